#!/bin/bash
# round 5, GPU call A: replay-mode parity over every game + forced rare tree events + the host's size
mkdir -p gpurun_out
{ nproc; free -g | head -2; } > gpurun_out/host.txt 2>&1
timeout 1500 python -m pytest tests/test_replay_all_games_gpu.py tests/test_tree_stress_gpu.py "tests/test_train_gpu.py::test_column_sums_finished_inside_the_producer_do_not_change_the_values" -q -m gpu --durations=0 -p no:cacheprovider > gpurun_out/replay_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/replay_tests.log
tail -40 gpurun_out/replay_tests.log
cat gpurun_out/host.txt gpurun_out/replay_all_games.jsonl
