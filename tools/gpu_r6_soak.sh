#!/bin/bash
# round 6: the free-running schedule's order of events differs from run to run -- the parity tests that exercise it, several times over
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6soak
export TMPDIR=/tmp
for i in ${SOAK_RUNS:-1 2 3 4 5 6}; do
  SECONDS=0
  timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_replay_all_games_gpu.py --deselect tests/test_baseline_configs_gpu.py --deselect tests/test_bench_multirank_gpu.py > gpurun_out/r6soak/run_$i.log 2>&1
  echo "run $i rc $? seconds $SECONDS: $(tail -1 gpurun_out/r6soak/run_$i.log)" | tee -a gpurun_out/r6soak/summary.txt
done
timeout 900 python -m pytest tests/test_replay_all_games_gpu.py -q -m gpu -x > gpurun_out/r6soak/replay_again.log 2>&1; echo "replay again rc $?: $(tail -1 gpurun_out/r6soak/replay_again.log)" | tee -a gpurun_out/r6soak/summary.txt
