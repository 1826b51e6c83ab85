#!/bin/bash
# round 6, call K2: replay threads within the cgroup's CPU quota; explore! that runs ahead (call J's content)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6k
export TMPDIR=/tmp
for t in 16 32; do
  SECONDS=0
  AZ_REPLAY_THREADS=$t timeout 500 python -m pytest "tests/test_replay_all_games_gpu.py::test_config2_every_one_of_the_4096_games" -x -q -m gpu -s > gpurun_out/r6k/replay_c2_t$t.log 2>&1
  echo "threads $t rc $? seconds $SECONDS" | tee -a gpurun_out/r6k/replay_times.txt
  grep -o '"seconds_device_phase[^}]*}' gpurun_out/r6k/replay_c2_t$t.log | tail -1
done
bash tools/gpu_r6_j.sh
