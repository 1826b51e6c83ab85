#!/bin/bash
# round 6, call T: what a free-running wave spends where in the other configurations (every kernel class timed)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6t
export TMPDIR=/tmp
(timeout 300 python tools/run_config.py --game mancala --slots 8192 --sims 800 --waves 6000 --prof
 timeout 300 python tools/run_config.py --game connect-four --slots 128 --sims 600 --filters 128 --waves 3000 --prof
 timeout 300 python tools/run_config.py --game connect-four --slots 4096 --sims 600 --filters 128 --groups 2 --waves 1500 --prof
 timeout 300 python tools/run_config.py --game connect-four --slots 4096 --sims 400 --blocks 10 --filters 128 --bf16 --groups 2 --waves 3000 --prof) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6t/classes.txt
