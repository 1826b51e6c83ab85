#!/bin/bash
# round 6: k_tower16<.,128,NT=11> (222 registers) within 176: the 5x128 configurations (two slot groups: the other group's tree kernels beside the tower)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6vgpr3
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_net.py -x -q -m gpu 2>&1 | tail -1
timeout 300 python tools/phase_profile.py > gpurun_out/r6vgpr3/phase_5000.jsonl 2> gpurun_out/r6vgpr3/phase.err; tail -1 gpurun_out/r6vgpr3/phase_5000.jsonl
timeout 300 python tools/run_config.py --game connect-four --slots 4096 --groups 2 --filters 128 --sims 600 --waves 3000 2>&1 | tail -3
