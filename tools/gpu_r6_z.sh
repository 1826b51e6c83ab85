#!/bin/bash
# round 6, call Z: time series of the iteration's self-play phase (where the rate is lost), then call Y's 512-game leg
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6z
export TMPDIR=/tmp
timeout 300 python tools/phase_profile.py > gpurun_out/r6z/phase_5000_on_4096x2.jsonl 2> gpurun_out/r6z/phase.err; tail -1 gpurun_out/r6z/phase_5000_on_4096x2.jsonl
timeout 300 python tools/phase_profile.py --games 8192 > gpurun_out/r6z/phase_8192_on_4096x2.jsonl 2>> gpurun_out/r6z/phase.err; tail -1 gpurun_out/r6z/phase_8192_on_4096x2.jsonl
bash tools/gpu_r6_y.sh
