#!/bin/bash
# round 6, call Y: VERDICT r5 #7's 512-game leg, once: 512 whole games of the headline configuration's phase (4096 slots, 400 sims,
# ResNet 5x64 in the loop, free-running schedule) against the oracle playing them with its OWN fp32 network on the host's CPUs
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6y
export TMPDIR=/tmp
SECONDS=0
AZ_NSAMPLE_C2=512 timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py -q -m gpu -k config2 -s > gpurun_out/r6y/c2_512.log 2>&1
echo "rc $? seconds $SECONDS" >> gpurun_out/r6y/c2_512.log
tail -5 gpurun_out/r6y/c2_512.log
