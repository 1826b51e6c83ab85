#!/bin/bash
# round 6: rocprofv3 kernel stats of summary.phase's workload (16384 games on 4096 slots, one group, 5x64, 400 sims/move), final library
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6phase
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_phase -- python $R/tools/phase_profile.py --games 16384 --groups 1 --filters 64 --sims 400 --reset-every 1 --waves 1024 > $R/gpurun_out/r6phase/phase.jsonl 2> $R/gpurun_out/r6phase/phase.err)
f=$(find /tmp/prof_phase -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r6phase/phase_kernel_stats.csv
tail -1 gpurun_out/r6phase/phase.jsonl; head -9 gpurun_out/r6phase/phase_kernel_stats.csv | cut -c1-230
