#!/bin/bash
# rocprofv3 kernel stats of the replay-memory pipeline (tools/memory_bench.py); output under gpurun_out/r4_memory/
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4_memory; mkdir -p $OUT; export TMPDIR=/tmp
d=/tmp/prof_memory; rm -rf $d
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/tools/memory_bench.py --games 16384 > $OUT/stdout.txt 2>&1)
cp $(find $d -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep -v "^[WE]2026" $OUT/stdout.txt | tail -6
