#!/bin/bash
# rocprofv3 kernel stats of the optimiser step (tools/train_bench.py) at 5x64 and 5x128; outputs under gpurun_out/r4_train/
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4_train; mkdir -p $OUT; export TMPDIR=/tmp
for f in ${FILTERS:-64 128}; do
  d=/tmp/prof_train_$f; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/tools/train_bench.py --filters $f --steps 40 > $OUT/train_${f}_stdout.txt 2>&1)
  cp $(find $d -name '*kernel_stats.csv' | head -1) $OUT/train_${f}_kernel_stats.csv
  tail -2 $OUT/train_${f}_stdout.txt
done
