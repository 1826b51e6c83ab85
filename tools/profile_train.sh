#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r1f; mkdir -p $OUT; export TMPDIR=/tmp
for f in 128 64; do
  d=/tmp/prof_train_$f; rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/tools/train_bench.py --filters $f --steps 30 > $OUT/train_${f}_stdout.txt 2>&1)
  cp $(find $d -name '*kernel_stats.csv' | head -1) $OUT/train_${f}_kernel_stats.csv
done
