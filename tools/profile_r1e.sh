#!/bin/bash
# rocprofv3 kernel-trace summaries of the final round-1 kernels -> gpurun_out/r1e/ (copied to profiles/r1/ by hand)
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r1e
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, command...
  name=$1; shift
  d=/tmp/prof_$name
  rm -rf $d
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- "$@" > $OUT/${name}_stdout.txt 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
}
run bench_groups2 python $ROOT/bench.py --steps 400 --warmup 100 --no-cpu-baseline
run small_128w_5x128 python $ROOT/tools/run_config.py --game connect-four --slots 128 --filters 128 --sims 600 --waves 1200
run memory_1M python $ROOT/tools/memory_bench.py --games 65536
run arena_128 python $ROOT/tools/arena_bench.py --games 128 --workers 128 --sims 200
tail -n 3 $OUT/*_stdout.txt
