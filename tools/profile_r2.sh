#!/bin/bash
# rocprofv3 kernel-trace summaries of the round-2 kernels -> gpurun_out/$TAG/ (copied to profiles/r2/ by hand)
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=${TAG:-r2p}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, command...
  name=$1; shift
  d=/tmp/prof_$name
  rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- "$@" > $OUT/${name}_stdout.txt 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
}
for what in ${WHAT:-bench_groups2 tower16 small_128w_5x128}; do
  case $what in
    bench_groups2) run $what python $ROOT/bench.py --steps 400 --warmup 100 --no-cpu-baseline ;;
    bench_groups1) run $what python $ROOT/bench.py --groups 1 --steps 400 --warmup 100 --no-cpu-baseline ;;
    tower16) AZHIP_TOWER=16 run $what python $ROOT/tools/run_config.py --game connect-four --slots 4096 --sims 400 --waves 600 ;;
    small_128w_5x128) run $what python $ROOT/tools/run_config.py --game connect-four --slots 128 --filters 128 --sims 600 --waves 1200 ;;
    small_128w_5x128_g2) run $what python $ROOT/tools/run_config.py --game connect-four --slots 128 --filters 128 --groups 2 --sims 600 --waves 1200 ;;
    tree) timeout 300 python $ROOT/tools/tree_bench.py > $OUT/tree_bench.txt 2>&1 ;;
  esac
done
tail -n 2 $OUT/*_stdout.txt
