#!/bin/bash
# Round-4 evidence, one GPU call: (1) rocprofv3 --kernel-trace --stats of the driver's bench command (with the extra blocks),
# (2) PMC passes of the dominant kernels -- each counter set in its OWN run with --kernel-trace only (never combined with
# sys / hip / hsa traces): HBM traffic (FETCH_SIZE, WRITE_SIZE: KB, FETCH doubled on gfx950 per MI355X_MICROARCH.md), MFMA
# pipe time, LDS bank conflicts -- for the fp32 headline configuration (one slot group, AZHIP_TOWER=16: the kernel of the
# timed two-group configuration, 4096 leaves per launch) and for the bf16 10x128 configuration.
cd "$(dirname "$0")/.."
ROOT=$PWD; TAG=${TAG:-r4prof}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
d=/tmp/prof_bench; rm -rf $d
[ -z "$HEADLINE_ONLY" ] && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-iteration > $OUT/bench_line.json 2> $OUT/bench_stderr.txt)
f=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2); [ -n "$f" ] && cp $f $OUT/bench_kernel_stats.csv    # the bench process itself (the host-stepped example is a child with its own, smaller file)
# (1b) the timed region ALONE (--headline-only): the only k_tower16 launches of this process are the 2048-board launches of the two
# slot groups, so the trace's average duration is comparable with the line's roofline.avg_launch_ms (HIP events in the same run)
d=/tmp/prof_headline; rm -rf $d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/bench.py --steps 2000 --warmup 5 --headline-only > $OUT/bench_headline_line.json 2> $OUT/bench_headline_stderr.txt)
f=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2); [ -n "$f" ] && cp $f $OUT/bench_headline_kernel_stats.csv
[ -n "$HEADLINE_ONLY" ] && exit 0
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1)); d=/tmp/pmc_f32_$i; rm -rf $d
  (cd /tmp && AZHIP_TOWER=16 timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $ROOT/tools/run_config.py --game connect-four --slots 4096 --sims 400 --waves 120 > $OUT/f32_pass${i}_stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/f32_pass${i}_counters.csv
done
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1)); d=/tmp/pmc_bf16_$i; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $ROOT/tools/run_config.py --game connect-four --slots 4096 --sims 400 --waves 120 --blocks 10 --filters 128 --bf16 > $OUT/bf16_pass${i}_stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bf16_pass${i}_counters.csv
done
python3 $ROOT/tools/pmc_summary.py $OUT
