#!/bin/bash
# PMC passes of the round-2 kernels (separate runs, --kernel-trace only -- never combined with sys/hip/hsa traces):
# HBM traffic (FETCH_SIZE, WRITE_SIZE; KB units, FETCH doubled on gfx950 per MI355X_MICROARCH.md), MFMA pipe time, LDS
# bank conflicts.  Workload: bench.py's configuration with ONE slot group and AZHIP_TOWER=16, so that the kernel is the
# one of the timed two-group configuration (k_tower16, NT = 11) and every launch takes 4096 leaves.
cd "$(dirname "$0")/.."
ROOT=$PWD; TAG=${TAG:-r2pmc}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1)); d=/tmp/pmc_r2_$i; rm -rf $d
  (cd /tmp && AZHIP_TOWER=16 timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $ROOT/tools/run_config.py --game connect-four --slots 4096 --sims 400 --waves 120 > $OUT/pass${i}_stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/pass${i}_counters.csv
done
python3 $ROOT/tools/pmc_summary.py $OUT
