#!/bin/bash
# round 5, GPU call I (the round's last minute): HBM counters of the headline's tower kernel, k_tower16x2, on full 4096-board launches
# (evaluation cache off, lock-step first move: every launch holds every leaf) -- separate --pmc passes, --kernel-trace only
O=gpurun_out/r5i; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; i=0
for ctrs in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1)); d=/tmp/pmc_x2_$i; rm -rf $d
  (cd /tmp && AZHIP_EVAL_CACHE=0 timeout 22 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $R/tools/run_config.py --game connect-four --slots 4096 --sims 400 --waves 100 > $R/$O/f32x2_pass${i}_stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/f32x2_pass${i}_counters.csv
  tail -1 $O/f32x2_pass${i}_stdout.txt | cut -c1-200
done
ls -la $O
