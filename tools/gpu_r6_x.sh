#!/bin/bash
# round 6, call X: free-running arena, background search off / wave K
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r6x
for cfg in "32 3" "0 3" "0 8" "0 16" "8 3" "8 8"; do
  set -- $cfg
  echo "kbg $1 k $2: $(AZHIP_RUN_KBG=$1 AZHIP_RUN_K=$2 AZHIP_ARENA_TRACE=1 timeout 300 python tools/arena_once.py 2>&1 | grep 'free-running arena\|first' | tr '\n' ' ')"
done | tee $OUT/sweep.txt
echo "lock: $(AZHIP_ARENA_FR=0 timeout 300 python tools/arena_once.py 2>&1 | tail -1)" | tee -a $OUT/sweep.txt
