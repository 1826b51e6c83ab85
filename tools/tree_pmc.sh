#!/bin/bash
# PMC passes of k_tree alone (tools/tree_wave.py, hash oracle): where a wavefront's cycles go (waiting on memory, waiting to
# issue, issuing), instruction mix, instruction-cache behaviour, HBM traffic.  Separate passes, --kernel-trace only.
cd "$(dirname "$0")/.."
ROOT=$PWD; TAG=${TAG:-tree_pmc}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
SLOTS=${SLOTS:-4096}
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_IFETCH SQ_INST_CYCLES_VMEM_RD" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); d=/tmp/tree_pmc_$i; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $ROOT/tools/tree_wave.py --slots $SLOTS --waves 100 > $OUT/pass${i}_stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/pass${i}_counters.csv
  tail -2 $OUT/pass${i}_stdout.txt
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pass*_counters.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_tree" not in k: continue
        c = r["Counter_Name"]; acc[c][0] += float(r["Counter_Value"]); acc[c][1] += 1
    for c, (s, n) in acc.items():
        print("%-32s per launch %14.1f  (%d launches)" % (c, s / max(n, 1), n))
PY
