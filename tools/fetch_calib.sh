#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for small random accesses (tools/probes/fetch_calib.hip), separate PMC passes.
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/fetch_calib; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=/tmp/fc_$(echo $c | tr ' ' '_'); rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $ROOT/tools/probes/fetch_calib > $OUT/stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"]); acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-28s %-24s per launch %16.1f" % (k, c, s / n))
PY
done
grep asked $OUT/stdout.txt
