#!/bin/bash
# PMC passes (separate runs, kernel-trace only -- never combined with sys/hip/hsa traces) over the optimiser step
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_train; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=/tmp/pmc_train_$i; rm -rf $d
  (cd /tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $ROOT/tools/train_bench.py --filters 128 --steps 6 --games 4096 > $OUT/pass${i}_stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/pass${i}_counters.csv
done
python3 - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_train/pass*_counters.csv")):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Kernel Name")
        if not name or not any(k in name for k in ("k_conv16_layer", "k_wgrad16", "k_tr_", "k_wgrad_reduce")):
            continue
        acc[name.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"_dispatches": max(len(v) for v in d.values())} for k, d in acc.items()}
json.dump(out, open("gpurun_out/pmc_train/summary.json", "w"), indent=1)
for k, d in out.items():
    print(k[:70], {c: round(v, 1) for c, v in d.items()})
PY
