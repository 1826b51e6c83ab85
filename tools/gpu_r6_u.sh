#!/bin/bash
# round 6, call U: MFMA / LDS counters of the free-running headline's kernels in the steady state (separate --pmc passes, --kernel-trace only)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6u
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1)); d=/tmp/fr_pmc2_$i; rm -rf $d
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $R/bench.py --headline-only --no-prof --steps 100 --warmup 5 > $R/gpurun_out/r6u/pass${i}_line.json 2> $R/gpurun_out/r6u/pass${i}.err)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - <<PY
import csv, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$f")):
    n = (r.get("Kernel_Name") or "").replace("void ", "").split("(")[0]
    if any(k in n for k in ("k_tower16x2", "k_tree", "k_heads16", "k_move_fr")):
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {n: {c: sum(v[-200:]) / len(v[-200:]) for c, v in d.items()} for n, d in acc.items()}
json.dump(out, open("gpurun_out/r6u/pmc_pass$i.json", "w"), indent=1); print(json.dumps(out)[:1500])
PY
  rm -rf $d
done
