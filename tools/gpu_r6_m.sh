#!/bin/bash
# round 6, call M: why is the two-round tower slow?  per-kernel durations (rocprofv3 --kernel-trace --stats)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6m
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for tag in rounds_on rounds_off force19; do
  d=/tmp/prof_$tag; rm -rf $d
  envs=""; [ $tag = rounds_off ] && envs="AZHIP_TOWER_ROUNDS=0"; [ $tag = force19 ] && envs="AZHIP_TOWER=19"
  (cd /tmp && env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/bench.py --steps 300 --warmup 50 --headline-only --no-prof > $R/gpurun_out/r6m/${tag}_line.json 2> $R/gpurun_out/r6m/${tag}.err)
  tr=$(find $d -name "*kernel_trace.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$tr")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-3000:]
acc = collections.defaultdict(list)
for r in last:
    acc[r["Kernel_Name"].replace("void ", "").split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$tag")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("   %-72s n %5d avg %8.1f us  total %8.1f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
PY
  rm -rf $d
done 2>&1 | tee gpurun_out/r6m/kernels.txt
