#!/bin/bash
# Optimiser-step evidence in one GPU call: the trainer's parity tests, tools/train_bench.py under rocprofv3 --kernel-trace --stats,
# and the same bench without the profiler.  Output: gpurun_out/$TAG/{tests.txt,bench_rocprof.txt,bench.txt,kernel_stats.csv}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${TAG:-train}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || (cd $ROOT && timeout 300 python -m pytest tests/test_train_gpu.py -x -q > $OUT/tests.txt 2>&1 < /dev/null; tail -2 $OUT/tests.txt)
d=/tmp/prof_$TAG
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $ROOT/tools/train_bench.py --steps 30 ${BENCH_ARGS} > $OUT/bench_rocprof.txt 2>&1 < /dev/null)
f=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" 2>/dev/null | sort -n | tail -1 | cut -d" " -f2)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; head -14 $OUT/kernel_stats.csv | cut -c1-160; fi
(cd $ROOT && timeout 120 python tools/train_bench.py --steps 50 ${BENCH_ARGS} 2>&1 < /dev/null | grep -v "^[WE]2026" > $OUT/bench.txt; cat $OUT/bench.txt)
# one step's timeline (the last 170 dispatches of the profiled run): start and end relative to the first, per kernel
k=$(find $d -name "*kernel_trace.csv" -printf "%s %p\n" 2>/dev/null | sort -n | tail -1 | cut -d" " -f2)
if [ -n "$k" ]; then python - "$k" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_t" in r["Kernel_Name"] or "k_conv" in r["Kernel_Name"] or "k_wgrad" in r["Kernel_Name"] or "k_gemm" in r["Kernel_Name"]][-340:-170]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    print("%-40s q%-3s %9.1f %9.1f  %7.1f" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
fi
# PMC=1: one counter pass (its own run, --kernel-trace only): MFMA pipe time of the trainer's kernels (the profiler serialises
# the two streams, so these are the kernels alone)
if [ -n "$PMC" ]; then
  p=/tmp/pmc_$TAG; rm -rf $p
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $p -- python $ROOT/tools/train_bench.py --steps 10 ${BENCH_ARGS} > $OUT/pmc_stdout.txt 2>&1 < /dev/null)
  c=$(find $p -name '*counter_collection.csv' 2>/dev/null | head -1)
  if [ -n "$c" ]; then python - "$c" > $OUT/pmc_mfma_busy.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if v.get("GRBM_GUI_ACTIVE", 0) > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES sums 1024 SIMDs, GRBM_GUI_ACTIVE 8 XCDs
        print("%-50s launches %5d  %8.1f us  MFMA pipes busy %.3f of the kernel's cycles" % (k[:50], n[(k, "GRBM_GUI_ACTIVE")],
              v["GRBM_GUI_ACTIVE"] / 8 / n[(k, "GRBM_GUI_ACTIVE")] / 2100.0, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * v["GRBM_GUI_ACTIVE"])))
PY
  head -8 $OUT/pmc_mfma_busy.txt; fi
fi
