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
