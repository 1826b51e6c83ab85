#!/bin/bash
# round 5, GPU call E (the last one, ~8 min): the round's final bench.py (one slot group by default) -- (A) its timed region under
# rocprofv3 with the kernel trace kept, so that the tower's average over exactly the timed launches can be set against the line's HIP
# events (tools/trace_window.py), evaluation cache on (the default) and off; (B) the driver's command itself; (C) the same command
# with a subset of the extras under rocprofv3 --kernel-trace --stats; (D) the tests that run bench.py; (E) HBM counters of k_tree with
# the evaluation cache's probe / fill (separate --pmc passes, --kernel-trace only).
O=gpurun_out/r5e; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T0=$SECONDS
prof_headline() { # tag, env...
  local tag=$1; shift; local d=/tmp/prof_$tag; rm -rf $d
  (cd /tmp && env "$@" timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/bench.py --steps 400 --warmup 50 --headline-only > $R/$O/${tag}_line.json 2> $R/$O/${tag}.err)
  local st=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  local tr=$(find $d -name "*kernel_trace.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  [ -n "$st" ] && cp $st $O/${tag}_kernel_stats.csv
  [ -n "$tr" ] && python tools/trace_window.py $tr "k_tower" 400 $O/${tag}_timed_region_tower_dispatches.csv > $O/${tag}_timed_region_tower.json
  python - <<PY
import json
try:
    d=json.load(open("$O/${tag}_line.json")); r=d["roofline"]; w=json.load(open("$O/${tag}_timed_region_tower.json"))
    print("$tag: %.3f M sims/s, %.4f ms/step, unique %.3f | HIP events: %s %d launches x %.1f boards, avg %.2f us, frac %.4f (over wall %.4f) | rocprofv3 trace, last %s: avg %.2f us (whole process %.2f us over %d)" % (
        d["value"]/1e6, d["ms_per_step"], d["unique_leaf_frac"], r["kernel"], r["launches"], r["avg_boards_per_launch"], 1e3*r["avg_launch_ms"], r["frac"], r["frac_over_wall"],
        w["window"], w["avg_us"], w["avg_us_whole_process"], w["dispatches_in_process"]))
except Exception as ex:
    print("$tag failed:", ex); print(open("$O/${tag}.err").read()[-600:])
PY
}
prof_headline headline_cache_on
prof_headline headline_cache_off AZHIP_EVAL_CACHE=0
echo "[$((SECONDS-T0)) s]"
# B. the driver's command
timeout 420 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
echo "bench rc $? [$((SECONDS-T0)) s]"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_default_line.json")); print(json.dumps(d["summary"])[:3000]); print({k: (v.get("error") if isinstance(v, dict) and "error" in v else "ok") for k, v in d["extra"].items()})
except Exception as ex:
    print("bench line unreadable:", ex); print(open("$O/bench_default.err").read()[-1500:])
PY
# C. kernel stats of the bench command (a subset of the extras: the whole line's trace would be gigabytes)
d=/tmp/prof_bench; rm -rf $d
(cd /tmp && AZ_BENCH_ONLY=whole_phase,c2_5x128,workers_128_5x128 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/bench.py --no-cpu-baseline --no-iteration > $R/$O/bench_under_rocprof_line.json 2> $R/$O/bench_under_rocprof.err)
f=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -8 $f | cut -c1-160
rm -rf $d; echo "[$((SECONDS-T0)) s]"
# D. the tests that run bench.py / the C demo (bench.py's defaults changed after call D's full-suite run)
if [ $((SECONDS-T0)) -lt 520 ]; then
  timeout 120 python -m pytest tests/test_bench_multirank_gpu.py tests/test_c_demo.py -q -m gpu -p no:cacheprovider > $O/bench_tests.log 2>&1; echo "tests rc $?" >> $O/bench_tests.log; tail -3 $O/bench_tests.log
fi
# E. k_tree's HBM traffic with the evaluation cache in the kernel (hash oracle, cache forced on)
if [ $((SECONDS-T0)) -lt 560 ]; then
  i=0
  for ctrs in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1)); d=/tmp/tree_pmc_$i; rm -rf $d
    (cd /tmp && AZHIP_EVAL_CACHE=1 timeout 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $R/tools/tree_wave.py --slots 4096 --waves 100 > $R/$O/tree_cache_pass${i}_stdout.txt 2>&1)
    f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/tree_cache_pass${i}_counters.csv
  done
  python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/tree_cache_pass*_counters.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tree" in r.get("Kernel_Name", ""): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items(): print("k_tree (cache on) %-12s per launch %.1f KB (second half of %d launches)" % (c, sum(v[len(v)//2:]) / max(len(v[len(v)//2:]), 1), len(v)))
PY
fi
find $O -size +8M -delete; du -sh $O; echo "[$((SECONDS-T0)) s]"
