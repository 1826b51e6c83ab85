#!/bin/bash
# round 6, the round's evidence run: (A) the headline's timed region under rocprofv3 --kernel-trace --stats (the tower's average over
# exactly the timed launches against the line's HIP events: tools/trace_window.py), free-running (the default) and lock step;
# (B) the driver's command and the default command; (C) HBM counters (separate --pmc passes, --kernel-trace only) of the
# free-running wave's kernels.
O=gpurun_out/r6final; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T0=$SECONDS
prof_headline() { # tag, extra bench args, env...
  local tag=$1; shift; local args=$1; shift; local d=/tmp/prof_$tag; rm -rf $d
  (cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $R/bench.py --steps 400 --warmup 50 --headline-only $args > $R/$O/${tag}_line.json 2> $R/$O/${tag}.err)
  local st=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  local tr=$(find $d -name "*kernel_trace.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  [ -n "$st" ] && cp $st $O/${tag}_kernel_stats.csv
  [ -n "$tr" ] && python tools/trace_window.py $tr "k_tower" 400 $O/${tag}_timed_region_tower_dispatches.csv > $O/${tag}_timed_region_tower.json
  [ -n "$tr" ] && python tools/trace_window.py $tr "k_tree" 800 > $O/${tag}_timed_region_tree.json
  python - <<PY
import json
try:
    d=json.load(open("$O/${tag}_line.json")); r=d["roofline"]; w=json.load(open("$O/${tag}_timed_region_tower.json"))
    print("$tag: %.3f M sims/s, %.4f ms/step, %.2f sims/slot/wave, unique %.3f | HIP events: %s %d launches x %.1f boards, avg %.2f us, frac %.4f (over wall %.4f) | rocprofv3 trace, last %s: avg %.2f us (whole process %.2f us over %d)" % (
        d["value"]/1e6, d["ms_per_step"], d["sims_per_slot_per_wave"], d["unique_leaf_frac"], r["kernel"], r["launches"], r["avg_boards_per_launch"], 1e3*r["avg_launch_ms"], r["frac"], r["frac_over_wall"],
        w["window"], w["avg_us"], w["avg_us_whole_process"], w["dispatches_in_process"]))
except Exception as ex:
    print("$tag failed:", ex); print(open("$O/${tag}.err").read()[-600:])
PY
  rm -rf $d
}
prof_headline headline_free_running ""
prof_headline headline_lock_step "--lock-step"
echo "[$((SECONDS-T0)) s]"
# B. the driver's command, then the default command (2000 waves)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver.err
echo "driver-style bench rc $? [$((SECONDS-T0)) s]"
timeout 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_default_headline_line.json 2> $O/bench_default_headline.err
echo "default bench (no extras) rc $? [$((SECONDS-T0)) s]"
python - <<PY
import json
for f in ("bench_driver_line", "bench_default_headline_line"):
    try:
        d=json.load(open("$O/%s.json" % f))
        print(f, "value %.3f M" % (d["value"]/1e6), "frac", round(d["roofline"]["frac"],4), "over wall", round(d["roofline"]["frac_over_wall"],4), "sims/slot/wave", round(d["sims_per_slot_per_wave"],3),
              {k: round(d[k]["value"]) for k in ("value_long", "value_cache_off", "value_lock_step") if k in d})
        if "summary" in d: print(json.dumps(d["summary"])[:4000]); print({k: (v.get("error") if isinstance(v, dict) and "error" in v else "ok") for k, v in d["extra"].items()})
    except Exception as ex:
        print(f, "unreadable:", ex); print(open("$O/%s.err" % f.replace("_line","")).read()[-1500:])
PY
# C. HBM counters of the free-running wave's kernels
i=0
for ctrs in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1)); d=/tmp/fr_pmc_$i; rm -rf $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $R/tools/run_config.py --game connect-four --slots 4096 --sims 400 --waves 150 > $R/$O/f32_pass${i}_stdout.txt 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $O/f32_pass${i}_counters.csv
  rm -rf $d
done
python tools/pmc_summary.py $O > $O/pmc_summary_stdout.txt 2>&1; tail -8 $O/pmc_summary_stdout.txt
find $O -size +8M -delete; du -sh $O; echo "[$((SECONDS-T0)) s]"
