#!/bin/bash
# round 5, GPU call D (the session after the re-entry; ~13 min): the whole GPU suite incl. the round's new tests, the two k_tree
# experiments, the headline kernel ALONE under rocprofv3 (VERDICT r4 #2a), the evaluation cache on / off over a
# whole phase + the shipped network at the BASELINE batch, and three training iterations.
O=gpurun_out/r5d; mkdir -p $O; export TMPDIR=/tmp
{ nproc; free -g | head -2; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $O/host.txt 2>&1
# 1. every GPU test (no -x: all failures of this call are wanted)
timeout 780 python -m pytest tests -q -m gpu --durations=25 -p no:cacheprovider > $O/suite.log 2>&1
echo "suite rc $?" >> $O/suite.log; tail -45 $O/suite.log
[ -f gpurun_out/replay_all_games.jsonl ] && cp gpurun_out/replay_all_games.jsonl $O/
# 2. the driver's command (default line: headline, extras incl. whole phase / c2_5x128 / iteration, CPU baseline), evaluation cache on (default)
timeout 420 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
echo "bench rc $?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_default_line.json")); print(json.dumps(d["summary"])[:3000]); print({k: (v.get("error") if isinstance(v, dict) and "error" in v else "ok") for k, v in d["extra"].items()})
except Exception as ex:
    print("bench line unreadable:", ex); print(open("$O/bench_default.err").read()[-1500:])
PY
# 3. the headline's kernel alone (one slot group, forced to the two-group run's kernel), rocprofv3 kernel stats + the line's HIP events
d=/tmp/prof_h1; rm -rf $d
(cd /tmp && AZHIP_TOWER=16 AZHIP_EVAL_CACHE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --groups 1 --steps 400 --warmup 50 --headline-only > $GRAFT_REPO_ROOT/$O/headline_one_group_cache_off_line.json 2> $GRAFT_REPO_ROOT/$O/headline_one_group_cache_off.err)
f=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2); [ -n "$f" ] && cp $f $O/headline_one_group_cache_off_kernel_stats.csv && head -6 $f | cut -c1-200
d=/tmp/prof_h2; rm -rf $d
(cd /tmp && AZHIP_TOWER=16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --groups 1 --steps 400 --warmup 50 --headline-only > $GRAFT_REPO_ROOT/$O/headline_one_group_line.json 2> $GRAFT_REPO_ROOT/$O/headline_one_group.err)
f=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2); [ -n "$f" ] && cp $f $O/headline_one_group_kernel_stats.csv && head -6 $f | cut -c1-200
# 4. k_tree: read-modify-write backup vs no-return atomics (1: waited for, 2: not), slots in depth order
{ for knob in "" "AZHIP_TREE_ATOMIC=1" "AZHIP_TREE_ATOMIC=2" "AZHIP_TREE_SORT=1"; do
    echo "== ${knob:-baseline}"; env $knob TREE_BENCH_WARM=200 timeout 240 python tools/tree_bench.py 4096 1048576
  done; } > $O/ktree_experiments.txt 2>&1
cat $O/ktree_experiments.txt
# 5. the same phase with the evaluation cache off
AZHIP_EVAL_CACHE=0 AZ_BENCH_ONLY=whole_phase timeout 300 python bench.py --no-cpu-baseline --no-iteration > $O/bench_cache0.json 2> $O/bench_cache0.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_cache0.json")); r=d["roofline"]; w=d["extra"].get("whole_phase",{})
    print("cache=0 headline %.3f M sims/s ms/step %.3f unique %.3f frac %.3f kernel %s | alone frac %.3f avg ms %.4f | tree us/wave %s | phase %.3f M sims/s frac %.3f" % (
      d["value"]/1e6, d["ms_per_step"], d["unique_leaf_frac"], r["frac"], r["kernel"], d["roofline_kernel_alone"]["frac"], d["roofline_kernel_alone"]["avg_launch_ms"],
      d["roofline_tree"]["us_per_wave"], w.get("value",0)/1e6, w.get("roofline",{}).get("frac",-1)))
except Exception as ex:
    print("cache=0 failed:", ex); print(open("$O/bench_cache0.err").read()[-800:])
PY
# 6. does the loop learn at the shipped learning parameters?
timeout 240 python tools/iterations.py --iters 3 --games 512 --workers 512 > $O/iterations.jsonl 2> $O/iterations.err
echo "iterations rc $?"; tail -1 $O/iterations.jsonl | cut -c1-600
find /tmp/prof_h1 /tmp/prof_h2 -name "*.csv" -size +20M -delete 2>/dev/null
du -sh $O
