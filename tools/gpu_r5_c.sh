#!/bin/bash
# round 5, GPU call C: replay loop after the oracle's node-pool trees; whole-phase throughput vs cache size and slot groups; kernel stats of a phase
mkdir -p gpurun_out; rm -f gpurun_out/replay_all_games.jsonl
timeout 600 python -m pytest "tests/test_replay_all_games_gpu.py::test_config2_every_one_of_the_4096_games" tests/test_split_fallback_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/c_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/c_tests.log; tail -5 gpurun_out/c_tests.log; cat gpurun_out/replay_all_games.jsonl
run() { # name env...
  local name=$1; shift
  env "$@" AZ_BENCH_ONLY=whole_phase timeout 600 python bench.py --no-cpu-baseline --no-iteration $EXTRA > gpurun_out/c_$name.json 2> gpurun_out/c_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/c_$name.json"))
    r=d["roofline"]; w=d["extra"].get("whole_phase",{})
    print("$name: headline %.3f M sims/s ms/step %.3f unique %.3f towerfrac %.3f | tree us/wave %s | whole_phase %.3f M sims/s unique %.3f frac %.3f avg boards/launch %.0f avg launch ms %.3f" % (
      d["value"]/1e6, d["ms_per_step"], d.get("unique_leaf_frac",-1), r["frac"], d.get("roofline_tree",{}).get("us_per_wave"),
      w["value"]/1e6, w["unique_leaf_frac"], w["roofline"]["frac"], w["roofline"]["avg_boards_per_launch"], w["roofline"]["avg_launch_ms"]))
except Exception as ex:
    print("$name: failed", ex); print(open("gpurun_out/c_$name.err").read()[-600:])
PY
}
run log24_g2 AZHIP_EVAL_CACHE_LOG2=24
run log27_g2 AZHIP_EVAL_CACHE_LOG2=27
EXTRA="--groups 4" run log27_g4 AZHIP_EVAL_CACHE_LOG2=27
EXTRA="--groups 1" run log27_g1 AZHIP_EVAL_CACHE_LOG2=27
cd /tmp && export TMPDIR=/tmp
AZHIP_EVAL_CACHE_LOG2=27 AZ_BENCH_ONLY=whole_phase timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c_prof -o phase -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-iteration --no-prof > $GRAFT_REPO_ROOT/gpurun_out/c_prof_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/c_prof.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/c_prof -name "*kernel_stats.csv" | head -1); echo $f; head -12 $f | cut -c1-220
find gpurun_out/c_prof -name "*kernel_trace.csv" -delete; find gpurun_out/c_prof -name "*.db" -delete
