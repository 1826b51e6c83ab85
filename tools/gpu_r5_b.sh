#!/bin/bash
# round 5, GPU call B: the evaluation cache (correctness, what it costs k_tree, what it gains) + the in-C replay loop's speed
mkdir -p gpurun_out; rm -f gpurun_out/replay_all_games.jsonl
timeout 900 python -m pytest tests/test_eval_cache_gpu.py tests/test_tree_stress_gpu.py "tests/test_replay_all_games_gpu.py::test_config2_every_one_of_the_4096_games" -q -m gpu --durations=5 -p no:cacheprovider > gpurun_out/b_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/b_tests.log
tail -25 gpurun_out/b_tests.log; cat gpurun_out/replay_all_games.jsonl
{ echo "== k_tree alone, cache off"; AZHIP_EVAL_CACHE=0 timeout 300 python tools/tree_bench.py 4096 65536
  echo "== k_tree alone, cache on (2^24 entries)"; AZHIP_EVAL_CACHE=1 timeout 300 python tools/tree_bench.py 4096 65536
  echo "== k_tree alone, cache on (2^16 entries)"; AZHIP_EVAL_CACHE=1 AZHIP_EVAL_CACHE_LOG2=16 timeout 300 python tools/tree_bench.py 4096 ; } > gpurun_out/b_tree_bench.txt 2>&1
cat gpurun_out/b_tree_bench.txt
for c in 0 1; do
  AZHIP_EVAL_CACHE=$c AZ_BENCH_ONLY=whole_phase,c3 timeout 600 python bench.py --no-cpu-baseline --no-iteration > gpurun_out/b_bench_cache$c.json 2> gpurun_out/b_bench_cache$c.err
  python - <<PY
import json
d=json.load(open("gpurun_out/b_bench_cache$c.json"))
r=d["roofline"]; w=d["extra"].get("whole_phase",{}); c3=d["extra"].get("c3",{})
print("cache=$c headline %.3f M sims/s ms/step %.3f unique %.3f frac %.3f | alone frac %.3f | tree us/wave %.1f | whole_phase %s M sims/s unique %s frac %s | c3 %s unique %s" % (
  d["value"]/1e6, d["ms_per_step"], d.get("unique_leaf_frac",-1), r["frac"], d.get("roofline_kernel_alone",{}).get("frac",-1), d.get("roofline_tree",{}).get("us_per_wave",-1),
  w.get("value",0)/1e6 if "value" in w else w, w.get("unique_leaf_frac"), w.get("roofline",{}).get("frac"), c3.get("value"), c3.get("unique_leaf_frac")))
PY
done
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider --deselect tests/test_replay_all_games_gpu.py --deselect tests/test_eval_cache_gpu.py --deselect tests/test_tree_stress_gpu.py > gpurun_out/b_suite.log 2>&1
echo "suite rc $?" >> gpurun_out/b_suite.log; tail -8 gpurun_out/b_suite.log
