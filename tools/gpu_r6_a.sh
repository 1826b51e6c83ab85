#!/bin/bash
# round 6, call A: the free-running schedule's first contact with the device -- its own tests, the tree / cache / parameter parity
# tests that now run on it by default, then the headline in both group counts.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_free_running_gpu.py tests/test_tree_gpu.py tests/test_eval_cache_gpu.py tests/test_params_gpu.py tests/test_edge_inputs_gpu.py tests/test_errors_gpu.py tests/test_capacity_gpu.py -x -q -m gpu > gpurun_out/r6a/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r6a/tests.log
tail -5 gpurun_out/r6a/tests.log
for g in 2 1; do
  timeout 600 python bench.py --headline-only --steps 1000 --groups $g > gpurun_out/r6a/headline_g$g.json 2> gpurun_out/r6a/headline_g$g.err
  AZHIP_FREE_RUN=0 timeout 600 python bench.py --headline-only --steps 1000 --groups $g > gpurun_out/r6a/headline_lock_g$g.json 2> gpurun_out/r6a/headline_lock_g$g.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6a/headline*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        print(f, "%.3f M sims/s" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], "frac %.3f" % r.get("frac",0), "boards/launch %.0f" % r.get("avg_boards_per_launch",0), "unique %.3f" % d["unique_leaf_frac"], r.get("kernel"))
    except Exception as ex:
        print(f, "unreadable:", ex)
P
