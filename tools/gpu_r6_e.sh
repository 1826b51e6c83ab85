#!/bin/bash
# round 6, call E: background launches over the compacted list of needy slots -- parity tests, then the K sweep again
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6e
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_free_running_gpu.py tests/test_tree_gpu.py tests/test_eval_cache_gpu.py tests/test_params_gpu.py tests/test_tree_stress_gpu.py -x -q -m gpu > gpurun_out/r6e/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r6e/tests.log
tail -3 gpurun_out/r6e/tests.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --headline-only --steps 600 --prof-all ${ARGS} > gpurun_out/r6e/$name.json 2> gpurun_out/r6e/$name.err
}
for g in 1 2; do
  for k in "1 16" "2 16" "2 64" "4 32" "4 8" "8 32"; do
    set -- $k
    ARGS="--groups $g" run g${g}_k$1_b$2 AZHIP_RUN_K=$1 AZHIP_RUN_KBG=$2
  done
done
ARGS="--groups 2" run g2_k2_b64_x2 AZHIP_RUN_K=2 AZHIP_RUN_KBG=64 AZHIP_TOWER=21
ARGS="--groups 2" run g2_k4_b32_x2 AZHIP_RUN_K=4 AZHIP_RUN_KBG=32 AZHIP_TOWER=21
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6e/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        km=d.get("kernel_ms",{})
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], "boards/launch %.0f" % r.get("avg_boards_per_launch",0), "tower launch ms %.3f" % r.get("avg_launch_ms",0), r.get("kernel"), {k: round(v/d["steps"],4) for k,v in km.items()})
    except Exception as ex:
        print(f, "unreadable:", ex)
P
