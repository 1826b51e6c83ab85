#!/bin/bash
# round 6, call G (background search until the tower has run): one slot group, wave launch + tower + heads on one stream, move step + background search on a side stream
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6h
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_free_running_gpu.py tests/test_tree_gpu.py tests/test_eval_cache_gpu.py tests/test_split_fallback_gpu.py -x -q -m gpu > gpurun_out/r6h/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r6h/tests.log
tail -3 gpurun_out/r6h/tests.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --headline-only --steps 600 --prof-all ${ARGS} > gpurun_out/r6h/$name.json 2> gpurun_out/r6h/$name.err
}
for k in "1 256" "2 256" "3 256" "4 256"; do
  set -- $k
  ARGS="--groups 1" run g1_k$1_b$2 AZHIP_RUN_K=$1 AZHIP_RUN_KBG=$2
done
ARGS="--groups 1 --no-prof" run g1_default_noprof
ARGS="--groups 2" run g2_default
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6h/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        km=d.get("kernel_ms",{})
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], "boards/launch %.0f" % r.get("avg_boards_per_launch",0), "tower launch ms %.3f" % r.get("avg_launch_ms",0), r.get("kernel"), {k: round(v/d["steps"],4) for k,v in km.items()})
    except Exception as ex:
        print(f, "unreadable:", ex)
P
