#!/bin/bash
# round 6, call B: where a free-running wave's time goes -- every kernel class timed (HIP events), slot groups x simulations per launch
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
run() {  # name, env..., -- args
  local name=$1; shift
  env "$@" timeout 600 python bench.py --headline-only --steps 600 --prof-all ${ARGS} > gpurun_out/r6b/$name.json 2> gpurun_out/r6b/$name.err
}
for g in 1 2 4; do
  for k in 8 32; do
    ARGS="--groups $g" run g${g}_k${k} AZHIP_RUN_K=$k
  done
done
ARGS="--groups 1" run g1_lock AZHIP_FREE_RUN=0
ARGS="--groups 2" run g2_lock AZHIP_FREE_RUN=0
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6b/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        km=d.get("kernel_ms",{})
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], "boards/launch %.0f" % r.get("avg_boards_per_launch",0), "tower launch ms %.3f" % r.get("avg_launch_ms",0), r.get("kernel"), {k: round(v/d["steps"],4) for k,v in km.items()})
    except Exception as ex:
        print(f, "unreadable:", ex)
P
