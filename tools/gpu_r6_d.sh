#!/bin/bash
# round 6, call D: do the background launches slow the tower because they win issue arbitration?  priority 0 vs 3
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6d
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --headline-only --steps 600 --prof-all ${ARGS} > gpurun_out/r6d/$name.json 2> gpurun_out/r6d/$name.err
}
for g in 1 2; do
  for k in "2 16" "2 32" "4 8" "1 32"; do
    set -- $k
    ARGS="--groups $g" run g${g}_k$1_b$2_p0 AZHIP_RUN_K=$1 AZHIP_RUN_KBG=$2
  done
  ARGS="--groups $g" run g${g}_k2_b16_p3 AZHIP_RUN_K=2 AZHIP_RUN_KBG=16 AZHIP_BG_PRIO=3
done
ARGS="--groups 2" run g2_k2_b16_p0_x2 AZHIP_RUN_K=2 AZHIP_RUN_KBG=16 AZHIP_TOWER=21
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6d/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        km=d.get("kernel_ms",{})
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], "boards/launch %.0f" % r.get("avg_boards_per_launch",0), "tower launch ms %.3f" % r.get("avg_launch_ms",0), r.get("kernel"), {k: round(v/d["steps"],4) for k,v in km.items()})
    except Exception as ex:
        print(f, "unreadable:", ex)
P
