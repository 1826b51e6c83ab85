#!/bin/bash
# round 6, call I: how long should the background search run?  (move step on its own side stream)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6i
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --headline-only --steps 1000 --no-prof ${ARGS} > gpurun_out/r6i/$name.json 2> gpurun_out/r6i/$name.err
}
for k in "3 0" "3 2" "3 4" "3 8" "3 16" "3 32" "2 8" "4 8" "2 4"; do
  set -- $k
  ARGS="--groups 1" run g1_k$1_b$2 AZHIP_RUN_K=$1 AZHIP_RUN_KBG=$2
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6i/*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], "sims/step %.0f" % (d["value"]*d["ms_per_step"]/1e3))
    except Exception as ex:
        print(f, "unreadable:", ex)
P
