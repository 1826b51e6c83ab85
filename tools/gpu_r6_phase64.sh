#!/bin/bash
# round 6: the first seconds of summary.phase's workload with every kernel class timed: what slows the tower down there (1.04 ms against 0.80)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6z
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python tools/phase_profile.py --games 16384 --groups 1 --filters 64 --sims 400 --reset-every 1 --waves 1024 --prof --max-seconds 4.2 > gpurun_out/r6z/early_$tag.jsonl 2> gpurun_out/r6z/early_$tag.err; echo "== $tag"; python - <<P
import json
for l in open("gpurun_out/r6z/early_$tag.jsonl"):
    r = json.loads(l)
    if "kernels_us_x_launches" in r: print(r["t"], r["boards_per_wave"], r["ms_per_wave"], r["msims_per_s"], {k: v[0] for k, v in r["kernels_us_x_launches"].items()})
P
}
run default A=1
run no_bg AZHIP_RUN_KBG=0
run bg8 AZHIP_RUN_KBG=8
run lock_step AZHIP_FREE_RUN=0
run no_prio AZHIP_BG_PRIO=3
