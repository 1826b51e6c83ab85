#!/bin/bash
# round 6: the paired tower capped at 176 VGPRs, TWO slot groups (a k_tree wavefront now fits beside a tower workgroup's waves on a SIMD)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6vgpr
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 2000 --warmup 50 --headline-only --groups 2 > gpurun_out/r6vgpr/g2_$tag.json 2> gpurun_out/r6vgpr/g2_$tag.err
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6vgpr/g2_$tag.json")); r=d["roofline"]
    print("2 groups $tag: %.3f M sims/s, %.4f ms/step, %.3f sims/slot/wave, %s %.1f boards/launch, tower %.1f us" % (d["value"]/1e6, d["ms_per_step"], d["sims_per_slot_per_wave"], r["kernel"], r["avg_boards_per_launch"], 1e3*r["avg_launch_ms"]))
except Exception as ex: print("$tag failed", ex)
P
}
run default A=1
run tower21 AZHIP_TOWER=21
run tower21_k8 AZHIP_TOWER=21 AZHIP_RUN_K=8
run tower21_k8_bg0 AZHIP_TOWER=21 AZHIP_RUN_K=8 AZHIP_RUN_KBG=0
run tower21_lock AZHIP_TOWER=21 AZHIP_FREE_RUN=0
run tower16 AZHIP_TOWER=16
