#!/bin/bash
# round 6: simulations per wave launch (AZHIP_RUN_K) with the background launch at its default 32
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6k
export TMPDIR=/tmp
for k in 3 4 5 6 3 4 5; do
  AZHIP_RUN_K=$k timeout 300 python bench.py --steps 2000 --warmup 50 --headline-only > gpurun_out/r6k/k_$k.json 2> gpurun_out/r6k/k_$k.err
  python - <<P
import json
d=json.load(open("gpurun_out/r6k/k_$k.json")); r=d["roofline"]
print("K $k: %.3f M sims/s, %.4f ms/step, %.3f sims/slot/wave, %.1f boards/launch, tower %.1f us" % (d["value"]/1e6, d["ms_per_step"], d["sims_per_slot_per_wave"], r["avg_boards_per_launch"], 1e3*r["avg_launch_ms"]))
P
done | tee gpurun_out/r6k/sweep.txt
