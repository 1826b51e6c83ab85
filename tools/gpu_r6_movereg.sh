#!/bin/bash
# round 6: k_move_fr within 96 registers (does a busy move-step workgroup keep a tower workgroup off its CU?): steady state K = 3 / 4, the phase
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6mv
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_free_running_gpu.py -x -q -m gpu 2>&1 | tail -1
for k in 3 4 3 4; do
  AZHIP_RUN_K=$k timeout 300 python bench.py --steps 2000 --warmup 50 --headline-only > gpurun_out/r6mv/k_$k.json 2> gpurun_out/r6mv/k_$k.err
  python - <<P
import json
d=json.load(open("gpurun_out/r6mv/k_$k.json")); r=d["roofline"]
print("K $k: %.3f M sims/s, %.4f ms/step, %.3f sims/slot/wave, %s %.1f boards/launch, tower %.1f us" % (d["value"]/1e6, d["ms_per_step"], d["sims_per_slot_per_wave"], r["kernel"], r["avg_boards_per_launch"], 1e3*r["avg_launch_ms"]))
P
done | tee gpurun_out/r6mv/sweep.txt
timeout 300 python tools/phase_profile.py --games 16384 --groups 1 --filters 64 --sims 400 --reset-every 1 --waves 1024 > gpurun_out/r6mv/phase.jsonl 2> gpurun_out/r6mv/phase.err; tail -1 gpurun_out/r6mv/phase.jsonl | tee -a gpurun_out/r6mv/sweep.txt
