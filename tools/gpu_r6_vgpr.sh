#!/bin/bash
# round 6: the paired tower's 176-register form (k_tower16x2c) picked per wave where the background search would cost the 198-register
# form a round of workgroups: parity of the form, the steady state (must not change), the first seconds of a phase, the whole phase
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6vgpr
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_free_running_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python bench.py --steps 2000 --warmup 50 --headline-only > gpurun_out/r6vgpr/steady_$i.json 2> gpurun_out/r6vgpr/steady_$i.err
  python - <<P
import json
d=json.load(open("gpurun_out/r6vgpr/steady_$i.json")); r=d["roofline"]
print("steady: %.3f M sims/s, %.4f ms/step, %.3f sims/slot/wave, %s %.1f boards/launch, tower %.1f us" % (d["value"]/1e6, d["ms_per_step"], d["sims_per_slot_per_wave"], r["kernel"], r["avg_boards_per_launch"], 1e3*r["avg_launch_ms"]))
P
done | tee gpurun_out/r6vgpr/sweep.txt
timeout 300 python tools/phase_profile.py --games 16384 --groups 1 --filters 64 --sims 400 --reset-every 1 --waves 1024 --prof --max-seconds 6 > gpurun_out/r6vgpr/early.jsonl 2> gpurun_out/r6vgpr/early.err
python - <<P | tee -a gpurun_out/r6vgpr/sweep.txt
import json
for l in open("gpurun_out/r6vgpr/early.jsonl"):
    r = json.loads(l)
    if "kernels_us_x_launches" in r: print(r["t"], r["boards_per_wave"], r["ms_per_wave"], r["msims_per_s"], r["tower_kernel"], {k: v[0] for k, v in r["kernels_us_x_launches"].items()})
P
timeout 300 python tools/phase_profile.py --games 16384 --groups 1 --filters 64 --sims 400 --reset-every 1 --waves 1024 > gpurun_out/r6vgpr/phase.jsonl 2> gpurun_out/r6vgpr/phase.err; tail -1 gpurun_out/r6vgpr/phase.jsonl | tee -a gpurun_out/r6vgpr/sweep.txt
