#!/bin/bash
# round 6, call P: the stop word back (from its own stream): headline, Mancala, the 128-filter blocks, the iteration
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6p
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_free_running_gpu.py tests/test_eval_cache_gpu.py -x -q -m gpu > gpurun_out/r6p/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6p/tests.log; tail -2 gpurun_out/r6p/tests.log
AZ_BENCH_ONLY=c3,c2_5x128,c4_mancala,bf16_10x128,workers_128_5x128,whole_phase,iteration timeout 1200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r6p/bench.json 2> gpurun_out/r6p/bench.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r6p/bench.json"))
print("headline %.3f M" % (d["value"] / 1e6), "spw", round(d["sims_per_slot_per_wave"],3), {k: round(d[k]["value"]) for k in ("value_long", "value_cache_off", "value_lock_step") if k in d})
for k, v in d.get("extra", {}).items():
    print("   ", k, "%.3f M" % ((v.get("value") or 0) / 1e6), "ms/step %.3f" % (v.get("ms_per_step") or 0), "spw", round(v.get("sims_per_slot_per_wave") or 0, 3), "frac", round((v.get("roofline") or {}).get("frac", 0), 3), (v.get("roofline") or {}).get("kernel"), v.get("error"), v.get("seconds"), v.get("sims_per_sec_self_play"), v.get("phases_share"))
P
