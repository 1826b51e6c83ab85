#!/bin/bash
# round 6, call O: one or two slot groups for the other configurations (steady state, bench.py's own blocks)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6o
export TMPDIR=/tmp
for g in 1 2; do
  AZ_BENCH_GROUPS=$g AZ_BENCH_ONLY=c3,c2_5x128,c4_mancala,bf16_10x128 timeout 900 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r6o/groups_$g.json 2> gpurun_out/r6o/groups_$g.err
done
python - <<'P'
import json
for g in (1, 2):
    try:
        d = json.load(open("gpurun_out/r6o/groups_%d.json" % g))
        print("groups", g, "headline %.3f M" % (d["value"] / 1e6))
        for k, v in d.get("extra", {}).items():
            print("   ", k, "%.3f M" % (v.get("value", 0) / 1e6), "ms/step %.3f" % v.get("ms_per_step", 0), "sims/slot/wave", round(v.get("sims_per_slot_per_wave", 0), 3), "frac", round((v.get("roofline") or {}).get("frac", 0), 3), (v.get("roofline") or {}).get("kernel"), v.get("error"))
    except Exception as ex:
        print(g, "unreadable", ex)
P
