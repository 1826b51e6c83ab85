#!/bin/bash
# round 6, call J: explore! that runs ahead (hooks, arena): parity tests, then the arena / 128-worker / iteration blocks of bench.py
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6j
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_arena_gpu.py tests/test_tree_gpu.py tests/test_selfplay_gpu.py tests/test_host_fallback_gpu.py tests/test_capacity_gpu.py tests/test_reference_golden.py tests/test_numerics_gpu.py tests/test_split_fallback_gpu.py -x -q -m gpu > gpurun_out/r6j/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r6j/tests.log
tail -4 gpurun_out/r6j/tests.log
AZ_BENCH_ONLY=arena_128,workers_128_5x128,c2_5x128 timeout 900 python bench.py --no-cpu-baseline --no-variants --steps 200 > gpurun_out/r6j/bench_small.json 2> gpurun_out/r6j/bench_small.err
AZHIP_EXPLORE_K=1 AZ_BENCH_ONLY=arena_128 timeout 900 python bench.py --no-cpu-baseline --no-variants --steps 200 > gpurun_out/r6j/bench_arena_lock.json 2> gpurun_out/r6j/bench_arena_lock.err
python - <<'P'
import json
for f in ("bench_small", "bench_arena_lock"):
    try:
        d = json.load(open("gpurun_out/r6j/%s.json" % f))
        print(f, "headline %.3f M" % (d["value"] / 1e6))
        for k, v in d.get("extra", {}).items():
            print("  ", k, {x: v.get(x) for x in ("value", "seconds", "seconds_inside_simulate", "error", "sims_per_slot_per_wave", "ms_per_step")}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("kernel"))
    except Exception as ex:
        print(f, "unreadable", ex)
P
