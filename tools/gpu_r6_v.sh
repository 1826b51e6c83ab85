#!/bin/bash
# round 6, call V: the free-running arena: parity, then the arena / iteration blocks
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6v
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_arena_gpu.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r6v/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6v/tests.log; tail -15 gpurun_out/r6v/tests.log
AZ_BENCH_ONLY=arena_128 timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r6v/arena_fr.json 2> gpurun_out/r6v/arena_fr.err
AZHIP_ARENA_FR=0 AZ_BENCH_ONLY=arena_128 timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r6v/arena_lock.json 2> gpurun_out/r6v/arena_lock.err
python - <<'P'
import json
for f in ("arena_fr", "arena_lock"):
    try:
        d = json.load(open("gpurun_out/r6v/%s.json" % f)); print(f, d["extra"]["arena_128"])
    except Exception as ex:
        print(f, "unreadable", ex); print(open("gpurun_out/r6v/%s.err" % f).read()[-800:])
P
