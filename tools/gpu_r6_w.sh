#!/bin/bash
# round 6, call W: where a free-running arena's time goes (waves, kernel stats)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6w
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r6w
timeout 300 python -m pytest tests/test_arena_gpu.py -x -q -m gpu 2>&1 | tail -3
AZHIP_ARENA_TRACE=1 AZ_BENCH_ONLY=arena_128 timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > $OUT/arena_fr.json 2> $OUT/arena_fr.err
grep "free-running arena" $OUT/arena_fr.err; python -c "import json; print(json.load(open(\"$OUT/arena_fr.json\"))[\"extra\"][\"arena_128\"])"
for mode in 1 0; do
  (cd /tmp && AZHIP_ARENA_FR=$mode AZHIP_ARENA_TRACE=1 AZ_BENCH_ONLY=arena_128 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o arena -- python $OLDPWD/bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > $OUT/prof_$mode.json 2> $OUT/prof_$mode.err)
  f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/arena_fr${mode}_kernel_stats.csv 2>/dev/null
  echo "== AZHIP_ARENA_FR=$mode"; head -12 "$f" | cut -c1-200
done
