#!/bin/bash
# round 6, call S: side streams for every slot group: two groups against one, headline and the 128-filter blocks
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6s
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_free_running_gpu.py tests/test_eval_cache_gpu.py tests/test_tree_gpu.py tests/test_split_fallback_gpu.py -x -q -m gpu > gpurun_out/r6s/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6s/tests.log; tail -3 gpurun_out/r6s/tests.log
run() { local name=$1; shift; env "$@" timeout 600 python bench.py --headline-only --steps 2000 --no-prof ${ARGS} > gpurun_out/r6s/$name.json 2> gpurun_out/r6s/$name.err; }
ARGS="--groups 1" run g1
ARGS="--groups 2" run g2
ARGS="--groups 2" run g2_x2 AZHIP_TOWER=21
ARGS="--groups 2" run g2_k2 AZHIP_RUN_K=2
ARGS="--groups 2" run g2_k4 AZHIP_RUN_K=4
ARGS="--groups 4" run g4
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6s/*.json")):
    try:
        d=json.load(open(f))
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], "sims/slot/wave %.3f" % d["sims_per_slot_per_wave"], "unique %.3f" % d["unique_leaf_frac"])
    except Exception as ex:
        print(f, "unreadable:", ex)
P
for g in 1 2; do
  AZ_BENCH_GROUPS=$g AZ_BENCH_ONLY=c2_5x128,c4_mancala,bf16_10x128 timeout 900 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 > gpurun_out/r6s/blocks_g$g.json 2> gpurun_out/r6s/blocks_g$g.err
done
python - <<'P'
import json
for g in (1, 2):
    try:
        d = json.load(open("gpurun_out/r6s/blocks_g%d.json" % g))
        for k, v in d.get("extra", {}).items():
            print("groups", g, k, "%.3f M" % (v.get("value", 0) / 1e6), "ms/step %.3f" % v.get("ms_per_step", 0), "spw", round(v.get("sims_per_slot_per_wave", 0), 3), "frac", round((v.get("roofline") or {}).get("frac", 0), 3), (v.get("roofline") or {}).get("kernel"), v.get("error"))
    except Exception as ex:
        print(g, "unreadable", ex)
P
