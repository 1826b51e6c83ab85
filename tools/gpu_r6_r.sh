#!/bin/bash
# round 6, call R: both paired forms in one launch (k_tower16x2m): parity test, headline with and without
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6r
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_free_running_gpu.py tests/test_net.py -x -q -m gpu > gpurun_out/r6r/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6r/tests.log; tail -3 gpurun_out/r6r/tests.log
run() { local name=$1; shift; env "$@" timeout 600 python bench.py --headline-only --steps 2000 ${ARGS} > gpurun_out/r6r/$name.json 2> gpurun_out/r6r/$name.err; }
ARGS="" run mixed_on
ARGS="" run mixed_off AZHIP_TOWER_MIXED=0
ARGS="--no-prof" run mixed_on_noprof
ARGS="--no-prof" run mixed_off_noprof AZHIP_TOWER_MIXED=0
ARGS="--no-prof" run mixed_on_k4_noprof AZHIP_RUN_K=4
ARGS="--no-prof" run mixed_on_k2_noprof AZHIP_RUN_K=2
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6r/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], "sims/slot/wave %.3f" % d["sims_per_slot_per_wave"], "tower ms/step %.4f" % r.get("kernel_ms_per_step",0), r.get("kernel"), "frac %.3f" % r.get("frac",0), "wall %.3f" % r.get("frac_over_wall",0))
    except Exception as ex:
        print(f, "unreadable:", ex)
P
