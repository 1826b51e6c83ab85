#!/bin/bash
# round 6, call L: the second round of a two-round tower launch as 7-board workgroups -- bit-exactness of the new form, then the headline
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6l
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_net.py tests/test_free_running_gpu.py tests/test_eval_cache_gpu.py -x -q -m gpu > gpurun_out/r6l/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r6l/tests.log
tail -3 gpurun_out/r6l/tests.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --headline-only --steps 1000 ${ARGS} > gpurun_out/r6l/$name.json 2> gpurun_out/r6l/$name.err
}
ARGS="" run rounds_on_k3
ARGS="" run rounds_off_k3 AZHIP_TOWER_ROUNDS=0
ARGS="" run rounds_on_k2 AZHIP_RUN_K=2
ARGS="" run rounds_on_k4 AZHIP_RUN_K=4
ARGS="--no-prof" run rounds_on_k3_noprof
ARGS="--no-prof" run rounds_on_k2_noprof AZHIP_RUN_K=2
ARGS="--no-prof" run rounds_off_k3_noprof AZHIP_TOWER_ROUNDS=0
timeout 300 python tools/drift.py --windows 50 --waves 500 > gpurun_out/r6l/drift_free_running.jsonl 2>&1
timeout 300 python tools/drift.py --windows 30 --waves 1500 --lock-step > gpurun_out/r6l/drift_lock_step.jsonl 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6l/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], "sims/slot/wave %.3f" % d["sims_per_slot_per_wave"], "tower ms/step %.4f" % r.get("kernel_ms_per_step",0), "launches", r.get("launches"), "frac %.3f" % r.get("frac",0), "wall %.3f" % r.get("frac_over_wall",0))
    except Exception as ex:
        print(f, "unreadable:", ex)
P
awk 'NR%5==0' gpurun_out/r6l/drift_free_running.jsonl | cut -c1-220
echo; awk 'NR%5==0' gpurun_out/r6l/drift_lock_step.jsonl | cut -c1-220
