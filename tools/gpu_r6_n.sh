#!/bin/bash
# round 6, call N: the driver's command with the 3-game-length warm-up; 5x128 at the BASELINE batch with one and two slot groups
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6n
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_net.py tests/test_bench_multirank_gpu.py tests/test_memory_gpu.py -x -q -m gpu > gpurun_out/r6n/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6n/tests.log; tail -2 gpurun_out/r6n/tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --headline-only > gpurun_out/r6n/driver_style_$i.json 2> gpurun_out/r6n/driver_style_$i.err
done
timeout 300 python bench.py --headline-only --no-prof > gpurun_out/r6n/default_2000_noprof.json 2> gpurun_out/r6n/default_2000_noprof.err
timeout 300 python bench.py --headline-only > gpurun_out/r6n/default_2000.json 2> gpurun_out/r6n/default_2000.err
timeout 300 python bench.py --headline-only --lock-step --no-prof > gpurun_out/r6n/lock_2000_noprof.json 2> gpurun_out/r6n/lock_2000_noprof.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6n/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        print(f.split("/")[-1], "%.3f M" % (d["value"]/1e6), "steps", d["steps"], "ms/step %.4f" % d["ms_per_step"], "sims/slot/wave %.3f" % d["sims_per_slot_per_wave"], "unique %.3f" % d["unique_leaf_frac"], "evals/sim %.3f" % d["leaf_evals_per_sim"], "frac %.3f" % r.get("frac",0), "wall %.3f" % r.get("frac_over_wall",0), "warm waves", d["warmup_waves_run"])
    except Exception as ex:
        print(f, "unreadable:", ex)
P
for g in 1 2; do timeout 200 python tools/run_config.py --game connect-four --slots 4096 --sims 600 --filters 128 --groups $g --waves 1200 2>&1 | tail -1; done
for g in 1 2; do timeout 200 python tools/run_config.py --game mancala --slots 8192 --sims 800 --groups $g --waves 1600 2>&1 | tail -1; done
timeout 200 python tools/run_config.py --game connect-four --slots 4096 --sims 400 --blocks 10 --filters 128 --bf16 --groups 1 --waves 1200 2>&1 | tail -1
