#!/bin/bash
# round 6, call Q: (1) the iteration's self-play: worker count x slot groups; (2) HBM counters of the free-running wave's
# kernels in the steady state (separate --pmc passes, --kernel-trace only)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6q
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python - <<'P' > gpurun_out/r6q/iteration_workers.jsonl 2> gpurun_out/r6q/iteration_workers.err
import json, os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "alphazero.jl_amd"))
import azhip
from azhip.training import SelfPlayParams, self_play_step_device
gspec = azhip.ConnectFourSpec()
hp = azhip.ResNetHP(num_blocks=5, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
best = azhip.ResNet(gspec, hp, seed=1)
mcts = azhip.MctsParams(num_iters_per_turn=600, cpuct=2.0, prior_temperature=1.0, temperature=azhip.PLSchedule([0, 20, 30], [1.0, 1.0, 0.3]), dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0)
for workers, groups in ((4096, 2), (4096, 1), (3072, 2), (2560, 2), (2048, 2), (2048, 1), (5000, 2), (1024, 1)):
    sp = SelfPlayParams(mcts=mcts, sim=azhip.SimParams(num_games=5000, num_workers=workers, batch_size=workers // groups, use_gpu=True, reset_every=2))
    mem = azhip.MemoryBuffer(gspec, 400000)
    t0 = time.perf_counter()
    rep = self_play_step_device(gspec, best, sp, mem, seed=1)
    dt = time.perf_counter() - t0
    n = len(mem)
    mem.close()
    print(json.dumps({"workers": workers, "groups": groups, "seconds": round(dt, 2), "samples": n, "sims_per_sec": round(n * 600 / (n / rep.samples_gen_speed)), "simulate_seconds": round(n / rep.samples_gen_speed, 2)}), flush=True)
    from azhip import engine as E
    E.clear_engine_cache()
P
cat gpurun_out/r6q/iteration_workers.jsonl; tail -3 gpurun_out/r6q/iteration_workers.err
i=0
for ctrs in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1)); d=/tmp/fr_pmc_$i; rm -rf $d
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python $R/bench.py --headline-only --no-prof --steps 100 --warmup 5 > $R/gpurun_out/r6q/f32_pass${i}_line.json 2> $R/gpurun_out/r6q/f32_pass${i}.err)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - <<PY
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    n = (r.get("Kernel_Name") or "").replace("void ", "").split("(")[0]
    if any(k in n for k in ("k_tower", "k_tree", "k_heads", "k_move")):
        acc[n].append(float(r["Counter_Value"]))
out = {n: {"last_200_avg_KB": sum(v[-200:]) / len(v[-200:]), "dispatches": len(v)} for n, v in acc.items()}
import json; json.dump({"counter": "$ctrs", "kernels": out}, open("gpurun_out/r6q/pmc_$ctrs.json", "w"), indent=1); print("$ctrs", json.dumps(out)[:900])
PY
  rm -rf $d
done
python -c "
import json
d=json.load(open('gpurun_out/r6q/f32_pass1_line.json')); print('under pmc: boards/launch n/a, sims/slot/wave', d['sims_per_slot_per_wave'], 'evals/sim', d['leaf_evals_per_sim'], 'unique', d['unique_leaf_frac'])"
