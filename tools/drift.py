#!/usr/bin/env python
"""How long until a refill-forever phase is in its steady state?  Consecutive windows of the headline workload after engine start:
simulations per slot and wave, network evaluations per simulation, sims/s per window.
    python tools/drift.py [--windows 40] [--waves 500] [--lock-step]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip.network import ResNetHP, random_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=40)
ap.add_argument("--waves", type=int, default=500)
ap.add_argument("--lock-step", action="store_true")
a = ap.parse_args()
hp = ResNetHP(num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
e = azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=4096, batch_size=4096, num_iters_per_turn=400, gamma=1.0, cpuct=2.0,
                 dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                 num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32, lock_step=1 if a.lock_step else 0)
e.net_set_params(random_params(0, hp, seed=2026))
e.selfplay_begin(-1, 0)
s0 = e.selfplay_stats()
for w in range(a.windows):
    t0 = time.perf_counter()
    e.selfplay_step(a.waves)
    s1 = e.selfplay_stats()
    dt = time.perf_counter() - t0
    sims, ev, re = s1.simulations - s0.simulations, s1.leaf_evals - s0.leaf_evals, s1.evals_reused - s0.evals_reused
    print(json.dumps({"window": w, "sims_total_M": round(s1.simulations / 1e6, 1), "moves_per_slot": round(s1.moves / 4096, 2), "games": s1.games,
                      "sims_per_sec_M": round(sims / dt / 1e6, 3), "sims_per_slot_per_wave": round(sims / max(s1.slot_launches - s0.slot_launches, 1), 3),
                      "net_evals_per_sim": round((ev - re) / max(sims, 1), 4), "unique_leaf_frac": round((ev - re) / max(ev, 1), 4)}), flush=True)
    s0 = s1
e.selfplay_end()
e.close()
