"""Runs one BASELINE.json-style configuration for a number of waves and prints throughput + stats."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip.network import ResNetHP, random_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--game", default="mancala")
ap.add_argument("--slots", type=int, default=8192)
ap.add_argument("--sims", type=int, default=800)
ap.add_argument("--waves", type=int, default=1600)
ap.add_argument("--blocks", type=int, default=5)
ap.add_argument("--games", type=int, default=-1)
ap.add_argument("--groups", type=int, default=1)
ap.add_argument("--filters", type=int, default=64)
ap.add_argument("--bf16", action="store_true", help="bf16 tower (az_engine_cfg.net_bf16)")
ap.add_argument("--prof", action="store_true", help="time every kernel class with events (serialises the streams a little)")
a = ap.parse_args()
gid = {"connect-four": 0, "tictactoe": 1, "mancala": 2}[a.game]
hp = ResNetHP(a.blocks, a.filters, (3, 3), 32, 32)
e = azhip.Engine(game=gid, oracle=azhip.ORACLE_RESNET, num_workers=a.slots, batch_size=a.slots // a.groups, num_iters_per_turn=a.sims, cpuct=2.0,
                 dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1,
                 num_blocks=a.blocks, num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32,
                 max_moves_per_game=256 if gid == 2 else 0, net_bf16=1 if a.bf16 else 0)
e.net_set_params(random_params(gid, hp))
e.selfplay_begin(a.games, 0)
if a.prof:
    e.selfplay_step(a.sims)
    e.prof_enable(True)
    e.prof_reset()
s0 = e.selfplay_stats()
base = (s0.simulations, s0.waves, s0.nodes_traversed, s0.leaf_evals)      # --prof: the warm-up move is not part of the timed region
t0 = time.perf_counter()
done = 0
while done < a.waves and e.selfplay_active() > 0:
    n = min(a.sims, a.waves - done)
    e.selfplay_step(n)
    done += n
s = e.selfplay_stats()
dt = time.perf_counter() - t0
print("%s slots=%d sims/move=%d waves=%d: %.0f sims/s, %.2f ms/wave, depth %.2f, evals/sim %.3f, games done %d, moves %d"
      % (a.game, a.slots, a.sims, s.waves - base[1], (s.simulations - base[0]) / dt, 1e3 * dt / max(s.waves - base[1], 1),
         (s.nodes_traversed - base[2]) / max(s.simulations - base[0], 1), (s.leaf_evals - base[3]) / max(s.simulations - base[0], 1), s.games, s.moves))
if a.prof:
    print("  tower kernel:", e.net_last_kernel())
    for k, v in e.prof_get().items():
        if v["launches"]:
            print("  %-12s %6d launches  %8.1f us each" % (k, v["launches"], 1e3 * v["ms"] / v["launches"]))
