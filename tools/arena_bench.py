"""Time one arena evaluation (compare_networks, training.jl:159-172) at the reference's connect-four arena
parameters (games/connect-four/params.jl:31-44): 128 games, 128 workers, 600 sims/move, ResNet 5x128."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alphazero.jl_amd"))
import azhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=128)
ap.add_argument("--workers", type=int, default=128)
ap.add_argument("--sims", type=int, default=600)
ap.add_argument("--filters", type=int, default=128)
ap.add_argument("--blocks", type=int, default=5)
a = ap.parse_args()
gspec = azhip.ConnectFourSpec()
hp = azhip.ResNetHP(num_blocks=a.blocks, num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32)
c, b = azhip.ResNet(gspec, hp, seed=1), azhip.ResNet(gspec, hp, seed=2)
mp = azhip.MctsParams(num_iters_per_turn=a.sims, cpuct=2.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0,
                      temperature=azhip.ConstSchedule(0.2))
params = azhip.ArenaParams(mcts=mp, sim=azhip.SimParams(num_games=a.games, num_workers=a.workers, batch_size=a.workers,
                                                        use_gpu=True, reset_every=2, flip_probability=0.5,
                                                        alternate_colors=True), update_threshold=0.05)
t0 = time.perf_counter()
ev = azhip.compare_networks(gspec, c, b, params)
dt = time.perf_counter() - t0
plies = None
print("arena: %d games x %d sims  avgr %.3f  redundancy %.3f  %.2f s (%.2f s inside)" % (a.games, a.sims, ev.avgr, ev.redundancy, dt, ev.time))
