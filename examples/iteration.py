"""One AlphaZero iteration around the (not yet device-resident) optimiser step, entirely on the MI355X engine:
self-play -> device replay memory -> learning status of the current network -> arena against the best network.

    python examples/iteration.py [--games 256] [--workers 128] [--sims 100]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip.training import SelfPlayParams, evaluation_half_of_learning_step, self_play_step_device  # noqa: E402


def main(games=256, workers=128, sims=100, filters=64, seed=1, quiet=False):
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=5, num_filters=filters, num_policy_head_filters=32, num_value_head_filters=32)
    bestnn, curnn = azhip.ResNet(gspec, hp, seed=1), azhip.ResNet(gspec, hp, seed=2)
    sp = SelfPlayParams(
        mcts=azhip.MctsParams(num_iters_per_turn=sims, cpuct=2.0, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0,
                              temperature=azhip.PLSchedule([0, 20, 30], [1.0, 1.0, 0.3])),
        sim=azhip.SimParams(num_games=games, num_workers=workers, batch_size=workers // 2, use_gpu=True, reset_every=2))
    arena = azhip.ArenaParams(
        mcts=azhip.MctsParams(num_iters_per_turn=sims, cpuct=2.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0,
                              temperature=azhip.ConstSchedule(0.2)),
        sim=azhip.SimParams(num_games=max(2, games // 4), num_workers=max(2, min(workers, games // 4)), batch_size=max(2, min(workers, games // 4)),
                            use_gpu=True, reset_every=2, flip_probability=0.5, alternate_colors=True),
        update_threshold=0.05)
    lp = azhip.LearningParams(samples_weighing_policy=azhip.LOG_WEIGHT, l2_regularization=1e-4, loss_computation_batch_size=1024)
    memory = azhip.MemoryBuffer(gspec, 400_000)
    t0 = time.perf_counter()
    rep = self_play_step_device(gspec, bestnn, sp, memory, seed=seed)
    t1 = time.perf_counter()
    status, ev, replace = evaluation_half_of_learning_step(gspec, curnn, bestnn, memory, lp, arena, seed=seed)
    t2 = time.perf_counter()
    if not quiet:
        print("self-play: %d games -> %d samples (%d distinct boards) in %.2f s, %.0f samples/s, depth %.2f"
              % (games, rep.memory_size, rep.memory_num_distinct_boards, t1 - t0, rep.samples_gen_speed, rep.average_exploration_depth))
        print("learning status of the current network: L %.4f  Lp %.4f  Lv %.4f  Lreg %.4f  Linv %.4f  Hp %.4f  Hpnet %.4f"
              % (status.loss.L, status.loss.Lp, status.loss.Lv, status.loss.Lreg, status.loss.Linv, status.Hp, status.Hpnet))
        print("arena: average reward %.3f, redundancy %.3f -> %s (%.2f s for status + arena)"
              % (ev.avgr, ev.redundancy, "replace the best network" if replace else "keep the best network", t2 - t1))
    memory.close()
    return rep, status, ev, replace


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--workers", type=int, default=128)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--filters", type=int, default=64)
    a = ap.parse_args()
    main(a.games, a.workers, a.sims, a.filters)
