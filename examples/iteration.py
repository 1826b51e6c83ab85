"""AlphaZero training iterations (train!, src/training.jl:321-333) entirely on the MI355X engine:
self-play -> device replay memory -> learning step (batch updates + loss status + arena checkpoint) -> next iteration.

    python examples/iteration.py [--iters 2] [--games 256] [--workers 128] [--sims 100]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip.training import SelfPlayParams, train_iteration  # noqa: E402


def main(iters=2, games=256, workers=128, sims=100, filters=64, batch=256, seed=1, quiet=False):
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=5, num_filters=filters, num_policy_head_filters=32, num_value_head_filters=32)
    bestnn = azhip.ResNet(gspec, hp, seed=1)
    curnn = bestnn.copy_()
    sp = SelfPlayParams(
        mcts=azhip.MctsParams(num_iters_per_turn=sims, cpuct=2.0, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0,
                              temperature=azhip.PLSchedule([0, 20, 30], [1.0, 1.0, 0.3])),
        sim=azhip.SimParams(num_games=games, num_workers=workers, batch_size=max(1, workers // 2), use_gpu=True, reset_every=2,
                            lock_step=True))   # reset_every 2: which games share a tree is a race between free-running workers (as in the reference); lock step makes the run repeatable
    ng = max(2, games // 4)
    arena = azhip.ArenaParams(
        mcts=azhip.MctsParams(num_iters_per_turn=sims, cpuct=2.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0,
                              temperature=azhip.ConstSchedule(0.2)),
        sim=azhip.SimParams(num_games=ng, num_workers=max(2, min(workers, ng)), batch_size=max(2, min(workers, ng)),
                            use_gpu=True, reset_every=2, flip_probability=0.5, alternate_colors=True),
        update_threshold=0.05)
    lp = azhip.LearningParams(samples_weighing_policy=azhip.LOG_WEIGHT, l2_regularization=1e-4, loss_computation_batch_size=1024,
                              batch_size=batch, optimiser=azhip.Adam(lr=2e-3), min_checkpoints_per_epoch=1,
                              max_batches_per_checkpoint=2000, num_checkpoints=1)
    memory = azhip.MemoryBuffer(gspec, 400_000)
    out = []
    for it in range(iters):
        t0 = time.perf_counter()
        curnn, bestnn, rep, lr = train_iteration(gspec, curnn, bestnn, memory, sp, lp, arena, seed=seed + it)
        out.append((rep, lr))
        if not quiet:
            ck = lr.checkpoints[-1]
            print("iteration %d: %d samples in memory (%d distinct boards), %.0f samples/s | %d batch updates, loss %.3f -> %.3f, "
                  "status L %.3f -> %.3f | arena avg reward %+.3f -> %s | %.1f s"
                  % (it + 1, rep.memory_size, rep.memory_num_distinct_boards, rep.samples_gen_speed, len(lr.losses),
                     lr.losses[0] if len(lr.losses) else float("nan"), lr.losses[-1] if len(lr.losses) else float("nan"),
                     lr.initial_status.loss.L, ck.status_after.loss.L, ck.evaluation.avgr,
                     "new best network" if ck.nn_replaced else "best network kept", time.perf_counter() - t0))
    memory.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--workers", type=int, default=128)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--filters", type=int, default=64)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    main(a.iters, a.games, a.workers, a.sims, a.filters, a.batch)
