#!/usr/bin/env python
"""Does the loop learn at the shipped parameters?  

A few training iterations (train!'s loop body, src/training.jl:321-333) at the reference's shipped Connect-Four LEARNING parameters
(games/connect-four/params.jl:46-75: Adam 2e-3, L2 1e-4, batch 1024, LOG_WEIGHT, position averaging, symmetries, nonvalidity
penalty 1, one checkpoint per iteration, update_threshold 0.05) and MCTS parameters (600 sims, cpuct 2, eps 0.25, PLSchedule), with
a REDUCED number of games per iteration (default 1024 instead of 5000) so that three iterations take about a minute.  Prints one
JSON line per iteration: learning_status of the whole data set before / after batch_updates! (L, Lp, Lv, Lreg, Linv, Hp, Hpnet),
the mini-batch losses' trend, the arena result and whether the network was replaced; and a summary line.

    python tools/iterations.py [--iters 3] [--games 1024] [--workers 1024] [--filters 128] > profiles/r5/iterations.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import numpy as np  # noqa: E402

import azhip  # noqa: E402
from azhip.training import SelfPlayParams, train_iteration  # noqa: E402


def status(st):
    return {k: round(float(v), 5) for k, v in (("L", st.loss.L), ("Lp", st.loss.Lp), ("Lv", st.loss.Lv), ("Lreg", st.loss.Lreg),
                                               ("Linv", st.loss.Linv), ("Hp", st.Hp), ("Hpnet", st.Hpnet))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--workers", type=int, default=1024)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--sims", type=int, default=600)
    ap.add_argument("--arena-games", type=int, default=128)
    a = ap.parse_args()
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=5, num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32)
    best = azhip.ResNet(gspec, hp, seed=1)
    cur = best.copy_()
    mcts = azhip.MctsParams(num_iters_per_turn=a.sims, cpuct=2.0, prior_temperature=1.0, temperature=azhip.PLSchedule([0, 20, 30], [1.0, 1.0, 0.3]),
                            dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0)
    sp = SelfPlayParams(mcts=mcts, sim=azhip.SimParams(num_games=a.games, num_workers=a.workers, batch_size=max(1, a.workers // 2), use_gpu=True,
                                                      reset_every=2, flip_probability=0.0, alternate_colors=False))
    amcts = azhip.MctsParams(num_iters_per_turn=a.sims, cpuct=2.0, prior_temperature=1.0, temperature=azhip.ConstSchedule(0.2),
                             dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0)
    arena = azhip.ArenaParams(mcts=amcts, sim=azhip.SimParams(num_games=a.arena_games, num_workers=a.arena_games, batch_size=a.arena_games, use_gpu=True,
                                                               reset_every=2, flip_probability=0.5, alternate_colors=True), update_threshold=0.05)
    lp = azhip.LearningParams(use_position_averaging=True, samples_weighing_policy=azhip.LOG_WEIGHT, batch_size=1024, loss_computation_batch_size=1024,
                              optimiser=azhip.Adam(lr=2e-3), l2_regularization=1e-4, nonvalidity_penalty=1.0, min_checkpoints_per_epoch=1,
                              max_batches_per_checkpoint=2000, num_checkpoints=1)
    mem = azhip.MemoryBuffer(gspec, 400000)
    rows = []
    try:
        for it in range(a.iters):
            t0 = time.perf_counter()
            cur, best, rep, lr = train_iteration(gspec, cur, best, mem, sp, lp, arena, seed=1 + it)
            ck = lr.checkpoints[-1]
            ls = np.asarray(lr.losses, dtype=np.float64)
            q = max(1, len(ls) // 4)
            row = {"iteration": it + 1, "seconds": round(time.perf_counter() - t0, 2), "memory_size": rep.memory_size,
                   "distinct_boards": rep.memory_num_distinct_boards, "samples_per_sec": round(rep.samples_gen_speed, 1),
                   "avg_exploration_depth": round(rep.average_exploration_depth, 3), "optimiser_steps": int(len(ls)),
                   "minibatch_loss_first_quarter_mean": round(float(ls[:q].mean()), 4), "minibatch_loss_last_quarter_mean": round(float(ls[-q:].mean()), 4),
                   "status_before": status(lr.initial_status), "status_after": status(ck.status_after),
                   "arena_avg_reward": round(float(ck.evaluation.avgr), 4), "arena_redundancy": round(float(ck.evaluation.redundancy), 4),
                   "nn_replaced": bool(ck.nn_replaced)}
            rows.append(row)
            print(json.dumps(row), flush=True)
    finally:
        mem.close()
    print(json.dumps({"summary": "learning_status L of the whole data set before -> after each iteration's batch_updates!",
                      "L": [[r["status_before"]["L"], r["status_after"]["L"]] for r in rows],
                      "Lp": [[r["status_before"]["Lp"], r["status_after"]["Lp"]] for r in rows],
                      "Lv": [[r["status_before"]["Lv"], r["status_after"]["Lv"]] for r in rows],
                      "arena": [r["arena_avg_reward"] for r in rows], "replaced": [r["nn_replaced"] for r in rows],
                      "params": vars(a), "network": "ResNet 5x%d fp32, random initial weights" % a.filters}), flush=True)


if __name__ == "__main__":
    main()
