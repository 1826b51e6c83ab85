#!/usr/bin/env python
"""bench.py -- self-play MCTS simulations/sec on MI355X (BASELINE.json metric).

A *step* is one search wave of the hot path over one batch of game slots: select -> gather misses
-> ResNet tower + heads -> expand + backup for every one of the G = 4096 Connect-Four slots (one
run_simulation! each, src/mcts.jl:199-226), plus the move step of play_game (src/play.jl:308-313) after
every 400th wave and slot refill when games end.  Workload = BASELINE.json configs[1]:
Connect-Four, 400 sims/move, 4096 parallel games, ResNet 5x64 fp32, synthetic weights.

N > 1 (launched by torch.distributed.run, one rank per GPU): games shard embarrassingly -- every rank
runs its own 4096 slots with global game ids offset by rank, no collective in the timed region
(weak scaling); value = simulations of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))

import numpy as np  # noqa: E402

# algorithmic work per evaluated leaf, ResNet 5x64 heads 32/32 on 7x6x3 planes
TOWER_FLOP = 2 * 42 * 64 * (27 + 10 * 576 + 64)     # stem + 10 tower convs + both 1x1 head convs
HEADS_FLOP = 2 * (1344 * 7 + 1344 * 64 + 64)
assert TOWER_FLOP + HEADS_FLOP == 31645952
PEAK_FP32_MFMA_TFLOPS = 157.3                        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def pmc_traffic(boards_per_launch):
    """HBM bytes per k_tower launch from the committed PMC passes (profiles/r1/r1d_pmc_summary_groups1.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this workload at 4096 boards per launch; KB units,
    FETCH doubled on gfx950 as MI355X_MICROARCH.md prescribes).  The write side (head features) scales with
    the boards of a launch, the read side (weights, once per XCD L2) does not.  None if the file is absent."""
    p = os.path.join(ROOT, "profiles", "r1", "r1d_pmc_summary_groups1.json")
    try:
        d = json.load(open(p))
        k = [v for name, v in d.items() if "k_tower" in name][0]
        return 2.0 * k["FETCH_SIZE"] * 1024.0 + k["WRITE_SIZE"] * 1024.0 * boards_per_launch / 4096.0
    except Exception:
        return None


def cpu_baseline(blob, hp, nsims, seconds=12.0, threads=None):
    """The oracle (a port, not the reference: Julia is absent) on the host cores: independent Connect-Four
    searches of `nsims` simulations each (= one move of a game) with the fp32 ResNet, one search at a time per
    thread, started until `seconds` have elapsed (bounded sample: the default bench stays within minutes)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import azref as R
    threads = threads or min(os.cpu_count() or 1, 32)
    done = [0] * threads
    roots = [0] * threads
    R.lib()
    t0 = time.perf_counter()

    def work(t):
        r = t
        while time.perf_counter() - t0 < seconds:
            m = R.Mcts(R.C4, oracle=R.ORACLE_NET, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0,
                       net=(hp.num_blocks, hp.num_filters, hp.num_policy_head_filters, hp.num_value_head_filters, blob))
            m.explore(R.Game(R.C4), nsims, seed=1, game_id=r, move=0)
            done[t] += m.total_simulations
            roots[t] += 1
            r += threads
    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    return {"value": sum(done) / dt, "unit": "sims/s", "cores": threads, "kind": "port",
            "sample": "%d Connect-Four searches x %d sims (one move each), ResNet 5x64 fp32, %d threads, %.1f s"
                      % (sum(roots), nsims, threads, dt)}


def kernel_alone(args, blob, dev_index, waves=200):
    """The dominant kernel WITHOUT anything co-scheduled (extra evidence, outside the timed region; not part of `value`):
    the same workload with one slot group, so every network launch takes all slots and runs alone on its stream.
    With several groups the per-launch durations of `roofline` include time shared with the other group's kernels."""
    import azhip
    eng = azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, device=dev_index,
                       num_workers=args.slots, batch_size=args.slots, num_iters_per_turn=args.sims,
                       gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                       prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                       num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    eng.net_set_params(blob)
    eng.selfplay_begin(-1, first_game_id=1 << 28)
    eng.selfplay_step(40)
    s0 = eng.selfplay_stats()
    eng.prof_reset()
    eng.prof_enable(True, classes=("tower",))
    eng.selfplay_step(waves)
    s1 = eng.selfplay_stats()
    tw = eng.prof_get()["tower"]
    eng.prof_enable(False)
    eng.selfplay_end()
    eng.close()
    evals = s1.leaf_evals - s0.leaf_evals
    achieved = evals * TOWER_FLOP / (tw["ms"] * 1e-3) / 1e12 if tw["ms"] > 0 else 0.0
    return {"kernel": "k_tower16x2<ConnectFour,64,false>", "slot_groups": 1, "waves": waves, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "avg_launch_ms": tw["ms"] / max(tw["launches"], 1),
            "avg_boards_per_launch": evals / max(tw["launches"], 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--slots", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--groups", type=int, default=2, help="interleaved slot groups = num_workers / batch_size: 2 (default) overlaps the tree kernels of one half-batch with the network of the other, the reference's num_workers = 2 x batch_size; 1 = one 4096-leaf batch per wave")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the N > 1 code path on a single GPU)")
    ap.add_argument("--no-prof", action="store_true", help="do not wrap launches in HIP events")
    ap.add_argument("--prof-all", action="store_true", help="time every kernel class (default: only the dominant kernel, k_tower)")
    args = ap.parse_args()

    import torch
    import azhip
    from azhip.network import ResNetHP, random_params

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libazhip.so has no CPU fallback)")

    hp = ResNetHP(num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=2026)
    # games/connect-four/params.jl:24-30 with 400 sims (BASELINE.json configs[1])
    eng = azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, device=dev_index,
                       num_workers=args.slots, batch_size=args.slots // args.groups, num_iters_per_turn=args.sims,
                       gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                       prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                       num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    eng.net_set_params(blob)
    dev_name, ncu, hbm = eng.device_info()
    eng.selfplay_begin(-1, first_game_id=rank * (1 << 24))

    def barrier():
        if dist is not None:
            dist.barrier()

    eng.selfplay_step(args.warmup)
    s0 = eng.selfplay_stats()
    if not args.no_prof:
        eng.prof_reset()
        eng.prof_enable(True, classes=None if args.prof_all else ("tower",))
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.selfplay_step(args.steps)
    s1 = eng.selfplay_stats()          # synchronises the engine's stream
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    prof = eng.prof_get() if not args.no_prof else None
    eng.prof_enable(False)

    elapsed = t1 - t0
    sims = s1.simulations - s0.simulations
    evals = s1.leaf_evals - s0.leaf_evals
    trav = s1.nodes_traversed - s0.nodes_traversed
    moves = s1.moves - s0.moves
    local_evals = evals
    local_elapsed = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([sims, evals, trav, moves], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        sims, evals, trav, moves = [float(x) for x in c.tolist()]
    eng.selfplay_end()

    if rank == 0:
        out = {
            "metric": "self-play MCTS sims/sec (Connect-Four, %d parallel games per GPU)" % args.slots,
            "value": sims / elapsed, "unit": "sims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Connect-Four self-play, %d sims/move, %d parallel games per GPU, ResNet 5x64 fp32 "
                                   "(heads 32/32), cpuct 2, eps 0.25, alpha 1, PLSchedule([0,20,30],[1,1,.3]), reset_every 1, num_workers/batch_size = %d; "
                                   "step = one search wave (1 simulation per slot)" % (args.sims, args.slots, args.groups),
                       "slots_per_gpu": args.slots, "sims_per_move": args.sims, "slot_groups": args.groups, "leaves_per_network_launch": args.slots // args.groups, "parallelism": "dp%d (games sharded, no collective in the timed region)" % world,
                       "device": dev_name, "compute_units": ncu},
            "sims_per_sec_per_gpu": sims / elapsed / world,
            "samples_per_sec": moves / elapsed,
            "avg_exploration_depth": trav / max(sims, 1),
            "leaf_evals_per_sim": evals / max(sims, 1),
        }
        if prof is not None:
            tw = prof["tower"]
            flops = local_evals * TOWER_FLOP
            achieved = flops / (tw["ms"] * 1e-3) / 1e12 if tw["ms"] > 0 else 0.0
            out["roofline"] = {
                # one slot group: the paired 21-row-tile kernel; several groups: the 11-tile kernel (pick_tower, azhip.hip)
                "kernel": "k_tower16x2<ConnectFour,64,false>" if args.groups == 1 else "k_tower16<ConnectFour,64,false,11>",
                "bound": "mfma", "achieved": achieved,
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "traffic": pmc_traffic(local_evals / max(tw["launches"], 1)),
                "flop_per_board": TOWER_FLOP, "avg_launch_ms": tw["ms"] / max(tw["launches"], 1),
                "avg_boards_per_launch": local_evals / max(tw["launches"], 1),
                "launches": tw["launches"],
                # the same FLOPs over the WALL time of the timed region (all kernels, both slot groups): with several
                # groups the per-launch durations above overlap each other, this figure has no such ambiguity
                "step_achieved": flops / local_elapsed / 1e12, "step_frac": flops / local_elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            }
            out["kernel_ms"] = {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]}
        if prof is not None and world == 1 and args.groups > 1:
            out["roofline_kernel_alone"] = kernel_alone(args, blob, dev_index)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(blob, hp, args.sims)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
