#!/usr/bin/env python
"""bench.py -- self-play MCTS simulations/sec on MI355X (BASELINE.json metric).

A *step* is one search wave of the hot path over one batch of game slots: expand + backup of the leaves the
network answered -> select -> gather misses -> ResNet tower + heads, for the G = 4096 Connect-Four slots.
Round 6 (free-running, az_engine_cfg.lock_step = 0): a wave carries every slot through as many run_simulation!
calls (src/mcts.jl:199-226) as end on a terminal state or on a state the evaluation cache answers, up to its next
question for the network (~2 per slot and wave); a slot plays its move (src/play.jl:308-313) when ITS explore! is
complete and takes the next game id when ITS game ends.  Rounds 1-5 (lock step, `value_lock_step`): one simulation per
slot and wave, all slots move every 400th wave.  The metric counts simulations, not waves.  Workload = BASELINE.json
configs[1]: Connect-Four, 400 sims/move, 4096 parallel games, ResNet 5x64 fp32, synthetic weights.

N > 1 (one rank per GPU): games shard embarrassingly -- every rank runs its own 4096 slots with global
game ids offset by rank, no collective in the timed region (weak scaling); value = simulations of all
ranks / max-over-ranks time.  Launched either by torch.distributed.run (RANK / WORLD_SIZE in the
environment) or as plain `python bench.py --gpus N`: with WORLD_SIZE unset the script re-executes
itself under torch.distributed.run with N ranks (self_launch), so `--gpus N` always means N ranks.
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

_T0 = time.perf_counter()


def trace(msg):
    """AZ_BENCH_TRACE=1: progress marks on stderr (rank, seconds since start) -- where a run that does not finish stands"""
    if os.environ.get("AZ_BENCH_TRACE"):
        print("[bench rank %s +%.1f s] %s" % (os.environ.get("RANK", "0"), time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)

# algorithmic work per evaluated leaf, ResNet 5x64 heads 32/32 on 7x6x3 planes
TOWER_FLOP = 2 * 42 * 64 * (27 + 10 * 576 + 64)     # stem + 10 tower convs + both 1x1 head convs
HEADS_FLOP = 2 * (1344 * 7 + 1344 * 64 + 64)
assert TOWER_FLOP + HEADS_FLOP == 31645952
PEAK_FP32_MFMA_TFLOPS = 157.3                        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# The tower kernels skip the (16-row tile, tap) products whose tap falls off the board for the whole tile (Geo16,
# csrc/resnet16.h).  roofline.achieved / frac count the work DONE: the dense FLOPs of the real board rows x the fraction of the
# convolutions' products the kernels that ran executed (az_prof.exec_units / units, reported by the library per launch), so no
# fraction can exceed 1.  The dense convolution's count (every tap of every position) is kept beside it as
# dense_achieved / dense_frac -- a speed-up over a dense kernel, not a fraction of the machine.


PEAK_HBM_GBS = 8000.0                                # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2500.0                       # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (no sparsity)
# Little's law for the search-tree kernel: a slot's simulation is a CHAIN of dependent 128-byte node-line loads (one line in
# flight per slot); a dependent load that misses L2 takes ~840 shader cycles (tools/probes/chase_latency.hip)
# ~ 400 ns at the sustained clock.  G slots therefore cannot move more than G x 128 B per 400 ns, whatever the kernel does.
TREE_LINE_BYTES, TREE_LOAD_LATENCY_S = 128.0, 400e-9
GAME_DIMS = {0: ("Connect-Four", 42, 3), 1: ("Tic-tac-toe", 9, 3), 2: ("Mancala", 14, 5)}   # name, board positions, planes


def tower_flop(game, hp):
    """dense (algorithmic) FLOP of stem + residual tower + both 1x1 head convolutions per board"""
    _, P, Cin = GAME_DIMS[game]
    F = hp.num_filters
    return 2 * P * F * (9 * Cin + 2 * hp.num_blocks * 9 * F + hp.num_policy_head_filters + hp.num_value_head_filters)


def tower_roofline(game, hp, bf16, kernel, tw, evals, wall_s):
    """roofline object of the tower launches `tw` (az_prof class "tower") that evaluated `evals` boards within wall_s seconds.
    achieved = EXECUTED FLOPs / exclusive kernel time: per board the dense FLOPs of stem + head 1x1 convolutions and of the 3x3
    tower convolutions x the executed-product fraction of the kernels that ran; time = sum of the HIP-event durations of the
    launches, clipped to the wall time (towers of several slot groups overlap)."""
    _, P, Cin = GAME_DIMS[game]
    F = hp.num_filters
    conv, other = 2 * hp.num_blocks * 9 * F, 9 * Cin + hp.num_policy_head_filters + hp.num_value_head_filters
    dense = 2 * P * F * (conv + other)
    ex = tw["exec_units"] / tw["units"] if tw["units"] else 1.0
    executed = 2 * P * F * (conv * ex + other)
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    wall_ms = 1e3 * wall_s
    excl_ms = min(tw["ms"], wall_ms)
    achieved = evals * executed / (excl_ms * 1e-3) / 1e12 if excl_ms > 0 else 0.0
    boards = evals / max(tw["launches"], 1)
    out = {"kernel": kernel, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
           "flop_per_board": executed, "executed_product_frac": ex,
           "dense_flop_per_board": dense, "dense_achieved": achieved * dense / executed, "dense_frac": achieved * dense / executed / peak,
           "dense_note": "dense_* = the same time priced with the dense convolution's FLOPs (zero padding included): a speed-up over a dense kernel, can exceed 1, NOT a machine fraction",
           "launches": tw["launches"], "avg_boards_per_launch": boards, "avg_launch_ms": tw["ms"] / max(tw["launches"], 1),
           "launch_ms_sum": tw["ms"], "wall_ms": wall_ms, "exclusive_ms": excl_ms, "traffic": None}
    t = pmc_lookup(kernel, "bf16" if bf16 else "f32")
    if t is not None:
        out["traffic"] = t[0] * boards / t[1]
    return out


def pmc_lookup(kernel, config="f32"):
    """HBM bytes per launch of `kernel` from the latest separate rocprofv3 --pmc passes (profiles/r5/pmc_summary.json if the round took
    any, else r4's, r3's; tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE in KB, FETCH doubled on gfx950 as MI355X_MICROARCH.md prescribes --
    tools/fetch_calib.sh, profiles/r4/fetch_calibration.txt: 2 x FETCH_SIZE = 128-byte LINES requested, WRITE_SIZE exact);
    returns (bytes, units per launch) or None: counters cannot be read from inside this process."""
    for rnd in ("r6", "r5", "r4", "r3"):                             # this round's passes first
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")))
            ks = d.get("kernels", [])
            for k in [k for k in ks if k.get("config") == config] + ks:  # the pass of this configuration first
                if kernel.startswith(k["match"]):
                    return (2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0, k["units_per_launch"]
        except Exception:
            pass
    return None


HEADLINE_WARMUP_GAME_LENGTHS = 3.0
MEAN_MOVES = {0: 23, 1: 7, 2: 24}     # moves per self-play game at the shipped parameters (measured: 92 961 / 4096, 194 061 / 8192)


def mixing_warmup(game, sims, lengths=1.5):
    """Waves after which a batch of slots that all started at wave 0 is in the steady state of a long phase: slots at every stage
    of a game (1.5 mean game lengths: the first generation of games has ended at moves 7 ... 42, every slot is somewhere in its
    second or third game).  Round 5: with the evaluation cache the state of the batch matters -- 4096 games in lock step through
    the same opening ask for the same evaluations (86 % repeats in the second move), a long phase's batch does not."""
    return int(lengths * MEAN_MOVES[game] * sims)


def warm_up(eng, game, sims, slots, extra_waves=0, lengths=1.5):
    """Steps the phase until it is in the steady state mixing_warmup describes, whatever the schedule: a lock-step wave is one simulation
    per slot, a free-running one about two, so the warm-up is counted in SIMULATIONS (mixing_warmup x slots) and stepped in chunks.
    lengths: mean game lengths to play first.  1.5 (rounds 2-5) leaves the batch lumpy -- the 4096 games that started together still
    end in bunches, and network evaluations per simulation swing between 0.41 and 0.49 with the period of a game (profiles/r6/
    drift_lock_step.jsonl), i.e. +-10 % of throughput depending on WHEN a short window looks; after ~2.5 game lengths the swing is
    below 2 %.  The headline and its variants use 3, the extra blocks (their windows are not the driver's 20 waves) keep 1.5."""
    target = (mixing_warmup(game, sims, lengths) + sims // 2) * slots
    waves = 0

    def drain():
        # the games that end are queued for az_selfplay_collect; a refill-forever phase nobody collects grows that queue by 64 B per move
        # -- and the reallocation of a 20 MB vector inside a 20-wave timed region is a 7 ms stall (seen once in three runs): collect and drop
        eng.selfplay_collect(2 * slots)
    while eng.selfplay_stats().simulations < target:
        eng.selfplay_step(500)
        waves += 500
        drain()
    if extra_waves > waves:
        eng.selfplay_step(extra_waves - waves)
        waves = extra_waves
        drain()
    return waves


def steady_block(azhip, dev_index, label, game, slots, groups, sims, hp, waves, bf16=False, note=None, max_moves=0):
    """One extra configuration, measured like the headline: steady state of a long phase (mixing_warmup), `waves`
    timed search waves, tower launches timed with HIP events, roofline on the tower kernel."""
    groups = int(os.environ.get("AZ_BENCH_GROUPS", groups))         # A/B aid
    from azhip.network import random_params
    eng = azhip.Engine(game=game, oracle=azhip.ORACLE_RESNET, device=dev_index, num_workers=slots, batch_size=slots // groups,
                       num_iters_per_turn=sims, gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                       prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                       num_blocks=hp.num_blocks, num_filters=hp.num_filters, num_policy_head_filters=hp.num_policy_head_filters,
                       num_value_head_filters=hp.num_value_head_filters, net_bf16=1 if bf16 else 0, max_moves_per_game=max_moves)
    try:
        eng.net_set_params(random_params(game, hp, seed=2026))
        dev_bytes = eng.device_bytes()
        eng.selfplay_begin(-1, first_game_id=1 << 27)
        warm_up(eng, game, sims, slots)
        s0 = eng.selfplay_stats()
        eng.prof_reset()
        eng.prof_enable(True, classes=("tower",))
        t0 = time.perf_counter()
        eng.selfplay_step(waves)
        s1 = eng.selfplay_stats()
        dt = time.perf_counter() - t0
        prof = eng.prof_get()
        kernel = eng.net_last_kernel()
        eng.prof_enable(False)
        eng.selfplay_end()
    finally:
        eng.close()
    return block_report(label, game, hp, bf16, kernel, prof, s0, s1, dt, waves, slots, groups, sims, note, dev_bytes)


def block_report(label, game, hp, bf16, kernel, prof, s0, s1, dt, waves, slots, groups, sims, note, dev_bytes):
    sims_n, evals, trav, moves = (s1.simulations - s0.simulations, s1.leaf_evals - s0.leaf_evals,
                                  s1.nodes_traversed - s0.nodes_traversed, s1.moves - s0.moves)
    net_evals = evals - (s1.evals_reused - s0.evals_reused)         # boards the network evaluated: the rest came from the evaluation cache
    out = {"workload": "%s self-play, %d sims/move, %d parallel games, ResNet %dx%d %s, %d slot group(s)%s"
                       % (GAME_DIMS[game][0], sims, slots, hp.num_blocks, hp.num_filters, "bf16 tower (opt-in, NOT the reference's fp32 precision)" if bf16 else "fp32", groups, "; " + note if note else ""),
           "value": sims_n / dt, "unit": "sims/s", "steps": waves, "ms_per_step": 1e3 * dt / max(waves, 1), "dtype": "bf16" if bf16 else "f32",
           "samples_per_sec": moves / dt, "avg_exploration_depth": trav / max(sims_n, 1), "leaf_evals_per_sim": evals / max(sims_n, 1),
           "unique_leaf_frac": net_evals / max(evals, 1),
           "sims_per_slot_per_wave": sims_n / max(s1.slot_launches - s0.slot_launches, 1),
           "engine_device_GB": dev_bytes / 2.0**30,
           "roofline": tower_roofline(game, hp, bf16, kernel, prof["tower"], net_evals, dt)}
    return out


def host_stepped_c5(seconds=6.0):
    """BASELINE configs[4] end to end as far as it can go without OpenSpiel: examples/host_stepped_go9 (plain C++ over
    include/azhip.h: host rules + host trees on up to 16 host threads, 4096 workers in two halves that take turns between the
    host and the network, ResNet 10x128 bf16 on the GPU through az_net_forward at 1600 sims/move).  The rules are a labelled stand-in (OpenSpiel is not in the reference tree): no parity claim, the
    number says what the network seam sustains and how much of the wall time the host needs."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "host_stepped_go9")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    r = subprocess.run([exe, "--workers", "4096", "--sims", "1600", "--seconds", str(seconds)], capture_output=True, text=True, timeout=120)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        raise RuntimeError("host_stepped_go9 exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
    return json.loads(line)


def whole_phase(azhip, dev_index, blob, hp, slots, sims, groups, games_per_slot=4):
    """SURVEY.md §8(d)'s metric over a WHOLE self-play phase: 4 x `slots` Connect-Four games from the empty board to completion
    (az_selfplay_run, device-only; the reference plays 5000 games on 128 workers, games/connect-four/params.jl:15-25, so a slot
    sees many games per phase) -- first moves from empty trees, every move step, refills, trace write-out into the phase
    buffer and the drain at the end while the batch empties -- against the steady-state headline."""
    eng = azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, device=dev_index, num_workers=slots, batch_size=slots // groups,
                       num_iters_per_turn=sims, gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                       prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                       num_blocks=hp.num_blocks, num_filters=hp.num_filters, num_policy_head_filters=32, num_value_head_filters=32)
    try:
        eng.net_set_params(blob)
        dev_bytes = eng.device_bytes()
        from azhip._lib import SelfplayStats
        s0 = SelfplayStats()                                 # az_selfplay_run counts from zero
        eng.prof_reset()
        eng.prof_enable(True, classes=("tower",))
        t0 = time.perf_counter()
        games, _, ng, _, st = eng.selfplay_run(games_per_slot * slots, first_game_id=1 << 26, device_only=True)
        dt = time.perf_counter() - t0
        prof = eng.prof_get()
        kernel = eng.net_last_kernel()
        eng.prof_enable(False)
    finally:
        eng.close()
    out = block_report("whole_phase", azhip.GAME_CONNECT_FOUR, hp, False, kernel, prof, s0, st, dt, int(st.waves - s0.waves), slots, groups, sims,
                       "%d games (%d per slot) played from the empty board to the end, drain included" % (ng, games_per_slot), dev_bytes)
    out["games"] = int(ng)
    out["positions"] = int(st.moves - s0.moves)
    out["seconds"] = dt
    return out



def _hash_phase(azhip, dev_index, games):
    """synthetic Connect-Four positions for the replay-memory / trainer blocks: `games` short games on the hash oracle (8 sims/move)"""
    with azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_HASH, device=dev_index, num_workers=4096, batch_size=4096,
                      num_iters_per_turn=8, reset_every=1, dirichlet_noise_eps=0.25, cpuct=1.0, temperature=([0], [1.0])) as e:
        return e.selfplay_run(games)


def trainer_block(azhip, dev_index, filters, steps=60, batch=1024):
    """the optimiser step batch_updates! (src/learning.jl:123-141) at the reference's learning parameters
    (games/connect-four/params.jl:46-58: Adam 2e-3, L2 1e-4, batch 1024, LOG_WEIGHT), train-mode forward + backward + update on
    the device.  FLOPs = forward + data gradient + weight gradient of every convolution, dense count (the padded taps are
    skipped by the kernels, so the machine fraction is a lower bound on the dense-equivalent one)."""
    gspec = azhip.ConnectFourSpec()
    games, moves, ng, nm, _ = _hash_phase(azhip, dev_index, 8192)
    mem = azhip.MemoryBuffer(gspec, 4 * nm, device=dev_index)
    try:
        mem.push_records(games, moves, ng, nm, 1.0)
        hp = azhip.ResNetHP(num_blocks=5, num_filters=filters, num_policy_head_filters=32, num_value_head_filters=32)
        nn = azhip.ResNet(gspec, hp, seed=1)
        lp = azhip.LearningParams(samples_weighing_policy=azhip.LOG_WEIGHT, l2_regularization=1e-4, loss_computation_batch_size=1024,
                                  batch_size=batch, optimiser=azhip.Adam(lr=2e-3))
        with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True, device=dev_index) as tr:
            tr.batch_updates(3)
            t0 = time.perf_counter()
            ls = tr.batch_updates(steps)
            dt = time.perf_counter() - t0
            n = tr.num_samples()
    finally:
        mem.close()
    flop = 3 * 2 * batch * 42 * filters * (9 * 3 + 2 * 5 * 9 * filters + 64)
    tf = flop / (dt / steps) / 1e12
    return {"workload": "batch_updates! (train-mode forward, backward, Adam): Connect-Four, ResNet 5x%d fp32, batch %d, %d merged boards" % (filters, batch, n),
            "ms_per_step": 1e3 * dt / steps, "steps": steps, "samples_per_sec": batch * steps / dt, "dtype": "f32",
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS,
                         "flop_per_step": flop, "note": "dense FLOPs of forward + data gradient + weight gradient of all convolutions; heads' dense layers not counted"},
            "loss_first_last": [float(ls[0]), float(ls[-1])]}


def memory_block(azhip, dev_index, games=16384):
    """replay memory on the device -- push_trace! (src/memory.jl:74-87) of a phase's records, then the
    Trainer's data set: augment_with_symmetries, merge_by_state, convert_samples (src/memory.jl:89-114, src/learning.jl:98-121).
    HBM-bound integer/byte work: GB/s = bytes the algorithm touches (64 B record in, 112 B sample out; data set: every augmented
    sample read and written by two 128-bit sort passes of 16-byte pairs + the gather, tensors written once) / time."""
    gspec = azhip.ConnectFourSpec()
    g, m, ng, nm, _ = _hash_phase(azhip, dev_index, games)
    mem = azhip.MemoryBuffer(gspec, 4 * nm, device=dev_index)
    try:
        mem.push_records(g, m, ng, nm, 1.0)
        mem.empty()
        t0 = time.perf_counter()
        mem.push_records(g, m, ng, nm, 1.0)
        t_push = time.perf_counter() - t0
        d = mem.dataset(use_symmetries=True, use_position_averaging=True, weighing_policy=azhip.LOG_WEIGHT)
        d.close()
        t0 = time.perf_counter()
        d = mem.dataset(use_symmetries=True, use_position_averaging=True, weighing_policy=azhip.LOG_WEIGHT)
        t_ds = time.perf_counter() - t0
        merged = len(d)
        d.close()
    finally:
        mem.close()
    aug = 2 * nm
    # bytes touched: augment nm x 112 read + aug x 112 written; 16 radix passes over aug x 12 B (key half + index) read + written;
    # merge: aug x 112 read, merged x 112 written; tensors: merged x (112 read + (1 + 126 + 7 + 7 + 1) x 4 written)
    ds_bytes = nm * 112 + aug * 112 + 16 * 2 * aug * 12 + aug * 112 + merged * 112 + merged * (112 + 142 * 4)
    return {"workload": "replay memory on the device: %d games, %d positions (hash oracle, 8 sims/move)" % (ng, nm),
            "push_trace": {"samples": nm, "ms": 1e3 * t_push, "samples_per_sec": nm / t_push, "note": "includes the 64 B / position H2D copy of host records; az_memory_push_engine (device-only phases) skips it"},
            "data_set": {"samples_in": nm, "augmented": aug, "merged_boards": merged, "ms": 1e3 * t_ds, "samples_per_sec": nm / t_ds,
                         "roofline": {"bound": "hbm", "achieved": ds_bytes / t_ds / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ds_bytes / t_ds / 1e9 / PEAK_HBM_GBS,
                                      "algorithmic_bytes": ds_bytes, "note": "wall time of the call (host loop, launches and synchronisations included)"}}}


def _tower_fallbacks(azhip):
    """az_selfplay_stats.tower_fallbacks summed over the engines the mirror keeps cached (arena players, the self-play engine): > 0
    means a split tower gave up on a partner workgroup (50 ms of waiting) and those engines run unsplit since"""
    from azhip import engine as E
    return int(sum(e.selfplay_stats().tower_fallbacks for e in E._cache.values() if e._h is not None))


def _cached_kernels(azhip):
    """tower kernel of the last network launch of every cached engine (role: kernel)"""
    from azhip import engine as E
    return {k[0] or "engine%d" % i: e.net_last_kernel() for i, (k, e) in enumerate(E._cache.items()) if e._h is not None}


def arena_block(azhip, dev_index, filters=128, games=128, sims=600):
    """one checkpoint evaluation compare_networks (src/training.jl:159-172) at the reference's arena
    parameters (games/connect-four/params.jl:31-44): 128 games on 128 workers, 600 sims/move, two ResNet 5x128, flip 0.5,
    colours alternated, ConstSchedule(0.2), eps 0.05."""
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=5, num_filters=filters, num_policy_head_filters=32, num_value_head_filters=32)
    c, b = azhip.ResNet(gspec, hp, seed=1), azhip.ResNet(gspec, hp, seed=2)
    mp = azhip.MctsParams(num_iters_per_turn=sims, cpuct=2.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0, temperature=azhip.ConstSchedule(0.2))
    params = azhip.ArenaParams(mcts=mp, sim=azhip.SimParams(num_games=games, num_workers=games, batch_size=games, use_gpu=True, reset_every=2,
                                                            flip_probability=0.5, alternate_colors=True), update_threshold=0.05)
    t0 = time.perf_counter()
    ev = azhip.compare_networks(gspec, c, b, params, device=dev_index)
    dt = time.perf_counter() - t0
    return {"workload": "compare_networks: %d games, %d workers, %d sims/move, two ResNet 5x%d fp32 (random weights)" % (games, games, sims, filters),
            "seconds": dt, "seconds_inside_simulate": ev.time, "avgr": ev.avgr, "redundancy": ev.redundancy, "tower_fallbacks": _tower_fallbacks(azhip)}


def _status(st):
    return {"L": st.loss.L, "Lp": st.loss.Lp, "Lv": st.loss.Lv, "Lreg": st.loss.Lreg, "Linv": st.loss.Linv, "Hp": st.Hp, "Hpnet": st.Hpnet}


def iteration_block(azhip, dev_index, num_games=5000, workers=4096):
    """ONE training iteration (train!'s loop body, src/training.jl:321-333) at the reference's shipped Connect-Four parameters
    (games/connect-four/params.jl:5-75): self_play_step! 5000 games x 600 sims with ResNet 5x128 (reset_every 2, PLSchedule
    temperature, eps 0.25) -> push_trace! -> learning_step!: Trainer data set (symmetries, merge, tensors), learning_status,
    batch_updates! (Adam 2e-3, batch 1024, min_checkpoints_per_epoch 1, max_batches_per_checkpoint 2000, 1 checkpoint),
    learning_status, compare_networks (128 games).  Random initial weights, everything on the device.  Deviation from the shipped
    file: num_workers %d / batch_size %d instead of 128 / 64 -- the worker count is a throughput knob of the engine (it enters the
    results only through which games share a tree under reset_every 2); extra.workers_128_5x128 has the 128-worker rate."""
    from azhip.training import SelfPlayParams, learning_step, self_play_step_device
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=5, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    best = azhip.ResNet(gspec, hp, seed=1)
    cur = best.copy_()
    mcts = azhip.MctsParams(num_iters_per_turn=600, cpuct=2.0, prior_temperature=1.0, temperature=azhip.PLSchedule([0, 20, 30], [1.0, 1.0, 0.3]),
                            dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0)
    sp = SelfPlayParams(mcts=mcts, sim=azhip.SimParams(num_games=num_games, num_workers=workers, batch_size=workers // 2, use_gpu=True, reset_every=2,
                                                      flip_probability=0.0, alternate_colors=False))
    amcts = azhip.MctsParams(num_iters_per_turn=600, cpuct=2.0, prior_temperature=1.0, temperature=azhip.ConstSchedule(0.2), dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0)
    arena = azhip.ArenaParams(mcts=amcts, sim=azhip.SimParams(num_games=128, num_workers=128, batch_size=128, use_gpu=True, reset_every=2,
                                                               flip_probability=0.5, alternate_colors=True), update_threshold=0.05)
    lp = azhip.LearningParams(use_position_averaging=True, samples_weighing_policy=azhip.LOG_WEIGHT, batch_size=1024, loss_computation_batch_size=1024,
                              optimiser=azhip.Adam(lr=2e-3), l2_regularization=1e-4, nonvalidity_penalty=1.0, min_checkpoints_per_epoch=1,
                              max_batches_per_checkpoint=2000, num_checkpoints=1)
    mem = azhip.MemoryBuffer(gspec, 400000, device=dev_index)        # mem_buffer_size at iteration 0
    try:
        t0 = time.perf_counter()
        rep = self_play_step_device(gspec, best, sp, mem, seed=1)
        t_sp = time.perf_counter() - t0
        samples = len(mem)
        t_sim = samples / rep.samples_gen_speed if rep.samples_gen_speed > 0 else t_sp   # simulate_distributed alone (training.jl:284-287)
        t0 = time.perf_counter()
        cur, best, lr = learning_step(gspec, cur, best, mem, lp, arena, use_symmetries=True, seed=1)
        t_learn = time.perf_counter() - t0
    finally:
        mem.close()
    total = t_sp + t_learn
    phases = {"self_play (simulate)": t_sim, "push_trace! + memory report": t_sp - t_sim, "data set (symmetries, merge, tensors)": lr.time_convert,
              "learning_status (x2)": lr.time_loss + max(0.0, t_learn - lr.time_convert - lr.time_loss - lr.time_train - lr.time_eval),
              "batch_updates!": lr.time_train, "arena (compare_networks)": lr.time_eval}
    return {"workload": iteration_block.__doc__.split("\n")[0].strip() + " -- games/connect-four/params.jl:5-75, %d workers" % workers,
            "seconds": total, "games": num_games, "samples": samples, "sims_per_sec_self_play": samples * 600 / t_sim,
            "optimiser_steps": int(len(lr.losses)), "loss_first_last": [float(lr.losses[0]), float(lr.losses[-1])] if len(lr.losses) else None,
            # train-mode mini-batch losses, mean of each quarter of the epoch: the trend inside batch_updates! (single batches are noisy)
            "minibatch_loss_quarter_means": [float(np.mean(q)) for q in np.array_split(np.asarray(lr.losses, dtype=np.float64), 4)] if len(lr.losses) >= 4 else None,
            # learning_status (src/learning.jl:148-181) over the WHOLE data set, test-mode network, before batch_updates! and after:
            # what the iteration learnt (loss_first_last above are two single mini-batches of the train-mode network)
            "learning_status": {"before": _status(lr.initial_status), "after": _status(lr.checkpoints[-1].status_after) if lr.checkpoints else None},
            "arena_avgr": lr.checkpoints[0].evaluation.avgr if lr.checkpoints else None, "nn_replaced": bool(lr.nn_replaced), "tower_fallbacks": _tower_fallbacks(azhip), "last_tower_kernels": _cached_kernels(azhip),
            "phases_seconds": phases, "phases_share": {k: v / total for k, v in phases.items()},
            "reference": "README.md:76-78: 'about one hour' per iteration on the authors' desktop GPU -- quoted, NOT reproduced here (no Julia in the image)"}


def pmc_traffic(kernel, boards_per_launch):
    """HBM bytes per tower launch from THIS round's PMC passes of this workload (profiles/r2/pmc_tower_summary.json,
    written by tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs; KB units, FETCH doubled
    on gfx950 as MI355X_MICROARCH.md prescribes).  Only used when the summary is for the kernel that actually ran;
    None otherwise (the counters cannot be read from inside this process)."""
    p = os.path.join(ROOT, "profiles", "r2", "pmc_tower_summary.json")
    try:
        d = json.load(open(p))
        if d.get("kernel") != kernel:
            return None
        return (2.0 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024.0 * boards_per_launch / d["boards_per_launch"]
    except Exception:
        return None


def cpu_baseline(blob, hp, nsims, seconds=8.0):
    """The oracle (a port, not the reference: Julia is absent) on the host cores, four bounded samples of the same
    workload: independent Connect-Four searches of `nsims` simulations each (= one move of a game), one
    search at a time per thread, started until the sample's time is up.  `value` = all threads with the fp32 ResNet (the
    configuration the GPU number is quoted on); variants: 1 thread with the ResNet, and the uniform oracle
    (MCTS.RandomOracle: tree cost only, comparable to the reference's 11 us / simulation) on 1 / all threads."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import azref as R
    R.lib()
    # all the CPUs this process can use -- the affinity mask cut down to the cgroup's CPU quota (the GPU box shows 256
    # hardware threads and grants 16 CPUs' worth of time: more threads than that only take turns).  Rounds 2-5 used min(cpu_count, 32).
    ncpu = R.usable_cpus()
    net = (hp.num_blocks, hp.num_filters, hp.num_policy_head_filters, hp.num_value_head_filters, blob)

    def sample(threads, oracle, secs):
        done = [0] * threads
        roots = [0] * threads
        t0 = time.perf_counter()

        def work(t):
            r = t
            while time.perf_counter() - t0 < secs:
                m = R.Mcts(R.C4, oracle=oracle, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, net=net if oracle == R.ORACLE_NET else None)
                m.explore(R.Game(R.C4), nsims, seed=1, game_id=r, move=0)
                done[t] += m.total_simulations
                roots[t] += 1
                r += threads
        ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        dt = time.perf_counter() - t0
        return {"value": sum(done) / dt, "unit": "sims/s", "cores": threads,
                "oracle": "ResNet 5x64 fp32" if oracle == R.ORACLE_NET else "uniform (MCTS.RandomOracle)",
                "sample": "%d Connect-Four searches x %d sims (one move each), %d threads, %.1f s" % (sum(roots), nsims, threads, dt)}
    main = sample(ncpu, R.ORACLE_NET, seconds)
    out = dict(main, kind="port")
    out["sample"] = main["sample"] + ", ResNet 5x64 fp32"
    out["variants"] = [sample(1, R.ORACLE_NET, seconds * 0.75), sample(1, R.ORACLE_UNIFORM, seconds * 0.4),
                       sample(ncpu, R.ORACLE_UNIFORM, seconds * 0.4)]
    if ncpu != 32:
        out["variants"].append(sample(32, R.ORACLE_NET, seconds * 0.5))      # the configuration of rounds 2-5 (32 threads), for continuity
    out["hardware_threads"], out["usable_cpus"] = os.cpu_count(), ncpu
    out["cores_note"] = "cores = threads run = usable_cpus (affinity mask and cgroup cpu.max quota); hardware_threads is what the box has"
    out["us_per_sim_uniform_1_thread"] = 1e6 / out["variants"][1]["value"]
    return out


def alone_and_tree(args, blob, hp, dev_index, kernel, waves=200):
    """Extra evidence, measured live after the timed region (not part of `value`), one slot group, every kernel class timed with HIP
    events.  (i) the shipped (free-running) schedule: the tower kernel alone -- with one group nothing runs beside a tower launch but
    the side streams' move step and background search -- and what a wave spends in each kernel class; (ii) the search-tree kernel
    against the HBM roofline on the LOCK-STEP schedule (one simulation per slot and launch, nothing overlapped: HIP-event durations
    are exclusive and comparable with rounds 2-5): algorithmic bytes of SURVEY.md §8(d) without the network's share -- 148 B per
    traversed node + per new leaf 16 (probe miss) + 136 (node write) + 64 (P, V written by the network, read by the expansion)."""
    import azhip

    def leg(lock_step):
        eng = azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, device=dev_index,
                           num_workers=args.slots, batch_size=args.slots, num_iters_per_turn=args.sims,
                           gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                           prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                           num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32, lock_step=1 if lock_step else 0)
        try:
            eng.net_set_params(blob)
            eng.selfplay_begin(-1, first_game_id=1 << 28)
            warm_up(eng, azhip.GAME_CONNECT_FOUR, args.sims, args.slots, lengths=HEADLINE_WARMUP_GAME_LENGTHS)
            s0 = eng.selfplay_stats()
            eng.prof_reset()
            eng.prof_enable(True)
            t0 = time.perf_counter()
            eng.selfplay_step(waves)
            s1 = eng.selfplay_stats()
            dt = time.perf_counter() - t0
            prof = eng.prof_get()
            name = eng.net_last_kernel()
            eng.prof_enable(False)
            eng.selfplay_end()
        finally:
            eng.close()
        return s0, s1, prof, name, dt
    s0, s1, prof, name, dt = leg(False)
    evals, sims = s1.leaf_evals - s0.leaf_evals, s1.simulations - s0.simulations
    alone = tower_roofline(azhip.GAME_CONNECT_FOUR, hp, False, name, prof["tower"], evals - (s1.evals_reused - s0.evals_reused), 1e9)
    alone.update(slot_groups=1, waves=waves, schedule="free-running",
                 us_per_wave_by_kernel_class={c: 1e3 * prof[c]["ms"] / waves for c in prof if prof[c]["launches"]},
                 class_note="select = the wave's tree launch (on the critical path), expand = the background search and move = the move step "
                            "(side streams, under the tower: their durations overlap it), tower, heads; HIP events around every launch slow the wave itself by ~2 %",
                 sims_per_slot_per_wave=sims / max(s1.slot_launches - s0.slot_launches, 1), ms_per_wave=1e3 * dt / waves)
    s0, s1, prof, _, _ = leg(True)
    evals, sims, trav = s1.leaf_evals - s0.leaf_evals, s1.simulations - s0.simulations, s1.nodes_traversed - s0.nodes_traversed
    tree_cls = [c for c in ("select", "compact", "expand") if prof[c]["launches"]]
    tree_ms = sum(prof[c]["ms"] for c in tree_cls)
    tree_bytes = 148.0 * trav + (16 + 136 + 64) * evals + 16.0 * (sims - evals)
    gbs = tree_bytes / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
    tree = {"bound": "hbm", "schedule": "lock step (one simulation per slot and launch; exclusive launch times)",
            "kernels": {c: {"launches": prof[c]["launches"], "avg_us": 1e3 * prof[c]["ms"] / prof[c]["launches"]} for c in tree_cls},
            "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
            "bytes_per_sim": tree_bytes / max(sims, 1), "avg_exploration_depth": trav / max(sims, 1), "slots": args.slots,
            "us_per_wave": 1e3 * tree_ms / waves, "traffic": None,
            "free_running_wave_launch_us": alone["us_per_wave_by_kernel_class"].get("select"),
            "free_running_sims_per_slot_per_wave": alone["sims_per_slot_per_wave"]}
    # what 4096 dependent chains can move at all (see TREE_LOAD_LATENCY_S): the ceiling this kernel is judged against
    ceiling = args.slots * TREE_LINE_BYTES / TREE_LOAD_LATENCY_S / 1e9
    tree["littles_law_ceiling_GBs"] = ceiling
    tree["littles_law_ceiling_frac"] = ceiling / PEAK_HBM_GBS
    tree["frac_of_littles_law_ceiling"] = gbs / ceiling if ceiling > 0 else None
    t = pmc_lookup("k_tree")
    if t is not None:
        tree["traffic"] = t[0] * args.slots / t[1]
        tree["traffic_over_algorithmic"] = tree["traffic"] / (tree_bytes / waves)
    return alone, tree


def headline_variant(args, blob, dev_index, waves, lock_step=False, env=None):
    """The headline workload once more under another switch (not part of `value`): same engine parameters, same warm-up, `waves` timed
    waves without HIP events.  env: environment overrides the library reads when the engine is created (AZHIP_EVAL_CACHE=0)."""
    import azhip
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        eng = azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, device=dev_index,
                           num_workers=args.slots, batch_size=args.slots // args.groups, num_iters_per_turn=args.sims,
                           gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                           prior_temperature=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                           num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32, lock_step=1 if lock_step else 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        eng.net_set_params(blob)
        eng.selfplay_begin(-1, first_game_id=1 << 25)
        warm_up(eng, azhip.GAME_CONNECT_FOUR, args.sims, args.slots, lengths=HEADLINE_WARMUP_GAME_LENGTHS)
        s0 = eng.selfplay_stats()
        t0 = time.perf_counter()
        eng.selfplay_step(waves)
        s1 = eng.selfplay_stats()
        dt = time.perf_counter() - t0
        eng.selfplay_end()
    finally:
        eng.close()
    sims, evals = s1.simulations - s0.simulations, s1.leaf_evals - s0.leaf_evals
    return {"value": sims / dt, "unit": "sims/s", "waves": waves, "ms_per_wave": 1e3 * dt / waves, "schedule": "lock step" if lock_step else "free-running",
            "sims_per_slot_per_wave": sims / max(s1.slot_launches - s0.slot_launches, 1),
            "unique_leaf_frac": (evals - (s1.evals_reused - s0.evals_reused)) / max(evals, 1), "env": env or {}}


def iterations_block(azhip, dev_index, iters=3, games=1024, workers=1024):
    """`iters` training iterations (train!'s loop body, src/training.jl:321-333) at the reference's shipped Connect-Four learning and MCTS
    parameters (games/connect-four/params.jl:5-75; tools/iterations.py is the stand-alone form) with a REDUCED number of games per
    iteration, so that the driver's line shows whether the loop learns beyond the first epoch: learning_status of the
    whole data set before / after each iteration's batch_updates!, the arena result, the replacement decision, and the evaluation
    cache's hit rate as the network trains (unique_leaf_frac of the self-play phase)."""
    from azhip.training import SelfPlayParams, train_iteration
    from azhip import engine as E
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=5, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    best = azhip.ResNet(gspec, hp, seed=1)
    cur = best.copy_()
    mcts = azhip.MctsParams(num_iters_per_turn=600, cpuct=2.0, prior_temperature=1.0, temperature=azhip.PLSchedule([0, 20, 30], [1.0, 1.0, 0.3]),
                            dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0)
    sp = SelfPlayParams(mcts=mcts, sim=azhip.SimParams(num_games=games, num_workers=workers, batch_size=workers, use_gpu=True, reset_every=2,
                                                      flip_probability=0.0, alternate_colors=False))
    amcts = azhip.MctsParams(num_iters_per_turn=600, cpuct=2.0, prior_temperature=1.0, temperature=azhip.ConstSchedule(0.2), dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0)
    arena = azhip.ArenaParams(mcts=amcts, sim=azhip.SimParams(num_games=128, num_workers=128, batch_size=128, use_gpu=True, reset_every=2,
                                                               flip_probability=0.5, alternate_colors=True), update_threshold=0.05)
    lp = azhip.LearningParams(use_position_averaging=True, samples_weighing_policy=azhip.LOG_WEIGHT, batch_size=1024, loss_computation_batch_size=1024,
                              optimiser=azhip.Adam(lr=2e-3), l2_regularization=1e-4, nonvalidity_penalty=1.0, min_checkpoints_per_epoch=1,
                              max_batches_per_checkpoint=2000, num_checkpoints=1)
    mem = azhip.MemoryBuffer(gspec, 400000, device=dev_index)
    rows = []
    try:
        for it in range(iters):
            t0 = time.perf_counter()
            cur, best, rep, lr = train_iteration(gspec, cur, best, mem, sp, lp, arena, seed=1 + it)
            ck = lr.checkpoints[-1]
            st = next((e.selfplay_stats() for k, e in E._cache.items() if k[0] == "selfplay" and e._h is not None), None)
            rows.append({"iteration": it + 1, "seconds": time.perf_counter() - t0, "memory_size": rep.memory_size, "samples_per_sec": rep.samples_gen_speed,
                         "sims_per_sec_self_play": rep.samples_gen_speed * 600,
                         "unique_leaf_frac": (st.leaf_evals - st.evals_reused) / max(st.leaf_evals, 1) if st is not None else None,
                         "leaf_evals_per_sim": st.leaf_evals / max(st.simulations, 1) if st is not None else None,
                         "optimiser_steps": int(len(lr.losses)), "status_before": _status(lr.initial_status), "status_after": _status(ck.status_after),
                         "arena_avgr": float(ck.evaluation.avgr), "nn_replaced": bool(ck.nn_replaced)})
    finally:
        mem.close()
    return {"workload": "%d training iterations, %d self-play games each on %d workers (600 sims, ResNet 5x128, reset_every 2), shipped learning parameters "
                        "(games/connect-four/params.jl:46-75), arena 128 games; random initial weights" % (iters, games, workers),
            "iterations": rows,
            "L_before_after": [[r["status_before"]["L"], r["status_after"]["L"]] for r in rows],
            "arena_avgr": [r["arena_avgr"] for r in rows], "nn_replaced": [r["nn_replaced"] for r in rows],
            "unique_leaf_frac": [r["unique_leaf_frac"] for r in rows]}


def gather_leg(azhip, blob, dev_index, rank, world, games_per_rank=512, nsims=48):
    """RCCL trace gather (BASELINE configs[2]'s exchange step) on a short phase: `games_per_rank` Connect-Four games per
    rank with global game ids, device-only; then one collective puts every rank's samples into every rank's memory."""
    from azhip import comm
    trace("exchange: id broadcast + communicator")
    c = comm.Comm(dev_index, rank, world, comm.torch_broadcast_id(rank))
    trace("exchange: communicator up")
    eng = azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, device=dev_index, num_workers=games_per_rank,
                       batch_size=games_per_rank, num_iters_per_turn=nsims, cpuct=2.0, dirichlet_noise_eps=0.25,
                       dirichlet_noise_alpha=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)), reset_every=1, seed=1,
                       num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    if rank == 0:
        eng.net_set_params(blob)
    c.broadcast_params(eng, root=0)                                 # the weights reach the other ranks over RCCL
    trace("exchange: weights broadcast")
    mem = azhip.MemoryBuffer(azhip.ConnectFourSpec(), 1 << 22, device=dev_index)
    eng.selfplay_run(games_per_rank, first_game_id=rank * games_per_rank, device_only=True)
    trace("exchange: phase played")
    c.gather_push(eng, mem, 1.0)                                    # warm-up of the communicator's rings
    mem.empty()
    t0 = time.perf_counter()
    gs = c.gather_push(eng, mem, 1.0)
    dt = time.perf_counter() - t0
    ver, path = comm.version()
    out = {"collective": "ncclAllGather (RCCL) of device-resident az_move_rec / az_game_rec + push_trace! on the device",
           "library": path, "nccl_version_code": ver,
           "ranks": int(gs.ranks), "world": world, "rendezvous": "torch.distributed/%s" % __import__("torch").distributed.get_backend(), "games": gs.games, "samples": gs.moves, "bytes_received_per_rank": gs.bytes,
           "gather_ms": gs.gather_ms, "gather_and_push_ms": gs.total_ms, "wall_ms": 1e3 * dt, "memory_length": len(mem),
           "GB_per_s_per_rank": gs.bytes / max(gs.gather_ms, 1e-9) / 1e6}
    mem.close()
    eng.close()
    c.close()
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same flags>` (one rank per GPU; simulate_distributed's one process per
    worker, src/simulations.jl:252-290).  Refuses loudly when the node has fewer than N devices -- unless the test-only
    transport is selected (AZHIP_RCCL_LIB, tests/rccl_stub: RCCL itself refuses two ranks on one device), in which case the
    ranks share the devices there are.  Rank 0's JSON line is the child's stdout, passed through unchanged."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (libazhip.so has no CPU fallback)")
    if n > ndev and not os.environ.get("AZHIP_RCCL_LIB"):
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible; one rank per GPU (RCCL refuses several ranks on a device)" % (n, ndev))
    with socket.socket() as so:                                      # a free rendezvous port on the loopback
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), AZ_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--slots", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--groups", type=int, default=1, help="slot groups = num_workers / batch_size: 1 (default since round 5) = one network launch per wave for all 4096 slots (batch_size 4096, BASELINE's 'batch 4096'); 2 = two interleaved half-batches, the tree kernels of one under the network of the other -- the better form while every leaf went to the network (rounds 1-4: 4.9 vs 4.85 M sims/s), the worse one since the evaluation cache answers ~40 %% of the leaves and a half-batch no longer fills the chip (8.0 vs 8.5 M sims/s, profiles/r5)")
    ap.add_argument("--lock-step", action="store_true", help="az_engine_cfg.lock_step = 1: the schedule of rounds 1-5 (one simulation per slot and wave, moves in rounds)")
    ap.add_argument("--no-variants", action="store_true", help="skip the headline variants (2000-wave run, evaluation cache off, lock step: ~30 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra configurations (whole phase, C3, C4 Mancala, bf16 10x128, 128 workers) reported under `extra`")
    ap.add_argument("--no-iteration", action="store_true", help="skip extra.iteration (one whole training iteration at the reference's shipped Connect-Four parameters, ~1-2 min)")
    ap.add_argument("--iteration", action="store_true", help="run ONLY the headline and extra.iteration (skips the other extras and the CPU baseline)")
    ap.add_argument("--backend", default="gloo", help="torch.distributed backend of the RENDEZVOUS for N > 1: a barrier, two scalar reductions and the 128-byte RCCL id are all that goes through it, so gloo (CPU) is the default and the process holds exactly ONE RCCL instance, the one libazhip.so loads for az_comm_*; nccl = torch's bundled RCCL as well")
    ap.add_argument("--headline-only", action="store_true", help="only the timed region (no one-group leg, no extras, no CPU baseline): the run whose rocprofv3 kernel stats are comparable with roofline.avg_launch_ms (profiles/r4/bench_headline_kernel_stats.csv)")
    ap.add_argument("--no-prof", action="store_true", help="do not wrap launches in HIP events")
    ap.add_argument("--prof-all", action="store_true", help="time every kernel class (default: only the dominant kernel, k_tower)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)                                       # does not return
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%s: the two must agree" % (args.gpus, os.environ["WORLD_SIZE"]))

    import torch
    import azhip
    from azhip.network import ResNetHP, random_params

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    if world > ndev and not os.environ.get("AZHIP_RCCL_LIB"):
        raise SystemExit("bench.py: %d ranks but %d device(s) visible; one rank per GPU (RCCL refuses several ranks on a device)" % (world, ndev))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    elif os.environ.get("AZ_BENCH_GATHER"):
        # single-GPU rehearsal of the N > 1 exchange leg (a world of one rank; the collective code path is the same)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500), rank=0, world_size=1)
        args.backend = "gloo"
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
                       num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32, lock_step=1 if args.lock_step else 0)
    eng.net_set_params(blob)
    dev_name, ncu, hbm = eng.device_info()
    eng.selfplay_begin(-1, first_game_id=rank * (1 << 24))
    trace("engine ready (%d slots, %d group(s)), world %d" % (args.slots, args.groups, world))

    def barrier():
        if dist is not None:
            dist.barrier()

    # steady state whatever --warmup says: the state of a LONG phase -- slots at every stage of a game, the evaluation cache
    # holding what earlier games left in it -- not 4096 games in lock step through the same opening (mixing_warmup)
    warm = warm_up(eng, azhip.GAME_CONNECT_FOUR, args.sims, args.slots, extra_waves=args.warmup, lengths=HEADLINE_WARMUP_GAME_LENGTHS)
    s0 = eng.selfplay_stats()
    trace("warm-up done (%d waves)" % warm)
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
    trace("timed region done (%.3f s)" % (t1 - t0))
    prof = eng.prof_get() if not args.no_prof else None
    eng.prof_enable(False)
    eng_kernel = eng.net_last_kernel()

    elapsed = t1 - t0
    sims = s1.simulations - s0.simulations
    evals = s1.leaf_evals - s0.leaf_evals
    trav = s1.nodes_traversed - s0.nodes_traversed
    moves = s1.moves - s0.moves
    reused = s1.evals_reused - s0.evals_reused
    slot_launches = s1.slot_launches - s0.slot_launches
    local_evals = evals - reused                                     # boards this rank's network evaluated (the rest came from the evaluation cache)
    local_elapsed = elapsed
    per_rank = [sims / elapsed]
    if dist is not None:
        pr = [torch.zeros(2, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(pr, torch.tensor([sims, elapsed], dtype=torch.float64, device=red_dev))
        per_rank = [float(x[0] / x[1]) for x in pr]
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([sims, evals, trav, moves, reused], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        sims, evals, trav, moves, reused = [float(x) for x in c.tolist()]
    eng.selfplay_end()
    eng.close()                        # frees its ~10 GB before the extra legs build their own engine

    # After the timed region (N > 1): the exchange step of the path -- every rank plays a short bounded phase device-only
    # and az_comm_gather_push all-gathers the records over RCCL straight into every rank's device replay memory
    # (simulate_distributed's fetch + push_trace!, src/simulations.jl:280-289, src/memory.jl:74-87).  Not part of `value`.
    gather, gather_hung = None, False
    if dist is not None and not os.environ.get("AZ_BENCH_NO_GATHER"):
        # the headline number must survive a failing or hanging extra leg: it runs on a helper thread with a deadline
        box = {}

        def run_gather():
            try:
                torch.cuda.set_device(dev_index)                    # the current device is per thread
                box["r"] = gather_leg(azhip, blob, dev_index, rank, world)
            except Exception as ex:
                box["r"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if rank == 0:
            # the exchange leg is the first time real RCCL runs with W > 1: should the process die inside it, the measured number
            # is already on stderr (the ONE JSON line on stdout comes after the leg, with its `gather` object)
            print("[bench] provisional (before the exchange leg): %s" % json.dumps({"metric": "self-play MCTS sims/sec", "value": sims / elapsed, "unit": "sims/s",
                  "n_gpus": world, "steps": args.steps, "ms_per_step": 1e3 * elapsed / args.steps, "sims_per_sec_by_rank": per_rank}), file=sys.stderr, flush=True)
        trace("exchange leg starts")
        th = threading.Thread(target=run_gather, daemon=True)
        th.start()
        th.join(240.0)
        gather_hung = th.is_alive()
        trace("exchange leg %s" % ("did NOT finish" if gather_hung else "done: %s" % json.dumps(box.get("r"))[:200]))
        gather = {"error": "the exchange leg did not finish within 240 s"} if gather_hung else box.get("r")

    if rank == 0:
        out = {
            "metric": "self-play MCTS sims/sec (Connect-Four, %d parallel games per GPU)" % args.slots,
            "value": sims / elapsed, "unit": "sims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_waves_run": warm,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Connect-Four self-play, %d sims/move, %d parallel games per GPU, ResNet 5x64 fp32 "
                                   "(heads 32/32), cpuct 2, eps 0.25, alpha 1, PLSchedule([0,20,30],[1,1,.3]), reset_every 1, num_workers/batch_size = %d; "
                                   "step = one search wave (%s); evaluation cache %s" % (args.sims, args.slots, args.groups,
                                       "lock step: 1 simulation per slot" if args.lock_step or os.environ.get("AZHIP_FREE_RUN") == "0" else "free-running: every slot up to its next network evaluation, sims_per_slot_per_wave simulations on average",
                                       "off (AZHIP_EVAL_CACHE=0)" if os.environ.get("AZHIP_EVAL_CACHE") == "0" else "on"),
                       "schedule": "lock step" if args.lock_step or os.environ.get("AZHIP_FREE_RUN") == "0" else "free-running",
                       "slots_per_gpu": args.slots, "sims_per_move": args.sims, "slot_groups": args.groups, "leaves_per_network_launch": args.slots // args.groups, "parallelism": "dp%d (games sharded, no collective in the timed region)" % world,
                       "device": dev_name, "compute_units": ncu},
            "sims_per_sec_per_gpu": sims / elapsed / world,
            "sims_per_sec_by_rank": per_rank,                         # each rank's own simulations / its own time
            "launcher": "self (python bench.py --gpus N -> torch.distributed.run)" if os.environ.get("AZ_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if dist is not None and world > 1 else "single process"),
            "samples_per_sec": moves / elapsed,
            "avg_exploration_depth": trav / max(sims, 1),
            "leaf_evals_per_sim": evals / max(sims, 1),
            # oracle calls the network evaluated / oracle calls: the others were answered by the engine's evaluation cache (the same
            # state evaluated before for another slot or in an earlier wave; bit-identical answers, tests/test_eval_cache_gpu.py)
            "unique_leaf_frac": (evals - reused) / max(evals, 1),
            # simulations a slot completes per wave (= per launch of the tree kernel): 1 in lock step
            "sims_per_slot_per_wave": sims / max(slot_launches * (world if dist is not None else 1), 1),
        }
        if prof is not None:
            out["roofline"] = tower_roofline(azhip.GAME_CONNECT_FOUR, hp, False, eng_kernel, prof["tower"], local_evals, local_elapsed)
            out["roofline"]["kernel_ms_per_step"] = out["roofline"]["exclusive_ms"] / args.steps
            # the same executed FLOPs over the WALL time of the timed region (tree kernels, dense heads, launch gaps included)
            out["roofline"]["achieved_over_wall"] = out["roofline"]["achieved"] * out["roofline"]["exclusive_ms"] / out["roofline"]["wall_ms"]
            out["roofline"]["frac_over_wall"] = out["roofline"]["achieved_over_wall"] / out["roofline"]["peak"]
            if args.groups > 1:
                out["roofline"]["time_basis"] = ("wall time of the timed region: the slot groups' tower launches run side by side, so avg_launch_ms x launches "
                                                 "(launch_ms_sum) exceeds the wall time and a per-launch average -- HIP events here, rocprofv3 in profiles/ -- is NOT "
                                                 "exclusive time; the same kernel alone, one launch at a time: roofline_kernel_alone (rocprof cross-check: "
                                                 "profiles/r5/headline_one_group_kernel_stats.csv)")
            if out["roofline"]["traffic"] is None:
                out["roofline"]["traffic"] = pmc_traffic(eng_kernel, out["roofline"]["avg_boards_per_launch"])
            out["kernel_ms"] = {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]}
        if prof is not None and world == 1 and not args.headline_only:
            alone, tree = alone_and_tree(args, blob, hp, dev_index, eng_kernel)
            out["roofline_kernel_alone"] = alone
            out["roofline_tree"] = tree
        if world == 1 and not args.headline_only and not args.no_variants and not args.iteration:
            # The driver times 20 waves (~18 ms): the same workload over 2000 waves gives that number its error bar; and the two switches a
            # reader wants beside `value` -- the evaluation cache off (every oracle call a network evaluation: the kernel-only comparable of
            # rounds 1-4) and the lock-step schedule of rounds 1-5
            try:
                out["value_long"] = headline_variant(args, blob, dev_index, 2000)
                out["value_cache_off"] = headline_variant(args, blob, dev_index, 1000, env={"AZHIP_EVAL_CACHE": "0"})
                out["value_lock_step"] = headline_variant(args, blob, dev_index, 2000, lock_step=True)
            except Exception as ex:
                out["value_variants_error"] = "%s: %s" % (type(ex).__name__, ex)
        if gather is not None:
            if world > 1 and "error" not in gather and gather.get("ranks") != world:
                gather["error"] = "the exchange saw %s ranks, the job has %d" % (gather.get("ranks"), world)
            out["gather"] = gather
        if world == 1 and not args.no_extras and not args.headline_only:
            # The rest of DESIGN.md §0's table, measured here so that the driver's line carries it (bounded: ~40 s in all).
            # None of it is part of `value`.  A failing block reports its error instead of taking the headline with it.
            mk = ResNetHP
            blocks = [
                ("whole_phase", lambda: whole_phase(azhip, dev_index, blob, hp, args.slots, args.sims, args.groups)),
                ("c3", lambda: steady_block(azhip, dev_index, "c3", azhip.GAME_CONNECT_FOUR, 4096, args.groups, 600, hp, 200,
                                            note="BASELINE configs[2] per GPU (games/connect-four/params.jl:25)")),
                ("c2_5x128", lambda: steady_block(azhip, dev_index, "c2_5x128", azhip.GAME_CONNECT_FOUR, 4096, 2, 600,
                                                  mk(num_blocks=5, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32), 120,
                                                  note="the reference's shipped network and sims/move (games/connect-four/params.jl:7-30) at the BASELINE batch")),
                ("c4_mancala", lambda: steady_block(azhip, dev_index, "c4_mancala", azhip.GAME_MANCALA, 8192, 1, 800, hp, 200, max_moves=256,
                                                    note="BASELINE configs[3] (games/mancala/params.jl:23-29), bug-compatible flip_colors")),
                ("bf16_10x128", lambda: steady_block(azhip, dev_index, "bf16_10x128", azhip.GAME_CONNECT_FOUR, 4096, 2, 400,
                                                     mk(num_blocks=10, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32), 200, bf16=True,
                                                     note="BASELINE configs[4]'s network on Connect-Four boards")),
                ("workers_128_5x128", lambda: steady_block(azhip, dev_index, "workers_128_5x128", azhip.GAME_CONNECT_FOUR, 128, 1, 600,
                                                           mk(num_blocks=5, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32), 400,
                                                           note="the reference's shipped self-play parameters (games/connect-four/params.jl:7-30: 128 workers, 5x128)")),
            ]
            blocks.append(("c5_host_stepped", lambda: host_stepped_c5()))
            # SURVEY §8(f) rows, driver-observed
            blocks += [("trainer_5x64", lambda: trainer_block(azhip, dev_index, 64)), ("trainer_5x128", lambda: trainer_block(azhip, dev_index, 128)),
                       ("memory_pipeline", lambda: memory_block(azhip, dev_index)), ("arena_128", lambda: arena_block(azhip, dev_index))]
            if not args.no_iteration:
                blocks.append(("iteration", lambda: iteration_block(azhip, dev_index, workers=int(os.environ.get("AZ_BENCH_ITER_WORKERS", "4096")))))
                blocks.append(("iterations", lambda: iterations_block(azhip, dev_index)))
            if args.iteration:
                blocks = [b for b in blocks if b[0] in ("iteration", "iterations")]
            if os.environ.get("AZ_BENCH_ONLY"):                      # A/B aid: a comma-separated subset of the extra blocks
                blocks = [b for b in blocks if b[0] in os.environ["AZ_BENCH_ONLY"].split(",")]
            out["extra"] = {}
            for name, fn in blocks:
                trace("extra block %s" % name)
                try:
                    out["extra"][name] = fn()
                except Exception as ex:
                    out["extra"][name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            # SURVEY §8(d)'s metric is the PHASE (all games from the empty board to the end, move steps, refills, drain): its number, and
            # what one training iteration learns, as top-level keys (VERDICT r4 #2c: a truncated `extra` must not hide them)
            wp = out["extra"].get("whole_phase", {})
            if "value" in wp:
                out["phase"] = {"sims_per_sec": wp["value"], "games": wp.get("games"), "positions": wp.get("positions"), "seconds": wp.get("seconds"),
                                "unique_leaf_frac": wp.get("unique_leaf_frac"), "leaf_evals_per_sim": wp.get("leaf_evals_per_sim"),
                                "tower_frac_of_fp32_mfma_peak": wp.get("roofline", {}).get("frac")}
            it = out["extra"].get("iteration", {})
            if "learning_status" in it:
                out["learning"] = it["learning_status"]
        if world == 1 and not args.no_cpu_baseline and not args.iteration and not args.headline_only:
            out["cpu_baseline"] = cpu_baseline(blob, hp, args.sims)
        if "extra" in out:
            # The figures a reader of the line's END should not have to dig for (the driver keeps the tail of stdout): the headline,
            # SURVEY 8(d)'s whole phase, the shipped network at the BASELINE batch, what one iteration learnt, the tree kernel.
            ex = out["extra"]

            def pick(name, *keys):
                b = ex.get(name, {})
                return {k: (b.get("roofline", {}).get(k[9:]) if k.startswith("roofline.") else b.get(k)) for k in keys} if "error" not in b else {"error": b["error"]}
            out["summary"] = {
                "headline_sims_per_sec": out["value"], "headline_unique_leaf_frac": out["unique_leaf_frac"], "headline_sims_per_slot_per_wave": out["sims_per_slot_per_wave"],
                "value_long": (out.get("value_long") or {}).get("value"), "value_cache_off": (out.get("value_cache_off") or {}).get("value"),
                "value_lock_step": (out.get("value_lock_step") or {}).get("value"),
                "iterations": {k: (ex.get("iterations") or {}).get(k) for k in ("L_before_after", "arena_avgr", "nn_replaced", "unique_leaf_frac", "error") if (ex.get("iterations") or {}).get(k) is not None},
                "headline_tower_frac_of_fp32_mfma_peak": out.get("roofline", {}).get("frac"),
                "kernel_alone_frac": out.get("roofline_kernel_alone", {}).get("frac"), "kernel_alone_avg_launch_ms": out.get("roofline_kernel_alone", {}).get("avg_launch_ms"),
                "tree_us_per_wave": out.get("roofline_tree", {}).get("us_per_wave"), "tree_frac_of_hbm_peak": out.get("roofline_tree", {}).get("frac"),
                "phase": out.get("phase"),
                "c2_5x128": pick("c2_5x128", "value", "ms_per_step", "unique_leaf_frac", "roofline.frac", "roofline.kernel"),
                "workers_128_5x128": pick("workers_128_5x128", "value", "roofline.frac", "roofline.kernel"),
                "c4_mancala": pick("c4_mancala", "value", "engine_device_GB", "roofline.frac"),
                "iteration": pick("iteration", "seconds", "sims_per_sec_self_play", "arena_avgr", "nn_replaced", "phases_share"),
                "learning": out.get("learning"),
                "cpu_baseline_sims_per_sec": out.get("cpu_baseline", {}).get("value"),
            }
            if "roofline" in out:
                # ... and as flat scalars of `roofline` (the driver's record keeps that object's scalar fields): VERDICT r4 #2c
                sm, rf = out["summary"], out["roofline"]
                rf["also_kernel_alone_frac"], rf["also_kernel_alone_avg_launch_ms"] = sm["kernel_alone_frac"], sm["kernel_alone_avg_launch_ms"]
                rf["also_tree_us_per_wave"], rf["also_tree_frac_of_hbm_peak"] = sm["tree_us_per_wave"], sm["tree_frac_of_hbm_peak"]
                ph = sm["phase"] or {}
                rf["also_phase_sims_per_sec"], rf["also_phase_unique_leaf_frac"], rf["also_phase_tower_frac"] = ph.get("sims_per_sec"), ph.get("unique_leaf_frac"), ph.get("tower_frac_of_fp32_mfma_peak")
                rf["also_c2_5x128_sims_per_sec"], rf["also_c2_5x128_tower_frac"] = sm["c2_5x128"].get("value"), sm["c2_5x128"].get("roofline.frac")
                rf["also_iteration_seconds"] = sm["iteration"].get("seconds")
                rf["also_value_long"], rf["also_value_cache_off"], rf["also_value_lock_step"] = sm["value_long"], sm["value_cache_off"], sm["value_lock_step"]
                rf["also_sims_per_slot_per_wave"] = out["sims_per_slot_per_wave"]
                lb, la = ((sm["learning"] or {}).get("before") or {}), ((sm["learning"] or {}).get("after") or {})
                rf["also_learning_loss_before"], rf["also_learning_loss_after"] = lb.get("L"), la.get("L")
        print(json.dumps(out), flush=True)
    trace("line printed" if rank == 0 else "done")
    if gather_hung:
        os._exit(0)                                                 # a helper thread is stuck in a collective: no orderly shutdown
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
