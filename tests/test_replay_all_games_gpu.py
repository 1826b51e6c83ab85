"""EVERY game of a BASELINE phase against the oracle, not a sample (SURVEY.md §7 hard part 3: replay mode).

tests/test_baseline_configs_gpu.py replays 64 of a phase's games with the oracle's OWN fp32 network on the CPU (about a second of
CPU network per move): rare per-slot events -- a tag collision in the slot's hash table, a probe that wraps, a node pool near its
bound -- show up only in the games that happen to be sampled.  Here the oracle's trees (oracle/azref.c: the same `simulate` loop,
the same recursive run_simulation!) consume the DEVICE network's P / V through az_net_evaluate_keys -- which tests/test_net.py
holds bit-identical to the in-loop tower for every tower form -- so a whole phase costs the CPU only its tree walks, on all host
threads, and every game of the phase is compared record by record: state keys, visit counts, actions, rewards, node counts,
simulation / traversal counters, final states.  What this leg does NOT check is the network itself (the 64-game leg and
tests/test_net.py do); what it adds is the tree, the move step, the RNG streams and the capacity machinery of all 4096 / 8192
slots.  tests/test_replay_mode.py (CPU) shows replay mode neutral: fed with the oracle's own answers it reproduces the direct run.

A phase's evaluations repeat: the replay asks the device only for states it has not seen (`evaluated`), the trees consume
`oracle_calls` -- their ratio is the share of distinct states among a phase's leaf evaluations (written to gpurun_out/ when that
directory exists: the measurement VERDICT r4 #3 asks for before any evaluation cache is built)."""
import ctypes as C
import json
import os
import time

import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu

C4_SCHED = ((0, 20, 30), (1.0, 1.0, 0.3))          # games/connect-four/params.jl:24-30, games/mancala/params.jl:23-29
GAME_DT = np.dtype([("game_id", "<i4"), ("slot", "<i4"), ("num_moves", "<i4"), ("first_move", "<i4"), ("nodes", "<i8"),
                    ("total_simulations", "<i8"), ("total_nodes_traversed", "<i8"), ("final_key", "<u8", 2)])
HERE = os.path.dirname(os.path.abspath(__file__))


def _views(games, moves, ng, nm):
    assert C.sizeof(games[0]) == GAME_DT.itemsize and C.sizeof(moves[0]) == 64
    g = np.frombuffer(memoryview(games).cast("B"), dtype=GAME_DT, count=ng)
    m = np.frombuffer(memoryview(moves).cast("B"), dtype=np.uint8, count=nm * 64).reshape(nm, 64)
    return g, m


def _chunk_size(nsims, plies, reset_every, want):
    """workers per replay so that their trees fit the host: <= nsims * plies * reset_every nodes per worker, 240 B each, in a table
    at most half full and doubled while it grows"""
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 32 << 30
    per = nsims * plies * max(1, reset_every) * 240 * 3
    return int(max(64, min(want, (avail // 2) // per)))


def _compare(hg, hm, rg, rm, what):
    """hg / hm: the device's records of the games rg describes (same ids); every field but `slot` and `first_move`"""
    assert len(hg) == len(rg)
    for f in ("game_id", "num_moves", "nodes", "total_simulations", "total_nodes_traversed", "final_key"):
        bad = np.nonzero((hg[f] != rg[f]).reshape(len(hg), -1).any(axis=1))[0]
        assert bad.size == 0, "%s: %d games differ in %s, first: id %d device %r oracle %r" % (
            what, bad.size, f, int(hg["game_id"][bad[0]]), hg[f][bad[0]], rg[f][bad[0]])
    hidx = np.concatenate([np.arange(a, a + n) for a, n in zip(hg["first_move"], hg["num_moves"])])
    ridx = np.concatenate([np.arange(a, a + n) for a, n in zip(rg["first_move"], rg["num_moves"])])
    diff = np.nonzero((hm[hidx] != rm[ridx]).any(axis=1))[0]
    if diff.size:
        k = int(diff[0])
        gi = int(np.searchsorted(np.cumsum(hg["num_moves"]), k, side="right"))
        raise AssertionError("%s: %d move records differ, first in game %d: device %r oracle %r" % (
            what, diff.size, int(hg["game_id"][gi]), hm[hidx[k]].view("<i4").tolist(), rm[ridx[k]].view("<i4").tolist()))


def _phase_vs_replay(name, game_hip, game_ref, slots, groups, nsims, first_id, num_games, plies, reset_every=1, flip=0.0,
                     blocks=5, filters=64):
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=blocks, num_filters=filters, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(game_hip, hp, seed=2026)
    t0 = time.time()
    with azhip.Engine(game=game_hip, oracle=azhip.ORACLE_RESNET, num_workers=slots, batch_size=slots // groups,
                      num_iters_per_turn=nsims, gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                      prior_temperature=1.0, temperature=C4_SCHED, reset_every=reset_every, flip_probability=flip, seed=1,
                      num_blocks=blocks, num_filters=filters, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        games, moves, ng, nm, stats = e.selfplay_run(num_games, first_game_id=first_id)
        t_dev = time.time() - t0
        assert ng == num_games and stats.aborted_games == 0
        hg, hm = _views(games, moves, ng, nm)
        assert np.array_equal(hg["game_id"], np.arange(first_id, first_id + num_games))
        kw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=C4_SCHED[0], temp_ys=C4_SCHED[1], reset_every=reset_every,
                  seed=1, flip_probability=flip)
        # reset_every = 1: a game depends on its id alone, so the phase is replayed in chunks of workers that fit the host's memory
        # (each chunk's games on fresh workers, exactly as the device's slots played them: one game per slot).  Otherwise the
        # worker <-> game assignment matters and all workers are replayed together, with the assignment the device reports.
        one_per_slot = reset_every == 1 and num_games <= slots
        chunk = _chunk_size(nsims, plies, reset_every, num_games) if one_per_slot else num_games
        ev = R.Evals(26)                                             # 64 M states x 64 B: forgets everything when 60 % full
        # the evaluator is the device library's az_net_evaluate_keys itself, called from the replay loop with no Python in between
        evaluator = (C.cast(azhip._lib.lib().az_net_evaluate_keys, C.c_void_p).value, e._h.value)
        t1 = time.time()
        tot = dict(evaluated=0, oracle_calls=0, steps=0)
        secs = {}
        for c0 in range(0, num_games, chunk):
            n = min(chunk, num_games - c0)
            # which worker played which game is a race in the reference (util.jl:181-188): the replay takes the outcome the device reports
            rg_, rm_, rnm, info = R.replay(game_ref, evaluator, n, n if one_per_slot else slots, nsims, evals=ev,
                                           first_game_id=first_id + c0, assignment=None if one_per_slot else hg["slot"].astype(np.int32), **kw)
            for k in tot:
                tot[k] += info[k]
            for k, v in info["seconds"].items():
                secs[k] = round(secs.get(k, 0.0) + v, 2)
            rg, rm = _views(rg_, rm_, n, rnm)
            _compare(hg[c0:c0 + n], hm, rg, rm, "%s games %d..%d" % (name, first_id + c0, first_id + c0 + n - 1))
        t_rep = time.time() - t1
        cnt = ev.counters()
        ev.close()
    assert tot["oracle_calls"] == stats.leaf_evals                   # the trees consumed exactly the evaluations the device computed
    rec = dict(case=name, games=num_games, moves=int(nm), simulations=int(stats.simulations), leaf_evals=int(stats.leaf_evals),
               distinct_states_evaluated=tot["evaluated"], distinct_over_leaf_evals=tot["evaluated"] / max(1, stats.leaf_evals),
               table_wipes=cnt["wipes"], replay_steps=tot["steps"], replay_chunk=chunk, seconds_device_phase=round(t_dev, 2),
               seconds_replay=round(t_rep, 2), seconds_replay_parts=secs, replay_threads=info["threads"], host_threads=os.cpu_count())
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "replay_all_games.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    print(json.dumps(rec))
    return rec


def test_config2_every_one_of_the_4096_games():
    """BASELINE configs[1]: Connect-Four, 4096 slots in two groups, 400 sims/move, ResNet 5x64 -- all 4096 games"""
    import azhip
    r = _phase_vs_replay("C2", azhip.GAME_CONNECT_FOUR, R.C4, 4096, 2, 400, 0, 4096, 42)
    assert r["simulations"] == 400 * r["moves"]


def test_config3_shard_every_game_with_flips_and_trees_kept_over_two_games():
    """BASELINE configs[2] as rank 3 of 8 sees it (600 sims/move, global ids from 3 x 4096), with the two options the shipped
    parameters leave off switched ON: flip_probability 0.5 (play.jl:305-307) and reset_every 2 (a worker's tree persists over
    two games, simulations.jl:235-237) -- 2048 games on 1024 slots, so every worker plays two and all of them are compared"""
    import azhip
    _phase_vs_replay("C3 flips, reset_every 2", azhip.GAME_CONNECT_FOUR, R.C4, 1024, 2, 600, 3 * 4096, 2048, 42, reset_every=2, flip=0.5)


def test_config3_shard_every_one_of_the_4096_games():
    """BASELINE configs[2], one rank's whole 4096-game shard at 600 sims/move (shipped options: no flips, reset_every 1)"""
    import azhip
    _phase_vs_replay("C3 shard", azhip.GAME_CONNECT_FOUR, R.C4, 4096, 2, 600, 3 * 4096, 4096, 42)


def test_config4_mancala_every_one_of_the_8192_games():
    """BASELINE configs[3]: Mancala (free turns, variable masks, bug-compatible flip_colors), 8192 slots, 800 sims/move"""
    import azhip
    _phase_vs_replay("C4 Mancala", azhip.GAME_MANCALA, R.MANCALA, 8192, 2, 800, 0, 8192, 128)
