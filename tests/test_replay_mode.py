"""Replay mode of the CPU oracle (SURVEY.md §7 hard part 3), checked on the CPU.

Replay mode = the oracle's `simulate` (same lock-step loop, same tree code: oracle/azref.c `azr_sim_step`) with every oracle(state)
answer supplied by the caller.  Before it is used to check the device (tests/test_replay_all_games_gpu.py: every game of a BASELINE
phase against the device network's P / V) it has to be shown neutral: fed with the oracle's OWN answers -- the hash oracle, the
fp32 network -- it must reproduce the direct run record by record, whatever the table size (forgetting answers), the number of
workers, flips, or tree persistence (reset_every)."""
import ctypes as C

import numpy as np
import pytest

import azref as R

SCHED = ((0, 4, 8), (1.0, 1.0, 0.3))


def _bytes(games, moves, ng, nm):
    return (bytes(memoryview(games).cast("B")[:ng * C.sizeof(R.GameRec)]),
            bytes(memoryview(moves).cast("B")[:nm * C.sizeof(R.MoveRec)]))


@pytest.mark.parametrize("game,nsims,games,workers,reset_every,flip", [
    (R.C4, 40, 24, 24, 1, 0.0),
    (R.C4, 40, 24, 8, 2, 0.5),           # workers take several games, trees persist over two of them, every other turn flipped
    (R.TTT, 30, 40, 16, 3, 1.0),         # seven symmetries, every turn flipped
    (R.MANCALA, 50, 12, 12, 1, 0.0),     # free turns (pswitch = false), variable masks
])
def test_replay_fed_with_the_hash_oracle_equals_the_direct_run(game, nsims, games, workers, reset_every, flip):
    kw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=SCHED[0], temp_ys=SCHED[1], reset_every=reset_every, seed=5,
              first_game_id=100, flip_probability=flip)
    dg, dm, dnm = R.simulate(game, R.ORACLE_HASH, games, workers, nsims, **kw)
    calls = []

    def evaluate(keys):
        calls.append(len(keys))
        assert len({(int(a), int(b)) for a, b in keys}) == len(keys)          # distinct states only
        return R.hash_oracle_keys(game, keys)
    rg, rm, rnm, info = R.replay(game, evaluate, games, workers, nsims, **kw)
    assert rnm == dnm and _bytes(rg, rm, games, rnm) == _bytes(dg, dm, games, dnm)
    assert info["evaluated"] == sum(calls) <= info["oracle_calls"]            # the table answers repeated states
    total_sims = sum(dg[i].num_moves for i in range(games)) * nsims
    assert info["oracle_calls"] <= total_sims


def test_a_table_that_keeps_forgetting_changes_nothing():
    """the smallest table the driver accepts (8 entries per worker): it is wiped again and again, states are asked for again"""
    kw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=SCHED[0], temp_ys=SCHED[1], reset_every=1, seed=9)
    dg, dm, dnm = R.simulate(R.C4, R.ORACLE_HASH, 8, 8, 64, **kw)
    ev = R.Evals(6)                                                            # 64 entries for 8 workers
    rg, rm, rnm, info = R.replay(R.C4, lambda k: R.hash_oracle_keys(R.C4, k), 8, 8, 64, evals=ev, **kw)
    c = ev.counters()
    ev.close()
    assert c["wipes"] > 10 and c["asked"] == c["answered"] == info["evaluated"]
    assert rnm == dnm and _bytes(rg, rm, 8, rnm) == _bytes(dg, dm, 8, dnm)


def test_a_shared_table_serves_several_replays():
    """chunks of one phase replayed one after the other over ONE table (how the GPU test bounds host memory): later chunks
    ask for fewer states, records equal the unchunked direct run's (reset_every = 1: a game depends on its id alone)"""
    kw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=SCHED[0], temp_ys=SCHED[1], reset_every=1, seed=2)
    dg, dm, dnm = R.simulate(R.C4, R.ORACLE_HASH, 16, 16, 32, **kw)
    direct = {dg[i].game_id: [bytes(dm[dg[i].first_move + k]) for k in range(dg[i].num_moves)] for i in range(16)}
    ev = R.Evals(16)
    asked = []
    for first in (0, 8):
        rg, rm, rnm, info = R.replay(R.C4, lambda k: R.hash_oracle_keys(R.C4, k), 8, 8, 32, evals=ev, first_game_id=first, **kw)
        asked.append(info["evaluated"])
        for i in range(8):
            assert [bytes(rm[rg[i].first_move + k]) for k in range(rg[i].num_moves)] == direct[rg[i].game_id]
            assert (rg[i].nodes, rg[i].total_simulations, rg[i].total_nodes_traversed) == \
                   (dg[first + i].nodes, dg[first + i].total_simulations, dg[first + i].total_nodes_traversed)
    ev.close()
    assert asked[1] < asked[0]


def test_replay_fed_with_the_oracle_network_equals_the_direct_run():
    rng = np.random.default_rng(3)
    hp = (1, 8, 4, 4)                                                          # a small ResNet: the oracle's network takes any width
    blob = rng.uniform(-0.3, 0.3, R.net_num_params(R.TTT, *hp)).astype(np.float32)
    n = R.net_num_params(R.TTT, *hp)
    assert blob.size == n
    kw = dict(cpuct=1.5, noise_eps=0.25, noise_alpha=0.5, temp_xs=(0,), temp_ys=(1.0,), reset_every=1, seed=4)
    dg, dm, dnm = R.simulate(R.TTT, R.ORACLE_NET, 6, 6, 24, net=hp + (blob,), **kw)
    rg, rm, rnm, info = R.replay(R.TTT, lambda k: R.net_evaluate_keys(R.TTT, hp, blob, k), 6, 6, 24, **kw)
    assert rnm == dnm and _bytes(rg, rm, 6, rnm) == _bytes(dg, dm, 6, dnm)


def test_the_evaluator_may_be_a_c_function():
    """the form the GPU test uses (the device library's az_net_evaluate_keys + its engine): here the oracle's own hash oracle
    behind the same signature, several host threads"""
    kw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=SCHED[0], temp_ys=SCHED[1], reset_every=1, seed=12)
    dg, dm, dnm = R.simulate(R.MANCALA, R.ORACLE_HASH, 40, 40, 30, **kw)
    fn = C.cast(R.lib().azr_eval_hash, C.c_void_p).value
    for threads in (1, 3):
        rg, rm, rnm, info = R.replay(R.MANCALA, (fn, R.MANCALA), 40, 40, 30, threads=threads, **kw)
        assert info["threads"] == threads and rnm == dnm and _bytes(rg, rm, 40, rnm) == _bytes(dg, dm, 40, dnm)


def test_an_evaluator_that_fails_stops_the_replay():
    def bad(keys):
        raise ValueError("no device")
    with pytest.raises(ValueError):
        R.replay(R.TTT, bad, 4, 4, 8, seed=1)
