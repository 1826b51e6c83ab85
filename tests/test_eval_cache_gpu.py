"""The engine's evaluation cache (csrc/tree.h ECEnt, round 5) changes no record.

The reference evaluates every oracle query on its own (src/simulations.jl:23-38, src/networks/network.jl:308-315).  A test-mode
evaluation is a pure function of the state, and every tower form of this library gives the same bits for it (tests/test_net.py),
so an answer computed once -- for another slot, in an earlier wave -- is THE answer: the engine keeps answers by state in a
direct-mapped table and sends a state to the network only when the table does not hold it.  What must hold, and is tested here:
  * every record of a phase is the same with the cache on, off, tiny (every claim evicts somebody) or large -- against the
    oracle's direct run (exact hash oracle, forced on with AZHIP_EVAL_CACHE=1: the cache is otherwise only used where an evaluation
    is expensive) and cache-on against cache-off with the ResNet in the loop;
  * leaf_evals (the reference's count of oracle calls) does not change, evals_reused says how many of them the network was spared;
  * az_net_set_params forgets the old network's answers.
Every other GPU test of the suite that runs the ResNet oracle runs with the cache ON (the default) and is compared with the
oracle's own network: tests/test_baseline_configs_gpu.py, tests/test_replay_all_games_gpu.py (every game of the BASELINE phases)."""
import ctypes as C

import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu

SCHED = ((0, 6, 12), (1.0, 1.0, 0.3))


def _recs(games, moves, ng, cumulative=True):
    """cumulative=False: without the per-worker counters (self_play_measurements, training.jl:269-273) -- they depend on which games the
    worker played before, i.e. on the outcome of the id race (util.jl:181-188), which two free-running runs need not share"""
    out = {}
    for i in range(ng):
        g = games[i]
        out[g.game_id] = (g.num_moves, g.nodes, (g.total_simulations, g.total_nodes_traversed) if cumulative else None, tuple(g.final_key),
                          [bytes(moves[g.first_move + k]) for k in range(g.num_moves)])
    return out


@pytest.mark.parametrize("game_name,log2", [("c4", 4), ("c4", 10), ("c4", 20), ("ttt", 6), ("mancala", 8)])
def test_hash_oracle_phases_with_the_cache_forced_on_equal_the_oracle(monkeypatch, game_name, log2):
    """log2 = 4: sixteen entries for 64 slots -- every wave fights for them (claims, evictions, stale claims, fills that find
    their entry taken); 20: nothing is ever evicted.  Two slot groups, trees kept over two games, flips on where the game has symmetries."""
    import azhip
    gh, gr, nsims, flip = {"c4": (azhip.GAME_CONNECT_FOUR, R.C4, 120, 0.5), "ttt": (azhip.GAME_TICTACTOE, R.TTT, 60, 1.0),
                           "mancala": (azhip.GAME_MANCALA, R.MANCALA, 100, 0.0)}[game_name]
    monkeypatch.setenv("AZHIP_EVAL_CACHE", "1")
    monkeypatch.setenv("AZHIP_EVAL_CACHE_LOG2", str(log2))
    with azhip.Engine(game=gh, oracle=azhip.ORACLE_HASH, num_workers=64, batch_size=32, num_iters_per_turn=nsims, cpuct=2.0,
                      dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=2, flip_probability=flip, seed=3) as e:
        g, m, ng, nm, st = e.selfplay_run(192)
        dev = _recs(g, m, ng)
        asg = R.assignment_of(g, 192)                                # the outcome of the id race this phase took (util.jl:181-188)
        # a second phase on the same engine: the table is warm (and the launch numbers go on)
        g2, m2, ng2, nm2, st2 = e.selfplay_run(64, first_game_id=1000)
        dev2 = _recs(g2, m2, ng2)
        asg2 = R.assignment_of(g2, 64, 1000)
    kw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=SCHED[0], temp_ys=SCHED[1], reset_every=2, seed=3, flip_probability=flip)
    rg, rm, rnm = R.simulate(gr, R.ORACLE_HASH, 192, 64, nsims, assignment=asg, **kw)
    assert dev == _recs(rg, rm, 192)
    rg2, rm2, _ = R.simulate(gr, R.ORACLE_HASH, 64, 64, nsims, first_game_id=1000, assignment=asg2, **kw)
    assert dev2 == _recs(rg2, rm2, 64)
    assert 0 < st.evals_reused < st.leaf_evals and st2.evals_reused > 0
    if log2 >= 20:
        assert st.evals_reused > 0.2 * st.leaf_evals                 # 64 games from the same opening share a lot


def test_resnet_phase_is_the_same_with_and_without_the_cache(monkeypatch):
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=2, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=11)
    out = {}
    for mode in ("0", "1", "tiny"):
        monkeypatch.setenv("AZHIP_EVAL_CACHE", "0" if mode == "0" else "1")
        monkeypatch.setenv("AZHIP_EVAL_CACHE_LOG2", "8" if mode == "tiny" else "22")
        with azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_workers=512, batch_size=256, num_iters_per_turn=100,
                          cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=1, seed=5,
                          num_blocks=2, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32) as e:
            e.net_set_params(blob)
            g, m, ng, nm, st = e.selfplay_run(1024)                  # two games per slot: the second ones start on a warm table
            out[mode] = (_recs(g, m, ng, cumulative=False), st.leaf_evals, st.evals_reused, st.simulations)
    assert out["0"][0] == out["1"][0] == out["tiny"][0]
    assert out["0"][1] == out["1"][1] == out["tiny"][1] and out["0"][3] == out["1"][3]   # the reference's counts do not move
    assert out["0"][2] == 0 and out["1"][2] > 0.25 * out["1"][1] and 0 < out["tiny"][2] < out["1"][2]


def test_new_parameters_empty_the_cache():
    """the answers of the old network must not survive az_net_set_params: a warm engine that gets new weights plays exactly the
    games a fresh engine with those weights plays"""
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    a, b = random_params(azhip.GAME_TICTACTOE, hp, seed=1), random_params(azhip.GAME_TICTACTOE, hp, seed=2)
    kw = dict(game=azhip.GAME_TICTACTOE, oracle=azhip.ORACLE_RESNET, num_workers=32, batch_size=32, num_iters_per_turn=40, cpuct=1.5,
              dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, reset_every=1, seed=8, num_blocks=1, num_filters=64,
              num_policy_head_filters=32, num_value_head_filters=32)
    with azhip.Engine(**kw) as e:
        e.net_set_params(a)
        ga, ma, nga, _, sta = e.selfplay_run(64)
        e.net_set_params(b)
        gb, mb, ngb, _, stb = e.selfplay_run(64)
        warm_b = _recs(gb, mb, ngb, cumulative=False)
        first_a = _recs(ga, ma, nga, cumulative=False)
    with azhip.Engine(**kw) as e:
        e.net_set_params(b)
        g, m, ng, _, st = e.selfplay_run(64)
        fresh_b = _recs(g, m, ng, cumulative=False)
    assert warm_b == fresh_b and warm_b != first_a and sta.evals_reused > 0 and stb.evals_reused > 0 and st.evals_reused > 0
    assert stb.leaf_evals == st.leaf_evals
