"""Capacity policy of the search trees.  The reference's tree is a Dict that grows as it is used
(src/mcts.jl:124-151); here a slot's nodes are an array with a bound.
  * A slot that reaches its bound -- tree nodes (max_nodes_per_slot / device memory) or move records (max_moves_per_game) --
    is RETIRED: its game is dropped and reported (az_selfplay_stats.aborted_games, az_selfplay_aborted), the slot plays ONE
    replacement game (id | 0x40000000, its own RNG streams) with an empty tree so that the phase still returns num_games games
    (round 4; a replacement that overflows too is given up), the phase finishes and every completed game is returned
    (round 2: the whole az_selfplay_run failed).
  * Big pools (Mancala at BASELINE configs[3]: 107 GB of worst-case nodes) live in a virtual range whose 2 MB chunks are mapped
    on demand at the move steps: results are identical to the plain pool's, the memory held is what the games needed."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def _records(games, moves, ng):
    return {games[i].game_id: [(tuple(moves[games[i].first_move + k].key), list(moves[games[i].first_move + k].N), moves[games[i].first_move + k].action)
                              for k in range(games[i].num_moves)] for i in range(ng)}


def test_a_full_node_pool_retires_the_slot_and_the_phase_finishes():
    import azhip
    kw = dict(game=R.C4, oracle=azhip.ORACLE_UNIFORM, num_workers=4, batch_size=4, num_iters_per_turn=32, cpuct=2.0, dirichlet_noise_eps=0.25,
              reset_every=1, seed=11)
    with azhip.Engine(**kw) as full:
        g0, m0, n0, _, st0 = full.selfplay_run(12)
    want = _records(g0, m0, n0)
    nodes = sorted(g0[i].nodes for i in range(n0))
    assert st0.aborted_games == 0 and n0 == 12
    cap = int(nodes[-3])                                             # room for all but the games with the largest trees
    with azhip.Engine(max_nodes_per_slot=cap, **kw) as e:
        g, m, ng, nm, st = e.selfplay_run(12)                        # status OK: the phase is not lost
        aborted = e.selfplay_aborted()
    BIT = 0x40000000
    orig_aborted = [a for a in aborted if not a & BIT]
    given_up = [a for a in aborted if a & BIT]                       # replacements that overflowed as well
    assert st.aborted_games == len(aborted) >= 1 and len(orig_aborted) >= 1
    assert ng == 12 - len(given_up)                                  # the phase still owes (and returns) 12 games
    got = _records(g, m, ng)
    played = {gid & ~BIT for gid in got}
    assert played | {a & ~BIT for a in given_up} == set(range(12))
    assert all((gid & BIT) == 0 or (gid & ~BIT) in orig_aborted for gid in got) and not (set(got) & set(aborted))
    # a game's trace depends on its id alone (fresh tree per game, RNG keyed by game id): every ORIGINAL game that completed is
    # the unbounded run's game, whichever slot played it; a game is aborted exactly when its tree needs more than `cap` nodes
    for gid, rec in got.items():
        if not gid & BIT:
            assert rec == want[gid], gid
    need = {g0[i].game_id: g0[i].nodes for i in range(n0)}
    assert all(need[gid] > cap - 32 for gid in orig_aborted) and all(need[gid] <= cap for gid in got if not gid & BIT)
    assert nm == sum(len(r) for r in got.values()) and st.moves >= nm   # st.moves also counts the moves the aborted games made


def test_a_game_longer_than_the_move_record_is_retired_too():
    import azhip
    with azhip.Engine(game=R.MANCALA, oracle=azhip.ORACLE_UNIFORM, num_workers=3, batch_size=3, num_iters_per_turn=16, reset_every=1, seed=2,
                      max_moves_per_game=24) as e:
        g, m, ng, nm, st = e.selfplay_run(9)
        aborted = e.selfplay_aborted()
    given_up = [a for a in aborted if a & 0x40000000]
    assert st.aborted_games == len(aborted) >= 1 and ng == 9 - len(given_up)
    assert all(g[i].num_moves <= 24 for i in range(ng))


def test_mapped_on_demand_pool_gives_the_plain_pool_s_games_and_holds_less_memory(monkeypatch):
    import azhip
    kw = dict(game=R.MANCALA, oracle=azhip.ORACLE_HASH, num_workers=6, batch_size=3, num_iters_per_turn=600, cpuct=2.0, dirichlet_noise_eps=0.25,
              reset_every=1, seed=5, max_moves_per_game=256, temperature=((0, 10), (1.0, 0.5)))
    out = {}
    for vmm in ("0", "1", "1 dense side records"):
        monkeypatch.setenv("AZHIP_VMM", vmm[0])
        if len(vmm) > 1:
            monkeypatch.setenv("AZHIP_VMM_KEYS", "0")                # round 4's form: nodes mapped on demand, [G][cap] side records beside them
        with azhip.Engine(**kw) as e:
            g, m, ng, nm, st = e.selfplay_run(8)
            out[vmm] = (_records(g, m, ng), st.aborted_games, e.device_bytes(), max(g[i].nodes for i in range(ng)))
    monkeypatch.delenv("AZHIP_VMM_KEYS")
    assert out["0"][0] == out["1"][0] == out["1 dense side records"][0] and out["0"][1] == out["1"][1] == 0
    # (r5) the side records follow the node chunks: 32 B per node of the chunks that exist instead of 32 B x the worst case
    # (dense: 6 x 600 x 128 x 32 B = 14.7 MB; mapped: 2 MB granules of four 16 384-record pieces, as far as the six trees grew)
    assert out["1 dense side records"][2] > out["1"][2], (out["1 dense side records"][2], out["1"][2])
    assert out["1"][3] > 16384                                       # at least one tree grew past its first 2 MB chunk
    pool = 6 * 600 * 128 * 128                                       # the plain pool: slots x sims x 128 plies x 128 B
    assert out["0"][2] - out["1"][2] > pool // 3, (out["0"][2], out["1"][2])

    # the oracle plays the same games: the mapped pool is not only self-consistent
    games, moves, _ = R.simulate(R.MANCALA, R.ORACLE_HASH, 8, 6, 600, cpuct=2.0, noise_eps=0.25, reset_every=1, seed=5, temp_xs=(0, 10), temp_ys=(1.0, 0.5))
    ref = {games[i].game_id: [(tuple(moves[games[i].first_move + k].key), list(moves[games[i].first_move + k].N), moves[games[i].first_move + k].action)
                              for k in range(games[i].num_moves)] for i in range(8)}
    assert out["1"][0] == ref


def test_explore_on_a_mapped_pool_grows_past_its_first_chunk(monkeypatch):
    """MCTS.explore! through the hook (az_mcts_explore) on a mapped-on-demand pool: 30 000 simulations from one root need more
    nodes than the 16 384 of the first 2 MB chunk -- the chunks are mapped before the waves run, the tree equals the plain pool's"""
    import azhip
    out = {}
    for vmm in ("0", "1"):
        monkeypatch.setenv("AZHIP_VMM", vmm)
        with azhip.Engine(game=R.C4, oracle=azhip.ORACLE_HASH, num_workers=2, batch_size=2, num_iters_per_turn=8, cpuct=2.0,
                          dirichlet_noise_eps=0.0, max_nodes_per_slot=40000) as e:
            key = e.init_key()
            e.mcts_explore([key, key], 30000)
            N, W, P, V, mask = e.mcts_node_stats(1, key)
            out[vmm] = (list(N), list(W), e.mcts_counters(1))
    assert out["0"] == out["1"] and out["1"][2][2] > 16384 and sum(out["1"][0]) == 29999
