"""Events of the device trees that are rare at the shipped sizes, forced.

A slot's Dict{State,StateInfo} (src/mcts.jl:126) is an open-addressed table of 64-bit entries  epoch(16) | tag(16) | node + 1  with
the exact key compare in a side array, child links of 18 bits memoise its answers, and MCTS.reset! is a new epoch (csrc/tree.h).
Three things the BASELINE-size runs meet once in millions of probes -- or never -- are made the common case here and compared,
record by record, with the oracle's Dict walk:
  * tag collisions: AZHIP_HT_TAG_BITS narrows the tag to 1 bit (or none), so every probe chain holds unequal states with equal tags
    and is decided by the key compare alone; the pool is sized so that the table runs at its highest load;
  * link overflow: max_nodes_per_slot above 2^18 - 2 -- links are never written, every edge is a table probe;
  * epoch wrap: AZHIP_HT_EPOCH0 starts the 16-bit epoch a few resets before 0xffff, where the table is really cleared."""
import ctypes as C

import pytest

import azref as R

pytestmark = pytest.mark.gpu

SCHED = ((0, 6, 12), (1.0, 1.0, 0.3))


def _bytes(games, moves, ng):
    out = {}
    for i in range(ng):
        g = games[i]
        out[g.game_id] = (g.num_moves, g.nodes, g.total_simulations, g.total_nodes_traversed, tuple(g.final_key),
                          [bytes(moves[g.first_move + k]) for k in range(g.num_moves)])
    return out


def _both(game_hip, game_ref, workers, games, nsims, reset_every, flip=0.0, **engine_kw):
    import azhip
    with azhip.Engine(game=game_hip, oracle=azhip.ORACLE_HASH, num_workers=workers, batch_size=workers, num_iters_per_turn=nsims,
                      cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=reset_every,
                      flip_probability=flip, seed=7, **engine_kw) as e:
        g, m, ng, nm, st = e.selfplay_run(games)
        assert ng == games and st.aborted_games == 0
        dev = _bytes(g, m, ng)
        asg = R.assignment_of(g, games)                              # the outcome of the id race the phase took (util.jl:181-188)
    rg, rm, rnm = R.simulate(game_ref, R.ORACLE_HASH, games, workers, nsims, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0,
                             temp_xs=SCHED[0], temp_ys=SCHED[1], reset_every=reset_every, seed=7, flip_probability=flip, assignment=asg)
    assert C.sizeof(rm[0]) == 64
    assert dev == _bytes(rg, rm, games)
    return dev


@pytest.mark.parametrize("bits", [1, 0])
def test_tag_collisions_everywhere_and_a_table_at_its_highest_load(monkeypatch, bits):
    import azhip
    monkeypatch.setenv("AZHIP_HT_TAG_BITS", str(bits))
    # trees kept over two games; find the pool the games need, then run again with exactly that much: the table (the smallest power
    # of two >= 1.5 x the pool) is then as full as it ever gets
    dev = _both(azhip.GAME_CONNECT_FOUR, R.C4, 8, 32, 150, 2, flip=0.5)
    need = max(v[1] for v in dev.values())
    cap = 1 << (need - 1).bit_length()                               # pool = a power of two: the table is 2 x pool, load up to need / (2 cap)
    if cap * 3 // 4 >= need:
        cap = cap * 3 // 4                                           # ... or 1.5 x smaller when that still fits: load up to 2/3
    _both(azhip.GAME_CONNECT_FOUR, R.C4, 8, 32, 150, 2, flip=0.5, max_nodes_per_slot=int(max(cap, 1024)))
    _both(azhip.GAME_MANCALA, R.MANCALA, 8, 16, 120, 1)


def test_pools_above_the_link_range_probe_every_edge():
    import azhip
    _both(azhip.GAME_CONNECT_FOUR, R.C4, 4, 8, 300, 1, max_nodes_per_slot=(1 << 18) + 1000)
    _both(azhip.GAME_TICTACTOE, R.TTT, 4, 12, 100, 3, flip=1.0, max_nodes_per_slot=(1 << 18) + 1000)


def test_the_table_epoch_wraps(monkeypatch):
    import azhip
    monkeypatch.setenv("AZHIP_HT_EPOCH0", str(0xffff - 4))           # the 4th reset of every slot reaches 0xffff: tables really cleared
    _both(azhip.GAME_CONNECT_FOUR, R.C4, 4, 40, 60, 1)
    _both(azhip.GAME_TICTACTOE, R.TTT, 4, 60, 40, 2)
