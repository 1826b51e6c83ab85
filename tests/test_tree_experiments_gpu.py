"""Round-5 forms of k_tree that are off by default (profiles/r5/README.md): whatever they do to the
kernel's time, every record of a phase must stay the oracle's.
  * AZHIP_TREE_ATOMIC = 1 | 2: the backup W += q, N += 1 (update_state_info!, src/mcts.jl:190-194) as no-return atomics performed by L2
    instead of a read-modify-write in the lane (one update per (node, action), slot and wave, so the same IEEE add);
  * AZHIP_TREE_SORT = 1: at every move step the slots of a slot group are handed to k_tree's lane groups in the order of the depth of
    their last explore! (results are by slot: a game's trace does not depend on which lane group advanced it)."""
import ctypes as C

import pytest

import azref as R

pytestmark = pytest.mark.gpu

SCHED = ((0, 6, 12), (1.0, 1.0, 0.3))


def _bytes(games, moves, ng):
    return {games[i].game_id: (games[i].num_moves, games[i].nodes, games[i].total_simulations, games[i].total_nodes_traversed, tuple(games[i].final_key),
                               [bytes(moves[games[i].first_move + k]) for k in range(games[i].num_moves)]) for i in range(ng)}


def _phase(game_hip, game_ref, workers, batch, games, nsims, reset_every, flip):
    import azhip
    with azhip.Engine(game=game_hip, oracle=azhip.ORACLE_HASH, num_workers=workers, batch_size=batch, num_iters_per_turn=nsims, cpuct=2.0,
                      dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=reset_every, flip_probability=flip, seed=11) as e:
        g, m, ng, nm, st = e.selfplay_run(games)
        assert ng == games and st.aborted_games == 0
        dev = _bytes(g, m, ng)
        asg = R.assignment_of(g, games)
    rg, rm, _ = R.simulate(game_ref, R.ORACLE_HASH, games, workers, nsims, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=SCHED[0], temp_ys=SCHED[1],
                           reset_every=reset_every, seed=11, flip_probability=flip, assignment=asg)
    assert C.sizeof(rm[0]) == 64
    assert dev == _bytes(rg, rm, games)


@pytest.mark.parametrize("knob,value", [("AZHIP_TREE_ATOMIC", "1"), ("AZHIP_TREE_ATOMIC", "2"), ("AZHIP_TREE_SORT", "1")])
def test_every_record_is_still_the_oracle_s(monkeypatch, knob, value):
    import azhip
    monkeypatch.setenv(knob, value)
    _phase(azhip.GAME_CONNECT_FOUR, R.C4, 96, 48, 300, 120, 2, 0.5)      # two slot groups, trees kept over two games, flips
    _phase(azhip.GAME_MANCALA, R.MANCALA, 40, 40, 64, 100, 1, 0.0)        # pswitch: the sign of q along the path
    _phase(azhip.GAME_TICTACTOE, R.TTT, 24, 24, 96, 64, 1, 0.5)         # 16 lanes per slot
