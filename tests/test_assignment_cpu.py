"""The oracle replays a given outcome of the reference's game-id race (src/util.jl:181-188): azr_sim_set_assignment."""
import numpy as np
import pytest

import azref as R

KW = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=(0, 4), temp_ys=(1.0, 0.5), seed=6)


def _recs(games, moves, n, cumulative=True):
    out = {}
    for i in range(n):
        g = games[i]
        head = (g.num_moves, g.nodes, tuple(g.final_key)) + ((g.slot, g.total_simulations, g.total_nodes_traversed) if cumulative else ())
        out[g.game_id] = (head, [bytes(moves[g.first_move + k]) for k in range(g.num_moves)])
    return out


@pytest.mark.parametrize("reset_every", [1, 2, 0])
def test_the_lock_step_outcome_replayed_as_an_assignment_is_the_lock_step_run(reset_every):
    g, m, nm = R.simulate(R.TTT, R.ORACLE_HASH, 30, 4, 24, reset_every=reset_every, **KW)
    asg = R.assignment_of(g, 30)
    g2, m2, nm2 = R.simulate(R.TTT, R.ORACLE_HASH, 30, 4, 24, reset_every=reset_every, assignment=asg, **KW)
    assert nm == nm2 and _recs(g, m, 30) == _recs(g2, m2, 30)


def test_another_outcome_of_the_race():
    """round-robin instead of finishing order: with reset_every = 1 a game depends on its id alone (the per-worker counters aside);
    with trees kept over games the records change, and the replay mode (external evaluator) follows the same assignment"""
    asg = np.arange(30, dtype=np.int32) % 4
    g, m, nm = R.simulate(R.TTT, R.ORACLE_HASH, 30, 4, 24, reset_every=1, **KW)
    g2, m2, nm2 = R.simulate(R.TTT, R.ORACLE_HASH, 30, 4, 24, reset_every=1, assignment=asg, **KW)
    assert _recs(g, m, 30, cumulative=False) == _recs(g2, m2, 30, cumulative=False)
    assert [g2[i].slot for i in range(30)] == asg.tolist()
    g3, m3, nm3 = R.simulate(R.TTT, R.ORACLE_HASH, 30, 4, 24, reset_every=0, assignment=asg, **KW)
    g4, m4, nm4, info = R.replay(R.TTT, lambda k: R.hash_oracle_keys(R.TTT, k), 30, 4, 24, reset_every=0, assignment=asg, **KW)
    assert nm3 == nm4 and _recs(g3, m3, 30) == _recs(g4, m4, 30)
    g5, m5, _ = R.simulate(R.TTT, R.ORACLE_HASH, 30, 4, 24, reset_every=0, **KW)
    assert _recs(g3, m3, 30) != _recs(g5, m5, 30)                    # which games share a tree matters


def test_assignments_the_reference_could_not_produce_are_refused():
    with pytest.raises(ValueError):
        R.simulate(R.TTT, R.ORACLE_HASH, 6, 4, 8, assignment=[1, 0, 2, 3, 0, 1], **KW)      # the first workers start games 0, 1, 2 ...
    with pytest.raises(ValueError):
        R.simulate(R.TTT, R.ORACLE_HASH, 6, 4, 8, assignment=[0, 1, 2, 3, 4, 0], **KW)      # no such worker
