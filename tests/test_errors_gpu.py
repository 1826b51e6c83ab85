"""Error behaviour of the engine on the GPU: capacities and call order come back as statuses (AzError), the
engine stays usable, nothing crashes (SURVEY.md §8b 'Errors')."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def test_capacity_and_state_errors():
    import azhip
    from azhip import _lib as L
    with azhip.Engine(game=0, oracle=azhip.ORACLE_UNIFORM, num_workers=4, batch_size=4, num_iters_per_turn=50,
                      max_nodes_per_slot=20) as e:
        key = e.init_key()
        with pytest.raises(azhip.AzError) as ei:
            e.mcts_explore([key], 100, eta=np.zeros((1, 9)))        # 100 new nodes > 20
        assert ei.value.status == L.AZ_ERR_CAPACITY and "node pool" in str(ei.value)
        e.mcts_reset()
        e.mcts_explore([key], 10, eta=np.zeros((1, 9)))              # engine still usable
        assert e.mcts_counters(0)[2] == 10
        with pytest.raises(azhip.AzError) as ei:
            e.selfplay_step(1)                                       # begin not called
        assert ei.value.status == L.AZ_ERR_STATE
        with pytest.raises(azhip.AzError) as ei:
            e.mcts_explore([key] * 5, 4)                             # more roots than slots
        assert ei.value.status == L.AZ_ERR_BAD_ARG
        with pytest.raises(azhip.AzError) as ei:
            e.mcts_node_stats(0, (123, 456))                         # KeyError analogue
        assert ei.value.status == L.AZ_ERR_BAD_ARG
    # a finished position cannot be a search root
    g = R.Game(R.TTT)
    for a in (0, 3, 1, 4, 2):
        g.play(a)
    assert g.terminated()
    with azhip.Engine(game=1, oracle=azhip.ORACLE_UNIFORM, num_workers=2, batch_size=2, num_iters_per_turn=8) as e:
        with pytest.raises(azhip.AzError):
            e.mcts_explore([g.key()], 4)
        e.selfplay_begin(2, 0)
        with pytest.raises(azhip.AzError):
            e.mcts_explore([e.init_key()], 4)                        # self-play in progress
        with pytest.raises(azhip.AzError):
            e.selfplay_begin(2, 0)
        while e.selfplay_active():
            e.selfplay_step(8)
        with pytest.raises(azhip.AzError) as ei:
            e.selfplay_collect(1, 1)                                 # caller buffers too small
        assert ei.value.status == L.AZ_ERR_CAPACITY
        games, moves, ng, nm = e.selfplay_collect(2)
        assert ng == 2 and nm >= 10
        e.selfplay_end()


def test_game_too_long_is_reported():
    """round 3: a game that outgrows its move record no longer fails the phase -- it is retired and reported as aborted"""
    import azhip
    with azhip.Engine(game=2, oracle=azhip.ORACLE_UNIFORM, num_workers=4, batch_size=4, num_iters_per_turn=8,
                      max_moves_per_game=5) as e:
        g, m, ng, nm, st = e.selfplay_run(4)
        # every game and its one replacement (id | bit 30) outgrow the record: all given up, all eight reported, the call returns
        assert ng == 0 and st.aborted_games == 8 and sorted(e.selfplay_aborted()) == [0, 1, 2, 3] + [0x40000000 + i for i in range(4)]
