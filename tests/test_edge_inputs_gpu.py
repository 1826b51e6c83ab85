"""Empty, single and ragged inputs through the C ABI (the edge cases a reference test suite would hold for this path):
zero boards, one board, counts that are not multiples of any tile, fewer games than slots, one game, empty memories and data sets."""
import ctypes as C

import numpy as np
import pytest

import azref as R
from test_net import ResNetHP, batch_of, random_params, random_positions

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bf16", [0, 1])
def test_forward_on_zero_one_and_ragged_batches(bf16):
    import azhip
    hp = ResNetHP(num_blocks=2, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(R.C4, hp, seed=3)
    envs = random_positions(R.C4, 45, 4)
    X, A = batch_of(R.C4, envs)
    with azhip.Engine(game=R.C4, oracle=azhip.ORACLE_RESNET, num_workers=64, batch_size=64, num_iters_per_turn=8, num_blocks=2,
                      num_filters=64, num_policy_head_filters=32, num_value_head_filters=32, net_bf16=bf16) as e:
        e.net_set_params(blob)
        Pall, Vall, _ = e.net_forward(X, A)
        P0, V0, _ = e.net_forward(X[:0], A[:0])                      # no board at all
        assert P0.shape[0] == 0 and V0.shape[0] == 0
        for n in (1, 3, 5, 17, 33, 45):                              # 1 board; not a multiple of 4 / 8 / 16 / 32 boards per tile
            P, V, _ = e.net_forward(X[:n], A[:n])
            assert np.array_equal(P, Pall[:n]) and np.array_equal(V, Vall[:n]), n     # a board's result does not depend on its batch
        Pk, Vk = e.net_evaluate_keys(np.zeros((0, 2), dtype=np.uint64))
        assert Pk.shape[0] == 0 and Vk.shape[0] == 0
        Xe, Ae = e.encode(np.zeros((0, 2), dtype=np.uint64))
        assert Xe.shape[0] == 0 and Ae.shape[0] == 0
    if not bf16:
        Pr, Vr, _ = R.net_forward_normalized(R.C4, (2, 64, 32, 32), blob, X, A)
        assert np.array_equal(Pall, Pr) and np.array_equal(Vall, Vr)


@pytest.mark.parametrize("ngames,workers,batch", [(1, 8, 8), (3, 8, 4), (5, 4, 2), (9, 9, 3)])
def test_fewer_games_than_slots_and_ragged_groups(ngames, workers, batch):
    import azhip
    with azhip.Engine(game=R.TTT, oracle=azhip.ORACLE_HASH, num_workers=workers, batch_size=batch, num_iters_per_turn=20, cpuct=1.0,
                      dirichlet_noise_eps=0.25, reset_every=1, seed=4, temperature=((0,), (1.0,))) as e:
        g, m, ng, nm, st = e.selfplay_run(ngames)
    rg, rm, rnm = R.simulate(R.TTT, R.ORACLE_HASH, ngames, workers, 20, cpuct=1.0, noise_eps=0.25, reset_every=1, seed=4, temp_xs=(0,), temp_ys=(1.0,),
                              assignment=R.assignment_of(g, ngames))
    assert ng == ngames and nm == rnm and st.aborted_games == 0
    for i in range(ngames):
        assert (g[i].game_id, g[i].num_moves) == (rg[i].game_id, rg[i].num_moves)
        for k in range(g[i].num_moves):
            a, b = m[g[i].first_move + k], rm[rg[i].first_move + k]
            assert tuple(a.key) == tuple(b.key) and list(a.N) == list(b.N) and a.action == b.action


def test_empty_memory_and_empty_pushes():
    import azhip
    gspec = azhip.TicTacToeSpec()
    mem = azhip.MemoryBuffer(gspec, 100)
    assert len(mem) == 0 and mem.cur_batch_size() == 0
    with mem.dataset(use_symmetries=True, use_position_averaging=True) as d:     # get_experience of an empty buffer
        assert len(d) == 0 and d.sum_n == 0
    with azhip.Engine(game=R.TTT, oracle=azhip.ORACLE_UNIFORM, num_workers=2, batch_size=2, num_iters_per_turn=8, reset_every=1) as e:
        g, m, ng, nm, st = e.selfplay_run(2)
        mem.push_records(g, m, 0, 0, 1.0)                             # a phase without games
        assert len(mem) == 0
        mem.push_records(g, m, ng, nm, 1.0)
        assert len(mem) == nm
        mem.new_batch()
        with mem.dataset(last_batch=True) as d:                        # last_batch right after new_batch!: empty
            assert len(d) == 0
        # the exchange with a world of one rank and a phase of ONE game
        from azhip import comm
        g1, _, ng1, _, _ = e.selfplay_run(1, first_game_id=7, device_only=True)
        with comm.Comm(0, 0, 1, comm.unique_id()) as c:
            gs = c.gather_push(e, mem, 1.0)
        assert gs.games == 1 and gs.ranks == 1 and len(mem) == nm + gs.moves
    mem.close()
