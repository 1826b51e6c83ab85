"""Parameter coverage of the self-play path on the GPU vs the oracle: every MctsParams / SimParams field that
changes arithmetic (src/params.jl:49-57,92-101) and the ragged / empty edge cases."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def _compare(game, oracle, ngames, workers, nsims, eng_kw, ref_kw):
    import azhip
    with azhip.Engine(game=game, oracle=oracle, num_workers=workers, batch_size=workers, num_iters_per_turn=nsims, **eng_kw) as e:
        dg, dm, ng, ndm, stats = e.selfplay_run(ngames)
    assert ng == ngames
    # which worker played which game is a race in the reference (util.jl:181-188); the oracle replays the outcome the device reports
    games, moves, nm = R.simulate(game, oracle, ngames, workers, nsims, assignment=R.assignment_of(dg, ngames), **ref_kw)
    assert ndm == nm
    for i in range(ngames):
        a, b = games[i], dg[i]
        assert (a.game_id, a.num_moves, a.nodes, a.total_simulations, a.total_nodes_traversed) == \
               (b.game_id, b.num_moves, b.nodes, b.total_simulations, b.total_nodes_traversed), i
        for k in range(a.num_moves):
            x, y = moves[a.first_move + k], dm[b.first_move + k]
            assert tuple(x.key) == tuple(y.key) and list(x.N) == list(y.N) and x.action == y.action and x.reward == y.reward, (i, k)
    return stats


@pytest.mark.parametrize("game", [0, 1, 2])
@pytest.mark.parametrize("prior_t,gamma,temp", [(0.5, 0.9, ((0,), (0.0,))), (0.0, 1.0, ((0, 3), (1.0, 0.5))), (2.0, 0.95, ((0,), (1.0,)))])
def test_prior_temperature_gamma_and_move_temperature(game, prior_t, gamma, temp):
    """prior_temperature (util.jl:98-110 on the oracle's priors, mcts.jl:157-161), gamma (mcts.jl:220), move
    temperature 0 = first argmax (util.jl:101-104) and fractional temperatures (pow + renormalise)."""
    _compare(game, R.ORACLE_HASH, 6, 3, 30,
             dict(gamma=gamma, cpuct=1.5, dirichlet_noise_eps=0.3, dirichlet_noise_alpha=0.5, prior_temperature=prior_t,
                  temperature=temp, reset_every=1, seed=21, max_moves_per_game=200 if game == 2 else 0),
             dict(gamma=gamma, cpuct=1.5, noise_eps=0.3, noise_alpha=0.5, prior_temperature=prior_t,
                  temp_xs=temp[0], temp_ys=temp[1], reset_every=1, seed=21))


def test_reset_every_never_and_ragged_game_counts():
    """reset_every = nothing (trees persist across a worker's games, simulations.jl:235-237), more workers than
    games (util.jl:181-188: idle workers), one worker for many games."""
    kw = dict(cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, seed=5)
    rkw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, seed=5)
    _compare(1, R.ORACLE_HASH, 9, 2, 40, dict(reset_every=0, max_nodes_per_slot=4000, **kw), dict(reset_every=0, **rkw))
    _compare(1, R.ORACLE_UNIFORM, 3, 8, 40, dict(reset_every=1, **kw), dict(reset_every=1, **rkw))
    _compare(0, R.ORACLE_HASH, 5, 1, 30, dict(reset_every=3, **kw), dict(reset_every=3, **rkw))


@pytest.mark.parametrize("game,flip", [(0, 0.5), (1, 0.5), (0, 1.0), (1, 1.0)])
def test_flip_probability_in_self_play(game, flip):
    """play_game's per-turn random symmetry (play.jl:305-307, game.jl:329-336) inside the device's self-play loop: states
    recorded before the flip, counts spread over the un-flipped state's mask by rank (learning.jl:31-33), the symmetry's
    index + 1 in N[AZ_MAX_ACTIONS]; tic-tac-toe draws among 7 images, trees persist over two games (reset_every 2), so
    a flipped root is sometimes found in the table and sometimes not."""
    kw = dict(cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, seed=9, temperature=((0, 4), (1.0, 0.5)))
    rkw = dict(cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, seed=9, temp_xs=(0, 4), temp_ys=(1.0, 0.5))
    import azhip
    games, moves, nm = R.simulate(game, R.ORACLE_HASH, 10, 4, 48, reset_every=2, flip_probability=flip, **rkw)
    flips = sum(1 for k in range(nm) if moves[k].N[R.AMAX])
    assert flips == nm if flip == 1.0 else 0 < flips < nm
    _compare(game, R.ORACLE_HASH, 10, 4, 48, dict(reset_every=2, flip_probability=flip, **kw),
             dict(reset_every=2, flip_probability=flip, **rkw))
    with pytest.raises(azhip.AzError, match="symmetries"):          # mancala declares none: the assert of game.jl:332
        with azhip.Engine(game=2, oracle=R.ORACLE_HASH, num_workers=2, batch_size=2, num_iters_per_turn=8, flip_probability=0.5) as e:
            e.selfplay_run(2)


def test_noise_alpha_below_one_uses_the_boosted_gamma_sampler():
    """Dirichlet(n, 0.03)-style sparse noise: Gamma(alpha < 1) = Gamma(alpha + 1) * U^(1/alpha) on both sides."""
    _compare(0, R.ORACLE_UNIFORM, 4, 4, 40, dict(cpuct=2.0, dirichlet_noise_eps=0.5, dirichlet_noise_alpha=0.1, seed=3),
             dict(cpuct=2.0, noise_eps=0.5, noise_alpha=0.1, seed=3))


def test_empty_and_single_inputs():
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=2, batch_size=2, num_iters_per_turn=4,
                      num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(random_params(0, hp))
        P, V, Pinv = e.net_forward(np.zeros((0, 3, 6, 7), np.float32), np.zeros((0, 7), np.float32))
        assert P.shape == (0, 7) and V.shape == (0,)
        X, A = e.encode(np.zeros((0, 2), np.uint64))
        assert X.shape == (0, 3, 6, 7)
        # a batch that is not a multiple of the 3-board tile or the 32-board head tile
        keys = np.array([e.init_key()] * 5, dtype=np.uint64)
        Pk, Vk = e.net_evaluate_keys(keys)
        assert np.all(Pk == Pk[0]) and np.all(Vk == Vk[0]) and abs(Pk[0].sum() - 1) < 1e-5
