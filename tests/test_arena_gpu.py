"""Arena parity (GPU): az_arena_run (pit_networks, src/training.jl:130-144; TwoPlayers, src/play.jl:248-282;
flip_probability / alternate_colors, src/play.jl:305-307, src/simulations.jl:221-223) against the oracle's
azr_arena -- every record, reward and the redundancy must be identical."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def _same_records(games, moves, ng, g_ref, m_ref, nm_ref, nA):
    assert ng == len(g_ref)
    for i in range(ng):
        a, b = games[i], g_ref[i]
        assert (a.game_id, a.slot, a.num_moves) == (b.game_id, b.slot, b.num_moves), i
        assert (a.final_key[0], a.final_key[1]) == (b.final_key[0], b.final_key[1]), i
        for k in range(a.num_moves):
            x, y = moves[a.first_move + k], m_ref[b.first_move + k]
            assert (x.key[0], x.key[1]) == (y.key[0], y.key[1]), (i, k)
            assert list(x.N) == list(y.N)[:len(list(x.N))], (i, k)
            assert x.action == y.action and x.reward == y.reward, (i, k)


def _engine_kw(pl, game, workers, batch, reset_every, flip, seed):
    return dict(game=game, oracle=pl["oracle"], num_workers=workers, batch_size=batch, num_iters_per_turn=pl["nsims"],
                cpuct=pl.get("cpuct", 1.0), dirichlet_noise_eps=pl.get("noise_eps", 0.0),
                dirichlet_noise_alpha=pl.get("noise_alpha", 1.0), temperature=(list(pl.get("temp_xs", (0,))), list(pl.get("temp_ys", (1.0,)))),
                reset_every=reset_every, flip_probability=flip, seed=pl.get("seed", seed))


@pytest.mark.parametrize("game,ngames,workers,batch,flip,alt", [
    (R.TTT, 20, 8, 8, 0.5, True), (R.C4, 12, 6, 3, 0.5, True), (R.C4, 9, 4, 4, 0.0, False),
    (R.MANCALA, 6, 3, 3, 0.0, True), (R.TTT, 7, 16, 16, 1.0, False)])
def test_arena_matches_oracle_hash_players(game, ngames, workers, batch, flip, alt):
    """two MctsPlayers with DIFFERENT MctsParams (Benchmark.Duel style) over the synthetic oracle"""
    import azhip
    c = dict(oracle=R.ORACLE_HASH, nsims=30, cpuct=2.0, noise_eps=0.05, noise_alpha=1.0, temp_xs=(0,), temp_ys=(0.2,))
    b = dict(oracle=R.ORACLE_HASH, nsims=12, cpuct=1.0, noise_eps=0.25, noise_alpha=0.7, temp_xs=(0, 4), temp_ys=(1.0, 0.5))
    count = [0]
    with azhip.Engine(**_engine_kw(c, game, workers, batch, 2, flip, 21)) as ec, \
            azhip.Engine(**_engine_kw(b, game, workers, batch, 2, flip, 21)) as eb:
        games, moves, ng, nm, rew, red = ec.arena_run(eb, ngames, alternate_colors=alt,
                                                     progress=lambda: count.__setitem__(0, count[0] + 1))
        # which worker plays which game is an outcome of the reference's id race (util.jl:181-188); the arena reports the one it took per
        # game (az_game_rec.slot) and the oracle replays it.  (The arena hands ids out in worker order: the oracle's default assignment.)
        g_ref, m_ref, nm_ref, rew_ref, red_ref = R.arena(game, ngames, workers, c, b, alternate_colors=alt, flip_probability=flip,
                                                         reset_every=2, seed=21, assignment=R.assignment_of(games, ngames))
        assert count[0] == ngames and nm == nm_ref
        _same_records(games, moves, ng, g_ref, m_ref, nm_ref, R.NUM_ACTIONS[game])
        assert np.array_equal(rew, rew_ref) and red == red_ref
        # the engines stay usable: a second run from a different id range is independent of the first
        games2, moves2, ng2, nm2, rew2, _ = ec.arena_run(eb, 4, first_game_id=100, alternate_colors=alt)
        g2, m2, nm2_ref, rew2_ref, _ = R.arena(game, 4, workers, c, b, alternate_colors=alt, flip_probability=flip,
                                               reset_every=2, seed=21, first_game_id=100, assignment=R.assignment_of(games2, 4, 100))
        _same_records(games2, moves2, ng2, g2, m2, nm2_ref, R.NUM_ACTIONS[game])
        assert np.array_equal(rew2, rew2_ref)


def test_pit_networks_two_resnets_matches_oracle():
    """pit_networks / compare_networks with two different ResNets (training.jl:130-172), connect-four arena
    parameters in miniature (flip 0.5, alternate colours, reset_every 2, const temperature 0.2, eps 0.05)."""
    import azhip
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=2, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    contender, baseline = azhip.ResNet(gspec, hp, seed=5), azhip.ResNet(gspec, hp, seed=6)
    mp = azhip.MctsParams(num_iters_per_turn=24, cpuct=2.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0,
                          temperature=azhip.ConstSchedule(0.2))
    ap = azhip.ArenaParams(mcts=mp, sim=azhip.SimParams(num_games=8, num_workers=4, batch_size=4, use_gpu=True, reset_every=2,
                                                       flip_probability=0.5, alternate_colors=True), update_threshold=0.05)

    class H:
        n = 0

        def checkpoint_game_played(self):
            H.n += 1
    ev = azhip.compare_networks(gspec, contender, baseline, ap, H(), seed=13)
    pl = dict(oracle=R.ORACLE_NET, nsims=24, cpuct=2.0, noise_eps=0.05, noise_alpha=1.0, temp_xs=(0,), temp_ys=(0.2,))
    _, _, _, rew_ref, red_ref = R.arena(R.C4, 8, 4, dict(pl, net=(2, 64, 32, 32, contender.params())),
                                        dict(pl, net=(2, 64, 32, 32, baseline.params())), alternate_colors=True,
                                        flip_probability=0.5, reset_every=2, seed=13, assignment=ev.workers)
    assert H.n == 8 and np.array_equal(ev.rewards, rew_ref) and ev.redundancy == red_ref
    assert ev.avgr == float(np.mean(rew_ref)) and ev.baseline_rewards is None and ev.time > 0
    # a network against itself, colours alternating: same trees on both sides is NOT the same as self-play, but
    # the result must be deterministic
    ev2 = azhip.compare_networks(gspec, contender, contender, ap, None, seed=13)
    ev3 = azhip.compare_networks(gspec, contender, contender, ap, None, seed=13)
    assert np.array_equal(ev2.rewards, ev3.rewards)


def test_arena_errors():
    import azhip
    from azhip import _lib as L
    kw = dict(oracle=L.ORACLE_HASH, num_workers=4, batch_size=4, num_iters_per_turn=8)
    with azhip.Engine(game=L.GAME_MANCALA, flip_probability=0.5, **kw) as a, azhip.Engine(game=L.GAME_MANCALA, **kw) as b:
        with pytest.raises(L.AzError, match="symmetries"):        # game.jl:332 assert
            a.arena_run(b, 2)
        with pytest.raises(L.AzError, match="two engines"):
            a.arena_run(a, 2)
        with pytest.raises(L.AzError, match="symmetries"):        # self-play: the same assert
            a.selfplay_run(2)
        b.arena_run(a, 2)                                         # baseline's flip setting is not consulted
    with azhip.Engine(game=L.GAME_TICTACTOE, **kw) as a, azhip.Engine(game=L.GAME_CONNECT_FOUR, **kw) as b:
        with pytest.raises(L.AzError, match="different games"):
            a.arena_run(b, 2)
    with azhip.Engine(game=L.GAME_TICTACTOE, **kw) as a, \
            azhip.Engine(game=L.GAME_TICTACTOE, **dict(kw, num_workers=2, batch_size=2)) as b:
        with pytest.raises(L.AzError, match="fewer workers"):
            a.arena_run(b, 8)
        a.arena_run(b, 2)                                         # 2 workers suffice for 2 games


@pytest.mark.parametrize("player", ["full", "network_only"])
def test_benchmark_duels_match_oracle(player):
    """The shipped connect-four benchmark in miniature (games/connect-four/params.jl:66-100): Benchmark.Duel(Full |
    NetworkOnly(0.5), MctsRollouts) through Benchmark.run -- az_arena_run with a rollout-oracle engine and, for
    NetworkOnly, an engine without search -- equals the oracle's arena."""
    import azhip
    from azhip import Benchmark
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=9)
    arena_mcts = azhip.MctsParams(num_iters_per_turn=20, cpuct=2.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0,
                                  temperature=azhip.ConstSchedule(0.2))
    base_mcts = azhip.MctsParams(num_iters_per_turn=50, cpuct=1.0, dirichlet_noise_ϵ=0.05, dirichlet_noise_α=1.0,
                                 temperature=azhip.ConstSchedule(0.2))
    sim = azhip.SimParams(num_games=10, num_workers=5, batch_size=5, use_gpu=True, reset_every=2, flip_probability=0.5,
                          alternate_colors=False)
    pl = Benchmark.Full(arena_mcts) if player == "full" else Benchmark.NetworkOnly(τ=0.5)
    duel = Benchmark.Duel(pl, Benchmark.MctsRollouts(base_mcts), sim)
    count = [0]
    ev = Benchmark.run(gspec, nn, duel, progress=lambda: count.__setitem__(0, count[0] + 1), seed=17)
    net = (1, 64, 32, 32, nn.params())
    c = dict(oracle=R.ORACLE_NET, nsims=20, cpuct=2.0, noise_eps=0.05, noise_alpha=1.0, temp_xs=(0,), temp_ys=(0.2,), net=net) \
        if player == "full" else dict(oracle=R.ORACLE_NET, nsims=0, noise_eps=0.0, temp_xs=(0,), temp_ys=(0.5,), net=net)
    b = dict(oracle=R.ORACLE_ROLLOUT, nsims=50, cpuct=1.0, noise_eps=0.05, noise_alpha=1.0, temp_xs=(0,), temp_ys=(0.2,))
    _, _, _, rew_ref, red_ref = R.arena(R.C4, 10, 5, c, b, alternate_colors=False, flip_probability=0.5, reset_every=2, seed=17)
    assert count[0] == 10 and np.array_equal(ev.rewards, rew_ref) and ev.redundancy == red_ref
    assert ev.legend == ("AlphaZero / MCTS (50 rollouts)" if player == "full" else "Network Only / MCTS (50 rollouts)")
    st = Benchmark.TernaryOutcomeStatistics.of(ev.rewards)
    assert st.num_won + st.num_draw + st.num_lost == 10


def test_network_player_records_and_errors():
    """records of a NetworkPlayer engine carry the Float32 policy; self-play and explore refuse an engine without search"""
    import azhip
    from azhip import _lib as L
    hp = dict(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    from azhip.network import ResNetHP, random_params
    blob = random_params(R.TTT, ResNetHP(1, 64, (3, 3), 32, 32), seed=3)
    kw = dict(game=L.GAME_TICTACTOE, num_workers=4, batch_size=4, temperature=([0], [0.5]), seed=8, cpuct=1.0, reset_every=1, **hp)
    with azhip.Engine(oracle=L.ORACLE_RESNET, num_iters_per_turn=0, **kw) as a, \
            azhip.Engine(oracle=L.ORACLE_ROLLOUT, num_iters_per_turn=40, dirichlet_noise_eps=0.05, **kw) as b:
        a.net_set_params(blob)
        games, moves, ng, nm, rew, red = a.arena_run(b, 6, alternate_colors=True)
        g_ref, m_ref, nm_ref, rew_ref, red_ref = R.arena(
            R.TTT, 6, 4, dict(oracle=R.ORACLE_NET, nsims=0, temp_xs=(0,), temp_ys=(0.5,), net=(1, 64, 32, 32, blob), seed=8),
            dict(oracle=R.ORACLE_ROLLOUT, nsims=40, noise_eps=0.05, temp_xs=(0,), temp_ys=(0.5,), seed=8), alternate_colors=True, seed=8)
        _same_records(games, moves, ng, g_ref, m_ref, nm_ref, 9)
        assert np.array_equal(rew, rew_ref) and red == red_ref
        assert any(moves[i].N[L.MAX_ACTIONS] & 0x100 for i in range(nm))
        with pytest.raises(L.AzError, match="NetworkPlayer"):
            a.selfplay_run(2)
    with pytest.raises(L.AzError, match="num_iters_per_turn"):
        azhip.Engine(oracle=L.ORACLE_HASH, num_iters_per_turn=0, game=L.GAME_TICTACTOE, num_workers=4, batch_size=4)
