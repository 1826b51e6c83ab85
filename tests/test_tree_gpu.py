"""Search-tree parity on the GPU: HIP kernels vs the CPU oracle, bit-exact (integer visit counts,
Float64 W, Float32 P) with NN-free oracles so that tree parity does not depend on network rounding."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu

GAMES = {0: "connect-four", 1: "tictactoe", 2: "mancala"}


def _engine(game, oracle, **kw):
    import azhip
    kw.setdefault("num_workers", 8)
    kw.setdefault("batch_size", 8)
    kw.setdefault("num_iters_per_turn", 64)
    return azhip.Engine(game=game, oracle=oracle, **kw)


@pytest.mark.parametrize("game,cpuct,nsims,expect", [
    (1, 1.0, 64, [7] * 9), (0, 2.0, 400, [57] * 7), (0, 2.0, 600, [86, 86, 86, 86, 85, 85, 85])])
def test_appendix_d_root_counts(game, cpuct, nsims, expect):
    """RandomOracle, no noise, fresh tree."""
    with _engine(game, 0, cpuct=cpuct, dirichlet_noise_eps=0.0) as e:
        key = e.init_key()
        e.mcts_explore([key], nsims, eta=np.zeros((1, 9)))
        N, W, P, V, mask = e.mcts_node_stats(0, key)
        assert list(N) == expect
        ts, tt, nn = e.mcts_counters(0)
        assert ts == nsims and nn == nsims


@pytest.mark.parametrize("game", [0, 1, 2])
@pytest.mark.parametrize("oracle", [0, 1, 3])
def test_explore_matches_oracle(game, oracle):
    """explore! from random reachable roots, uniform / hash / rollout oracle (MCTS.RolloutOracle, mcts.jl:35-60: the
    playouts draw from the RNG contract's stream of (seed, game, move, simulation)), given eta: N, W, P, Vest equal
    bit for bit."""
    rng = np.random.default_rng(7 + game)
    nslots, nsims = 8, 200
    roots, envs = [], []
    for s in range(nslots):
        while True:
            g = R.Game(game)
            for _ in range(int(rng.integers(0, 8))):
                if g.terminated():
                    break
                g.play(int(rng.choice(g.available_actions())))
            if not g.terminated():
                break
        roots.append(g.key()); envs.append(g)
    nA = R.NUM_ACTIONS[game]
    eta_full = np.zeros((nslots, 9))
    etas = []
    for s, g in enumerate(envs):
        acts = g.available_actions()
        eta = rng.dirichlet(np.ones(len(acts)))
        etas.append(eta)
        eta_full[s, acts] = eta
    with _engine(game, oracle, cpuct=1.7, dirichlet_noise_eps=0.25, gamma=0.97) as e:
        e.mcts_explore(roots, nsims, eta=eta_full, game_ids=np.arange(nslots) + 40, moves=np.arange(nslots) % 3)
        for s, g in enumerate(envs):
            m = R.Mcts(game, oracle=oracle, gamma=0.97, cpuct=1.7, noise_eps=0.25)
            m.explore(g, nsims, eta=etas[s], seed=1, game_id=s + 40, move=s % 3)
            N, W, P, V = m.root_stats(g)
            Nd, Wd, Pd, Vd, mask = e.mcts_node_stats(s, roots[s])
            acts = g.available_actions()
            assert list(Nd[acts]) == list(N), (GAMES[game], s)
            assert np.array_equal(Wd[acts], W)
            assert np.array_equal(Pd[acts], P)
            assert Vd == V
            ts, tt, nn = e.mcts_counters(s)
            assert (ts, tt, nn) == (m.total_simulations, m.total_nodes_traversed, m.num_nodes)


@pytest.mark.parametrize("game,nsims,ngames,workers,batch", [(1, 64, 32, 32, 32), (0, 100, 24, 8, 8), (2, 60, 12, 8, 8),
                                                              (1, 64, 40, 32, 8), (0, 100, 24, 8, 4)])
@pytest.mark.parametrize("oracle", [0, 1, 3])
def test_selfplay_traces_match_oracle(game, nsims, ngames, workers, batch, oracle, monkeypatch):
    """Whole self-play phase (simulate): every move record, visit count, action, reward, node count.
    batch < workers runs workers/batch interleaved slot groups on separate streams: same results."""
    kw = dict(gamma=1.0, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=(0, 4, 8), temp_ys=(1.0, 1.0, 0.3))
    if oracle == 1 and batch == workers:
        monkeypatch.setenv("AZHIP_GRAPH", "1")        # the opt-in hipGraph replay of wave pairs (single slot group) gives the same records
    with _engine(game, oracle, num_workers=workers, batch_size=batch, num_iters_per_turn=nsims, cpuct=2.0,
                 dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=((0, 4, 8), (1.0, 1.0, 0.3)),
                 reset_every=2, seed=11, max_moves_per_game=200 if game == 2 else 0) as e:
        dg, dm, ng, ndm, stats = e.selfplay_run(ngames)
    assert ng == ngames
    # (which worker takes which game is a race in the reference, util.jl:181-188: the oracle replays the outcome the device reports)
    games, moves, nm = R.simulate(game, oracle, ngames, workers, nsims, reset_every=2, seed=11, assignment=R.assignment_of(dg, ngames), **kw)
    assert ndm == nm
    for i in range(ngames):
        a, b = games[i], dg[i]
        assert (a.game_id, a.slot, a.num_moves, a.nodes, a.total_simulations, a.total_nodes_traversed) == \
               (b.game_id, b.slot, b.num_moves, b.nodes, b.total_simulations, b.total_nodes_traversed), i
        assert tuple(a.final_key) == tuple(b.final_key)
        for k in range(a.num_moves):
            x, y = moves[a.first_move + k], dm[b.first_move + k]
            assert tuple(x.key) == tuple(y.key) and list(x.N) == list(y.N), (i, k)
            assert x.action == y.action and x.reward == y.reward, (i, k)
    # phase statistics: every explore! is nsims simulations, so simulations = nsims x move records; a leaf is
    # evaluated at most once per simulation; total_simulations is cumulative per worker, so its sum over the workers' last
    # games is the phase total
    assert stats.games == ngames and stats.moves == nm and stats.simulations == nsims * nm
    assert 0 < stats.leaf_evals <= stats.simulations
    last = {}
    for g in games:
        last[g.slot] = max(last.get(g.slot, 0), g.total_simulations)
    assert sum(last.values()) == stats.simulations
