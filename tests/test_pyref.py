"""Cross-check of the C oracle against an independent pure-Python restatement (oracle/pyref.py): the three
games' rules and the whole search (noise at the root, gamma != 1, Mancala's free turns / pswitch, hash
oracle) must agree exactly -- visit counts, W (Float64) and priors (Float32)."""
import numpy as np
import pytest

import azref as R
import pyref as Y


@pytest.mark.parametrize("game", [0, 1, 2])
def test_game_rules_agree(game):
    G = Y.GAMES[game]
    rng = np.random.default_rng(10 + game)
    for _ in range(150):
        g, o = G.init(), R.Game(game)
        while not Y.finished(G, g):
            assert not o.terminated() and G.key(g) == o.key()
            assert list(o.actions_mask()) == G.mask(g) and Y.white_playing(G, g) == o.white_playing()
            a = int(rng.choice([i for i, ok in enumerate(G.mask(g)) if ok]))
            g = G.play(g, a); o.play(a)
            assert G.reward(g) == o.white_reward()
        assert o.terminated() and G.key(g) == o.key()


@pytest.mark.parametrize("game,oracle,nsims", [(0, 1, 300), (1, 1, 200), (2, 1, 300), (2, 0, 200), (0, 0, 250)])
def test_search_agrees(game, oracle, nsims):
    G = Y.GAMES[game]
    rng = np.random.default_rng(game * 7 + oracle)
    for trial in range(4):
        g, o = G.init(), R.Game(game)
        for _ in range(int(rng.integers(0, 6))):
            acts = [i for i, ok in enumerate(G.mask(g)) if ok]
            a = int(rng.choice(acts))
            g2 = G.play(g, a)
            if Y.finished(G, g2):
                break
            g = g2; o.play(a)
        n = sum(G.mask(g))
        eta = rng.dirichlet(np.ones(n))
        y = Y.Mcts(G, Y.hash_oracle if oracle == 1 else Y.uniform_oracle, gamma=0.95, cpuct=1.3, eps=0.25)
        y.explore(g, nsims, list(eta))
        m = R.Mcts(game, oracle=oracle, gamma=0.95, cpuct=1.3, noise_eps=0.25)
        m.explore(o, nsims, eta=eta)
        N, W, P, V = m.root_stats(o)
        yN, yW, yP, yV = y.root_stats(g)
        assert list(N) == yN and list(W) == yW and [np.float32(p) for p in P] == yP and np.float32(V) == yV
        assert (m.total_simulations, m.total_nodes_traversed, m.num_nodes) == (y.total_simulations, y.total_nodes_traversed, len(y.tree))


@pytest.mark.parametrize("game", [0, 1])
def test_symmetries_agree(game):
    """GI.symmetries restated twice from games/*/game.jl (pyref: position maps composed like generate_dihedral_symmetries;
    azref.c: the coordinate form): same images in the same order on random positions."""
    G = Y.GAMES[game]
    rng = np.random.default_rng(game + 1)
    assert len(G.symmetries(G.init())) == R.lib().azr_num_symmetries(game)
    for _ in range(60):
        g = G.init()
        for _ in range(int(rng.integers(0, 9))):
            g2 = G.play(g, int(rng.choice([i for i, ok in enumerate(G.mask(g)) if ok])))
            if Y.finished(G, g2):
                break
            g = g2
        st = R.unpack_key(game, G.key(g))
        for k, img in enumerate(G.symmetries(g)):
            cells, cur = R.symmetry(game, list(st.cells), st.curplayer, k)
            o = R.State()
            for j, c in enumerate(cells):
                o.cells[j] = c
            o.curplayer = cur
            assert R.Game(game, o).key() == G.key(img), (game, k)
