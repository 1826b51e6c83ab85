"""Seam 2: games WITHOUT a device twin keep their rules and their tree on the host and use
the HIP network through az_net_forward (= Network.forward_normalized on host tensors, src/networks/network.jl:264-271,
which is what the reference's inference server calls for any GameInterface / OpenSpiel game, src/openspiel.jl:7-173).

The host game here is "three in a row loses, four wins" Connect-Four -- different rules from the device twin, same 7x6x3
planes, so the 7x6 tower geometry serves it.  The host search is the independent pure-Python restatement of
src/mcts.jl (oracle/pyref.py) with an oracle that batches nothing: one az_net_forward call per new leaf, exactly like
`Network.evaluate(nn, state)` (network.jl:283-293).  The same search driven by the CPU oracle's network must produce the
same tree, visit count for visit count: the HIP forward is bit-identical to the fp32 contract."""
import numpy as np
import pytest

import azref as R
import pyref as Y

pytestmark = pytest.mark.gpu


class OddConnectFour(Y.ConnectFour):
    """host-only rules: making exactly three in a line (and not four) LOSES at once; four in a line wins as usual"""

    @classmethod
    def play(cls, g, col):
        b, cur, _, _ = g
        row = 0
        while b[col + 7 * row] != 0:
            row += 1
        b = list(b)
        b[col + 7 * row] = cur
        runs = [1 + cls._connected(b, cur, col, row, dc, dr) + cls._connected(b, cur, col, row, -dc, -dr)
                for dc, dr in ((1, 1), (1, -1), (1, 0), (0, 1))]
        win, lose = max(runs) >= 4, max(runs) == 3
        fin = win or lose or all(b[c + 35] != 0 for c in range(7))
        return (tuple(b), 3 - cur, fin, cur if win else (3 - cur) if lose else 0)


def planes_and_mask(g):
    """GI.vectorize_state / actions_mask of the host game: [empty, current player, opponent] planes, (C, H, W) memory"""
    b, cur = g[0], g[1]
    X = np.zeros((3, 6, 7), dtype=np.float32)
    for col in range(7):
        for row in range(6):
            v = b[col + 7 * row]
            X[0 if v == 0 else 1 if v == cur else 2, row, col] = 1.0
    return X, np.array(OddConnectFour.mask(g), dtype=np.float32)


def test_host_tree_on_the_hip_network_matches_the_cpu_network():
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=2, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=33)
    calls = [0]
    with azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_workers=1, batch_size=1, num_iters_per_turn=2,
                      num_blocks=2, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)

        def hip_oracle(G, g):
            X, A = planes_and_mask(g)
            P, V, _ = e.net_forward(X[None], A[None])
            calls[0] += 1
            return [np.float32(p) for p, ok in zip(P[0], A) if ok], np.float32(V[0])

        def cpu_oracle(G, g):
            X, A = planes_and_mask(g)
            P, V, _ = R.net_forward_normalized(R.C4, (2, 64, 32, 32), blob, X[None], A[None])
            return [np.float32(p) for p, ok in zip(P[0], A) if ok], np.float32(V[0])

        rng = np.random.default_rng(2)
        g = OddConnectFour.init()
        lost_by_three = False
        for move in range(12):                                     # a few moves of a game, trees kept between moves
            if Y.finished(OddConnectFour, g):
                break
            acts = [i for i, ok in enumerate(OddConnectFour.mask(g)) if ok]
            eta = list(rng.dirichlet(np.ones(len(acts))))
            trees = []
            for oracle in (hip_oracle, cpu_oracle):
                m = Y.Mcts(OddConnectFour, oracle, cpuct=2.0, eps=0.25)
                m.explore(g, 60, eta)
                trees.append(m)
            a, b = trees
            assert a.root_stats(g)[0] == b.root_stats(g)[0] and a.root_stats(g)[1] == b.root_stats(g)[1]
            assert a.tree.keys() == b.tree.keys() and a.total_nodes_traversed == b.total_nodes_traversed
            N = a.root_stats(g)[0]
            g = OddConnectFour.play(g, acts[int(np.argmax(N))])
            lost_by_three = lost_by_three or (g[2] and g[3] == g[1])  # the mover lost: the rule the device twin does not have
    assert calls[0] > 300
