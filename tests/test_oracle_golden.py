"""The oracle against every known-answer the path has: Appendix D's RNG-free vectors,
the PLSchedule vector of src/schedule.jl:82-87, the Philox known-answer test, the numerics contract vs
libm, and the GameInterface invariants of src/scripts/test_game.jl on the oracle's games.  CPU only."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import azref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")
D = json.load(open(os.path.join(GOLD, "appendix_d.json")))
GID = {"connect-four": R.C4, "tictactoe": R.TTT, "mancala": R.MANCALA}


def test_numerics_selftest_and_philox_kat():
    L = R.lib()
    assert L.azr_numerics_selftest() == 0
    for ctr, key, out in D["philox4x32_10"]["kat"]:
        h = lambda v: int(v, 16) if isinstance(v, str) else v
        c = (C.c_uint32 * 4)(*[h(x) for x in ctr]); k = (C.c_uint32 * 2)(*[h(x) for x in key]); o = (C.c_uint32 * 4)()
        L.azr_philox(c, k, o)
        assert ["%08x" % x for x in o] == out


def _ulp32(a, b):
    return abs(int(np.float32(a).view(np.int32)) - int(np.float32(b).view(np.int32)))


def test_transcendentals_close_to_libm():
    L = R.lib()
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.uniform(-80, 10, 3000), rng.uniform(-1, 1, 1000)]).astype(np.float32):
        assert _ulp32(L.azr_expf(float(x)), np.exp(np.float64(x))) <= 2, x
    for x in rng.uniform(-12, 12, 3000).astype(np.float32):
        assert abs(L.azr_tanhf(float(x)) - math.tanh(float(x))) < 3e-7
    assert L.azr_expf(-100.0) == 0.0 and L.azr_tanhf(20.0) == 1.0 and L.azr_tanhf(-20.0) == -1.0
    for x in np.exp(rng.uniform(-40, 40, 3000)):
        assert abs(L.azr_log(float(x)) - math.log(x)) <= 2.3e-16 * max(1.0, abs(math.log(x)))
    for x in rng.uniform(-700, 700, 3000):
        assert abs(L.azr_exp(float(x)) / math.exp(x) - 1.0) < 5e-16
    for x, y in zip(rng.uniform(0, 1, 2000), rng.uniform(0.5, 5, 2000)):
        assert abs(L.azr_pow(float(x), float(y)) - x ** y) <= 1e-14 * max(x ** y, 1e-300)
    assert L.azr_pow(0.0, 3.0) == 0.0


def test_rng_streams_are_deterministic_and_sane():
    L = R.lib()
    eta = np.zeros(7); eta2 = np.zeros(7)
    L.azr_dirichlet(5, 3, 9, 7, 1.0, eta.ctypes.data_as(C.c_void_p))
    L.azr_dirichlet(5, 3, 9, 7, 1.0, eta2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(eta, eta2) and abs(eta.sum() - 1) < 1e-12 and (eta > 0).all()
    L.azr_dirichlet(5, 3, 10, 7, 1.0, eta2.ctypes.data_as(C.c_void_p))
    assert not np.array_equal(eta, eta2)
    # Dirichlet(7, alpha) marginal mean 1/7, variance (1/7)(6/7)/(7 alpha + 1)
    for alpha in (1.0, 0.3, 2.5):
        xs = np.zeros((4000, 7)); tmp = np.zeros(7)
        for g in range(4000):
            L.azr_dirichlet(1, g, 0, 7, alpha, tmp.ctypes.data_as(C.c_void_p)); xs[g] = tmp
        assert abs(xs.mean() - 1 / 7) < 1e-9 and abs(xs[:, 0].mean() - 1 / 7) < 0.01
        assert abs(xs[:, 0].var() - (1 / 7) * (6 / 7) / (7 * alpha + 1)) < 0.004
    us = np.array([L.azr_move_uniform(1, g, 0) for g in range(4000)])
    assert 0 <= us.min() and us.max() < 1 and abs(us.mean() - 0.5) < 0.02


@pytest.mark.parametrize("case", D["root_counts"], ids=lambda c: "%s-%d" % (c["game"], c["nsims"]))
def test_appendix_d_root_counts(case):
    g = R.Game(GID[case["game"]])
    m = R.Mcts(g.game, cpuct=case["cpuct"])
    m.explore(g, case["nsims"], eta=np.zeros(9))
    N, W, P, V = m.root_stats(g)
    assert list(N) == case["N"] and (W == 0).all()
    assert m.num_nodes == case["nodes"] and m.total_nodes_traversed == case["traversed"]
    assert m.total_simulations == case["nsims"] and sum(N) == case["nsims"] - 1      # Appendix A.2


@pytest.mark.parametrize("case", D["games"], ids=lambda c: c["game"])
def test_appendix_d_whole_games(case):
    game = GID[case["game"]]
    games, moves, nm = R.simulate(game, R.ORACLE_UNIFORM, 1, 1, case["nsims"], cpuct=case["cpuct"], temp_ys=(0.0,), reset_every=0)
    g = games[0]
    assert [moves[i].action + 1 for i in range(g.num_moves)] == case["moves"]
    assert moves[g.num_moves - 1].reward == case["white_reward"]
    assert (g.nodes, g.total_simulations, g.total_nodes_traversed) == (case["nodes"], case["total_simulations"], case["total_nodes_traversed"])
    for i, exp in enumerate(case["root_N"]):
        key = tuple(moves[i].key)
        env = R.Game(game, R.unpack_key(game, key))
        assert [moves[i].N[a] for a in env.available_actions()] == exp


def test_plschedule_known_answer():
    k = D["plschedule"]
    xs = (C.c_int * 3)(*k["xs"]); ys = (C.c_int * 3)(*k["ys"])
    assert [R.lib().azr_plschedule_int(xs, ys, 3, i) for i in k["at"]] == k["expect"]
    from azhip.params import ConstSchedule, PLSchedule
    s = PLSchedule(k["xs"], k["ys"])
    assert [s[i] for i in k["at"]] == k["expect"]
    f = PLSchedule([0, 20, 30], [1.0, 1.0, 0.3])
    yd = (C.c_double * 3)(1.0, 1.0, 0.3); xd = (C.c_int * 3)(0, 20, 30)
    for i in range(-2, 45):
        assert f[i] == R.lib().azr_plschedule(xd, yd, 3, i)
    assert ConstSchedule(0.5)[7] == 0.5 and f[25] == 1.0 + (0.3 - 1.0) / 10 * 5


def test_temperature_and_sampling_rules():
    """util.jl:68-110"""
    L = R.lib()
    pi = np.array([0.2, 0.5, 0.3]); out = np.zeros(3)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L.azr_apply_temperature(vp(pi), 3, C.c_double(1.0), vp(out)); assert np.array_equal(out, pi)
    L.azr_apply_temperature(vp(pi), 3, C.c_double(0.0), vp(out)); assert list(out) == [0, 1, 0]
    tie = np.array([0.4, 0.4, 0.2]); L.azr_apply_temperature(vp(tie), 3, C.c_double(0.0), vp(out)); assert list(out) == [1, 0, 0]
    L.azr_apply_temperature(vp(pi), 3, C.c_double(0.5), vp(out)); assert np.allclose(out, pi ** 2 / (pi ** 2).sum(), rtol=1e-14)
    assert [L.azr_rand_categorical(vp(pi), 3, C.c_float(u)) for u in (0.0, 0.19, 0.2, 0.69, 0.7, 0.999)] == [0, 0, 1, 1, 2, 2]
    z = np.zeros(4)
    assert [L.azr_rand_categorical(vp(z), 4, C.c_float(u)) for u in (0.1, 0.3, 0.6, 0.9)] == [0, 1, 2, 3]   # uniform when sum == 0
    from azhip.play import apply_temperature, rand_categorical
    assert np.array_equal(apply_temperature(pi, 0), [0, 1, 0]) and rand_categorical(pi, np.float32(0.69)) == 1


@pytest.mark.parametrize("game", [R.C4, R.TTT, R.MANCALA])
def test_game_interface_invariants(game):
    """src/scripts/test_game.jl:37-110 on 100 random games of the oracle's restatement."""
    rng = np.random.default_rng(game)
    w, h, c = R.DIMS[game]
    for _ in range(100):
        g = R.Game(game)
        nmoves = 0
        while not g.terminated():
            st = g.state()
            g2 = R.Game(game, st)                                  # init(gspec, state) round trip
            assert g2.key() == g.key() and g2.white_playing() == g.white_playing()
            mask = g.actions_mask()
            assert mask.any() and len(mask) == R.NUM_ACTIONS[game]  # game_terminated || any(mask)
            x = g.vectorize()
            assert x.dtype == np.float32 and x.size == w * h * c
            assert R.unpack_key(game, g.key()).cells[:] == st.cells[:]
            cl = g.clone(); a = int(rng.choice(g.available_actions())); cl.play(a)
            assert g.key() == R.Game(game, st).key()               # state persistence
            g.play(a); nmoves += 1
            assert g.key() == cl.key()
        assert g.white_reward() in (-1.0, 0.0, 1.0)
        assert nmoves <= (42 if game == R.C4 else 9 if game == R.TTT else 200)


def test_connect_four_symmetry_and_known_positions():
    """symmask == mask[sigma] (test_game.jl:23-27) and the 6000 benchmark positions are legal, non-terminal."""
    lines = open(os.path.join(GOLD, "c4_positions.txt")).read().split()
    assert len(lines) == 6000 and max(map(len, lines)) <= 41
    for mv in lines[::7]:
        g, gm = R.Game(R.C4), R.Game(R.C4)
        for ch in mv:
            assert not g.terminated() and g.actions_mask()[int(ch) - 1]
            g.play(int(ch) - 1); gm.play(7 - int(ch))
        assert not g.terminated()
        assert list(gm.actions_mask()) == list(g.actions_mask()[::-1])
        assert np.array_equal(gm.vectorize().reshape(3, 6, 7), g.vectorize().reshape(3, 6, 7)[:, :, ::-1])


def test_connect_four_rules_reproduce_the_reference_solver_scores():
    """A pin to data the reference ships: games/connect-four/benchmark/Test_L*_R* hold positions WITH their exact
    game-theoretic scores (Pons' convention; scripts/pons_benchmark.jl:49-98 measures the network against them).  A plain
    negamax over the oracle's own Connect-Four rules -- legal moves, the four-in-a-row test of `play!`, termination on a
    full board (games/connect-four/game.jl:50-146) -- reproduces every recorded score of the 1000 end-game positions
    (Test_L3_R1) and of the middle-game positions (Test_L2_R1) it can solve within the node budget.  The device twin of
    the game is tied to these rules move by move (tests/test_game_gpu.py)."""
    lines = [l.split() for l in open(os.path.join(GOLD, "c4_scores.txt"))]
    assert len(lines) == 1300
    solved = {"Test_L3_R1": 0, "Test_L2_R1": 0}
    for name, mv, sc in lines:
        got, nodes = R.c4_solve([int(c) - 1 for c in mv], 10_000_000 if name == "Test_L3_R1" else 200_000)
        if got == 98:
            continue                                               # over the node budget (middle game only)
        assert got == int(sc), (name, mv, sc, got)
        solved[name] += 1
    assert solved["Test_L3_R1"] == 1000 and solved["Test_L2_R1"] >= 200
    assert R.c4_solve([3, 3, 3, 3, 3, 3, 3])[0] == 99             # a full column: illegal move string


def test_tictactoe_rules_enumerate_the_known_game_tree():
    """Every Tic-tac-toe game under the oracle's rules (games/tictactoe/game.jl: first player WHITE, three in a row wins,
    full board = draw): 255 168 games -- 131 184 won by the first player, 77 904 by the second, 46 080 draws -- over
    549 946 nodes: the textbook counts."""
    res, nodes = {1.0: 0, -1.0: 0, 0.0: 0}, [0]

    def rec(g):
        nodes[0] += 1
        if g.terminated():
            res[g.white_reward()] += 1
            return
        for a in g.available_actions():
            c = g.clone()
            c.play(int(a))
            rec(c)
    rec(R.Game(R.TTT))
    assert (res[1.0], res[-1.0], res[0.0], nodes[0]) == (131184, 77904, 46080, 549946)


def test_mancala_flip_colors_bug_is_reproduced():
    """games/mancala/game.jl:224-229: with black to move the planes show the INITIAL board."""
    g = R.Game(R.MANCALA)
    g.play(0)                       # 3 seeds from house 1 -> last seed in the store? no: houses; then black
    while g.white_playing() and not g.terminated():
        g.play(int(g.available_actions()[0]))
    x = g.vectorize().reshape(5, 14)
    assert not g.white_playing()
    assert list(x[0]) == [3] * 6 + [0] + [3] * 6 + [0]
