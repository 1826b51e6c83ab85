"""CPU checks of the oracle's arena restatement (oracle/azref.c azr_arena; src/training.jl:130-144,
src/simulations.jl:207-244,296-311, src/play.jl:248-315): symmetries against the reference's definitions,
reward / colour bookkeeping, redundancy, replay of every recorded game through the game rules."""
import numpy as np
import pytest

import azref as R

# games/tictactoe/game.jl:149-160 evaluated by hand: sym[p] (0-based) for rot, rot2, rot3, flip, flip.rot, ...
TTT_SYMS = [[6, 3, 0, 7, 4, 1, 8, 5, 2], [8, 7, 6, 5, 4, 3, 2, 1, 0], [2, 5, 8, 1, 4, 7, 0, 3, 6], [6, 7, 8, 3, 4, 5, 0, 1, 2],
            [0, 3, 6, 1, 4, 7, 2, 5, 8], [2, 1, 0, 5, 4, 3, 8, 7, 6], [8, 5, 2, 7, 4, 1, 6, 3, 0]]


def test_symmetries_follow_the_reference_definitions():
    cells = list(range(10, 19))
    for k in range(7):
        out, cur = R.symmetry(R.TTT, cells, 2, k)
        assert out[:9] == [cells[q] for q in TTT_SYMS[k]] and cur == 2
    c4 = [(i * 7) % 3 for i in range(42)]
    out, cur = R.symmetry(R.C4, c4, 1, 0)                       # flipped_board, games/connect-four/game.jl:243-250
    assert all(out[c + 7 * r] == c4[6 - c + 7 * r] for c in range(7) for r in range(6)) and cur == 1


def _replay(game, g, moves):
    """re-run one recorded arena game through the oracle's game rules, applying the recorded symmetries"""
    env = R.Game(game)
    for k in range(g.num_moves):
        m = moves[g.first_move + k]
        assert env.key() == (m.key[0], m.key[1])
        sym = m.N[R.AMAX]
        if sym:
            cells, cur = R.symmetry(game, list(env.state().cells), env.state().curplayer, sym - 1)
            st = R.State()
            for i, c in enumerate(cells):
                st.cells[i] = c
            st.curplayer = cur
            env = R.Game(game, st)
        mask = env.actions_mask()
        assert mask[m.action] and sum(m.N[a] for a in range(R.NUM_ACTIONS[game])) > 0
        assert all(m.N[a] == 0 for a in range(R.NUM_ACTIONS[game]) if not mask[a])
        env.play(m.action)
        assert abs(env.white_reward() - m.reward) == 0
    assert env.terminated() and env.key() == (g.final_key[0], g.final_key[1])
    return env.white_reward()


def test_arena_bookkeeping():
    pl = dict(oracle=R.ORACLE_HASH, nsims=30, cpuct=2.0, noise_eps=0.05, noise_alpha=1.0, temp_xs=(0,), temp_ys=(0.2,))
    for game, flip in ((R.TTT, 0.5), (R.C4, 0.5), (R.MANCALA, 0.0)):
        n = 12
        games, moves, nm, rewards, red = R.arena(game, n, 5, pl, dict(pl, nsims=8), alternate_colors=True,
                                                 flip_probability=flip, reset_every=2, seed=3)
        keys = []
        for i in range(n):
            g = games[i]
            assert g.game_id == i
            wr = _replay(game, g, moves)
            flipped = (i + 1) % 2 == 1                            # simulations.jl:221-223 (sim_id is 1-based)
            assert rewards[i] == (-wr if flipped else wr)
            keys += [(moves[g.first_move + k].key[0], moves[g.first_move + k].key[1]) for k in range(g.num_moves)]
            keys.append((g.final_key[0], g.final_key[1]))
        assert abs(red - (1.0 - len(set(keys)) / len(keys))) < 1e-15
        assert nm == sum(games[i].num_moves for i in range(n))
        if flip:
            assert any(moves[i].N[R.AMAX] for i in range(nm)) and not all(moves[i].N[R.AMAX] for i in range(nm))
    # the stronger searcher wins the series on tic-tac-toe regardless of colours
    _, _, _, rewards, _ = R.arena(R.TTT, 40, 8, dict(pl, nsims=60), dict(pl, nsims=3), alternate_colors=True, seed=4)
    assert np.mean(rewards) > 0.3


def test_arena_without_flips_or_swaps_is_white_vs_black():
    """alternate_colors=false, flip 0: the contender is always WHITE, so its first-move tree statistics equal a
    plain explore! from the initial state (mcts.jl:239-245) with the game's noise stream."""
    pl = dict(oracle=R.ORACLE_HASH, nsims=25, cpuct=1.5, noise_eps=0.25, noise_alpha=1.0)
    games, moves, nm, rewards, red = R.arena(R.C4, 3, 3, pl, dict(pl, nsims=10), seed=9)
    for i in range(3):
        m = moves[games[i].first_move]
        e = R.Mcts(R.C4, R.ORACLE_HASH, cpuct=1.5, noise_eps=0.25, noise_alpha=1.0)
        g0 = R.Game(R.C4)
        e.explore(g0, 25, seed=9, game_id=i, move=0)
        N = e.root_stats(g0)[0]
        assert list(m.N[:7]) == [int(x) for x in N]
        wr = _replay(R.C4, games[i], moves)
        assert rewards[i] == wr


def _u64(seed, game, move, purpose, draw):
    """one az_rng_f64 draw restated from the RNG contract (include/az_numerics.h): Philox block -> 52-bit uniform"""
    import ctypes as C
    c = (C.c_uint32 * 4)(game, move, purpose, draw)
    k = (C.c_uint32 * 2)(seed & 0xffffffff, seed >> 32)
    o = (C.c_uint32 * 4)()
    R.lib().azr_philox(c, k, o)
    return ((((o[0] << 32) | o[1]) >> 12) + 0.5) * 2.0 ** -52


def test_rollout_oracle_against_python_restatement():
    """MCTS.RolloutOracle (mcts.jl:35-60): the C oracle's tree after explore! equals an independent Python MCTS
    (oracle/pyref.py games + search) whose oracle plays the rollout with draws taken from the Philox contract."""
    import pyref
    seed, gid, move = 5, 3, 2
    for game in (R.C4, R.TTT, R.MANCALA):
        G = pyref.GAMES[game]
        sim = [0]

        def rollout_oracle(G_, g):
            acts = [a for a, ok in enumerate(G_.mask(g)) if ok]
            P = [np.float32(1.0 / len(acts))] * len(acts)
            wp = pyref.white_playing(G_, g)
            draw = sim[0] * 1024
            wr = 0.0
            while not pyref.finished(G_, g):                       # rollout!, mcts.jl:41-50 (rewards are 0 before the end)
                av = [a for a, ok in enumerate(G_.mask(g)) if ok]
                k = min(int(_u64(seed, gid, move, 4, draw) * len(av)), len(av) - 1)
                draw += 1
                g = G_.play(g, av[k])
                wr = G_.reward(g)
            return P, np.float32(wr if wp else -wr)
        m = pyref.Mcts(G, rollout_oracle, gamma=1.0, cpuct=1.3)
        g0 = G.init()
        for a in ((3, 3, 2) if game == R.C4 else (4, 0) if game == R.TTT else (2, 5)):
            g0 = G.play(g0, a)
        for i in range(80):
            sim[0] = i
            m.explore(g0, 1, None)
        e = R.Mcts(game, R.ORACLE_ROLLOUT, cpuct=1.3)
        env = R.Game(game)
        for a in ((3, 3, 2) if game == R.C4 else (4, 0) if game == R.TTT else (2, 5)):
            env.play(a)
        e.explore(env, 80, eta=np.zeros(9), seed=seed, game_id=gid, move=move)
        N, W, P, V = e.root_stats(env)
        Np, Wp, Pp, Vp = m.root_stats(g0)
        assert [int(x) for x in N] == Np and [float(x) for x in W] == Wp and float(V) == float(Vp), game


def test_network_only_and_rollout_players_in_the_arena():
    """Benchmark.Duel(NetworkOnly / Full, MctsRollouts) (benchmark.jl:124-192) through the oracle's arena: games
    replay through the rules, NetworkPlayer moves carry the policy bits, rollout players beat the random net."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "alphazero.jl_amd"))
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(R.TTT, hp, seed=3)
    net = (1, 64, 32, 32, blob)
    netonly = dict(oracle=R.ORACLE_NET, nsims=0, temp_xs=(0,), temp_ys=(0.5,), net=net)
    roll = dict(oracle=R.ORACLE_ROLLOUT, nsims=40, cpuct=1.0, noise_eps=0.05, noise_alpha=1.0, temp_xs=(0,), temp_ys=(0.2,))
    games, moves, nm, rewards, red = R.arena(R.TTT, 16, 8, netonly, roll, alternate_colors=True, flip_probability=0.5, seed=8)
    nnet = 0
    for i in range(16):
        _replay_any(R.TTT, games[i], moves)
        for k in range(games[i].num_moves):
            m = moves[games[i].first_move + k]
            if m.N[R.AMAX] & 0x100:
                nnet += 1
                P = np.array(m.N[:9], dtype=np.int32).view(np.float32)
                assert abs(float(P.sum()) - 1.0) < 1e-5 and (P >= 0).all()
    assert nnet > 0 and np.mean(rewards) < 0        # 40 rollouts per move beat an untrained policy


def _replay_any(game, g, moves):
    env = R.Game(game)
    for k in range(g.num_moves):
        m = moves[g.first_move + k]
        assert env.key() == (m.key[0], m.key[1])
        sym = m.N[R.AMAX] & 0xff
        if sym:
            cells, cur = R.symmetry(game, list(env.state().cells), env.state().curplayer, sym - 1)
            st = R.State()
            for i, c in enumerate(cells):
                st.cells[i] = c
            st.curplayer = cur
            env = R.Game(game, st)
        assert env.actions_mask()[m.action]
        env.play(m.action)
    assert env.terminated() and env.key() == (g.final_key[0], g.final_key[1])


@pytest.mark.parametrize("game,flip", [(R.C4, 0.5), (R.TTT, 0.7), (R.TTT, 1.0)])
def test_self_play_flips_follow_play_game(game, flip):
    """azr_simulate with flip_probability (play.jl:299-313), rebuilt move by move from the oracle's single-step entry
    points: trace.states[i] is the state BEFORE the turn's flip, the turn is flipped iff the first f64 draw of the
    (game, move, FLIP) stream is below p, the image is symmetries[floor(u2 * n)], the player thinks and plays on the
    image, and the visit counts land on the un-flipped state's available actions by rank (learning.jl:31-33)."""
    import ctypes as C
    nsims, seed, n = 20, 6, 5
    kw = dict(cpuct=1.5, noise_eps=0.25, noise_alpha=1.0, seed=seed)
    games, moves, nm = R.simulate(game, R.ORACLE_HASH, n, n, nsims, reset_every=1, flip_probability=flip, **kw)
    L = R.lib()
    L.azr_stream_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    nsym = L.azr_num_symmetries(game)
    A = R.NUM_ACTIONS[game]
    nflip = 0
    for i in range(n):
        g = games[i]
        env, tree = R.Game(game), R.Mcts(game, R.ORACLE_HASH, cpuct=1.5, noise_eps=0.25, noise_alpha=1.0)
        for k in range(g.num_moves):
            m = moves[g.first_move + k]
            assert env.key() == (m.key[0], m.key[1]), (i, k)
            pre_avail = list(env.available_actions())
            u64, u32 = np.zeros(2), np.zeros(2, np.float32)
            L.azr_stream_uniforms(seed, i, k, 3, 2, u64.ctypes.data, u32.ctypes.data)       # purpose 3 = FLIP
            if u64[0] < flip:
                ks = min(int(u64[1] * nsym), nsym - 1)
                assert m.N[R.AMAX] == ks + 1, (i, k)
                st = env.state()
                cells, cur = R.symmetry(game, list(st.cells), st.curplayer, ks)
                img = R.State()
                for j, c in enumerate(cells):
                    img.cells[j] = c
                img.curplayer = cur
                env = R.Game(game, img)
                nflip += 1
            else:
                assert m.N[R.AMAX] == 0, (i, k)
            tree.explore(env, nsims, seed=seed, game_id=i, move=k)
            N = tree.root_stats(env)[0]
            assert len(N) == len(pre_avail)
            want = [0] * A
            for a, c in zip(pre_avail, N):
                want[a] = int(c)
            assert [m.N[a] for a in range(A)] == want, (i, k)
            assert env.actions_mask()[m.action]                      # the action is one of the IMAGE's
            env.play(m.action)
            assert env.white_reward() == m.reward
        assert env.terminated() and env.key() == (g.final_key[0], g.final_key[1])
    assert nflip == nm if flip == 1.0 else 0 < nflip < nm
