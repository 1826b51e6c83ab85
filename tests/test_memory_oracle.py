"""CPU checks of the oracle's replay-memory / learning-status restatement (oracle/azref.c; src/memory.jl:20-138,
src/learning.jl:17-90,148-190) against independent Python / torch-fp64 restatements written from the reference."""
import math

import numpy as np
import torch

import azref as R
from azhip.network import ResNetHP, random_params, split_params
from azhip.trace import policy_from_visits
from test_net import torch_forward_normalized

EPS32 = float(np.finfo(np.float32).eps)


def _phase(game, ngames=8, workers=4, nsims=24, seed=3):
    games, moves, nm = R.simulate(game, R.ORACLE_HASH, ngames, workers, nsims, noise_eps=0.25, seed=seed)
    S = []
    for i in range(ngames):
        g = games[i]
        ss = R.samples_from_trace(game, moves, g.first_move, g.num_moves, 0.9)
        S += [ss[k] for k in range(g.num_moves)]
    return games, moves, S


def test_push_trace_samples():
    """memory.jl:74-87: last position first, z discounted and side relative, t = plies to the end, pi = MCTS.policy"""
    for game in (R.C4, R.TTT, R.MANCALA):
        games, moves, S = _phase(game)
        k = 0
        for i in range(len(games)):
            g = games[i]
            n = g.num_moves
            wr = 0.0
            for j in reversed(range(n)):
                m = moves[g.first_move + j]
                wr = 0.9 * wr + float(m.reward)
                e = S[k + (n - 1 - j)]
                env = R.Game(game, R.unpack_key(game, (m.key[0], m.key[1])))
                assert (e.key[0], e.key[1]) == (m.key[0], m.key[1]) and e.n == 1 and e.t == float(n - j)
                assert e.z == (wr if env.white_playing() else -wr)
                mask = env.actions_mask()
                pi = policy_from_visits(list(m.N[:R.NUM_ACTIONS[game]]), mask)
                assert np.array_equal(np.array(e.pi[:R.NUM_ACTIONS[game]])[mask], pi) and all(e.pi[a] == 0 for a in range(R.NUM_ACTIONS[game]) if not mask[a])
            k += n


def _py_merge(S, nA):
    """merge_by_state / merge_samples (memory.jl:89-114) with Python floats, insertion-ordered dict"""
    groups = {}
    for e in S:
        groups.setdefault((e.key[0], e.key[1]), []).append(e)
    out = {}
    for key, es in groups.items():
        pi = [es[0].pi[a] for a in range(nA)]
        z, t, n = es[0].z, es[0].t, es[0].n
        for e in es[1:]:
            pi = [x + e.pi[a] for a, x in enumerate(pi)]
            z += e.z; t += e.t; n += e.n
        c = float(len(es))
        out[key] = ([x / c for x in pi], z / c, t / c, n)
    return out


def test_symmetries_and_merge():
    from azhip.game import ConnectFourSpec
    spec = ConnectFourSpec()
    games, moves, S = _phase(R.C4)
    A = R.augment_with_symmetries(R.C4, S)
    n = len(S)
    assert len(A) == 2 * n
    for i in range(n):
        a, b = A[i], A[n + i]
        assert (a.key[0], a.key[1]) == (S[i].key[0], S[i].key[1])
        (ka, kb), sigma = spec.symmetries((S[i].key[0], S[i].key[1]))[0]          # game.jl:252-257 restated in azhip/game.py
        assert (b.key[0], b.key[1]) == (ka, kb) and sigma == [7, 6, 5, 4, 3, 2, 1]
        assert [b.pi[j] for j in range(7)] == [S[i].pi[6 - j] for j in range(7)] and (b.z, b.t, b.n) == (S[i].z, S[i].t, S[i].n)
    M = R.merge_by_state(R.C4, A)
    ref = _py_merge(A, 7)
    assert len(M) == len(ref) and sum(m.n for m in M) == 2 * n
    keys = [(m.key[0], m.key[1]) for m in M]
    assert keys == sorted(keys)
    for m in M:
        pi, z, t, cnt = ref[(m.key[0], m.key[1])]
        assert [m.pi[a] for a in range(7)] == pi and (m.z, m.t, m.n) == (z, t, cnt)
    assert max(m.n for m in M) >= len(games)          # the initial position (and its mirror = itself) merges all games
    # tic-tac-toe: 7 images per sample, action permutation = board permutation
    _, _, St = _phase(R.TTT)
    At = R.augment_with_symmetries(R.TTT, St)
    assert len(At) == 8 * len(St)
    from test_arena_oracle import TTT_SYMS
    for i in (0, 3, len(St) - 1):
        for k in range(7):
            img = At[len(St) + 7 * i + k]
            assert [img.pi[j] for j in range(9)] == [St[i].pi[TTT_SYMS[k][j]] for j in range(9)]
            cells, cur = R.symmetry(R.TTT, list(R.unpack_key(R.TTT, (St[i].key[0], St[i].key[1])).cells),
                                    R.unpack_key(R.TTT, (St[i].key[0], St[i].key[1])).curplayer, k)
            st = R.State()
            for q, c in enumerate(cells):
                st.cells[q] = c
            st.curplayer = cur
            assert R.Game(R.TTT, st).key() == (img.key[0], img.key[1])


def test_convert_samples():
    games, moves, S = _phase(R.C4)
    M = R.merge_by_state(R.C4, R.augment_with_symmetries(R.C4, S))
    for policy in (0, 1, 2):
        W, X, A, P, V = R.convert_samples(R.C4, policy, M)
        for i, e in enumerate(M):
            w = 1.0 if policy == 0 else math.log2(e.n) + 1 if policy == 1 else float(e.n)       # learning.jl:22-29
            assert abs(W[i] - np.float32(w)) <= np.spacing(np.float32(w))
            if e.n & (e.n - 1) == 0:
                assert W[i] == np.float32(w)                                                     # exact on powers of two
            env = R.Game(R.C4, R.unpack_key(R.C4, (e.key[0], e.key[1])))
            assert np.array_equal(X[i].ravel(), env.vectorize()) and np.array_equal(A[i] > 0, env.actions_mask())
            assert np.array_equal(P[i], np.array(e.pi[:7], dtype=np.float64).astype(np.float32)) and V[i] == np.float32(e.z)


def _torch_status(game, hp, blob, data, l2, cinv, renorm, batch):
    """losses + learning_status (learning.jl:59-90,148-181) in fp64 torch, written from the reference's formulas"""
    W, X, A, P, V = [torch.tensor(np.asarray(x), dtype=torch.float64) for x in data]
    n = len(W)
    Wmean = W.mean()
    Hp = -(P * torch.log(P + EPS32) * W[:, None]).sum() / W.sum()
    p = split_params(game, hp, blob)
    reg = sum(float((np.asarray(v, dtype=np.float64) ** 2).sum()) for k, v in p.items() if not (k.endswith(".mean") or k.endswith(".var")))
    acc = np.zeros(6)
    wtot = 0.0
    for b0 in range(0, n, batch):
        sl = slice(b0, min(n, b0 + batch))
        Ph, Vh, Pinv = [torch.tensor(x) for x in torch_forward_normalized(game, hp, blob, X[sl].numpy(), A[sl].numpy())]
        w = W[sl]
        Lp = -(P[sl] * torch.log(Ph + EPS32) * w[:, None]).sum() / w.sum() - Hp
        Lv = (((Vh / renorm - V[sl] / renorm) ** 2) * w).sum() / w.sum()
        Lreg = l2 * reg
        Linv = cinv * (Pinv * w).sum() / w.sum()
        L = (w.mean() / Wmean) * (Lp + Lv + Lreg + Linv)
        Hn = -(Ph * torch.log(Ph + EPS32) * w[:, None]).sum() / w.sum()
        acc += np.array([float(L), float(Lp), float(Lv), float(Lreg), float(Linv), float(Hn)]) * float(w.sum())
        wtot += float(w.sum())
    return acc / wtot, float(Hp), float(Wmean)


def test_learning_status_vs_torch_fp64():
    for game, policy, batch in ((R.C4, 1, 50), (R.TTT, 2, 1000), (R.MANCALA, 0, 17)):
        hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
        blob = random_params(game, hp, seed=4)
        _, _, S = _phase(game, ngames=6, workers=3, nsims=16)
        M = R.merge_by_state(game, R.augment_with_symmetries(game, S))
        data = R.convert_samples(game, policy, M)
        st = R.learning_status(game, (1, 64, 32, 32), blob, data, l2=1e-4, nonvalidity_penalty=1.0, rewards_renormalization=2.0, batch=batch)
        ref, Hp, Wmean = _torch_status(game, hp, blob, data, 1e-4, 1.0, 2.0, batch)
        got = np.array([st.L, st.Lp, st.Lv, st.Lreg, st.Linv, st.Hpnet])
        assert np.allclose(got, ref, rtol=2e-5, atol=2e-6), (got, ref)
        assert abs(st.Hp - Hp) < 1e-5 and abs(st.Wmean - Wmean) < 1e-6
    st0 = R.learning_status(R.C4, (1, 64, 32, 32), random_params(R.C4, ResNetHP(1, 64, (3, 3), 32, 32), seed=4),
                            R.convert_samples(R.C4, 0, _phase(R.C4)[2]), l2=0.0, nonvalidity_penalty=0.0)
    assert st0.Lreg == 0.0 and st0.Linv == 0.0 and abs(st0.L - (st0.Lp + st0.Lv)) < 1e-6
