"""Independent checks of the numerics the oracle and the kernels SHARE (include/az_numerics.h): a misreading there is
invisible to every HIP-vs-oracle comparison ("What's weak" 1), so the rules are restated here from their
sources in plain Python / numpy, without the header:
  * Philox4x32-10 counter layout and the (0,1) / [0,1) uniforms;
  * the Dirichlet draw: Marsaglia polar normals -> Marsaglia & Tsang Gamma (alpha < 1 boosted by U^(1/alpha)) ->
    normalisation, with libm's log / sqrt / pow (the header restates them in IEEE operations; agreement to ~1e-13 shows
    both the restated transcendentals and the sampler's control flow are what they claim), and its moments;
  * the categorical rule of Distributions.jl's DiscreteNonParametric sampler (`while cp <= u && i < n`) on Float32
    cumulative sums, against numpy's searchsorted and against the empirical frequencies it must produce."""
import ctypes as C
import math

import numpy as np

import azref as R

M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


class Stream:
    """counter = (game id, move, purpose, draw index), key = 64-bit seed; one block per draw"""

    def __init__(self, seed, game, move, purpose):
        self.key = (seed & M32, seed >> 32)
        self.ctr = [game, move, purpose, 0]

    def block(self):
        o = philox4x32_10(self.ctr, self.key)
        self.ctr[3] += 1
        return o

    def f64(self):
        o = self.block()
        return ((((o[0] << 32) | o[1]) >> 12) + 0.5) * 2.0 ** -52

    def f32(self):
        return np.float32((self.block()[2] >> 8) * 2.0 ** -24)


def randn(r):
    while True:
        a, b = 2.0 * r.f64() - 1.0, 2.0 * r.f64() - 1.0
        s = a * a + b * b
        if 0.0 < s < 1.0:
            return a * math.sqrt(-2.0 * math.log(s) / s)


def rand_gamma(r, alpha):
    boost = 1.0
    if alpha < 1.0:
        boost = r.f64() ** (1.0 / alpha)
        alpha += 1.0
    d = alpha - 1.0 / 3.0
    c = 1.0 / math.sqrt(9.0 * d)
    while True:
        x = randn(r)
        v = 1.0 + c * x
        if v <= 0.0:
            continue
        v = v * v * v
        if math.log(r.f64()) < 0.5 * x * x + d * (1.0 - v + math.log(v)):
            return boost * d * v


def dirichlet(seed, game, move, n, alpha):
    r = Stream(seed, game, move, 1)             # AZ_RNG_NOISE
    g = [rand_gamma(r, alpha) for _ in range(n)]
    s = 0.0
    for x in g:
        s += x
    return np.array([x / s for x in g])


def oracle_dirichlet(seed, game, move, n, alpha):
    eta = np.zeros(n)
    R.lib().azr_dirichlet(seed, game, move, n, alpha, eta.ctypes.data_as(C.c_void_p))
    return eta


def test_dirichlet_draws_agree_with_an_independent_restatement():
    for alpha in (1.0, 0.3, 0.03, 2.5, 10.0):
        for n in (2, 6, 7, 9):
            for game, move, seed in ((0, 0, 1), (17, 5, 1), (4095, 41, 0x1234567890abcdef), (123456, 7, 77)):
                a, b = oracle_dirichlet(seed, game, move, n, alpha), dirichlet(seed, game, move, n, alpha)
                assert np.allclose(a, b, rtol=1e-12, atol=1e-300), (alpha, n, game, move, np.abs(a - b).max())
                assert abs(a.sum() - 1.0) < 1e-14 and (a >= 0).all()


def test_dirichlet_moments():
    """E[eta_i] = 1/n, Var[eta_i] = (n - 1) / (n^2 (n alpha + 1)) -- 4000 draws per case"""
    for alpha, n in ((1.0, 7), (0.3, 7), (3.0, 6)):
        X = np.array([oracle_dirichlet(1, g, 0, n, alpha) for g in range(4000)])
        var = (n - 1) / (n * n * (n * alpha + 1))
        assert np.abs(X.mean(0) - 1.0 / n).max() < 5 * math.sqrt(var / 4000)
        assert np.abs(X.var(0) / var - 1.0).max() < 0.15


def test_move_uniform_is_the_float32_word_of_the_move_stream():
    L = R.lib()
    for game, move, seed in ((0, 0, 1), (9, 33, 1), (4000, 2, 99)):
        assert L.azr_move_uniform(seed, game, move) == float(Stream(seed, game, move, 2).f32())   # AZ_RNG_MOVE


def test_categorical_rule_against_searchsorted_and_frequencies():
    L = R.lib()
    rng = np.random.default_rng(5)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for _ in range(300):
        n = int(rng.integers(2, 10))
        p = rng.dirichlet(np.ones(n) * rng.choice([0.2, 1.0, 5.0]))
        pf = p.astype(np.float32)
        cum = np.zeros(n, dtype=np.float32)             # the Float32 running sum the sampler walks
        acc = np.float32(0)
        for i in range(n):
            acc = np.float32(acc + pf[i])
            cum[i] = acc
        for u in list(rng.random(20).astype(np.float32)) + [np.float32(0), cum[0], cum[-2], np.float32(1 - 2 ** -24)]:
            want = min(int(np.searchsorted(cum, u, side="right")), n - 1)   # first index with cum > u, clamped to the last
            assert L.azr_rand_categorical(vp(p), n, C.c_float(float(u))) == want, (p, u)
    # frequencies: the rule samples index i with probability p_i
    p = np.array([0.05, 0.4, 0.0, 0.25, 0.3])
    us = (np.arange(1 << 14) + 0.5) / (1 << 14)
    counts = np.bincount([L.azr_rand_categorical(vp(p), 5, C.c_float(float(u))) for u in us], minlength=5) / len(us)
    assert np.abs(counts - p).max() < 2e-4 and counts[2] == 0
