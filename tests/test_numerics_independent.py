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
import os

import numpy as np

import azref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the oracle no longer includes the product's header.  oracle/ref_numerics.h (the oracle's own implementation) against
# include/az_numerics.h (what libazhip.so compiles, host side and kernels), bit by bit, both built here with gcc.
_CONTRACT_SHIM = r"""
#include "az_numerics.h"
float c_expf(float x) { return az_expf(x); }
float c_tanhf(float x) { return az_tanhf(x); }
double c_log(double x) { return az_log(x); }
double c_log2(double x) { return az_log2(x); }
float c_logf(float x) { return az_logf(x); }
double c_exp(double x) { return az_exp(x); }
double c_pow(double x, double y) { return az_pow(x, y); }
void c_philox(const uint32_t* c, const uint32_t* k, uint32_t* o) { az_philox4x32_10(c, k, o); }
uint64_t c_hash_key(uint64_t a, uint64_t b) { return az_hash_key(a, b); }
int c_categorical(const float* p, int n, float u) { return az_categorical_f32(p, n, u); }
void c_dirichlet(uint64_t seed, uint32_t game, uint32_t move, int n, double alpha, double* eta) {
  az_rng r = az_rng_make(seed, game, move, AZ_RNG_NOISE); az_dirichlet(&r, n, alpha, eta);
}
void c_stream_uniforms(uint64_t seed, uint32_t game, uint32_t move, uint32_t purpose, int n, double* u64, float* u32) {
  az_rng r = az_rng_make(seed, game, move, purpose);
  for (int i = 0; i < n; ++i) u64[i] = az_rng_f64(&r);
  for (int i = 0; i < n; ++i) u32[i] = az_rng_f32(&r);
}
int c_selftest(void) { return az_numerics_selftest(); }
"""


def _contract_lib(tmp_path_factory):
    import ctypes
    import subprocess
    d = tmp_path_factory.mktemp("contract")
    (d / "shim.c").write_text(_CONTRACT_SHIM)
    so = d / "libcontract.so"
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-ffp-contract=off", "-mavx2", "-mfma", "-shared", "-I", os.path.join(ROOT, "include"),
                           "-o", str(so), str(d / "shim.c"), "-lm"])
    return ctypes.CDLL(str(so))


def test_the_oracle_s_own_numerics_equal_the_product_header_bit_for_bit(tmp_path_factory):
    import ctypes as C
    import re
    src = open(os.path.join(ROOT, "oracle", "azref.c")).read()
    assert "az_numerics.h\"" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S)          # the oracle includes only its own ref_numerics.h
    assert "az_numerics" not in open(os.path.join(ROOT, "oracle", "ref_numerics.h")).read().split("#ifndef")[1]
    P, O = _contract_lib(tmp_path_factory), R.lib()
    assert P.c_selftest() == 0 and O.azr_numerics_selftest() == 0
    rng = np.random.default_rng(4)

    def cmp1(pf, of, xs, ctype, npdt):
        pf.restype = of.restype = ctype
        pf.argtypes = of.argtypes = [ctype]
        a = np.array([pf(ctype(float(x))) for x in xs], dtype=npdt)
        b = np.array([of(ctype(float(x))) for x in xs], dtype=npdt)
        assert np.array_equal(a.view(np.uint32 if npdt == np.float32 else np.uint64), b.view(np.uint32 if npdt == np.float32 else np.uint64))
    f32 = np.concatenate([rng.uniform(-100, 100, 4000), rng.normal(0, 3, 4000), [0.0, -0.0, 88.0, 89.0, -86.0, -87.0, 9.0, 9.5, 1e-30, np.inf, -np.inf]]).astype(np.float32)
    cmp1(P.c_expf, O.azr_expf, f32, C.c_float, np.float32)
    cmp1(P.c_tanhf, O.azr_tanhf, f32, C.c_float, np.float32)
    pos = np.concatenate([np.exp(rng.uniform(-700, 700, 4000)), rng.uniform(0, 2, 4000), [1.0, 2.0, 0.5, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308]])
    cmp1(P.c_log, O.azr_log, pos, C.c_double, np.float64)
    cmp1(P.c_log2, O.azr_log2, pos[pos >= 1.0], C.c_double, np.float64)
    cmp1(P.c_logf, O.azr_logf, pos[(pos > 1e-30) & (pos < 1e30)].astype(np.float32), C.c_float, np.float32)
    cmp1(P.c_exp, O.azr_exp, np.concatenate([rng.uniform(-720, 720, 6000), [0.0, 709.0, 710.0, -708.0, -709.0]]), C.c_double, np.float64)
    P.c_pow.restype = O.azr_pow.restype = C.c_double
    P.c_pow.argtypes = O.azr_pow.argtypes = [C.c_double, C.c_double]
    for x, y in zip(rng.uniform(0, 1, 3000), rng.uniform(0.1, 12, 3000)):
        assert P.c_pow(x, y) == O.azr_pow(x, y)
    assert P.c_pow(0.0, 2.0) == O.azr_pow(0.0, 2.0) == 0.0
    # Philox blocks, stream uniforms (f64 then f32 of one stream: the draw counter), Dirichlet draws, the categorical walk, the key hash
    for _ in range(200):
        c = (C.c_uint32 * 4)(*rng.integers(0, 2**32, 4, dtype=np.uint64).tolist())
        k = (C.c_uint32 * 2)(*rng.integers(0, 2**32, 2, dtype=np.uint64).tolist())
        o1, o2 = (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
        P.c_philox(c, k, o1); O.azr_philox(c, k, o2)
        assert list(o1) == list(o2)
    sig = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    P.c_stream_uniforms.argtypes = O.azr_stream_uniforms.argtypes = sig
    for purpose in (1, 2, 3, 4, 5):
        seed, game, move = int(rng.integers(0, 2**63)), int(rng.integers(0, 2**31)), int(rng.integers(0, 300))
        a64, b64, a32, b32 = np.zeros(40), np.zeros(40), np.zeros(40, np.float32), np.zeros(40, np.float32)
        P.c_stream_uniforms(seed, game, move, purpose, 40, a64.ctypes.data, a32.ctypes.data)
        O.azr_stream_uniforms(seed, game, move, purpose, 40, b64.ctypes.data, b32.ctypes.data)
        assert np.array_equal(a64, b64) and np.array_equal(a32, b32) and (a64 > 0).all() and (a64 < 1).all()
    P.c_dirichlet.argtypes = O.azr_dirichlet.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_void_p]
    for alpha in (0.03, 0.3, 1.0, 1.7, 10.0):
        for n in (1, 2, 6, 7, 9):
            for g in range(12):
                e1, e2 = np.zeros(n), np.zeros(n)
                P.c_dirichlet(77 + g, g, 3 * g, n, alpha, e1.ctypes.data); O.azr_dirichlet(77 + g, g, 3 * g, n, alpha, e2.ctypes.data)
                assert np.array_equal(e1, e2) and abs(e1.sum() - 1) < 1e-12
    P.c_categorical.argtypes = O.azr_categorical.argtypes = [C.c_void_p, C.c_int, C.c_float]
    for _ in range(500):
        n = int(rng.integers(1, 10))
        p = rng.dirichlet(np.ones(n)).astype(np.float32)
        u = np.float32(rng.uniform())
        assert P.c_categorical(p.ctypes.data, n, u) == O.azr_categorical(p.ctypes.data, n, u)
    assert P.c_categorical(np.array([0.5, 0.5], np.float32).ctypes.data, 2, np.float32(1.0)) == O.azr_categorical(np.array([0.5, 0.5], np.float32).ctypes.data, 2, np.float32(1.0)) == 1
    P.c_hash_key.restype = O.azr_hash_key.restype = C.c_uint64
    P.c_hash_key.argtypes = O.azr_hash_key.argtypes = [C.c_uint64, C.c_uint64]
    for a, b in rng.integers(0, 2**63, (300, 2), dtype=np.uint64).tolist():
        assert P.c_hash_key(a, b) == O.azr_hash_key(a, b)
