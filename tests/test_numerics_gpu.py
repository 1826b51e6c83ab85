"""The numerics contract (include/az_numerics.h) on gfx950 vs the host, bit for bit: IEEE f64 sqrt and division,
fp32 division, az_log / az_exp / az_pow, az_expf / az_tanhf and a whole Dirichlet draw (Philox + Gamma sampler)."""
import ctypes as C

import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def _dev(e, op, x, y=None):
    from azhip._lib import check, lib
    f = lib().az_debug_math
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(np.zeros_like(x) if y is None else y, dtype=np.float64)
    out = np.zeros_like(x)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(f(e._h, op, vp(x), vp(y), x.size, vp(out)))
    return out


def test_device_math_is_bit_identical_to_host():
    import azhip
    L = R.lib()
    rng = np.random.default_rng(0)
    with azhip.Engine(game=0, oracle=azhip.ORACLE_UNIFORM, num_workers=1, batch_size=1, num_iters_per_turn=2) as e:
        # sqrt of every visit-count total up to 2^22 and of random doubles: correctly rounded == numpy
        n = np.arange(0, 1 << 22, dtype=np.float64)
        assert np.array_equal(_dev(e, 0, n), np.sqrt(n))
        x = np.exp(rng.uniform(-50, 50, 500000))
        assert np.array_equal(_dev(e, 0, x), np.sqrt(x))
        # f64 and f32 division
        a, b = rng.uniform(-1e3, 1e3, 500000), np.exp(rng.uniform(-20, 20, 500000))
        assert np.array_equal(_dev(e, 1, a, b), a / b)
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        assert np.array_equal(_dev(e, 7, a32, b32).astype(np.float32), a32 / b32)
        # transcendental functions of the contract: device == oracle build of the same header
        xs = np.exp(rng.uniform(-40, 40, 20000))
        assert np.array_equal(_dev(e, 2, xs), np.array([L.azr_log(float(v)) for v in xs]))
        xe = rng.uniform(-700, 700, 20000)
        assert np.array_equal(_dev(e, 3, xe), np.array([L.azr_exp(float(v)) for v in xe]))
        px, py = rng.uniform(0, 1, 20000), rng.uniform(0.3, 5, 20000)
        assert np.array_equal(_dev(e, 4, px, py), np.array([L.azr_pow(float(u), float(v)) for u, v in zip(px, py)]))
        xf = rng.uniform(-90, 20, 20000).astype(np.float32)
        assert np.array_equal(_dev(e, 5, xf).astype(np.float32), np.array([L.azr_expf(float(v)) for v in xf], dtype=np.float32))
        xt = rng.uniform(-12, 12, 20000).astype(np.float32)
        assert np.array_equal(_dev(e, 6, xt).astype(np.float32), np.array([L.azr_tanhf(float(v)) for v in xt], dtype=np.float32))
        # Dirichlet(7, 0.3) draw number 3 for 2000 (seed, game) pairs
        seeds, games = rng.integers(0, 1 << 30, 2000).astype(np.float64), rng.integers(0, 1 << 20, 2000).astype(np.float64)
        eta = np.zeros(7)
        ref = []
        for s, g in zip(seeds, games):
            L.azr_dirichlet(int(s), int(g), 3, 7, 0.3, eta.ctypes.data_as(C.c_void_p))
            ref.append(eta[3])
        assert np.array_equal(_dev(e, 8, seeds, games), np.array(ref))
