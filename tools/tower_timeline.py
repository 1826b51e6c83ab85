"""Debug aid: per-workgroup s_memtime stamps of one k_tower16 launch (az_debug_tower_timeline)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip._lib import check, lib  # noqa: E402
from azhip.network import ResNetHP, random_params  # noqa: E402

hp = ResNetHP(5, 64, (3, 3), 32, 32)
e = azhip.Engine(game=0, oracle=2, num_workers=4096, batch_size=4096, num_iters_per_turn=8, num_blocks=5,
                 num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
e.net_set_params(random_params(0, hp))
f = lib().az_debug_tower_timeline
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
for n in (4096, 2048, 1024):
    nb = (n + 3) // 4
    out = np.zeros((nb, 8), dtype=np.uint64)
    check(f(e._h, n, out.ctypes.data_as(C.c_void_p), out.size))
    t = out[:, :4].astype(np.int64)
    last = 3
    t0 = t[:, 0].min()
    start, end = t[:, 0] - t0, t[:, last] - t0
    dur = t[:, last] - t[:, 0]
    print("n", n, "workgroups", nb, "kernel span", end.max(), "workgroup duration mean/min/max", dur.mean().round(), dur.min(), dur.max(), "(100 MHz ticks)")
    print(" segments mean (stem, tower, head conv + feature store):", np.diff(t[:, :last + 1], axis=1).mean(axis=0).round())
    print(" start quantiles", np.percentile(start, [0, 25, 50, 75, 90, 100]).round())
    print(" end quantiles", np.percentile(end, [0, 25, 50, 75, 90, 100]).round())
