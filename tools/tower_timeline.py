"""Debug aid: per-workgroup cycle stamps of one k_tower16 launch (az_debug_tower_timeline, Net16Dev::dbg).

    python tools/tower_timeline.py [--filters 64|128] [--nt 11|3] [--n 4096,2048,...]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip._lib import check, lib  # noqa: E402
from azhip.network import ResNetHP, random_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--filters", type=int, default=64)
ap.add_argument("--nt", type=int, default=11)
ap.add_argument("--n", default="4096,2048,1024")
ap.add_argument("--blocks", type=int, default=5)
ap.add_argument("--bf16", action="store_true")
a = ap.parse_args()
ns = [int(x) for x in a.n.split(",")]
hp = ResNetHP(a.blocks, a.filters, (3, 3), 32, 32)
e = azhip.Engine(game=0, oracle=2, num_workers=max(ns), batch_size=max(ns), num_iters_per_turn=8, num_blocks=a.blocks,
                 num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32, net_bf16=1 if a.bf16 else 0)
e.net_set_params(random_params(0, hp))
f = lib().az_debug_tower_timeline
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
tb = 4 if a.nt == 11 else 1
wg_per_tile = 2 if a.nt == 2 else 1                                # nt = 2: the split tower, two workgroups per board
for n in ns:
    nb = wg_per_tile * ((n + tb - 1) // tb)
    out = np.zeros((nb, 8), dtype=np.uint64)
    check(f(e._h, n, a.nt, out.ctypes.data_as(C.c_void_p), out.size))
    t = out.astype(np.int64)
    t = t[t[:, 3] != 0]                                                  # padding workgroups of the split tower leave no stamps
    t0 = t[:, 0].min()
    start, end = t[:, 0] - t0, t[:, 3] - t0
    dur = t[:, 3] - t[:, 0]
    print("F", a.filters, "NT", a.nt, "n", n, "workgroups", nb, "kernel span", end.max(), "workgroup duration mean/min/max", dur.mean().round(), dur.min(), dur.max(), "(100 MHz ticks)")
    print(" segments mean (stem, tower, head conv + feature store):", np.diff(t[:, :4], axis=1).mean(axis=0).round())
    print(" layer 2 (conv, wait at barrier, epilogue):", (t[:, 4] - t[:, 6]).mean().round(1), (t[:, 5] - t[:, 4]).mean().round(1), (t[:, 7] - t[:, 5]).mean().round(1))
    print(" start quantiles", np.percentile(start, [0, 25, 50, 75, 90, 100]).round())
    print(" end quantiles", np.percentile(end, [0, 25, 50, 75, 90, 100]).round())
