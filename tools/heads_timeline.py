"""Debug aid: per-wavefront cycle stamps of one split k_heads16 launch (az_debug_heads_timeline)."""
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
ap.add_argument("--filters", type=int, default=128)
ap.add_argument("--n", type=int, default=128)
a = ap.parse_args()
hp = ResNetHP(1, a.filters, (3, 3), 32, 32)
e = azhip.Engine(game=0, oracle=2, num_workers=a.n, batch_size=a.n, num_iters_per_turn=8, num_blocks=1,
                 num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32)
e.net_set_params(random_params(0, hp))
f = lib().az_debug_heads_timeline
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
waves = a.filters // 16                               # wavefronts per workgroup (value tiles; the policy workgroup uses one of them)
nb = 2 * ((a.n + 15) // 16)
out = np.zeros((nb, waves, 4), dtype=np.uint64)
check(f(e._h, a.n, out.ctypes.data_as(C.c_void_p), out.size))
t = out.astype(np.int64)
for role, name in ((0, "value workgroups"), (1, "policy workgroups")):
    tr = t[role::2]
    t0 = tr[:, :, 0].min(axis=1, keepdims=True)
    print("k_heads16 (split), %d filters, %d boards, %s: cycles from the workgroup's first stamp, per wavefront" % (a.filters, a.n, name))
    for k, lab in enumerate(("start", "chain done", "barrier", "end")):
        print("  %-11s" % lab, (tr[:, :, k] - t0).mean(axis=0).round())
