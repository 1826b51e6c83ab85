"""Debug aid: cycle stamps of k_tree's first wavefront (az_debug_tree_stamps) over a number of waves."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip._lib import check, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slots", type=int, default=4096)
ap.add_argument("--waves", type=int, default=60)
a = ap.parse_args()
e = azhip.Engine(game=0, oracle=azhip.ORACLE_HASH, num_workers=a.slots, batch_size=a.slots, num_iters_per_turn=200, cpuct=2.0,
                 dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, reset_every=1, max_nodes_per_slot=600)
f = lib().az_debug_tree_stamps
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p]
e.selfplay_begin(-1, 0)
e.selfplay_step(150)
check(f(e._h, None))
rows = []
for _ in range(a.waves):
    e.selfplay_step(1)
    out = np.zeros(16, dtype=np.uint64)
    check(f(e._h, out.ctypes.data_as(C.c_void_p)))
    rows.append(out.astype(np.int64))
t = np.array(rows)
d = np.diff(t[:, :7], axis=1)
print("k_tree, %d slots, hash oracle: cycles between stamps of block 0 / wavefront 0, median over %d waves" % (a.slots, a.waves))
for name, col in zip(("phase A (expand + backup)", "fence + slot state + root record", "descent", "leaf stores", "block atomics + barrier", "eval slot stores"), range(6)):
    print("  %-34s %8.0f" % (name, np.median(d[:, col])))
print("  %-34s %8.0f" % ("total", np.median(t[:, 6] - t[:, 0])))
print("  first ply: record arrived -> scores %.0f, argmax %.0f, play! + reward + path entry %.0f; rest of the descent %.0f"
      % (np.median(t[:, 7] - t[:, 2]), np.median(t[:, 8] - t[:, 7]), np.median(t[:, 9] - t[:, 8]), np.median(t[:, 3] - t[:, 9])))
