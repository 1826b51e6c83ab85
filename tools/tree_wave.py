"""k_tree alone at ONE slot count (NN-free hash oracle): the workload of tools/tree_bench.py for profilers (rocprofv3 --pmc)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slots", type=int, default=4096)
ap.add_argument("--waves", type=int, default=200)
ap.add_argument("--sims", type=int, default=200)
a = ap.parse_args()
e = azhip.Engine(game=0, oracle=azhip.ORACLE_HASH, num_workers=a.slots, batch_size=a.slots, num_iters_per_turn=a.sims, cpuct=2.0,
                 dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, reset_every=1, max_nodes_per_slot=a.sims * 3)
e.selfplay_begin(-1, 0)
e.selfplay_step(100)
e.prof_reset(); e.prof_enable(True)
e.selfplay_step(a.waves)
p = e.prof_get()
e.selfplay_end()
print("G=%d k_tree %.2f us/wave" % (a.slots, 1e3 * (p["select"]["ms"] + p["expand"]["ms"]) / a.waves))
e.close()
