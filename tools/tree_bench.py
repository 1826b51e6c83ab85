"""The tree kernel alone (NN-free hash oracle) at increasing slot counts: HIP-event time of k_tree and the
algorithmic HBM bytes of SURVEY.md §8(d) (148 B per traversed node + per-leaf terms without the network part)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402

GS = [int(x) for x in sys.argv[1:]] or [4096, 65536, 262144, 1048576]   # AZHIP_EVAL_CACHE=1 (+ AZHIP_EVAL_CACHE_LOG2): with the evaluation cache's probe / fill in k_tree
for G in GS:
    nsims, waves = 200, 400
    e = azhip.Engine(game=0, oracle=azhip.ORACLE_HASH, num_workers=G, batch_size=G, num_iters_per_turn=nsims, cpuct=2.0,
                     dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, reset_every=1, max_nodes_per_slot=nsims * 3)
    e.selfplay_begin(-1, 0)
    e.selfplay_step(int(os.environ.get("TREE_BENCH_WARM", "100")))   # AZHIP_TREE_SORT orders the slots at move steps (every nsims waves): TREE_BENCH_WARM=200 times sorted waves only
    s0 = e.selfplay_stats()
    e.prof_reset(); e.prof_enable(True)
    e.selfplay_step(waves)
    s1 = e.selfplay_stats()
    p = e.prof_get()
    e.selfplay_end()
    sims = s1.simulations - s0.simulations
    trav = s1.nodes_traversed - s0.nodes_traversed
    evals = s1.leaf_evals - s0.leaf_evals
    d = trav / sims
    tree_bytes = 148 * trav + (16 + 136 + 64) * evals + 16 * (sims - evals)   # bench.py's roofline_tree figure
    t_ms = p["select"]["ms"] + p["expand"]["ms"]
    print("G=%7d depth %.2f | k_tree %.1f us/wave %.0f GB/s (%.1f %% of 8 TB/s) | synth %.1f us | %.1f M sims/s tree-only"
          % (G, d, 1e3 * t_ms / waves, tree_bytes / (t_ms * 1e-3) / 1e9, 100 * tree_bytes / (t_ms * 1e-3) / 8e12,
             1e3 * p["synth"]["ms"] / waves, sims / (sum(v["ms"] for v in p.values()) * 1e-3) / 1e6)
          + (" | cache: %.1f %% of leaf evaluations reused" % (100.0 * (s1.evals_reused - s0.evals_reused) / max(evals, 1)) if os.environ.get("AZHIP_EVAL_CACHE") == "1" else ""))
    e.close()
