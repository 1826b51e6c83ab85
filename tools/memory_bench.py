"""Time the device replay-memory path at scale: push_trace! of a self-play phase, the Trainer's data set
(symmetries + merge_by_state + convert_samples) and learning_status (5x64 ResNet, test mode)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=16384)
ap.add_argument("--filters", type=int, default=64)
a = ap.parse_args()
gspec = azhip.ConnectFourSpec()
with azhip.Engine(game=0, oracle=azhip.ORACLE_HASH, num_workers=4096, batch_size=4096, num_iters_per_turn=8, reset_every=1,
                  dirichlet_noise_eps=0.25, cpuct=1.0, temperature=([0], [1.0])) as e:
    t0 = time.perf_counter()
    games, moves, ng, nm, stats = e.selfplay_run(a.games)
    print("generated %d games, %d positions in %.2f s (hash oracle, 8 sims/move)" % (ng, nm, time.perf_counter() - t0))
mem = azhip.MemoryBuffer(gspec, 4 * nm)
t0 = time.perf_counter()
mem.push_records(games, moves, ng, nm, 1.0)
t1 = time.perf_counter()
print("push_trace!: %d samples in %.1f ms = %.1f M samples/s (incl. the 64 B/sample H2D copy)" % (nm, 1e3 * (t1 - t0), nm / (t1 - t0) / 1e6))
for _ in range(2):
    t0 = time.perf_counter()
    d = mem.dataset(use_symmetries=True, use_position_averaging=True, weighing_policy=azhip.LOG_WEIGHT)
    t1 = time.perf_counter()
    print("data set: %d samples -> x2 symmetries -> %d merged boards (+ W,X,A,P,V tensors) in %.1f ms = %.1f M input samples/s"
          % (nm, len(d), 1e3 * (t1 - t0), nm / (t1 - t0) / 1e6))
    d.close()
hp = azhip.ResNetHP(num_blocks=5, num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32)
nn = azhip.ResNet(gspec, hp, seed=1)
lp = azhip.LearningParams(samples_weighing_policy=azhip.LOG_WEIGHT, l2_regularization=1e-4, loss_computation_batch_size=1024)
with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
    tr.learning_status()
    t0 = time.perf_counter()
    st = tr.learning_status()
    t1 = time.perf_counter()
    print("learning_status: %d boards, 5x%d net, batches of 1024: %.1f ms = %.2f M boards/s   L=%.4f Lp=%.4f Lv=%.4f Hp=%.4f"
          % (tr.num_samples(), a.filters, 1e3 * (t1 - t0), tr.num_samples() / (t1 - t0) / 1e6, st.loss.L, st.loss.Lp, st.loss.Lv, st.Hp))
