"""Time the optimiser step (batch_updates!) at the reference's connect-four learning parameters
(games/connect-four/params.jl:46-58: batch 1024, Adam 2e-3, L2 1e-4, ResNet 5x128)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=8192)
ap.add_argument("--filters", type=int, default=128)
ap.add_argument("--blocks", type=int, default=5)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
gspec = azhip.ConnectFourSpec()
with azhip.Engine(game=0, oracle=azhip.ORACLE_HASH, num_workers=4096, batch_size=4096, num_iters_per_turn=8, reset_every=1,
                  dirichlet_noise_eps=0.25, cpuct=1.0, temperature=([0], [1.0])) as e:
    games, moves, ng, nm, stats = e.selfplay_run(a.games)
mem = azhip.MemoryBuffer(gspec, 4 * nm)
mem.push_records(games, moves, ng, nm, 1.0)
hp = azhip.ResNetHP(num_blocks=a.blocks, num_filters=a.filters, num_policy_head_filters=32, num_value_head_filters=32)
nn = azhip.ResNet(gspec, hp, seed=1)
lp = azhip.LearningParams(samples_weighing_policy=azhip.LOG_WEIGHT, l2_regularization=1e-4, loss_computation_batch_size=1024,
                          batch_size=a.batch, optimiser=azhip.Adam(lr=2e-3))
with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
    st0 = tr.learning_status()
    tr.batch_updates(3)
    t0 = time.perf_counter()
    ls = tr.batch_updates(a.steps)
    dt = time.perf_counter() - t0
    flop = 3 * 2 * a.batch * 42 * (9 * 3 * a.filters + 2 * a.blocks * 9 * a.filters * a.filters + a.filters * 64)   # fwd + dgrad + wgrad of the convolutions
    print("batch_updates!: %d samples, batch %d, 5x%d: %.2f ms / step (%.1f TFLOP/s of convolution work), loss %.4f -> %.4f"
          % (tr.num_samples(), a.batch, a.filters, 1e3 * dt / a.steps, flop / (dt / a.steps) / 1e12, ls[0], ls[-1]))
    nn2 = azhip.ResNet(gspec, hp, params=tr.trained_params())
with azhip.Trainer(gspec, nn2, mem, lp, use_symmetries=True) as tr2:
    st1 = tr2.learning_status()
print("learning status before: L %.4f Lp %.4f Lv %.4f | after %d steps: L %.4f Lp %.4f Lv %.4f"
      % (st0.loss.L, st0.loss.Lp, st0.loss.Lv, a.steps + 3, st1.loss.L, st1.loss.Lp, st1.loss.Lv))
