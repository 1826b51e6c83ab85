"""Soak test of the optimiser step's two-stream schedule: 400 batch updates at 5x128 / batch 1024 with the weight gradients on
their own stream, with AZHIP_TRAIN_ONE_STREAM=1, and once more -- losses and trained parameters must be bit-identical (a race
between the streams would show as a difference somewhere in 1200 steps).  Run on an MI355X: python tools/train_soak.py"""
import os, sys, numpy as np
sys.path.insert(0, "/root/repo/alphazero.jl_amd")
import azhip
gspec = azhip.ConnectFourSpec()
with azhip.Engine(game=0, oracle=azhip.ORACLE_HASH, num_workers=1024, batch_size=1024, num_iters_per_turn=8, reset_every=1, dirichlet_noise_eps=0.25, cpuct=1.0, temperature=([0], [1.0])) as e:
    games, moves, ng, nm, stats = e.selfplay_run(2048)
mem = azhip.MemoryBuffer(gspec, 4 * nm)
mem.push_records(games, moves, ng, nm, 1.0)
hp = azhip.ResNetHP(num_blocks=5, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
nn = azhip.ResNet(gspec, hp, seed=1)
lp = azhip.LearningParams(samples_weighing_policy=azhip.LOG_WEIGHT, l2_regularization=1e-4, loss_computation_batch_size=1024, batch_size=1024, optimiser=azhip.Adam(lr=2e-3))
out = []
for mode in ("0", "1", "0"):
    os.environ["AZHIP_TRAIN_ONE_STREAM"] = mode
    with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
        ls = tr.batch_updates(400)
        out.append((np.array(ls), tr.trained_params().copy()))
print("losses equal:", np.array_equal(out[0][0], out[1][0]), np.array_equal(out[0][0], out[2][0]), "params equal:", np.array_equal(out[0][1], out[1][1]), np.array_equal(out[0][1], out[2][1]), "last loss", out[0][0][-1])
