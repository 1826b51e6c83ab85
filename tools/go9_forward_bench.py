"""BASELINE configs[4] at the network seam: az_net_forward on 9x9x4 planes with 82 actions, ResNet 10x128, fp32 vs bf16
(host buffers in and out, i.e. PCIe included -- this is the host-stepped path of games without a device twin)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))
import azhip  # noqa: E402
from azhip.network import ResNetHP, random_params  # noqa: E402

rng = np.random.default_rng(1)
n = 4096
cell = rng.integers(0, 3, size=(n, 9, 9))
X = np.zeros((n, 4, 9, 9), dtype=np.float32)
for c in range(3):
    X[:, c] = (cell == c)
A = np.ones((n, 82), dtype=np.float32)
A[:, :81] = (cell == 2).reshape(n, 81)
hp = ResNetHP(num_blocks=10, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
blob = random_params(azhip.GAME_GO9_PLANES, hp, seed=1)
for bf in (0, 1):
    with azhip.Engine(game=azhip.GAME_GO9_PLANES, oracle=azhip.ORACLE_RESNET, num_workers=1, batch_size=1, num_iters_per_turn=2, num_blocks=10,
                      num_filters=128, num_policy_head_filters=32, num_value_head_filters=32, net_bf16=bf) as e:
        e.net_set_params(blob)
        e.net_forward(X, A)
        t0 = time.perf_counter()
        for _ in range(5):
            e.net_forward(X, A)
        dt = (time.perf_counter() - t0) / 5
        flop = 2 * 81 * 128 * (36 + 20 * 1152 + 64) * n
        print("9x9x4 planes, 82 actions, ResNet 10x128 %s: az_net_forward(%d boards) %.2f ms = %.2f M boards/s, %.0f TFLOP/s  [%s]"
              % ("bf16" if bf else "fp32", n, 1e3 * dt, n / dt / 1e6, flop / dt / 1e12, e.net_last_kernel()))
