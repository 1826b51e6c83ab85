"""BASELINE configs[4] at the network seam: 9 x 9 boards with 4 planes and 82 actions (OpenSpiel Go through
src/openspiel.jl), ResNet 10x128, fp32 and bf16.  Such games have no device twin -- their rules and tree stay on the
host (tests/test_host_fallback_gpu.py shows that path) -- so the engine offers the tensor geometry only
(AZ_GAME_GO9_PLANES): az_net_set_params / az_net_forward = Network.forward_normalized (src/networks/network.jl:264-271).
The fp32 tower is bit-identical to the oracle's fp32 chain for every tower kernel (11-tile, 6-tile = one board per
workgroup, 21-tile, and the 32x32 design); the bf16 tower holds the bf16 tolerance; search entry points refuse the game."""
import numpy as np
import pytest

import azref as R
from azhip.network import ResNetHP, random_params

pytestmark = pytest.mark.gpu


def random_go_batch(n, seed):
    rng = np.random.default_rng(seed)
    X = np.zeros((n, 4, 9, 9), dtype=np.float32)
    cell = rng.integers(0, 3, size=(n, 9, 9))
    for c in range(3):
        X[:, c] = (cell == c)
    X[:, 3] = rng.integers(0, 2, size=(n, 1, 1))                    # the to-play plane is constant over the board
    A = np.zeros((n, 82), dtype=np.float32)
    A[:, :81] = (cell == 2).reshape(n, 81) & (rng.random((n, 81)) < 0.9)   # empty points, a few of them illegal (ko / suicide)
    A[:, 81] = 1.0                                                  # pass is always legal
    return X, A


@pytest.mark.parametrize("nblocks,F,n,tower,heads", [(2, 64, 23, "16", ""), (2, 64, 5, "3", ""), (2, 64, 40, "21", "32"), (1, 64, 9, "32", ""),
                                                     (2, 128, 17, "16", "32"), (10, 128, 6, "", ""), (1, 128, 21, "", "16"), (2, 128, 9, "2", "")])
def test_go9_planes_fp32_bit_exact(nblocks, F, n, tower, heads, monkeypatch):
    """`heads`: "" = the engine's choice (k_heads16 at these sizes: 6 policy tiles of 16 logits), "32" = k_heads_mfma"""
    import azhip
    if tower:
        monkeypatch.setenv("AZHIP_TOWER", tower)
    if heads:
        monkeypatch.setenv("AZHIP_HEADS", heads)
    hp = ResNetHP(num_blocks=nblocks, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_GO9_PLANES, hp, seed=19)
    assert blob.size == R.net_num_params(R.GO9, nblocks, F, 32, 32)
    X, A = random_go_batch(n, 3)
    with azhip.Engine(game=azhip.GAME_GO9_PLANES, oracle=azhip.ORACLE_RESNET, num_workers=1, batch_size=1, num_iters_per_turn=2,
                      num_blocks=nblocks, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32) as e:
        assert e.num_actions == 82 and e.state_dim == (9, 9, 4)
        e.net_set_params(blob)
        P, V, Pinv = e.net_forward(X, A)
        kernel = e.net_last_kernel()
        with pytest.raises(azhip.AzError):
            e.selfplay_run(1)                                       # no device twin: the search refuses
        with pytest.raises(azhip.AzError):
            e.net_evaluate_keys(np.zeros((1, 2), dtype=np.uint64))
    assert "Go9Planes" in kernel and (not tower or ("NT=6" in kernel) == (tower == "3"))   # one 9x9 board needs 6 row tiles
    assert tower != "2" or kernel.startswith("k_tower16s<")
    Pr, Vr, Pir = R.net_forward_normalized(R.GO9, (nblocks, F, 32, 32), blob, X, A)
    assert np.array_equal(P, Pr), np.abs(P - Pr).max()
    assert np.array_equal(V, Vr) and np.array_equal(Pinv, Pir)
    assert np.all(P[A == 0] == 0) and np.allclose(P.sum(1), 1, atol=1e-5)


def test_go9_planes_bf16_10x128():
    import azhip
    hp = ResNetHP(num_blocks=10, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_GO9_PLANES, hp, seed=23)
    X, A = random_go_batch(64, 5)
    with azhip.Engine(game=azhip.GAME_GO9_PLANES, oracle=azhip.ORACLE_RESNET, num_workers=1, batch_size=1, num_iters_per_turn=2,
                      num_blocks=10, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32, net_bf16=1) as e:
        e.net_set_params(blob)
        P, V, Pinv = e.net_forward(X, A)
        assert e.net_last_kernel().startswith("k_tower16b<Go9Planes,128")
    Pr, Vr, _ = R.net_forward_normalized(R.GO9, (10, 128, 32, 32), blob, X, A)
    assert np.abs(P - Pr).max() < 4e-2 and np.abs(V - Vr).max() < 8e-2, (np.abs(P - Pr).max(), np.abs(V - Vr).max())
    assert np.all(P[A == 0] == 0) and np.allclose(P.sum(1), 1, atol=1e-5)
