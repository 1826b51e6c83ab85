"""Network parity.  Three legs:
  * the oracle's fp32 forward vs an independent PyTorch fp64 restatement of the Flux model
    (src/networks/architectures/resnet.jl:53-92, network.jl:264-271), tolerance 1e-5  [CPU]
  * the HIP tower/heads kernels vs the oracle: bit-exact fp32 (the fp32 contract of include/azhip.h),
    and within 1e-5 of the fp64 restatement  [GPU]
"""
import numpy as np
import pytest
import torch

import azref as R
from azhip.network import ResNetHP, num_parameters, random_params, split_params

TOL = 1e-5   # BASELINE.json: "within 1e-5 on fp32 policy/value outputs"


def torch_forward_normalized(game, hp, blob, X, A):
    """Independent fp64 restatement with torch ops (conv2d = cross-correlation, so kernels are flipped)."""
    p = {k: torch.tensor(np.ascontiguousarray(v), dtype=torch.float64) for k, v in split_params(game, hp, blob).items()}
    x = torch.tensor(X, dtype=torch.float64)

    def conv(x, W, b, pad):
        w = W.flip(0, 1).permute(3, 2, 1, 0).contiguous()      # (co, ci, ky, kx), true convolution
        return torch.nn.functional.conv2d(x, w, b, padding=pad)

    def bn(x, pre):
        g, be, mu, var = p[pre + ".gamma"], p[pre + ".beta"], p[pre + ".mean"], p[pre + ".var"]
        s = (1, -1, 1, 1)
        return g.view(s) * (x - mu.view(s)) / torch.sqrt(var.view(s) + 1e-5) + be.view(s)

    x = torch.relu(bn(conv(x, p["stem.conv.W"], p["stem.conv.b"], 1), "stem.bn"))
    for b in range(hp.num_blocks):
        y = torch.relu(bn(conv(x, p["block%d.conv1.W" % b], p["block%d.conv1.b" % b], 1), "block%d.bn1" % b))
        y = bn(conv(y, p["block%d.conv2.W" % b], p["block%d.conv2.b" % b], 1), "block%d.bn2" % b)
        x = torch.relu(y + x)
    N = x.shape[0]
    hp_ = torch.relu(bn(conv(x, p["phead.conv.W"], p["phead.conv.b"], 0), "phead.bn")).reshape(N, -1)
    logits = hp_ @ p["phead.dense.W"].T + p["phead.dense.b"]
    pol = torch.softmax(logits, dim=1)
    hv = torch.relu(bn(conv(x, p["vhead.conv.W"], p["vhead.conv.b"], 0), "vhead.bn")).reshape(N, -1)
    v1 = torch.relu(hv @ p["vhead.dense1.W"].T + p["vhead.dense1.b"])
    val = torch.tanh(v1 @ p["vhead.dense2.W"].T + p["vhead.dense2.b"]).reshape(N)
    A = torch.tensor(A, dtype=torch.float64)
    pm = pol * A
    sp = pm.sum(dim=1, keepdim=True)
    eps32 = float(np.finfo(np.float32).eps)
    return (pm / (sp + eps32)).numpy(), val.numpy(), (1 - sp).reshape(N).numpy()


def random_positions(game, n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        g = R.Game(game)
        for _ in range(int(rng.integers(0, 30))):
            if g.terminated():
                break
            g.play(int(rng.choice(g.available_actions())))
        if not g.terminated():
            out.append(g)
    return out


def batch_of(game, envs):
    w, h, c = R.DIMS[game]
    X = np.stack([g.vectorize().reshape(c, h, w) for g in envs])
    A = np.stack([g.actions_mask().astype(np.float32) for g in envs])
    return X, A


@pytest.mark.parametrize("game,hp", [
    (R.C4, ResNetHP(num_blocks=2, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)),
    (R.TTT, ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)),
    (R.MANCALA, ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)),
    (R.C4, ResNetHP(num_blocks=1, num_filters=16, num_policy_head_filters=2, num_value_head_filters=1)),
])
def test_oracle_forward_vs_torch_fp64(game, hp):
    blob = random_params(game, hp, seed=5)
    assert blob.size == num_parameters(game, hp) == R.net_num_params(game, hp.num_blocks, hp.num_filters, hp.num_policy_head_filters, hp.num_value_head_filters)
    X, A = batch_of(game, random_positions(game, 6, 1))
    P, V, Pinv = R.net_forward_normalized(game, (hp.num_blocks, hp.num_filters, hp.num_policy_head_filters, hp.num_value_head_filters), blob, X, A)
    Pt, Vt, Pit = torch_forward_normalized(game, hp, blob, X, A)
    assert np.abs(P - Pt).max() < TOL and np.abs(V - Vt).max() < TOL and np.abs(Pinv - Pit).max() < TOL
    assert np.all(P[A == 0] == 0) and np.allclose(P.sum(1), 1, atol=1e-5)


def test_param_count_matches_reference_doc():
    """5x64 with 32/32 heads = 472 328 trainable parameters, 5x128 = 1 672 328 ("about 1.6M",
    docs/src/tutorial/connect_four.md:60-61)."""
    from azhip.game import ConnectFourSpec
    from azhip.network import ResNet
    for F, n in ((64, 472328), (128, 1672328)):
        nn = ResNet(ConnectFourSpec(), ResNetHP(num_blocks=5, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32))
        assert nn.num_parameters() == n


@pytest.mark.gpu
@pytest.mark.parametrize("game,nblocks,n,F,heads", [(R.C4, 5, 64, 64, (32, 32)), (R.C4, 1, 7, 64, (32, 32)), (R.TTT, 2, 33, 64, (32, 32)),
                                                      (R.MANCALA, 2, 20, 64, (32, 32)), (R.C4, 2, 40, 128, (32, 32)),
                                                      (R.C4, 1, 9, 64, (2, 1)), (R.TTT, 1, 5, 128, (16, 8))])
@pytest.mark.parametrize("tower", ["16", "32", "3", "21", "2", "7", "19", "20"])      # 19 (r6): the 7-board paired form (64 filters); 20 (r6): the 8-board paired form within 176 registers
def test_hip_forward_bit_exact_vs_oracle(game, nblocks, n, F, heads, tower, monkeypatch):
    """F = 128 is the shipped connect-four network (games/connect-four/params.jl:7-13); heads (2, 1) are the
    ResNetHP defaults (resnet.jl:30-37) and take the VALU dense-head kernel.  All tower kernels (k_tower16 on
    16x16x4 MFMA with 11 or 3 row tiles per workgroup, k_tower on 32x32x2, "2" = k_tower16s: two workgroups per board
    tile exchanging channel halves after every layer, 128 filters only -- the engine's own choice elsewhere) are forced
    in turn: the engine otherwise picks one per launch size."""
    import azhip
    monkeypatch.setenv("AZHIP_TOWER", tower)
    npf, nvf = heads
    hp = ResNetHP(num_blocks=nblocks, num_filters=F, num_policy_head_filters=npf, num_value_head_filters=nvf)
    blob = random_params(game, hp, seed=2026)
    envs = random_positions(game, n, 3)
    X, A = batch_of(game, envs)
    with azhip.Engine(game=game, oracle=azhip.ORACLE_RESNET, num_workers=8, batch_size=8, num_iters_per_turn=8,
                      num_blocks=nblocks, num_filters=F, num_policy_head_filters=npf, num_value_head_filters=nvf) as e:
        e.net_set_params(blob)
        assert np.array_equal(e.net_get_params(), blob)
        P, V, Pinv = e.net_forward(X, A)
        keys = np.array([g.key() for g in envs], dtype=np.uint64)
        Pk, Vk = e.net_evaluate_keys(keys)
        Xd, Ad = e.encode(keys)
        if tower == "7":             # the exact-fit variant: 7 row tiles = 8 Mancala boards, 9 = 16 Tic-tac-toe boards; Connect-Four (r5): the half-size form, 6 tiles = 2 boards
            assert e.net_last_kernel().endswith("NT=%d>" % {R.MANCALA: 7, R.TTT: 9, R.C4: 6}[game]), e.net_last_kernel()
    assert np.array_equal(Xd, X) and np.array_equal(Ad, A)          # vectorize_state / actions_mask twins
    Pr, Vr, Pir = R.net_forward_normalized(game, (nblocks, F, npf, nvf), blob, X, A)
    Pt, Vt, Pit = torch_forward_normalized(game, hp, blob, X, A)
    # tolerance leg (BASELINE.json): 1e-5 vs the fp64 restatement
    assert np.abs(P - Pt).max() < TOL and np.abs(V - Vt).max() < TOL and np.abs(Pinv - Pit).max() < TOL
    # contract leg: bit-exact vs the oracle's fp32 chain
    assert np.array_equal(P, Pr), np.abs(P - Pr).max()
    assert np.array_equal(V, Vr), np.abs(V - Vr).max()
    assert np.array_equal(Pinv, Pir)
    assert np.array_equal(Pk, P) and np.array_equal(Vk, V)           # fused encode path == planes path


@pytest.mark.gpu
def test_engine_reports_the_tower_kernel_it_launched():
    import azhip
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    X, A = batch_of(R.C4, random_positions(R.C4, 9, 4))
    with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=8, batch_size=8, num_iters_per_turn=8,
                      num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32) as e:
        assert e.net_last_kernel() == ""
        e.net_set_params(random_params(R.C4, hp, seed=12))
        e.net_forward(X, A)
        assert e.net_last_kernel() == "k_tower16<ConnectFour,64,NT=3>"   # 9 boards: one board per workgroup


def test_oracle_matches_committed_golden():
    """tests/golden/net_golden.npz pins the fp32 contract (summation order) on CPU."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "net_golden.npz"))
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(R.C4, hp, seed=7)
    P, V, Pinv = R.net_forward_normalized(R.C4, (1, 64, 32, 32), blob, z["X"], z["A"])
    assert np.array_equal(P, z["P"]) and np.array_equal(V, z["V"]) and np.array_equal(Pinv, z["Pinv"])


@pytest.mark.gpu
def test_hip_matches_committed_golden():
    import os
    import azhip
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "net_golden.npz"))
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=4, batch_size=4, num_iters_per_turn=4,
                      num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(random_params(R.C4, hp, seed=7))
        P, V, Pinv = e.net_forward(z["X"], z["A"])
        Pk, Vk = e.net_evaluate_keys(z["keys"])
    assert np.array_equal(P, z["P"]) and np.array_equal(V, z["V"]) and np.array_equal(Pinv, z["Pinv"])
    assert np.array_equal(Pk, z["P"]) and np.array_equal(Vk, z["V"])


@pytest.mark.gpu
def test_set_params_twice_replaces_the_network():
    """Network.copy per self-play phase (training.jl:278-279) re-uploads weights: the second upload wins."""
    import azhip
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    b1, b2 = random_params(R.C4, hp, seed=1), random_params(R.C4, hp, seed=2)
    X, A = batch_of(R.C4, random_positions(R.C4, 5, 9))
    with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=4, batch_size=4, num_iters_per_turn=4,
                      num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32) as e:
        with pytest.raises(azhip.AzError):
            e.net_forward(X, A)                      # no parameters yet
        e.net_set_params(b1)
        P1, V1, _ = e.net_forward(X, A)
        e.net_set_params(b2)
        P2, V2, _ = e.net_forward(X, A)
        with pytest.raises(azhip.AzError):
            e.net_set_params(b1[:-1])                # wrong blob size
    assert np.array_equal(P1, R.net_forward_normalized(R.C4, (1, 64, 32, 32), b1, X, A)[0])
    assert np.array_equal(P2, R.net_forward_normalized(R.C4, (1, 64, 32, 32), b2, X, A)[0]) and not np.array_equal(P1, P2)


@pytest.mark.gpu
@pytest.mark.parametrize("heads_kernel", ["16", "32"])
@pytest.mark.parametrize("game,F,n", [(R.C4, 64, 37), (R.C4, 128, 50), (R.TTT, 64, 19), (R.MANCALA, 128, 33), (R.C4, 64, 1)])
def test_dense_head_kernels_bit_exact_vs_oracle(game, F, n, heads_kernel, monkeypatch):
    """The dense heads have two MFMA kernels: k_heads16 (16-board tiles on 16x16x4, small launches) and k_heads_mfma
    (32-board tiles on 32x32x2, full launches); the engine picks by launch size.  Both are forced here (AZHIP_HEADS) on
    batches with a partial last tile: same ascending-k fp32 chain, so both equal the oracle bit for bit."""
    import azhip
    monkeypatch.setenv("AZHIP_HEADS", heads_kernel)
    hp = ResNetHP(num_blocks=1, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(game, hp, seed=77)
    X, A = batch_of(game, random_positions(game, n, 8))
    with azhip.Engine(game=game, oracle=azhip.ORACLE_RESNET, num_workers=8, batch_size=8, num_iters_per_turn=8,
                      num_blocks=1, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        P, V, Pinv = e.net_forward(X, A)
    Pr, Vr, Pir = R.net_forward_normalized(game, (1, F, 32, 32), blob, X, A)
    assert np.array_equal(P, Pr), np.abs(P - Pr).max()
    assert np.array_equal(V, Vr) and np.array_equal(Pinv, Pir)


@pytest.mark.gpu
@pytest.mark.parametrize("game,nblocks,n", [(R.C4, 5, 128), (R.C4, 2, 1), (R.C4, 3, 77), (R.TTT, 2, 33), (R.MANCALA, 2, 100)])
def test_split_tower_bit_exact_vs_oracle(game, nblocks, n):
    """k_tower16s (128 filters, launches of at most num_cu workgroups): pairs of workgroups exchange halves of every
    layer's channels through HBM with agent-scope atomics.  Bit-exact like every other tower kernel; the kernel is the
    engine's own pick at these sizes; repeated launches reuse the exchange areas (epochs)."""
    import azhip
    hp = ResNetHP(num_blocks=nblocks, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(game, hp, seed=41)
    envs = random_positions(game, n, 11)
    X, A = batch_of(game, envs)
    with azhip.Engine(game=game, oracle=azhip.ORACLE_RESNET, num_workers=8, batch_size=8, num_iters_per_turn=8,
                      num_blocks=nblocks, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        outs = [e.net_forward(X, A) for _ in range(3)]
        assert e.net_last_kernel().startswith("k_tower16s<")
        Pk, Vk = e.net_evaluate_keys(np.array([g.key() for g in envs], dtype=np.uint64))
    Pr, Vr, Pir = R.net_forward_normalized(game, (nblocks, 128, 32, 32), blob, X, A)
    for P, V, Pinv in outs:
        assert np.array_equal(P, Pr), np.abs(P - Pr).max()
        assert np.array_equal(V, Vr) and np.array_equal(Pinv, Pir)
    assert np.array_equal(Pk, Pr) and np.array_equal(Vk, Vr)


@pytest.mark.gpu
def test_split_tower_epoch_wrap(monkeypatch):
    """The exchange words of k_tower16s carry 24 bits of the launch epoch; before they repeat the library clears the
    areas and skips the epoch whose tags would equal cleared memory.  Started three launches before the wrap, eight
    forward passes (of different sizes, so that some areas go unused for a while) stay bit-exact."""
    import azhip
    monkeypatch.setenv("AZHIP_XCH_EPOCH0", str(0xFFFFFF - 3))
    hp = ResNetHP(num_blocks=2, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(R.C4, hp, seed=43)
    envs = random_positions(R.C4, 60, 12)
    X, A = batch_of(R.C4, envs)
    Pr, Vr, _ = R.net_forward_normalized(R.C4, (2, 128, 32, 32), blob, X, A)
    with azhip.Engine(game=R.C4, oracle=azhip.ORACLE_RESNET, num_workers=64, batch_size=64, num_iters_per_turn=8,
                      num_blocks=2, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        for k in range(8):
            n = (60, 7, 33, 60, 2, 60, 19, 60)[k]
            P, V, _ = e.net_forward(X[:n], A[:n])
            assert e.net_last_kernel().startswith("k_tower16s<")
            assert np.array_equal(P, Pr[:n]) and np.array_equal(V, Vr[:n]), k


@pytest.mark.gpu
def test_split_tower_gives_up_instead_of_hanging():
    """The workgroups of k_tower16s wait for each other's halves with a BOUNDED poll.  az_debug_exchange_timeout launches
    it with one workgroup missing: the call must come back with an error (about a second), not hang the GPU, and the
    engine must work afterwards."""
    import ctypes as C
    import time
    import azhip
    from azhip._lib import lib
    hp = ResNetHP(num_blocks=1, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(R.C4, hp, seed=47)
    X, A = batch_of(R.C4, random_positions(R.C4, 20, 13))
    with azhip.Engine(game=R.C4, oracle=azhip.ORACLE_RESNET, num_workers=8, batch_size=8, num_iters_per_turn=8,
                      num_blocks=1, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        f = lib().az_debug_exchange_timeout
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        t0 = time.perf_counter()
        st = f(e._h)
        dt = time.perf_counter() - t0
        assert st != 0 and b"partner" in lib().az_last_error() and dt < 30.0, (st, dt)
        P, V, _ = e.net_forward(X, A)                               # the next launches are unaffected
    Pr, Vr, _ = R.net_forward_normalized(R.C4, (1, 128, 32, 32), blob, X, A)
    assert np.array_equal(P, Pr) and np.array_equal(V, Vr)
