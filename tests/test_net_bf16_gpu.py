"""The bfloat16 tower (az_engine_cfg.net_bf16, csrc/resnet16b.h; BASELINE configs[4] "ResNet 10x128 bf16").

Not the fp32 contract: weights of the F -> F convolutions and every activation the tower stores are rounded to bf16,
products accumulate in fp32 inside v_mfma_f32_16x16x32_bf16, folded batch norm / residual / ReLU run in fp32.  Two legs:
  (a) against a torch restatement of EXACTLY that scheme (bf16-rounded weights and stored activations, float64
      accumulation): 3e-3 on P and V -- what is left is the accumulation order and the rare activation that sits on a bf16
      rounding boundary;
  (b) against the fp32 network (the CPU oracle): the documented bf16 tolerance, 4e-2 on P and 8e-2 on V for 10-block
      random networks (bf16 keeps 8 significant bits; errors grow with depth).
Self-play with a bf16 network is deterministic and well-formed; it is not comparable move for move with the fp32 oracle."""
import numpy as np
import pytest
import torch

import azref as R
from azhip.network import ResNetHP, random_params, split_params
from test_net import batch_of, random_positions

pytestmark = pytest.mark.gpu


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float64)


def torch_forward_bf16(game, hp, blob, X, A):
    p = {k: torch.tensor(np.ascontiguousarray(v), dtype=torch.float64) for k, v in split_params(game, hp, blob).items()}

    def conv(x, W, b, pad, round_w):
        w = W.flip(0, 1).permute(3, 2, 1, 0).contiguous()
        return torch.nn.functional.conv2d(x, bf16(w) if round_w else w, None, padding=pad), b

    def bn(xb, pre):
        x, b = xb
        g, be, mu, var = (p[pre + "." + k].to(torch.float32) for k in ("gamma", "beta", "mean", "var"))
        scale = g / torch.sqrt(var + torch.tensor(1e-5, dtype=torch.float32))            # the kernels fold in fp32
        shift = (b.to(torch.float32) - mu) * scale + be
        s = (1, -1, 1, 1)
        return x * scale.to(torch.float64).view(s) + shift.to(torch.float64).view(s)

    x = torch.tensor(X, dtype=torch.float64)
    x = bf16(torch.relu(bn(conv(x, p["stem.conv.W"], p["stem.conv.b"], 1, False), "stem.bn")))      # the stem stays fp32, its output is stored in bf16
    for b in range(hp.num_blocks):
        y = bf16(torch.relu(bn(conv(x, p["block%d.conv1.W" % b], p["block%d.conv1.b" % b], 1, True), "block%d.bn1" % b)))
        y = bn(conv(y, p["block%d.conv2.W" % b], p["block%d.conv2.b" % b], 1, True), "block%d.bn2" % b)
        x = bf16(torch.relu(y + x))
    N = x.shape[0]
    hp_ = torch.relu(bn(conv(x, p["phead.conv.W"], p["phead.conv.b"], 0, True), "phead.bn")).to(torch.float32).to(torch.float64).reshape(N, -1)
    logits = hp_ @ p["phead.dense.W"].T + p["phead.dense.b"]
    pol = torch.softmax(logits, dim=1)
    hv = torch.relu(bn(conv(x, p["vhead.conv.W"], p["vhead.conv.b"], 0, True), "vhead.bn")).to(torch.float32).to(torch.float64).reshape(N, -1)
    v1 = torch.relu(hv @ p["vhead.dense1.W"].T + p["vhead.dense1.b"])
    val = torch.tanh(v1 @ p["vhead.dense2.W"].T + p["vhead.dense2.b"]).reshape(N)
    A = torch.tensor(A, dtype=torch.float64)
    pm = pol * A
    sp = pm.sum(dim=1, keepdim=True)
    return (pm / (sp + float(np.finfo(np.float32).eps))).numpy(), val.numpy()


@pytest.mark.parametrize("game,nblocks,F,n,tower", [(R.C4, 10, 128, 40, "16"), (R.C4, 10, 128, 9, "3"), (R.C4, 5, 64, 37, ""),
                                                     (R.TTT, 2, 64, 50, "16"), (R.MANCALA, 3, 128, 30, ""),
                                                     (R.C4, 10, 128, 43, "22"), (R.MANCALA, 3, 128, 57, "22"), (R.TTT, 2, 128, 80, "22")])   # 22 row tiles: 8 Connect-Four boards per workgroup
def test_bf16_tower_matches_its_own_scheme_and_tracks_fp32(game, nblocks, F, n, tower, monkeypatch):
    import azhip
    if tower:
        monkeypatch.setenv("AZHIP_TOWER", tower)
    hp = ResNetHP(num_blocks=nblocks, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(game, hp, seed=77)
    envs = random_positions(game, n, 6)
    X, A = batch_of(game, envs)
    with azhip.Engine(game=game, oracle=azhip.ORACLE_RESNET, num_workers=8, batch_size=8, num_iters_per_turn=8, num_blocks=nblocks,
                      num_filters=F, num_policy_head_filters=32, num_value_head_filters=32, net_bf16=1) as e:
        e.net_set_params(blob)
        P, V, Pinv = e.net_forward(X, A)
        Pk, Vk = e.net_evaluate_keys(np.array([g.key() for g in envs], dtype=np.uint64))
        assert e.net_last_kernel().startswith("k_tower16b")
    assert np.array_equal(P, Pk) and np.array_equal(V, Vk)           # planes path == fused encode path
    Pe, Ve = torch_forward_bf16(game, hp, blob, X, A)
    Pr, Vr, _ = R.net_forward_normalized(game, (nblocks, F, 32, 32), blob, X, A)
    print("bf16 %d x %d game %d: vs scheme dP %.2e dV %.2e | vs fp32 dP %.2e dV %.2e" % (nblocks, F, game, np.abs(P - Pe).max(), np.abs(V - Ve).max(), np.abs(P - Pr).max(), np.abs(V - Vr).max()))
    tol = 3e-3 if nblocks <= 3 else 1.5e-2                            # the gap to the emulation grows with depth (MFMA's internal fp32 summation vs float64)
    assert np.abs(P - Pe).max() < tol and np.abs(V - Ve).max() < tol, (np.abs(P - Pe).max(), np.abs(V - Ve).max())
    assert np.abs(P - Pr).max() < 4e-2 and np.abs(V - Vr).max() < 8e-2, (np.abs(P - Pr).max(), np.abs(V - Vr).max())
    assert np.all(P[A == 0] == 0) and np.allclose(P.sum(1), 1, atol=1e-5)


def test_bf16_self_play_is_deterministic_and_well_formed():
    import azhip
    hp = ResNetHP(num_blocks=10, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=5)
    runs = []
    for _ in range(2):
        with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=16, batch_size=8, num_iters_per_turn=32, cpuct=2.0,
                          dirichlet_noise_eps=0.25, reset_every=1, seed=9, num_blocks=10, num_filters=128, num_policy_head_filters=32,
                          num_value_head_filters=32, net_bf16=1) as e:
            e.net_set_params(blob)
            g, m, ng, nm, st = e.selfplay_run(20)
            runs.append([(g[i].game_id, g[i].num_moves, [(tuple(m[g[i].first_move + k].key), list(m[g[i].first_move + k].N), m[g[i].first_move + k].action)
                                                         for k in range(g[i].num_moves)]) for i in range(ng)])
            assert st.simulations == 32 * nm and all(sum(m[g[i].first_move].N) == 31 for i in range(ng))
    assert runs[0] == runs[1]


def test_full_launch_picks_the_22_tile_form_and_equals_the_11_tile_form(monkeypatch):
    """4096 boards at 128 filters: the engine's own choice is k_tower16b<..., NT=22> (8 boards per workgroup).  Tiling does not
    enter the arithmetic -- every output element is the same sequence of MFMAs over (tap, k step) -- so its outputs must equal
    the 11-tile form's bit for bit (which the cases above hold to the emulation and to the fp32 oracle)."""
    import azhip
    hp = ResNetHP(num_blocks=3, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(R.C4, hp, seed=78)
    envs = random_positions(R.C4, 4096, 6)
    X, A = batch_of(R.C4, envs)
    out = {}
    for tower in ("", "16"):
        if tower:
            monkeypatch.setenv("AZHIP_TOWER", tower)
        with azhip.Engine(game=R.C4, oracle=azhip.ORACLE_RESNET, num_workers=4096, batch_size=4096, num_iters_per_turn=8, num_blocks=3,
                          num_filters=128, num_policy_head_filters=32, num_value_head_filters=32, net_bf16=1) as e:
            e.net_set_params(blob)
            out[tower] = e.net_forward(X, A)[:2] + (e.net_last_kernel(),)
    assert "NT=22" in out[""][2] and "NT=11" in out["16"][2], (out[""][2], out["16"][2])
    assert np.array_equal(out[""][0], out["16"][0]) and np.array_equal(out[""][1], out["16"][1])
