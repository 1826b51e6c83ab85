"""The optimiser step on the device (az_trainer_*; src/learning.jl:59-141, src/networks/flux.jl:68-95) against an
independent fp64 torch restatement: train-mode forward (BatchNorm with batch statistics), `losses`, autograd gradients
in Flux parameter order, Adam trajectories, running statistics.  Floating point: fp32 device vs fp64 reference, the
tolerances are relative to the largest gradient entry of each parameter array."""
import numpy as np
import pytest
import torch

import azref as R
from azhip.network import param_layout, split_params

pytestmark = pytest.mark.gpu
EPS32 = float(np.finfo(np.float32).eps)
SPECS = {0: "ConnectFourSpec", 1: "TicTacToeSpec", 2: "MancalaSpec"}


def _memory(game, ngames, seed):
    import azhip
    gspec = getattr(azhip, SPECS[game])()
    with azhip.Engine(game=game, oracle=azhip.ORACLE_HASH, num_workers=8, batch_size=8, num_iters_per_turn=16,
                      dirichlet_noise_eps=0.25, cpuct=1.0, reset_every=1, temperature=([0], [1.0]), seed=seed,
                      max_moves_per_game=200 if game == 2 else 0) as e:
        games, moves, ng, nm, _ = e.selfplay_run(ngames)
    mem = azhip.MemoryBuffer(gspec, 100000)
    mem.push_records(games, moves, ng, nm, 1.0)
    return gspec, mem


class _MaskedRelu(torch.autograd.Function):
    """relu whose DERIVATIVE is a given 0/1 mask (the forward value is relu(z) itself)"""
    @staticmethod
    def forward(ctx, z, mask):
        ctx.save_for_backward(mask)
        return torch.relu(z)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


class TorchNet:
    """fp64 restatement of the Flux ResNet in TRAIN mode + `losses` (learning.jl:67-90), parameters in Julia shapes.
    masks (optional): name -> 0/1 tensor; the ReLU of that name then differentiates with the given mask instead of (z > 0)."""

    def __init__(self, game, hp, blob, masks=None):
        self.game, self.hp = game, hp
        self.masks, self.own_masks = masks or {}, {}
        self.names = [n for n, _ in param_layout(game, hp)]
        self.p = {k: torch.tensor(np.ascontiguousarray(v), dtype=torch.float64, requires_grad=not (k.endswith(".mean") or k.endswith(".var")))
                  for k, v in split_params(game, hp, blob).items()}
        self.batch_stats = {}

    def blob(self, grads=False):
        out = []
        for n in self.names:
            t = self.p[n]
            a = (t.grad if grads else t.detach()) if (not grads or t.requires_grad) else None
            a = np.zeros(tuple(t.shape)) if a is None else a.numpy()
            out.append(np.asarray(a).reshape(-1, order="F"))
        return np.concatenate(out)

    def _conv(self, x, pre, pad):
        W, b = self.p[pre + ".W"], self.p[pre + ".b"]
        return torch.nn.functional.conv2d(x, W.flip(0, 1).permute(3, 2, 1, 0).contiguous(), b, padding=pad)

    def _bn(self, x, pre):
        g, be = self.p[pre + ".gamma"], self.p[pre + ".beta"]
        mu = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        self.batch_stats[pre] = (mu.detach(), var.detach(), x.numel() // x.shape[1])
        s = (1, -1, 1, 1)
        return g.view(s) * (x - mu.view(s)) / torch.sqrt(var.view(s) + 1e-5) + be.view(s)

    def _relu(self, z, name):
        self.own_masks[name] = (z.detach() > 0).to(torch.float64)
        if name in self.masks:
            return _MaskedRelu.apply(z, self.masks[name])
        return torch.relu(z)

    def forward(self, X):
        x = self._relu(self._bn(self._conv(X, "stem.conv", 1), "stem.bn"), "tower0")
        for b in range(self.hp.num_blocks):
            y = self._relu(self._bn(self._conv(x, "block%d.conv1" % b, 1), "block%d.bn1" % b), "tower%d" % (2 * b + 1))
            y = self._bn(self._conv(y, "block%d.conv2" % b, 1), "block%d.bn2" % b)
            x = self._relu(y + x, "tower%d" % (2 * b + 2))
        N = x.shape[0]
        hp_ = self._relu(self._bn(self._conv(x, "phead.conv", 0), "phead.bn"), "phead").reshape(N, -1)
        logits = hp_ @ self.p["phead.dense.W"].T + self.p["phead.dense.b"]
        hv = self._relu(self._bn(self._conv(x, "vhead.conv", 0), "vhead.bn"), "vhead").reshape(N, -1)
        v1 = self._relu(hv @ self.p["vhead.dense1.W"].T + self.p["vhead.dense1.b"], "v1")
        val = torch.tanh(v1 @ self.p["vhead.dense2.W"].T + self.p["vhead.dense2.b"]).reshape(N)
        return torch.softmax(logits, dim=1), val

    def losses(self, batch, Wmean, Hp, l2, cinv, rho):
        W, X, A, P, V = [torch.tensor(np.asarray(x), dtype=torch.float64) for x in batch]
        pol, val = self.forward(X)
        pm = pol * A
        sp = pm.sum(dim=1, keepdim=True)
        Ph = pm / (sp + EPS32)
        pinv = (1 - sp).reshape(-1)
        Lp = -(P * torch.log(Ph + EPS32) * W[:, None]).sum() / W.sum() - Hp
        Lv = (((val / rho - V / rho) ** 2) * W).sum() / W.sum()
        Lreg = l2 * sum((t ** 2).sum() for t in self.p.values() if t.requires_grad)
        Linv = cinv * (pinv * W).sum() / W.sum()
        scale = W.mean() / Wmean
        return scale * (Lp + Lv + Lreg + Linv), (Lp, Lv, Lreg, Linv, scale)


def _device_relu_masks(tr, hp, ref):
    """the ReLU masks (a > 0) of the device's last forward pass, shaped like the reference's (az_debug_trainer_activation, train.hip).
    The device keeps activations as [board x position][channel]; the position order is found by agreement with the reference."""
    import ctypes as C
    from azhip import _lib as L
    f = L.lib().az_debug_trainer_activation
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    ntower = 1 + 2 * hp.num_blocks
    names = ["tower%d" % l for l in range(ntower)] + ["phead", "vhead", "v1"]
    out, flips = {}, 0
    for which, name in enumerate(names):
        own = ref.own_masks[name]
        a = np.zeros(own.numel(), dtype=np.float32)
        L.check(f(tr._trainer(), which, a.ctypes.data_as(C.c_void_p), a.size))
        if own.dim() == 4:
            n, c, h, w = own.shape
            cands = [torch.tensor(a.reshape(n, h, w, c) > 0).permute(0, 3, 1, 2), torch.tensor(a.reshape(n, w, h, c) > 0).permute(0, 3, 2, 1)]
            m = max(cands, key=lambda t: int((t.to(torch.float64) == own).sum()))
        else:
            m = torch.tensor(a.reshape(tuple(own.shape)) > 0)
        m = m.to(torch.float64).contiguous()
        d = int((m != own).sum())
        assert d <= max(8, own.numel() // 20000), (name, d, own.numel())   # the same network: only units within rounding of zero differ
        flips += d
        out[name] = m
    return out, flips


def _rel_err_by_array(game, hp, got, want, tol=1e-3, l2_tol=None):
    """every array: max |got - want| <= tol * max |want| (+ 2e-6); l2_tol: also ||got - want||_2 <= l2_tol * ||want||_2.
    Deep towers: with ~0.5 M activations per layer a handful of them sit within rounding of zero, an fp32 chain and an fp64 chain
    then disagree on their ReLU masks, and ONE flipped unit moves single weight-gradient entries by a few 1e-3 of the array's
    largest (seen: 4.4e-3 on block4.conv1.W of a 5x128 tower with 96 samples).  Round 3 loosened the tolerance to 1e-2 there; round 4
    gives the fp64 reference the DEVICE's masks instead (_device_relu_masks) and keeps 1e-3."""
    worst, off = 0.0, 0
    for name, shape in param_layout(game, hp):
        n = int(np.prod(shape))
        g, w = got[off:off + n], want[off:off + n]
        off += n
        if name.endswith(".mean") or name.endswith(".var"):
            assert not g.any()
            continue
        denom = max(np.abs(w).max(), 1e-7)
        worst = max(worst, np.abs(g - w).max() / denom) if np.abs(w).max() > 1e-6 else worst
        assert np.abs(g - w).max() <= tol * denom + 2e-6, (name, np.abs(g - w).max(), denom)
        if l2_tol is not None:
            assert np.linalg.norm(g - w) <= l2_tol * np.linalg.norm(w) + 1e-7, (name, np.linalg.norm(g - w), np.linalg.norm(w))
    return worst


def test_gradients_with_default_heads_and_no_blocks():
    """ResNetHP defaults (2 policy / 1 value head filters, resnet.jl:30-37) and a tower without residual blocks"""
    import azhip
    for nblocks, heads in ((0, (32, 32)), (1, (2, 1))):
        gspec, mem = _memory(1, 12, 5)
        hp = azhip.ResNetHP(num_blocks=nblocks, num_filters=64, num_policy_head_filters=heads[0], num_value_head_filters=heads[1])
        nn = azhip.ResNet(gspec, hp, seed=6)
        lp = azhip.LearningParams(samples_weighing_policy=1, l2_regularization=1e-4, loss_computation_batch_size=64, batch_size=20)
        with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
            data = tr.data.tensors()
            idx = np.arange(20) * 3
            loss, parts, grad = tr.gradients(idx)
            ref = TorchNet(1, hp, nn.params())
            L, (Lp, Lv, Lreg, Linv, scale) = ref.losses([x[idx] for x in data], float(tr.Wmean), float(tr.Hp), 1e-4, 1.0, 1.0)
            (L - scale * Lreg).backward()
            assert abs(loss - L.item()) < 2e-5 * max(1.0, abs(L.item()))
            _rel_err_by_array(1, hp, grad.astype(np.float64), ref.blob(grads=True))
            assert np.isfinite(tr.batch_updates(3)).all()
        mem.close()


@pytest.mark.parametrize("game,nblocks,F,B,policy", [(1, 1, 64, 24, 1), (0, 2, 64, 16, 0), (2, 1, 64, 20, 2), (0, 1, 128, 12, 1),
                                                     (0, 1, 128, 203, 1), (0, 2, 64, 333, 0), (1, 1, 64, 500, 2),    # many workgroups, ragged tails
                                                     (0, 5, 128, 96, 1), (2, 3, 64, 130, 0),    # deep towers: the three-buffer gradient ring of the backward pass wraps
                                                     (1, 1, 128, 70, 1), (2, 2, 128, 45, 0)])   # 128 filters on the small boards: k_wgrad16's 4-wavefront form with 5 / 3 boards per LDS chunk
def test_gradients_match_torch_autograd(game, nblocks, F, B, policy):
    import azhip
    gspec, mem = _memory(game, 12 if B < 100 else 60, 3)
    hp = azhip.ResNetHP(num_blocks=nblocks, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=8)
    lp = azhip.LearningParams(samples_weighing_policy=policy, l2_regularization=1e-4, loss_computation_batch_size=64, batch_size=B,
                              rewards_renormalization=2.0, nonvalidity_penalty=1.0)
    with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=game != 2) as tr:
        data = tr.data.tensors()
        n = len(data[0])
        rng = np.random.default_rng(5)
        idx = rng.choice(n, size=B, replace=False)
        loss, parts, grad = tr.gradients(idx)
        batch = [x[idx] for x in data]
        ref = TorchNet(game, hp, nn.params())
        L, (Lp, Lv, Lreg, Linv, scale) = ref.losses(batch, float(tr.Wmean), float(tr.Hp), 1e-4, 1.0, 2.0)
        (L - scale * Lreg).backward()                              # the device gradient excludes the L2 term (added in the update)
        want = ref.blob(grads=True)
        assert abs(loss - L.item()) < 2e-5 * max(1.0, abs(L.item()))
        assert np.allclose(parts, [Lp.item(), Lv.item(), Lreg.item(), Linv.item(), scale.item()], rtol=5e-5, atol=5e-6), (parts, Lp.item(), Lv.item())
        if nblocks >= 3:
            # the same reference differentiated with the ReLU masks the device actually had: the comparison is then between
            # two chains of the SAME piecewise-linear function, and the 1e-3 of the shallow cases holds for every entry
            masks, flips = _device_relu_masks(tr, hp, ref)
            ref2 = TorchNet(game, hp, nn.params(), masks=masks)
            L2, (_, _, Lreg2, _, scale2) = ref2.losses(batch, float(tr.Wmean), float(tr.Hp), 1e-4, 1.0, 2.0)
            (L2 - scale2 * Lreg2).backward()
            assert abs(L2.item() - L.item()) < 1e-12 * max(1.0, abs(L.item()))      # the masks change derivatives, not values
            _rel_err_by_array(game, hp, grad.astype(np.float64), ref2.blob(grads=True), tol=1e-3, l2_tol=3e-4)
            _rel_err_by_array(game, hp, grad.astype(np.float64), want, tol=1e-2, l2_tol=1e-3)   # and against the reference's own masks, as before
        else:
            _rel_err_by_array(game, hp, grad.astype(np.float64), want, tol=1e-3)
        # the probe does not move the parameters or the running statistics
        assert np.array_equal(tr.trained_params(), nn.params())
    mem.close()


@pytest.mark.parametrize("reset_at", [None, 2])
def test_adam_steps_follow_torch(reset_at):
    """batch_updates!: three Adam steps on the device vs torch.optim.Adam on the fp64 restatement (same batches through the
    shuffling contract), incl. the L2 term and the BatchNorm running statistics (ResNetHP.batch_norm_momentum, unbiased running variance)"""
    import azhip
    game, B = 1, 32
    gspec, mem = _memory(game, 24, 4)
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=2)
    lp = azhip.LearningParams(samples_weighing_policy=1, l2_regularization=1e-3, loss_computation_batch_size=64, batch_size=B,
                              optimiser=azhip.Adam(lr=1e-3))
    with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
        data = tr.data.tensors()
        n = len(data[0])
        st0 = tr.learning_status()
        if reset_at is None:
            ls = tr.batch_updates(3, seed=11)
        else:                                                             # two calls: the optimiser state restarts, the batch stream continues
            ls = np.concatenate([tr.batch_updates(reset_at, seed=11), tr.batch_updates(3 - reset_at)])
        got = tr.trained_params()
        # replay the contract's shuffle: Fisher-Yates from the last index, draw k -> floor(u * (i + 1)), purpose 5, game word = epoch
        from test_arena_oracle import _u64
        perm = list(range(n))
        for k, i in enumerate(range(n - 1, 0, -1)):
            j = min(int(_u64(11, 0, 0, 5, k) * (i + 1)), i)
            perm[i], perm[j] = perm[j], perm[i]
        ref = TorchNet(game, hp, nn.params())
        train = [t for t in ref.p.values() if t.requires_grad]
        opt = torch.optim.Adam(train, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
        run = {k: v.detach().clone() for k, v in ref.p.items() if k.endswith(".mean") or k.endswith(".var")}
        losses = []
        for s in range(3):
            if s == reset_at:                                             # Flux.setup inside train!: a fresh state per call
                opt = torch.optim.Adam(train, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
            idx = perm[s * B:(s + 1) * B]
            opt.zero_grad()
            L, _ = ref.losses([x[idx] for x in data], float(tr.Wmean), float(tr.Hp), 1e-3, 1.0, 1.0)
            L.backward()
            opt.step()
            losses.append(L.item())
            mom = hp.batch_norm_momentum                                  # ResNetHP default 0.6 (resnet.jl:30-37)
            for pre, (mu, var, m) in ref.batch_stats.items():
                run[pre + ".mean"] = (1 - mom) * run[pre + ".mean"] + mom * mu
                run[pre + ".var"] = (1 - mom) * run[pre + ".var"] + mom * var * (m / (m - 1))
        for k, v in run.items():
            ref.p[k] = v
        want = ref.blob()
        assert np.allclose(ls, losses, rtol=2e-4, atol=2e-5), (ls, losses)
        assert np.abs(got - want).max() < 2e-4, np.abs(got - want).max()
        assert np.abs(got - nn.params()).max() > 5e-4               # it did move (3 steps of lr 1e-3)
        # installing the trained parameters lowers the loss on the training data
        nn2 = azhip.ResNet(gspec, hp, params=got)
    with azhip.Trainer(gspec, nn2, mem, lp, use_symmetries=True) as tr2:
        more = tr2.batch_updates(40, seed=11)
        st1 = tr2.learning_status()
    assert np.isfinite(more).all() and more[-5:].mean() < more[:5].mean()
    assert st0.loss.L == pytest.approx(st0.loss.L) and np.isfinite(st1.loss.L)
    mem.close()


def test_cyclic_nesterov_steps_follow_the_reference_rule():
    """CyclicNesterov (network.jl:163-180, flux.jl:78-94): Optimisers.Nesterov with lr / momentum from CyclicSchedule
    (schedule.jl:130-134); Flux.adjust! follows update!, so step i uses the schedule values of index i-1."""
    import azhip
    game, B, n = 1, 32, 6
    gspec, mem = _memory(game, 24, 6)
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=3)
    opt = azhip.CyclicNesterov(lr_base=1e-3, lr_high=1e-2, lr_low=5e-4, momentum_low=0.8, momentum_high=0.9)
    lp = azhip.LearningParams(samples_weighing_policy=0, l2_regularization=1e-4, loss_computation_batch_size=64, batch_size=B, optimiser=opt)

    def pl(xs, ys, i):
        pt = max([k for k in range(4) if xs[k] <= i], default=-1)
        if pt < 0:
            return ys[0]
        if pt == 3 or xs[pt + 1] == xs[pt]:
            return ys[pt]
        return ys[pt] + (ys[pt + 1] - ys[pt]) / (xs[pt + 1] - xs[pt]) * (i - xs[pt])
    xs = [1, int(0.45 * n), int(0.9 * n), n]
    with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
        data = tr.data.tensors()
        N = len(data[0])
        ls = tr.batch_updates(n, seed=2)
        got = tr.trained_params()
        from test_arena_oracle import _u64
        perm = list(range(N))
        for k, i in enumerate(range(N - 1, 0, -1)):
            j = min(int(_u64(2, 0, 0, 5, k) * (i + 1)), i)
            perm[i], perm[j] = perm[j], perm[i]
        ref = TorchNet(game, hp, nn.params())
        train = [t for t in ref.p.values() if t.requires_grad]
        vel = [torch.zeros_like(t) for t in train]
        for s in range(n):
            lr = 5e-4 if s == 0 else pl(xs, [1e-3, 1e-2, 1e-3, 5e-4], s)
            rho = 0.9 if s == 0 else pl(xs, [0.9, 0.8, 0.9, 0.9], s)
            for t in train:
                t.grad = None
            L, _ = ref.losses([x[perm[s * B:(s + 1) * B]] for x in data], float(tr.Wmean), float(tr.Hp), 1e-4, 1.0, 1.0)
            L.backward()
            assert abs(float(L.detach()) - ls[s]) < 3e-4 * max(1.0, abs(float(L.detach()))), (s, float(L.detach()), ls[s])
            with torch.no_grad():
                for t, v in zip(train, vel):
                    newdx = -rho * rho * v + (1 + rho) * lr * t.grad              # Optimisers.apply!(::Nesterov)
                    v.mul_(rho).sub_(lr * t.grad)
                    t.sub_(newdx)
        want = ref.blob()
        mask = np.ones(len(want), dtype=bool)
        off = 0
        for name, shape in param_layout(game, hp):
            k = int(np.prod(shape))
            if name.endswith(".mean") or name.endswith(".var"):
                mask[off:off + k] = False
            off += k
        assert np.abs(got[mask] - want[mask]).max() < 3e-4, np.abs(got[mask] - want[mask]).max()
    mem.close()


def test_trainer_errors_and_lifecycle():
    import ctypes as C
    import azhip
    from azhip import _lib as L
    gspec, mem = _memory(1, 6, 1)
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=1)
    lp = azhip.LearningParams(samples_weighing_policy=0, l2_regularization=0.0, loss_computation_batch_size=8, batch_size=1 << 20)
    with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=False) as tr:
        assert tr.batch_size() == tr.num_samples()                  # min(batch_size, #samples), learning.jl:113
        ls = tr.batch_updates(2)                                    # every epoch is one full batch
        assert np.isfinite(ls).all() and tr.batch_updates(0).size == 0
        with pytest.raises(L.AzError, match="out of range"):
            tr.gradients(np.full(tr.batch_size(), 10 ** 6))
        cfg = L.TrainCfg()
        L.check(L.lib().az_train_cfg_init(C.byref(cfg)))
        h = C.c_void_p()
        cfg.optimiser = 7
        assert L.lib().az_trainer_create(tr._eng._h, tr.data._h, C.byref(cfg), C.byref(h)) == L.AZ_ERR_BAD_ARG and b"optimiser" in L.lib().az_last_error()
        cfg.optimiser, cfg.struct_size = 0, 12
        assert L.lib().az_trainer_create(tr._eng._h, tr.data._h, C.byref(cfg), C.byref(h)) == L.AZ_ERR_BAD_ARG
        L.check(L.lib().az_train_cfg_init(C.byref(cfg)))
        cfg.batch_size = 1
        assert L.lib().az_trainer_create(tr._eng._h, tr.data._h, C.byref(cfg), C.byref(h)) == L.AZ_ERR_BAD_ARG and b"batch" in L.lib().az_last_error()
        with azhip.Engine(game=L.GAME_TICTACTOE, oracle=L.ORACLE_HASH, num_workers=4, batch_size=4, num_iters_per_turn=4) as e2:
            L.check(L.lib().az_train_cfg_init(C.byref(cfg)))
            assert L.lib().az_trainer_create(e2._h, tr.data._h, C.byref(cfg), C.byref(h)) == L.AZ_ERR_STATE
    assert L.lib().az_trainer_destroy(None) == 0
    mem.close()


def test_weight_gradient_stream_does_not_change_the_values(monkeypatch):
    """round 3: k_wgrad16 runs on a second stream beside the batch-norm backward passes of the next layer; with
    AZHIP_TRAIN_ONE_STREAM=1 it stays in line.  Same kernels, same partial sums: the gradients are bit-identical."""
    import azhip
    gspec, mem = _memory(0, 40, 3)
    hp = azhip.ResNetHP(num_blocks=4, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=3)
    lp = azhip.LearningParams(samples_weighing_policy=1, l2_regularization=1e-4, loss_computation_batch_size=64, batch_size=150)
    grads = []
    for one in ("0", "1"):
        monkeypatch.setenv("AZHIP_TRAIN_ONE_STREAM", one)
        with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
            idx = np.arange(150) * 3 % len(tr.data.tensors()[0])
            g = [tr.gradients(idx)[2].copy() for _ in range(3)]     # repeated: a race would not repeat itself
            assert all(np.array_equal(g[0], x) for x in g[1:])
            grads.append(g[0])
    assert np.array_equal(grads[0], grads[1])
    mem.close()


def test_column_sums_finished_inside_the_producer_do_not_change_the_values(monkeypatch):
    """round 4's opt-in form (AZHIP_TRAIN_FINISH_INSIDE=1: the second stage of every column sum in the producer's last workgroup,
    release / acquire on its counter -- ADVICE r4) adds the same partials in the same order as the separate launch: trained
    parameters bit-identical, and repeatable"""
    import azhip
    gspec, mem = _memory(0, 40, 3)
    hp = azhip.ResNetHP(num_blocks=3, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    lp = azhip.LearningParams(samples_weighing_policy=1, l2_regularization=1e-4, loss_computation_batch_size=64, batch_size=150)
    outs = []
    for inside in ("0", "1", "1"):
        monkeypatch.setenv("AZHIP_TRAIN_FINISH_INSIDE", inside)
        nn = azhip.ResNet(gspec, hp, seed=3)
        with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=True) as tr:
            idx = np.arange(150) * 3 % len(tr.data.tensors()[0])
            outs.append(tr.gradients(idx)[2].copy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    mem.close()
