"""Network plugin mirror (src/networks/network.jl:30-328, src/networks/flux.jl, architectures/resnet.jl).

`ResNet` keeps its parameters as ONE fp32 blob in Flux array order (the layout az_net_set_params takes)
and evaluates through the HIP tower/heads kernels.  Training (`train!`, optimisers) is outside the
hot path and raises NotImplementedError.
"""
from dataclasses import dataclass, replace

import numpy as np

from . import _lib as L
from .engine import Engine


@dataclass(frozen=True)
class ResNetHP:
    """resnet.jl:30-37"""
    num_blocks: int
    num_filters: int
    conv_kernel_size: tuple = (3, 3)
    num_policy_head_filters: int = 2
    num_value_head_filters: int = 1
    batch_norm_momentum: float = 0.6


_DIMS = {L.GAME_CONNECT_FOUR: (7, 6, 3, 7), L.GAME_TICTACTOE: (3, 3, 3, 9), L.GAME_MANCALA: (14, 1, 5, 6),
         L.GAME_GO9_PLANES: (9, 9, 4, 82)}


def param_layout(game, hp):
    """[(name, Flux shape)] in blob order; shapes are Julia (column-major) shapes."""
    W, H, C, A = _DIMS[game]
    P, F, npf, nvf = W * H, hp.num_filters, hp.num_policy_head_filters, hp.num_value_head_filters

    def bn(prefix, n):
        return [(prefix + ".gamma", (n,)), (prefix + ".beta", (n,)), (prefix + ".mean", (n,)), (prefix + ".var", (n,))]

    out = [("stem.conv.W", (3, 3, C, F)), ("stem.conv.b", (F,))] + bn("stem.bn", F)
    for b in range(hp.num_blocks):
        for k in (1, 2):
            out += [("block%d.conv%d.W" % (b, k), (3, 3, F, F)), ("block%d.conv%d.b" % (b, k), (F,))] + bn("block%d.bn%d" % (b, k), F)
    out += [("phead.conv.W", (1, 1, F, npf)), ("phead.conv.b", (npf,))] + bn("phead.bn", npf)
    out += [("phead.dense.W", (A, P * npf)), ("phead.dense.b", (A,))]
    out += [("vhead.conv.W", (1, 1, F, nvf)), ("vhead.conv.b", (nvf,))] + bn("vhead.bn", nvf)
    out += [("vhead.dense1.W", (F, P * nvf)), ("vhead.dense1.b", (F,)), ("vhead.dense2.W", (1, F)), ("vhead.dense2.b", (1,))]
    return out


def num_parameters(game, hp):
    return int(sum(int(np.prod(s)) for _, s in param_layout(game, hp)))


def random_params(game, hp, seed=2026):
    """Synthetic weights: Glorot-uniform conv/dense kernels, U(-0.1,0.1) biases and
    BatchNorm statistics perturbed by U(-0.1,0.1) around (gamma, beta, mean, var) = (1, 0, 0, 1), all
    drawn from numpy's Philox(seed)."""
    rng = np.random.Generator(np.random.Philox(seed))
    parts = []
    for name, shape in param_layout(game, hp):
        n = int(np.prod(shape))
        if name.endswith(".W"):
            if len(shape) == 4:
                fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
            else:
                fan_out, fan_in = shape
            s = np.sqrt(6.0 / (fan_in + fan_out))
            parts.append(rng.uniform(-s, s, n))
        elif name.endswith(".gamma") or name.endswith(".var"):
            parts.append(1.0 + rng.uniform(-0.1, 0.1, n))
        else:
            parts.append(rng.uniform(-0.1, 0.1, n))
    return np.concatenate(parts).astype(np.float32)


def split_params(game, hp, blob):
    """blob -> {name: ndarray with the Julia shape (Fortran order view)}"""
    out, off = {}, 0
    for name, shape in param_layout(game, hp):
        n = int(np.prod(shape))
        out[name] = np.asarray(blob[off:off + n]).reshape(shape, order="F")
        off += n
    assert off == len(blob)
    return out


class ResNet:
    """ResNet <: TwoHeadNetwork (resnet.jl:45-92): ResNet(gspec, hyper)."""

    HyperParams = ResNetHP

    def __init__(self, gspec, hyper, params=None, seed=2026, device=0, bf16=False):
        """bf16: evaluate the residual tower in bfloat16 (az_engine_cfg.net_bf16, csrc/resnet16b.h); the parameters stay fp32"""
        self.gspec = gspec
        self.hyper = hyper
        self.device = device
        self.bf16 = bool(bf16)
        self._params = random_params(gspec.game_id, hyper, seed) if params is None else np.asarray(params, dtype=np.float32).copy()
        if self._params.size != num_parameters(gspec.game_id, hyper):
            raise ValueError("parameter blob has the wrong size")
        self._on_gpu = False
        self._test_mode = False
        self._engine = None

    # -- Network interface (network.jl:30-206) --
    def hyperparams(self):
        return self.hyper

    def game_spec(self):
        return self.gspec

    def params(self):
        return self._params

    def num_parameters(self):
        # trainable parameters only (BatchNorm mean/var are state): network.jl:218-220
        lay = param_layout(self.gspec.game_id, self.hyper)
        return int(sum(int(np.prod(s)) for n, s in lay if not (n.endswith(".mean") or n.endswith(".var"))))

    def regularized_params(self):
        p = split_params(self.gspec.game_id, self.hyper, self._params)
        return [v for k, v in p.items() if k.endswith(".W")]

    def on_gpu(self):
        return self._on_gpu

    def to_gpu(self):
        nn = self.copy_()
        nn._on_gpu = True
        return nn

    def to_cpu(self):
        nn = self.copy_()
        nn._on_gpu = False
        return nn

    def set_test_mode(self, mode=True):
        self._test_mode = bool(mode)

    def copy_(self):
        nn = ResNet(self.gspec, self.hyper, params=self._params, device=self.device, bf16=self.bf16)
        nn._on_gpu, nn._test_mode = self._on_gpu, self._test_mode
        return nn

    def gc(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def train(self, *a, **k):
        raise NotImplementedError("train! is outside the self-play hot path (SURVEY.md §8f)")

    # -- evaluation --
    def engine_options(self):
        h = self.hyper
        if tuple(h.conv_kernel_size) != (3, 3):
            raise ValueError("only 3x3 kernels are supported")
        return dict(num_blocks=h.num_blocks, num_filters=h.num_filters,
                    num_policy_head_filters=h.num_policy_head_filters,
                    num_value_head_filters=h.num_value_head_filters, net_bf16=1 if self.bf16 else 0)

    def _eng(self):
        if self._engine is None:
            self._engine = Engine(game=self.gspec.game_id, oracle=L.ORACLE_RESNET, device=self.device,
                                  num_workers=1, batch_size=1, num_iters_per_turn=2, **self.engine_options())
            self._engine.net_set_params(self._params)
        return self._engine

    def forward_normalized(self, X, A):
        """network.jl:264-271.  X: (N, C, H, W) == the Julia W x H x C x N array in memory; A: (N, nA)."""
        return self._eng().net_forward(X, A)

    def evaluate_batch(self, batch):
        """network.jl:308-315: list of states -> list of (P over available actions, V)."""
        keys = np.array([s for s in batch], dtype=np.uint64).reshape(-1, 2)
        P, V = self._eng().net_evaluate_keys(keys)
        _, A = self._eng().encode(keys)
        return [(P[i][A[i] > 0], float(V[i])) for i in range(len(batch))]

    def evaluate(self, state):
        """network.jl:287-298"""
        return self.evaluate_batch([state])[0]

    __call__ = evaluate


def copy(nn, on_gpu, test_mode):
    """Network.copy(nn; on_gpu, test_mode), network.jl:323-328"""
    nn = nn.copy_()
    nn = nn.to_gpu() if on_gpu else nn.to_cpu()
    nn.set_test_mode(test_mode)
    return nn


def with_filters(hp, **kw):
    return replace(hp, **kw)


MAGIC = b"AZHIPNET"


def save_params(path, nn):
    """Raw checkpoint of a network: magic, game id, ResNetHP fields (int32), parameter count (int64), Float32 little-endian
    blob in Flux parameter order -- the file julia/AlphaZeroHIP.jl's save_params writes (checkpoint interop)."""
    import struct
    h = nn.hyper
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<5iq", nn.gspec.game_id, h.num_blocks, h.num_filters, h.num_policy_head_filters, h.num_value_head_filters,
                            nn.params().size))
        f.write(np.ascontiguousarray(nn.params(), dtype="<f4").tobytes())


def load_params(path, gspec):
    import struct
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not an AZHIPNET parameter file")
        game, nb, nf, npf, nvf, n = struct.unpack("<5iq", f.read(28))
        if game != gspec.game_id:
            raise ValueError("parameter file is for game %d" % game)
        blob = np.frombuffer(f.read(4 * n), dtype="<f4").astype(np.float32)
    hp = ResNetHP(num_blocks=nb, num_filters=nf, num_policy_head_filters=npf, num_value_head_filters=nvf)
    return ResNet(gspec, hp, params=blob)
