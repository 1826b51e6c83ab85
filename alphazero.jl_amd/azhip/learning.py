"""Learning-side mirror (src/learning.jl): Trainer's data, losses, learning_status, samples_report, batch_updates!.

Everything runs on the device: the data path of a Trainer (symmetry augmentation, merge_by_state, convert_samples),
the loss evaluation with the network in test mode (Report.LearningStatus) and the optimiser steps (forward with
batch statistics, backward, Adam / CyclicNesterov: az_trainer_*)."""
import ctypes as C
from dataclasses import dataclass

from . import _lib as L
from .engine import Engine
from .memory import Dataset, MemoryBuffer

CONSTANT_WEIGHT, LOG_WEIGHT, LINEAR_WEIGHT = L.WEIGHT_CONSTANT, L.WEIGHT_LOG, L.WEIGHT_LINEAR


@dataclass
class LearningParams:
    """params.jl:235-248 (the fields the evaluation half reads)"""
    samples_weighing_policy: int
    l2_regularization: float
    loss_computation_batch_size: int
    batch_size: int = 1024
    use_gpu: bool = True
    use_position_averaging: bool = True
    rewards_renormalization: float = 1.0
    nonvalidity_penalty: float = 1.0
    optimiser: object = None
    min_checkpoints_per_epoch: int = 1
    max_batches_per_checkpoint: int = 2000
    num_checkpoints: int = 1


@dataclass
class Adam:
    """network.jl:182-190"""
    lr: float = 2e-3


@dataclass
class CyclicNesterov:
    """network.jl:163-180"""
    lr_base: float
    lr_high: float
    lr_low: float
    momentum_low: float
    momentum_high: float


@dataclass
class Loss:
    """Report.Loss"""
    L: float
    Lp: float
    Lv: float
    Lreg: float
    Linv: float


@dataclass
class LearningStatus:
    """Report.LearningStatus"""
    loss: Loss
    Hp: float
    Hpnet: float


@dataclass
class Samples:
    """Report.Samples"""
    num_samples: int
    num_boards: int
    Wtot: float
    status: LearningStatus


class Trainer:
    """Trainer(gspec, network, samples, params; test_mode) (learning.jl:98-121) over a device MemoryBuffer:
    merge_by_state (if use_position_averaging), convert_samples, Wmean, Hp; the network in test mode."""

    def __init__(self, gspec, network, mem: MemoryBuffer, params: LearningParams, use_symmetries=False, last_batch=False, device=0):
        self.gspec, self.params = gspec, params
        self.data = Dataset(mem, last_batch, use_symmetries, params.use_position_averaging, params.samples_weighing_policy)
        self.Wmean, self.Hp = self.data.Wmean, self.data.Hp
        self._hyper = network.hyper
        kw = network.engine_options()
        self._eng = Engine(game=gspec.game_id, oracle=L.ORACLE_RESNET, device=device, num_workers=8, batch_size=8,
                           num_iters_per_turn=2, **kw)
        self._eng.net_set_params(network.params())

    def close(self):
        if getattr(self, "_tr", None):
            L.lib().az_trainer_destroy(self._tr)
            self._tr = None
        self.data.close()
        self._eng.close()

    # ---- the optimiser step (learning.jl:123-141) ----
    def _trainer(self, optimiser=None, batch_norm_momentum=None, seed=1):
        if getattr(self, "_tr", None):
            return self._tr
        cfg = L.TrainCfg()
        L.check(L.lib().az_train_cfg_init(C.byref(cfg)))
        opt = optimiser if optimiser is not None else getattr(self.params, "optimiser", None) or Adam()
        if isinstance(opt, Adam):
            cfg.optimiser, cfg.lr = L.OPT_ADAM, opt.lr
        else:
            cfg.optimiser = L.OPT_CYCLIC_NESTEROV
            cfg.lr_base, cfg.lr_high, cfg.lr_low = opt.lr_base, opt.lr_high, opt.lr_low
            cfg.momentum_low, cfg.momentum_high = opt.momentum_low, opt.momentum_high
        p = self.params
        cfg.l2_regularization, cfg.nonvalidity_penalty = p.l2_regularization, p.nonvalidity_penalty
        cfg.rewards_renormalization, cfg.batch_size = p.rewards_renormalization, p.batch_size
        cfg.batch_norm_momentum = self._hyper.batch_norm_momentum if batch_norm_momentum is None else batch_norm_momentum   # ResNetHP, resnet.jl:30-37
        cfg.seed = seed
        h = C.c_void_p()
        L.check(L.lib().az_trainer_create(self._eng._h, self.data._h, C.byref(cfg), C.byref(h)))
        self._tr = h
        return h

    def batch_size(self):
        return min(self.params.batch_size, self.num_samples())

    def batch_updates(self, n, **kw):
        """batch_updates!(tr, n): n optimiser steps; returns the losses (Float32) like the reference's `ls`"""
        import numpy as np
        tr = self._trainer(**kw)
        ls = np.zeros(n, dtype=np.float32)
        L.check(L.lib().az_trainer_batch_updates(tr, int(n), ls.ctypes.data_as(C.c_void_p)))
        return ls

    def gradients(self, sample_idx, **kw):
        """parity hook: (loss, (Lp, Lv, Lreg, Linv, scale), data gradient in Flux parameter order) of one batch, no update"""
        import numpy as np
        tr = self._trainer(**kw)
        idx = np.ascontiguousarray(sample_idx, dtype=np.int32)
        assert len(idx) == self.batch_size()
        n = self._eng.net_num_params()
        g = np.zeros(n, dtype=np.float32)
        parts = np.zeros(5, dtype=np.float32)
        loss = C.c_float()
        L.check(L.lib().az_trainer_gradients(tr, idx.ctypes.data_as(C.c_void_p), C.byref(loss), parts.ctypes.data_as(C.c_void_p),
                                             g.ctypes.data_as(C.c_void_p), n))
        return loss.value, parts, g

    def trained_params(self, **kw):
        """get_trained_network(tr): the parameter blob after the updates so far"""
        import numpy as np
        tr = self._trainer(**kw)
        out = np.zeros(self._eng.net_num_params(), dtype=np.float32)
        L.check(L.lib().az_trainer_get_params(tr, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def num_samples(self):
        return len(self.data)

    def num_batches_total(self):
        return self.num_samples() // self.params.batch_size

    def install_trained(self):
        """make the loss evaluation (and get_trained_network) see the parameters after the updates so far"""
        if getattr(self, "_tr", None):
            self._eng.net_set_params(self.trained_params())

    def learning_status(self):
        """learning_status(tr) (learning.jl:170-181), network in TEST mode (running statistics) as memory_report does.
        Deviation, documented: inside learning_step! the reference's Trainer holds the network in train mode, so its
        status figures use batch statistics (and its evaluation passes move the running statistics)."""
        p = self.params
        out = L.LearningStatusRec()
        L.check(L.lib().az_learning_status(self._eng._h, self.data._h, float(p.l2_regularization), float(p.nonvalidity_penalty),
                                           float(p.rewards_renormalization), int(p.loss_computation_batch_size), C.byref(out)))
        return LearningStatus(Loss(out.L, out.Lp, out.Lv, out.Lreg, out.Linv), out.Hp, out.Hpnet)

    def samples_report(self):
        """samples_report(tr) (learning.jl:183-190); num_boards needs the merged count"""
        status = self.learning_status()
        if self.params.use_position_averaging:
            num_boards = len(self.data)
        else:
            raise NotImplementedError("num_boards of an un-merged Trainer: build a second data set with use_position_averaging")
        return Samples(self.data.sum_n, num_boards, self.data.Wtot, status)


@dataclass
class StageSamples:
    """Report.StageSamples"""
    min_remaining_length: float
    max_remaining_length: float
    samples_stats: Samples


@dataclass
class Memory:
    """Report.Memory"""
    latest_batch: Samples
    all_samples: Samples
    per_game_stage: list


def memory_report(mem: MemoryBuffer, nn, learning_params: LearningParams, num_game_stages: int, device=0):
    """memory_report(mem, nn, learning_params, params::MemAnalysisParams) (learning.jl:192-216): samples_report of all
    samples, of the latest batch, and of num_game_stages slices of the samples sorted by remaining game length t.
    The network is evaluated in test mode.  Stage slices are staged through a scratch device buffer."""
    import math
    gspec = mem.gspec

    def report(m, last_batch=False):
        with Trainer(gspec, nn, m, learning_params, last_batch=last_batch, device=device) as tr:
            return tr.samples_report()
    all_samples = report(mem)
    latest = report(mem, last_batch=True) if mem.cur_batch_size() > 0 else all_samples
    with mem.dataset() as d:
        raw = d.raw_samples()
        es = sorted((raw[i] for i in range(len(d))), key=lambda e: e.t)            # sort!(es, by = e -> e.t), stable
    csize = math.ceil(len(es) / num_game_stages)
    stages = []
    for k in range(0, len(es), csize):
        part = es[k:k + csize]
        scratch = MemoryBuffer(gspec, len(part), device=device)
        try:
            scratch.push_samples(part)
            stages.append(StageSamples(min(e.t for e in part), max(e.t for e in part), report(scratch)))
        finally:
            scratch.close()
    return Memory(latest, all_samples, stages)
