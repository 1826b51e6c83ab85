"""Learning-side mirror, evaluation half (src/learning.jl): Trainer's data, losses, learning_status, samples_report.

The backward pass / optimiser step (batch_updates!, Network.train!) is not on the device yet (SURVEY.md §8f
rank 1); what is: the whole data path of a Trainer (symmetry augmentation, merge_by_state, convert_samples) and the
loss evaluation with the network in test mode -- Report.LearningStatus as learning_step! / memory_report print it."""
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
        kw = network.engine_options()
        self._eng = Engine(game=gspec.game_id, oracle=L.ORACLE_RESNET, device=device, num_workers=8, batch_size=8,
                           num_iters_per_turn=2, **kw)
        self._eng.net_set_params(network.params())

    def close(self):
        self.data.close()
        self._eng.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def num_samples(self):
        return len(self.data)

    def num_batches_total(self):
        return self.num_samples() // self.params.batch_size

    def learning_status(self):
        """learning_status(tr) (learning.jl:170-181)"""
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
