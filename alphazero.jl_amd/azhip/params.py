"""Parameters of the path: MctsParams / SimParams (src/params.jl:49-57,92-101) and the schedules they
use (src/schedule.jl:18-25,49-80)."""
import math
from dataclasses import dataclass, field
from typing import Optional


class ConstSchedule:
    """schedule.jl:18-25"""

    def __init__(self, value):
        self.value = value

    def __getitem__(self, i):
        return self.value

    def breakpoints(self):
        return [0], [self.value]


class PLSchedule:
    """Piecewise linear schedule, schedule.jl:49-80 (integer schedules round up)."""

    def __init__(self, xs, ys=None):
        if ys is None:
            xs, ys = [0], [xs]
        assert len(xs) > 0 and len(xs) == len(ys)
        self.xs, self.ys = list(xs), list(ys)

    def __getitem__(self, i):
        ptidx = None
        for k, x in enumerate(self.xs):
            if x <= i:
                ptidx = k
        if ptidx is None:
            return self.ys[0]
        if ptidx == len(self.xs) - 1:
            return self.ys[-1]
        x0, y0, x1, y1 = self.xs[ptidx], self.ys[ptidx], self.xs[ptidx + 1], self.ys[ptidx + 1]
        y = y0 + (y1 - y0) / (x1 - x0) * (i - x0)
        if all(isinstance(v, int) for v in self.ys):
            y = int(math.ceil(y))
        return y

    def breakpoints(self):
        return self.xs, self.ys


@dataclass
class MctsParams:
    """params.jl:49-57"""
    num_iters_per_turn: int
    dirichlet_noise_ϵ: float
    dirichlet_noise_α: float
    gamma: float = 1.0
    cpuct: float = 1.0
    temperature: object = field(default_factory=lambda: ConstSchedule(1.0))
    prior_temperature: float = 1.0


@dataclass
class ArenaParams:
    """params.jl:139-143"""
    mcts: "MctsParams"
    sim: "SimParams"
    update_threshold: float


@dataclass
class SimParams:
    """params.jl:92-101"""
    num_games: int
    num_workers: int
    batch_size: int
    use_gpu: bool = False
    fill_batches: bool = True
    reset_every: Optional[int] = 1
    flip_probability: float = 0.0
    alternate_colors: bool = False
    # not a field of the reference: az_engine_cfg.lock_step (include/azhip.h).  False = free-running workers (which worker plays which
    # game is an outcome of the reference's own race, util.jl:181-188, reported per game); True = rounds in lock step, fixed assignment
    lock_step: bool = False


def check_sim_params(p: SimParams, arena=False):
    """the part of check_params (params.jl:361-384) that concerns this path"""
    if p.batch_size > p.num_workers:
        raise ValueError("batch_size must be <= num_workers")
    if not 0.0 <= p.flip_probability <= 1.0:
        raise ValueError("flip_probability must be in [0, 1]")


def engine_options(mcts: MctsParams, sim: SimParams, seed=1, arena=False):
    """MctsParams + SimParams -> az_engine_cfg fields (SURVEY.md §8b 'Config mapping')."""
    check_sim_params(sim, arena=arena)
    xs, ys = mcts.temperature.breakpoints()
    return dict(gamma=mcts.gamma, cpuct=mcts.cpuct, dirichlet_noise_eps=mcts.dirichlet_noise_ϵ,
                dirichlet_noise_alpha=mcts.dirichlet_noise_α, prior_temperature=mcts.prior_temperature,
                num_iters_per_turn=mcts.num_iters_per_turn, temperature=(xs, ys),
                num_workers=sim.num_workers, batch_size=sim.batch_size,
                reset_every=0 if sim.reset_every is None else sim.reset_every,
                fill_batches=1 if sim.fill_batches else 0, flip_probability=sim.flip_probability, seed=seed,
                lock_step=1 if getattr(sim, "lock_step", False) else 0)
