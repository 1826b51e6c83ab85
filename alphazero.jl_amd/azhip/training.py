"""The caller of the path: self_play_step! (src/training.jl:275-300) and its report (src/report.jl:203-209).

Everything else of training.jl (learning step, arena, checkpoints) stays in the reference; this mirror only shows
that the seam `simulate_distributed -> push_trace! -> Report.SelfPlay` is served by the engine."""
import time
from dataclasses import dataclass

import numpy as np

from .memory import push_trace
from .network import copy as network_copy
from .params import MctsParams, SimParams
from .play import MctsPlayer
from .simulations import Simulator, self_play_measurements, simulate_distributed


@dataclass
class SelfPlayParams:
    """params.jl:110-118"""
    mcts: MctsParams
    sim: SimParams


@dataclass
class SelfPlayReport:
    """Report.SelfPlay, report.jl:203-209"""
    samples_gen_speed: float      # samples (positions) per second
    average_exploration_depth: float
    mcts_memory_footprint: int
    memory_size: int
    memory_num_distinct_boards: int


def broadcast_params(nn, src=0, group=None):
    """Weights to every rank before a self-play phase: torch.distributed.broadcast of the fp32
    parameter blob (RCCL over xGMI with the nccl backend), in place of the reference shipping the whole network
    inside the @spawnat closure (simulations.jl:271-281)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return nn
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(nn.params(), dtype=np.float32).copy()).to(dev)
    dist.broadcast(t, src=src, group=group)
    nn._params[:] = t.cpu().numpy()
    nn.gc()
    return nn


def self_play_step(gspec, bestnn, params: SelfPlayParams, memory, game_played=None, seed=1):
    """self_play_step!(env, handler): plays params.sim.num_games games with the best network, pushes the traces
    into `memory` (a list of TrainingSample) and returns the Report.SelfPlay numbers."""
    def make_oracle():
        return network_copy(bestnn, on_gpu=params.sim.use_gpu, test_mode=True)
    simulator = Simulator(lambda oracle: MctsPlayer(gspec, oracle, params.mcts), make_oracle, self_play_measurements)
    t0 = time.perf_counter()
    results = simulate_distributed(simulator, gspec, params.sim, game_simulated=game_played, seed=seed)
    elapsed = time.perf_counter() - t0
    n0 = len(memory)
    for x in results:
        push_trace(memory, x["trace"], params.mcts.gamma)
    new = len(memory) - n0
    return SelfPlayReport(samples_gen_speed=new / elapsed,
                          average_exploration_depth=float(np.mean([x["edepth"] for x in results])),
                          mcts_memory_footprint=int(max(x["mem"] for x in results)),
                          memory_size=len(memory),
                          memory_num_distinct_boards=len({s.s for s in memory}))
