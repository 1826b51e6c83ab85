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


def repack_by_game_id(g, m):
    """gathered (game, move) record arrays -> the same records in game-id order with contiguous move ranges, so
    that every rank's replay memory receives the samples in the same order whatever the number of ranks"""
    order = np.argsort(g["game_id"], kind="stable")
    mm = np.concatenate([m[g["first_move"][i]:g["first_move"][i] + g["num_moves"][i]] for i in order]) if len(order) else m[:0]
    g = g[order].copy()
    if len(g):
        g["first_move"] = np.concatenate([[0], np.cumsum(g["num_moves"])[:-1]])
    return g, mm


def self_play_step_device(gspec, bestnn, params: SelfPlayParams, memory, game_played=None, seed=1, group=None, comm=None):
    """self_play_step!(env, handler) with the replay memory on the device.  Returns the Report.SelfPlay numbers.

    One process: the phase runs device-only -- the move records never leave HBM (the engine's phase buffer ->
    az_memory_push_engine does push_trace! for every game on the GPU); the host sees the game records (56 B per game).
    Several ranks with a native communicator (`comm`: azhip.comm.Comm, RCCL behind the C ABI): every rank simulates its
    shard device-only and az_comm_gather_push all-gathers the records device to device into every rank's memory.
    Several ranks with only a torch.distributed group (the gloo CPU tests): records are gathered through torch."""
    import ctypes as C
    import sys

    # torch.distributed only when the caller already set it up (importing torch costs seconds and is plumbing here)
    dist = sys.modules.get("torch.distributed")

    from . import _lib as L
    from .simulations import gather_records, records_to_numpy, run_local, shard_games

    def make_oracle():
        return network_copy(bestnn, on_gpu=params.sim.use_gpu, test_mode=True)
    simulator = Simulator(lambda oracle: MctsPlayer(gspec, oracle, params.mcts), make_oracle, self_play_measurements)
    t0 = time.perf_counter()
    sim = params.sim
    first, device = 0, 0
    torch_group = comm is None and dist is not None and dist.is_available() and dist.is_initialized()
    if comm is not None:
        first, count = shard_games(sim.num_games, comm.world, comm.rank)
        sim = SimParams(**{**sim.__dict__, "num_games": count})
        device = getattr(comm, "device", 0)
    elif torch_group:
        import torch
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        first, count = shard_games(sim.num_games, world, rank)
        sim = SimParams(**{**sim.__dict__, "num_games": count})
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    games, moves, ng, nm, stats, eng = run_local(simulator, gspec, sim, first, game_played, device, seed, device_only=not torch_group)
    # `elapsed` is the time of simulate_distributed alone (training.jl:284-287: its fetch of the workers' results is inside,
    # push_trace! is not): the clock stops here, plus the gather where there is one
    elapsed = time.perf_counter() - t0
    # A slot that ran out of tree nodes / move records (the reference's Dict has no bound, src/mcts.jl:124-151) was retired and its
    # game played again under a replacement id, so the count is right unless a replacement overflowed too; long games are the ones
    # that go, which biases the data: warn, and refuse when it is more than a few.  The verdict is formed AFTER the
    # exchange, from numbers every rank sees alike (a rank raising on its own left the others in the all-gather):
    # `asked` games were wanted over all ranks, `came` came back, `replaced` of those carry the replacement bit.
    def abort_policy(asked, came, replaced):
        aborted = replaced + (asked - came)                         # games played again + games given up for good
        if not aborted:
            return
        import warnings
        msg = ("azhip: %d of %d self-play games were aborted (tree node pool / move record full) and played again under replacement "
               "ids; %d games came back.  Raise max_nodes_per_slot / max_moves_per_game." % (aborted, asked, came))
        if aborted * 20 > max(asked, 1) or came < asked:
            raise L.AzError(L.AZ_ERR_CAPACITY, msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
    if comm is None and not torch_group:
        abort_policy(sim.num_games, ng, stats.aborted_games)        # one process: nothing is pushed when the phase is refused
    memory.new_batch()
    depth = footprint = None
    if comm is not None:
        # gather first (no memory: nothing is pushed), judge the phase from numbers every rank sees alike, THEN gather into the memory
        # (a refused phase must not leave its samples behind; the records of a phase are a few MB, the second collective ~1 ms)
        probe = comm.gather_push(eng, None, params.mcts.gamma)
        abort_policy(params.sim.num_games, int(probe.games), int(probe.replaced_games))   # every rank alike: nobody is left inside a collective
        gs = comm.gather_push(eng, memory, params.mcts.gamma)
        nm = gs.moves
        elapsed += gs.gather_ms * 1e-3
        # mean / maximum over ALL ranks' games (training.jl:293-294), from the gathered game records
        depth, footprint = float(gs.mean_game_depth), int(gs.max_nodes)
        eng.release_phase()                                         # the cached engine keeps its trees' pools, not the records
    elif torch_group:
        tg = time.perf_counter()
        g, mm = repack_by_game_id(*gather_records(*records_to_numpy(games, moves, ng, nm), group))
        ng, nm = len(g), len(mm)
        abort_policy(params.sim.num_games, ng, int(np.count_nonzero(g["game_id"] & L.REPLACEMENT_GAME_BIT)))
        elapsed += time.perf_counter() - tg
        games = (L.GameRec * max(ng, 1)).from_buffer_copy(g.tobytes() or bytes(C.sizeof(L.GameRec)))
        moves = (L.MoveRec * max(nm, 1)).from_buffer_copy(mm.tobytes() or bytes(C.sizeof(L.MoveRec)))
        memory.push_records(games, moves, ng, nm, params.mcts.gamma)
    else:
        memory.push_engine(eng, params.mcts.gamma)
        nm = stats.moves
        eng.release_phase()
    if depth is None:
        depth = float(np.mean([games[i].total_nodes_traversed / max(games[i].total_simulations, 1) for i in range(ng)])) if ng else 0.0
        footprint = int(max((games[i].nodes for i in range(ng)), default=0))
    with memory.dataset(use_position_averaging=True) as d:
        distinct = len(d)
    return SelfPlayReport(samples_gen_speed=nm / elapsed, average_exploration_depth=depth,
                          mcts_memory_footprint=footprint,
                          memory_size=len(memory), memory_num_distinct_boards=distinct)


def evaluation_half_of_learning_step(gspec, curnn, bestnn, memory, learning_params, arena_params, use_symmetries=True,
                                     handler=None, seed=1):
    """What learning_step! (training.jl:193-259) does around the optimiser: Trainer over the (augmented, merged)
    experience, learning_status, then the checkpoint evaluation compare_networks(curnn, bestnn) and the replacement
    decision (avgr >= update_threshold).  Returns (LearningStatus, Evaluation, replace::bool)."""
    from .arena import compare_networks
    from .learning import Trainer
    with Trainer(gspec, curnn, memory, learning_params, use_symmetries=use_symmetries) as tr:
        status = tr.learning_status()
    ev = compare_networks(gspec, curnn, bestnn, arena_params, handler, seed=seed)
    return status, ev, ev.avgr >= arena_params.update_threshold


@dataclass
class Checkpoint:
    """Report.Checkpoint"""
    batch_id: int
    evaluation: object
    status_after: object
    nn_replaced: bool


@dataclass
class LearningReport:
    """Report.Learning (report.jl)"""
    time_convert: float
    time_loss: float
    time_train: float
    time_eval: float
    initial_status: object
    losses: np.ndarray
    checkpoints: list
    nn_replaced: bool


def learning_step(gspec, curnn, bestnn, memory, learning_params, arena_params=None, use_symmetries=True, handler=None, seed=1):
    """learning_step!(env, handler) (training.jl:193-259) on the device: Trainer over the (augmented, merged) experience,
    num_checkpoints x [batch_updates!(nbatches) -> learning_status -> compare_networks(curnn, bestnn)] with the
    replacement rule avgr >= best so far (initially update_threshold).  Returns (curnn, bestnn, LearningReport)."""
    from .arena import compare_networks
    from .learning import Trainer
    from .network import ResNet
    lp, ap = learning_params, arena_params

    def call(name, *a):                                           # Handlers.<name>(handler, ...), training.jl:40-75 (all optional)
        f = getattr(handler, name, None)
        if f is not None:
            f(*a)
    t0 = time.perf_counter()
    tr = Trainer(gspec, curnn, memory, lp, use_symmetries=use_symmetries)
    tconvert = time.perf_counter() - t0
    tloss = ttrain = teval = 0.0
    try:
        init_status = status = tr.learning_status()
        call("learning_started")
        nbatches = lp.max_batches_per_checkpoint
        if lp.min_checkpoints_per_epoch:
            nbatches = min(nbatches, tr.num_batches_total() // lp.min_checkpoints_per_epoch)
        best_evalr = None if ap is None else ap.update_threshold
        losses, checkpoints, replaced = [], [], False
        for k in range(1, lp.num_checkpoints + 1):
            call("updates_started", status)
            t1 = time.perf_counter()
            losses.append(tr.batch_updates(nbatches, seed=seed))
            t2 = time.perf_counter()
            tr.install_trained()
            status = tr.learning_status()
            t3 = time.perf_counter()
            call("updates_finished", status)
            ttrain += t2 - t1
            tloss += t3 - t2
            curnn = ResNet(gspec, curnn.hyper, params=tr.trained_params())      # get_trained_network
            if ap is None:
                bestnn, replaced = curnn.copy_(), True
            else:
                call("checkpoint_started")
                ev = compare_networks(gspec, curnn, bestnn, ap, handler if hasattr(handler, "checkpoint_game_played") else None, seed=seed + k)
                teval += ev.time
                success = ev.avgr >= best_evalr
                if success:
                    bestnn, best_evalr, replaced = curnn.copy_(), ev.avgr, True
                checkpoints.append(Checkpoint(k * nbatches, ev, status, success))
                call("checkpoint_finished", checkpoints[-1])
    finally:
        tr.close()
    report = LearningReport(tconvert, tloss, ttrain, teval, init_status,
                            np.concatenate(losses) if losses else np.zeros(0, np.float32), checkpoints, replaced)
    call("learning_finished", report)
    return curnn, bestnn, report


def train_iteration(gspec, curnn, bestnn, memory, self_play: SelfPlayParams, learning_params, arena_params, use_symmetries=True,
                    seed=1, handler=None):
    """One pass of train!'s loop body (training.jl:321-333): self_play_step! then learning_step!, both on the device."""
    sp = self_play_step_device(gspec, bestnn, self_play, memory, seed=seed)
    curnn, bestnn, lr = learning_step(gspec, curnn, bestnn, memory, learning_params, arena_params, use_symmetries, handler, seed)
    return curnn, bestnn, sp, lr
