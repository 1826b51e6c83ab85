"""Simulation runtime mirror (src/simulations.jl:179-290, src/training.jl:269-273).

`simulate` hands the whole self-play phase to the engine (az_selfplay_run): the reference's
num_workers tasks + inference server (batchifier.jl) become num_workers device slots searched in
lock-step.  `simulate_distributed` shards the games over the ranks of a torch.distributed group (one
process per GPU) and gathers the packed records with all_gather (RCCL over xGMI on the GPU box)."""
import numpy as np

from . import _lib as L
from .engine import Engine, cached_engine
from .mcts import memory_footprint_per_node, oracle_kind
from .params import SimParams, engine_options
from .play import MctsPlayer
from .trace import trace_from_records


class Simulator:
    """Simulator(make_player, make_oracles, measure), simulations.jl:179-183"""

    def __init__(self, make_player, make_oracles, measure):
        self.make_player, self.make_oracles, self.measure = make_player, make_oracles, measure


def record_trace(t, cf, p):
    return {"trace": t, "colors_flipped": cf}


class _WorkerView:
    """What `measure(trace, colors_flipped, player)` may read of the worker's player at game end."""

    class _M:
        def __init__(self, gspec, rec):
            self._g, self._r = gspec, rec

        def approximate_memory_footprint(self):
            return memory_footprint_per_node(self._g) * int(self._r.nodes)

        def average_exploration_depth(self):
            return 0 if self._r.total_simulations == 0 else self._r.total_nodes_traversed / self._r.total_simulations

    def __init__(self, gspec, rec):
        self.mcts = _WorkerView._M(gspec, rec)
        self.worker = int(rec.slot)      # which worker played the game (az_game_rec.slot): the outcome of the id race, util.jl:181-188


def self_play_measurements(trace, _, player):
    """training.jl:269-273; + `worker`: the worker that played the game (not in the reference's report: its assignment is a race)"""
    return {"trace": trace, "mem": player.mcts.approximate_memory_footprint(),
            "edepth": player.mcts.average_exploration_depth(), "worker": getattr(player, "worker", -1)}


def _engine_for(gspec, player, p, device, seed):
    if not isinstance(player, MctsPlayer):
        raise TypeError("the device path simulates MctsPlayer self-play only")
    kind = oracle_kind(player.oracle)
    kw = engine_options(player.params, p, seed=seed)
    if kind == L.ORACLE_RESNET:
        kw.update(player.oracle.engine_options())
    e = cached_engine("selfplay", game=gspec.game_id, oracle=kind, device=device, **kw)   # kept across phases (engine.py)
    if kind == L.ORACLE_RESNET:
        e.net_set_params(player.oracle.params())
    return e


def run_local(simulator, gspec, p: SimParams, first_game_id=0, game_simulated=None, device=0, seed=1, device_only=False):
    """The body of simulate(): returns the engine's raw (games, moves, ngames, nmoves, stats) and the (cached, live) engine.
    device_only: the move records stay in the engine's HBM phase buffer (moves is None)."""
    oracles = simulator.make_oracles()
    player = simulator.make_player(oracles)
    e = _engine_for(gspec, player, p, device, seed)
    return e.selfplay_run(p.num_games, first_game_id=first_game_id, progress=game_simulated, device_only=device_only) + (e,)


def results_from_records(simulator, gspec, games, moves, ngames, mask_engine=None):
    nA = gspec.num_actions()
    eng = mask_engine or gspec._eng()
    keys = np.array([[moves[g.first_move + k].key[0], moves[g.first_move + k].key[1]]
                     for g in (games[i] for i in range(ngames)) for k in range(g.num_moves)], dtype=np.uint64).reshape(-1, 2)
    _, A = eng.encode(keys) if len(keys) else (None, np.zeros((0, nA)))
    masks = {(int(k[0]), int(k[1])): A[i] > 0 for i, k in enumerate(keys)}
    out = []
    for i in range(ngames):
        g = games[i]
        trace = trace_from_records(g, moves, nA, lambda key: masks[key])
        out.append(simulator.measure(trace, False, _WorkerView(gspec, g)))
    return out


def simulate(simulator, gspec, p: SimParams, game_simulated=None, first_game_id=0, device=0, seed=1):
    """simulate(::Simulator, gspec, ::SimParams; game_simulated), simulations.jl:207-244"""
    games, moves, ng, nm, stats, _ = run_local(simulator, gspec, p, first_game_id, game_simulated, device, seed)
    return results_from_records(simulator, gspec, games, moves, ng)


def shard_games(num_games, world, rank):
    """simulate_distributed's split (simulations.jl:268-278): divrem, remainder to the first worker.
    Returns (first_game_id, count) so that global ids are contiguous in rank order."""
    num_each, rem = divmod(num_games, world)
    if num_each < 1:
        raise ValueError("num_games must be >= number of ranks")
    counts = [num_each + (rem if r == 0 else 0) for r in range(world)]
    return sum(counts[:rank]), counts[rank]


def gather_records(games_arr, moves_arr, group=None):
    """All-gather two numpy structured arrays (game records, move records) over the process group:
    counts first, then one padded byte buffer per array.  Game `first_move` indices are rebased.
    With NCCL (= RCCL on ROCm) the buffers are staged on the current GPU."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    cnt = torch.tensor([len(games_arr), len(moves_arr)], dtype=torch.int64, device=dev)
    cnts = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    cnts = [c.cpu().tolist() for c in cnts]

    def gather(arr, idx):
        isz = arr.dtype.itemsize
        mx = max(c[idx] for c in cnts) * isz
        buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=dev)
        raw = torch.from_numpy(np.frombuffer(arr.tobytes(), dtype=np.uint8).copy())
        buf[:raw.numel()] = raw.to(dev)
        outs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf, group=group)
        return [np.frombuffer(o.cpu().numpy().tobytes()[:cnts[r][idx] * isz], dtype=arr.dtype).copy() for r, o in enumerate(outs)]

    gs, ms = gather(games_arr, 0), gather(moves_arr, 1)
    off = 0
    for g, m in zip(gs, ms):
        g["first_move"] += off
        off += len(m)
    return np.concatenate(gs), np.concatenate(ms)


def records_to_numpy(games, moves, ngames, nmoves):
    g = np.frombuffer(bytes(games), dtype=GAME_DTYPE)[:ngames].copy()
    m = np.frombuffer(bytes(moves), dtype=MOVE_DTYPE)[:nmoves].copy()
    return g, m


GAME_DTYPE = np.dtype([("game_id", "<i4"), ("slot", "<i4"), ("num_moves", "<i4"), ("first_move", "<i4"), ("nodes", "<i8"),
                       ("total_simulations", "<i8"), ("total_nodes_traversed", "<i8"), ("final_key", "<u8", (2,))])
MOVE_DTYPE = np.dtype([("key", "<u8", (2,)), ("N", "<i4", (L.MAX_ACTIONS + 1,)), ("action", "<i4"), ("reward", "<f4")])


class _Rec:
    """attribute view of one numpy record (so trace_from_records works on gathered arrays)"""

    def __init__(self, r):
        self._r = r

    def __getattr__(self, k):
        return self._r[k]


def simulate_distributed(simulator, gspec, p: SimParams, game_simulated=None, group=None, seed=1):
    """simulate_distributed, simulations.jl:252-290: one rank per GPU, games split by shard_games, global
    game ids (so results do not depend on the number of ranks), records all-gathered to every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return simulate(simulator, gspec, p, game_simulated=game_simulated, seed=seed)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count = shard_games(p.num_games, world, rank)
    lp = SimParams(**{**p.__dict__, "num_games": count})
    device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    games, moves, ng, nm, stats, _ = run_local(simulator, gspec, lp, first, game_simulated, device, seed)
    g, m = records_to_numpy(games, moves, ng, nm)
    G, M = gather_records(g, m, group)
    order = np.argsort(G["game_id"], kind="stable")
    return results_from_records(simulator, gspec, [_Rec(G[i]) for i in order], [_Rec(x) for x in M], len(order))
