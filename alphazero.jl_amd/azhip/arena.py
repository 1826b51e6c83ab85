"""Arena mirror: pit_networks / compare_networks (src/training.jl:130-172) over the device engines.

The reference pits TwoPlayers(MctsPlayer(contender), MctsPlayer(baseline)) through `simulate` with a
two-network inference server (simulations.jl:70-99).  Here each network gets its own engine (its own trees
per worker) and az_arena_run steps both: every ply the workers are split by the player to move and each engine
explores its share.  SimParams.flip_probability and alternate_colors are honoured (play.jl:305-307,
simulations.jl:221-223)."""
import time
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from .engine import cached_engine
from .mcts import oracle_kind
from .network import copy as network_copy
from .params import ArenaParams, ConstSchedule, MctsParams, engine_options
from .play import MctsPlayer, NetworkPlayer, PlayerWithTemperature, TwoPlayers
from .trace import Trace


@dataclass
class Evaluation:
    """Report.Evaluation, report.jl:73-80"""
    legend: str
    avgr: float
    redundancy: float
    rewards: np.ndarray
    baseline_rewards: Optional[np.ndarray]
    time: float
    workers: Optional[np.ndarray] = None      # not in the reference's report: the worker that played each game (az_game_rec.slot) -- with
                                              # reset_every != 1 which games share a tree is an outcome of the id race (util.jl:181-188)


def _engine(gspec, player, sim, device, seed, role):
    """one engine per player: MctsPlayer (any device oracle) or PlayerWithTemperature(NetworkPlayer(nn), schedule)
    -- the latter is an engine without search (num_iters_per_turn = 0)."""
    if isinstance(player, MctsPlayer):
        oracle, params = player.oracle, player.params
    elif isinstance(player, PlayerWithTemperature) and isinstance(player.player, NetworkPlayer):
        oracle = player.player.network
        params = MctsParams(num_iters_per_turn=0, dirichlet_noise_ϵ=0.0, dirichlet_noise_α=1.0, temperature=player.temperature)
    elif isinstance(player, NetworkPlayer):
        oracle = player.network
        params = MctsParams(num_iters_per_turn=0, dirichlet_noise_ϵ=0.0, dirichlet_noise_α=1.0, temperature=ConstSchedule(1.0))
    else:
        raise TypeError("the device arena pits MctsPlayers and NetworkPlayers (MinMax / Human players stay on the host)")
    kind = oracle_kind(oracle)
    kw = engine_options(params, sim, seed=seed, arena=True)
    if kind == L.ORACLE_RESNET:
        kw.update(oracle.engine_options())
    e = cached_engine(role, game=gspec.game_id, oracle=kind, device=device, **kw)   # kept across checkpoints (engine.py)
    if kind == L.ORACLE_RESNET:
        e.net_set_params(oracle.params())
    return e


def arena_traces(games, moves, ngames, num_actions):
    """records -> [(Trace, colors_flipped is NOT stored: derive from the game id)].  Trace.states are the
    un-flipped states like the reference's; policies are full-width visit frequencies of the state the
    player actually saw (the reference pairs them with the un-flipped state too, play.jl:305-313)."""
    out = []
    for i in range(ngames):
        g = games[i]
        t = Trace((int(moves[g.first_move].key[0]), int(moves[g.first_move].key[1])))
        for k in range(g.num_moves):
            m = moves[g.first_move + k]
            n = np.array(m.N[:num_actions], dtype=np.int32)
            n = n.view(np.float32).astype(np.float64) if m.N[L.MAX_ACTIONS] & 0x100 else n.astype(np.float64)   # NetworkPlayer: f32 policy bits
            nxt = (int(moves[g.first_move + k + 1].key[0]), int(moves[g.first_move + k + 1].key[1])) \
                if k + 1 < g.num_moves else (int(g.final_key[0]), int(g.final_key[1]))
            t.push(n / n.sum(), float(m.reward), nxt)
        out.append(t)
    return out


def pit_players(gspec, players: TwoPlayers, sim, game_simulated=None, device=0, seed=1, first_game_id=0):
    """simulate(Simulator(make_oracles, record_trace) do ... TwoPlayers(white, black)) + rewards_and_redundancy:
    returns (rewards from players.white's side, redundancy, traces)."""
    if not gspec.two_players():
        raise ValueError("pit_players needs a two-player game")
    ec, eb = _engine(gspec, players.white, sim, device, seed, "arena-white"), _engine(gspec, players.black, sim, device, seed, "arena-black")
    games, moves, ng, nm, rewards, red = ec.arena_run(eb, sim.num_games, first_game_id=first_game_id,
                                                      alternate_colors=sim.alternate_colors, progress=game_simulated)
    pit_players.last_workers = np.array([games[i].slot for i in range(ng)], dtype=np.int32)   # by game id (az_arena_run sorts the records)
    return rewards, red, arena_traces(games, moves, ng, gspec.num_actions())


def pit_networks(gspec, contender, baseline, params: ArenaParams, handler=None, device=0, seed=1):
    """training.jl:130-144 -> (rewards vector, redundancy)"""
    white = MctsPlayer(gspec, network_copy(contender, on_gpu=params.sim.use_gpu, test_mode=True), params.mcts)
    black = MctsPlayer(gspec, network_copy(baseline, on_gpu=params.sim.use_gpu, test_mode=True), params.mcts)
    cb = (lambda: handler.checkpoint_game_played()) if handler is not None else None
    rewards, red, _ = pit_players(gspec, TwoPlayers(white, black), params.sim, cb, device, seed)
    return rewards, red


def compare_networks(gspec, contender, baseline, params: ArenaParams, handler=None, device=0, seed=1):
    """training.jl:159-172 (two-player branch; the device games are all two-player)"""
    t0 = time.perf_counter()
    rewards, red = pit_networks(gspec, contender, baseline, params, handler, device, seed)
    return Evaluation("Most recent NN versus best NN so far", float(np.mean(rewards)), red, rewards, None,
                      time.perf_counter() - t0, getattr(pit_players, "last_workers", None))
