"""Benchmark mirror (src/benchmark.jl): Duel / Single evaluations between standard players, on the device arena.

Players (benchmark.jl:124-192): Full(params) = MctsPlayer + the network, MctsRollouts(params) = MctsPlayer +
MCTS.RolloutOracle, NetworkOnly(τ) = PlayerWithTemperature(NetworkPlayer(nn), ConstSchedule(τ)).  MinMaxTS needs
the host-side minmax player (src/minmax.jl) and is not provided."""
import time
from dataclasses import dataclass

import numpy as np

from . import mcts as MCTS
from .arena import Evaluation, pit_players
from .network import copy as network_copy
from .params import ConstSchedule, MctsParams, SimParams
from .play import MctsPlayer, NetworkPlayer, PlayerWithTemperature, TwoPlayers


@dataclass
class Full:
    params: MctsParams
    name = "AlphaZero"

    def instantiate(self, gspec, nn):
        return MctsPlayer(gspec, nn, self.params)


@dataclass
class MctsRollouts:
    params: MctsParams

    @property
    def name(self):
        return "MCTS (%d rollouts)" % self.params.num_iters_per_turn

    def instantiate(self, gspec, nn):
        return MctsPlayer(gspec, MCTS.RolloutOracle(gspec), self.params)


@dataclass
class NetworkOnly:
    τ: float = 1.0
    name = "Network Only"

    def instantiate(self, gspec, nn):
        return PlayerWithTemperature(NetworkPlayer(nn), ConstSchedule(self.τ))


@dataclass
class Duel:
    """Benchmark.Duel(player, baseline, sim), benchmark.jl:55-61"""
    player: object
    baseline: object
    sim: SimParams

    @property
    def name(self):
        return "%s / %s" % (self.player.name, self.baseline.name)


def run(gspec, bestnn, duel: Duel, gamma=1.0, progress=None, device=0, seed=1):
    """Benchmark.run(env, duel, progress) (benchmark.jl:78-99) -> Report.Evaluation; `gamma` is
    env.params.self_play.mcts.gamma and must equal the players' (the engine discounts with the contender's)."""
    nn = network_copy(bestnn, on_gpu=duel.sim.use_gpu, test_mode=True)
    players = TwoPlayers(duel.player.instantiate(gspec, nn), duel.baseline.instantiate(gspec, nn))
    t0 = time.perf_counter()
    rewards, red, _ = pit_players(gspec, players, duel.sim, progress, device, seed)
    return Evaluation(duel.name, float(np.mean(rewards)), red, rewards, None, time.perf_counter() - t0)


@dataclass
class TernaryOutcomeStatistics:
    """benchmark.jl:105-122"""
    num_won: int
    num_draw: int
    num_lost: int

    @staticmethod
    def of(rewards):
        r = np.asarray(rewards)
        return TernaryOutcomeStatistics(int((r > 0).sum()), int((r == 0).sum()), int((r < 0).sum()))
