"""Player abstraction of the path (src/play.jl:156-214,298-315): MctsPlayer, think, play_game."""
import time

import numpy as np

from . import mcts as MCTS
from .params import MctsParams
from .trace import Trace


def apply_temperature(pi, tau):
    """Util.apply_temperature, src/util.jl:98-110"""
    pi = np.asarray(pi, dtype=np.float64)
    if tau == 1:
        return pi
    if tau == 0:
        res = np.zeros_like(pi)
        res[int(np.argmax(pi))] = 1
        return res
    res = pi ** (1.0 / tau)
    return res / res.sum()


def fix_probvec(pi):
    """Util.fix_probvec, src/util.jl:68-81"""
    p = np.asarray(pi, dtype=np.float32)
    s = np.float32(0)
    for x in p:
        s = np.float32(s + x)
    if not (abs(float(s) - 1.0) <= 0.00034526698 * max(abs(float(s)), 1.0)):
        p = np.full(len(p), np.float32(1) / np.float32(len(p)), dtype=np.float32) if s == 0 else (p / s).astype(np.float32)
    return p


def rand_categorical(pi, u):
    """Util.rand_categorical (util.jl:87-90) with the Float32 uniform supplied"""
    p = fix_probvec(pi)
    cp, i = p[0], 0
    while cp <= u and i < len(p) - 1:
        i += 1
        cp = np.float32(cp + p[i])
    return i


class MctsPlayer:
    """MctsPlayer(gspec, oracle, params::MctsParams; timeout=nothing), play.jl:156-214"""

    def __init__(self, gspec, oracle, params: MctsParams, timeout=None, seed=1):
        assert params.num_iters_per_turn > 0
        if timeout is not None and not timeout >= 0:
            raise ValueError("timeout must be None or a number of seconds >= 0")
        self.gspec, self.oracle, self.params, self.timeout = gspec, oracle, params, timeout
        self.niters, self.τ, self.seed = params.num_iters_per_turn, params.temperature, seed
        self._mcts = None

    @property
    def mcts(self):
        if self._mcts is None:
            p = self.params
            self._mcts = MCTS.Env(self.gspec, self.oracle, gamma=p.gamma, cpuct=p.cpuct, noise_ϵ=p.dirichlet_noise_ϵ,
                                  noise_α=p.dirichlet_noise_α, prior_temperature=p.prior_temperature, seed=self.seed)
        return self._mcts

    def think(self, game):
        if self.timeout is None:                      # fixed number of MCTS simulations, play.jl:198
            self.mcts.explore(game, self.niters)
        else:                                         # whole explore! calls (each with its own noise draw) until the time is up,
            start = time.monotonic()                  # play.jl:199-204; wall-clock driven, so not reproducible by construction
            while time.monotonic() - start < self.timeout:
                self.mcts.explore(game, self.niters)
        return self.mcts.policy(game)

    def player_temperature(self, game, turn):
        return self.τ[turn]

    def reset_player(self):
        if self._mcts is not None:
            self._mcts.reset()


class NetworkPlayer:
    """NetworkPlayer(nn) (play.jl:226-235): plays the network's policy directly, no search."""

    def __init__(self, network):
        self.network = network

    def think(self, game):
        P, _ = self.network.evaluate(game.current_state())
        return game.available_actions(), np.asarray(P)

    def player_temperature(self, game, turn):
        return 1.0

    def reset_player(self):
        pass


class PlayerWithTemperature:
    """PlayerWithTemperature(player, temperature) (play.jl:136-150): overrides the move-selection temperature."""

    def __init__(self, player, temperature):
        self.player, self.temperature = player, temperature

    def think(self, game):
        return self.player.think(game)

    def player_temperature(self, game, turn):
        return self.temperature[turn]

    def reset_player(self):
        self.player.reset_player()


class TwoPlayers:
    """TwoPlayers(white, black), play.jl:248-282: behaves as `white` when white is to play, else as `black`."""

    def __init__(self, white, black):
        self.white, self.black = white, black

    def _cur(self, game):
        return self.white if game.white_playing() else self.black

    def think(self, game):
        return self._cur(game).think(game)

    def player_temperature(self, game, turn):
        return self._cur(game).player_temperature(game, turn)

    def reset_player(self):
        self.white.reset_player()
        self.black.reset_player()


def flipped_colors(p: TwoPlayers):
    """play.jl:253"""
    return TwoPlayers(p.black, p.white)


def play_game(gspec, player, flip_probability=0.0, rng=None):
    """play_game, play.jl:298-315 -- host-stepped (one device search per move).  Self-play at scale goes
    through simulations.simulate instead, which runs this loop for thousands of games on the GPU."""
    rng = rng or np.random.Generator(np.random.Philox(1))
    game = gspec.init()
    trace = Trace(game.current_state())
    while True:
        if game.game_terminated():
            return trace
        if flip_probability != 0.0 and rng.random() < flip_probability:      # play.jl:305-307
            game.apply_random_symmetry(rng)
        actions, pi_target = player.think(game)
        tau = player.player_temperature(game, len(trace))
        pi_sample = apply_temperature(pi_target, tau)
        a = actions[rand_categorical(pi_sample, np.float32(rng.random(dtype=np.float32)))]
        game.play(a)
        trace.push(pi_target, game.white_reward(), game.current_state())
