"""GameInterface mirror (src/game.jl:34-336) over the device twins of alphazero.jl_amd/csrc/games.h.

States are the packed 16-byte keys of include/azhip.h (tuples of two Python ints); every rule
evaluation (play!, game_terminated, white_reward, actions_mask, vectorize_state) runs on the GPU
through az_game_play / az_game_encode, so the invariants of src/scripts/test_game.jl exercise the same
code the search kernels use."""
import numpy as np

from . import _lib as L
from .engine import Engine

BLACK_BIT = 1 << 63


class GameSpec:
    game_id = None
    name = None
    _engines = {}

    def _eng(self):
        e = GameSpec._engines.get(self.game_id)
        if e is None:
            e = Engine(game=self.game_id, oracle=L.ORACLE_UNIFORM, num_workers=1, batch_size=1, num_iters_per_turn=2)
            GameSpec._engines[self.game_id] = e
        return e

    # --- static properties (game.jl:34-75, 243-336) ---
    def two_players(self):
        return True

    def actions(self):
        return list(range(1, self.num_actions() + 1))     # 1-based like the reference

    def num_actions(self):
        return {L.GAME_CONNECT_FOUR: 7, L.GAME_TICTACTOE: 9, L.GAME_MANCALA: 6}[self.game_id]

    def state_dim(self):
        return {L.GAME_CONNECT_FOUR: (7, 6, 3), L.GAME_TICTACTOE: (3, 3, 3), L.GAME_MANCALA: (14, 1, 5)}[self.game_id]

    def init(self, state=None):
        return GameEnv(self, state)

    def vectorize_state(self, state):
        X, _ = self._eng().encode([state])
        w, h, c = self.state_dim()
        return np.transpose(X[0], (2, 1, 0)).copy()       # (W, H, C) like the Julia array

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self))


class ConnectFourSpec(GameSpec):
    game_id, name = L.GAME_CONNECT_FOUR, "connect-four"

    def symmetries(self, state):
        """games/connect-four/game.jl:247-257: column mirror, sigma = 7..1"""
        a, b = state

        def mirror(x):
            flag, x = x & BLACK_BIT, x & ~BLACK_BIT
            out = 0
            for c in range(7):
                out |= ((x >> (7 * c)) & 0x7f) << (7 * (6 - c))
            return out | flag
        return [((mirror(a), mirror(b)), list(range(7, 0, -1)))]


class TicTacToeSpec(GameSpec):
    game_id, name = L.GAME_TICTACTOE, "tictactoe"


class MancalaSpec(GameSpec):
    game_id, name = L.GAME_MANCALA, "mancala"


SPECS = {"connect-four": ConnectFourSpec, "tictactoe": TicTacToeSpec, "mancala": MancalaSpec}


class GameEnv:
    """AbstractGameEnv (game.jl:77-175)."""

    def __init__(self, spec, state=None):
        self._spec = spec
        if state is None:
            self._state = spec._eng().init_key()
            self._terminated, self._reward = False, 0.0
        else:
            self.set_state(state)

    def spec(self):
        return self._spec

    def set_state(self, state):
        state = (int(state[0]), int(state[1]))
        nxt, term, rew = self._spec._eng().play([state], [-1])
        self._state, self._terminated, self._reward = state, bool(term[0]), float(rew[0])

    def clone(self):
        g = GameEnv.__new__(GameEnv)
        g._spec, g._state, g._terminated, g._reward = self._spec, self._state, self._terminated, self._reward
        return g

    def current_state(self):
        return self._state

    def game_terminated(self):
        return self._terminated

    def white_playing(self):
        return not (self._state[0] & BLACK_BIT)

    def white_reward(self):
        return self._reward

    def actions_mask(self):
        _, A = self._spec._eng().encode([self._state])
        return A[0] > 0

    def available_actions(self):
        return [a for a, m in zip(self._spec.actions(), self.actions_mask()) if m]

    def play(self, action):
        """GI.play!(game, action) with a 1-based action like the reference."""
        nxt, term, rew = self._spec._eng().play([self._state], [int(action) - 1])
        self._state = (int(nxt[0][0]), int(nxt[0][1]))
        self._terminated, self._reward = bool(term[0]), float(rew[0])

    def vectorize_state(self):
        return self._spec.vectorize_state(self._state)

    def apply_random_symmetry(self, rng):
        """GI.apply_random_symmetry! (game.jl:329-336)."""
        syms = self._spec.symmetries(self._state) if hasattr(self._spec, "symmetries") else []
        assert syms, "no symmetries were declared for this game"
        symstate, _ = syms[int(rng.integers(len(syms)))]
        self.set_state(symstate)
