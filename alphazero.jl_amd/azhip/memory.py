"""TrainingSample / push_trace! (src/memory.jl:20-26,74-87) and the packed sample records that cross
ranks after a self-play phase."""
from dataclasses import dataclass

import numpy as np

BLACK_BIT = 1 << 63


@dataclass
class TrainingSample:
    s: tuple
    π: np.ndarray
    z: float
    t: float
    n: int = 1


def push_trace(mem, trace, gamma):
    """push_trace!(mem, trace, gamma): appends one TrainingSample per position, last position first."""
    n = len(trace)
    wr = 0.0
    for i in reversed(range(n)):
        wr = gamma * wr + trace.rewards[i]
        s = trace.states[i]
        wp = not (s[0] & BLACK_BIT)
        z = wr if wp else -wr
        mem.append(TrainingSample(s, trace.policies[i], z, float(n - i), 1))
    return n


def sample_dtype(num_actions):
    return np.dtype([("key", "<u8", (2,)), ("N", "<i4", (num_actions,)), ("z", "<f8"), ("t", "<f8"), ("game", "<i4"), ("n", "<i4")])


def pack_samples(games, moves, ngames, num_actions, gamma):
    """Engine records -> flat sample array (state key, visit counts, z, t): the on-wire record of the
    trace gather.  z/t follow push_trace! exactly (az_push_trace)."""
    total = sum(games[i].num_moves for i in range(ngames))
    out = np.zeros(total, dtype=sample_dtype(num_actions))
    k = 0
    for i in range(ngames):
        g = games[i]
        wr = 0.0
        for j in reversed(range(g.num_moves)):
            m = moves[g.first_move + j]
            wr = gamma * wr + float(m.reward)
            wp = not (int(m.key[0]) & BLACK_BIT)
            r = out[k + j]
            r["key"] = (m.key[0], m.key[1])
            r["N"] = list(m.N[:num_actions])
            r["z"] = wr if wp else -wr
            r["t"] = float(g.num_moves - j)
            r["game"] = g.game_id
            r["n"] = 1
        k += g.num_moves
    return out
