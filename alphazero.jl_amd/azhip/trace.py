"""Trace (src/trace.jl:17-47) and its reconstruction from the engine's packed move records."""
import numpy as np


class Trace:
    def __init__(self, init_state):
        self.states = [init_state]
        self.policies = []
        self.rewards = []

    def push(self, pi, r, s):
        """Base.push!(t, π, r, s), trace.jl:34-38"""
        self.states.append(s)
        self.policies.append(pi)
        self.rewards.append(r)

    def valid(self):
        return len(self.policies) == len(self.rewards) == len(self.states) - 1

    def __len__(self):
        return len(self.rewards)

    def total_reward(self, gamma=1.0):
        return sum(gamma ** i * r for i, r in enumerate(self.rewards))


def policy_from_visits(N, mask):
    """MCTS.policy (src/mcts.jl:255-271): N_i / sum(N) over AVAILABLE actions, then renormalised."""
    n = np.asarray(N, dtype=np.int64)[np.asarray(mask, dtype=bool)]
    ntot = int(n.sum())
    pi = n.astype(np.float64) / float(ntot)
    s = 0.0
    for x in pi:              # sequential Float64 sum like Base.sum on a short vector
        s += float(x)
    return pi / s


def trace_from_records(game_rec, moves, num_actions, mask_fn):
    """game record + its move records -> Trace{State} with states as packed keys.
    mask_fn(key) -> availability mask of the state (needed because unavailable actions have N = 0 too)."""
    first, n = game_rec.first_move, game_rec.num_moves
    t = Trace((int(moves[first].key[0]), int(moves[first].key[1])))
    for k in range(n):
        m = moves[first + k]
        key = (int(m.key[0]), int(m.key[1]))
        pi = policy_from_visits(list(m.N[:num_actions]), mask_fn(key))
        nxt = (int(moves[first + k + 1].key[0]), int(moves[first + k + 1].key[1])) if k + 1 < n else \
            (int(game_rec.final_key[0]), int(game_rec.final_key[1]))
        t.push(pi, float(m.reward), nxt)
    return t
