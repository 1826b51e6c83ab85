"""MCTS module mirror (src/mcts.jl): Env / explore! / policy / reset! on ONE device slot.

This is the fine-grained seam (SURVEY.md §8b seam 5) used by parity tests and by the host-stepped
play_game; the production self-play path is `simulations.simulate`, which keeps the whole loop on the GPU."""
import numpy as np

from . import _lib as L
from .engine import Engine
from .trace import policy_from_visits


class RandomOracle:
    """MCTS.RandomOracle (mcts.jl:62-72): uniform prior, V = 0."""
    kind = L.ORACLE_UNIFORM

    def __init__(self, gspec=None):
        self.gspec = gspec


class RolloutOracle:
    """MCTS.RolloutOracle(gspec) (mcts.jl:35-60): uniform prior, V = outcome of one uniformly random playout
    (gamma = 1, the value Benchmark.MctsRollouts builds it with)."""
    kind = L.ORACLE_ROLLOUT

    def __init__(self, gspec=None, gamma=1.0):
        if gamma != 1.0:
            raise ValueError("the device rollout oracle implements gamma = 1")
        self.gspec = gspec


class HashOracle:
    """Synthetic exact oracle (priors/value derived from the state key); for parity tests."""
    kind = L.ORACLE_HASH

    def __init__(self, gspec=None):
        self.gspec = gspec


def oracle_kind(oracle):
    from .network import ResNet
    if isinstance(oracle, ResNet):
        return L.ORACLE_RESNET
    if isinstance(oracle, (RandomOracle, HashOracle, RolloutOracle)):
        return oracle.kind
    raise TypeError("oracle must be a ResNet, MCTS.RandomOracle, MCTS.RolloutOracle or MCTS.HashOracle: arbitrary host callables "
                    "cannot run inside the device search (use the reference's CPU MCTS with ResNet.evaluate_batch)")


class Env:
    """MCTS.Env(gspec, oracle; gamma, cpuct, noise_ϵ, noise_α, prior_temperature), mcts.jl:124-151"""

    def __init__(self, gspec, oracle, gamma=1.0, cpuct=1.0, noise_ϵ=0.0, noise_α=1.0, prior_temperature=1.0,
                 seed=1, max_nodes=1 << 16, device=0):
        self.gspec, self.oracle = gspec, oracle
        self.gamma, self.cpuct, self.noise_ϵ, self.noise_α, self.prior_temperature = gamma, cpuct, noise_ϵ, noise_α, prior_temperature
        kw = {}
        kind = oracle_kind(oracle)
        if kind == L.ORACLE_RESNET:
            kw = oracle.engine_options()
        self._e = Engine(game=gspec.game_id, oracle=kind, device=device, num_workers=1, batch_size=1,
                         num_iters_per_turn=2, gamma=gamma, cpuct=cpuct, dirichlet_noise_eps=noise_ϵ,
                         dirichlet_noise_alpha=noise_α, prior_temperature=prior_temperature, seed=seed,
                         max_nodes_per_slot=max_nodes, reset_every=0, **kw)
        if kind == L.ORACLE_RESNET:
            self._e.net_set_params(oracle.params())
        self._calls = 0

    def explore(self, game, nsims, eta=None):
        """MCTS.explore!(env, game, nsims), mcts.jl:239-245; eta (by available-action rank) overrides the draw."""
        e = None
        if eta is not None:
            mask = game.actions_mask()
            e = np.zeros((1, L.MAX_ACTIONS))
            e[0, np.nonzero(mask)[0]] = np.asarray(eta, dtype=np.float64)
        self._e.mcts_explore([game.current_state()], nsims, eta=e, game_ids=[0], moves=[self._calls])
        self._calls += 1

    def policy(self, game):
        """MCTS.policy(env, game), mcts.jl:255-271"""
        N, W, P, V, mask = self._e.mcts_node_stats(0, game.current_state())
        m = [(mask >> a) & 1 for a in range(self.gspec.num_actions())]
        return [a for a, ok in zip(self.gspec.actions(), m) if ok], policy_from_visits(N, m)

    def tree_stats(self, state):
        """env.tree[state]: (N, W, P) by available-action rank and Vest (used by the explorer UI)."""
        N, W, P, V, mask = self._e.mcts_node_stats(0, state)
        m = np.array([(mask >> a) & 1 for a in range(self.gspec.num_actions())], dtype=bool)
        return N[m], W[m], P[m], V

    def reset(self):
        """MCTS.reset!, mcts.jl:278-281"""
        self._e.mcts_reset()

    @property
    def total_simulations(self):
        return self._e.mcts_counters(0)[0]

    @property
    def total_nodes_traversed(self):
        return self._e.mcts_counters(0)[1]

    def num_nodes(self):
        return self._e.mcts_counters(0)[2]

    def average_exploration_depth(self):
        """mcts.jl:293-296"""
        ts, tt, _ = self._e.mcts_counters(0)
        return 0 if ts == 0 else tt / ts

    def approximate_memory_footprint(self):
        """mcts.jl:319-321 with the per-node size of the DEVICE record (memory_footprint_per_node)"""
        return memory_footprint_per_node(self.gspec) * self.num_nodes()


def memory_footprint_per_node(gspec):
    nA = gspec.num_actions()
    hb = 2 if nA <= 8 else 4
    rec = ((18 * nA + hb - 1) // hb * hb + hb + 31) // 32 * 32      # NodeL (csrc/tree.h): 16 B per action + child links
    return rec + 32 + 12            # node record + side record (key, Vest) + its share of the 1.5x hash table (8 B entries)
