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


# ---------------------------------------------------------------------------------------------------------
# MemoryBuffer on the device (memory.jl:34-60) and the Trainer's data (learning.jl:98-121)
import ctypes as _C

from . import _lib as _L


class MemoryBuffer:
    """MemoryBuffer(gspec, size): a circular buffer of TrainingSamples in HBM (az_memory_*).

    push_records takes the packed records of a self-play phase (Engine.selfplay_run / gather_records) and does
    push_trace! for every game on the device; get_experience() / last_batch() return device data sets."""

    def __init__(self, gspec, size, device=0):
        self.gspec = gspec
        h = _C.c_void_p()
        _L.check(_L.lib().az_memory_create(gspec.game_id, device, int(size), _C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _L.lib().az_memory_destroy(self._h)
            self._h = None

    __del__ = close

    def push_records(self, games, moves, ngames, nmoves, gamma):
        tb = _L.TraceBuf()
        tb.games, tb.games_cap, tb.num_games = games, ngames, ngames
        tb.moves, tb.moves_cap, tb.num_moves = moves, nmoves, nmoves
        _L.check(_L.lib().az_memory_push(self._h, _C.byref(tb), float(gamma)))

    def push_engine(self, engine, gamma):
        """push_trace! for every game of the engine's last bounded self-play phase, from its device-resident records"""
        _L.check(_L.lib().az_memory_push_engine(self._h, engine._h, float(gamma)))

    def push_samples(self, samples):
        """push!(mem.buf, e) for host TrainingSamples (or raw _lib.Sample records).

        az_sample.pi is indexed by FULL action index.  A host TrainingSample carries either a full-width π (what
        Dataset.samples() returns) or the reference's compact π over the AVAILABLE actions only (what push_trace builds
        from a Trace, memory.jl:74-87): the compact form is scattered through the state's action mask; any other length
        is an error."""
        n = len(samples)
        nA = self.gspec.num_actions()
        arr = (_L.Sample * max(n, 1))()
        for i, e in enumerate(samples):
            if isinstance(e, _L.Sample):
                _C.memmove(_C.byref(arr[i]), _C.byref(e), _C.sizeof(_L.Sample))
                continue
            arr[i].key[0], arr[i].key[1] = int(e.s[0]), int(e.s[1])
            pi = np.asarray(e.π, dtype=np.float64)
            if len(pi) != nA:
                mask = np.asarray(self.gspec.init((int(e.s[0]), int(e.s[1]))).actions_mask(), dtype=bool)
                if len(pi) != int(mask.sum()):
                    raise ValueError("sample %d: π has %d entries, the state has %d available actions of %d" % (i, len(pi), int(mask.sum()), nA))
                full = np.zeros(nA)
                full[mask] = pi
                pi = full
            for a in range(nA):
                arr[i].pi[a] = float(pi[a])
            arr[i].z, arr[i].t, arr[i].n = float(e.z), float(e.t), int(e.n)
        _L.check(_L.lib().az_memory_push_samples(self._h, arr, n))

    def _lens(self):
        a, b = _C.c_int64(), _C.c_int64()
        _L.check(_L.lib().az_memory_length(self._h, _C.byref(a), _C.byref(b)))
        return a.value, b.value

    def __len__(self):
        return self._lens()[0]

    def cur_batch_size(self):
        return self._lens()[1]

    def new_batch(self):
        _L.check(_L.lib().az_memory_new_batch(self._h))

    def empty(self):
        _L.check(_L.lib().az_memory_empty(self._h))

    def dataset(self, last_batch=False, use_symmetries=False, use_position_averaging=False, weighing_policy=_L.WEIGHT_CONSTANT):
        return Dataset(self, last_batch, use_symmetries, use_position_averaging, weighing_policy)

    def get_experience(self):
        """get_experience(mem) as TrainingSample list (host copy; the device path keeps working on Dataset)"""
        with self.dataset() as d:
            return d.samples()

    def last_batch(self):
        with self.dataset(last_batch=True) as d:
            return d.samples()


class Dataset:
    """The (W, X, A, P, V) data of a Trainer plus its samples, resident on the device (az_dataset_*)."""

    def __init__(self, mem, last_batch, use_symmetries, use_position_averaging, weighing_policy):
        self.gspec = mem.gspec
        h = _C.c_void_p()
        _L.check(_L.lib().az_dataset_create(mem._h, 1 if last_batch else 0, 1 if use_symmetries else 0,
                                            1 if use_position_averaging else 0, int(weighing_policy), _C.byref(h)))
        self._h = h
        info = _L.DatasetInfo()
        _L.check(_L.lib().az_dataset_get_info(h, _C.byref(info)))
        self.num_samples, self.sum_n, self.Wtot, self.Wmean, self.Hp = info.num_samples, info.sum_n, info.Wtot, info.Wmean, info.Hp

    def close(self):
        if getattr(self, "_h", None):
            _L.lib().az_dataset_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __len__(self):
        return self.num_samples

    def raw_samples(self):
        n = self.num_samples
        out = (_L.Sample * max(n, 1))()
        _L.check(_L.lib().az_dataset_read(self._h, 0, n, out, None, None, None, None, None))
        return out

    def samples(self):
        nA = self.gspec.num_actions()
        raw = self.raw_samples()
        return [TrainingSample((int(raw[i].key[0]), int(raw[i].key[1])), np.array(raw[i].pi[:nA]), raw[i].z, raw[i].t, int(raw[i].n))
                for i in range(self.num_samples)]

    def tensors(self):
        """convert_samples: (W, X, A, P, V) Float32 arrays, sample index first (= the reference's last dimension)"""
        n, nA = self.num_samples, self.gspec.num_actions()
        w, h, c = self.gspec.state_dim()
        W = np.zeros(n, dtype=np.float32); X = np.zeros((n, c, h, w), dtype=np.float32)
        A = np.zeros((n, nA), dtype=np.float32); P = np.zeros((n, nA), dtype=np.float32); V = np.zeros(n, dtype=np.float32)
        vp = lambda a: a.ctypes.data_as(_C.c_void_p)
        _L.check(_L.lib().az_dataset_read(self._h, 0, n, None, vp(W), vp(X), vp(A), vp(P), vp(V)))
        return W, X, A, P, V
