"""Engine: one MI355X self-play engine (a thin object wrapper over the C ABI, numpy in / numpy out)."""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib as L
from ._lib import EngineCfg, GameRec, MoveRec, Prof, SelfplayStats, TraceBuf, check, lib


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def default_cfg(**kw):
    cfg = EngineCfg()
    check(lib().az_engine_cfg_init(C.byref(cfg)))
    for k, v in kw.items():
        if k == "temperature":
            xs, ys = v
            if len(xs) != len(ys) or not 1 <= len(xs) <= L.SCHED_MAX:
                raise ValueError("temperature schedule needs 1..%d breakpoints" % L.SCHED_MAX)
            cfg.temperature_len = len(xs)
            for i, (x, y) in enumerate(zip(xs, ys)):
                cfg.temperature_xs[i] = int(x)
                cfg.temperature_ys[i] = float(y)
        else:
            if not hasattr(cfg, k):
                raise TypeError("unknown engine option %r" % k)
            setattr(cfg, k, v)
    return cfg


class Engine:
    def __init__(self, cfg=None, **kw):
        self.cfg = cfg if cfg is not None else default_cfg(**kw)
        h = C.c_void_p()
        check(lib().az_engine_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        n = C.c_int32()
        check(lib().az_game_num_actions(self.cfg.game, C.byref(n)))
        self.num_actions = n.value
        w, hh, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().az_game_state_dim(self.cfg.game, C.byref(w), C.byref(hh), C.byref(c)))
        self.state_dim = (w.value, hh.value, c.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().az_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- device / profiling -----------------------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, mem = C.c_int32(), C.c_int64()
        check(lib().az_device_info(self._h, name, 256, C.byref(ncu), C.byref(mem)))
        return name.value.decode(), ncu.value, mem.value

    def net_last_kernel(self):
        """name of the tower kernel that served the most recent network launch"""
        name = C.create_string_buffer(128)
        check(lib().az_net_last_kernel(self._h, name, 128))
        return name.value.decode()

    def prof_enable(self, on=True, classes=None):
        """classes: iterable of kernel class names (KERNEL_CLASSES) to time; None = all"""
        flag = 1 if on else 0
        if on and classes:
            mask = 0
            for c in classes:
                mask |= 1 << L.KERNEL_CLASSES.index(c)
            flag |= mask << 1
        check(lib().az_prof_enable(self._h, flag))

    def prof_reset(self):
        check(lib().az_prof_reset(self._h))

    def prof_get(self):
        p = Prof()
        check(lib().az_prof_get(self._h, C.byref(p)))
        return {name: {"launches": p.launches[i], "ms": p.ms[i], "units": p.units[i], "exec_units": p.exec_units[i]}
                for i, name in enumerate(L.KERNEL_CLASSES)}

    # ---- game plugin ------------------------------------------------------------------------
    def init_key(self):
        k = (C.c_uint64 * 2)()
        check(lib().az_game_init_key(self.cfg.game, k))
        return int(k[0]), int(k[1])

    def encode(self, keys):
        """GI.vectorize_state + actions_mask for an (n, 2) uint64 key array -> X (n, C, H, W), A (n, nA)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, 2)
        n = keys.shape[0]
        w, h, c = self.state_dim
        X = np.zeros((n, c, h, w), dtype=np.float32)
        A = np.zeros((n, self.num_actions), dtype=np.float32)
        check(lib().az_game_encode(self._h, _vp(keys), n, _vp(X), _vp(A)))
        return X, A

    def play(self, keys, actions):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, 2)
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        n = keys.shape[0]
        nxt = np.zeros((n, 2), dtype=np.uint64)
        term = np.zeros(n, dtype=np.int8)
        rew = np.zeros(n, dtype=np.float32)
        check(lib().az_game_play(self._h, _vp(keys), _vp(actions), n, _vp(nxt), _vp(term), _vp(rew)))
        return nxt, term.astype(bool), rew

    # ---- network ----------------------------------------------------------------------------
    def net_num_params(self):
        n = C.c_int64()
        check(lib().az_net_num_params(self._h, C.byref(n)))
        return n.value

    def net_set_params(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        check(lib().az_net_set_params(self._h, _vp(blob), blob.size))

    def net_get_params(self):
        out = np.zeros(self.net_num_params(), dtype=np.float32)
        check(lib().az_net_get_params(self._h, _vp(out), out.size))
        return out

    def net_forward(self, X, A):
        """Network.forward_normalized: X (N, C, H, W) [= Julia WHCN memory], A (N, nA) -> P, V, Pinv."""
        X = np.ascontiguousarray(X, dtype=np.float32)
        A = np.ascontiguousarray(A, dtype=np.float32)
        N = X.shape[0]
        P = np.zeros((N, self.num_actions), dtype=np.float32)
        V = np.zeros(N, dtype=np.float32)
        Pinv = np.zeros(N, dtype=np.float32)
        check(lib().az_net_forward(self._h, _vp(X), _vp(A), N, _vp(P), _vp(V), _vp(Pinv)))
        return P, V, Pinv

    def net_evaluate_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, 2)
        N = keys.shape[0]
        P = np.zeros((N, self.num_actions), dtype=np.float32)
        V = np.zeros(N, dtype=np.float32)
        check(lib().az_net_evaluate_keys(self._h, _vp(keys), N, _vp(P), _vp(V)))
        return P, V

    # ---- MCTS hooks ---------------------------------------------------------------------------
    def mcts_reset(self):
        check(lib().az_mcts_reset(self._h))

    def mcts_explore(self, root_keys, nsims, eta=None, game_ids=None, moves=None):
        keys = np.ascontiguousarray(root_keys, dtype=np.uint64).reshape(-1, 2)
        n = keys.shape[0]
        e = g = m = None
        if eta is not None:
            e = np.zeros((n, L.MAX_ACTIONS), dtype=np.float64)
            eta = np.asarray(eta, dtype=np.float64).reshape(n, -1)
            e[:, :eta.shape[1]] = eta
        if game_ids is not None:
            g = np.ascontiguousarray(game_ids, dtype=np.uint32)
        if moves is not None:
            m = np.ascontiguousarray(moves, dtype=np.uint32)
        check(lib().az_mcts_explore(self._h, _vp(keys), n, nsims, None if e is None else _vp(e),
                                    None if g is None else _vp(g), None if m is None else _vp(m)))

    def mcts_node_stats(self, slot, key):
        k = (C.c_uint64 * 2)(int(key[0]), int(key[1]))
        nA = self.num_actions
        N = np.zeros(nA, dtype=np.int32)
        W = np.zeros(nA, dtype=np.float64)
        P = np.zeros(nA, dtype=np.float32)
        V = C.c_float()
        mask = C.c_uint32()
        check(lib().az_mcts_node_stats(self._h, slot, k, _vp(N), _vp(W), _vp(P), C.byref(V), C.byref(mask)))
        return N, W, P, V.value, mask.value

    def mcts_counters(self, slot):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().az_mcts_counters(self._h, slot, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- self-play ------------------------------------------------------------------------------
    def _trace_buf(self, games_cap, moves_cap):
        games = (GameRec * max(games_cap, 1))()
        moves = (MoveRec * max(moves_cap, 1))()
        tb = TraceBuf()
        tb.games, tb.games_cap, tb.moves, tb.moves_cap = games, games_cap, moves, moves_cap
        return tb, games, moves

    def max_moves(self):
        if self.cfg.max_moves_per_game > 0:
            return self.cfg.max_moves_per_game
        return {L.GAME_CONNECT_FOUR: 42, L.GAME_TICTACTOE: 9, L.GAME_MANCALA: 256, L.GAME_GO9_PLANES: 1}[self.cfg.game]

    def selfplay_run(self, num_games, first_game_id=0, progress=None, device_only=False):
        """simulate(): returns (games, moves, ngames, nmoves, stats); games sorted by game id.
        device_only: the move records stay in HBM (MemoryBuffer.push_engine / Comm.gather_push read them there);
        only the game records come back (moves is None, nmoves 0)."""
        tb, games, moves = self._trace_buf(num_games, 0 if device_only else num_games * self.max_moves())
        if device_only:
            tb.moves, moves = None, None
        stats = SelfplayStats()
        cb = L.PROGRESS_CB((lambda user: progress()) if progress else (lambda user: None))
        check(lib().az_selfplay_run(self._h, num_games, first_game_id, C.byref(tb), cb, None, C.byref(stats)))
        return games, moves, tb.num_games, tb.num_moves, stats

    def selfplay_begin(self, num_games=-1, first_game_id=0):
        check(lib().az_selfplay_begin(self._h, num_games, first_game_id))

    def selfplay_step(self, nwaves):
        check(lib().az_selfplay_step(self._h, nwaves))

    def selfplay_collect(self, games_cap, moves_cap=None):
        tb, games, moves = self._trace_buf(games_cap, moves_cap or games_cap * self.max_moves())
        check(lib().az_selfplay_collect(self._h, C.byref(tb)))
        return games, moves, tb.num_games, tb.num_moves

    def selfplay_stats(self):
        s = SelfplayStats()
        check(lib().az_selfplay_get_stats(self._h, C.byref(s)))
        return s

    def selfplay_active(self):
        n = C.c_int32()
        check(lib().az_selfplay_active(self._h, C.byref(n)))
        return n.value

    def device_bytes(self):
        """device memory this engine holds (az_engine_device_bytes)"""
        n = C.c_int64()
        check(lib().az_engine_device_bytes(self._h, C.byref(n)))
        return n.value

    def release_phase(self):
        """drop the device-resident records of the last phase once they have been pushed / gathered"""
        check(lib().az_engine_release_phase(self._h))

    def selfplay_end(self):
        check(lib().az_selfplay_end(self._h))

    def selfplay_aborted(self):
        """ids of the games the current / last phase aborted (slot out of tree nodes or move records)"""
        n = C.c_int32()
        check(lib().az_selfplay_aborted(self._h, None, 0, C.byref(n)))
        ids = (C.c_int32 * max(n.value, 1))()
        check(lib().az_selfplay_aborted(self._h, ids, n.value, C.byref(n)))
        return [ids[i] for i in range(n.value)]

    # ---- arena ----------------------------------------------------------------------------------
    def arena_run(self, baseline, num_games, first_game_id=0, alternate_colors=False, progress=None, traces=True):
        """pit_networks: self = contender's engine, baseline = the other player's engine.
        Returns (games, moves, ngames, nmoves, rewards, redundancy)."""
        tb, games, moves = self._trace_buf(num_games, num_games * self.max_moves()) if traces else (None, None, None)
        rewards = np.zeros(num_games, dtype=np.float64)
        red = C.c_double()
        cb = L.PROGRESS_CB((lambda user: progress()) if progress else (lambda user: None))
        check(lib().az_arena_run(self._h, baseline._h, num_games, first_game_id, 1 if alternate_colors else 0,
                                 C.byref(tb) if traces else None, _vp(rewards), C.byref(red), cb, None))
        return games, moves, (tb.num_games if traces else 0), (tb.num_moves if traces else 0), rewards, red.value


# ---- engine cache --------------------------------------------------------------------------------------------------
# An engine owns its slots' node pools (10 GB at 4096 Connect-Four slots x 400 simulations): building one per
# self_play_step / pit_networks / Trainer call pays hipMalloc + table clears every time and can hold several at once.
# Engines are therefore kept by configuration and only their parameters are replaced between phases (az_net_set_params =
# Network.copy per phase, training.jl:278-279).  The key is everything az_engine_create looks at: a role (the two players
# of an arena get two engines), the az_engine_cfg bytes AND the environment overrides the library reads at creation
# (AZHIP_TOWER / AZHIP_HEADS / AZHIP_GRAPH / AZHIP_XCH_EPOCH0 -- an engine built under one override must not serve a call
# made under another).  Bounded by count and by device bytes (az_engine_device_bytes), least recently used first out; a
# cached engine does not keep its last phase's records (az_engine_release_phase at eviction time is too late: callers
# release them once pushed, see training.py).
CACHE_MAX = 4
CACHE_MAX_BYTES = int(float(os.environ.get("AZHIP_CACHE_GB", "96")) * (1 << 30))
CREATE_ENV = ("AZHIP_TOWER", "AZHIP_HEADS", "AZHIP_GRAPH", "AZHIP_XCH_EPOCH0", "AZHIP_VMM", "AZHIP_POOL_GB", "AZHIP_XCH_FAIL_AT", "AZHIP_POOLED_QUEUE",
              "AZHIP_HT_TAG_BITS", "AZHIP_HT_EPOCH0", "AZHIP_EVAL_CACHE", "AZHIP_EVAL_CACHE_LOG2", "AZHIP_VMM_KEYS", "AZHIP_TREE_ATOMIC", "AZHIP_TREE_SORT")   # (AZHIP_FREE_RUN / AZHIP_RUN_K / AZHIP_FR_ROUND are read per phase, not at creation)
_cache = {}


def _cache_bytes():
    return sum(e.device_bytes() for e in _cache.values() if e._h is not None)


def _evict():
    k = next(iter(_cache))
    if os.environ.get("AZHIP_TRACE_CACHE"):
        print("azhip cache: evicting '%s' (%.1f GB)" % (k[0], _cache[k].device_bytes() / 2**30), file=sys.stderr)
    _cache.pop(k).close()


def cached_engine(role="", **kw):
    import ctypes
    cfg = default_cfg(**kw)
    key = (role, bytes(ctypes.string_at(ctypes.addressof(cfg), ctypes.sizeof(cfg))), tuple(os.environ.get(k, "") for k in CREATE_ENV))
    e = _cache.pop(key, None)
    if e is None or e._h is None:
        while len(_cache) >= CACHE_MAX:
            _evict()
        e = Engine(cfg=cfg)
        need = e.device_bytes()
        while _cache and _cache_bytes() + need > CACHE_MAX_BYTES:
            _evict()
        if os.environ.get("AZHIP_TRACE_CACHE"):
            print("azhip cache: new engine '%s' %.1f GB (cache now %.1f GB in %d engines)" % (role, need / 2**30, (_cache_bytes() + need) / 2**30, len(_cache) + 1), file=sys.stderr)
    _cache[key] = e                                                 # most recently used last
    return e


def clear_engine_cache():
    while _cache:
        _cache.popitem()[1].close()
