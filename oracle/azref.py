"""ctypes binding of the CPU oracle (oracle/libazref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py.  The product package (alphazero.jl_amd/azhip) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libazref.so")

C4, TTT, MANCALA = 0, 1, 2
GO9 = 3           # tensor geometry only (9 x 9 x 4 planes, 82 actions): the network restatement takes any dimensions
ORACLE_UNIFORM, ORACLE_HASH, ORACLE_NET, ORACLE_ROLLOUT = 0, 1, 2, 3
ORACLE_EXTERNAL = 4   # replay mode: oracle(state) answers supplied by the caller (azr_sim_step / azr_sim_feed)
AMAX = 9
CELLS = 42


def build(force=False):
    src = os.path.join(HERE, "azref.c")
    hdr = os.path.join(HERE, "ref_numerics.h")            # the oracle's own numerics (not the product's include/az_numerics.h)
    if (force or not os.path.exists(LIB)
            or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-s", "-C", HERE, "libazref.so"])
    return LIB


class State(C.Structure):
    _fields_ = [("cells", C.c_uint8 * CELLS), ("curplayer", C.c_uint8), ("pad", C.c_uint8)]


class Env(C.Structure):
    _fields_ = [("game", C.c_int), ("s", State), ("finished", C.c_uint8), ("winner", C.c_uint8),
                ("amask", C.c_uint8 * AMAX)]


class SimParams(C.Structure):
    _fields_ = [("gamma", C.c_double), ("cpuct", C.c_double), ("noise_eps", C.c_double),
                ("noise_alpha", C.c_double), ("prior_temperature", C.c_double),
                ("num_iters_per_turn", C.c_int), ("temp_len", C.c_int), ("temp_xs", C.c_int * 8),
                ("temp_ys", C.c_double * 8), ("num_games", C.c_int), ("num_workers", C.c_int),
                ("reset_every", C.c_int), ("seed", C.c_uint64), ("game", C.c_int),
                ("oracle_kind", C.c_int), ("nblocks", C.c_int), ("F", C.c_int), ("npf", C.c_int),
                ("nvf", C.c_int), ("blob", C.POINTER(C.c_float)), ("first_game_id", C.c_int),
                ("flip_probability", C.c_double)]


class MoveRec(C.Structure):
    _fields_ = [("key", C.c_uint64 * 2), ("N", C.c_int32 * (AMAX + 1)), ("action", C.c_int32),
                ("reward", C.c_float)]


class GameRec(C.Structure):
    _fields_ = [("game_id", C.c_int32), ("slot", C.c_int32), ("num_moves", C.c_int32),
                ("first_move", C.c_int32), ("nodes", C.c_int64), ("total_simulations", C.c_int64),
                ("total_nodes_traversed", C.c_int64), ("final_key", C.c_uint64 * 2)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.azr_white_reward.restype = C.c_double
        L.azr_net_num_params.restype = C.c_size_t
        L.azr_mcts_new.restype = C.c_void_p
        L.azr_mcts_new.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
        L.azr_mcts_set_net.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.azr_mcts_reset.argtypes = [C.c_void_p]
        L.azr_mcts_free.argtypes = [C.c_void_p]
        for f in ("azr_mcts_num_nodes", "azr_mcts_total_simulations", "azr_mcts_total_nodes_traversed",
                  "azr_mcts_oracle_calls"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.azr_mcts_explore.argtypes = [C.c_void_p, C.POINTER(Env), C.c_int, C.c_void_p, C.c_uint64,
                                       C.c_uint32, C.c_uint32]
        L.azr_mcts_root_stats.argtypes = [C.c_void_p, C.POINTER(State), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        L.azr_mcts_policy.argtypes = [C.c_void_p, C.POINTER(Env), C.c_void_p, C.c_void_p]
        L.azr_plschedule.restype = C.c_double
        L.azr_simulate.restype = C.c_int64
        L.azr_simulate.argtypes = [C.POINTER(SimParams), C.c_void_p, C.c_void_p, C.c_int64]
        L.azr_expf.restype = C.c_float
        L.azr_c4_solve.restype = C.c_int
        L.azr_c4_solve.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]
        L.azr_expf.argtypes = [C.c_float]
        L.azr_tanhf.restype = C.c_float
        L.azr_tanhf.argtypes = [C.c_float]
        for f in ("azr_log", "azr_exp"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double]
        L.azr_pow.restype = C.c_double
        L.azr_pow.argtypes = [C.c_double, C.c_double]
        L.azr_dirichlet.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_void_p]
        L.azr_move_uniform.restype = C.c_float
        L.azr_move_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.azr_rand_categorical.argtypes = [C.c_void_p, C.c_int, C.c_float]
        for f in ("azr_sizeof_state", "azr_sizeof_env", "azr_sizeof_move_rec", "azr_sizeof_game_rec",
                  "azr_sizeof_sim_params"):
            getattr(L, f).restype = C.c_size_t
        assert L.azr_sizeof_state() == C.sizeof(State)
        assert L.azr_sizeof_env() == C.sizeof(Env)
        assert L.azr_sizeof_move_rec() == C.sizeof(MoveRec)
        assert L.azr_sizeof_game_rec() == C.sizeof(GameRec)
        assert L.azr_sizeof_sim_params() == C.sizeof(SimParams)
        assert L.azr_numerics_selftest() == 0, "oracle built with fp contraction / without fma"
        _lib = L
    return _lib


NUM_ACTIONS = {C4: 7, TTT: 9, MANCALA: 6, GO9: 82}
DIMS = {C4: (7, 6, 3), TTT: (3, 3, 3), MANCALA: (14, 1, 5), GO9: (9, 9, 4)}


class Game:
    """GameInterface view of one oracle game environment (src/game.jl)."""

    def __init__(self, game, state=None):
        self.game = game
        self.env = Env()
        if state is None:
            lib().azr_init(C.byref(self.env), game)
        else:
            lib().azr_init_state(C.byref(self.env), game, C.byref(state))

    def clone(self):
        g = Game.__new__(Game)
        g.game = self.game
        g.env = Env.from_buffer_copy(self.env)
        return g

    def play(self, a):
        lib().azr_play(C.byref(self.env), int(a))

    def terminated(self):
        return bool(lib().azr_terminated(C.byref(self.env)))

    def white_playing(self):
        return bool(lib().azr_white_playing(C.byref(self.env)))

    def white_reward(self):
        return lib().azr_white_reward(C.byref(self.env))

    def actions_mask(self):
        return np.array(self.env.amask[:NUM_ACTIONS[self.game]], dtype=bool)

    def available_actions(self):
        return np.nonzero(self.actions_mask())[0]

    def state(self):
        return State.from_buffer_copy(self.env.s)

    def key(self):
        k = (C.c_uint64 * 2)()
        lib().azr_pack_key(self.game, C.byref(self.env.s), k)
        return int(k[0]), int(k[1])

    def vectorize(self):
        w, h, c = DIMS[self.game]
        out = np.zeros(w * h * c, dtype=np.float32)
        lib().azr_vectorize_state(self.game, C.byref(self.env.s), out.ctypes.data_as(C.c_void_p))
        return out  # Flux memory order: W fastest, then H, then C


def unpack_key(game, key):
    st = State()
    k = (C.c_uint64 * 2)(key[0], key[1])
    lib().azr_unpack_key(game, k, C.byref(st))
    return st


class Mcts:
    """MCTS.Env (src/mcts.jl:124-151)."""

    def __init__(self, game, oracle=ORACLE_UNIFORM, gamma=1.0, cpuct=1.0, noise_eps=0.0, noise_alpha=1.0,
                 prior_temperature=1.0, net=None):
        self.game = game
        self.h = lib().azr_mcts_new(game, oracle, gamma, cpuct, noise_eps, noise_alpha, prior_temperature)
        self._blob = None
        if net is not None:
            nblocks, F, npf, nvf, blob = net
            self._blob = np.ascontiguousarray(blob, dtype=np.float32)
            lib().azr_mcts_set_net(self.h, nblocks, F, npf, nvf, self._blob.ctypes.data_as(C.c_void_p))

    def __del__(self):
        if getattr(self, "h", None):
            lib().azr_mcts_free(self.h)
            self.h = None

    def explore(self, g, nsims, eta=None, seed=0, game_id=0, move=0):
        e = None
        if eta is not None:
            e = np.ascontiguousarray(eta, dtype=np.float64)
        lib().azr_mcts_explore(self.h, C.byref(g.env), nsims, None if e is None else e.ctypes.data_as(C.c_void_p),
                               seed, game_id, move)

    def root_stats(self, g):
        N = np.zeros(AMAX, dtype=np.int64)
        W = np.zeros(AMAX, dtype=np.float64)
        P = np.zeros(AMAX, dtype=np.float32)
        V = C.c_float()
        st = g.state()
        n = lib().azr_mcts_root_stats(self.h, C.byref(st), N.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p),
                                      P.ctypes.data_as(C.c_void_p), C.byref(V))
        if n < 0:
            raise KeyError("state not in tree")
        return N[:n], W[:n], P[:n], V.value

    def policy(self, g):
        acts = (C.c_int * AMAX)()
        pi = np.zeros(AMAX, dtype=np.float64)
        n = lib().azr_mcts_policy(self.h, C.byref(g.env), acts, pi.ctypes.data_as(C.c_void_p))
        if n < 0:
            raise KeyError("MCTS.explore! must be called before MCTS.policy")
        return list(acts[:n]), pi[:n]

    def reset(self):
        lib().azr_mcts_reset(self.h)

    num_nodes = property(lambda s: lib().azr_mcts_num_nodes(s.h))
    total_simulations = property(lambda s: lib().azr_mcts_total_simulations(s.h))
    total_nodes_traversed = property(lambda s: lib().azr_mcts_total_nodes_traversed(s.h))
    oracle_calls = property(lambda s: lib().azr_mcts_oracle_calls(s.h))


def net_num_params(game, nblocks, F, npf, nvf):
    w, h, c = DIMS[game]
    return lib().azr_net_num_params(w, h, c, NUM_ACTIONS[game], nblocks, F, npf, nvf)


def net_forward_normalized(game, hp, blob, X, A):
    """Network.forward_normalized on a batch: X (N, C, H, W) == Julia WHCN memory, A (N, nA)."""
    nblocks, F, npf, nvf = hp
    w, h, c = DIMS[game]
    nA = NUM_ACTIONS[game]
    X = np.ascontiguousarray(X, dtype=np.float32)
    A = np.ascontiguousarray(A, dtype=np.float32)
    N = X.shape[0]
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    assert blob.size == net_num_params(game, nblocks, F, npf, nvf)
    P = np.zeros((N, nA), dtype=np.float32)
    V = np.zeros(N, dtype=np.float32)
    Pinv = np.zeros(N, dtype=np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib().azr_net_forward_normalized(w, h, c, nA, nblocks, F, npf, nvf, vp(blob), vp(X), vp(A), N, vp(P), vp(V), vp(Pinv))
    return P, V, Pinv


def _sim_params(game, oracle, num_games, num_workers, nsims, gamma=1.0, cpuct=1.0, noise_eps=0.0, noise_alpha=1.0,
                prior_temperature=1.0, temp_xs=(0,), temp_ys=(1.0,), reset_every=1, seed=1, net=None, first_game_id=0, flip_probability=0.0):
    p = SimParams()
    p.gamma, p.cpuct, p.noise_eps, p.noise_alpha, p.prior_temperature = gamma, cpuct, noise_eps, noise_alpha, prior_temperature
    p.num_iters_per_turn = nsims
    p.temp_len = len(temp_xs)
    for i, (x, y) in enumerate(zip(temp_xs, temp_ys)):
        p.temp_xs[i] = x
        p.temp_ys[i] = y
    p.num_games, p.num_workers, p.reset_every, p.seed = num_games, num_workers, reset_every or 0, seed
    p.game, p.oracle_kind = game, oracle
    p.first_game_id = first_game_id
    p.flip_probability = flip_probability          # azr_simulate only; azr_arena takes its own argument
    blob = None
    if net is not None:
        p.nblocks, p.F, p.npf, p.nvf, blob = net[0], net[1], net[2], net[3], np.ascontiguousarray(net[4], dtype=np.float32)
        p.blob = blob.ctypes.data_as(C.POINTER(C.c_float))
    return p, blob


def assignment_of(device_games, num_games, first_game_id=0):
    """worker_of[i] = the worker (slot) that played game first_game_id + i, from the game records a device phase returned
    (az_game_rec.slot): the outcome of the reference's id race (util.jl:181-188) that phase took"""
    w = np.full(num_games, -1, dtype=np.int32)
    for i in range(num_games):
        g = device_games[i]
        w[g.game_id - first_game_id] = g.slot
    assert (w >= 0).all(), "assignment_of: a game is missing from the records"
    return w


def _worker_of(assignment, num_games):
    if assignment is None:
        return None, None
    w = np.ascontiguousarray(assignment, dtype=np.int32)
    assert w.shape == (num_games,)
    return w, w.ctypes.data_as(C.c_void_p)


def simulate(game, oracle, num_games, num_workers, nsims, assignment=None, **kw):
    """simulate (src/simulations.jl:207-244) in lock-step; returns (games, moves) record arrays.
    assignment: None = game ids in finishing order (ties by worker index), or worker_of[num_games] -- the outcome of the reference's
    id race to replay (assignment_of: what a free-running device phase reports)."""
    p, blob = _sim_params(game, oracle, num_games, num_workers, nsims, **kw)
    games = (GameRec * num_games)()
    cap = num_games * 512
    moves = (MoveRec * cap)()
    w, wp = _worker_of(assignment, num_games)
    L = lib()
    L.azr_simulate_assigned.restype = C.c_int64
    L.azr_simulate_assigned.argtypes = [C.POINTER(SimParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    nm = L.azr_simulate_assigned(C.byref(p), games, moves, cap, wp)
    if nm < 0:
        raise ValueError("simulate: not an assignment the reference's worker pool could produce")
    return games, moves, nm


class Evals:
    """The evaluation table of replay mode (state key -> P, V), shareable between several replays of one network."""

    def __init__(self, log2cap=22):
        lib().azr_evals_new.restype = C.c_void_p
        self.h = C.c_void_p(lib().azr_evals_new(int(log2cap)))

    def counters(self):
        out = (C.c_int64 * 4)()
        lib().azr_evals_counters(self.h, out)
        return dict(asked=out[0], answered=out[1], wipes=out[2], entries=out[3])

    def close(self):
        if self.h:
            lib().azr_evals_free(self.h)
            self.h = None

    __del__ = close


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))


def usable_cpus():
    """CPUs this process can really use: the affinity mask, cut down to the cgroup's CPU quota (cpu.max: the GPU box shows 256 hardware
    threads and grants 16 CPUs' worth of time; 64 spinning threads then take 71 s for a replay, 128 take 103 s, 256 do not finish)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
        except Exception:
            pass
    return n


def replay(game, evaluate, num_games, num_workers, nsims, evals=None, threads=None, assignment=None, **kw):
    """REPLAY MODE (SURVEY.md §7 hard part 3): `simulate` -- the same lock-step loop, the same tree code -- with every oracle answer
    supplied by the caller, e.g. by the device network behind az_net_evaluate_keys.  The oracle's trees then run on the other
    evaluator's numbers, at CPU-tree speed on `threads` host threads (default: as many as the process may really use -- usable_cpus -- with 16 workers or more each).
    evaluate: a callable  keys uint64[n, 2] -> (P float32[n, A] by full action index, V float32[n]),  or a pair (address, user) of a C
    function  int f(void* user, const uint64_t* keys, int32_t n, float* P, float* V)  (0 = ok) that is called without Python in between.
    Returns (games, moves, num_moves, info); info: steps (rounds of evaluation + move rounds), evaluated (states sent to the
    evaluator: each distinct state once while the table holds it), oracle_calls (evaluations the trees consumed: what a direct run
    computes), rounds (move rounds)."""
    L = lib()
    p, blob = _sim_params(game, ORACLE_EXTERNAL, num_games, num_workers, nsims, **kw)
    games = (GameRec * num_games)()
    cap = num_games * 512
    moves = (MoveRec * cap)()
    L.azr_sim_new.restype = C.c_void_p
    L.azr_sim_new.argtypes = [C.POINTER(SimParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.azr_sim_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
    L.azr_sim_free.argtypes = [C.c_void_p]
    L.azr_sim_num_moves.restype = C.c_int64
    L.azr_sim_num_moves.argtypes = [C.c_void_p]
    L.azr_sim_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.azr_sim_times.argtypes = [C.c_void_p, C.c_void_p]
    own = evals is None
    G = min(num_workers, num_games)
    if own:
        evals = Evals(max(16, int(np.ceil(np.log2(16 * max(1, G))))))
    nA = NUM_ACTIONS[game]
    failure = []
    if callable(evaluate):
        def _cb(user, keys, n, P, V):
            try:
                k = np.ctypeslib.as_array(keys, shape=(n, 2))
                Pn, Vn = evaluate(k)
                np.ctypeslib.as_array(P, shape=(n, nA))[:] = np.asarray(Pn, dtype=np.float32).reshape(n, nA)
                np.ctypeslib.as_array(V, shape=(n,))[:] = np.asarray(Vn, dtype=np.float32).reshape(n)
                return 0
            except BaseException as ex:                  # an exception must not unwind through the C loop: stop it, re-raise below
                failure.append(ex)
                return -1
        cb = EVAL_FN(_cb)
        fn, user = C.cast(cb, C.c_void_p), None
    else:
        fn, user = C.c_void_p(evaluate[0]), C.c_void_p(evaluate[1])
    if threads is None:
        # the workers spin at a barrier between the rounds: never more threads than CPUs this process may run on (AZ_REPLAY_THREADS overrides)
        threads = int(os.environ.get("AZ_REPLAY_THREADS", 0)) or max(1, min(usable_cpus(), (G + 15) // 16))
    h = C.c_void_p(L.azr_sim_new(C.byref(p), games, moves, cap, evals.h))
    nev = C.c_int64(0)
    w, wp = _worker_of(assignment, num_games)
    if w is not None:
        L.azr_sim_set_assignment.argtypes = [C.c_void_p, C.c_void_p]
        if L.azr_sim_set_assignment(h, wp) != 0:
            L.azr_sim_free(h)
            raise ValueError("replay: not an assignment the reference's worker pool could produce")
    try:
        rc = L.azr_sim_run(h, fn, user, int(threads), C.byref(nev))
        if failure:
            raise failure[0]
        if rc != 0:
            raise RuntimeError("replay: the evaluator returned status %d" % rc)
        nm = L.azr_sim_num_moves(h)
        out = (C.c_int64 * 3)()
        L.azr_sim_counters(h, out)
        tm = (C.c_double * 4)()
        L.azr_sim_times(h, tm)
        info = dict(rounds=out[0], steps=out[1], oracle_calls=out[2], evaluated=int(nev.value), threads=int(threads),
                    seconds=dict(turns=round(tm[0], 3), waiting_for_threads=round(tm[1], 3), serial=round(tm[2], 3), evaluator=round(tm[3], 3)))
    finally:
        L.azr_sim_free(h)
        if own:
            evals.close()
    return games, moves, nm, info


def hash_oracle_keys(game, keys):
    """the synthetic hash oracle for a batch of state keys: (P [n, A] full width, V [n])"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, 2)
    n = keys.shape[0]
    P = np.zeros((n, NUM_ACTIONS[game]), dtype=np.float32); V = np.zeros(n, dtype=np.float32)
    lib().azr_hash_oracle_keys(game, keys.ctypes.data_as(C.c_void_p), C.c_int64(n), P.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
    return P, V


def net_evaluate_keys(game, hp, blob, keys):
    """Network.evaluate_batch (network.jl:308-315) of the oracle's fp32 network for a batch of state keys"""
    keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, 2)
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    n = keys.shape[0]
    P = np.zeros((n, NUM_ACTIONS[game]), dtype=np.float32); V = np.zeros(n, dtype=np.float32)
    lib().azr_net_evaluate_keys(game, hp[0], hp[1], hp[2], hp[3], blob.ctypes.data_as(C.c_void_p), keys.ctypes.data_as(C.c_void_p),
                                C.c_int64(n), P.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
    return P, V


def arena(game, num_games, num_workers, contender, baseline, alternate_colors=False, flip_probability=0.0,
          reset_every=1, seed=1, first_game_id=0, assignment=None):
    """pit_networks (src/training.jl:130-144).  contender / baseline: dicts of _sim_params keywords
    (oracle, nsims, cpuct, noise_eps, ..., temp_xs, temp_ys, net, seed).
    Returns (games, moves, num_moves, rewards, redundancy)."""
    ps, keep = [], []
    for pl in (contender, baseline):
        kw = dict(pl)
        oracle, nsims = kw.pop("oracle"), kw.pop("nsims")
        kw.setdefault("seed", seed)
        p, blob = _sim_params(game, oracle, num_games, num_workers, nsims, reset_every=reset_every, **kw)
        ps.append(p)
        keep.append(blob)
    games = (GameRec * num_games)()
    cap = num_games * 512
    moves = (MoveRec * cap)()
    rewards = np.zeros(num_games, dtype=np.float64)
    red = C.c_double()
    w, wp = _worker_of(assignment, num_games)       # the outcome of the id race to replay (assignment_of), or None
    lib().azr_arena_assigned.restype = C.c_int64
    lib().azr_arena_assigned.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    nm = lib().azr_arena_assigned(C.byref(ps[0]), C.byref(ps[1]), int(alternate_colors), float(flip_probability), first_game_id,
                                  games, moves, cap, rewards.ctypes.data_as(C.c_void_p), C.byref(red), wp)
    if nm < 0:
        raise ValueError("arena: not an assignment the reference's worker pool could produce")
    return games, moves, nm, rewards, red.value


def symmetry(game, state_cells, curplayer, k):
    """GI.symmetries(gspec, state)[k] on raw cells (for tests)."""
    st, out = State(), State()
    for i, c in enumerate(state_cells):
        st.cells[i] = c
    st.curplayer = curplayer
    lib().azr_symmetry(game, C.byref(st), k, C.byref(out))
    return list(out.cells), out.curplayer


# ---------------------------------------------------------------- replay memory + learning status
class Sample(C.Structure):
    """TrainingSample (src/memory.jl:20-26), pi by full action index"""
    _fields_ = [("key", C.c_uint64 * 2), ("pi", C.c_double * AMAX), ("z", C.c_double), ("t", C.c_double), ("n", C.c_int64)]


class LearningStatus(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("L", "Lp", "Lv", "Lreg", "Linv", "Hp", "Hpnet", "Wmean")]


def samples_from_trace(game, moves, first, n, gamma=1.0):
    """push_trace! (memory.jl:74-87) of one game's move records -> list-like (Sample * n)"""
    out = (Sample * max(n, 1))()
    arr = (MoveRec * max(n, 1))(*[moves[first + k] for k in range(n)])
    lib().azr_samples_from_trace(game, arr, n, C.c_double(gamma), out)
    return out


def _arr(samples):
    n = len(samples)
    a = (Sample * max(n, 1))()
    for i, s in enumerate(samples):
        C.memmove(C.byref(a[i]), C.byref(s), C.sizeof(Sample))
    return a, n


def augment_with_symmetries(game, samples):
    a, n = _arr(samples)
    nsym = lib().azr_num_symmetries(game)
    out = (Sample * max(n * (1 + nsym), 1))()
    lib().azr_augment_with_symmetries.restype = C.c_int64
    m = lib().azr_augment_with_symmetries(game, a, C.c_int64(n), out)
    return [out[i] for i in range(m)]


def merge_by_state(game, samples):
    a, n = _arr(samples)
    out = (Sample * max(n, 1))()
    lib().azr_merge_by_state.restype = C.c_int64
    m = lib().azr_merge_by_state(game, a, C.c_int64(n), out)
    return [out[i] for i in range(m)]


def convert_samples(game, policy, samples):
    a, n = _arr(samples)
    w, h, c = DIMS[game]
    nA = NUM_ACTIONS[game]
    W = np.zeros(n, dtype=np.float32); X = np.zeros((n, c, h, w), dtype=np.float32)
    A = np.zeros((n, nA), dtype=np.float32); P = np.zeros((n, nA), dtype=np.float32); V = np.zeros(n, dtype=np.float32)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    lib().azr_convert_samples(game, policy, a, C.c_int64(n), vp(W), vp(X), vp(A), vp(P), vp(V))
    return W, X, A, P, V


def learning_status(game, hp, blob, data, l2=1e-4, nonvalidity_penalty=1.0, rewards_renormalization=1.0, batch=1024):
    """learning_status(tr) (learning.jl:158-181) for converted data (W, X, A, P, V); hp = (nblocks, F, npf, nvf)"""
    W, X, A, P, V = [np.ascontiguousarray(x, dtype=np.float32) for x in data]
    w, h, c = DIMS[game]
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    out = LearningStatus()
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    lib().azr_learning_status(w, h, c, NUM_ACTIONS[game], hp[0], hp[1], hp[2], hp[3], vp(blob), vp(W), vp(X), vp(A), vp(P), vp(V),
                              C.c_int64(len(W)), C.c_double(l2), C.c_double(nonvalidity_penalty), C.c_double(rewards_renormalization),
                              C.c_int64(batch), C.byref(out))
    return out


def c4_solve(moves, node_limit=20_000_000):
    """exact Connect-Four value (Pons' score convention) of the position after `moves` (0-based columns), by negamax over
    the oracle's own rules; returns (score, nodes).  98 = node limit, 99 = illegal / finished position.  Test infrastructure."""
    import numpy as _np
    m = _np.asarray(moves, dtype=_np.int32)
    nodes = C.c_longlong(0)
    sc = lib().azr_c4_solve(m.ctypes.data_as(C.c_void_p), len(m), int(node_limit), C.byref(nodes))
    return sc, nodes.value
