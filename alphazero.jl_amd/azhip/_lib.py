"""ctypes binding of libazhip.so (the C ABI of include/azhip.h).

The HIP extension is the product: if the shared library is missing this module raises at import
time -- there is NO CPU fallback (the oracle under oracle/ is test infrastructure only).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "csrc"))
LIB_PATH = os.environ.get("AZHIP_LIB", os.path.join(CSRC, "libazhip.so"))   # override: A/B builds

ABI_VERSION = 4              # include/azhip.h AZ_ABI_VERSION: struct layouts below are version 4's
REPLACEMENT_GAME_BIT = 0x40000000
AZ_OK, AZ_ERR_BAD_ARG, AZ_ERR_CAPACITY, AZ_ERR_HIP, AZ_ERR_STATE = 0, -1, -2, -3, -4
GAME_CONNECT_FOUR, GAME_TICTACTOE, GAME_MANCALA = 0, 1, 2
GAME_GO9_PLANES = 3          # network-only geometry (9, 9, 4), 82 actions: az_net_forward for host-stepped 9x9 Go
ORACLE_UNIFORM, ORACLE_HASH, ORACLE_RESNET, ORACLE_ROLLOUT = 0, 1, 2, 3
MAX_ACTIONS = 9
SCHED_MAX = 8
PROF_NUM = 8
KERNEL_CLASSES = ("select", "compact", "tower", "heads", "expand", "move", "synth", "start")


class AzError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("azhip status %d: %s" % (status, msg))
        self.status = status


class EngineCfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("game", C.c_int32), ("oracle", C.c_int32),
        ("gamma", C.c_double), ("cpuct", C.c_double), ("dirichlet_noise_eps", C.c_double),
        ("dirichlet_noise_alpha", C.c_double), ("prior_temperature", C.c_double),
        ("num_iters_per_turn", C.c_int32), ("temperature_len", C.c_int32),
        ("temperature_xs", C.c_int32 * SCHED_MAX), ("temperature_ys", C.c_double * SCHED_MAX),
        ("num_workers", C.c_int32), ("batch_size", C.c_int32), ("reset_every", C.c_int32),
        ("fill_batches", C.c_int32), ("flip_probability", C.c_double), ("seed", C.c_uint64),
        ("max_nodes_per_slot", C.c_int32), ("max_moves_per_game", C.c_int32),
        ("num_blocks", C.c_int32), ("num_filters", C.c_int32),
        ("num_policy_head_filters", C.c_int32), ("num_value_head_filters", C.c_int32), ("net_bf16", C.c_int32), ("lock_step", C.c_int32),
    ]


class MoveRec(C.Structure):
    _fields_ = [("key", C.c_uint64 * 2), ("N", C.c_int32 * (MAX_ACTIONS + 1)), ("action", C.c_int32),
                ("reward", C.c_float)]


class GameRec(C.Structure):
    _fields_ = [("game_id", C.c_int32), ("slot", C.c_int32), ("num_moves", C.c_int32),
                ("first_move", C.c_int32), ("nodes", C.c_int64), ("total_simulations", C.c_int64),
                ("total_nodes_traversed", C.c_int64), ("final_key", C.c_uint64 * 2)]


class TraceBuf(C.Structure):
    _fields_ = [("games", C.POINTER(GameRec)), ("games_cap", C.c_int64), ("num_games", C.c_int64),
                ("moves", C.POINTER(MoveRec)), ("moves_cap", C.c_int64), ("num_moves", C.c_int64)]


class SelfplayStats(C.Structure):
    _fields_ = [("simulations", C.c_int64), ("nodes_traversed", C.c_int64), ("leaf_evals", C.c_int64),
                ("moves", C.c_int64), ("games", C.c_int64), ("waves", C.c_int64), ("seconds", C.c_double),
                ("aborted_games", C.c_int64), ("tower_fallbacks", C.c_int64), ("evals_reused", C.c_int64), ("slot_launches", C.c_int64)]


class Prof(C.Structure):
    _fields_ = [("launches", C.c_int64 * PROF_NUM), ("ms", C.c_double * PROF_NUM), ("units", C.c_int64 * PROF_NUM),
                ("exec_units", C.c_double * PROF_NUM)]


class Sample(C.Structure):
    """az_sample = TrainingSample (memory.jl:20-26), pi by full action index"""
    _fields_ = [("key", C.c_uint64 * 2), ("pi", C.c_double * MAX_ACTIONS), ("z", C.c_double), ("t", C.c_double),
                ("n", C.c_int64)]


class DatasetInfo(C.Structure):
    _fields_ = [("num_samples", C.c_int64), ("sum_n", C.c_int64), ("Wtot", C.c_double), ("Wmean", C.c_float),
                ("Hp", C.c_float)]


class LearningStatusRec(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("L", "Lp", "Lv", "Lreg", "Linv", "Hp", "Hpnet")]


WEIGHT_CONSTANT, WEIGHT_LOG, WEIGHT_LINEAR = 0, 1, 2
OPT_ADAM, OPT_CYCLIC_NESTEROV = 0, 1


class TrainCfg(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("optimiser", C.c_int32), ("lr", C.c_float), ("lr_base", C.c_float),
                ("lr_high", C.c_float), ("lr_low", C.c_float), ("momentum_low", C.c_float), ("momentum_high", C.c_float),
                ("l2_regularization", C.c_double), ("nonvalidity_penalty", C.c_double), ("rewards_renormalization", C.c_double),
                ("batch_size", C.c_int32), ("batch_norm_momentum", C.c_float), ("seed", C.c_uint64)]

PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p)

# every symbol include/azhip.h declares: name -> argtypes (restype is int unless noted)
_VP, _I32, _I64, _U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
class GatherStats(C.Structure):
    _fields_ = [("games", C.c_int64), ("moves", C.c_int64), ("bytes", C.c_int64), ("gather_ms", C.c_double), ("total_ms", C.c_double),
                ("ranks", C.c_int64), ("total_simulations", C.c_int64), ("total_nodes_traversed", C.c_int64), ("max_nodes", C.c_int64),
                ("mean_game_depth", C.c_double), ("replaced_games", C.c_int64)]


AZ_ERR_COMM = -5
COMM_ID_BYTES = 128

SYMBOLS = {
    "az_last_error": None,
    "az_abi_version": [],
    "az_abi_struct_size": [_I32],
    "az_engine_cfg_init": [C.POINTER(EngineCfg)],
    "az_engine_create": [C.POINTER(EngineCfg), C.POINTER(_VP)],
    "az_engine_destroy": [_VP],
    "az_game_num_actions": [C.c_int, C.POINTER(_I32)],
    "az_game_state_dim": [C.c_int, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)],
    "az_game_init_key": [C.c_int, C.POINTER(C.c_uint64)],
    "az_game_encode": [_VP, _VP, _I32, _VP, _VP],
    "az_game_play": [_VP, _VP, _VP, _I32, _VP, _VP, _VP],
    "az_net_num_params": [_VP, C.POINTER(_I64)],
    "az_net_set_params": [_VP, _VP, _I64],
    "az_net_get_params": [_VP, _VP, _I64],
    "az_net_forward": [_VP, _VP, _VP, _I32, _VP, _VP, _VP],
    "az_net_evaluate_keys": [_VP, _VP, _I32, _VP, _VP],
    "az_mcts_reset": [_VP],
    "az_mcts_explore": [_VP, _VP, _I32, _I32, _VP, _VP, _VP],
    "az_mcts_node_stats": [_VP, _I32, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_mcts_counters": [_VP, _I32, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)],
    "az_selfplay_run": [_VP, _I32, _I32, C.POINTER(TraceBuf), PROGRESS_CB, _VP, C.POINTER(SelfplayStats)],
    "az_selfplay_begin": [_VP, _I32, _I32],
    "az_selfplay_step": [_VP, _I32],
    "az_selfplay_collect": [_VP, C.POINTER(TraceBuf)],
    "az_selfplay_get_stats": [_VP, C.POINTER(SelfplayStats)],
    "az_selfplay_active": [_VP, C.POINTER(_I32)],
    "az_selfplay_end": [_VP],
    "az_selfplay_aborted": [_VP, _VP, _I32, C.POINTER(_I32)],
    "az_arena_run": [_VP, _VP, _I32, _I32, _I32, C.POINTER(TraceBuf), _VP, C.POINTER(C.c_double), PROGRESS_CB, _VP],
    "az_push_trace": [_VP, _I32, C.c_double, _VP, _VP],
    "az_memory_create": [_I32, _I32, _I64, C.POINTER(_VP)],
    "az_memory_destroy": [_VP],
    "az_memory_push": [_VP, C.POINTER(TraceBuf), C.c_double],
    "az_memory_push_samples": [_VP, _VP, _I64],
    "az_memory_length": [_VP, C.POINTER(_I64), C.POINTER(_I64)],
    "az_memory_new_batch": [_VP],
    "az_memory_empty": [_VP],
    "az_dataset_create": [_VP, _I32, _I32, _I32, _I32, C.POINTER(_VP)],
    "az_dataset_destroy": [_VP],
    "az_dataset_get_info": [_VP, C.POINTER(DatasetInfo)],
    "az_dataset_read": [_VP, _I64, _I64, _VP, _VP, _VP, _VP, _VP, _VP],
    "az_learning_status": [_VP, _VP, C.c_double, C.c_double, C.c_double, _I64, C.POINTER(LearningStatusRec)],
    "az_train_cfg_init": [C.POINTER(TrainCfg)],
    "az_trainer_create": [_VP, _VP, C.POINTER(TrainCfg), C.POINTER(_VP)],
    "az_trainer_destroy": [_VP],
    "az_trainer_batch_updates": [_VP, _I32, _VP],
    "az_trainer_get_params": [_VP, _VP, _I64],
    "az_trainer_gradients": [_VP, _VP, C.POINTER(C.c_float), _VP, _VP, _I64],
    "az_prof_enable": [_VP, _I32],
    "az_prof_get": [_VP, C.POINTER(Prof)],
    "az_prof_reset": [_VP],
    "az_device_info": [_VP, C.c_char_p, _I32, C.POINTER(_I32), C.POINTER(_I64)],
    "az_net_last_kernel": [_VP, C.c_char_p, _I32],
    "az_memory_push_engine": [_VP, _VP, C.c_double],
    "az_comm_unique_id": [_VP],
    "az_comm_init": [_I32, _I32, _I32, _VP, C.POINTER(_VP)],
    "az_comm_destroy": [_VP],
    "az_comm_version": [C.POINTER(_I32), C.c_char_p, _I32],
    "az_engine_device_bytes": [_VP, C.POINTER(C.c_int64)],
    "az_engine_release_phase": [_VP],
    "az_comm_gather_push": [_VP, _VP, _VP, C.c_double, C.POINTER(GatherStats)],
    "az_comm_broadcast_params": [_VP, _VP, _I32],
}

# az_struct_id order of include/azhip.h
STRUCTS = [("az_engine_cfg", EngineCfg), ("az_move_rec", MoveRec), ("az_game_rec", GameRec), ("az_trace_buf", TraceBuf),
           ("az_selfplay_stats", SelfplayStats), ("az_sample", Sample), ("az_dataset_info", DatasetInfo),
           ("az_learning_status_t", LearningStatusRec), ("az_train_cfg", TrainCfg), ("az_gather_stats", GatherStats), ("az_prof", Prof)]
_lib = None


def lib():
    """Load libazhip.so; raises ImportError loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libazhip.so is missing at %s -- build it with `python __graft_entry__.py` or "
                "`make -C alphazero.jl_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in SYMBOLS.items():
            f = getattr(L, name)     # AttributeError if a declared symbol is not exported
            if name == "az_last_error":
                f.restype = C.c_char_p
                f.argtypes = []
            else:
                f.restype = C.c_int
                f.argtypes = args
        if L.az_abi_version() != ABI_VERSION:
            raise ImportError("libazhip.so ABI version %d, this mirror is written against %d (include/azhip.h AZ_ABI_VERSION)" % (L.az_abi_version(), ABI_VERSION))
        # a stale mirror must fail here, not be written out of bounds by the library
        for which, (name, st) in enumerate(STRUCTS):
            if L.az_abi_struct_size(which) != C.sizeof(st):
                raise ImportError("%s: the library's struct is %d bytes, this mirror's %d" % (name, L.az_abi_struct_size(which), C.sizeof(st)))
        _lib = L
    return _lib


def check(status):
    if status != AZ_OK:
        raise AzError(status, lib().az_last_error().decode("utf-8", "replace"))
