"""The C-ABI library loads on a CPU-only box and exports every symbol include/azhip.h declares, with the
record layouts the header states.  No compute call is made here (there is no GPU and no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from azhip import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "azhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(az_[a-z0-9_]+)\s*\(", src)) - {"az_progress_cb"}


def test_every_declared_symbol_is_exported_and_bound():
    lib = L.lib()
    declared = header_functions()
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.az_abi_version() == 4 == L.ABI_VERSION


def test_struct_sizes_of_the_mirror_are_the_library_s():
    """az_selfplay_stats / az_gather_stats / az_prof grew; a mirror built against an older header must not load"""
    lib = L.lib()
    sizes = {name: lib.az_abi_struct_size(i) for i, (name, _) in enumerate(L.STRUCTS)}
    assert sizes == {name: C.sizeof(st) for name, st in L.STRUCTS}
    assert (sizes["az_engine_cfg"], sizes["az_move_rec"], sizes["az_game_rec"], sizes["az_selfplay_stats"], sizes["az_gather_stats"],
            sizes["az_prof"], sizes["az_sample"]) == (224, 64, 56, 88, 88, 256, 112)
    assert lib.az_abi_struct_size(len(L.STRUCTS)) == -1


def test_record_layouts_match_the_oracle_records():
    import azref as R
    assert C.sizeof(L.MoveRec) == 64 == C.sizeof(R.MoveRec)
    assert C.sizeof(L.GameRec) == 56 == C.sizeof(R.GameRec)
    for a, b in zip(L.MoveRec._fields_, R.MoveRec._fields_):
        assert a[0] == b[0] and C.sizeof(a[1]) == C.sizeof(b[1])
    from azhip.simulations import GAME_DTYPE, MOVE_DTYPE
    assert GAME_DTYPE.itemsize == 56 and MOVE_DTYPE.itemsize == 64


def test_cfg_defaults_are_the_connect_four_experiment():
    """games/connect-four/params.jl:5-30"""
    cfg = L.EngineCfg()
    L.check(L.lib().az_engine_cfg_init(C.byref(cfg)))
    assert cfg.struct_size == C.sizeof(L.EngineCfg)
    assert (cfg.cpuct, cfg.dirichlet_noise_eps, cfg.dirichlet_noise_alpha, cfg.num_iters_per_turn) == (2.0, 0.25, 1.0, 600)
    assert list(cfg.temperature_xs[:3]) == [0, 20, 30] and list(cfg.temperature_ys[:3]) == [1.0, 1.0, 0.3]
    assert (cfg.num_workers, cfg.batch_size, cfg.reset_every) == (128, 64, 2)


def test_pure_host_entry_points():
    lib = L.lib()
    n = C.c_int32()
    for game, na, dims in ((0, 7, (7, 6, 3)), (1, 9, (3, 3, 3)), (2, 6, (14, 1, 5))):
        L.check(lib.az_game_num_actions(game, C.byref(n))); assert n.value == na
        w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(lib.az_game_state_dim(game, C.byref(w), C.byref(h), C.byref(c))); assert (w.value, h.value, c.value) == dims
    k = (C.c_uint64 * 2)()
    L.check(lib.az_game_init_key(2, k)); assert (k[0], k[1]) == (0x030303030303, 0x030303030303)
    assert lib.az_game_num_actions(9, C.byref(n)) == L.AZ_ERR_BAD_ARG and b"game" in lib.az_last_error()


def test_errors_are_statuses_not_crashes():
    """bad configuration -> AZ_ERR_BAD_ARG with a message; without a GPU engine creation reports AZ_ERR_HIP."""
    from azhip.engine import default_cfg
    lib = L.lib()
    h = C.c_void_p()
    for kw, frag in ((dict(batch_size=300), b"batch_size"), (dict(num_iters_per_turn=1), b"num_iters_per_turn"),
                     (dict(flip_probability=1.5), b"flip_probability"), (dict(game=7), b"game")):
        cfg = default_cfg(**kw)
        assert lib.az_engine_create(C.byref(cfg), C.byref(h)) == L.AZ_ERR_BAD_ARG
        assert frag in lib.az_last_error() and not h.value
    import torch
    if not torch.cuda.is_available():
        cfg = default_cfg()
        st = lib.az_engine_create(C.byref(cfg), C.byref(h))
        assert st in (L.AZ_ERR_HIP, L.AZ_ERR_BAD_ARG) and lib.az_last_error() and not h.value
        with pytest.raises(L.AzError):
            import azhip
            azhip.Engine()
    assert lib.az_engine_destroy(None) == 0


def test_push_trace_host_function_matches_oracle_and_mirror():
    """az_push_trace == push_trace! (src/memory.jl:74-87) == the oracle's restatement"""
    import numpy as np
    import azref as R
    games, moves, nm = R.simulate(R.MANCALA, R.ORACLE_HASH, 3, 2, 20, cpuct=2.0, noise_eps=0.25, seed=4)
    for gi in range(3):
        g = games[gi]
        recs = (L.MoveRec * g.num_moves)()
        C.memmove(recs, C.byref(moves, g.first_move * 64), g.num_moves * 64)
        z = np.zeros(g.num_moves); t = np.zeros(g.num_moves); zr = np.zeros(g.num_moves); tr = np.zeros(g.num_moves)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        L.check(L.lib().az_push_trace(recs, g.num_moves, 0.9, vp(z), vp(t)))
        R.lib().azr_push_trace(R.MANCALA, C.byref(moves, g.first_move * 64), g.num_moves, C.c_double(0.9), vp(zr), vp(tr))
        assert np.array_equal(z, zr) and np.array_equal(t, tr)
        assert t[0] == g.num_moves and t[-1] == 1 and abs(z[-1]) in (0.0, 1.0)


def test_new_record_layouts():
    """az_sample (112 B, also the oracle's record), az_train_cfg (size written by az_train_cfg_init), the small result structs"""
    import azref as R
    assert C.sizeof(L.Sample) == 112 == C.sizeof(R.Sample) == R.lib().azr_sizeof_sample()
    cfg = L.TrainCfg()
    L.check(L.lib().az_train_cfg_init(C.byref(cfg)))
    assert cfg.struct_size == C.sizeof(L.TrainCfg) and cfg.optimiser == L.OPT_ADAM and cfg.batch_size == 1024
    assert abs(cfg.lr - 2e-3) < 1e-9 and cfg.l2_regularization == 1e-4       # games/connect-four/params.jl:46-58
    assert C.sizeof(L.DatasetInfo) == 32 and C.sizeof(L.LearningStatusRec) == 28
    assert L.lib().az_train_cfg_init(None) == L.AZ_ERR_BAD_ARG


def test_transport_stub_exports_the_entry_points_comm_hip_binds():
    """tests/rccl_stub (TEST infrastructure: several ranks on one GPU) must offer exactly the symbols csrc/comm.hip looks up"""
    import ctypes
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "rccl_stub")], stdout=subprocess.DEVNULL)
    src = open(os.path.join(root, "alphazero.jl_amd", "csrc", "comm.hip")).read()
    wanted = set(re.findall(r'dlsym\(so, "(nccl[A-Za-z]+)"\)', src))
    assert wanted == {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclAllGather", "ncclBroadcast", "ncclGetErrorString", "ncclGetVersion"}
    wanted.discard("ncclGetVersion")                                 # optional: only az_comm_version asks for it
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(root, "tests", "rccl_stub", "librccl_stub.so")], text=True)
    have = set(re.findall(r"\bT (nccl[A-Za-z]+)", out))
    assert wanted <= have, wanted - have
    # the product never names the stub: only the environment variable does
    for f in ("comm.hip", "azhip.hip"):
        assert "rccl_stub" not in open(os.path.join(root, "alphazero.jl_amd", "csrc", f)).read().replace("tests/rccl_stub", "")
