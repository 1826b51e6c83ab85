"""The N > 1 path of the device pieces on a GPU box: two ranks (gloo group, both on cuda:0 -- the box exposes one GPU)
run the sharded self-play step into their own device replay memory; every rank must end up with the samples of the
unsharded run, in the same order (game ids are global, the gathered records are re-packed by id)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params():
    import azhip
    from azhip.training import SelfPlayParams
    return SelfPlayParams(
        mcts=azhip.MctsParams(num_iters_per_turn=16, cpuct=2.0, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0,
                              temperature=azhip.PLSchedule([0, 4], [1.0, 0.5])),
        sim=azhip.SimParams(num_games=10, num_workers=4, batch_size=4, use_gpu=True, reset_every=1))


def _samples(mem):
    with mem.dataset() as d:
        raw = d.raw_samples()
        return np.array([[raw[i].key[0], raw[i].key[1], raw[i].n] + [np.float64(x).view(np.uint64) for x in list(raw[i].pi) + [raw[i].z, raw[i].t]]
                         for i in range(len(d))], dtype=np.uint64)


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [os.path.join(ROOT, "alphazero.jl_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import azhip
    from azhip.training import self_play_step_device
    gspec = azhip.TicTacToeSpec()
    nn = azhip.ResNet(gspec, azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=4)
    mem = azhip.MemoryBuffer(gspec, 10000)
    rep = self_play_step_device(gspec, nn, _params(), mem, seed=6)
    np.save(os.path.join(out_dir, "S%d.npy" % rank), _samples(mem))
    assert rep.memory_size == len(mem)
    mem.close()
    dist.destroy_process_group()


def test_sharded_self_play_step_fills_identical_memories(tmp_path):
    port = 29700 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    import azhip
    from azhip.training import self_play_step_device
    gspec = azhip.TicTacToeSpec()
    nn = azhip.ResNet(gspec, azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=4)
    mem = azhip.MemoryBuffer(gspec, 10000)
    self_play_step_device(gspec, nn, _params(), mem, seed=6)          # single process, all 10 games
    want = _samples(mem)
    mem.close()
    S0, S1 = np.load(tmp_path / "S0.npy"), np.load(tmp_path / "S1.npy")
    assert np.array_equal(S0, S1) and np.array_equal(S0, want)


def test_native_comm_single_rank_gather_and_broadcast():
    """az_comm_* (csrc/comm.hip) with a world of ONE rank -- what a 1-GPU box can run of the RCCL path: ncclCommInitRank,
    the all-gathers of az_comm_gather_push (device-resident records -> the rank's device memory, global game-id order)
    and the parameter broadcast.  The samples equal the ones the host path pushes for the same phase.  Ranks > 1 need one
    GPU each (RCCL refuses two ranks on a device): the driver's multi-GPU bench exercises them (bench.py --gpus N)."""
    import azhip
    from azhip import comm
    gspec = azhip.ConnectFourSpec()
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=4)
    kw = dict(game=0, oracle=azhip.ORACLE_RESNET, num_workers=6, batch_size=3, num_iters_per_turn=16, cpuct=2.0, dirichlet_noise_eps=0.25,
              reset_every=1, seed=5, num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    with comm.Comm(0, 0, 1, comm.unique_id()) as c, azhip.Engine(**kw) as e, azhip.Engine(**kw) as e2:
        e.net_set_params(nn.params())
        games, moves, ng, nm, st = e.selfplay_run(9, first_game_id=40)
        a, b = azhip.MemoryBuffer(gspec, 10000), azhip.MemoryBuffer(gspec, 10000)
        a.push_records(games, moves, ng, nm, 1.0)
        gs = c.gather_push(e, b, 1.0)
        assert (gs.games, gs.moves) == (9, nm) and gs.bytes >= nm * 64 and gs.gather_ms > 0
        assert np.array_equal(_samples(a), _samples(b))
        assert c.gather_push(e, None, 1.0).moves == nm             # a rank without a memory still takes part in the collective
        with pytest.raises(azhip.AzError):
            c.broadcast_params(e2, root=0)                          # the root must hold parameters
        c.broadcast_params(e, root=0)
        assert np.array_equal(e.net_get_params(), nn.params())
        a.close(); b.close()


STUB = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")


def _native_worker(rank, world, port, out_dir, stub=False):
    """stub: all ranks on cuda:0 with the test-only transport (tests/rccl_stub) in place of RCCL -- the code under test,
    csrc/comm.hip, is the shipped one"""
    sys.path[:0] = [os.path.join(ROOT, "alphazero.jl_amd")]
    if stub:
        os.environ["AZHIP_RCCL_LIB"] = STUB
        os.environ["AZSTUB_SLOT_MB"] = "1"                                # az_comm_broadcast_params' 1 MB+ blob goes in pieces
    dev = 0 if stub else rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)       # carries the 128-byte unique id only
    import azhip
    from azhip import comm
    from azhip.training import self_play_step_device
    gspec = azhip.TicTacToeSpec()
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=4 if rank == 0 else 99)             # rank 1 starts with other weights: the broadcast must replace them
    with comm.Comm(dev, rank, world, comm.torch_broadcast_id(rank)) as c:
        mem = azhip.MemoryBuffer(gspec, 10000, device=dev)
        kw = dict(game=1, oracle=azhip.ORACLE_RESNET, num_workers=4, batch_size=4, num_iters_per_turn=16, num_blocks=1, num_filters=64,
                  num_policy_head_filters=32, num_value_head_filters=32, device=dev)
        with azhip.Engine(**kw) as e:
            # failure agreement: the root holds no parameters yet -- EVERY rank gets an error (the root its own, the others
            # AZ_ERR_COMM) instead of the non-roots waiting in ncclBroadcast for ever; the communicator stays usable
            try:
                c.broadcast_params(e, root=0)
                raise AssertionError("broadcast from a root without parameters succeeded")
            except azhip.AzError as ex:
                assert ("no parameters" in str(ex)) if rank == 0 else ("rank 0 cannot take part" in str(ex)), str(ex)
            # ... and a rank without a device-resident phase stops the gather on all ranks, nothing exchanged
            try:
                c.gather_push(e, None, 1.0)
                raise AssertionError("gather without a phase succeeded")
            except azhip.AzError as ex:
                assert "no device-resident phase" in str(ex), str(ex)
            if rank == 0:
                e.net_set_params(nn.params())
            c.broadcast_params(e, root=0)                                    # ncclBroadcast of the blob
            got = e.net_get_params()
        nn0 = azhip.ResNet(gspec, hp, params=got)
        rep = self_play_step_device(gspec, nn0, _params(), mem, seed=6, comm=c)
        np.save(os.path.join(out_dir, "N%d.npy" % rank), _samples(mem))
        np.save(os.path.join(out_dir, "R%d.npy" % rank), np.array([rep.average_exploration_depth, rep.mcts_memory_footprint], dtype=np.float64))
        np.save(os.path.join(out_dir, "W%d.npy" % rank), got)
        assert rep.memory_size == len(mem)
        mem.close()
    dist.destroy_process_group()


def test_native_comm_two_ranks_on_two_gpus(tmp_path):
    """az_comm_* with a world of TWO ranks, one GPU each (RCCL over xGMI): weights broadcast from rank 0, shards played
    device-only, ncclAllGather of the records straight into both ranks' device memories -- identical to the unsharded
    single-GPU run.  Needs two visible GPUs (RCCL refuses two ranks on one device): skipped on the 1-GPU test box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the 1-GPU box runs test_native_comm_single_rank_gather_and_broadcast and the gloo test)")
    port = 29900 + os.getpid() % 2000
    mp.spawn(_native_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    import azhip
    from azhip.training import self_play_step_device
    gspec = azhip.TicTacToeSpec()
    nn = azhip.ResNet(gspec, azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=4)
    mem = azhip.MemoryBuffer(gspec, 10000)
    self_play_step_device(gspec, nn, _params(), mem, seed=6)
    want = _samples(mem)
    mem.close()
    N0, N1 = np.load(tmp_path / "N0.npy"), np.load(tmp_path / "N1.npy")
    assert np.array_equal(N0, N1) and np.array_equal(N0, want)
    assert np.array_equal(np.load(tmp_path / "W0.npy"), nn.params()) and np.array_equal(np.load(tmp_path / "W1.npy"), nn.params())


@pytest.mark.parametrize("world", [2, 4])
def test_native_comm_several_ranks_share_one_gpu_through_the_transport_seam(tmp_path, world):
    """The shipped exchange (csrc/comm.hip: az_comm_gather_push / az_comm_broadcast_params) with a world of 2 and 4 ranks on
    the ONE GPU of the test box: AZHIP_RCCL_LIB points the library's loader at tests/rccl_stub (host shared memory behind the
    six nccl entry points; RCCL itself refuses two ranks per device).  10 games split 6 + 4 / 4 + 2 + 2 + 2 (divrem, remainder
    to rank 0: the padded segments differ in fill), every rank's device memory must hold exactly the samples of the unsharded
    run in the same order, the broadcast weights must arrive, Report.SelfPlay's depth / footprint must cover all games, and
    the failure-agreement paths must return on all ranks."""
    import subprocess
    if not os.path.exists(STUB):
        subprocess.check_call(["make", "-C", os.path.dirname(STUB)])
    port = 30100 + os.getpid() % 2000 + world
    mp.spawn(_native_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    import azhip
    from azhip.training import self_play_step_device
    gspec = azhip.TicTacToeSpec()
    nn = azhip.ResNet(gspec, azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=4)
    mem = azhip.MemoryBuffer(gspec, 10000)
    rep = self_play_step_device(gspec, nn, _params(), mem, seed=6)
    want = _samples(mem)
    mem.close()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("N%d.npy" % r)), want), r
        assert np.array_equal(np.load(tmp_path / ("W%d.npy" % r)), nn.params()), r
        got = np.load(tmp_path / ("R%d.npy" % r))
        # the largest tree is a property of the games (reset_every = 1); the exploration depth of a game is its WORKER's
        # running average (MCTS.reset! keeps the counters, mcts.jl:278-281), so it depends on which games shared a slot:
        # every rank must report the same mean over ALL games, not the unsharded run's
        assert got[1] == rep.mcts_memory_footprint and got[0] > 0.5, (r, got, rep)
        assert np.array_equal(got, np.load(tmp_path / "R0.npy")), r
