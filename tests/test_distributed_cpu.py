"""The N > 1 path on CPU: world_size-2 gloo group, shard -> (synthetic) local records -> gather_records.
The records are produced by the oracle so the test also checks that the union of the shards equals the
unsharded run (results keyed by GLOBAL game id do not depend on the number of ranks)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [os.path.join(ROOT, "alphazero.jl_amd"), os.path.join(ROOT, "oracle")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import azref as R
    from azhip.simulations import GAME_DTYPE, MOVE_DTYPE, gather_records, shard_games
    first, count = shard_games(7, world, rank)
    # the oracle keys its RNG streams (noise, move, flip) by GLOBAL game id: the rank simulates its own shard, ids
    # first .. first+count-1 (simulations.jl:268-278), with play_game's random symmetries switched on
    games, moves, nm = R.simulate(R.TTT, R.ORACLE_HASH, count, count, 20, cpuct=1.5, noise_eps=0.25, seed=3, first_game_id=first,
                                  flip_probability=0.4)
    g = np.frombuffer(bytes(games), dtype=GAME_DTYPE)[:count].copy()
    assert list(g["game_id"]) == list(range(first, first + count))
    # the oracle writes a game's moves when the game ends: pack them in game-id order like the engine's trace buffer
    m = np.concatenate([np.frombuffer(bytes(moves), dtype=MOVE_DTYPE)[r["first_move"]:r["first_move"] + r["num_moves"]] for r in g])
    g["first_move"] = np.cumsum([0] + list(g["num_moves"][:-1]))
    G, M = gather_records(g, m)
    np.save(os.path.join(out_dir, "G%d.npy" % rank), G)
    np.save(os.path.join(out_dir, "M%d.npy" % rank), M)
    dist.destroy_process_group()


def test_gather_records_gloo_world2(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sys.path[:0] = [os.path.join(ROOT, "oracle")]
    import azref as R
    from azhip.simulations import GAME_DTYPE, MOVE_DTYPE
    games, moves, nm = R.simulate(R.TTT, R.ORACLE_HASH, 7, 7, 20, cpuct=1.5, noise_eps=0.25, seed=3, flip_probability=0.4)   # unsharded
    assert any(moves[k].N[R.AMAX] for k in range(nm))                   # some turns were flipped
    ref_g = np.frombuffer(bytes(games), dtype=GAME_DTYPE)[:7]
    ref_m = np.frombuffer(bytes(moves), dtype=MOVE_DTYPE)[:nm]
    G0, M0 = np.load(tmp_path / "G0.npy"), np.load(tmp_path / "M0.npy")
    G1, M1 = np.load(tmp_path / "G1.npy"), np.load(tmp_path / "M1.npy")
    assert np.array_equal(G0, G1) and np.array_equal(M0, M1)          # every rank holds the same gather
    assert list(G0["game_id"]) == list(range(7)) and len(M0) == nm
    for r, q in zip(G0, ref_g):
        a = M0[r["first_move"]:r["first_move"] + r["num_moves"]]
        b = ref_m[q["first_move"]:q["first_move"] + q["num_moves"]]
        assert np.array_equal(a, b) and tuple(r["final_key"]) == tuple(q["final_key"])


def _bcast_worker(rank, world, port, out_dir):
    sys.path[:0] = [os.path.join(ROOT, "alphazero.jl_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azhip import ConnectFourSpec, ResNet, ResNetHP, broadcast_params
    nn = ResNet(ConnectFourSpec(), ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=100 + rank)
    broadcast_params(nn, src=0)
    np.save(os.path.join(out_dir, "W%d.npy" % rank), nn.params())
    dist.destroy_process_group()


def test_broadcast_params_gloo_world2(tmp_path):
    """weights of rank 0 reach every rank before the self-play phase"""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_bcast_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sys.path[:0] = [os.path.join(ROOT, "alphazero.jl_amd")]
    from azhip.network import ResNetHP, random_params
    w0, w1 = np.load(tmp_path / "W0.npy"), np.load(tmp_path / "W1.npy")
    ref = random_params(0, ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=100)
    assert np.array_equal(w0, ref) and np.array_equal(w1, ref)


class _FakeMemory:
    """what self_play_step_device needs of a MemoryBuffer, on the host"""

    def __init__(self):
        self.pushed = []

    def new_batch(self):
        pass

    def push_records(self, games, moves, ng, nm, gamma):
        self.pushed.append((ng, nm, sorted(int(games[i].game_id) for i in range(ng))))

    def __len__(self):
        return sum(p[1] for p in self.pushed)

    def dataset(self, **kw):
        import contextlib
        n = len(self)

        class D:
            def __len__(self):
                return n
        return contextlib.nullcontext(D())


def _abort_worker(rank, world, port, out_dir, num_games, replaced_on_rank1):
    """self_play_step_device over a gloo group with the local phase replaced by oracle-made records; rank 1 reports
    `replaced_on_rank1` of its games under replacement ids (games its slots aborted and played again)"""
    import ctypes as C
    import json
    import warnings
    sys.path[:0] = [os.path.join(ROOT, "alphazero.jl_amd"), os.path.join(ROOT, "oracle")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import azref as R
    from azhip import _lib as L
    from azhip import simulations as S
    from azhip import training as T
    from azhip.params import MctsParams, SimParams

    def fake_run_local(simulator, gspec, p, first_game_id=0, game_simulated=None, device=0, seed=1, device_only=False):
        games, moves, nm = R.simulate(R.TTT, R.ORACLE_HASH, p.num_games, p.num_games, 12, cpuct=1.5, noise_eps=0.25, seed=5, first_game_id=first_game_id)
        g = (L.GameRec * max(p.num_games, 1)).from_buffer_copy(bytes(games)[:C.sizeof(L.GameRec) * max(p.num_games, 1)])
        m = (L.MoveRec * max(nm, 1)).from_buffer_copy(bytes(moves)[:C.sizeof(L.MoveRec) * max(nm, 1)])
        if rank == 1:
            for i in range(replaced_on_rank1):
                g[i].game_id |= L.REPLACEMENT_GAME_BIT
        st = L.SelfplayStats()
        st.aborted_games = replaced_on_rank1 if rank == 1 else 0
        st.moves = nm
        return g, m, p.num_games, nm, st, None
    S.run_local = fake_run_local
    mem = _FakeMemory()
    params = T.SelfPlayParams(mcts=MctsParams(num_iters_per_turn=12, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0), sim=SimParams(num_games=num_games, num_workers=4, batch_size=4, use_gpu=False))
    outcome = {"rank": rank}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            T.self_play_step_device(None, None, params, mem)
            outcome["raised"] = None
        except L.AzError as ex:
            outcome["raised"] = str(ex)
        outcome["warned"] = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    outcome["pushed"] = mem.pushed
    # the group must still be whole: a rank that had left on its own would have left the other in all_gather, and this
    # barrier is where a survivor of such a split would hang
    dist.barrier()
    json.dump(outcome, open(os.path.join(out_dir, "abort%d.json" % rank), "w"))
    dist.destroy_process_group()


@pytest.mark.parametrize("num_games,replaced,expect_raise", [(6, 2, True), (60, 1, False)])
def test_abort_verdict_is_formed_after_the_gather_on_every_rank_alike(tmp_path, num_games, replaced, expect_raise):
    """a rank whose slots aborted games used to raise BEFORE the collective and leave the other ranks in the
    all-gather.  Now every rank gathers first and forms the verdict from the gathered records: above 5 % every rank refuses -- with the
    same message -- and nothing is pushed anywhere; below, every rank warns and pushes the same games."""
    import json
    port = 33500 + (os.getpid() + num_games) % 2000
    mp.spawn(_abort_worker, args=(2, port, str(tmp_path), num_games, replaced), nprocs=2, join=True)
    o = [json.load(open(tmp_path / ("abort%d.json" % r))) for r in range(2)]
    if expect_raise:
        assert o[0]["raised"] and o[0]["raised"] == o[1]["raised"] and "%d of %d" % (replaced, num_games) in o[0]["raised"]
        assert o[0]["pushed"] == o[1]["pushed"] == []
    else:
        assert o[0]["raised"] is None and o[1]["raised"] is None
        assert len(o[0]["warned"]) == len(o[1]["warned"]) == 1 and o[0]["warned"] == o[1]["warned"]
        assert o[0]["pushed"] == o[1]["pushed"] and o[0]["pushed"][0][0] == num_games
