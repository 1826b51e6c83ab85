"""Host-side mirror logic that needs no GPU: parameter mapping, sharding, trace / sample reconstruction
from packed records (exercised on records produced by the oracle, which share the ABI layout)."""
import ctypes as C

import numpy as np
import pytest

import azref as R
from azhip import MctsParams, PLSchedule, ConstSchedule, SimParams
from azhip.memory import pack_samples, push_trace
from azhip.params import check_sim_params, engine_options
from azhip.simulations import shard_games
from azhip.trace import policy_from_visits, trace_from_records


def test_engine_options_mapping():
    """SURVEY.md §8b config mapping: MctsParams/SimParams -> az_engine_cfg"""
    m = MctsParams(num_iters_per_turn=400, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0, cpuct=2.0,
                   temperature=PLSchedule([0, 20, 30], [1.0, 1.0, 0.3]))
    s = SimParams(num_games=5000, num_workers=128, batch_size=64, use_gpu=True, reset_every=2)
    o = engine_options(m, s, seed=9)
    assert o["num_iters_per_turn"] == 400 and o["cpuct"] == 2.0 and o["temperature"] == ([0, 20, 30], [1.0, 1.0, 0.3])
    assert o["num_workers"] == 128 and o["batch_size"] == 64 and o["reset_every"] == 2 and o["seed"] == 9
    assert engine_options(MctsParams(2, 0.0, 1.0), SimParams(1, 1, 1, reset_every=None))["reset_every"] == 0
    assert engine_options(MctsParams(2, 0.0, 1.0, temperature=ConstSchedule(0.5)), SimParams(1, 1, 1))["temperature"] == ([0], [0.5])
    with pytest.raises(ValueError):
        check_sim_params(SimParams(10, 4, 8))                  # batch_size <= num_workers, params.jl:361-384
    with pytest.raises(ValueError):
        check_sim_params(SimParams(10, 8, 8, flip_probability=0.3))
    check_sim_params(SimParams(10, 8, 8, flip_probability=0.3), arena=True)    # the arena honours flips (play.jl:305-307)
    with pytest.raises(ValueError):
        check_sim_params(SimParams(10, 8, 8, flip_probability=1.3), arena=True)


def test_shard_games_is_the_reference_split():
    """simulations.jl:268-278: divrem, remainder to the first worker; ids contiguous in rank order"""
    for n, w in ((32768, 8), (10, 3), (7, 7), (4101, 4)):
        shards = [shard_games(n, w, r) for r in range(w)]
        assert sum(c for _, c in shards) == n and shards[0][1] == n // w + n % w
        assert all(c == n // w for _, c in shards[1:])
        assert [f for f, _ in shards] == list(np.cumsum([0] + [c for _, c in shards[:-1]]))
    with pytest.raises(ValueError):
        shard_games(3, 4, 0)


def test_policy_from_visits_is_mcts_policy():
    """mcts.jl:255-271 vs the oracle"""
    g = R.Game(R.C4)
    for a in (3, 3, 3, 3, 3, 3, 2):
        g.play(a)                                   # column 3 is now full
    m = R.Mcts(R.C4, oracle=R.ORACLE_HASH, cpuct=2.0)
    m.explore(g, 150, eta=np.zeros(9))
    acts, pi = m.policy(g)
    N, _, _, _ = m.root_stats(g)
    full = np.zeros(7, dtype=np.int64); full[acts] = N
    mine = policy_from_visits(full, g.actions_mask())
    assert 3 not in acts and np.array_equal(mine, pi) and abs(pi.sum() - 1) < 1e-15


def test_trace_and_samples_from_records():
    games, moves, nm = R.simulate(R.TTT, R.ORACLE_HASH, 6, 3, 30, cpuct=1.5, noise_eps=0.25, seed=2, temp_ys=(1.0,))
    mask_fn = lambda key: R.Game(R.TTT, R.unpack_key(R.TTT, key)).actions_mask()
    mem = []
    for i in range(6):
        t = trace_from_records(games[i], moves, 9, mask_fn)
        assert t.valid() and len(t) == games[i].num_moves
        assert t.states[0] == (0, 0) and t.states[-1] == tuple(games[i].final_key)
        assert all(abs(p.sum() - 1) < 1e-12 for p in t.policies) and t.rewards[-1] in (-1.0, 0.0, 1.0)
        assert all(r == 0 for r in t.rewards[:-1])
        n0 = len(mem)
        assert push_trace(mem, t, 1.0) == len(t)
        # last position first, z is relative to the player to move (memory.jl:74-87)
        last = mem[n0]
        assert last.t == 1.0 and last.s == t.states[-2]
        wp = not (last.s[0] >> 63)
        assert last.z == (t.rewards[-1] if wp else -t.rewards[-1])
    packed = pack_samples(games, moves, 6, 9, 1.0)
    assert len(packed) == nm == len(mem)
    by_key = {}
    for smp in mem:
        by_key.setdefault(smp.s, []).append(smp)
    for r in packed[:50]:
        cands = by_key[(int(r["key"][0]), int(r["key"][1]))]
        assert any(c.z == r["z"] and c.t == r["t"] for c in cands)


def test_repack_by_game_id_orders_games_and_keeps_their_moves():
    """the distributed self-play step feeds the replay memory in game-id order whatever the rank layout"""
    from azhip.simulations import GAME_DTYPE, MOVE_DTYPE
    from azhip.training import repack_by_game_id
    # two "ranks": rank 0 played games 3,0, rank 1 played games 2,1 (finish order), gathered back to back
    lens = {3: 2, 0: 3, 2: 1, 1: 4}
    g = np.zeros(4, dtype=GAME_DTYPE)
    m = np.zeros(sum(lens.values()), dtype=MOVE_DTYPE)
    off = 0
    for i, gid in enumerate((3, 0, 2, 1)):
        g[i]["game_id"], g[i]["num_moves"], g[i]["first_move"] = gid, lens[gid], off
        for k in range(lens[gid]):
            m[off + k]["action"] = 10 * gid + k
        off += lens[gid]
    G, M = repack_by_game_id(g, m)
    assert list(G["game_id"]) == [0, 1, 2, 3] and list(G["first_move"]) == [0, 3, 7, 8] and list(G["num_moves"]) == [3, 4, 1, 2]
    assert list(M["action"]) == [0, 1, 2, 10, 11, 12, 13, 20, 30, 31]
    G0, M0 = repack_by_game_id(g[:0], m[:0])
    assert len(G0) == 0 and len(M0) == 0


def test_parameter_file_round_trip(tmp_path):
    from azhip.game import ConnectFourSpec, TicTacToeSpec
    from azhip.network import ResNet, ResNetHP, load_params, save_params
    nn = ResNet(ConnectFourSpec(), ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32), seed=3)
    p = str(tmp_path / "net.azhip")
    save_params(p, nn)
    back = load_params(p, ConnectFourSpec())
    assert np.array_equal(back.params(), nn.params()) and back.hyper == nn.hyper
    with pytest.raises(ValueError):
        load_params(p, TicTacToeSpec())
