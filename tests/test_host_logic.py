import os
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
    check_sim_params(SimParams(10, 8, 8, flip_probability=0.3))                # self-play and the arena honour flips (play.jl:305-307)
    check_sim_params(SimParams(10, 8, 8, flip_probability=0.3), arena=True)
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


def test_tower_row_permutation_tables():
    """Geo16 (csrc/resnet16.h): the tower kernels order a workgroup's rows by border class so that taps which fall off
    the board for a whole 16-row tile are skipped.  Host-side check of the tables the kernels use: the permutation is a
    bijection onto (board, position), every neighbour entry is the row of the true neighbour or one of the 8 zero rows, and
    the product counts are the ones DESIGN.md quotes (Connect-Four: 85 of 99 for 4 boards, 159 of 189 for 8; Mancala 31 of 99).
    Round 3: inside a class the rows follow a linear residue function, so the 8 tap-shifted rows of a ds_read_b128 pass (half a
    tile) are distinct mod 8 (= hit different LDS banks): extra LDS cycles per convolution 119 -> 18 (Connect-Four, 4 boards),
    0 for 8 boards and Mancala."""
    import ctypes as C
    import numpy as np
    from azhip import _lib as L
    lib = L.lib()
    f = lib.az_debug_tower_geometry
    f.restype = C.c_int
    f.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    dims = {0: (7, 6), 1: (3, 3), 2: (14, 1), 3: (9, 9)}
    expect = {(0, 0): 85, (0, 1): 26, (0, 2): 159, (2, 0): 31}
    max_extra = {(0, 0): 18, (0, 2): 0, (2, 0): 0, (3, 0): 9, (1, 0): 12}   # extra LDS cycles over all passes of one convolution
    for game, (W, H) in dims.items():
        P = W * H
        for which, ntiles in ((0, 11), (1, 3 if P <= 48 else (P + 15) // 16), (2, 21)):
            out = np.zeros(10 * 21 * 16, dtype=np.uint16)
            rows, prod = C.c_int32(), C.c_int32()
            L.check(f(game, which, out.ctypes.data_as(C.c_void_p), out.size, C.byref(rows), C.byref(prod)))
            R = rows.value
            assert R == ntiles * 16
            pos, nbr = out[:R].astype(int), out[R:10 * R].astype(int).reshape(9, R)
            TB = R // P
            real = pos[pos != 0xffff]
            assert sorted(real) == list(range(TB * P))                       # every (board, position) exactly once
            row_of = {int(p): i for i, p in enumerate(pos) if p != 0xffff}
            products, extra = 0, 0
            for tile in range(ntiles):
                for tap in range(9):
                    dy, dx = tap // 3 - 1, tap % 3 - 1
                    used = False
                    for r in range(tile * 16, tile * 16 + 16):
                        want = None                                           # one of the zero rows R .. R + 7
                        if pos[r] != 0xffff:
                            b, q = divmod(int(pos[r]), P)
                            x, y = q % W + dx, q // W + dy
                            if 0 <= x < W and 0 <= y < H:
                                want = row_of[b * P + y * W + x]
                                used = True
                        assert (nbr[tap, r] == want) if want is not None else (R <= nbr[tap, r] < R + 8), (game, which, tile, tap, r)
                    products += used
                    if used:
                        for half in (0, 8):                                   # a pass = 8 rows: worst bank multiplicity - 1 extra cycles
                            by_res = {}
                            for r in range(tile * 16 + half, tile * 16 + half + 8):
                                by_res.setdefault(int(nbr[tap, r]) % 8, set()).add(int(nbr[tap, r]))
                            extra += max(len(v) for v in by_res.values()) - 1
            if (game, which) in max_extra:
                assert extra <= max_extra[(game, which)], (game, which, extra)
            assert products == prod.value <= 9 * ntiles
            if (game, which) in expect:
                assert prod.value == expect[(game, which)]
            if game == 2:
                assert prod.value <= 3 * ntiles                              # 14x1 board: the six taps with dy != 0 never apply


def test_register_budget_of_the_kernels_that_share_a_cu():
    """Two design points rest on kernels fitting on a CU TOGETHER (512 VGPRs per SIMD lane, allocated in blocks of 8): the
    headline step (two k_tower16 workgroups + k_tree: 2 x 176 + 80) and the optimiser step's backward pass (k_conv16_layer at two
    wavefronts per SIMD + one 4-wavefront k_wgrad16 workgroup: 2 x 160 + 192 = 512 to the register; two k_wgrad16 workgroups + a
    wavefront of the batch-norm passes).  A compiler that spends eight registers more breaks neither parity nor a test on the GPU,
    only the overlap: check the built code objects (skipped when csrc/*.o have not been built)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "alphazero.jl_amd", "csrc", "train.o")):
        pytest.skip("csrc/*.o not built")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py")], capture_output=True, text=True).stdout
    regs = {}
    for line in out.splitlines():
        m = re.match(r"(\S.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+).*spill (\d+)", line)
        if m:
            regs[m.group(1)] = (-(-(int(m.group(2)) + int(m.group(3))) // 8) * 8, int(m.group(4)))
    def alloc(prefix):
        hit = [v for k, v in regs.items() if k.startswith(prefix)]
        assert hit, prefix
        assert all(sp == 0 for _, sp in hit), (prefix, hit)
        return max(a for a, _ in hit)
    assert 2 * alloc("k_tower16<ConnectFour, 64, false, 11>") + alloc("k_tree<ConnectFour>") <= 512
    conv = max(alloc("k_conv16_layer<ConnectFour, 128, true"), alloc("k_conv16_layer<ConnectFour, 128, false"))
    wg = alloc("k_wgrad16<ConnectFour, 128, 0, 2, 48>")
    assert 2 * conv + wg <= 512, (conv, wg)
    assert 2 * wg + alloc("k_tr_colsum1v") <= 512 and 2 * wg + 2 * alloc("k_tr_bn_bwd<4>") <= 512
    assert conv + wg + alloc("k_wgrad_reduce") <= 512 + conv      # the reduction beside one convolution wavefront pair and the weight gradient


def test_engine_cache_policy(monkeypatch):
    """azhip.engine.cached_engine (host logic, no device): engines are kept per (role, az_engine_cfg bytes, creation-time
    environment), at most CACHE_MAX of them and CACHE_MAX_BYTES of device memory, least recently used first out."""
    from azhip import engine as E

    class Fake:
        made = []

        def __init__(self, cfg=None, **kw):
            self.cfg, self._h, self.closed = cfg, object(), False
            self.bytes = int(cfg.num_workers) << 20                # 1 MB per worker
            Fake.made.append(self)

        def device_bytes(self):
            return self.bytes

        def close(self):
            self._h, self.closed = None, True

    monkeypatch.setattr(E, "Engine", Fake)
    monkeypatch.setattr(E, "_cache", {})
    monkeypatch.setattr(E, "CACHE_MAX_BYTES", 1000 << 20)
    a = E.cached_engine("white", num_workers=100)
    assert E.cached_engine("white", num_workers=100) is a and len(Fake.made) == 1       # same role + configuration: reused
    b = E.cached_engine("black", num_workers=100)                                         # another role: its own engine
    c = E.cached_engine("white", num_workers=100, seed=7)                                 # any cfg byte differs: a new engine
    assert b is not a and c is not a and len(E._cache) == 3
    monkeypatch.setenv("AZHIP_TOWER", "3")
    d = E.cached_engine("white", num_workers=100)                                         # built under an override: not the same engine
    assert d is not a and len(E._cache) == 4
    monkeypatch.delenv("AZHIP_TOWER")
    assert E.cached_engine("white", num_workers=100) is a                                 # ... and `a` is now the most recently used
    e = E.cached_engine("selfplay", num_workers=200)                                      # a fifth: the least recently used (b) goes
    assert b.closed and not a.closed and len(E._cache) == E.CACHE_MAX == 4
    big = E.cached_engine("selfplay", num_workers=700)                                    # over the byte budget: oldest first until it fits
    assert c.closed and d.closed and not big.closed and E._cache_bytes() <= 1000 << 20
    assert sum(1 for f in Fake.made if not f.closed) == len(E._cache)
    a.close()                                                                              # a closed engine in the cache is rebuilt, not handed out
    assert E.cached_engine("white", num_workers=100) is not a
    E.clear_engine_cache()
    assert not E._cache and all(f.closed for f in Fake.made)
