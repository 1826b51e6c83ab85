"""Self-play parity with the ResNet oracle in the loop (GPU) and full-size properties.

Because the HIP network is bit-identical to the oracle's fp32 chain (tests/test_net.py), whole
self-play phases can be compared record by record: visit counts, actions, rewards, node counts."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def _hp(nblocks, F=64):
    from azhip import ResNetHP
    return ResNetHP(num_blocks=nblocks, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32)


@pytest.mark.parametrize("game,spec,ngames,workers,batch,nsims,F,tower", [
    (R.C4, "ConnectFourSpec", 12, 6, 6, 40, 64, "16"), (R.TTT, "TicTacToeSpec", 10, 4, 4, 24, 64, ""),
    (R.MANCALA, "MancalaSpec", 6, 3, 3, 24, 64, "32"), (R.C4, "ConnectFourSpec", 12, 6, 3, 40, 64, ""),
    (R.C4, "ConnectFourSpec", 6, 4, 2, 24, 128, "16"), (R.C4, "ConnectFourSpec", 6, 4, 2, 24, 128, ""),
    (R.C4, "ConnectFourSpec", 24, 24, 24, 12, 128, "")])     # 24 workers in one group: k_tower16s pairs + split k_heads16 tiles in the wave path
def test_simulate_with_resnet_matches_oracle(game, spec, ngames, workers, batch, nsims, F, tower, monkeypatch):
    """Simulator / simulate (simulations.jl:179-244) with MctsPlayer + ResNet, vs the oracle's simulate.
    tower: AZHIP_TOWER override ("" = the engine's own choice, which is the 3-row-tile k_tower16 at these sizes)."""
    import azhip
    if tower:
        monkeypatch.setenv("AZHIP_TOWER", tower)
    gspec = getattr(azhip, spec)()
    hp = _hp(2, F)
    nn = azhip.ResNet(gspec, hp, seed=11)
    mp = azhip.MctsParams(num_iters_per_turn=nsims, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0, cpuct=2.0,
                          temperature=azhip.PLSchedule([0, 6, 10], [1.0, 1.0, 0.3]), gamma=1.0)
    sp = azhip.SimParams(num_games=ngames, num_workers=workers, batch_size=batch, use_gpu=True, reset_every=2)
    from azhip.network import copy as netcopy
    sim = azhip.Simulator(lambda oracle: azhip.MctsPlayer(gspec, oracle, mp), lambda: netcopy(nn, on_gpu=True, test_mode=True),
                          azhip.self_play_measurements)
    count = [0]
    res = azhip.simulate(sim, gspec, sp, game_simulated=lambda: count.__setitem__(0, count[0] + 1), seed=5)
    assert count[0] == ngames == len(res)
    # which worker plays which game is a race in the reference (util.jl:181-188); the oracle replays the outcome the device took
    games, moves, nm = R.simulate(game, R.ORACLE_NET, ngames, workers, nsims, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0,
                                  temp_xs=(0, 6, 10), temp_ys=(1.0, 1.0, 0.3), reset_every=2, seed=5,
                                  net=(2, F, 32, 32, nn.params()), assignment=[r["worker"] for r in res])
    from azhip.trace import trace_from_records
    for i in range(ngames):
        t = res[i]["trace"]
        ref = trace_from_records(games[i], moves, R.NUM_ACTIONS[game],
                                 lambda key: R.Game(game, R.unpack_key(game, key)).actions_mask())
        assert t.states == ref.states and t.rewards == ref.rewards, i
        assert all(np.array_equal(a, b) for a, b in zip(t.policies, ref.policies)), i
        g = games[i]
        assert res[i]["edepth"] == g.total_nodes_traversed / g.total_simulations


def test_simulate_with_flips_and_the_network_matches_oracle():
    """simulate with SimParams.flip_probability = 0.5 (play.jl:305-307) and the ResNet in the loop, through the Python mirror:
    Trace.states are the un-flipped states, Trace.policies the image's policy vectors (by rank), as in the reference's traces."""
    import azhip
    from azhip.network import copy as netcopy
    from azhip.trace import trace_from_records
    gspec = azhip.ConnectFourSpec()
    nn = azhip.ResNet(gspec, _hp(2), seed=11)
    mp = azhip.MctsParams(num_iters_per_turn=32, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0, cpuct=2.0,
                          temperature=azhip.PLSchedule([0, 6, 10], [1.0, 1.0, 0.3]), gamma=1.0)
    sp = azhip.SimParams(num_games=8, num_workers=4, batch_size=4, use_gpu=True, reset_every=2, flip_probability=0.5)
    sim = azhip.Simulator(lambda oracle: azhip.MctsPlayer(gspec, oracle, mp), lambda: netcopy(nn, on_gpu=True, test_mode=True),
                          azhip.self_play_measurements)
    res = azhip.simulate(sim, gspec, sp, seed=5)
    games, moves, nm = R.simulate(R.C4, R.ORACLE_NET, 8, 4, 32, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=(0, 6, 10),
                                  temp_ys=(1.0, 1.0, 0.3), reset_every=2, seed=5, net=(2, 64, 32, 32, nn.params()), flip_probability=0.5,
                                  assignment=[r["worker"] for r in res])
    assert 0 < sum(1 for k in range(nm) if moves[k].N[R.AMAX]) < nm
    for i in range(8):
        t = res[i]["trace"]
        ref = trace_from_records(games[i], moves, 7, lambda key: R.Game(R.C4, R.unpack_key(R.C4, key)).actions_mask())
        assert t.states == ref.states and t.rewards == ref.rewards, i
        assert all(np.array_equal(a, b) for a, b in zip(t.policies, ref.policies)), i
        assert res[i]["edepth"] == games[i].total_nodes_traversed / games[i].total_simulations


def test_mcts_env_and_play_game_mirror():
    """MCTS.Env explore!/policy (mcts.jl:239-271) with the ResNet, vs the oracle; play_game runs to the end."""
    import azhip
    gspec = azhip.ConnectFourSpec()
    nn = azhip.ResNet(gspec, _hp(1), seed=3)
    env = azhip.MCTS.Env(gspec, nn, cpuct=2.0, noise_ϵ=0.25, noise_α=1.0)
    g = gspec.init()
    for a in (4, 4, 3):
        g.play(a)
    eta = np.array([0.1, 0.2, 0.05, 0.05, 0.3, 0.2, 0.1])
    env.explore(g, 120, eta=eta)
    acts, pi = env.policy(g)
    og = R.Game(R.C4)
    for a in (3, 3, 2):
        og.play(a)
    m = R.Mcts(R.C4, oracle=R.ORACLE_NET, cpuct=2.0, noise_eps=0.25, net=(1, 64, 32, 32, nn.params()))
    m.explore(og, 120, eta=eta)
    oacts, opi = m.policy(og)
    assert acts == [a + 1 for a in oacts] and np.array_equal(pi, opi)
    N, W, P, V = env.tree_stats(g.current_state())
    oN, oW, oP, oV = m.root_stats(og)
    assert np.array_equal(N, oN) and np.array_equal(W, oW) and np.array_equal(P, oP) and V == oV
    assert env.total_simulations == 120 and env.average_exploration_depth() == m.total_nodes_traversed / 120
    env.reset()
    with pytest.raises(azhip.AzError):
        env.policy(g)                                   # "MCTS.explore! must be called before MCTS.policy"
    player = azhip.MctsPlayer(gspec, azhip.MCTS.RandomOracle(gspec), azhip.MctsParams(32, 0.25, 1.0, cpuct=2.0))
    t = azhip.play_game(gspec, player)
    assert t.valid() and 7 <= len(t) <= 42 and t.rewards[-1] in (-1.0, 0.0, 1.0)
    p, v = nn.evaluate(g.current_state())
    assert len(p) == 7 and abs(p.sum() - 1) < 1e-5 and -1 <= v <= 1


def test_timeout_search_and_flips_in_the_host_stepped_play_game():
    """think with a timeout (play.jl:199-204): whole explore! calls until the wall clock says stop, so the tree holds a
    positive multiple of num_iters_per_turn simulations and timeout = 0 leaves the tree empty (MCTS.policy then errors
    like the reference).  play_game's flip_probability (play.jl:305-307) on the host-stepped loop: the trace stays a
    valid game (every state is the image-or-not successor of the one before)."""
    import azhip
    gspec = azhip.ConnectFourSpec()
    mp = azhip.MctsParams(16, 0.25, 1.0, cpuct=2.0)
    player = azhip.MctsPlayer(gspec, azhip.MCTS.RandomOracle(gspec), mp, timeout=0.05)
    g = gspec.init()
    acts, pi = player.think(g)
    n = player.mcts.total_simulations
    assert n >= 16 and n % 16 == 0 and abs(pi.sum() - 1) < 1e-12 and len(acts) == 7
    lazy = azhip.MctsPlayer(gspec, azhip.MCTS.RandomOracle(gspec), mp, timeout=0.0)
    with pytest.raises(azhip.AzError):
        lazy.think(g)
    with pytest.raises(ValueError):
        azhip.MctsPlayer(gspec, azhip.MCTS.RandomOracle(gspec), mp, timeout=-1.0)
    fixed = azhip.MctsPlayer(gspec, azhip.MCTS.RandomOracle(gspec), mp)
    t = azhip.play_game(gspec, fixed, flip_probability=0.5, rng=np.random.default_rng(4))
    assert 7 <= len(t) <= 42 and t.rewards[-1] in (-1.0, 0.0, 1.0)
    mirror = lambda s: gspec.symmetries(s)[0][0]

    def succ(state):
        out = set()
        for a in gspec.init(state).available_actions():
            e = gspec.init(state)
            e.play(a)
            out.add(e.current_state())
        return out
    flips = 0
    for i in range(len(t)):
        plain, image = succ(t.states[i]), succ(mirror(t.states[i]))
        assert t.states[i + 1] in plain | image, i
        flips += t.states[i + 1] not in plain
    assert flips > 0                                                # some successor is reachable only through the image


@pytest.mark.parametrize("flip", [0.0, 0.5])
def test_full_size_slot_count_independence(flip):
    """BASELINE configs[1] size: 4096 slots, 400 sims/move, ResNet 5x64 (flip 0.5: with play_game's random symmetries, whose
    draws are keyed by game id and move like everything else).  Size-independent properties:
    (i) two runs are identical (determinism); (ii) game g's trace does not depend on how many slots run
    beside it: games 0..47 of the 4096-slot run equal a 48-slot run (RNG keyed by game id, reset_every 1);
    (iii) conservation: sum of root visits = sims - 1 on the first move, every policy sums to 1,
    simulations = waves x active slots."""
    import azhip
    from azhip.network import random_params
    blob = random_params(azhip.GAME_CONNECT_FOUR, _hp(5), seed=2026)
    kw = dict(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_iters_per_turn=400, cpuct=2.0,
              dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=((0, 20, 30), (1.0, 1.0, 0.3)),
              reset_every=1, seed=1, num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32, flip_probability=flip,
              lock_step=1)       # "400 waves = one move of every slot" is the lock-step schedule; the free-running one: test_free_running_gpu.py

    def first_moves(G, nmoves):
        with azhip.Engine(num_workers=G, batch_size=G, **kw) as e:
            e.net_set_params(blob)
            e.selfplay_begin(-1, 0)
            recs = []
            e.selfplay_step(400 * nmoves)
            st = e.selfplay_stats()
            # roots after nmoves moves are in the per-slot trace; read them through the node stats hook
            out = []
            for s in range(48):
                out.append(e.mcts_counters(s))
            e.selfplay_end()
            return out, st

    a, sa = first_moves(4096, 2)
    b, sb = first_moves(4096, 2)
    c, sc = first_moves(48, 2)
    assert a == b and (sa.simulations, sa.nodes_traversed, sa.leaf_evals) == (sb.simulations, sb.nodes_traversed, sb.leaf_evals)
    assert a == c                                            # per-slot sims, traversed, nodes identical for games 0..47
    assert sa.simulations == 4096 * 800 and sc.simulations == 48 * 800 and sa.moves == 2 * 4096
    assert sa.leaf_evals <= sa.simulations and all(x[0] == 800 for x in a)


def test_self_play_step_report():
    """self_play_step! (training.jl:275-300): traces pushed into the memory, Report.SelfPlay fields."""
    import azhip
    gspec = azhip.TicTacToeSpec()
    nn = azhip.ResNet(gspec, _hp(1), seed=4)
    params = azhip.SelfPlayParams(
        mcts=azhip.MctsParams(num_iters_per_turn=16, dirichlet_noise_ϵ=0.25, dirichlet_noise_α=1.0, cpuct=1.0),
        sim=azhip.SimParams(num_games=12, num_workers=4, batch_size=2, use_gpu=True, reset_every=1))
    mem, played = [], [0]
    rep = azhip.self_play_step(gspec, nn, params, mem, game_played=lambda: played.__setitem__(0, played[0] + 1))
    assert played[0] == 12 and rep.memory_size == len(mem) >= 12 * 5
    assert rep.samples_gen_speed > 0 and 0 < rep.average_exploration_depth < 9 and rep.mcts_memory_footprint > 0
    assert 1 <= rep.memory_num_distinct_boards <= rep.memory_size
    assert all(abs(s.z) <= 1 and s.t >= 1 and abs(s.π.sum() - 1) < 1e-12 for s in mem)


def test_shards_by_first_game_id_equal_the_unsharded_run():
    """a rank simulates a contiguous range of GLOBAL game ids (first_game_id); the union of the
    shards is the single-engine run, whatever the number of ranks (reset_every = 1)."""
    import azhip
    from azhip.network import random_params
    hp = _hp(1)
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=8)
    kw = dict(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_workers=6, batch_size=3, num_iters_per_turn=24,
              cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=((0, 8), (1.0, 0.4)), reset_every=1,
              seed=9, num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)

    def run(first, count, workers):
        k = dict(kw, num_workers=workers, batch_size=workers)
        with azhip.Engine(**k) as e:
            e.net_set_params(blob)
            g, m, ng, nm, _ = e.selfplay_run(count, first_game_id=first)
            return [(g[i].game_id, [(tuple(m[g[i].first_move + k].key), list(m[g[i].first_move + k].N), m[g[i].first_move + k].action)
                                    for k in range(g[i].num_moves)]) for i in range(ng)]
    whole = run(0, 11, 6)
    from azhip.simulations import shard_games
    parts = []
    for r in range(3):
        first, count = shard_games(11, 3, r)
        parts += run(first, count, 4)
    assert [g for g, _ in whole] == list(range(11)) == [g for g, _ in parts]
    assert whole == parts


def test_very_large_engine_plays_the_same_games_and_conserves_its_counters():
    """65 536 Connect-Four slots (16 x the BASELINE batch, 2048 tree-kernel workgroups): the statistics are accumulated per workgroup
    (three same-address atomics per workgroup and wave were what bounded k_tree at this scale) and summed by the host.  A game's
    trace depends on its id alone, so games 0..11 must be the oracle's games; the counters must be conserved over the whole phase."""
    import azhip
    kw = dict(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_HASH, num_iters_per_turn=12, cpuct=2.0, dirichlet_noise_eps=0.25,
              dirichlet_noise_alpha=1.0, temperature=((0, 6), (1.0, 0.5)), reset_every=1, seed=21)
    with azhip.Engine(num_workers=65536, batch_size=65536, **kw) as e:
        g, m, ng, nm, st = e.selfplay_run(65536)
    assert ng == 65536 and st.simulations == 12 * st.moves == 12 * nm and st.aborted_games == 0
    assert st.nodes_traversed == sum(g[i].total_nodes_traversed for i in range(ng))
    rg, rm, _ = R.simulate(R.C4, R.ORACLE_HASH, 12, 12, 12, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=(0, 6), temp_ys=(1.0, 0.5),
                           reset_every=1, seed=21)
    for i in range(12):
        assert (g[i].game_id, g[i].num_moves, g[i].nodes) == (rg[i].game_id, rg[i].num_moves, rg[i].nodes), i
        for k in range(g[i].num_moves):
            a, b = m[g[i].first_move + k], rm[rg[i].first_move + k]
            assert tuple(a.key) == tuple(b.key) and list(a.N) == list(b.N) and a.action == b.action, (i, k)
