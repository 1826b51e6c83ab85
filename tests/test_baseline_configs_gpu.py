"""HIP engine vs the CPU oracle at the BASELINE.json configurations' REAL parameters.

The engines are created at the configurations' full slot counts and network sizes (C2: 4096 slots, 400 sims/move,
ResNet 5x64; C3: a rank's 4096-slot shard with 600 sims/move and global game ids; C4: Mancala, 8192 slots, 800
sims/move); a sample of whole games is compared move record by move record with the oracle's simulate (visit counts,
sampled action, reward, node counts, simulation / traversal counters -- all integers or exact floats).  RNG streams are
keyed by GLOBAL game id and trees reset after every game (reset_every = 1), so a game's trace does not depend on
how many slots run beside it (tests/test_selfplay_gpu.py::test_full_size_slot_count_independence): the sampled games
stand for every game of the full phase.  The oracle runs its workers on host threads (OpenMP), ~30 s per case."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu

C4_SCHED = ((0, 20, 30), (1.0, 1.0, 0.3))          # games/connect-four/params.jl:24-30, games/mancala/params.jl:23-29


def _run_case(game_hip, game_ref, slots, groups, nsims, ngames, first_id, sched, whole_phase, full_games=None):
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=5, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(game_hip, hp, seed=2026)
    with azhip.Engine(game=game_hip, oracle=azhip.ORACLE_RESNET, num_workers=slots, batch_size=slots // groups,
                      num_iters_per_turn=nsims, gamma=1.0, cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0,
                      prior_temperature=1.0, temperature=sched, reset_every=1, seed=1, num_blocks=5, num_filters=64,
                      num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        n_run = full_games if whole_phase else ngames
        games, moves, ng, nm, stats = e.selfplay_run(n_run, first_game_id=first_id)
        assert ng == n_run
        hip = []
        for i in range(ngames):
            g = games[i]
            hip.append((g.game_id, g.num_moves, g.nodes, g.total_simulations, g.total_nodes_traversed, tuple(g.final_key),
                        [(tuple(moves[g.first_move + k].key), list(moves[g.first_move + k].N), moves[g.first_move + k].action,
                          moves[g.first_move + k].reward) for k in range(g.num_moves)]))
    rg, rm, rnm = R.simulate(game_ref, R.ORACLE_NET, ngames, ngames, nsims, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0,
                             temp_xs=sched[0], temp_ys=sched[1], reset_every=1, seed=1, net=(5, 64, 32, 32, blob),
                             first_game_id=first_id)
    for i in range(ngames):
        g = rg[i]
        ref = (g.game_id, g.num_moves, g.nodes, g.total_simulations, g.total_nodes_traversed, tuple(g.final_key),
               [(tuple(rm[g.first_move + k].key), list(rm[g.first_move + k].N), rm[g.first_move + k].action,
                 rm[g.first_move + k].reward) for k in range(g.num_moves)])
        assert hip[i][:2] == ref[:2], (i, hip[i][:2], ref[:2])
        for k, (a, b) in enumerate(zip(hip[i][6], ref[6])):
            assert a == b, "game %d move %d: HIP %r != oracle %r" % (ref[0], k, a, b)
        assert hip[i][:6] == ref[:6], (i, hip[i][:6], ref[:6])
        assert sum(ref[6][0][1]) == nsims - 1                      # first move of a fresh tree: sum N = nsims - 1 (Appendix A.2)
    return stats


def test_config2_connect_four_4096_slots_400_sims():
    """BASELINE configs[1]: the WHOLE 4096-game phase runs on the device (4096 slots, two slot groups as bench.py
    does); games 0..7 are compared with the oracle."""
    import azhip
    st = _run_case(azhip.GAME_CONNECT_FOUR, R.C4, 4096, 2, 400, 8, 0, C4_SCHED, True, full_games=4096)
    assert st.games == 4096 and st.simulations == 400 * st.moves


def test_config3_rank_shard_600_sims():
    """BASELINE configs[2] as rank 3 of 8 sees it: 4096 slots, 600 sims/move, global game ids from 3 * 4096."""
    import azhip
    _run_case(azhip.GAME_CONNECT_FOUR, R.C4, 4096, 2, 600, 6, 3 * 4096, C4_SCHED, False)


def test_config4_mancala_8192_slots_800_sims():
    """BASELINE configs[3]: Mancala (variable action mask, free turns, bug-compatible flip_colors), 800 sims/move,
    an 8192-slot engine (games/mancala/params.jl:23-29: cpuct 2, eps 0.25, alpha 1, PLSchedule([0, 20, 30], [1, 1, 0.3]))."""
    import azhip
    _run_case(azhip.GAME_MANCALA, R.MANCALA, 8192, 2, 800, 6, 0, C4_SCHED, False)


def test_config1_tictactoe_32_games_64_sims():
    """BASELINE configs[0] (the reference's own CPU-runnable case: Tic-tac-toe, 64 sims/move, 32 parallel games) on the device:
    ALL 32 games of a 32-slot engine against the oracle, record by record (search constants as in the other cases)."""
    import azhip
    st = _run_case(azhip.GAME_TICTACTOE, R.TTT, 32, 1, 64, 32, 0, ((0,), (1.0,)), True, full_games=32)
    assert st.games == 32 and st.simulations == 64 * st.moves
