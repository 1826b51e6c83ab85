"""Free-running self-play phases (round 6; az_engine_cfg.lock_step = 0, csrc/tree.h k_tree / k_move_fr).

A worker of the reference runs its simulations, its moves and its games at its own pace: `explore!` is a sequential loop
(src/mcts.jl:239-245), `play_game` one move after the other (src/play.jl:298-315), and the next game id goes to whichever worker asks
first (src/util.jl:181-188).  A free-running slot does the same: inside one launch of the tree kernel it completes every simulation
that ends on a terminal state or on a state the evaluation cache answers, it moves when ITS explore! is complete, and it takes its
next game from an atomic counter.  What must hold:
  * every record of a game is the lock-step schedule's (and the oracle's) whenever the tree is reset after every game -- a game then
    depends on its id alone;
  * with trees kept over several games the records are the oracle's for the worker -> game assignment the device reports;
  * the statistics are conserved; the knobs (simulations per launch, waves between two looks of the host) change no record."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu

SCHED = ((0, 6, 12), (1.0, 1.0, 0.3))


def _by_id(games, moves, ng, cumulative=False):
    """records by game id; `cumulative`: with the per-worker counters (they depend on which games the worker played before)"""
    out = {}
    for i in range(ng):
        g = games[i]
        head = (g.num_moves, g.nodes, tuple(g.final_key)) + ((g.slot, g.total_simulations, g.total_nodes_traversed) if cumulative else ())
        out[g.game_id] = (head, [bytes(moves[g.first_move + k]) for k in range(g.num_moves)])
    return out


def _resnet_kw(F=64, blocks=2):
    return dict(num_blocks=blocks, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32)


@pytest.mark.parametrize("game_name,workers,batch,games,nsims", [("c4", 256, 128, 1024, 100), ("ttt", 64, 64, 512, 48), ("mancala", 96, 48, 256, 80)])
def test_free_running_and_lock_step_play_the_same_games(monkeypatch, game_name, workers, batch, games, nsims):
    """ResNet in the loop, evaluation cache on, four games per slot: every game equals the lock-step run's game of the same id, for
    several values of the two scheduling knobs; the free-running runs really ran ahead (more than one simulation per slot and launch)."""
    import azhip
    from azhip.network import ResNetHP, random_params
    gh = {"c4": azhip.GAME_CONNECT_FOUR, "ttt": azhip.GAME_TICTACTOE, "mancala": azhip.GAME_MANCALA}[game_name]
    hp = ResNetHP(**_resnet_kw())
    blob = random_params(gh, hp, seed=11)
    out, stats = {}, {}
    for mode, env in (("lock", {}), ("free", {}), ("free k=1", {"AZHIP_RUN_K": "1"}), ("free k=64 round=7", {"AZHIP_RUN_K": "64", "AZHIP_FR_ROUND": "7"})):
        for k in ("AZHIP_RUN_K", "AZHIP_FR_ROUND"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with azhip.Engine(game=gh, oracle=azhip.ORACLE_RESNET, num_workers=workers, batch_size=batch, num_iters_per_turn=nsims, cpuct=2.0,
                          dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=1, seed=5,
                          lock_step=1 if mode == "lock" else 0, max_moves_per_game=200 if game_name == "mancala" else 0, **_resnet_kw()) as e:
            e.net_set_params(blob)
            g, m, ng, nm, st = e.selfplay_run(games)
            assert ng == games and st.aborted_games == 0 and st.games == games and st.moves == nm
            assert st.simulations == nsims * nm                      # every explore! is num_iters_per_turn simulations, whatever the schedule
            out[mode], stats[mode] = _by_id(g, m, ng), st
    for mode in out:
        assert out[mode] == out["lock"], mode
        assert (stats[mode].simulations, stats[mode].nodes_traversed, stats[mode].leaf_evals) == \
               (stats["lock"].simulations, stats["lock"].nodes_traversed, stats["lock"].leaf_evals), mode
    assert stats["lock"].simulations == stats["lock"].slot_launches or stats["lock"].slot_launches >= stats["lock"].simulations   # one simulation per slot and wave (idle slots of the drain counted)
    assert stats["free"].waves < 0.8 * stats["lock"].waves           # the same simulations in fewer launches
    assert stats["free k=64 round=7"].waves <= stats["free"].waves


@pytest.mark.parametrize("reset_every,flip", [(2, 0.5), (0, 0.0), (3, 1.0)])
def test_trees_kept_over_several_games_follow_the_reported_assignment(reset_every, flip):
    """reset_every != 1: which games share a tree is the outcome of the id race (util.jl:181-188).  The device reports the one it took
    (az_game_rec.slot); the oracle, given that assignment, produces every record -- the per-worker counters included."""
    import azhip
    with azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_HASH, num_workers=48, batch_size=24, num_iters_per_turn=90, cpuct=2.0,
                      dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=reset_every, flip_probability=flip, seed=9,
                      max_nodes_per_slot=90 * 42 * 8 if reset_every == 0 else 0) as e:
        g, m, ng, nm, st = e.selfplay_run(240)
        assert ng == 240 and st.aborted_games == 0
        dev = _by_id(g, m, ng, cumulative=True)
        asg = R.assignment_of(g, 240)
    per_slot = np.bincount(asg, minlength=48)
    assert per_slot.sum() == 240 and per_slot.min() >= 1
    rg, rm, rnm = R.simulate(R.C4, R.ORACLE_HASH, 240, 48, 90, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=SCHED[0], temp_ys=SCHED[1],
                             reset_every=reset_every, seed=9, flip_probability=flip, assignment=asg)
    assert rnm == nm and dev == _by_id(rg, rm, 240, cumulative=True)
    # a worker plays its games in increasing id order (its next id is always the counter's next value)
    for s in range(48):
        ids = [i for i in range(240) if asg[i] == s]
        assert ids == sorted(ids)


def test_stepping_form_collects_finished_games_and_survives_a_full_staging_area():
    """az_selfplay_begin(-1) / step / collect: an unbounded phase keeps its finished games in a staging area of one game per slot that
    the host drains at every look; Tic-tac-toe games of 8 simulations per move finish faster than that when the host looks rarely --
    the slots then wait for room with their last move unplayed.  Every collected game is the bounded run's game of the same id."""
    import azhip
    kw = dict(game=azhip.GAME_TICTACTOE, oracle=azhip.ORACLE_HASH, num_workers=32, batch_size=32, num_iters_per_turn=8, cpuct=1.5,
              dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=((0,), (1.0,)), reset_every=1, seed=13)
    with azhip.Engine(**kw) as e:
        g0, m0, n0, _, _ = e.selfplay_run(600)
        want = _by_id(g0, m0, n0)
        e.selfplay_begin(-1, 0)
        got, collected = {}, 0
        for _ in range(6):
            e.selfplay_step(100)                                     # ~100 waves between two looks: more games than the staging area holds
            games, moves, ng, nm = e.selfplay_collect(4096)
            got.update(_by_id(games, moves, ng))
            collected += ng
        st = e.selfplay_stats()
        e.selfplay_end()
    assert collected == len(got) >= 64 and st.games == collected
    for gid, rec in got.items():
        if gid in want:
            assert rec == want[gid], gid
    assert sum(1 for gid in got if gid in want) >= 64


def test_retired_slots_get_their_replacement_in_a_free_running_phase():
    """a tight node pool: slots retire inside k_tree, the host hands out the replacement games at its next look, the phase returns
    what it owes; the games that completed are the unbounded run's"""
    import azhip
    kw = dict(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_UNIFORM, num_workers=8, batch_size=4, num_iters_per_turn=32, cpuct=2.0,
              dirichlet_noise_eps=0.25, reset_every=1, seed=11)
    with azhip.Engine(**kw) as full:
        g0, m0, n0, _, st0 = full.selfplay_run(40)
    want = _by_id(g0, m0, n0)
    cap = int(sorted(g0[i].nodes for i in range(n0))[-6])
    with azhip.Engine(max_nodes_per_slot=cap, **kw) as e:
        g, m, ng, nm, st = e.selfplay_run(40)
        aborted = e.selfplay_aborted()
    BIT = 0x40000000
    given_up = [a for a in aborted if a & BIT]
    assert st.aborted_games == len(aborted) >= 1 and ng == 40 - len(given_up)
    got = _by_id(g, m, ng)
    for gid, rec in got.items():
        if not gid & BIT:
            assert rec == want[gid], gid


def test_the_mixed_8_and_7_board_tower_launch_changes_no_record(monkeypatch):
    """4096 slots in ONE group: a free-running wave's batch (~3800 boards) CAN be served by k_tower16x2m -- 256 workgroups of 8 boards and up
    to 256 of 7 in one launch (csrc/resnet16.h; measured slower than two rounds of 8-board workgroups and off by default: AZHIP_TOWER_MIXED=1).
    Every fp32 tower form computes the same bits, so the phase's records must be those of the plain 8-board launches and of the lock-step
    schedule; and the mixed launch must really have been in use."""
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(**_resnet_kw(blocks=1))
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=3)
    kw = dict(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_workers=4096, batch_size=4096, num_iters_per_turn=48, cpuct=2.0,
              dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=1, seed=17, **_resnet_kw(blocks=1))
    out = {}
    for mode in ("mixed", "plain", "lock"):
        monkeypatch.setenv("AZHIP_TOWER_MIXED", "0" if mode == "plain" else "1")
        with azhip.Engine(lock_step=1 if mode == "lock" else 0, **kw) as e:
            e.net_set_params(blob)
            if mode == "mixed":
                e.selfplay_begin(-1, 0)
                e.selfplay_step(150)
                seen = e.net_last_kernel()
                e.selfplay_end()
                assert seen.startswith("k_tower16x2m<"), seen
            g, m, ng, nm, st = e.selfplay_run(6000)
            assert ng == 6000 and st.aborted_games == 0
            out[mode] = _by_id(g, m, ng)
    assert out["mixed"] == out["plain"] == out["lock"]


def test_the_176_register_paired_tower_changes_no_record(monkeypatch):
    """4096 slots in ONE group: where the background search's workgroups would cost the 198-register paired tower a round of workgroups
    (a k_tree wavefront does not fit beside its two on a SIMD), wave_net_f serves the wave with k_tower16x2c, the same kernel body
    within 176 registers (csrc/resnet16.h, net_impl.h; profiles/r6/README.md §12).  The phase's records must not depend on which
    waves that happens in: the engine's own choice, the capped form forced for every launch (AZHIP_TOWER=20) and the lock-step
    schedule give the same games."""
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(**_resnet_kw(blocks=1))
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=5)
    kw = dict(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_workers=4096, batch_size=4096, num_iters_per_turn=48, cpuct=2.0,
              dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=SCHED, reset_every=1, seed=23, **_resnet_kw(blocks=1))
    out, kernels = {}, set()
    for mode in ("own", "capped", "lock"):
        if mode == "capped":
            monkeypatch.setenv("AZHIP_TOWER", "20")
        else:
            monkeypatch.delenv("AZHIP_TOWER", raising=False)
        with azhip.Engine(lock_step=1 if mode == "lock" else 0, **kw) as e:
            e.net_set_params(blob)
            if mode != "lock":
                e.selfplay_begin(-1, 0)
                for _ in range(12):
                    e.selfplay_step(25)
                    kernels.add((mode, e.net_last_kernel().split("<")[0]))
                e.selfplay_end()
            g, m, ng, nm, st = e.selfplay_run(6000)
            assert ng == 6000 and st.aborted_games == 0
            out[mode] = _by_id(g, m, ng)
    assert out["own"] == out["capped"] == out["lock"]
    assert ("capped", "k_tower16x2c") in kernels and {k for md, k in kernels if md == "own"} <= {"k_tower16x2", "k_tower16x2c", "k_tower16x2m", "k_tower16", "k_tower"}, kernels   # (16x2m: the mixed launch, if an earlier test of this process switched it on)
