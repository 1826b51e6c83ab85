"""The split tower (k_tower16s: two workgroups per board that exchange halves of every layer) must DEGRADE, not fail:
when a workgroup's partner is not co-resident -- another engine, a trainer, another process holds the CUs -- the bounded wait
gives up, and the engine then (i) notices without a synchronisation per wave (a host-mapped word), (ii) lets every queued launch
pass as a no-op (k_tree stands still and is counted), (iii) evaluates the pending leaves again with the unsplit kernel, (iv) replays
the waves that stood still, (v) keeps the split off, and reports it (az_selfplay_stats.tower_fallbacks).  The fault is injected
(AZHIP_XCH_FAIL_AT = n: the n-th split launch of the engine loses a partner; the wait is 50 ms of wall time).  Every record of the
phase must equal the undisturbed run's -- all tower kernels produce the same bits -- and the oracle's."""
import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def _records(games, moves, ng):
    return {games[i].game_id: [(tuple(moves[games[i].first_move + k].key), list(moves[games[i].first_move + k].N), moves[games[i].first_move + k].action)
                              for k in range(games[i].num_moves)] for i in range(ng)}


def _phase(monkeypatch, fail_at, workers, batch, ngames=6, nsims=24, cache=False):
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=2, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=7)
    if fail_at:
        monkeypatch.setenv("AZHIP_XCH_FAIL_AT", str(fail_at))
    else:
        monkeypatch.delenv("AZHIP_XCH_FAIL_AT", raising=False)
    # the injected fault is "the n-th split launch loses a workgroup": it is only FELT when that workgroup's board exists, i.e. when
    # the launch is full -- so every leaf must go to the network here (round 5: the evaluation cache would answer some of them)
    monkeypatch.setenv("AZHIP_EVAL_CACHE", "1" if cache else "0")
    with azhip.Engine(game=azhip.GAME_CONNECT_FOUR, oracle=azhip.ORACLE_RESNET, num_workers=workers, batch_size=batch, num_iters_per_turn=nsims,
                      cpuct=2.0, dirichlet_noise_eps=0.25, dirichlet_noise_alpha=1.0, temperature=((0,), (1.0,)), reset_every=1, seed=3,
                      num_blocks=2, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32) as e:
        e.net_set_params(blob)
        games, moves, ng, nm, st = e.selfplay_run(ngames)
        kernel = e.net_last_kernel()
    return _records(games, moves, ng), st, kernel, blob


@pytest.mark.parametrize("workers,batch,fail_at", [(6, 6, 9), (6, 3, 14), (6, 6, 1)])
def test_a_lost_partner_costs_two_seconds_not_the_phase(monkeypatch, workers, batch, fail_at):
    clean, st0, k0, blob = _phase(monkeypatch, 0, workers, batch)
    assert "k_tower16s" in k0 and st0.tower_fallbacks == 0          # the split form serves these launches
    hurt, st1, k1, _ = _phase(monkeypatch, fail_at, workers, batch)
    assert st1.tower_fallbacks == 1 and "k_tower16s" not in k1      # gave up once, unsplit from then on
    assert hurt == clean and st1.simulations == st0.simulations and st1.games == st0.games == 6
    # and both are the oracle's games
    rg, rm, _ = R.simulate(R.C4, R.ORACLE_NET, 6, workers, 24, cpuct=2.0, noise_eps=0.25, noise_alpha=1.0, temp_xs=(0,), temp_ys=(1.0,),
                           reset_every=1, seed=3, net=(2, 128, 32, 32, blob))
    assert clean == _records(rg, rm, 6)


def test_a_lost_partner_with_the_evaluation_cache_on(monkeypatch):
    """the same with the cache answering part of every wave: whether the lost workgroup's board exists is then up to the wave, so
    the fall-back may or may not engage -- the records must be the undisturbed run's either way (recover_split evaluates the
    pending network leaves again; leaves the cache answered keep their answers)"""
    clean, st0, _, _ = _phase(monkeypatch, 0, 6, 3, cache=False)
    for fail_at in (3, 14, 40):
        hurt, st1, _, _ = _phase(monkeypatch, fail_at, 6, 3, cache=True)
        assert hurt == clean and st1.simulations == st0.simulations and st1.leaf_evals == st0.leaf_evals and st1.tower_fallbacks in (0, 1)


def test_the_network_seam_retries_without_the_split(monkeypatch):
    """az_net_evaluate_keys (the Network seam, one launch at a time): the launch that loses its partner is repeated unsplit"""
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=2, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(azhip.GAME_CONNECT_FOUR, hp, seed=7)
    keys = np.array([[0, 0], [1, 1 << 63], [1 | (1 << 7), 0]], dtype=np.uint64)
    out = {}
    for fail_at in (0, 2):
        if fail_at:
            monkeypatch.setenv("AZHIP_XCH_FAIL_AT", str(fail_at))
        else:
            monkeypatch.delenv("AZHIP_XCH_FAIL_AT", raising=False)
        with azhip.Engine(game=0, oracle=azhip.ORACLE_RESNET, num_workers=4, batch_size=4, num_iters_per_turn=8, num_blocks=2, num_filters=128,
                          num_policy_head_filters=32, num_value_head_filters=32) as e:
            e.net_set_params(blob)
            a = e.net_evaluate_keys(keys)
            b = e.net_evaluate_keys(keys)                            # launch 2: the injected one
            c = e.net_evaluate_keys(keys)
            out[fail_at] = (a, b, c, e.net_last_kernel())
    for x, y in zip(out[0][:3], out[2][:3]):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    assert "k_tower16s" in out[0][3] and "k_tower16s" not in out[2][3]


def test_two_engines_count_together_for_co_residency():
    """an arena drives two engines side by side: whether pairs of workgroups are co-resident depends on BOTH engines' launches"""
    import azhip
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=1, num_filters=128, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(0, hp, seed=1)
    kw = dict(game=0, oracle=azhip.ORACLE_RESNET, num_workers=96, batch_size=96, num_iters_per_turn=4, num_blocks=1, num_filters=128,
              num_policy_head_filters=32, num_value_head_filters=32)
    def full_launch_kernel(e, other=None):                           # steady state: every one of the 96 slots has a leaf
        if other is not None:
            other.selfplay_begin(-1, 0)                              # the other engine has a search in progress
        e.selfplay_begin(-1, 0)
        e.selfplay_step(6)
        k = e.net_last_kernel()
        e.selfplay_end()
        if other is not None:
            other.selfplay_end()
        return k
    with azhip.Engine(**kw) as a:
        a.net_set_params(blob)
        alone = full_launch_kernel(a)                                # 2 x 96 workgroups fit 256 CUs
        with azhip.Engine(seed=9, **kw) as b:
            b.net_set_params(blob)
            idle = full_launch_kernel(b)                             # `a` exists but is idle (a host keeps engines cached): not counted
            together = full_launch_kernel(b, other=a)                # both searching: 2 engines x 2 x 96 workgroups do not fit
        again = full_launch_kernel(a)                                # the second engine is gone: split again
    assert "k_tower16s" in again and "k_tower16s" in idle
    assert "k_tower16s" in alone and "k_tower16s" not in together
