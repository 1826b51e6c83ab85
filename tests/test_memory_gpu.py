"""Device replay memory + learning status (GPU) against the oracle: az_memory_* / az_dataset_* / az_learning_status
(src/memory.jl:20-138, src/learning.jl:17-121,148-190).  Samples (Float64 averages included) and the Float32
tensors must be identical; the loss figures agree to Float32 rounding of differently ordered Float64 sums."""
import ctypes as C

import numpy as np
import pytest

import azref as R

pytestmark = pytest.mark.gpu


def _selfplay(game, ngames, workers, nsims, seed):
    import azhip
    with azhip.Engine(game=game, oracle=azhip.ORACLE_HASH, num_workers=workers, batch_size=workers, num_iters_per_turn=nsims,
                      dirichlet_noise_eps=0.25, cpuct=1.0, reset_every=1, temperature=([0], [1.0]), seed=seed,
                      max_moves_per_game=200 if game == 2 else 0) as e:
        return e.selfplay_run(ngames)


def _oracle_samples(game, games, moves, ng, gamma):
    S = []
    for i in range(ng):
        g = games[i]
        arr = (R.MoveRec * g.num_moves)()
        for k in range(g.num_moves):
            m = moves[g.first_move + k]
            arr[k].key[0], arr[k].key[1] = m.key[0], m.key[1]
            for a in range(10):
                arr[k].N[a] = m.N[a]
            arr[k].action, arr[k].reward = m.action, m.reward
        ss = R.samples_from_trace(game, arr, 0, g.num_moves, gamma)
        S += [ss[k] for k in range(g.num_moves)]
    return S


def _same_samples(dev, ref, nA):
    assert len(dev) == len(ref)
    for a, b in zip(dev, ref):
        assert (a.key[0], a.key[1]) == (b.key[0], b.key[1])
        assert list(a.pi[:nA]) == list(b.pi[:nA]) and (a.z, a.t, a.n) == (b.z, b.t, b.n)


SPECS = {0: "ConnectFourSpec", 1: "TicTacToeSpec", 2: "MancalaSpec"}


@pytest.mark.parametrize("game", [0, 1, 2])
def test_push_and_experience_match_oracle(game):
    import azhip
    gspec = getattr(azhip, SPECS[game])()
    nA = R.NUM_ACTIONS[game]
    games, moves, ng, nm, _ = _selfplay(game, 10, 5, 24, 5)
    ref = _oracle_samples(game, games, moves, ng, 0.95)
    mem = azhip.MemoryBuffer(gspec, 10000)
    mem.push_records(games, moves, ng, nm, 0.95)
    assert len(mem) == nm == mem.cur_batch_size()
    with mem.dataset() as d:
        _same_samples(d.raw_samples(), ref, nA)
    # symmetries + merge + weights: samples and tensors identical to the oracle's
    use_sym = game != 2
    aug = R.augment_with_symmetries(game, ref) if use_sym else ref
    merged = R.merge_by_state(game, aug)
    for policy in (0, 1, 2):
        with mem.dataset(use_symmetries=use_sym, use_position_averaging=True, weighing_policy=policy) as d:
            _same_samples(d.raw_samples(), merged, nA)
            W, X, A, P, V = d.tensors()
            Wr, Xr, Ar, Pr, Vr = R.convert_samples(game, policy, merged)
            assert all(np.array_equal(x, y) for x, y in ((W, Wr), (X, Xr), (A, Ar), (P, Pr), (V, Vr)))
            assert d.sum_n == len(aug) and abs(d.Wtot - float(Wr.astype(np.float64).sum())) < 1e-9 * max(1.0, d.Wtot)
            assert abs(d.Wmean - Wr.astype(np.float64).mean()) < 1e-6
    with mem.dataset(use_symmetries=use_sym) as d:                    # no merge: [samples ; images]
        _same_samples(d.raw_samples(), aug, nA)
    samples = mem.get_experience()
    assert len(samples) == nm and samples[0].s == (ref[0].key[0], ref[0].key[1]) and samples[0].n == 1
    mem.close()


@pytest.mark.parametrize("game", [0, 1])
def test_flipped_self_play_reaches_the_data_set_like_the_oracles(game):
    """play_game's random symmetries (play.jl:305-307) end to end on both sides, independently: device self-play with
    flip_probability 0.5 -> az_memory_push -> symmetries + merge + tensors, against the oracle's own flipped simulate ->
    push_trace! -> augment_with_symmetries -> merge_by_state -> convert_samples.  The samples pair the un-flipped state with the
    image's policy by rank, as the reference's do (learning.jl:31-33)."""
    import azhip
    gspec = getattr(azhip, SPECS[game])()
    nA = R.NUM_ACTIONS[game]
    with azhip.Engine(game=game, oracle=azhip.ORACLE_HASH, num_workers=5, batch_size=5, num_iters_per_turn=24, dirichlet_noise_eps=0.25, cpuct=1.0,
                      reset_every=2, temperature=([0], [1.0]), seed=5, flip_probability=0.5) as e:
        games, moves, ng, nm, _ = e.selfplay_run(10)
    rg, rm, rnm = R.simulate(game, R.ORACLE_HASH, 10, 5, 24, cpuct=1.0, noise_eps=0.25, reset_every=2, seed=5, flip_probability=0.5,
                              assignment=R.assignment_of(games, 10))
    assert rnm == nm and any(rm[k].N[R.AMAX] for k in range(rnm))
    ref = _oracle_samples(game, rg, rm, 10, 0.95)
    mem = azhip.MemoryBuffer(gspec, 10000)
    mem.push_records(games, moves, ng, nm, 0.95)
    with mem.dataset() as d:
        _same_samples(d.raw_samples(), ref, nA)
    ref2 = R.merge_by_state(game, R.augment_with_symmetries(game, ref))
    with mem.dataset(use_symmetries=True, use_position_averaging=True, weighing_policy=azhip.LOG_WEIGHT) as d:
        _same_samples(d.raw_samples(), ref2, nA)
        for a, b in zip(d.tensors(), R.convert_samples(game, 1, ref2)):
            assert np.array_equal(a, b)
    mem.close()


def test_long_segments_merge_in_buffer_order():
    """600 short games: the opening positions (x8 symmetric images that coincide) form segments of thousands of
    samples, which take the wavefront-parallel merge kernel; the averages must still be the sequential ones."""
    import azhip
    gspec = azhip.TicTacToeSpec()
    games, moves, ng, nm, _ = _selfplay(1, 600, 64, 8, 3)
    ref = R.merge_by_state(1, R.augment_with_symmetries(1, _oracle_samples(1, games, moves, ng, 0.9)))
    mem = azhip.MemoryBuffer(gspec, 100000)
    mem.push_records(games, moves, ng, nm, 0.9)
    with mem.dataset(use_symmetries=True, use_position_averaging=True, weighing_policy=azhip.LOG_WEIGHT) as d:
        _same_samples(d.raw_samples(), ref, 9)
        assert max(s.n for s in ref) >= 4800 and sum(1 for s in ref if s.n >= 256) >= 3
        assert np.array_equal(d.tensors()[0], R.convert_samples(1, 1, ref)[0])
    mem.close()


def test_large_data_set_sorts_scans_and_sums_across_many_tiles():
    """merge_by_state on ~100 k augmented samples: the hand-written radix sort (csrc/prims.h: 8 passes over ~50 tiles, two stable
    sorts for the 128-bit key), the multi-level scan and the tiled double sum, against the oracle's sequential merge"""
    import azhip
    gspec = azhip.ConnectFourSpec()
    games, moves, ng, nm, _ = _selfplay(0, 2500, 512, 8, 9)
    ref = R.merge_by_state(0, R.augment_with_symmetries(0, _oracle_samples(0, games, moves, ng, 1.0)))
    assert 2 * nm > 80000
    mem = azhip.MemoryBuffer(gspec, 200000)
    mem.push_records(games, moves, ng, nm, 1.0)
    with mem.dataset(use_symmetries=True, use_position_averaging=True, weighing_policy=azhip.LOG_WEIGHT) as d:
        _same_samples(d.raw_samples(), ref, 7)
        W, X, A, P, V = d.tensors()
        Wr = R.convert_samples(0, 1, ref)[0]
        assert np.array_equal(W, Wr) and d.sum_n == 2 * nm
        assert abs(d.Wtot - float(Wr.astype(np.float64).sum())) < 1e-9 * d.Wtot
    mem.close()


def test_circular_buffer_semantics():
    """CircularBuffer(size) + cur_batch_size / last_batch / new_batch! / empty! (memory.jl:34-60)"""
    import azhip
    gspec = azhip.TicTacToeSpec()
    g1 = _selfplay(1, 6, 3, 16, 1)
    g2 = _selfplay(1, 7, 3, 16, 2)
    r1 = _oracle_samples(1, g1[0], g1[1], g1[2], 1.0)
    r2 = _oracle_samples(1, g2[0], g2[1], g2[2], 1.0)
    cap = len(r1) + 5
    mem = azhip.MemoryBuffer(gspec, cap)
    mem.push_records(g1[0], g1[1], g1[2], g1[3], 1.0)
    mem.new_batch()
    assert (len(mem), mem.cur_batch_size()) == (len(r1), 0)
    mem.push_records(g2[0], g2[1], g2[2], g2[3], 1.0)
    allref = (r1 + r2)[-cap:]
    assert len(mem) == cap and mem.cur_batch_size() == min(len(r2), cap)
    with mem.dataset() as d:
        _same_samples(d.raw_samples(), allref, 9)
    with mem.dataset(last_batch=True) as d:
        _same_samples(d.raw_samples(), r2[-cap:], 9)
    mem.push_records(g1[0], g1[1], g1[2], g1[3], 1.0)                # more than one wrap in total
    with mem.dataset() as d:
        _same_samples(d.raw_samples(), (r1 + r2 + r1)[-cap:], 9)
    tiny = azhip.MemoryBuffer(gspec, 3)                               # one push larger than the buffer
    tiny.push_records(g2[0], g2[1], g2[2], g2[3], 1.0)
    with tiny.dataset() as d:
        _same_samples(d.raw_samples(), r2[-3:], 9)
    mem.empty()
    assert (len(mem), mem.cur_batch_size()) == (0, 0)
    with mem.dataset() as d:
        assert len(d) == 0
    mem.close(); tiny.close()


@pytest.mark.parametrize("game,F,policy,batch", [(0, 64, 1, 64), (1, 64, 2, 1 << 20), (2, 64, 0, 37), (0, 128, 1, 500)])
def test_learning_status_matches_oracle(game, F, policy, batch):
    """Trainer + learning_status (learning.jl:98-121,158-181) on the device vs the oracle's Float32 restatement"""
    import azhip
    gspec = getattr(azhip, SPECS[game])()
    hp = azhip.ResNetHP(num_blocks=2, num_filters=F, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=21)
    games, moves, ng, nm, _ = _selfplay(game, 24, 8, 24, 9)
    mem = azhip.MemoryBuffer(gspec, 100000)
    mem.push_records(games, moves, ng, nm, 1.0)
    use_sym = game != 2
    lp = azhip.LearningParams(samples_weighing_policy=policy, l2_regularization=1e-4, loss_computation_batch_size=batch,
                              rewards_renormalization=1.0, nonvalidity_penalty=1.0)
    with azhip.Trainer(gspec, nn, mem, lp, use_symmetries=use_sym) as tr:
        st = tr.learning_status()
        rep = tr.samples_report()
        data = tr.data.tensors()
    ref = R.learning_status(game, (2, F, 32, 32), nn.params(), data, l2=1e-4, nonvalidity_penalty=1.0,
                            rewards_renormalization=1.0, batch=batch)
    got = np.array([st.loss.L, st.loss.Lp, st.loss.Lv, st.loss.Lreg, st.loss.Linv, st.Hp, st.Hpnet])
    want = np.array([ref.L, ref.Lp, ref.Lv, ref.Lreg, ref.Linv, ref.Hp, ref.Hpnet])
    assert np.allclose(got, want, rtol=2e-6, atol=1e-7), (got, want)
    assert rep.num_samples == nm * (1 + (R.lib().azr_num_symmetries(game) if use_sym else 0)) and rep.num_boards == len(data[0])
    mem.close()


def test_memory_errors():
    import azhip
    from azhip import _lib as L
    gspec = azhip.TicTacToeSpec()
    with pytest.raises(L.AzError, match="capacity"):
        azhip.MemoryBuffer(gspec, 0)
    mem = azhip.MemoryBuffer(gspec, 10)
    with pytest.raises(L.AzError, match="policy"):
        mem.dataset(weighing_policy=7)
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    lp = azhip.LearningParams(samples_weighing_policy=0, l2_regularization=0.0, loss_computation_batch_size=8)
    with azhip.Trainer(gspec, azhip.ResNet(gspec, hp, seed=1), mem, lp) as tr:
        with pytest.raises(L.AzError, match="empty"):
            tr.learning_status()
    games, moves, ng, nm, _ = _selfplay(1, 2, 2, 8, 1)
    games[0].first_move = 10 ** 6
    with pytest.raises(L.AzError, match="outside"):
        mem.push_records(games, moves, ng, nm, 1.0)
    mem.close()


def test_iteration_example_runs_end_to_end():
    """examples/iteration.py: two full iterations (self-play -> device memory -> batch updates -> status -> arena), tiny
    sizes; deterministic"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("iteration", os.path.join(os.path.dirname(__file__), "..", "examples", "iteration.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r1 = mod.main(iters=2, games=16, workers=8, sims=12, batch=32, quiet=True)
    r2 = mod.main(iters=2, games=16, workers=8, sims=12, batch=32, quiet=True)
    for (sp1, lr1), (sp2, lr2) in zip(r1, r2):
        assert sp1.memory_size == sp2.memory_size > 16 and sp1.memory_num_distinct_boards <= sp1.memory_size
        assert np.array_equal(lr1.losses, lr2.losses) and len(lr1.losses) >= 1 and np.isfinite(lr1.losses).all()
        assert lr1.initial_status == lr2.initial_status and lr1.nn_replaced == lr2.nn_replaced
        assert np.array_equal(lr1.checkpoints[0].evaluation.rewards, lr2.checkpoints[0].evaluation.rewards)
        assert len(lr1.checkpoints[0].evaluation.rewards) == 4
    assert r1[1][0].memory_size > r1[0][0].memory_size


def test_push_samples_and_memory_report():
    """host TrainingSamples into the device buffer (wrap-around included) and memory_report (learning.jl:192-216)"""
    import azhip
    gspec = azhip.TicTacToeSpec()
    games, moves, ng, nm, _ = _selfplay(1, 12, 4, 16, 2)
    ref = _oracle_samples(1, games, moves, ng, 1.0)
    mem = azhip.MemoryBuffer(gspec, 1000)
    mem.push_records(games, moves, ng, nm, 1.0)
    host = mem.get_experience()
    m2 = azhip.MemoryBuffer(gspec, len(host) - 3)                     # smaller than what is pushed: the oldest fall out
    m2.push_samples(host[:5])
    m2.push_samples(host[5:])
    with m2.dataset() as d:
        _same_samples(d.raw_samples(), ref[3:], 9)
    assert m2.cur_batch_size() == 0
    hp = azhip.ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    nn = azhip.ResNet(gspec, hp, seed=4)
    lp = azhip.LearningParams(samples_weighing_policy=1, l2_regularization=1e-4, loss_computation_batch_size=32)
    rep = azhip.memory_report(mem, nn, lp, num_game_stages=3)
    assert rep.all_samples.num_samples == nm and rep.latest_batch.num_samples == nm and len(rep.per_game_stage) == 3
    assert sum(s.samples_stats.num_samples for s in rep.per_game_stage) == nm
    ts = [(s.min_remaining_length, s.max_remaining_length) for s in rep.per_game_stage]
    assert ts[0][0] == 1.0 and all(a[1] <= b[0] for a, b in zip(ts, ts[1:]))
    # the stage statistics equal the oracle's learning status of the same slice
    es = sorted(ref, key=lambda e: e.t)
    csize = -(-len(es) // 3)
    part = R.merge_by_state(1, es[:csize])
    want = R.learning_status(1, (1, 64, 32, 32), nn.params(), R.convert_samples(1, 1, part), l2=1e-4, batch=32)
    got = rep.per_game_stage[0].samples_stats.status
    assert np.allclose([got.loss.L, got.loss.Lp, got.loss.Lv, got.Hp, got.Hpnet], [want.L, want.Lp, want.Lv, want.Hp, want.Hpnet], rtol=2e-6, atol=1e-7)
    mem.close(); m2.close()


def test_push_trace_samples_with_unavailable_actions_reach_the_device_full_width():
    """push_trace() builds TrainingSamples whose π covers the AVAILABLE actions only (memory.jl:74-87) while
    az_sample.pi is indexed by full action index.  MemoryBuffer.push_samples scatters the compact π through the state's
    action mask: the device samples equal the ones az_memory_push derives from the same records (Tic-tac-toe: every
    state after the first move has an unavailable action)."""
    import azhip
    from azhip.memory import push_trace
    from azhip.trace import trace_from_records
    gspec = azhip.TicTacToeSpec()
    games, moves, ng, nm, _ = _selfplay(1, 6, 3, 16, 1)
    host = []
    for i in range(ng):
        tr = trace_from_records(games[i], moves, 9, lambda key: gspec.init(key).actions_mask())
        assert any(len(p) < 9 for p in tr.policies)
        push_trace(host, tr, 1.0)
    a, b = azhip.MemoryBuffer(gspec, 1000), azhip.MemoryBuffer(gspec, 1000)
    a.push_records(games, moves, ng, nm, 1.0)
    b.push_samples(host)
    with a.dataset() as da, b.dataset() as db:
        _same_samples(db.raw_samples(), [da.raw_samples()[i] for i in range(nm)], 9)
    with pytest.raises(ValueError):
        bad = host[1]
        b.push_samples([type(bad)(bad.s, bad.π[:-1], bad.z, bad.t, bad.n)])
    a.close(); b.close()


@pytest.mark.parametrize("game", [0, 2])
def test_device_only_phase_pushes_the_same_samples_as_the_host_path(game):
    """training.jl:284-299 without the host hop: az_selfplay_run with out->moves == NULL keeps the move
    records in the engine's HBM phase buffer and returns game records only; az_memory_push_engine runs push_trace! from
    there.  Same samples, same order, as pushing the host copy of the same phase -- and as the oracle's."""
    import azhip
    kw = dict(game=game, oracle=azhip.ORACLE_HASH, num_workers=5, batch_size=5, num_iters_per_turn=24, dirichlet_noise_eps=0.25,
              cpuct=1.0, reset_every=1, temperature=([0], [1.0]), seed=3, max_moves_per_game=200 if game == 2 else 0)
    gspec = getattr(azhip, SPECS[game])()
    with azhip.Engine(**kw) as e:
        games, moves, ng, nm, st = e.selfplay_run(17, first_game_id=100)
        host = azhip.MemoryBuffer(gspec, 100000)
        host.push_records(games, moves, ng, nm, 0.9)
        dev0 = azhip.MemoryBuffer(gspec, 100000)
        dev0.push_engine(e, 0.9)                                   # the phase buffer is kept whether or not the host asked for a copy
        g2, m2, ng2, nm2, st2 = e.selfplay_run(17, first_game_id=100, device_only=True)
        assert m2 is None and nm2 == 0 and ng2 == 17 and st2.moves == nm
        assert [(g2[i].game_id, g2[i].num_moves, g2[i].first_move) for i in range(17)] == [(games[i].game_id, games[i].num_moves, -1) for i in range(17)]
        dev = azhip.MemoryBuffer(gspec, 100000)
        dev.push_engine(e, 0.9)
        small = azhip.MemoryBuffer(gspec, nm - 7)                  # ring smaller than the phase: the oldest samples fall out
        small.push_engine(e, 0.9)
    ref = _oracle_samples(game, games, moves, ng, 0.9)
    nA = gspec.num_actions()
    with host.dataset() as dh, dev.dataset() as dd, dev0.dataset() as d0, small.dataset() as ds:
        _same_samples(dh.raw_samples(), ref, nA)
        _same_samples(dd.raw_samples(), ref, nA)
        _same_samples(d0.raw_samples(), ref, nA)
        _same_samples(ds.raw_samples(), ref[7:], nA)
    assert dev.cur_batch_size() == nm
    with azhip.Engine(**kw) as e2:
        with pytest.raises(azhip.AzError):
            dev.push_engine(e2, 1.0)                               # no phase has been played on this engine
    for m in (host, dev, dev0, small):
        m.close()
