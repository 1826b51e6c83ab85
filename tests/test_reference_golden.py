"""Golden vectors produced by the REFERENCE (tools/gen_golden.jl run with Julia + AlphaZero.jl) -> tests/golden/ref_*.json.

The build container has no Julia, so the files cannot be generated here: when they are absent these tests SKIP with
"parity unpinned" (the oracle is then pinned only to restatements).  When a box with Julia has produced
them, the same tests pin
  * the CPU oracle (oracle/azref.c) -- `not gpu`,
  * the HIP engine through the C ABI -- `gpu`
to the reference's own MCTS.explore! / policy / play_game / Categorical sampler / Flux ResNet outputs:
visit counts, W, priors, policies and sampled actions exactly (the injected oracles are exact functions of the state),
network outputs within the 1e-5 of BASELINE.json.
The loader itself is exercised on every run: the same schema is written by an independent pure-Python restatement
(oracle/pyref.py, and the fp64 torch network of tests/test_net.py) into a temporary directory and consumed by the same
checks."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import azref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ORACLES = {"uniform": 0, "hash": 1}
UNPINNED = ("parity unpinned: tests/golden/%s is absent -- run `julia --project=<AlphaZero.jl> tools/gen_golden.jl` "
            "on a box with Julia to pin the oracle and the HIP engine to the reference's own outputs")


def load(dirname, name):
    p = os.path.join(dirname, name)
    if not os.path.exists(p):
        pytest.skip(UNPINNED % name)
    return json.load(open(p))


def key_of(k):
    return int(k[0]), int(k[1])


def full_eta(game, key, eta):
    """eta by rank among the available actions -> by full action index (az_mcts_explore's convention)"""
    mask = R.Game(game, R.unpack_key(game, key)).actions_mask()
    out = np.zeros(9)
    out[np.nonzero(mask)[0]] = eta
    return out


def image_of(game, g, k):
    """the k-th entry of GI.symmetries for the oracle game g, as a new oracle game"""
    st = g.state()
    cells, cur = R.symmetry(game, list(st.cells), st.curplayer, k)
    img = R.State()
    for j, c in enumerate(cells):
        img.cells[j] = c
    img.curplayer = cur
    return R.Game(game, img)


# ------------------------------------------------------------------------------------------------ the checks
def check_mcts_cpu(dirname):
    for c in load(dirname, "ref_mcts.json")["cases"]:
        g = R.Game(c["game"])
        for a in c["prefix"]:
            g.play(a)
        assert g.key() == key_of(c["root_key"])                    # state encoding (julia/AlphaZeroHIP.jl encode_state)
        m = R.Mcts(c["game"], oracle=ORACLES[c["oracle"]], gamma=c["gamma"], cpuct=c["cpuct"], noise_eps=c["eps"],
                   prior_temperature=c["prior_temperature"])
        m.explore(g, c["nsims"], eta=np.array(c["eta"]))
        N, W, P, V = m.root_stats(g)
        assert list(N) == c["N"], (c["game"], c["oracle"], list(N), c["N"])
        assert list(W) == c["W"] and [float(x) for x in P] == c["P"] and float(V) == c["Vest"]
        acts, pi = m.policy(g)
        assert acts == c["actions"] and list(pi) == c["pi"]
        assert (m.total_simulations, m.total_nodes_traversed, m.num_nodes) == (c["total_simulations"], c["total_nodes_traversed"], c["num_nodes"])


def check_play_cpu(dirname):
    from azhip.params import ConstSchedule, PLSchedule
    from azhip.play import apply_temperature
    L = R.lib()
    for c in load(dirname, "ref_play.json")["cases"]:
        game = c["game"]
        m = R.Mcts(game, oracle=ORACLES[c["oracle"]], cpuct=c["cpuct"], noise_eps=0.25)
        sched = ConstSchedule(c["temp_ys"][0]) if len(c["temp_xs"]) == 1 else PLSchedule(c["temp_xs"], c["temp_ys"])
        g = R.Game(game)
        flips = c.get("flips") or [0] * len(c["actions"])           # play_game's per-turn symmetry: 0 = none, else 1 + its index in GI.symmetries
        for k, action in enumerate(c["actions"]):
            assert g.key() == key_of(c["states"][k])                # trace.states[k]: the state BEFORE the turn's flip (play.jl:305-313)
            if flips[k]:
                g = image_of(game, g, flips[k] - 1)                 # the player thinks and plays on the image; policy / N are by rank
            m.explore(g, c["nsims"], eta=np.array(c["etas"][k]))    # the tree persists between the moves of a game
            acts, pi = m.policy(g)
            assert list(pi) == c["policies"][k], (game, k)
            assert list(m.root_stats(g)[0]) == c["N"][k]
            # play.jl:309-311: temperature of move k, then the categorical draw with the recorded uniform
            pis = np.ascontiguousarray(apply_temperature(pi, sched[k]), dtype=np.float64)
            idx = L.azr_rand_categorical(pis.ctypes.data_as(C.c_void_p), len(pis), C.c_float(c["us"][k]))
            assert acts[idx] == action, (game, k, acts[idx], action)
            assert L.azr_move_uniform(int(c["seed"]), c["game_id"], k) == c["us"][k]      # u IS the contract's draw for (seed, game, move)
            g.play(action)
            assert g.white_reward() == c["rewards"][k]
        assert g.terminated() and g.key() == key_of(c["states"][-1])
        assert (m.total_simulations, m.total_nodes_traversed, m.num_nodes) == (c["total_simulations"], c["total_nodes_traversed"], c["num_nodes"])


def net_batch(c, dirname):
    game = c["game"]
    blob = np.fromfile(os.path.join(dirname, c["blob_file"]), dtype="<f4")
    envs = [R.Game(game, R.unpack_key(game, key_of(k))) for k in c["states"]]
    w, h, ch = R.DIMS[game]
    X = np.stack([g.vectorize().reshape(ch, h, w) for g in envs])
    A = np.stack([g.actions_mask().astype(np.float32) for g in envs])
    return blob, envs, X, A


def check_net_cpu(dirname):
    for c in load(dirname, "ref_net.json")["cases"]:
        blob, envs, X, A = net_batch(c, dirname)
        hp = (c["num_blocks"], c["num_filters"], c["num_policy_head_filters"], c["num_value_head_filters"])
        assert blob.size == R.net_num_params(c["game"], *hp)        # the Flux flattening order has the expected size
        P, V, _ = R.net_forward_normalized(c["game"], hp, blob, X, A)
        for i, g in enumerate(envs):
            assert np.abs(P[i][g.actions_mask()] - np.array(c["P"][i])).max() < 1e-5 and abs(V[i] - c["V"][i]) < 1e-5, (c["game"], i)


def check_mcts_gpu(dirname):
    import azhip
    for c in load(dirname, "ref_mcts.json")["cases"]:
        key = key_of(c["root_key"])
        with azhip.Engine(game=c["game"], oracle=ORACLES[c["oracle"]], num_workers=2, batch_size=2, num_iters_per_turn=c["nsims"],
                          gamma=c["gamma"], cpuct=c["cpuct"], dirichlet_noise_eps=c["eps"], prior_temperature=c["prior_temperature"]) as e:
            e.mcts_explore([key], c["nsims"], eta=full_eta(c["game"], key, c["eta"])[None, :])
            N, W, P, V, mask = e.mcts_node_stats(0, key)
            av = [a for a in range(e.num_actions) if (mask >> a) & 1]
            assert av == c["actions"] and [int(N[a]) for a in av] == c["N"]
            assert [float(W[a]) for a in av] == c["W"] and [float(P[a]) for a in av] == c["P"] and float(V) == c["Vest"]
            assert e.mcts_counters(0) == (c["total_simulations"], c["total_nodes_traversed"], c["num_nodes"])


def check_play_gpu(dirname):
    import azhip
    for c in load(dirname, "ref_play.json")["cases"]:
        game = c["game"]
        with azhip.Engine(game=game, oracle=ORACLES[c["oracle"]], num_workers=1, batch_size=1, num_iters_per_turn=c["nsims"],
                          cpuct=c["cpuct"], dirichlet_noise_eps=0.25, max_nodes_per_slot=c["nsims"] * (len(c["actions"]) + 1)) as e:
            flips = c.get("flips") or [0] * len(c["actions"])
            for k in range(len(c["actions"])):
                key = key_of(c["states"][k])
                if flips[k]:
                    key = image_of(game, R.Game(game, R.unpack_key(game, key)), flips[k] - 1).key()
                e.mcts_explore([key], c["nsims"], eta=full_eta(game, key, c["etas"][k])[None, :])   # slot 0 keeps its tree
                N, W, P, V, mask = e.mcts_node_stats(0, key)
                assert [int(N[a]) for a in range(e.num_actions) if (mask >> a) & 1] == c["N"][k], (game, k)
            assert e.mcts_counters(0) == (c["total_simulations"], c["total_nodes_traversed"], c["num_nodes"])


def check_net_gpu(dirname):
    import azhip
    for c in load(dirname, "ref_net.json")["cases"]:
        blob, envs, X, A = net_batch(c, dirname)
        with azhip.Engine(game=c["game"], oracle=azhip.ORACLE_RESNET, num_workers=4, batch_size=4, num_iters_per_turn=4,
                          num_blocks=c["num_blocks"], num_filters=c["num_filters"], num_policy_head_filters=c["num_policy_head_filters"],
                          num_value_head_filters=c["num_value_head_filters"]) as e:
            e.net_set_params(blob)
            P, V = e.net_evaluate_keys(np.array([key_of(k) for k in c["states"]], dtype=np.uint64))
        for i, g in enumerate(envs):
            assert np.abs(P[i][g.actions_mask()] - np.array(c["P"][i])).max() < 1e-5 and abs(V[i] - c["V"][i]) < 1e-5, (c["game"], i)


# ------------------------------------------------------------------------------------------------ against the reference
def test_oracle_vs_reference_mcts():
    check_mcts_cpu(GOLDEN)


def test_oracle_vs_reference_play_game():
    check_play_cpu(GOLDEN)


def test_oracle_vs_reference_network():
    check_net_cpu(GOLDEN)


@pytest.mark.gpu
def test_hip_vs_reference_mcts():
    check_mcts_gpu(GOLDEN)


@pytest.mark.gpu
def test_hip_vs_reference_play_game():
    check_play_gpu(GOLDEN)


@pytest.mark.gpu
def test_hip_vs_reference_network():
    check_net_gpu(GOLDEN)


# ------------------------------------------------------------------------------------------------ loader self-test
def write_mock_golden(dirname):
    """the schema of tools/gen_golden.jl, filled by the independent Python restatements (NOT the reference)"""
    import pyref as Y
    from azhip.params import ConstSchedule, PLSchedule
    from azhip.play import apply_temperature, rand_categorical
    rng = np.random.default_rng(3)
    mcts_cases, play_cases = [], []
    for game, oname, nsims, cpuct, gamma, eps, nprefix in ((0, "hash", 120, 2.0, 1.0, 0.25, 3), (1, "hash", 64, 1.0, 0.9, 0.5, 1),
                                                           (2, "hash", 150, 2.0, 0.97, 0.25, 4), (0, "uniform", 80, 2.0, 1.0, 0.0, 0)):
        G = Y.GAMES[game]
        g, prefix = G.init(), []
        for _ in range(nprefix):
            a = int(rng.choice([i for i, ok in enumerate(G.mask(g)) if ok]))
            g2 = G.play(g, a)
            if Y.finished(G, g2):
                break
            g = g2
            prefix.append(a)
        eta = list(rng.dirichlet(np.ones(sum(G.mask(g)))))
        y = Y.Mcts(G, Y.hash_oracle if oname == "hash" else Y.uniform_oracle, gamma=gamma, cpuct=cpuct, eps=eps)
        y.explore(g, nsims, eta)
        N, W, P, V = y.root_stats(g)
        pi = [n / sum(N) for n in N]
        s = 0.0
        for x in pi:
            s += x
        mcts_cases.append(dict(game=game, oracle=oname, nsims=nsims, cpuct=cpuct, gamma=gamma, eps=eps, prior_temperature=1.0,
                               prefix=prefix, root_key=[str(k) for k in G.key(g)], eta=eta,
                               actions=[i for i, ok in enumerate(G.mask(g)) if ok], N=N, W=W, P=[float(p) for p in P], Vest=float(V),
                               pi=[x / s for x in pi], total_simulations=y.total_simulations,
                               total_nodes_traversed=y.total_nodes_traversed, num_nodes=len(y.tree)))
    for game, nsims, cpuct, xs, ys, seed, gid, flip_p in ((1, 40, 1.0, [0], [1.0], 1, 3, 0.0), (0, 30, 2.0, [0, 4, 8], [1.0, 0.5, 0.0], 7, 12345, 0.0),
                                                          (0, 30, 2.0, [0, 10], [1.0, 0.5], 3, 77, 0.5), (1, 40, 1.0, [0], [1.0], 2, 5, 0.6)):
        G = Y.GAMES[game]
        y = Y.Mcts(G, Y.hash_oracle, cpuct=cpuct, eps=0.25)
        sched = ConstSchedule(ys[0]) if len(xs) == 1 else PLSchedule(xs, ys)
        g = G.init()
        case = dict(game=game, oracle="hash", nsims=nsims, cpuct=cpuct, temp_xs=xs, temp_ys=ys, seed=str(seed), game_id=gid, etas=[], us=[],
                    states=[[str(k) for k in G.key(g)]], policies=[], rewards=[], actions=[], N=[], flips=[])
        k = 0
        while not Y.finished(G, g):
            flip = 0
            if flip_p and rng.random() < flip_p:                    # play.jl:305-307 with the image chosen here (data, like eta)
                syms = G.symmetries(g)
                flip = 1 + int(rng.integers(len(syms)))
                g = syms[flip - 1]
            case["flips"].append(flip)
            acts = [i for i, ok in enumerate(G.mask(g)) if ok]
            eta = list(rng.dirichlet(np.ones(len(acts))))
            u = R.lib().azr_move_uniform(seed, gid, k)
            y.explore(g, nsims, eta)
            N = y.root_stats(g)[0]
            pi = np.array([n / sum(N) for n in N])
            s = 0.0
            for x in pi:
                s += float(x)
            pi = pi / s
            a = acts[rand_categorical(apply_temperature(pi, sched[k]), np.float32(u))]
            g = G.play(g, a)
            case["etas"].append(eta); case["us"].append(float(u)); case["N"].append(N); case["policies"].append(list(pi))
            case["actions"].append(a); case["rewards"].append(float(G.reward(g))); case["states"].append([str(x) for x in G.key(g)])
            k += 1
        case.update(total_simulations=y.total_simulations, total_nodes_traversed=y.total_nodes_traversed, num_nodes=len(y.tree))
        play_cases.append(case)
    json.dump(dict(generator="mock: oracle/pyref.py", cases=mcts_cases), open(os.path.join(dirname, "ref_mcts.json"), "w"))
    json.dump(dict(generator="mock: oracle/pyref.py", cases=play_cases), open(os.path.join(dirname, "ref_play.json"), "w"))
    # network: the fp64 torch restatement of the Flux model
    from azhip.network import ResNetHP, random_params
    from test_net import batch_of, random_positions, torch_forward_normalized
    net_cases = []
    for game in (0, 2):
        hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
        blob = random_params(game, hp, seed=21)
        envs = random_positions(game, 6, 8)
        X, A = batch_of(game, envs)
        P, V, _ = torch_forward_normalized(game, hp, blob, X, A)
        blob.astype("<f4").tofile(os.path.join(dirname, "ref_net_blob_%d.f32" % game))
        net_cases.append(dict(game=game, num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32,
                              blob_file="ref_net_blob_%d.f32" % game, states=[[str(k) for k in g.key()] for g in envs],
                              P=[[float(x) for x in P[i][g.actions_mask()]] for i, g in enumerate(envs)], V=[float(v) for v in V]))
    json.dump(dict(generator="mock: torch fp64", cases=net_cases), open(os.path.join(dirname, "ref_net.json"), "w"))


@pytest.fixture(scope="module")
def mock_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("mock_golden"))
    write_mock_golden(d)
    return d


def test_loader_selftest_cpu(mock_dir):
    check_mcts_cpu(mock_dir)
    check_play_cpu(mock_dir)
    check_net_cpu(mock_dir)


@pytest.mark.gpu
def test_loader_selftest_gpu(mock_dir):
    check_mcts_gpu(mock_dir)
    check_play_gpu(mock_dir)
    check_net_gpu(mock_dir)


def test_unpinned_is_reported_not_hidden():
    """without the reference's files the reference tests must SKIP with the words `parity unpinned`, never pass silently"""
    if all(os.path.exists(os.path.join(GOLDEN, f)) for f in ("ref_mcts.json", "ref_play.json", "ref_net.json")):
        pytest.skip("reference golden files present: parity is pinned")
    with pytest.raises(pytest.skip.Exception, match="parity unpinned"):
        load(GOLDEN, "ref_mcts.json")


def test_check_golden_front_end(tmp_path):
    """tools/check_golden.py: exit 0 on a directory the loader accepts (here: the pyref-written miniature), 1 while files are missing"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(GOLDEN), "..", "tools", "check_golden.py")
    r = subprocess.run([sys.executable, tool, "--selftest"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count("ok ") == 3 and "NOT the reference's" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, tool, "--dir", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and r.stdout.count("MISSING") == 3 and "parity unpinned" in r.stdout
