#!/usr/bin/env python
"""Regenerates the fixtures of tests/golden/ (run in the BUILD container, where /root/reference exists).

  c4_scores.txt      move strings AND recorded solver scores of Test_L3_R1 (1000) and Test_L2_R1 (first 300)
  c4_positions.txt   the move strings of games/connect-four/benchmark/Test_L*_R* (first column; 6 x 1000
                     positions the reference's own Pons benchmark replays, scripts/pons_benchmark.jl:49-98)
  appendix_d.json    hand-transcribed from SURVEY.md Appendix D, src/schedule.jl:82-87 and Random123's KAT
                     (not generated: it is the independent pin of the oracle)
  net_golden.npz     oracle outputs (fp32 contract) of a fixed tiny ResNet on fixed positions: a regression
                     pin so that an accidental change of the summation-order contract is caught on CPU
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "alphazero.jl_amd"), os.path.join(ROOT, "oracle")]


def positions():
    files = sorted(glob.glob("/root/reference/games/connect-four/benchmark/Test_L*_R*"))
    assert len(files) == 6
    with open(os.path.join(HERE, "c4_positions.txt"), "w") as f:
        for p in files:
            for line in open(p):
                f.write(line.split()[0] + "\n")
    # positions WITH the solver scores the reference ships beside them (second column): the 1000 end-game positions of
    # Test_L3_R1 and the first 300 middle-game positions of Test_L2_R1 -- what tests/test_oracle_golden.py solves exactly
    with open(os.path.join(HERE, "c4_scores.txt"), "w") as f:
        for name, count in (("Test_L3_R1", 1000), ("Test_L2_R1", 300)):
            for i, line in enumerate(open("/root/reference/games/connect-four/benchmark/" + name)):
                if i < count:
                    mv, sc = line.split()
                    f.write("%s %s %d\n" % (name, mv, int(sc)))


def net_golden():
    import azref as R
    from azhip.network import ResNetHP, random_params
    hp = ResNetHP(num_blocks=1, num_filters=64, num_policy_head_filters=32, num_value_head_filters=32)
    blob = random_params(R.C4, hp, seed=7)
    rng = np.random.default_rng(0)
    envs = []
    while len(envs) < 4:
        g = R.Game(R.C4)
        for _ in range(int(rng.integers(0, 25))):
            if g.terminated():
                break
            g.play(int(rng.choice(g.available_actions())))
        if not g.terminated():
            envs.append(g)
    X = np.stack([g.vectorize().reshape(3, 6, 7) for g in envs])
    A = np.stack([g.actions_mask().astype(np.float32) for g in envs])
    P, V, Pinv = R.net_forward_normalized(R.C4, (1, 64, 32, 32), blob, X, A)
    np.savez(os.path.join(HERE, "net_golden.npz"), keys=np.array([g.key() for g in envs], dtype=np.uint64), X=X, A=A, P=P, V=V, Pinv=Pinv)


if __name__ == "__main__":
    if os.path.isdir("/root/reference"):
        positions()
    net_golden()
    print("golden fixtures written")
