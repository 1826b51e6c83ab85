#!/usr/bin/env python
"""Dry run of the reference-golden loader: what a maintainer runs right after `tools/gen_golden.jl`.

    python tools/check_golden.py [--dir tests/golden] [--gpu]     validate ref_mcts.json / ref_play.json / ref_net.json (+ blob)
    python tools/check_golden.py --selftest                      write a miniature set of files through the independent
                                                                   pure-Python restatement (oracle/pyref.py, fp64 torch) into
                                                                   a temporary directory and validate THAT: proves the loader
                                                                   and every schema branch without Julia

Every check is the one `pytest tests/test_reference_golden.py` runs (imported from there); this front end only reports per
file instead of skipping, and exits non-zero when a file is missing or a value differs -- "parity unpinned" until it exits 0
on files Julia produced."""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "alphazero.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--gpu", action="store_true", help="also run the HIP engine against the files (needs an MI355X)")
    a = ap.parse_args()
    import pytest
    import test_reference_golden as T
    d = a.dir
    if a.selftest:
        d = tempfile.mkdtemp(prefix="azgolden_")
        T.write_mock_golden(d)
        print("selftest: miniature golden files written by oracle/pyref.py + fp64 torch into", d)
    checks = [("ref_mcts.json", "CPU oracle vs reference MCTS.explore! / policy", T.check_mcts_cpu),
              ("ref_play.json", "CPU oracle vs reference play_game traces", T.check_play_cpu),
              ("ref_net.json", "CPU oracle vs reference Flux ResNet (1e-5)", T.check_net_cpu)]
    if a.gpu:
        checks += [("ref_mcts.json", "HIP engine vs reference MCTS", T.check_mcts_gpu), ("ref_play.json", "HIP engine vs reference play_game", T.check_play_gpu),
                   ("ref_net.json", "HIP engine vs reference network", T.check_net_gpu)]
    bad = 0
    for name, what, fn in checks:
        if not os.path.exists(os.path.join(d, name)):
            print("MISSING  %-14s %s  (parity unpinned)" % (name, what))
            bad += 1
            continue
        try:
            fn(d)
            print("ok       %-14s %s" % (name, what))
        except pytest.skip.Exception as ex:
            print("SKIPPED  %-14s %s: %s" % (name, what, ex))
            bad += 1
        except AssertionError as ex:
            print("DIFFERS  %-14s %s: %s" % (name, what, str(ex)[:300]))
            bad += 1
    print("parity %s" % ("PINNED to the files in " + d if bad == 0 and not a.selftest else "unpinned" if bad else "loader self-test passed (files are NOT the reference's)"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
