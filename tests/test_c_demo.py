"""The C ABI from plain C: examples/c_abi_demo.c compiles against include/azhip.h with gcc, links libazhip.so, and
(on a GPU) plays a self-play phase whose traces equal the oracle's; without a GPU it reports AZ_ERR_HIP and exits 2."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alphazero.jl_amd", "csrc")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + CSRC, "-lazhip",
                           "-Wl,-rpath," + CSRC, "-o", exe])
    return exe


def test_c_demo_builds_and_fails_gracefully_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "az_engine_create" in r.stderr


@pytest.mark.gpu
def test_c_demo_traces_equal_the_oracle(tmp_path):
    import azref as R
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    games, moves, nm = R.simulate(R.TTT, R.ORACLE_UNIFORM, 6, 4, 64, cpuct=1.0, noise_eps=0.25, reset_every=1, seed=7)
    lines = [l for l in r.stdout.splitlines() if l.startswith("game ") and ":" in l and "targets" not in l]
    assert len(lines) == 6
    for g, line in enumerate(lines):
        acts = [int(x) for x in line.split(":")[1].split("white")[0].split()]
        assert acts == [moves[games[g].first_move + k].action + 1 for k in range(games[g].num_moves)]
