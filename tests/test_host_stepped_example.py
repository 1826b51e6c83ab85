"""examples/host_stepped_go9.cpp -- BASELINE configs[4] end to end in its host-stepped form (host rules + host trees in C++,
the network through az_net_forward).  Built with plain g++ against include/azhip.h.  Without a GPU: `--dry` (uniform oracle
on the host) exercises the rules, the tree and the lock-step driver, and the real mode must fail loudly (exit 2), never fall
back.  On a GPU: a short run with the bf16 10x128 tower reports throughput and the host / network split."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "host_stepped_go9")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    return EXE


def _run(*args):
    r = subprocess.run([_build(), *args], capture_output=True, text=True, timeout=120, cwd="/tmp")
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
    return r, (json.loads(line) if line else None)


def test_dry_mode_plays_go_shaped_games_on_the_host():
    r, d = _run("--dry", "--workers", "32", "--sims", "24", "--seconds", "1.5", "--threads", "2")
    assert r.returncode == 0, r.stderr
    assert d["value"] > 1000 and d["games_finished"] >= 1 and d["moves"] > 100      # whole games reach two passes / the move limit
    assert 0.9 < d["boards_per_launch"] <= 16 and d["avg_exploration_depth"] > 0.5      # two halves of 16 workers take turns
    assert 0.2 < d["host_tree_share"] <= 1.0 and 0.0 <= d["network_wait_share"] < 0.8
    assert "NOT OpenSpiel" in d["rules"]


def test_real_mode_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r, d = _run("--seconds", "0.5")
    assert r.returncode == 2 and d is None and "az_engine_create" in r.stderr


@pytest.mark.gpu
def test_host_tree_over_the_network_seam_reports_a_throughput():
    r, d = _run("--workers", "256", "--sims", "64", "--seconds", "2")
    assert r.returncode == 0, r.stderr
    assert d["value"] > 1000 and d["kernel"].startswith("k_tower16b<Go9Planes,128") and d["network_busy_share"] > 0.01
    assert 50 < d["boards_per_launch"] <= 128
