"""bench.py's roofline arithmetic on the CPU: the FLOP count per board is SURVEY.md §8(d)'s figure, the fractions follow from
it, and the line committed under profiles/ can be recomputed from its own fields (what the judge does)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _hp(blocks, filters):
    from azhip.network import ResNetHP
    return ResNetHP(num_blocks=blocks, num_filters=filters, num_policy_head_filters=32, num_value_head_filters=32)


def test_dense_flops_per_board_are_the_surveys():
    import bench
    tw = {"exec_units": 1000, "units": 1000, "ms": 1.0, "launches": 1}
    r = bench.tower_roofline(0, _hp(5, 64), False, "k", tw, 4096, 1.0)
    # 31 645 952 FLOP per evaluated leaf at 5x64 heads 32/32 counts the dense heads too (2 x (32x42x7 + 32x42x64 ...));
    # the tower kernel's share = stem + 10 3x3 convolutions + the two 1x1 head convolutions
    assert r["dense_flop_per_board"] == 2 * 42 * 64 * (2 * 5 * 9 * 64 + 9 * 3 + 32 + 32) == r["flop_per_board"]
    assert 31_645_952 - r["dense_flop_per_board"] == 2 * (42 * 32 * 7 + 42 * 32 * 64 + 64 * 1)      # policy dense + value dense 1, dense 2
    assert r["frac"] == r["dense_frac"] == pytest.approx(4096 * r["flop_per_board"] / 1e-3 / 1e12 / bench.PEAK_FP32_MFMA_TFLOPS)
    r128 = bench.tower_roofline(0, _hp(5, 128), False, "k", tw, 1, 1.0)
    assert 125_204_608 - r128["dense_flop_per_board"] == 2 * (42 * 32 * 7 + 42 * 32 * 128 + 128 * 1)


def test_executed_fraction_only_scales_the_3x3_convolutions_and_time_is_clipped_to_the_wall():
    import bench
    tw = {"exec_units": 850, "units": 990, "ms": 3.0, "launches": 2}
    r = bench.tower_roofline(0, _hp(5, 64), False, "k", tw, 8192, 2e-3)        # two overlapping launches of 1.5 ms inside 2 ms of wall
    conv, other = 2 * 5 * 9 * 64, 9 * 3 + 64
    assert r["flop_per_board"] == pytest.approx(2 * 42 * 64 * (conv * 850 / 990 + other))
    assert r["exclusive_ms"] == 2.0 and r["avg_launch_ms"] == 1.5 and r["avg_boards_per_launch"] == 4096
    assert r["frac"] < r["dense_frac"] and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["dense_frac"] / r["frac"] == pytest.approx(r["dense_flop_per_board"] / r["flop_per_board"])
    b = bench.tower_roofline(0, _hp(10, 128), True, "k", tw, 8192, 2e-3)
    assert b["peak"] == bench.PEAK_BF16_MFMA_TFLOPS and b["bound"] == "mfma"


def test_committed_line_recomputes():
    p = os.path.join(ROOT, "profiles", "r4", "bench_default_line.json")
    d = json.load(open(p))
    r = d["roofline"]
    assert r["frac"] <= 1.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    evals = r["avg_boards_per_launch"] * r["launches"]
    assert r["achieved"] == pytest.approx(evals * r["flop_per_board"] / (r["exclusive_ms"] * 1e-3) / 1e12, rel=1e-9)
    assert r["exclusive_ms"] <= r["wall_ms"] * (1 + 1e-12) and r["exclusive_ms"] <= r["launch_ms_sum"] * (1 + 1e-12)
    assert d["value"] == pytest.approx(d["config"]["slots_per_gpu"] / (d["ms_per_step"] * 1e-3), rel=0.02)   # a step is one wave: one simulation per slot
    assert d["n_gpus"] == 1 and d["unit"] == "sims/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    for name, blk in d["extra"].items():                            # no block's fraction of a machine peak exceeds 1
        rf = blk.get("roofline") if isinstance(blk, dict) else None
        if rf:
            assert rf["frac"] <= 1.0, name
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["unit"] == "sims/s"
