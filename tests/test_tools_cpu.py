"""CPU checks of the measurement tooling of round 5 (no GPU, no oracle): the timed-region window of a rocprofv3 kernel trace
(tools/trace_window.py), the committed round-5 bench line recomputed from its own fields, and the address map of a mapped-on-demand
pool's side records (csrc/tree.h side_at, csrc/azhip.hip vm_map) restated in Python."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trace_window_takes_the_last_launches_of_the_kernel(tmp_path):
    trace = tmp_path / "1_kernel_trace.csv"
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp"])
        for i in range(300):                                            # warm-up launches of 800 ns, then 100 timed ones of 400 ns, interleaved with another kernel
            w.writerow(["KERNEL_DISPATCH", 1, 1, 7, "void k_tower16x2<ConnectFour, 64, false>(Net16Dev, GEnv const*)", i, 10_000 * i, 10_000 * i + (800 if i < 200 else 400)])
            w.writerow(["KERNEL_DISPATCH", 1, 1, 9, "void k_tree<ConnectFour>(DView, DParams, int, int, int)", i, 10_000 * i + 900, 10_000 * i + 950])
    rows = tmp_path / "rows.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_window.py"), str(trace), "k_tower", "100", str(rows)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["dispatches_in_process"] == 300 and d["window"] == "last 100 dispatches"
    assert abs(d["avg_us"] - 0.4) < 1e-9 and abs(d["avg_us_whole_process"] - (200 * 0.8 + 100 * 0.4) / 300) < 1e-9
    got = list(csv.reader(open(rows)))
    assert len(got) == 101 and all("k_tower16x2" in g[4] for g in got[1:])


def test_the_committed_round_5_line_recomputes_from_its_own_fields():
    """profiles/r5/bench_default_line.json: roofline.frac = executed FLOPs x boards / sum of launch times / peak; nothing above 1; the
    flat also_* scalars (what the driver's record keeps) equal the nested figures they stand for"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r5", "bench_default_line.json")))
    r = d["roofline"]
    assert d["config"]["slot_groups"] == 1 and r["exclusive_ms"] <= r["wall_ms"] and abs(r["exclusive_ms"] - r["launch_ms_sum"]) < 1e-9
    boards = r["avg_boards_per_launch"] * r["launches"]
    achieved = r["flop_per_board"] * boards / (r["exclusive_ms"] * 1e-3) / 1e12
    assert abs(achieved - r["achieved"]) < 1e-6 * achieved and abs(r["frac"] - achieved / 157.3) < 1e-9
    assert abs(r["frac_over_wall"] - r["frac"] * r["exclusive_ms"] / r["wall_ms"]) < 1e-9
    # executed FLOPs per board: dense stem / head convolutions + the tower's 3x3 convolutions x the executed product fraction
    conv, other = 2 * 5 * 9 * 64, 9 * 3 + 32 + 32
    assert abs(r["flop_per_board"] - 2 * 42 * 64 * (conv * r["executed_product_frac"] + other)) < 1.0
    assert r["dense_flop_per_board"] == 2 * 42 * 64 * (conv + other) == 31454976          # SURVEY 8(d)'s dense count
    fr = [r["frac"], d["roofline_kernel_alone"]["frac"], d["roofline_tree"]["frac"]] + [v["roofline"]["frac"] for v in d["extra"].values() if isinstance(v, dict) and "roofline" in v]
    assert all(0.0 < x <= 1.0 for x in fr), fr
    # the evaluation cache's bookkeeping: boards the network evaluated = unique_leaf_frac x oracle calls
    assert abs(boards - d["unique_leaf_frac"] * d["leaf_evals_per_sim"] * d["value"] * d["ms_per_step"] * 1e-3 * d["steps"]) < 1e-6 * boards
    assert r["also_phase_sims_per_sec"] == d["extra"]["whole_phase"]["value"] == d["phase"]["sims_per_sec"] == d["summary"]["phase"]["sims_per_sec"]
    assert r["also_c2_5x128_sims_per_sec"] == d["extra"]["c2_5x128"]["value"] and r["also_kernel_alone_frac"] == d["roofline_kernel_alone"]["frac"]
    assert r["also_learning_loss_before"] == d["extra"]["iteration"]["learning_status"]["before"]["L"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_side_records_of_a_mapped_pool_have_one_place_each_inside_mapped_granules():
    """DView::keys of a mapped-on-demand pool (tree.h side_at, azhip.hip vm_map), restated: side record (slot, idx) lives at word
    (idx >> sh) * key_row + slot * key_stride + (idx & mask) * 4 of a virtual range of rows x G pieces; the host maps the 2 MB granule a
    piece lies in when it maps node chunk (row, slot).  Every record of a backed chunk must lie in a mapped granule, no two records may
    share a place, and a group view's offset (slot o .. o + Gh) must address the same words as the whole-engine view."""
    CH = 2 << 20
    for G, node_bytes in ((6, 128), (8192, 128), (7, 128)):
        chunk_nodes = CH // node_bytes
        sh, mask = chunk_nodes.bit_length() - 1, chunk_nodes - 1
        piece = chunk_nodes * 32
        assert CH % piece == 0
        key_row, key_stride = G * piece // 8, piece // 8
        rows = 5
        granules = set()
        backed = [(r, s) for r in range(rows) for s in range(G) if (r * 31 + s * 17) % 3 != 0 or r == 0]   # an arbitrary set of backed chunks
        for r, s in backed:
            granules.add((r * G + s) * piece // CH)                     # vm_map's granule of the chunk's piece
        seen = set()
        for r, s in backed[:: max(1, len(backed) // 200)]:
            for off in (0, 1, chunk_nodes // 2, chunk_nodes - 1):
                idx = r * chunk_nodes + off
                word = (idx >> sh) * key_row + s * key_stride + (idx & mask) * 4
                byte = word * 8
                assert byte // CH in granules and (byte + 31) // CH in granules
                assert byte not in seen
                seen.add(byte)
                o = (s // 2) * 2                                        # a slot-group view starting at slot o
                assert o * key_stride + (idx >> sh) * key_row + (s - o) * key_stride + (idx & mask) * 4 == word
        assert rows * G * piece <= ((rows * G * piece + CH - 1) // CH) * CH
