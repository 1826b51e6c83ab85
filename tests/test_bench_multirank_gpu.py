"""bench.py's N > 1 path as the driver launches it (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`),
rehearsed on the ONE GPU of the test box: two ranks share cuda:0 and the library's exchange runs over the test-only transport
(AZHIP_RCCL_LIB -> tests/rccl_stub; RCCL itself refuses two ranks per device).  What it proves: the rendezvous over gloo, the
barrier + max-over-ranks timing, the summed counters, and the exchange leg after the timed region (weights broadcast, device-only
phases with global game ids, az_comm_gather_push into every rank's device memory) run end to end with W = 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env, tmp_path, limit=240):
    """runs bench.py with its progress marks on (AZ_BENCH_TRACE), output into files: a run that does not finish within `limit` seconds
    fails with the marks of every rank in the message instead of a bare TimeoutExpired (the 8-second run once sat for two minutes
    behind other work on the test box and nothing said where)"""
    out, err = tmp_path / "stdout.txt", tmp_path / "stderr.txt"
    with open(out, "w") as fo, open(err, "w") as fe:
        try:
            rc = subprocess.run(cmd, stdout=fo, stderr=fe, timeout=limit, env=dict(env, AZ_BENCH_TRACE="1"), cwd=ROOT).returncode
        except subprocess.TimeoutExpired:
            rc = None
    so, se = out.read_text(), err.read_text()
    assert rc is not None, "bench.py did not finish within %d s; progress marks:\n%s" % (limit, "\n".join(ln for ln in se.splitlines() if "bench rank" in ln)[-3000:] or se[-3000:])
    return rc, so, se


def test_two_ranks_on_one_gpu_print_one_line_with_the_gather_leg(tmp_path):
    stub = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")
    if not os.path.exists(stub):
        subprocess.check_call(["make", "-C", os.path.dirname(stub)])
    env = dict(os.environ, AZHIP_RCCL_LIB=stub, HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29000 + os.getpid() % 900
    rc, so, se = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                       "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--slots", "512"], env, tmp_path)
    lines = [ln for ln in so.splitlines() if ln.startswith("{")]
    assert rc == 0 and len(lines) == 1, so[-2000:] + se[-2000:]                                # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["sims_per_sec_per_gpu"] * 2 - d["value"]) < 1e-6 * d["value"]
    g = d["gather"]
    assert "error" not in g, g
    assert g["ranks"] == 2 and g["world"] == 2 and g["rendezvous"].endswith("gloo")
    assert g["games"] == 1024 and g["memory_length"] == g["samples"] > 1024 * 7                  # every rank holds all ranks' samples
    assert "extra" not in d and "cpu_baseline" not in d                                         # those legs belong to N = 1


def test_plain_bench_with_gpus_2_launches_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` exactly as a driver without a launcher calls it: the script re-executes itself under
    torch.distributed.run with two ranks (stub transport on the one GPU) and rank 0 prints ONE line with n_gpus = 2."""
    stub = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")
    if not os.path.exists(stub):
        subprocess.check_call(["make", "-C", os.path.dirname(stub)])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AZHIP_RCCL_LIB=stub, HSA_ENABLE_IPC_MODE_LEGACY="0")
    rc, so, se = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--slots", "512"], env, tmp_path)
    lines = [ln for ln in so.splitlines() if ln.startswith("{")]
    assert rc == 0 and len(lines) == 1, so[-2000:] + se[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launcher"].startswith("self") and len(d["sims_per_sec_by_rank"]) == 2
    assert all(v > 0 for v in d["sims_per_sec_by_rank"]) and d["value"] <= sum(d["sims_per_sec_by_rank"]) * (1 + 1e-9)
    g = d["gather"]
    assert "error" not in g and g["ranks"] == 2 and g["library"].endswith("librccl_stub.so"), g


def test_plain_bench_refuses_more_ranks_than_devices_without_the_stub():
    """no silent 1-GPU number under an n_gpus = 8 label: more ranks than devices is an error (RCCL: one rank per device)"""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "AZHIP_RCCL_LIB")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "device(s) visible" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
