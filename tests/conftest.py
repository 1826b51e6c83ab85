import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alphazero.jl_amd"))   # package `azhip`
sys.path.insert(0, os.path.join(ROOT, "oracle"))             # the CPU oracle (tests only)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a fresh checkout has no binaries (they are git-ignored): build the HIP library once (hipcc cross-compiles gfx950
    # without a GPU, ~2 min); the oracle builds itself on first use (oracle/azref.py)
    lib = os.path.join(ROOT, "alphazero.jl_amd", "csrc", "libazhip.so")
    if not os.path.exists(lib) and "AZHIP_LIB" not in os.environ:
        import subprocess
        subprocess.check_call(["make", "-j8", "-C", os.path.dirname(lib), "libazhip.so"])


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True, scope="module")
def _drop_cached_engines():
    """azhip keeps engines alive by configuration between phases (azhip/engine.py); tests of one module may share them,
    the next module starts with an empty cache (node pools of up to 10 GB each must not pile up over the suite)."""
    yield
    eng = sys.modules.get("azhip.engine")
    if eng is not None:
        eng.clear_engine_cache()
