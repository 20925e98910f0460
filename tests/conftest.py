"""Shared fixtures.  `-m "not gpu"` runs on the CPU-only builder container; `-m gpu` on a MI355X.

Only this directory (plus __graft_entry__.smoke and bench.py's cpu_baseline leg) may import
the CPU oracle under oracle/.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import t360_oracle as O
    O.build(ref=True)
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("maps", "lowpass", "frames"):
        p = os.path.join(GOLDEN_DIR, name + ".json")
        if os.path.exists(p):
            with open(p) as f:
                out[name] = json.load(f)
    return out


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a GPU box; building happens in __graft_entry__.build()."""
    from transform360_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def noise_plane(h, w, seed, pad=0):
    """uint8 noise plane whose row stride is w + pad (to exercise linesize handling)."""
    rng = np.random.default_rng(seed)
    buf = rng.integers(0, 256, (h, w + pad), dtype=np.uint8)
    return buf[:, :w]
