import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"
GOLDEN_CASES = ["syn_1138", "gen_real", "skew", "pattern_sym", "integer_wide", "tall_empty_rows", "dense_row", "powerlaw"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu on the GPU box")


def _gpu_visible() -> bool:
    return os.path.exists("/dev/kfd") and os.path.isdir("/dev/dri")


def pytest_collection_modifyitems(config, items):
    # On a box without any AMD GPU node the gpu tests cannot run at all; on a GPU box they must
    # run against the HIP library and fail loudly if it is missing (no fallback anywhere).
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no AMD GPU device node on this machine")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        d = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
        return {k: d[k] for k in d.files}
    return load


def ref_vectors(rows, cols):
    """x_j=(j+1)/(j+2), y_i=-2(i+1)/(i+2) -- cpu/src/main.cpp:173-178."""
    j = np.arange(cols, dtype=np.float32)
    i = np.arange(rows, dtype=np.float32)
    x = ((j + 1) / (j + 2)).astype(np.float32)
    y = (np.float32(-2.0) * (i + 1) / (i + 2)).astype(np.float32)
    return x, y


ALPHA, BETA = 0.85, -2.06          # cpu/src/main.cpp:147-148
ALPHA_HOST, BETA_HOST = 0.55, -2.05  # common/src/spmv-host.cpp:43-44
TOL = 1e-5                          # BASELINE.json north_star: "within 1e-5 rel fp32 on y"
