"""Shared fixtures.  Tests marked ``gpu`` need an MI355X and call the product through its C ABI;
everything else runs on the CPU (oracle, host ingest, ABI surface, gloo multi-process logic)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

FIXTURES = os.path.join(ROOT, "fixtures")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def scenes(pkg):
    """Lazily built scenes by fixture name (teapot / model / dragon80k), colour seed 1."""
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pkg.load_scene(os.path.join(FIXTURES, name + ".obj"), colour_seed=1)
        return cache[name]

    return get


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    """One render context on device 0 for the whole session (GPU tests only)."""
    ctx = pkg.RenderContext(0)
    yield ctx
    ctx.close()


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a).view(np.uint32)


def rrmse(a: np.ndarray, b: np.ndarray) -> float:
    """Relative RMSE over RGB: sqrt(sum (a-b)^2 / sum b^2)."""
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b**2).sum()))
