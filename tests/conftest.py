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


def render_both_ways(pkg, ctx, cam, params, want, cnt_o, start=None, nan_ok=False, what=""):
    """Renders `params` five times and holds every image to the oracle's bit for bit:
      counted   with work counters, otherwise as asked
      uncut     with work counters and RT_FLAG_NO_CULL: the reference's literal traversal, lane by lane
      timed     exactly as asked, WITHOUT counters -- the instantiation a timed run executes (its traversal steps end in
                hand-scheduled code that counting builds do not take)
      packets / counted packets   as timed / counted with RT_FLAG_PACKETS: the primary rays of every tile traced as one packet
                (hand-written loop / its C++ twin) and the paths continued from the hit queues, whether or not the
                library's own estimate would have chosen that for this scene and image size
    The uncut render must also show the oracle's six work counters exactly.  The other counting renders may skip subtrees a
    bounce ray cannot reach: they must trace the same rays, find the same hits and finish the same paths (counters 2, 3, 5)
    with no more box tests, triangle tests and node visits (0, 1, 4) than the reference runs.  Returns the default counting
    render and its counters."""
    out = None
    default_pipeline = not (params.flags & (pkg.RT_FLAG_MEGAKERNEL | pkg.RT_FLAG_WAVEFRONT))
    variants = ("counted", "uncut", "timed") + (("packets", "counted packets") if default_pipeline else ())
    for variant in variants:
        p = type(params).from_buffer_copy(params)
        if variant in ("counted", "uncut", "counted packets"):
            p.flags |= pkg.RT_FLAG_COUNTERS
        if variant == "uncut":
            p.flags |= pkg.RT_FLAG_NO_CULL
        if "packets" in variant:
            p.flags |= pkg.RT_FLAG_PACKETS
        ctx.reset_counters()
        got = ctx.render(cam, p, None if start is None else start.copy())
        cnt = ctx.counters()
        same = bits(got) == bits(want)
        if nan_ok:
            same |= np.isnan(got) & np.isnan(want)
        if not same.all():
            bad = (~same).any(axis=-1)
            ys, xs = np.nonzero(bad)
            raise AssertionError(f"{what} ({variant}): {int(bad.sum())} of {bad.size} pixels differ; first at "
                                 f"(x={xs[0]}, y={ys[0]}): got {got[ys[0], xs[0]]} want {want[ys[0], xs[0]]}")
        if variant == "uncut":
            assert np.array_equal(cnt, cnt_o), f"{what}: work counters of the uncut render differ: gpu {cnt} oracle {cnt_o}"
        elif p.flags & pkg.RT_FLAG_COUNTERS:
            assert np.array_equal(cnt[[2, 3, 5]], cnt_o[[2, 3, 5]]), f"{what} ({variant}): rays/hits/paths differ: gpu {cnt} oracle {cnt_o}"
            assert (cnt[[0, 1, 4]] <= cnt_o[[0, 1, 4]]).all(), f"{what} ({variant}): more work than the reference: gpu {cnt} oracle {cnt_o}"
            if variant == "counted":
                out = (got, cnt)
            else:   # the packets give every ray the reference's own visit order: same work as lane by lane
                assert np.array_equal(cnt, out[1]), f"{what}: packets counted {cnt}, lane by lane {out[1]}"
        else:
            assert not cnt.any(), f"{what}: a render without RT_FLAG_COUNTERS counted {cnt}"
    return out
