"""The N > 1 path on CPU: world-size-2 gloo job with the oracle standing in for the GPU renderer."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_partition_is_exact(pkg):
    from opengl_raytracer_amd import partition

    for world in (1, 2, 4, 8):
        for first, steps in ((0, 1), (0, 5), (3, 4)):
            assert partition.all_frames(world, first, steps) == list(range(first * world, (first + steps) * world))
    with pytest.raises(ValueError):
        partition.rank_frames(2, 2, 0, 1)


@pytest.mark.parametrize("reduce_bands,launch_frames", [(1, 16), (3, 16), (3, 2)])
def test_two_rank_sum_reduce_matches_sequential_mean(tmp_path, reduce_bands, launch_frames):
    """partition.render_partitioned (what bench.py runs on RCCL) under gloo with the oracle as renderer: one reduce of
    the whole image; the image finished in three row bands whose reduces overlap the next band's rendering; and the same
    with a launch of two frames, so that the first two of the rank's three frames go out as a whole-image launch and only
    the last launch's frame is rendered band by band."""
    out = tmp_path / "result.npz"
    port = str(29641 + 10 * reduce_bands + launch_frames)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "tests", "dist_worker.py"), "3", str(out), "samples", str(reduce_bands),
           str(launch_frames)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = np.load(out)
    assert float(res["err"]) < 1e-6            # sum/N vs running mean: fp32 rounding only
    assert float(res["rays"]) == float(res["rays_seq"])   # every frame rendered exactly once
    head_calls = 1 if (reduce_bands > 1 and launch_frames < 3) else 0
    assert int(res["calls"]) == reduce_bands + head_calls and bool(res["same_as_unbanded"])


def test_two_rank_tile_partition_gathers_the_exact_image(tmp_path):
    out = tmp_path / "tiles.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29643", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29643", os.path.join(ROOT, "tests", "dist_worker.py"), "3", str(out), "tiles"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = np.load(out)
    assert bool(res["same"])                               # bands are disjoint: the gathered image is bit-exact
    assert float(res["rays"]) == float(res["rays_seq"])


def test_band_rows_partition_the_image(pkg):
    from opengl_raytracer_amd import partition

    for height in (1, 7, 8, 9, 217, 1080, 2160):
        for count in (1, 2, 3, 8):
            rows = [partition.band_rows(i, count, height) for i in range(count)]
            assert rows[0][0] == 0 and rows[-1][1] == height
            assert all(rows[i][1] == rows[i + 1][0] for i in range(count - 1))
            assert all(r0 % 8 == 0 for r0, _ in rows if r0 < height)
            for interleaved in (False, True):
                covered = [0] * height
                for i in range(count):
                    for r0, r1 in partition.band_row_ranges(i, count, height, interleaved):
                        assert 0 <= r0 < r1 <= height
                        for r in range(r0, r1):
                            covered[r] += 1
                assert all(c == 1 for c in covered)
