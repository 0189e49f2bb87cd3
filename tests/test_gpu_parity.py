"""GPU parity tests: the HIP path through the C ABI against the CPU oracle on the same inputs.

Bars: integer / index results (primary-hit triangle ids, work counters) bit-exact; floating-point
results bit-exact as well, because device and oracle share the pinned arithmetic of
include/rt_math.h -- and in any case within the stated tolerance of relative RMSE <= 1e-3 on the
accumulated image, which is asserted separately so that a future non-bit-exact kernel would still
be held to it.
"""
import numpy as np
import pytest

from conftest import bits, render_both_ways, rrmse

pytestmark = pytest.mark.gpu

RRMSE_TOL = 1e-3

CAMERAS = {
    "teapot": dict(pos=(0.0, 0.0, -2.0)),            # the reference's CAMERA_START_POS
    "dragon80k": dict(pos=(0.0, 0.0, -5.5)),
    "model": dict(pos=(0.0, 0.0, -2.0)),
}


def _assert_image_equal(got, want, what):
    if not np.array_equal(bits(got), bits(want)):
        diff = (bits(got) != bits(want)).any(axis=-1)
        ys, xs = np.nonzero(diff)
        raise AssertionError(
            f"{what}: {int(diff.sum())} of {diff.size} pixels differ; first at (x={xs[0]}, y={ys[0]}): "
            f"got {got[ys[0], xs[0]]} want {want[ys[0], xs[0]]}; rRMSE={rrmse(got, want):.3e}"
        )


def test_math_device_equals_host(pkg, oracle, gpu_ctx):
    rng = np.random.RandomState(0)
    x = np.concatenate([
        rng.uniform(0.0, 1.0, 20000), rng.uniform(0.0, 6.3, 20000), rng.uniform(-40.0, 2.0, 20000),
        10.0 ** rng.uniform(-38, 38, 20000), -(10.0 ** rng.uniform(-10, 3, 5000)),
        np.array([0.0, 1.0, 2.0 ** -32, 1.0 - 2.0 ** -24, np.inf, np.nan, 1e-45, 1e-40, 6.2831852, 0.5]),
    ]).astype(np.float32)
    dev = gpu_ctx.selftest_math(x)
    host = oracle.math(x)
    names = ["log", "cos", "exp", "sqrt", "rcp", "pow0.35"]
    for k, name in enumerate(names):
        d, h = bits(dev[k]), bits(host[k])
        nan_both = np.isnan(dev[k]) & np.isnan(host[k])
        bad = (d != h) & ~nan_both
        assert not bad.any(), f"{name}: {int(bad.sum())} inputs differ, e.g. x={x[bad][0]!r}: dev {dev[k][bad][0]!r} host {host[k][bad][0]!r}"


@pytest.mark.parametrize("name", ["teapot", "model", "dragon80k"])
def test_normals(pkg, oracle, gpu_ctx, scenes, name):
    scene = scenes(name)
    gpu_ctx.upload_scene(scene)
    got = gpu_ctx.get_normals()
    want = oracle.normals(scene)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("name,size", [("model", (1920, 1080)), ("teapot", (1920, 1080)), ("dragon80k", (1920, 1080)),
                                       ("teapot", (333, 217))])
def test_primary_hits_bit_exact(pkg, oracle, gpu_ctx, scenes, name, size):
    scene = scenes(name)
    cam = pkg.make_camera(**CAMERAS[name])
    gpu_ctx.upload_scene(scene)
    ids, t = gpu_ctx.primary_hits(cam, *size)
    ids_o, t_o = oracle.primary_hits(scene, cam, *size)
    assert np.array_equal(ids, ids_o), f"{int((ids != ids_o).sum())} primary-hit ids differ"
    assert np.array_equal(bits(t), bits(t_o))


def _pipeline_flags(pkg, name):
    return {"paths": 0, "wavefront": pkg.RT_FLAG_WAVEFRONT, "megakernel": pkg.RT_FLAG_MEGAKERNEL}[name]


@pytest.mark.parametrize("flags_name", ["paths", "wavefront", "megakernel"])
@pytest.mark.parametrize("name", ["teapot", "dragon80k"])
def test_frame0_full_size(pkg, oracle, gpu_ctx, scenes, name, flags_name):
    """Benchmark configurations 0 and 1: 1920x1080, 1 spp, 100 bounces, Russian roulette, bug-compatible."""
    scene = scenes(name)
    cam = pkg.make_camera(**CAMERAS[name])
    params = pkg.make_params(1920, 1080, flags=_pipeline_flags(pkg, flags_name))
    gpu_ctx.upload_scene(scene)
    want, cnt_o = oracle.render(scene, cam, params)
    got, cnt = render_both_ways(pkg, gpu_ctx, cam, params, want, cnt_o, what=f"{name} {flags_name}")
    assert rrmse(got, want) <= RRMSE_TOL
    assert np.all(got[..., 3] == 2.0)
    if flags_name == "paths" and name == "dragon80k":   # the cull is at work: a third fewer box tests
        assert cnt[0] < 0.75 * cnt_o[0]


@pytest.mark.parametrize("variant", [
    dict(frames=4),
    dict(frames=11),                      # more than one batch of the persistent kernel
    dict(frames=4, flags="wave"),
    dict(frames=3, flags="fix"),
    dict(frames=3, flags="wave+fix"),
    dict(frames=2, flags="mega"),
    dict(frames=2, flags="mega+fix"),
    dict(frames=2, rays_per_pixel=3),     # samples of a pixel traced one after the other by its lane of the path kernel
    dict(frames=5, rays_per_pixel=4),     # ... over more than one launch (16 staged samples per pixel and launch)
    dict(frames=2, rays_per_pixel=2, flags="fix"),
    dict(frames=2, rays_per_pixel=3, flags="sum", frame_stride=2, frame_base=0),
    dict(frames=2, rays_per_pixel=2, frame_base=3),
    dict(frames=1, rays_per_pixel=17),    # more samples than a launch stages: the megakernel takes it
    dict(frames=2, rays_per_pixel=3, flags="mega"),
    dict(frames=1, ray_bounces=2),
    dict(frames=1, ray_bounces=1),
    dict(frames=1, ray_bounces=0),
    dict(frames=1, ray_bounces=5, flags="fix"),
    dict(frames=2, ray_bounces=3, flags="wave"),
    dict(frames=1, ray_bounces=2, flags="wave"),
    dict(frames=3, flags="wave+sum", frame_stride=2, frame_base=0),
    dict(frames=1, render_mode=2),
    dict(frames=1, render_mode=3),
    dict(frames=1, render_mode=4),
    dict(frames=3, flags="sum", frame_stride=2, frame_base=0),
    dict(frames=2, frame_base=5, frame_stride=3),
])
@pytest.mark.parametrize("name,size", [("teapot", (480, 270)), ("dragon80k", (333, 217))])
def test_variants_small(pkg, oracle, gpu_ctx, scenes, name, size, variant):
    variant = dict(variant)
    fl = variant.pop("flags", "")
    flags = 0
    if "fix" in fl:
        flags |= pkg.RT_FLAG_FIX_INVDIR
    if "mega" in fl:
        flags |= pkg.RT_FLAG_MEGAKERNEL
    if "wave" in fl:
        flags |= pkg.RT_FLAG_WAVEFRONT
    if "sum" in fl:
        flags |= pkg.RT_FLAG_SUM
    scene = scenes(name)
    cam = pkg.make_camera(pitch=0.1, yaw=-0.15, **CAMERAS[name])
    params = pkg.make_params(size[0], size[1], flags=flags, **variant)
    gpu_ctx.upload_scene(scene)
    start = None
    if params.frame_base != 0:
        start = np.random.RandomState(3).uniform(0, 1, (size[1], size[0], 4)).astype(np.float32)
    want, cnt_o = oracle.render(scene, cam, params, None if start is None else start.copy())
    render_both_ways(pkg, gpu_ctx, cam, params, want, cnt_o, start=start, what=f"{name} {variant} {fl}")


def test_pitched_camera_sees_quads(pkg, oracle, gpu_ctx, scenes):
    """models/model.obj: the y = +-10 quads are outside the default frustum; look up at one."""
    scene = scenes("model")
    cam = pkg.make_camera(pos=(0.0, 0.0, 0.0), pitch=1.2)
    gpu_ctx.upload_scene(scene)
    ids, t = gpu_ctx.primary_hits(cam, 640, 360)
    ids_o, t_o = oracle.primary_hits(scene, cam, 640, 360)
    assert np.array_equal(ids, ids_o) and np.array_equal(bits(t), bits(t_o))
    assert len(np.unique(ids[ids >= 0])) >= 2


def test_resume_accumulation_equals_one_shot(pkg, gpu_ctx, scenes):
    """frames 0..5 in one call == frames 0..2 then 3..5 resumed from the returned running mean."""
    scene = scenes("teapot")
    cam = pkg.make_camera()
    gpu_ctx.upload_scene(scene)
    one = gpu_ctx.render(cam, pkg.make_params(320, 180, frames=6))
    part = gpu_ctx.render(cam, pkg.make_params(320, 180, frames=3))
    part = gpu_ctx.render(cam, pkg.make_params(320, 180, frames=3, frame_base=3), part)
    assert np.array_equal(bits(one), bits(part))


@pytest.mark.parametrize("name", ["teapot", "model", "dragon80k"])
def test_gpu_against_committed_golden_vectors(pkg, gpu_ctx, scenes, name):
    """The HIP path against the committed fixtures directly (tests/golden/oracle_*.npz, written by make_golden.py): primary
    hit ids and distances, three accumulated frames with their work counters, two frames with RT_FLAG_FIX_INVDIR."""
    import os
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", f"oracle_{name}.npz"))
    w, h = [int(x) for x in g["size"]]
    cam = pkg.make_camera(pos=tuple(float(x) for x in g["pos"]))
    gpu_ctx.upload_scene(scenes(name))
    ids, t = gpu_ctx.primary_hits(cam, w, h)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(bits(t), bits(g["t"]))
    for frames, extra, key_img, key_cnt in ((3, 0, "img", "counters"), (2, pkg.RT_FLAG_FIX_INVDIR, "img_fix", "counters_fix")):
        gpu_ctx.reset_counters()
        img = gpu_ctx.render(cam, pkg.make_params(w, h, frames=frames, flags=extra | pkg.RT_FLAG_COUNTERS | pkg.RT_FLAG_NO_CULL))
        assert np.array_equal(bits(img), bits(g[key_img]))
        assert np.array_equal(gpu_ctx.counters(), g[key_cnt])
        plain = gpu_ctx.render(cam, pkg.make_params(w, h, frames=frames, flags=extra))   # the non-counting kernels
        assert np.array_equal(bits(plain), bits(g[key_img]))


def test_single_frame_calls_pipeline_and_mix_with_batches(pkg, oracle, gpu_ctx, scenes):
    """rt_raytrace_frame calls are pipelined inside the library (the path kernels of consecutive frames overlap on
    internal streams).  A frame-by-frame loop, a batched call in the middle of it, a scene swap and a counting call must
    all see each other's results: the final image equals the oracle's running mean of the same frames bit for bit."""
    W, H = 320, 180
    cam = pkg.make_camera()
    teapot, dragon = scenes("teapot"), scenes("dragon80k")
    bufs = [pkg.DeviceBuffer(W * H * 16), pkg.DeviceBuffer(W * H * 16)]
    p1 = pkg.make_params(W, H)
    for scene in (teapot, dragon, teapot):
        gpu_ctx.upload_scene(scene)
        f = 0
        for _ in range(5):                                  # frames 0..4, one call each
            gpu_ctx.raytrace_frame(cam, p1, f, bufs[(f + 1) & 1].ptr, bufs[f & 1].ptr)
            f += 1
        last = bufs[(f - 1) & 1]
        gpu_ctx.render_device(cam, pkg.make_params(W, H, frames=4, frame_base=f), last.ptr)   # frames 5..8 batched, in place
        f += 4
        other = bufs[0] if last is bufs[1] else bufs[1]
        gpu_ctx.raytrace_frame(cam, pkg.make_params(W, H, flags=pkg.RT_FLAG_COUNTERS), f, last.ptr, other.ptr)  # frame 9, counting
        f += 1
        for _ in range(3):                                  # frames 10..12
            src, dst = (other, last) if (f & 1) == 0 else (last, other)
            gpu_ctx.raytrace_frame(cam, p1, f, src.ptr, dst.ptr)
            f += 1
        gpu_ctx.synchronize()
        got = dst.to_array((H, W, 4), np.float32)
        want, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=f))
        assert np.array_equal(bits(got), bits(want))
    for b in bufs:
        b.free()


def test_pipelined_frames_of_two_image_sizes_do_not_share_staging(pkg, oracle, gpu_ctx, scenes):
    """Frame-by-frame calls of two image sizes interleaved on one context without any host synchronisation: the path
    kernels of up to two calls are in flight together, each in its own fixed share of the staging area, so a large
    frame launched while a small one still waits for its accumulate step must not overwrite the small one's samples."""
    scene = scenes("dragon80k")
    cam = pkg.make_camera(**CAMERAS["dragon80k"])
    gpu_ctx.upload_scene(scene)
    big, small = (1920, 1080), (64, 64)
    pb, ps = pkg.make_params(*big), pkg.make_params(*small)
    gpu_ctx.render_device(cam, pkg.make_params(*big, frames=16), pkg.DeviceBuffer(big[0] * big[1] * 16).ptr)  # sizes the staging area
    gpu_ctx.synchronize()
    B = [pkg.DeviceBuffer(big[0] * big[1] * 16) for _ in range(2)]
    S = [pkg.DeviceBuffer(small[0] * small[1] * 16) for _ in range(2)]
    n = 6
    for f in range(n):       # big f, small f, big f+1, ... : consecutive launches alternate between the two slots
        gpu_ctx.raytrace_frame(cam, pb, f, B[(f + 1) & 1].ptr, B[f & 1].ptr)
        gpu_ctx.raytrace_frame(cam, ps, f, S[(f + 1) & 1].ptr, S[f & 1].ptr)
        gpu_ctx.raytrace_frame(cam, ps, f, S[(f + 1) & 1].ptr, S[f & 1].ptr)   # same frame again: shifts the slot parity
    gpu_ctx.synchronize()
    got_b = B[(n - 1) & 1].to_array((big[1], big[0], 4), np.float32)
    got_s = S[(n - 1) & 1].to_array((small[1], small[0], 4), np.float32)
    want_s, _ = oracle.render(scene, cam, pkg.make_params(*small, frames=n))
    assert np.array_equal(bits(got_s), bits(want_s))
    want_b = gpu_ctx.render(cam, pkg.make_params(*big, frames=n))
    assert np.array_equal(bits(got_b), bits(want_b))
    for b in B + S:
        b.free()


@pytest.mark.parametrize("fix", [False, True])
def test_reuse_primary_is_bit_identical_and_traces_fewer_rays(pkg, oracle, gpu_ctx, scenes, fix):
    """RT_FLAG_REUSE_PRIMARY (opt-in): the primary hits are traced once per (camera, size, scene) and every path starts at
    its first bounce.  Same pixels as the oracle; the counters show the primary pass once plus the bounce rays; a camera
    move, a scene swap and a frame-by-frame continuation all invalidate or reuse the stored hits correctly."""
    W, H, n = 333, 217, 5
    flags = pkg.RT_FLAG_REUSE_PRIMARY | (pkg.RT_FLAG_FIX_INVDIR if fix else 0)
    base = pkg.RT_FLAG_FIX_INVDIR if fix else 0
    for name in ("dragon80k", "teapot"):
        scene = scenes(name)
        gpu_ctx.upload_scene(scene)
        for yaw in (0.0, 0.2):                       # the second camera must not see the first one's hits
            cam = pkg.make_camera(yaw=yaw, **CAMERAS[name])
            want, cnt_o = oracle.render(scene, cam, pkg.make_params(W, H, frames=n, flags=base))
            gpu_ctx.reset_counters()
            got = gpu_ctx.render(cam, pkg.make_params(W, H, frames=n, flags=flags | pkg.RT_FLAG_COUNTERS))
            cnt = gpu_ctx.counters()
            _assert_image_equal(got, want, f"{name} yaw={yaw} reuse")
            # hits as in the oracle; the primary pass adds one "path" per pixel; rays = all rays minus (n - 1) x the
            # primary rays (one per pixel and frame)
            assert cnt[5] == cnt_o[5] + W * H and cnt[3] == cnt_o[3]
            assert cnt[2] == cnt_o[2] - (n - 1) * W * H
            plain = gpu_ctx.render(cam, pkg.make_params(W, H, frames=n, flags=flags))          # non-counting kernels
            _assert_image_equal(plain, want, f"{name} yaw={yaw} reuse, no counters")
        # frame-by-frame continuation with the same camera: the stored hits serve every call
        cont = gpu_ctx.render(cam, pkg.make_params(W, H, frames=2, frame_base=n, flags=flags), got.copy())
        want2, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=n + 2, flags=base))
        _assert_image_equal(cont, want2, f"{name} continuation")
    # one call per frame (the reference's loop): the first call traces the primary hits, the others reuse them
    cam = pkg.make_camera(yaw=-0.1, **CAMERAS["teapot"])
    bufs = [pkg.DeviceBuffer(W * H * 16), pkg.DeviceBuffer(W * H * 16)]
    p1 = pkg.make_params(W, H, flags=flags)
    for f in range(4):
        gpu_ctx.raytrace_frame(cam, p1, f, bufs[(f + 1) & 1].ptr, bufs[f & 1].ptr)
    gpu_ctx.synchronize()
    want, _ = oracle.render(scenes("teapot"), cam, pkg.make_params(W, H, frames=4, flags=base))
    _assert_image_equal(bufs[1].to_array((H, W, 4), np.float32), want, "frame loop with reuse")
    for b in bufs:
        b.free()
    # bands and the flag together; other pipelines refuse it
    cam = pkg.make_camera(**CAMERAS["teapot"])
    whole = gpu_ctx.render(cam, pkg.make_params(W, H, frames=3, flags=flags))
    union = np.zeros_like(whole)
    from opengl_raytracer_amd import partition
    for g in range(3):
        part = gpu_ctx.render(cam, pkg.make_params(W, H, frames=3, flags=flags, band=(g, 3), interleaved=True))
        for r0, r1 in partition.band_row_ranges(g, 3, H, True):
            union[r0:r1] = part[r0:r1]
    assert np.array_equal(bits(union), bits(whole))
    with pytest.raises(RuntimeError):
        gpu_ctx.render(cam, pkg.make_params(64, 64, flags=flags | pkg.RT_FLAG_MEGAKERNEL))


def test_sample_partition_sum_matches_running_mean(pkg, gpu_ctx, scenes):
    """The multi-GPU decomposition on one GPU: G interleaved frame subsets summed, divided by the frame
    count, against the sequential running mean -- equal to fp32 rounding, far inside 1e-3."""
    scene = scenes("teapot")
    cam = pkg.make_camera()
    gpu_ctx.upload_scene(scene)
    n, G = 8, 4
    mean = gpu_ctx.render(cam, pkg.make_params(320, 180, frames=n))
    total = np.zeros_like(mean)
    for g in range(G):
        # a non-zero frame_base continues from the passed buffer: zeros give this rank's own sum
        part = np.zeros_like(mean)
        total += gpu_ctx.render(cam, pkg.make_params(320, 180, frames=n // G, frame_base=g, frame_stride=G,
                                                     flags=pkg.RT_FLAG_SUM), part)
    assert rrmse(total / n, mean) < 1e-6


def test_malformed_scenes_are_rejected(pkg, gpu_ctx, scenes):
    good = scenes("model")
    for what in ("child_out_of_range", "tri_out_of_range", "cycle"):
        nodes = good.nodes.copy()
        inner = np.nonzero(nodes[:, 8] == 0.0)[0]
        leaf = np.nonzero(nodes[:, 8] == 1.0)[0]
        if what == "child_out_of_range":
            nodes[inner[0], 9] = 1e6
        elif what == "tri_out_of_range":
            nodes[leaf[0], 10] = good.num_tris + 5
        else:
            nodes[inner[-1], 9] = 0.0
        bad = pkg.Scene(nodes, good.triv0, good.triv1, good.triv2, good.colours)
        with pytest.raises(RuntimeError):
            gpu_ctx.upload_scene(bad)
    gpu_ctx.upload_scene(good)  # context still usable


def test_render_without_scene_fails(pkg):
    ctx = pkg.RenderContext(0)
    with pytest.raises(RuntimeError, match="no scene"):
        ctx.render(pkg.make_camera(), pkg.make_params(64, 64))
    ctx.close()


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(x) for x in f.readline().split()]
        assert float(f.readline()) < 0
        return np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)


@pytest.mark.parametrize("batch", [False, True])
def test_headless_driver_loop_semantics(pkg, oracle, scenes, tmp_path, batch):
    """The C++ host mirror (host/rt_headless): scripted camera input resets the accumulation, frames
    accumulate through ping-pong buffers, and the dumped image equals the oracle's for the final camera."""
    import os
    import subprocess
    from conftest import ROOT, FIXTURES
    exe = os.path.join(ROOT, "opengl-raytracer_amd", "host", "rt_headless")
    if not os.path.exists(exe):
        pytest.skip("host/rt_headless not built")
    out = tmp_path / "frame.pfm"
    cmd = [exe, os.path.join(FIXTURES, "teapot.obj"), "--size", "320", "180", "--script", "wait:3;yaw:0.2,w:0.5,wait:2",
           "--out", str(out)] + (["--batch"] if batch else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Average FPS" in r.stdout and "Raytracer initialised" in r.stdout
    got = _read_pfm(out)
    cam = pkg.make_camera(pos=(0.0, 0.0, -1.0), yaw=0.2)   # W held 0.5 s at 2 units/s along the old view axis
    want, _ = oracle.render(scenes("teapot"), cam, pkg.make_params(320, 180, frames=2))
    assert np.array_equal(bits(got), bits(np.ascontiguousarray(want[..., :3])))


def _headless(args, timeout=180):
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "opengl-raytracer_amd", "host", "rt_headless")
    if not os.path.exists(exe):
        pytest.skip("host/rt_headless not built")
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_headless_script_mode_keys(pkg, oracle, scenes, tmp_path):
    """Keys 1-4 of the reference (main.cpp:232-247) as "mode:N" script items: the render mode changes and the
    accumulation restarts.  Two frames in mode 1, then key 3 and two frames of the box-test heat map; back to mode 1."""
    import os
    from conftest import FIXTURES
    obj = os.path.join(FIXTURES, "teapot.obj")
    out = tmp_path / "m3.pfm"
    _headless([obj, "--size", "320", "180", "--script", "wait:2;mode:3,wait:2", "--out", str(out)])
    cam = pkg.make_camera()
    want, _ = oracle.render(scenes("teapot"), cam, pkg.make_params(320, 180, frames=2, render_mode=3))
    assert np.array_equal(bits(_read_pfm(out)), bits(np.ascontiguousarray(want[..., :3])))
    out = tmp_path / "m1.pfm"
    _headless([obj, "--size", "320", "180", "--script", "mode:4,wait:1;mode:1,wait:3", "--out", str(out)])
    want, _ = oracle.render(scenes("teapot"), cam, pkg.make_params(320, 180, frames=3))
    assert np.array_equal(bits(_read_pfm(out)), bits(np.ascontiguousarray(want[..., :3])))


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_headless_rank_world_partition(pkg, oracle, scenes, tmp_path, mode):
    """The multi-GPU split of the C++ host on one GPU: three ranks run one after the other, each writes its partial
    image, tools/combine_partials.py puts them together.  Tiles: bit-equal to the unpartitioned image; samples: equal to
    the running mean up to fp32 rounding."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, FIXTURES
    obj = os.path.join(FIXTURES, "teapot.obj")
    world, frames = 3, 7
    parts = []
    for r in range(world):
        part = tmp_path / f"part{r}.pfm"
        _headless([obj, "--size", "320", "184", "--script", f"yaw:0.1,wait:{frames}", "--rank", str(r), "--world", str(world),
                   "--partition", mode, "--out", str(part)])
        parts.append(str(part))
    out = tmp_path / "combined.pfm"
    tool = os.path.join(ROOT, "opengl-raytracer_amd", "tools", "combine_partials.py")
    cmd = [sys.executable, tool, mode] + ([str(frames)] if mode == "samples" else []) + [str(out)] + parts
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    got = _read_pfm(out)
    want, _ = oracle.render(scenes("teapot"), pkg.make_camera(yaw=0.1), pkg.make_params(320, 184, frames=frames))
    want = np.ascontiguousarray(want[..., :3])
    if mode == "tiles":
        assert np.array_equal(bits(got), bits(want))
    else:
        err = float(np.sqrt(((got.astype(np.float64) - want) ** 2).sum() / (want.astype(np.float64) ** 2).sum()))
        assert err < 1e-6


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_headless_rccl_reduce_single_rank(pkg, oracle, scenes, tmp_path, mode):
    """The C++ host's own collective: rt_headless --rccl creates an RCCL communicator (here of one rank, the only size a
    one-GPU box allows), renders its share with RT_FLAG_SUM / its band, calls ncclReduce on the device frame buffer on the
    context's stream, and rank 0 scales and writes the image."""
    import os
    from conftest import FIXTURES
    out = tmp_path / "rccl.pfm"
    _headless([os.path.join(FIXTURES, "teapot.obj"), "--size", "320", "184", "--script", "yaw:0.1,wait:5", "--world", "1", "--rank", "0",
               "--partition", mode, "--rccl", str(tmp_path / "nccl_id"), "--out", str(out)], timeout=300)
    got = _read_pfm(out)
    want, _ = oracle.render(scenes("teapot"), pkg.make_camera(yaw=0.1), pkg.make_params(320, 184, frames=5))
    want = np.ascontiguousarray(want[..., :3])
    if mode == "tiles":
        assert np.array_equal(bits(got), bits(want))
    else:
        err = float(np.sqrt(((got.astype(np.float64) - want) ** 2).sum() / (want.astype(np.float64) ** 2).sum()))
        assert err < 1e-6


def test_config0_teapot_primary_plus_one_bounce(pkg, oracle, gpu_ctx, scenes):
    """Benchmark configuration 0: teapot, 1920x1080, 1 spp, primary + 1 diffuse bounce (ray_bounces = 2)."""
    scene = scenes("teapot")
    cam = pkg.make_camera()
    p = pkg.make_params(1920, 1080, ray_bounces=2)
    gpu_ctx.upload_scene(scene)
    want, cnt_o = oracle.render(scene, cam, p)
    got, _ = render_both_ways(pkg, gpu_ctx, cam, p, want, cnt_o, what="teapot 2 bounces")
    assert rrmse(got, want) <= RRMSE_TOL


def test_config2_4k_progressive_and_sample_partition(pkg, oracle, gpu_ctx, scenes):
    """Benchmark configuration 2: dragon stand-in at 3840x2160, progressive accumulation, sample-partitioned.
    Bit-exact against the oracle for the first two frames at full size; then the full 256 spp through
    size-independent properties: (1) resuming in two halves equals one shot, (2) eight interleaved partial SUMs
    divided by 256 equal the sequential running mean to fp32 rounding (the multi-GPU decomposition)."""
    scene = scenes("dragon80k")
    cam = pkg.make_camera(**CAMERAS["dragon80k"])
    W, H = 3840, 2160
    gpu_ctx.upload_scene(scene)
    got = gpu_ctx.render(cam, pkg.make_params(W, H, frames=2))
    want, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=2))
    _assert_image_equal(got, want, "4K, 2 frames")
    del want
    full = gpu_ctx.render(cam, pkg.make_params(W, H, frames=256))
    half = gpu_ctx.render(cam, pkg.make_params(W, H, frames=128))
    half = gpu_ctx.render(cam, pkg.make_params(W, H, frames=128, frame_base=128), half)
    assert np.array_equal(bits(full), bits(half))
    total = np.zeros_like(full, dtype=np.float64)
    G = 8
    for g in range(G):
        part = np.zeros_like(full)
        total += gpu_ctx.render(cam, pkg.make_params(W, H, frames=256 // G, frame_base=g, frame_stride=G,
                                                     flags=pkg.RT_FLAG_SUM), part)
    # 256 running-mean updates in fp32 drift by ~2e-6 relative; the stated tolerance is 1e-3
    assert rrmse((total / 256.0).astype(np.float32), full) < 1e-5
    assert np.all(full[..., 3] == 2.0)


def _tiny_scene(pkg, tris, nodes=None):
    """Scene from explicit triangles; nodes default to a single root leaf covering all of them."""
    tris = np.asarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    T = tris.shape[0]
    pad = lambda v: np.concatenate([v, np.zeros((T, 1), np.float32)], axis=1)
    if nodes is None:
        lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
        nodes = np.array([[*lo, 0, *hi, 0, 1.0, 0.0, T - 1, 0]], dtype=np.float32)
    return pkg.Scene(nodes, pad(tris[:, 0]), pad(tris[:, 1]), pad(tris[:, 2]), pkg.make_colours(5, T))


@pytest.mark.parametrize("case", ["root_is_leaf", "empty_leaf", "empty_leaf_inverted_box", "degenerate_triangle", "nan_column"])
@pytest.mark.parametrize("pipeline", ["paths", "wavefront", "megakernel"])
def test_edge_scenes(pkg, oracle, gpu_ctx, case, pipeline):
    """Ragged inputs: a tree that is a single leaf, a leaf with an empty triangle range (b < a) -- also with its box given
    upside down (max < min), which the slab test of raytrace.glsl:113-125 reads like the box the right way up --, a
    degenerate triangle (NaN normal, normals.glsl:18), and boxes ending exactly on the plane x = camera.x (0 * inf = NaN
    in the slab test of the centre pixel column)."""
    big = [[-3, -2, 5], [3, -2, 5], [0, 3, 6]]
    if case == "root_is_leaf":
        scene = _tiny_scene(pkg, [big])
    elif case == "degenerate_triangle":
        scene = _tiny_scene(pkg, [big, [[0, 0, 4], [0, 0, 4], [1, 1, 4]]])
    elif case == "nan_column":
        scene = _tiny_scene(pkg, [[[0, -2, 5], [3, -2, 5], [0, 3, 6]], [[-3, -2, 7], [0, -2, 7], [0, 3, 7]]])
    else:
        lo, hi = [-3, -2, 5], [3, 3, 6]
        nodes = np.array([[*lo, 0, *hi, 0, 0.0, 1.0, 2.0, 0],      # root: children 1, 2
                          [*lo, 0, *hi, 0, 1.0, 0.0, 0.0, 0],      # leaf with triangle 0
                          [*lo, 0, *hi, 0, 1.0, 1.0, 0.0, 0]],     # leaf with the empty range [1, 0]
                         dtype=np.float32)
        if case == "empty_leaf_inverted_box":
            nodes[2, 0:3], nodes[2, 4:7] = hi, lo
        scene = _tiny_scene(pkg, [big], nodes)
    cam = pkg.make_camera(pos=(0.0, 0.0, -2.0))
    p = pkg.make_params(128, 96, frames=2, flags=_pipeline_flags(pkg, pipeline))
    gpu_ctx.upload_scene(scene)
    want, cnt_o = oracle.render(scene, cam, p)
    render_both_ways(pkg, gpu_ctx, cam, p, want, cnt_o, nan_ok=True, what=f"{case} {pipeline}")
    ids, t = gpu_ctx.primary_hits(cam, 128, 96)
    ids_o, t_o = oracle.primary_hits(scene, cam, 128, 96)
    assert np.array_equal(ids, ids_o) and np.array_equal(bits(t), bits(t_o))
    assert (ids >= 0).any()


def test_api_argument_errors(pkg, gpu_ctx, scenes):
    import ctypes as C
    lib = gpu_ctx.lib
    gpu_ctx.upload_scene(scenes("model"))
    cam = pkg.make_camera()
    bad = [pkg.make_params(0, 10), pkg.make_params(10, -1), pkg.make_params(64, 64, rays_per_pixel=0),
           pkg.make_params(64, 64, render_mode=9), pkg.make_params(20000, 20000)]
    out = np.zeros((64, 64, 4), dtype=np.float32)
    for p in bad:
        rc = lib.rt_render(gpu_ctx.ctx, C.byref(cam), C.byref(p), pkg.fptr(out))
        assert rc in (-1, -4) and len(lib.rt_last_error(gpu_ctx.ctx)) > 0
    assert lib.rt_render(gpu_ctx.ctx, None, C.byref(pkg.make_params(64, 64)), pkg.fptr(out)) == -1
    assert lib.rt_render(gpu_ctx.ctx, C.byref(cam), C.byref(pkg.make_params(64, 64)), None) == -1
    assert lib.rt_upload_scene(gpu_ctx.ctx, None) == -1
    assert lib.rt_create(99, C.byref(C.c_void_p())) == -1
    # the context is still usable afterwards
    gpu_ctx.render(cam, pkg.make_params(64, 64))


def test_two_contexts_are_independent(pkg, oracle, scenes):
    a, b = pkg.RenderContext(0), pkg.RenderContext(0)
    a.upload_scene(scenes("teapot"))
    b.upload_scene(scenes("model"))
    cam = pkg.make_camera()
    p = pkg.make_params(160, 90, frames=2)
    ia, ib = a.render(cam, p), b.render(cam, p)
    wa, _ = oracle.render(scenes("teapot"), cam, p)
    wb, _ = oracle.render(scenes("model"), cam, p)
    assert np.array_equal(bits(ia), bits(wa)) and np.array_equal(bits(ib), bits(wb))
    a.close()
    b.close()


@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("size,bands", [((333, 217), 3), ((640, 360), 8), ((64, 20), 5), ((200, 1080 // 4), 2)])
def test_band_partition_union_equals_whole_image(pkg, oracle, gpu_ctx, scenes, size, bands, interleaved):
    """Image-tile partition (benchmark configuration 4): each band rendered by its own call, into the rows
    partition.band_rows() names and nowhere else; the union is bit-identical to the whole image and to the oracle."""
    from opengl_raytracer_amd import partition
    scene = scenes("teapot")
    cam = pkg.make_camera(yaw=0.1)
    W, H = size
    gpu_ctx.upload_scene(scene)
    whole = gpu_ctx.render(cam, pkg.make_params(W, H, frames=3))
    want, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=3))
    assert np.array_equal(bits(whole), bits(want))
    sentinel = np.float32(-7.0)
    union = np.full((H, W, 4), sentinel, dtype=np.float32)
    covered = np.zeros(H, dtype=int)
    for g in range(bands):
        buf = np.full((H, W, 4), sentinel, dtype=np.float32)
        # frame_base 0 ignores the buffer's content inside the band; outside the band it must stay untouched
        out = gpu_ctx.render(cam, pkg.make_params(W, H, frames=3, band=(g, bands), interleaved=interleaved), buf)
        mine = np.zeros(H, dtype=bool)
        for r0, r1 in partition.band_row_ranges(g, bands, H, interleaved):
            mine[r0:r1] = True
        covered += mine
        assert np.all(out[~mine] == sentinel)
        union[mine] = out[mine]
    assert np.all(covered == 1)
    assert np.array_equal(bits(union), bits(whole))
    # SUM mode over bands into ONE buffer
    acc = np.zeros((H, W, 4), dtype=np.float32)
    for g in range(bands):
        acc = gpu_ctx.render(cam, pkg.make_params(W, H, frames=2, flags=pkg.RT_FLAG_SUM, band=(g, bands), interleaved=interleaved), acc)
    want_sum = gpu_ctx.render(cam, pkg.make_params(W, H, frames=2, flags=pkg.RT_FLAG_SUM))
    assert np.array_equal(bits(acc), bits(want_sum))


def test_band_argument_errors(pkg, gpu_ctx, scenes):
    gpu_ctx.upload_scene(scenes("model"))
    cam = pkg.make_camera()
    for bad in (dict(band=(3, 3)), dict(band=(0, 2), flags=pkg.RT_FLAG_MEGAKERNEL), dict(band=(0, 2), rays_per_pixel=2)):
        with pytest.raises(RuntimeError):
            gpu_ctx.render(cam, pkg.make_params(64, 64, **bad))


def test_large_image_uses_smaller_batches(pkg, gpu_ctx, scenes):
    """At 8192x4096 the staged samples of 16 frames would take 8.6 GB; the library renders fewer frames per launch
    (4 GiB of staging at most, shared by two overlapping launches: 3 frames each here).  Rendering 10 frames in one call
    (four launches) must equal ten single-frame dispatches bit for bit.  A taller image is rendered first: it stages
    2 x 2 frames of 786 MB, less than the 2 x 3 x 537 MB the second image needs, so the staging area has to grow
    although the pixel count shrinks."""
    gpu_ctx.upload_scene(scenes("teapot"))
    cam = pkg.make_camera()
    tall = pkg.DeviceBuffer(8192 * 6000 * 16)
    gpu_ctx.render_device(cam, pkg.make_params(8192, 6000, frames=3), tall.ptr)
    gpu_ctx.synchronize()
    tall.free()
    W, H = 8192, 4096
    nbytes = W * H * 16
    a, b = pkg.DeviceBuffer(nbytes), pkg.DeviceBuffer(nbytes)
    gpu_ctx.render_device(cam, pkg.make_params(W, H, frames=10), a.ptr)
    for f in range(10):
        gpu_ctx.raytrace_frame(cam, pkg.make_params(W, H), f, b.ptr, b.ptr)
    gpu_ctx.synchronize()
    img_a = a.to_array((H, W, 4), np.float32)
    img_b = b.to_array((H, W, 4), np.float32)
    a.free()
    b.free()
    assert np.array_equal(bits(img_a), bits(img_b))
    assert img_a[..., 3].min() == 2.0 and img_a[..., :3].max() <= 1.0 and img_a[..., :3].min() >= 0.0


def test_upload_with_integer_links(pkg, oracle, scenes):
    """rt_upload_scene reads the node ids from rt_scene.links when given (float ids poisoned)."""
    scene = scenes("teapot")
    nodes = scene.nodes.copy()
    links = nodes[:, 9:11].astype(np.int32)
    nodes[:, 9:11] = -12345.0
    linked = pkg.Scene(nodes, scene.triv0, scene.triv1, scene.triv2, scene.colours, links=links)
    cam = pkg.make_camera()
    ctx = pkg.RenderContext(0)
    ctx.upload_scene(linked)
    p = pkg.make_params(160, 90, frames=2)
    want, _ = oracle.render(scene, cam, p)
    assert np.array_equal(bits(ctx.render(cam, p)), bits(want))
    # poisoned ids without links: not a tree, refused on the host
    with pytest.raises(RuntimeError):
        ctx.upload_scene(pkg.Scene(nodes, scene.triv0, scene.triv1, scene.triv2, scene.colours))
    ctx.close()


def test_steady_frame_loop_traces_ahead_and_stays_exact(pkg, oracle, gpu_ctx, scenes):
    """A steady frame-by-frame loop (same camera, consecutive frame counts) is served from frames the library traces
    ahead in batches.  Whatever interrupts the pattern -- a synchronise, a camera move, a jump in the frame count, a batched
    call, another image size, a new scene -- the accumulated image equals the oracle's running mean bit for bit."""
    W, H = 320, 180
    scene = scenes("dragon80k")
    gpu_ctx.upload_scene(scene)
    p1 = pkg.make_params(W, H)
    bufs = [pkg.DeviceBuffer(W * H * 16), pkg.DeviceBuffer(W * H * 16)]

    def loop(cam, first, n, params=p1, b=bufs):
        for f in range(first, first + n):
            gpu_ctx.raytrace_frame(cam, params, f, b[(f + 1) & 1].ptr, b[f & 1].ptr)
        return b[(first + n - 1) & 1]

    def image(buf, w=W, h=H):
        gpu_ctx.synchronize()
        return buf.to_array((h, w, 4), np.float32)

    cam = pkg.make_camera(**CAMERAS["dragon80k"])
    # 1. a long steady loop: batches of 2, 4, 8, 8, ... are traced ahead
    want, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=45))
    assert np.array_equal(bits(image(loop(cam, 0, 45))), bits(want))
    # 2. a synchronise in the middle of a batch, then on
    last = loop(cam, 0, 13)
    gpu_ctx.synchronize()
    last = loop(cam, 13, 9)
    want, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=22))
    assert np.array_equal(bits(image(last)), bits(want))
    # 3. the camera moves while frames of the old camera are staged: accumulation restarts at frame 0
    loop(cam, 0, 11)
    cam2 = pkg.make_camera(yaw=0.15, **CAMERAS["dragon80k"])
    want, _ = oracle.render(scene, cam2, pkg.make_params(W, H, frames=7))
    assert np.array_equal(bits(image(loop(cam2, 0, 7))), bits(want))
    # 4. a batched call in the middle (it takes the staging slots), then the loop goes on
    last = loop(cam, 0, 10)
    gpu_ctx.render_device(cam, pkg.make_params(W, H, frames=5, frame_base=10), last.ptr)   # frames 10..14 in place
    other = bufs[0] if last is bufs[1] else bufs[1]
    f = 15
    src, dst = last, other
    for _ in range(9):                                                                      # frames 15..23
        gpu_ctx.raytrace_frame(cam, p1, f, src.ptr, dst.ptr)
        src, dst = dst, src
        f += 1
    want, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=24))
    assert np.array_equal(bits(image(src)), bits(want))
    # 5. two loops of different image sizes interleaved call by call: neither is steady, both stay exact
    w2, h2 = 200, 120
    p2 = pkg.make_params(w2, h2)
    small = [pkg.DeviceBuffer(w2 * h2 * 16), pkg.DeviceBuffer(w2 * h2 * 16)]
    for f in range(9):
        gpu_ctx.raytrace_frame(cam, p1, f, bufs[(f + 1) & 1].ptr, bufs[f & 1].ptr)
        gpu_ctx.raytrace_frame(cam, p2, f, small[(f + 1) & 1].ptr, small[f & 1].ptr)
    want, _ = oracle.render(scene, cam, pkg.make_params(W, H, frames=9))
    want2, _ = oracle.render(scene, cam, pkg.make_params(w2, h2, frames=9))
    assert np.array_equal(bits(image(bufs[0])), bits(want))
    assert np.array_equal(bits(image(small[0], w2, h2)), bits(want2))
    # 6. a new scene while frames of the old one are in flight
    loop(cam, 0, 12)
    teapot = scenes("teapot")
    gpu_ctx.upload_scene(teapot)
    camt = pkg.make_camera(**CAMERAS["teapot"])
    want, _ = oracle.render(teapot, camt, pkg.make_params(W, H, frames=20))
    assert np.array_equal(bits(image(loop(camt, 0, 20))), bits(want))
    # 7. plain sums (RT_FLAG_SUM) and the fixed invDir take the same path
    ps = pkg.make_params(W, H, flags=pkg.RT_FLAG_SUM | pkg.RT_FLAG_FIX_INVDIR)
    zero = np.zeros((H, W, 4), dtype=np.float32)
    sb = [pkg.DeviceBuffer.from_array(zero), pkg.DeviceBuffer.from_array(zero)]
    last = loop(camt, 0, 18, ps, sb)
    want, _ = oracle.render(teapot, camt, pkg.make_params(W, H, frames=18, flags=pkg.RT_FLAG_SUM | pkg.RT_FLAG_FIX_INVDIR))
    got = image(last)
    assert rrmse(got, want) < 1e-6
    # 8. switched off and on again through the API
    gpu_ctx.set_frames_ahead(0)
    want, _ = oracle.render(teapot, camt, pkg.make_params(W, H, frames=9))
    assert np.array_equal(bits(image(loop(camt, 0, 9))), bits(want))
    gpu_ctx.set_frames_ahead(16)
    want, _ = oracle.render(teapot, camt, pkg.make_params(W, H, frames=70))
    assert np.array_equal(bits(image(loop(camt, 0, 70))), bits(want))
    gpu_ctx.set_frames_ahead(8)
    with pytest.raises(RuntimeError):
        gpu_ctx.set_frames_ahead(17)
    for b in bufs + small + sb:
        b.free()


def test_frame_loop_random_operation_sequences(pkg, gpu_ctx, scenes):
    """Seeded random sequences of frame calls (continuing, repeating or jumping in frame count), camera changes, batched
    calls, synchronisations and look-ahead settings, run on two contexts: one traces ahead, the other never does.  Their
    framebuffers must agree bit for bit after every sequence -- the look-ahead may never be observable."""
    W, H = 160, 96
    scene = scenes("dragon80k")
    plain = pkg.RenderContext(0)
    plain.set_frames_ahead(0)
    for ctx in (gpu_ctx, plain):
        ctx.upload_scene(scene)
    gpu_ctx.set_frames_ahead(8)
    p1 = pkg.make_params(W, H)
    cams = [pkg.make_camera(yaw=0.1 * k, **CAMERAS["dragon80k"]) for k in range(3)]
    for seed in range(6):
        rng = np.random.RandomState(100 + seed)
        ops = []
        frame, cam = 0, 0
        for _ in range(120):
            r = rng.uniform()
            if r < 0.78:
                ops.append(("frame", cam, frame))
                frame += 1
            elif r < 0.84:
                cam = int(rng.randint(0, 3))
                frame = 0
            elif r < 0.88:
                frame = max(0, frame + int(rng.randint(-3, 6)))          # a jump: repeated or skipped frame counts
            elif r < 0.92:
                ops.append(("sync",))
            elif r < 0.96:
                n = int(rng.randint(1, 6))
                ops.append(("batch", cam, frame, n))
                frame += n
            else:
                ops.append(("ahead", int(rng.choice([0, 2, 4, 8, 16]))))
        results = []
        for ctx in (gpu_ctx, plain):
            zero = np.zeros((H, W, 4), dtype=np.float32)
            bufs = [pkg.DeviceBuffer.from_array(zero), pkg.DeviceBuffer.from_array(zero)]
            cur = 0                                                        # index of the buffer holding the latest image
            for op in ops:
                if op[0] == "frame":
                    ctx.raytrace_frame(cams[op[1]], p1, op[2], bufs[cur].ptr, bufs[cur ^ 1].ptr)
                    cur ^= 1
                elif op[0] == "sync":
                    ctx.synchronize()
                elif op[0] == "batch":
                    ctx.render_device(cams[op[1]], pkg.make_params(W, H, frames=op[3], frame_base=op[2]), bufs[cur].ptr)
                elif op[0] == "ahead" and ctx is gpu_ctx:
                    ctx.set_frames_ahead(op[1])
            ctx.synchronize()
            results.append(bufs[cur].to_array((H, W, 4), np.float32))
            for b in bufs:
                b.free()
        assert np.array_equal(bits(results[0]), bits(results[1])), f"sequence {seed}"
    gpu_ctx.set_frames_ahead(8)
    plain.close()


def _chain_scene(pkg, n=40):
    """n triangles stacked behind one another in front of the default camera, under a hand-made tree that is one long chain:
    every inner node has the nearer n-1-l triangles as its first child and the farthest remaining one as a leaf, so a ray
    through the middle enters the chain first and leaves one pending entry per level."""
    tri = np.zeros((n, 3, 3), dtype=np.float32)
    for k in range(n):
        s = 0.6 + 0.01 * k
        tri[k] = ((-s, -s, 0.25 * k), (s, -s, 0.25 * k), (0.0, s, 0.25 * k))
    nodes = np.zeros((2 * n - 1, 12), dtype=np.float32)

    def put(i, lo, hi, leaf, a, b):
        v = tri[lo:hi + 1].reshape(-1, 3)
        nodes[i, 0:3], nodes[i, 4:7], nodes[i, 8:11] = v.min(0), v.max(0), (leaf, a, b)

    for lvl in range(n - 1):
        hi = n - 1 - lvl
        put(2 * lvl, 0, hi, 0.0, 2 * (lvl + 1), 2 * lvl + 1)
        put(2 * lvl + 1, hi, hi, 1.0, hi, hi)
    put(2 * (n - 1), 0, 0, 1.0, 0, 0)
    v = np.zeros((3, n, 4), dtype=np.float32)
    v[:, :, :3] = tri.transpose(1, 0, 2)
    return pkg.Scene(nodes, v[0], v[1], v[2], pkg.make_colours(7, n))


def test_packet_that_outgrows_its_stack_hands_the_tile_back(pkg, oracle, gpu_ctx):
    """A tile packet keeps 19 pending entries.  On a 39-level chain whose far child is pushed at every level the packets of the
    middle tiles give up and flag their tile; the path kernel then traces those primary rays lane by lane (its own stack spills
    to memory).  Pixels and counters must not show which way a tile went."""
    scene = _chain_scene(pkg, 40)
    cam = pkg.make_camera(pos=(0.0, 0.0, -0.8))
    gpu_ctx.upload_scene(scene)
    ids, t = gpu_ctx.primary_hits(cam, 200, 120)
    ids_o, t_o = oracle.primary_hits(scene, cam, 200, 120)
    assert np.array_equal(ids, ids_o) and np.array_equal(bits(t), bits(t_o))
    assert (ids_o >= 0).mean() > 0.1 and (ids_o < 0).mean() > 0.1      # some tiles see the stack of triangles, some only sky
    for extra in (0, pkg.RT_FLAG_FIX_INVDIR):
        params = pkg.make_params(200, 120, frames=3, flags=extra)
        want, cnt_o = oracle.render(scene, cam, params)
        render_both_ways(pkg, gpu_ctx, cam, params, want, cnt_o, what=f"chain {extra}")


@pytest.mark.parametrize("env", [
    {"RT_TUNE_PRIMARY_GRID": "1000"},                                # a multiple of 8 but not of 64: eight counters
    {"RT_TUNE_PRIMARY_GRID": "999", "RT_TUNE_PRIMARY_CHUNK": "3"},   # neither: one counter
    {"RT_TUNE_PRIMARY_GRID": "7"},                                   # fewer waves than a frame has tiles
    {"RT_TUNE_PRIMARY_CHUNK": "1"},
    {"RT_TUNE_PRIMARY_CHUNK": "64"},                                 # longer than a short launch has items per wave
    {"RT_TUNE_SIGNED_SLABS": "0"},                                   # every packet through the general slab tests
    {"RT_TUNE_LAST_BLOCKS": "1", "RT_TUNE_QUEUE_BLOCKS": "2"},
    {"RT_TUNE_PRIMARY_JOINED": "3", "RT_TUNE_PRIMARY_BLOCKS": "7"},   # other grids for the launch alone on the chip / next to a path kernel
])
def test_packet_kernel_work_distribution_variants(pkg, oracle, scenes, env):
    """k_primary's waves pull chunks of work items from counters in memory; how many counters, how long the chunks and how
    large the grid is must not show in the image (every item traced exactly once).  Forced through the tuning build; also
    the general form of the slab tests where the default picks the sign-specific ones, and the other workgroup counts of the
    path kernel behind it."""
    import os
    os.environ.update(env)
    try:
        ctx = pkg.RenderContext(0, tunable=True)
    finally:
        for k in env:
            del os.environ[k]
    for name, frames in (("dragon80k", 18), ("teapot", 3)):
        scene = scenes(name)
        ctx.upload_scene(scene)
        cam = pkg.make_camera(yaw=0.3, **CAMERAS[name])
        p = pkg.make_params(333, 217, frames=frames, flags=pkg.RT_FLAG_PACKETS)
        want, _ = oracle.render(scene, cam, p)
        got = ctx.render(cam, p)
        assert np.array_equal(bits(got), bits(want)), f"{name} {env}: {int((bits(got) != bits(want)).any(axis=-1).sum())} pixels differ"
        p1 = pkg.make_params(333, 217, frames=1, frame_base=frames, flags=pkg.RT_FLAG_PACKETS)
        want1, _ = oracle.render(scene, cam, p1, want.copy())
        got1 = ctx.render(cam, p1, got.copy())
        assert np.array_equal(bits(got1), bits(want1))
    ctx.close()
