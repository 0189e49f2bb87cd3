"""The path kernel's subtree cull (on unless RT_FLAG_NO_CULL): a bounce ray's slab tests follow the primary ray's line
(the stale invDir of raytrace.glsl:247-254), so the reference walks through subtrees the ray itself never comes near;
the kernel drops a child whose widened box the ray's true line misses.  That must never change a pixel, and it must
switch itself off for scenes whose pad it cannot bound: trees whose boxes do not nest, NaN vertices, large triangles."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import FIXTURES, bits, render_both_ways

pytestmark = pytest.mark.gpu


def _read_obj(pkg, name):
    lib = pkg.load_host()
    vp, fp = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, nf = C.c_int32(), C.c_int32()
    assert lib.rt_host_read_obj(os.path.join(FIXTURES, name + ".obj").encode(), C.byref(vp), C.byref(nv), C.byref(fp), C.byref(nf)) == 0
    v = np.ctypeslib.as_array(vp, shape=(nv.value, 3)).copy()
    f = np.ctypeslib.as_array(fp, shape=(nf.value, 3)).copy()
    lib.rt_host_free(vp)
    lib.rt_host_free(fp)
    return v, f


def _counted(pkg, ctx, cam, w, h, frames, flags=0, **kw):
    ctx.reset_counters()
    img = ctx.render(cam, pkg.make_params(w, h, frames=frames, flags=flags | pkg.RT_FLAG_COUNTERS, **kw))
    return img, ctx.counters()


@pytest.mark.parametrize("name,pos,pitch,yaw", [
    ("dragon80k", (0.0, 0.0, -5.5), 0.0, 0.0),
    ("dragon80k", (0.0, 0.3, -1.2), 0.2, 0.3),       # close: most of the image bounces
    ("dragon80k", (0.05, 0.1, 0.02), 0.0, 1.0),      # inside the model's box
    ("dragon80k", (0.0, 6.0, 0.0), 1.5, 0.0),        # from above, grazing the axis-parallel case
    ("teapot", (0.0, 0.0, -2.0), 0.0, 0.0),
    ("teapot", (0.0, 0.2, -0.6), 0.3, -0.4),
])
def test_cull_never_changes_a_pixel(pkg, oracle, gpu_ctx, scenes, name, pos, pitch, yaw):
    scene = scenes(name)
    gpu_ctx.upload_scene(scene)
    cam = pkg.make_camera(pos=pos, pitch=pitch, yaw=yaw)
    p = pkg.make_params(480, 270, frames=3)
    want, cnt_o = oracle.render(scene, cam, p)
    _, cnt = render_both_ways(pkg, gpu_ctx, cam, p, want, cnt_o, what=f"{name} {pos}")
    if cnt_o[3] > 1000:      # bounce rays exist: the cull has removed work
        assert cnt[0] < cnt_o[0] and cnt[4] < cnt_o[4]


@pytest.mark.parametrize("scale", [0.1, 10.0])
def test_cull_pad_follows_the_scene_scale(pkg, oracle, gpu_ctx, scale):
    """The pad grows with the scene (edge products, diagonal): the same model ten times smaller and larger."""
    v, f = _read_obj(pkg, "dragon80k")
    scene = pkg.build_scene((v * scale).astype(np.float32), f, colour_seed=3)
    gpu_ctx.upload_scene(scene)
    cam = pkg.make_camera(pos=(0.0, 0.0, -5.5 * scale), yaw=0.1)
    p = pkg.make_params(400, 240, frames=2)
    want, cnt_o = oracle.render(scene, cam, p)
    render_both_ways(pkg, gpu_ctx, cam, p, want, cnt_o, what=f"scale {scale}")


def test_cull_is_off_for_large_triangles_and_unnested_trees(pkg, oracle, gpu_ctx, scenes):
    """Where the pad cannot be bounded the default render does exactly the reference's work: a scene with large triangles
    (models/model.obj has 20-unit quads), and a tree with a box that does not contain what is below it (the reference
    follows such a tree as it is, and so must the kernel)."""
    model = scenes("model")
    gpu_ctx.upload_scene(model)
    cam = pkg.make_camera(pos=(0.0, 0.0, -2.0), pitch=0.4)
    want, cnt_o = oracle.render(model, cam, pkg.make_params(320, 180, frames=2))
    got, cnt = _counted(pkg, gpu_ctx, cam, 320, 180, 2)
    assert np.array_equal(bits(got), bits(want)) and np.array_equal(cnt, cnt_o)

    teapot = scenes("teapot")
    nodes = teapot.nodes.copy()
    inner = np.nonzero(np.abs(nodes[:, 8] - 1.0) > 1e-4)[0]
    victim = int(inner[len(inner) // 3])             # some inner node below the root: shrink its box to a quarter
    assert victim != 0
    centre = 0.5 * (nodes[victim, 0:3] + nodes[victim, 4:7])
    nodes[victim, 0:3] = centre + 0.25 * (nodes[victim, 0:3] - centre)
    nodes[victim, 4:7] = centre + 0.25 * (nodes[victim, 4:7] - centre)
    broken = pkg.Scene(nodes, teapot.triv0, teapot.triv1, teapot.triv2, teapot.colours, links=teapot.links)
    gpu_ctx.upload_scene(broken)
    cam = pkg.make_camera(pos=(0.0, 0.0, -2.0))
    want, cnt_o = oracle.render(broken, cam, pkg.make_params(320, 180, frames=2))
    got, cnt = _counted(pkg, gpu_ctx, cam, 320, 180, 2)
    assert np.array_equal(bits(got), bits(want)) and np.array_equal(cnt, cnt_o)
    # the intact tree right after it on the same context: culled again
    gpu_ctx.upload_scene(teapot)
    want, cnt_o = oracle.render(teapot, cam, pkg.make_params(320, 180, frames=2))
    got, cnt = _counted(pkg, gpu_ctx, cam, 320, 180, 2)
    assert np.array_equal(bits(got), bits(want)) and cnt[0] < cnt_o[0]


def test_cull_on_device_built_and_refitted_trees(pkg, oracle):
    """The device builder and the refit compute the pad on the device (edge maxima of the triangle emit kernel)."""
    v, f = _read_obj(pkg, "dragon80k")
    col = pkg.make_colours(4, f.shape[0])
    ctx = pkg.RenderContext(0)
    built, _ = ctx.build_scene_device(v, f, col)
    cam = pkg.make_camera(pos=(0.0, 0.0, -5.5), yaw=-0.2)
    p = pkg.make_params(400, 240, frames=2)
    want, cnt_o = oracle.render(built, cam, p)
    _, cnt = render_both_ways(pkg, ctx, cam, p, want, cnt_o, what="built")
    assert cnt[0] < 0.8 * cnt_o[0]
    v2 = (v + np.random.RandomState(2).normal(0.0, 0.004, v.shape)).astype(np.float32)
    d_v = pkg.DeviceBuffer.from_array(v2)
    refit, _ = ctx.refit_scene_device(d_v.ptr, download=True)
    d_v.free()
    want, cnt_o = oracle.render(refit, cam, p)
    _, cnt = render_both_ways(pkg, ctx, cam, p, want, cnt_o, what="refit")
    assert cnt[0] < 0.8 * cnt_o[0]
    # a NaN vertex switches the cull off (and the render still equals the oracle's, NaN for NaN)
    v3 = v2.copy()
    v3[int(f[100, 0])] = np.nan
    d_v = pkg.DeviceBuffer.from_array(v3)
    bad, _ = ctx.refit_scene_device(d_v.ptr, download=True)
    d_v.free()
    want, cnt_o = oracle.render(bad, cam, p)
    ctx.reset_counters()
    got = ctx.render(cam, pkg.make_params(400, 240, frames=2, flags=pkg.RT_FLAG_COUNTERS))
    same = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
    assert same.all() and np.array_equal(ctx.counters(), cnt_o)
    ctx.close()


@pytest.mark.parametrize("reach", ["0", "1"])
def test_cull_with_either_pad_rule(pkg, oracle, scenes, reach):
    """The pad is either one constant per scene or grows with a box's reach from the ray origin; the library picks by the
    scene's size against its triangles.  Both forced here (the tuning build of the library reads RT_TUNE_CULL_REACH when a
    context is created; the shipped one reads no environment)."""
    os.environ["RT_TUNE_CULL_REACH"] = reach
    try:
        ctx = pkg.RenderContext(0, tunable=True)
    finally:
        del os.environ["RT_TUNE_CULL_REACH"]
    for name, pos, yaw in (("dragon80k", (0.0, 0.0, -5.5), 0.0), ("dragon80k", (0.2, 0.4, -1.0), 0.5), ("teapot", (0.0, 0.0, -2.0), -0.2)):
        scene = scenes(name)
        ctx.upload_scene(scene)
        cam = pkg.make_camera(pos=pos, yaw=yaw)
        p = pkg.make_params(400, 240, frames=2)
        want, cnt_o = oracle.render(scene, cam, p)
        _, cnt = render_both_ways(pkg, ctx, cam, p, want, cnt_o, what=f"{name} reach={reach}")
        assert cnt[0] < cnt_o[0]
    ctx.close()


@pytest.mark.parametrize("name,extent", [("dragon80k", 3.0), ("teapot", 2.5)])
def test_cull_random_cameras(pkg, oracle, gpu_ctx, scenes, name, extent):
    """Twelve random cameras around and inside each model, four frames each: every image bit-identical to the oracle's with
    and without the cull."""
    scene = scenes(name)
    gpu_ctx.upload_scene(scene)
    rng = np.random.RandomState(17)
    for k in range(12):
        pos = tuple(float(x) for x in rng.uniform(-extent, extent, 3))
        cam = pkg.make_camera(pos=pos, pitch=float(rng.uniform(-1.2, 1.2)), yaw=float(rng.uniform(-3.1, 3.1)))
        p = pkg.make_params(240, 136, frames=4)
        want, cnt_o = oracle.render(scene, cam, p)
        render_both_ways(pkg, gpu_ctx, cam, p, want, cnt_o, what=f"{name} camera {k} at {pos}")
