"""rt_raytrace_frame_noalloc: the frame entry point for callers that cannot allocate device memory.

It shares its kernels with rt_raytrace_frame but keeps the staged samples and the traversal-stack spill area in a static
device array instead of buffers the context allocates.  The test drives both entry points through the C ABI on the same
inputs and requires bit-identical frames -- on two calls in a row, the second with a smaller image, so that whatever the
first call left in the shared array must not leak into the second -- then checks the documented limits.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _frames(pkg, ctx, fn, cam, W, H, nframes, flags=0):
    """nframes consecutive dispatches through ping-pong device buffers with `fn`; returns the last frame."""
    zeros = np.zeros((H, W, 4), dtype=np.float32)          # RT_FLAG_SUM adds to the previous frame from frame 0 on
    bufs = [pkg.DeviceBuffer.from_array(zeros), pkg.DeviceBuffer.from_array(zeros)]
    p = pkg.make_params(W, H, flags=flags)
    for f in range(nframes):
        fn(cam, p, f, bufs[(f + 1) & 1].ptr, bufs[f & 1].ptr)
    ctx.synchronize()
    out = bufs[(nframes - 1) & 1].to_array((H, W, 4), np.float32)
    for b in bufs:
        b.free()
    return out


def test_noalloc_frames_equal_allocating_entry_point(pkg, oracle, scenes):
    ctx = pkg.RenderContext(0)
    cam = pkg.make_camera(pos=(0.0, 0.0, -5.5), yaw=0.05)
    # dragon80k is 22 levels deep: deeper than the 8 stack entries kept in LDS, so the spill part of the array is in use
    ctx.upload_scene(scenes("dragon80k"))
    for (W, H, n) in ((1920, 1080, 3), (333, 217, 4)):      # second call: smaller input, same context, same array
        got = _frames(pkg, ctx, ctx.raytrace_frame_noalloc, cam, W, H, n)
        want = _frames(pkg, ctx, ctx.raytrace_frame, cam, W, H, n)
        assert np.array_equal(bits(got), bits(want)), f"{W}x{H}: {int((bits(got) != bits(want)).any(axis=-1).sum())} pixels differ"
        assert np.all(got[..., 3] == 2.0)
    # and against the oracle at the small size (frames 0..3, running mean)
    ref, _ = oracle.render(scenes("dragon80k"), cam, pkg.make_params(333, 217, frames=4))
    assert np.array_equal(bits(got), bits(ref))
    # sum mode + fix_invdir + counters through the same path
    fl = pkg.RT_FLAG_SUM | pkg.RT_FLAG_FIX_INVDIR | pkg.RT_FLAG_COUNTERS
    ctx.reset_counters()
    a = _frames(pkg, ctx, ctx.raytrace_frame_noalloc, cam, 200, 120, 2, fl)
    ca = ctx.counters()
    ctx.reset_counters()
    b = _frames(pkg, ctx, ctx.raytrace_frame, cam, 200, 120, 2, fl)
    cb = ctx.counters()
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(ca, cb)
    ctx.close()


def test_noalloc_limits(pkg, scenes):
    ctx = pkg.RenderContext(0)
    ctx.upload_scene(scenes("teapot"))
    cam = pkg.make_camera()
    buf = pkg.DeviceBuffer(64 * 64 * 16)
    lib = ctx.lib
    # beyond the static array: 4096 x 4096 pixels stage 268 MB
    rc = lib.rt_raytrace_frame_noalloc(ctx.ctx, C.byref(cam), C.byref(pkg.make_params(4096, 4096)), 0, None, C.c_void_p(buf.ptr))
    assert rc == -4 and b"scratch" in lib.rt_last_error(ctx.ctx)
    # other pipelines are refused, not silently rerouted
    for bad in (dict(flags=pkg.RT_FLAG_MEGAKERNEL), dict(flags=pkg.RT_FLAG_WAVEFRONT), dict(render_mode=2), dict(rays_per_pixel=2),
                dict(ray_bounces=0)):
        rc = lib.rt_raytrace_frame_noalloc(ctx.ctx, C.byref(cam), C.byref(pkg.make_params(64, 64, **bad)), 0, None, C.c_void_p(buf.ptr))
        assert rc == -1
    rc = lib.rt_raytrace_frame_noalloc(ctx.ctx, C.byref(cam), C.byref(pkg.make_params(64, 64)), 3, None, C.c_void_p(buf.ptr))
    assert rc == -1     # frame 3 needs the previous frame
    # still usable
    ctx.raytrace_frame_noalloc(cam, pkg.make_params(64, 64), 0, None, buf.ptr)
    ctx.synchronize()
    buf.free()
    ctx.close()
