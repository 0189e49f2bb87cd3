#!/usr/bin/env python3
"""Minimal render loop for profilers (rocprofv3 wraps `python3 prof_render.py ...` directly):
no torch, no CPU baseline -- scene upload, a few warm-up frames, N frames through the C ABI."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SCENES = {"dragon80k": ("dragon80k.obj", (0.0, 0.0, -5.5)), "teapot": ("teapot.obj", (0.0, 0.0, -2.0)),
          "grid": (None, (0.0, 0.0, -30.0))}  # grid: ~9.3M triangles, tree built on the GPU


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="dragon80k", choices=sorted(SCENES))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pipeline", default="paths", choices=["paths", "wavefront", "megakernel"])
    ap.add_argument("--fix-invdir", action="store_true")
    ap.add_argument("--rays-per-pixel", type=int, default=1)
    ap.add_argument("--primary-only", action="store_true", help="time rt_primary_hits instead of full frames")
    ap.add_argument("--counters", action="store_true", help="render with RT_FLAG_COUNTERS and print the work counters")
    ap.add_argument("--per-frame", action="store_true", help="one rt_raytrace_frame call per frame (ping-pong buffers), as the reference's loop does")
    ap.add_argument("--bands", type=int, default=0, help="time every band of an N-way tile partition on this one GPU")
    ap.add_argument("--standalone", type=int, default=0,
                    help="after the warm-up, this many calls of --frames frames each (at most one launch's worth), every one "
                         "synchronised before the next: launches that have the chip to themselves, for a kernel trace")
    a = ap.parse_args()
    pkg = entry.load_package()
    obj, pos = SCENES[a.scene]
    cam = pkg.make_camera(pos=pos)
    ctx = pkg.RenderContext(0, tunable=any(k.startswith(("RT_TUNE_", "RT_BUILD_")) for k in os.environ))
    if obj is None:
        import numpy as np

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import gen_scenes

        fx = os.path.join(ROOT, "fixtures")
        v, f = gen_scenes.grid([os.path.join(fx, "dragon80k.obj")] * 3 + [os.path.join(fx, "teapot.obj")], (6, 5, 5), 8.0, 7)
        ctx.build_scene_device(v.astype(np.float32), f.astype(np.uint32), pkg.make_colours(1, f.shape[0]), download=False)
    else:
        ctx.upload_scene(pkg.load_scene(os.path.join(ROOT, "fixtures", obj), colour_seed=1))
    flags = {"paths": 0, "wavefront": pkg.RT_FLAG_WAVEFRONT, "megakernel": pkg.RT_FLAG_MEGAKERNEL}[a.pipeline] | (pkg.RT_FLAG_FIX_INVDIR if a.fix_invdir else 0)
    if a.primary_only:
        t0 = time.perf_counter()
        for _ in range(a.frames):
            ctx.primary_hits(cam, a.width, a.height)
        print(f"primary_hits: {(time.perf_counter() - t0) / a.frames * 1e3:.3f} ms per call (incl. download)")
        return
    import numpy as np
    if a.counters:
        flags |= pkg.RT_FLAG_COUNTERS
    img = np.zeros((a.height, a.width, 4), dtype=np.float32)
    if a.bands:
        # what an N-GPU tile partition would see: the slowest band sets the frame time
        for inter in (False, True):
            ms = []
            for g in range(a.bands):
                p = pkg.make_params(a.width, a.height, frames=a.frames, flags=flags, band=(g, a.bands), interleaved=inter)
                best = 1e9
                for _ in range(4):  # first call warms up, the best of the others counts
                    t0 = time.perf_counter()
                    ctx.render(cam, p, img)
                    best = min(best, (time.perf_counter() - t0) / a.frames * 1e3)
                ms.append(best)
            print(f"{a.bands} {'interleaved' if inter else 'contiguous'} bands, ms per frame per band: "
                  + " ".join(f"{x:.3f}" for x in ms) + f" | max {max(ms):.3f} mean {sum(ms) / len(ms):.3f} balance {sum(ms) / len(ms) / max(ms):.2f}")
        ctx.close()
        return
    if a.standalone:
        buf = pkg.DeviceBuffer(a.width * a.height * 16)
        p = pkg.make_params(a.width, a.height, frames=a.frames, flags=flags, rays_per_pixel=a.rays_per_pixel)
        ctx.render_device(cam, pkg.make_params(a.width, a.height, frames=a.warmup, flags=flags, rays_per_pixel=a.rays_per_pixel), buf.ptr)
        ctx.synchronize()
        ms = []
        for _ in range(a.standalone):
            t0 = time.perf_counter()
            ctx.render_device(cam, p, buf.ptr)
            ctx.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        print(f"{a.pipeline} {a.scene} {a.width}x{a.height}, {a.rays_per_pixel} spp: {a.standalone} synchronised calls of {a.frames} frames: "
              f"mean {sum(ms) / len(ms):.3f} min {min(ms):.3f} max {max(ms):.3f} ms per call (host clock)")
        buf.free()
        ctx.close()
        return
    if a.per_frame:
        nbytes = a.width * a.height * 16
        bufs = [pkg.DeviceBuffer(nbytes), pkg.DeviceBuffer(nbytes)]
        p1 = pkg.make_params(a.width, a.height, flags=flags)
        for phase, n in (("warm-up", a.warmup), ("timed", a.frames)):
            ctx.synchronize()
            t0 = time.perf_counter()
            for f in range(n):
                ctx.raytrace_frame(cam, p1, f, bufs[(f + 1) & 1].ptr, bufs[f & 1].ptr)
            ctx.synchronize()
            dt = time.perf_counter() - t0
        print(f"{a.pipeline} {a.scene} {a.width}x{a.height}: {dt / a.frames * 1e3:.3f} ms per frame over {a.frames} single-frame calls")
        for b in bufs:
            b.free()
        ctx.close()
        return
    ctx.render(cam, pkg.make_params(a.width, a.height, frames=a.warmup, flags=flags), img)
    t0 = time.perf_counter()
    ctx.render(cam, pkg.make_params(a.width, a.height, frames=a.frames, flags=flags), img)
    dt = time.perf_counter() - t0
    print(f"{a.pipeline} {a.scene} {a.width}x{a.height}: {dt / a.frames * 1e3:.3f} ms per frame over {a.frames} frames (incl. one download)")
    if a.counters:
        print("counters [box tri rays hits pops paths]:", ctx.counters().tolist())
    ctx.close()


if __name__ == "__main__":
    main()
