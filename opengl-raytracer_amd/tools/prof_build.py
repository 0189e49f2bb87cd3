#!/usr/bin/env python3
"""Builds the ~9.3M-triangle grid scene (or a fixture OBJ) with the GPU BVH builder, for profilers and timing:
prof_build.py [--scene grid|dragon80k|teapot] [--repeat N] [--render FRAMES]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as entry  # noqa: E402
import gen_scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="grid")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--render", type=int, default=0)
    ap.add_argument("--rebuild", type=int, default=0, help="rebuilds from perturbed device-resident vertices")
    ap.add_argument("--gen", default=None, help="builder generation (sets RT_BUILD_GEN)")
    a = ap.parse_args()
    if a.gen:
        os.environ["RT_BUILD_GEN"] = a.gen
    pkg = entry.load_package()
    fx = os.path.join(ROOT, "fixtures")
    if a.scene == "grid":
        v, f = gen_scenes.grid([os.path.join(fx, "dragon80k.obj")] * 3 + [os.path.join(fx, "teapot.obj")], (6, 5, 5), 8.0, 7)
        pos = (0.0, 0.0, -30.0)
    else:
        v, f = gen_scenes.read_obj(os.path.join(fx, a.scene + ".obj"))
        pos = (0.0, 0.0, -5.5) if a.scene == "dragon80k" else (0.0, 0.0, -2.0)
    v, f = v.astype(np.float32), f.astype(np.uint32)
    col = pkg.make_colours(1, f.shape[0])
    ctx = pkg.RenderContext(0, tunable=any(k.startswith(("RT_TUNE_", "RT_BUILD_")) for k in os.environ))
    for r in range(a.repeat):
        t0 = time.perf_counter()
        rc = ctx.lib.rt_build_scene_device(ctx.ctx, pkg.fptr(v), v.shape[0], f.ctypes.data_as(pkg.c_uint32_p), f.shape[0], pkg.fptr(col))
        assert rc == 0, ctx.lib.rt_last_error(ctx.ctx)
        wall = time.perf_counter() - t0
        n, t, ms = C.c_int32(), C.c_int32(), C.c_float()
        ctx.lib.rt_get_built_scene(ctx.ctx, C.byref(n), C.byref(t), None, None, None, None, None, None, C.byref(ms))
        print(f"build {r}: {t.value} triangles -> {n.value} nodes: device {ms.value:.2f} ms, call {wall * 1e3:.1f} ms (incl. validation + upload)")
    if a.rebuild:
        for r in range(a.rebuild):
            d_w = pkg.DeviceBuffer.from_array((v + 0.01 * np.sin(v * 3.0 + float(r))).astype(np.float32))
            t0 = time.perf_counter()
            _, ms = ctx.rebuild_scene_device(d_w.ptr)
            print(f"rebuild {r}: device {ms:.2f} ms, call {(time.perf_counter() - t0) * 1e3:.2f} ms")
            d_w.free()
    if a.render:
        ctx.num_tris = f.shape[0]
        cam = pkg.make_camera(pos=pos)
        img = np.zeros((1080, 1920, 4), dtype=np.float32)
        ctx.render(cam, pkg.make_params(1920, 1080, frames=4), img)
        t0 = time.perf_counter()
        ctx.render(cam, pkg.make_params(1920, 1080, frames=a.render), img)
        print(f"render: {(time.perf_counter() - t0) / a.render * 1e3:.3f} ms per frame over {a.render} frames")
    ctx.close()


if __name__ == "__main__":
    main()
