#!/usr/bin/env python3
"""Duration of one path-kernel launch against the number of frames it renders (fixed cost per launch vs cost per frame).
usage: launch_scaling.py [scene]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "dragon80k"
    pkg = entry.load_package()
    pos = {"dragon80k": (0.0, 0.0, -4.5), "teapot": (0.0, 0.0, -2.0)}.get(scene, (0.0, 0.0, -2.0))
    cam = pkg.make_camera(pos=pos)
    ctx = pkg.RenderContext(0, tunable=any(k.startswith(("RT_TUNE_", "RT_BUILD_")) for k in os.environ))
    ctx.upload_scene(pkg.load_scene(os.path.join(ROOT, "fixtures", scene + ".obj"), colour_seed=1))
    W, H = 1920, 1080
    buf = pkg.DeviceBuffer(W * H * 16)
    ctx.render_device(cam, pkg.make_params(W, H, frames=32), buf.ptr)
    ctx.synchronize()
    ctx.set_profiling(True)
    names = pkg.KERNEL_CLASS_NAMES
    for n in (1, 2, 4, 8, 16, 32):
        best = 1e9
        for _ in range(5):
            ctx.render_device(cam, pkg.make_params(W, H, frames=n), buf.ptr)
            ms, launches = ctx.kernel_times()
            best = min(best, float(ms[names.index("paths")]))
        print(f"{n:3d} frames: {best:8.3f} ms per launch, {best / n:6.3f} ms per frame")
    # the same for small images, one frame: what a launch costs before there is any work to speak of
    for w, h in ((64, 64), (256, 256), (640, 360), (1280, 720)):
        best = 1e9
        for _ in range(5):
            ctx.render_device(cam, pkg.make_params(w, h, frames=1), buf.ptr)
            ms, launches = ctx.kernel_times()
            best = min(best, float(ms[names.index("paths")]))
        print(f"{w}x{h}, 1 frame: {best:8.3f} ms per launch")
    buf.free()
    ctx.close()


if __name__ == "__main__":
    main()
