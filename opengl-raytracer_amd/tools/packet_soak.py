#!/usr/bin/env python3
"""Random-camera soak of the tile-packet kernel: renders many cameras with RT_FLAG_PACKETS forced (k_primary + the queue
path kernel, the uncounted instantiation a timed run executes) and compares every pixel, bit for bit, with the render of the
same context that traces everything lane by lane with the reference's literal traversal (RT_FLAG_NO_CULL).  Cameras stand
anywhere in and around the model, so packets see every sign octant, rays that start inside boxes, tiles that are all sky
and tiles at the image's centre lines.  usage (GPU box, repo root): packet_soak.py [scene] [cameras] [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dragon80k"
cameras = int(sys.argv[2]) if len(sys.argv) > 2 else 100
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 18   # two launches: 16 frames and a remainder that joins them
pkg = entry.load_package()
scene = pkg.load_scene(os.path.join(ROOT, "fixtures", name + ".obj"), colour_seed=1)
ctx = pkg.RenderContext(0)
ctx.upload_scene(scene)
rng = np.random.RandomState(11)
bad_pixels = bad_cameras = 0
for k in range(cameras):
    W, H = [(640, 360), (333, 217), (960, 540)][k % 3]
    pos = tuple(float(x) for x in rng.uniform(-3.0, 3.0, 3))
    cam = pkg.make_camera(pos=pos, pitch=float(rng.uniform(-1.2, 1.2)), yaw=float(rng.uniform(-3.1, 3.1)))
    got = ctx.render(cam, pkg.make_params(W, H, frames=frames, flags=pkg.RT_FLAG_PACKETS))
    want = ctx.render(cam, pkg.make_params(W, H, frames=frames, flags=pkg.RT_FLAG_NO_CULL))
    d = (got.view(np.uint32) != want.view(np.uint32)).any(axis=-1)
    bad_pixels += int(d.sum())
    bad_cameras += int(d.any())
ctx.close()
print(f"{name}: {bad_pixels} differing pixels in {bad_cameras} of {cameras} cameras ({frames} frames each, 640x360 / 333x217 / 960x540 in turn), "
      f"build {pkg.build_id()}", flush=True)
