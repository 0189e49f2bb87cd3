#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

The reference has no tests, fixtures or golden vectors of its own and its render path (GLSL) cannot run
here, so two kinds of vectors exist:

1. ``ingest_*.json`` -- produced by the REFERENCE's own host objects (obj-reader.cpp, bvh.cpp,
   scene-loader.cpp compiled from /root/reference by oracle/Makefile into oracle/_ref/ref_ingest):
   FNV-1a hashes and sizes of the node / vertex buffers loadScene() yields for each fixture OBJ, the
   colours libstdc++'s uniform_real_distribution draws for seed 1, and GLM's camera matrices.  These pin
   the host ingest (librt_host.so) to the reference.  Needs the reference checkout; run in the build
   container only.
2. ``oracle_*.npz`` -- outputs of this repository's CPU oracle (small images, primary-hit ids).  They do
   not pin the oracle to the reference (nothing can, see oracle/rt_oracle.cpp); they pin the oracle and the
   math contract against silent drift between machines, compilers and edits.

Usage: python tests/golden/make_golden.py [--skip-ref]
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

FIX = os.path.join(ROOT, "fixtures")


def fnv(a) -> str:
    h = 1469598103934665603
    data = np.ascontiguousarray(a).view(np.uint8).ravel()
    # vectorised FNV-1a is awkward; the buffers are small enough for the C helper
    pkg = entry.load_package()
    return format(pkg.load_host().rt_host_hash64(data.ctypes.data, data.size), "016x")


def read_ref_dump(path):
    b = open(path, "rb").read()
    T, N = np.frombuffer(b[:8], dtype=np.int32)
    off = 16
    out = {}
    for name, cnt, w in (("nodes", N, 12), ("triv0", T, 4), ("triv1", T, 4), ("triv2", T, 4), ("colours", T, 4)):
        out[name] = np.frombuffer(b[off:off + cnt * w * 4], dtype=np.float32).reshape(cnt, w)
        off += cnt * w * 4
    out["rot"] = np.frombuffer(b[off:off + 36], dtype=np.float32)
    return out


def ref_goldens():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_ingest")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_ingest missing: make -C oracle ref (needs /root/reference)")
    cams = [(0.0, 0.0), (0.3, -0.7), (-1.2, 2.5), (1.5707963267948966, 0.1)]
    for name in ("teapot", "model", "dragon80k"):
        rec = {"source": "reference objects via oracle/_ref/ref_ingest", "colour_seed": 1}
        for k, (pitch, yaw) in enumerate(cams if name == "teapot" else cams[:1]):
            tmp = os.path.join(ROOT, "oracle", "_ref", f"{name}.bin")
            subprocess.check_call([exe, os.path.join(FIX, name + ".obj"), tmp, "1", repr(pitch), repr(yaw)],
                                  stdout=subprocess.DEVNULL)
            d = read_ref_dump(tmp)
            os.remove(tmp)
            if k == 0:
                rec.update(num_tris=int(d["triv0"].shape[0]), num_nodes=int(d["nodes"].shape[0]))
                for key in ("nodes", "triv0", "triv1", "triv2", "colours"):
                    rec["hash_" + key] = fnv(d[key])
                rec["first_colours"] = d["colours"][:2].tolist()
                rec["cameras"] = []
            rec["cameras"].append({"pitch": pitch, "yaw": yaw, "rot_bits": [int(x) for x in d["rot"].view(np.uint32)]})
        with open(os.path.join(HERE, f"ingest_{name}.json"), "w") as f:
            json.dump(rec, f, indent=1)
        print("wrote", f"ingest_{name}.json", rec["num_tris"], rec["num_nodes"])


def oracle_goldens():
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    cases = {
        "teapot": dict(pos=(0.0, 0.0, -2.0), size=(160, 90)),
        "dragon80k": dict(pos=(0.0, 0.0, -5.5), size=(160, 90)),
        "model": dict(pos=(0.0, 0.0, -2.0), size=(160, 90)),
    }
    for name, c in cases.items():
        scene = pkg.load_scene(os.path.join(FIX, name + ".obj"), colour_seed=1)
        cam = pkg.make_camera(pos=c["pos"])
        w, h = c["size"]
        ids, t = oracle.primary_hits(scene, cam, w, h)
        img, cnt = oracle.render(scene, cam, pkg.make_params(w, h, frames=3))
        img_fix, cnt_fix = oracle.render(scene, cam, pkg.make_params(w, h, frames=2, flags=pkg.RT_FLAG_FIX_INVDIR))
        np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), ids=ids, t=t, img=img, counters=cnt,
                            img_fix=img_fix, counters_fix=cnt_fix, size=np.array([w, h]), pos=np.array(c["pos"]))
        print("wrote", f"oracle_{name}.npz", cnt)


if __name__ == "__main__":
    if "--skip-ref" not in sys.argv:
        ref_goldens()
    oracle_goldens()
