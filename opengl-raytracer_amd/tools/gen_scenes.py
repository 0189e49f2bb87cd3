#!/usr/bin/env python3
"""Deterministic procedural scenes, written as OBJ in the grammar the reference reads
(``v x y z`` / ``f a b c``, obj-reader.cpp:29-36).

``dragon80k``  -- the 80,000-triangle stand-in for the "80K triangle dragon" the reference's README
quotes its frame rate on.  That mesh is not part of the reference repository (models/model.obj is a
7-triangle debug scene), so this generator produces a closed, bumpy (2,3) torus-knot tube of exactly
400 x 100 quads = 80,000 triangles, about 6.7 units across like the teapot, with 2*area of a typical
triangle ~3e-3 >> the path's absolute EPS = 1e-4 (raytrace.glsl:71,90).

``grid``  -- N copies of a base OBJ on a jittered 3-D grid (baked instancing; the reference has no
instancing), for the multi-million-triangle configuration.

The committed fixture fixtures/dragon80k.obj is the output of ``gen_scenes.py dragon80k``; the text
file, not this script, is the input of record (libm differences between machines cannot change it).
"""
from __future__ import annotations

import argparse
import sys

import numpy as np


def dragon80k(n_along: int = 400, n_around: int = 100):
    R, r, tube = 2.0, 0.8, 0.45
    p, q = 2, 3
    u = np.arange(n_along, dtype=np.float64) * (2.0 * np.pi / n_along)
    v = np.arange(n_around, dtype=np.float64) * (2.0 * np.pi / n_around)
    cu, su = np.cos(p * u), np.sin(p * u)
    cq, sq = np.cos(q * u), np.sin(q * u)
    rad = R + r * cq
    centre = np.stack([rad * cu, rad * su, r * sq], axis=1)
    # tangent (analytic derivative), torus-surface normal, binormal: smooth and periodic
    d_rad = -r * q * sq
    tangent = np.stack([d_rad * cu - rad * p * su, d_rad * su + rad * p * cu, r * q * cq], axis=1)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    n1 = np.stack([cq * cu, cq * su, sq], axis=1)
    n1 -= tangent * np.sum(n1 * tangent, axis=1, keepdims=True)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tangent, n1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    # scales / ridges / a slow swell: the "dragon skin"
    radius = tube * (1.0 + 0.18 * np.sin(9.0 * uu) * np.sin(5.0 * vv) + 0.10 * np.sin(31.0 * uu + 3.0 * vv)
                     + 0.22 * np.cos(3.0 * uu) + 0.06 * np.sin(17.0 * vv + 7.0 * uu))
    pos = (centre[:, None, :] + radius[..., None] * (np.cos(vv)[..., None] * n1[:, None, :]
                                                      + np.sin(vv)[..., None] * n2[:, None, :]))
    verts = pos.reshape(-1, 3)
    i = np.arange(n_along)[:, None]
    j = np.arange(n_around)[None, :]
    a = i * n_around + j
    b = ((i + 1) % n_along) * n_around + j
    c = ((i + 1) % n_along) * n_around + (j + 1) % n_around
    d = i * n_around + (j + 1) % n_around
    faces = np.stack([np.stack([a, b, c], axis=-1), np.stack([a, c, d], axis=-1)], axis=2).reshape(-1, 3)
    return verts, faces


def read_obj(path):
    vs, fs = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v" and len(t) >= 4:
                vs.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] == "f" and len(t) >= 4:
                fs.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    return np.array(vs, dtype=np.float64), np.array(fs, dtype=np.int64)


def grid(base_paths, copies, spacing, seed):
    """copies = (nx, ny, nz); bases are cycled; each copy gets a seeded jitter, rotation about y, scale."""
    rng = np.random.RandomState(seed)
    bases = [read_obj(p) for p in base_paths]
    nx, ny, nz = copies
    all_v, all_f, off = [], [], 0
    k = 0
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                v, f = bases[k % len(bases)]
                k += 1
                ang = rng.uniform(0.0, 2.0 * np.pi)
                s = rng.uniform(0.8, 1.2)
                jit = rng.uniform(-0.15, 0.15, size=3) * spacing
                ca, sa = np.cos(ang), np.sin(ang)
                rot = np.array([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]])
                centre = np.array([(ix - (nx - 1) / 2.0), (iy - (ny - 1) / 2.0), (iz - (nz - 1) / 2.0)]) * spacing
                all_v.append((v - v.mean(axis=0)) @ rot.T * s + centre + jit)
                all_f.append(f + off)
                off += v.shape[0]
    return np.concatenate(all_v), np.concatenate(all_f)


def write_obj(path, verts, faces, comment):
    with open(path, "w") as f:
        f.write(f"# {comment}\n")
        for x, y, z in verts:
            f.write(f"v {x:.6f} {y:.6f} {z:.6f}\n")
        f.write("\n")
        for a, b, c in faces:
            f.write(f"f {a + 1} {b + 1} {c + 1}\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    d = sub.add_parser("dragon80k")
    d.add_argument("out")
    d.add_argument("--along", type=int, default=400)
    d.add_argument("--around", type=int, default=100)
    g = sub.add_parser("grid")
    g.add_argument("out")
    g.add_argument("--base", nargs="+", required=True)
    g.add_argument("--copies", type=int, nargs=3, required=True)
    g.add_argument("--spacing", type=float, default=8.0)
    g.add_argument("--seed", type=int, default=7)
    a = ap.parse_args(argv)
    if a.cmd == "dragon80k":
        v, f = dragon80k(a.along, a.around)
        write_obj(a.out, v, f, f"procedural torus-knot stand-in for the 80K-triangle dragon: {len(f)} triangles")
    else:
        v, f = grid(a.base, tuple(a.copies), a.spacing, a.seed)
        write_obj(a.out, v, f, f"baked grid of {a.copies} copies: {len(f)} triangles")
    print(f"{a.out}: {len(v)} vertices, {len(f)} triangles, extent {v.min(axis=0)} .. {v.max(axis=0)}")


if __name__ == "__main__":
    sys.exit(main())
