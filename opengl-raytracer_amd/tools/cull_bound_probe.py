#!/usr/bin/env python3
"""How far from a triangle can a ray be and still be reported as a hit by the binary32 triangle test?

The subtree cull of the path kernel rests on a bound for exactly that distance (fp + 4*E*m2 + u*m2, see cull_pad_of in
csrc/rt_api.hip).  This probe attacks the bound numerically: it builds the worst cases the derivation names -- rays that graze
the triangle's plane (|det| just above the test's rejection threshold E, where the division by det amplifies every rounding),
origins as far away as the scene allows (|s| up to D), aimed a little outside an edge or a corner -- evaluates the test
in binary32 with the kernel's operation order (numpy float32 arithmetic rounds every operation, no contraction), and for every
reported hit measures in binary64 the true distance between the ray and the triangle.  It prints the largest distance found
next to the bound.  CPU only.

usage: cull_bound_probe.py [--samples N] [--D 9.47] [--m2 0.146] [--m1 0.0188] [--seed 1]
"""
import argparse

import numpy as np

E = np.float32(1e-4)
f32 = np.float32


def cross32(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)


def dot32(a, b):
    return (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2]


def tri_test32(o, d, a, e1, e2):
    """raytrace.glsl:79-110 in binary32, the order of rt_device.h: tri_dist_flat.  Returns (hit mask, t)."""
    with np.errstate(all="ignore"):
        p = cross32(d, e2)
        det = dot32(e1, p)
        inv = f32(1.0) / det
        s = o - a
        u = inv * dot32(s, p)
        q = cross32(s, e1)
        v = inv * dot32(d, q)
        t = inv * dot32(e2, q)
        rej = np.abs(det) < E
        rej |= (u < 0) & (np.abs(u) > E)
        rej |= (u > 1) & (np.abs(u - f32(1.0)) > E)
        rej |= (v < 0) & (np.abs(v) > E)
        rej |= (u + v > 1) & (np.abs((u + v) - f32(1.0)) > E)
        rej |= ~(t > E)
    return ~rej, t, det


def seg_ray_dist(o, d, p0, p1):
    """binary64 distance between the half-line o + t d (t >= 0) and the segment p0 p1, row-wise."""
    w = p1 - p0
    r = p0 - o
    a = (d * d).sum(1)
    b = (d * w).sum(1)
    c = (w * w).sum(1)
    dd = (d * r).sum(1)
    e = (w * r).sum(1)
    den = a * c - b * b
    with np.errstate(all="ignore"):
        sN = np.where(den > 1e-300, (b * dd - a * e) / den, 0.0)   # parameter on the segment
    sN = np.clip(sN, 0.0, 1.0)
    best = np.full(len(o), np.inf)
    for sc in (sN, np.zeros_like(sN), np.ones_like(sN)):
        pt = p0 + sc[:, None] * w
        t = np.maximum(((pt - o) * d).sum(1) / a, 0.0)
        best = np.minimum(best, np.linalg.norm(o + t[:, None] * d - pt, axis=1))
    # the ray's origin against the segment
    sO = np.clip(-(e) / np.where(c > 0, c, 1.0), 0.0, 1.0)
    best = np.minimum(best, np.linalg.norm(p0 + sO[:, None] * w - o, axis=1))
    return best


def ray_tri_dist(o, d, a, e1, e2):
    """binary64 distance between the half-line and the triangle."""
    n = np.cross(e1, e2)
    den = (n * d).sum(1)
    with np.errstate(all="ignore"):
        t = ((a - o) * n).sum(1) / den
    hitp = o + t[:, None] * d - a
    # barycentrics of the plane hit
    m11, m12, m22 = (e1 * e1).sum(1), (e1 * e2).sum(1), (e2 * e2).sum(1)
    r1, r2 = (hitp * e1).sum(1), (hitp * e2).sum(1)
    dm = m11 * m22 - m12 * m12
    with np.errstate(all="ignore"):
        u = (r1 * m22 - r2 * m12) / dm
        v = (r2 * m11 - r1 * m12) / dm
    inside = np.isfinite(t) & (t >= 0) & (u >= 0) & (v >= 0) & (u + v <= 1)
    dist = np.minimum.reduce([seg_ray_dist(o, d, a, a + e1), seg_ray_dist(o, d, a, a + e2), seg_ray_dist(o, d, a + e1, a + e2)])
    return np.where(inside, 0.0, dist)


def bound(D, m1, m2):
    u = 1.0 / 16777216.0
    e = 1e-4
    fp = (u / e) * m1 * (29.0 * D + 26.5 * m2) + 2.0 * u * (D + 4.1 * m2)
    return fp, fp + 4.0 * e * m2 + u * m2


def probe(samples, D, m1, m2, seed):
    rng = np.random.default_rng(seed)
    worst = 0.0
    hits = 0
    worst_case = None
    block = 200000
    for _ in range(max(1, samples // block)):
        n = block
        # triangle: edges up to m2 with product up to m1, any angle, anywhere
        l1 = m2 * rng.uniform(0.2, 1.0, n)
        l2 = np.minimum(m2 * rng.uniform(0.2, 1.0, n), m1 / l1)
        ang = rng.uniform(0.05, np.pi - 0.05, n)
        rot = np.linalg.qr(rng.normal(size=(n, 3, 3)))[0]
        e1 = np.einsum("nij,nj->ni", rot, np.stack([l1, 0 * l1, 0 * l1], 1))
        e2 = np.einsum("nij,nj->ni", rot, np.stack([l2 * np.cos(ang), l2 * np.sin(ang), 0 * l2], 1))
        nrm = np.einsum("nij,nj->ni", rot, np.tile([0.0, 0.0, 1.0], (n, 1)))
        a = rng.uniform(-D / 3, D / 3, (n, 3))
        # target: a point near the boundary of the triangle (a little inside or outside an edge or corner, in the plane)
        k = rng.integers(0, 3, n)
        w = rng.uniform(0, 1, n)
        w = np.where(rng.uniform(size=n) < 0.3, np.round(w), w)     # corners too
        edge_p = np.where((k == 0)[:, None], a + w[:, None] * e1, np.where((k == 1)[:, None], a + w[:, None] * e2, a + e1 + w[:, None] * (e2 - e1)))
        inplane = np.einsum("nij,nj->ni", rot, np.stack([np.cos(th := rng.uniform(0, 2 * np.pi, n)), np.sin(th), 0 * th], 1))
        off = 10.0 ** rng.uniform(-7, -2.5, n) * m2
        target = edge_p + off[:, None] * inplane
        # direction: grazing, |det| = |e1 x e2| * |cos(angle to the normal)| from just under E upwards
        area2 = np.linalg.norm(np.cross(e1, e2), axis=1)
        cosn = np.minimum(1.0, 10.0 ** rng.uniform(-0.3, 2.0, n) * 1e-4 / area2)
        tang = np.cross(nrm, inplane)
        phi = rng.uniform(0, 2 * np.pi, n)
        flat = np.cos(phi)[:, None] * inplane + np.sin(phi)[:, None] * tang
        d = np.sqrt(1 - cosn ** 2)[:, None] * flat + (cosn * rng.choice([-1.0, 1.0], n))[:, None] * nrm
        dist = D * rng.uniform(0.02, 1.0, n)
        o = target - dist[:, None] * d
        # binary32 inputs, as the kernel sees them (direction normalised in binary32 like primary / bounce directions)
        o32, a32, e132, e232 = o.astype(f32), a.astype(f32), e1.astype(f32), e2.astype(f32)
        d32 = d.astype(f32)
        d32 = d32 / np.sqrt(dot32(d32, d32))[:, None]
        hit, t, det = tri_test32(o32, d32, a32, e132, e232)
        if not hit.any():
            continue
        idx = np.nonzero(hit)[0]
        dd = ray_tri_dist(o32[idx].astype(np.float64), d32[idx].astype(np.float64), a32[idx].astype(np.float64),
                          e132[idx].astype(np.float64), e232[idx].astype(np.float64))
        hits += len(idx)
        j = int(np.argmax(dd))
        if dd[j] > worst:
            worst = float(dd[j])
            worst_case = dict(det=float(det[idx[j]]), s=float(np.linalg.norm(o32[idx[j]] - a32[idx[j]])), off=float(off[idx[j]]))
    return worst, hits, worst_case


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=4000000)
    ap.add_argument("--D", type=float, default=9.47)
    ap.add_argument("--m2", type=float, default=0.146)
    ap.add_argument("--m1", type=float, default=0.0188)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    worst, hits, case = probe(a.samples, a.D, a.m1, a.m2, a.seed)
    fp, total = bound(a.D, a.m1, a.m2)
    print(f"{a.samples} rays aimed at the rim of grazing triangles (D {a.D}, m1 {a.m1}, m2 {a.m2}): {hits} reported hits, "
          f"largest true distance ray - triangle {worst:.3e} ({case}); bound fp {fp:.3e}, fp + 4 E m2 + u m2 = {total:.3e}: "
          f"{total / worst if worst > 0 else float('inf'):.1f} x the worst case found")


if __name__ == "__main__":
    main()
