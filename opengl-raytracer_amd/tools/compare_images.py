#!/usr/bin/env python3
"""Image / hit-buffer comparison used when checking a render against another implementation of the path.

  compare_images.py a.pfm b.pfm                     relative RMSE over RGB: sqrt(sum (a-b)^2 / sum b^2)
  compare_images.py --ids a.npz b.npz               primary-hit ids (arrays `ids`, `t`): mismatching pixels, split into
                                                    documented epsilon ties (both distances within 1e-4 * t) and real ones

PFM files are the ones host/rt_headless writes (bottom row first, little endian)."""
import argparse
import sys

import numpy as np


def read_pfm(path):
    with open(path, "rb") as f:
        kind = f.readline().strip()
        if kind not in (b"PF", b"Pf"):
            raise ValueError(f"{path}: not a PFM file")
        w, h = [int(x) for x in f.readline().split()]
        scale = float(f.readline())
        ch = 3 if kind == b"PF" else 1
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4")
        return data.reshape(h, w, ch).astype(np.float32)


def rrmse(a, b):
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def compare_ids(ids_a, t_a, ids_b, t_b, rel=1e-4):
    differ = ids_a != ids_b
    both_hit = differ & (ids_a >= 0) & (ids_b >= 0)
    tie = both_hit & (np.abs(t_a - t_b) <= rel * np.abs(t_b))
    return {"pixels": int(ids_a.size), "mismatch": int(differ.sum()), "epsilon_ties": int(tie.sum()),
            "real": int((differ & ~tie).sum()), "first_real": [int(x) for x in np.argwhere(differ & ~tie)[:1].ravel()]}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--ids", action="store_true")
    ap.add_argument("--tol", type=float, default=1e-3, help="rRMSE tolerance (exit status 1 above it)")
    a = ap.parse_args(argv)
    if a.ids:
        x, y = np.load(a.a), np.load(a.b)
        r = compare_ids(x["ids"], x["t"], y["ids"], y["t"])
        print(r)
        return 0 if r["real"] == 0 else 1
    e = rrmse(read_pfm(a.a), read_pfm(a.b))
    print(f"rRMSE = {e:.3e} (tolerance {a.tol:g})")
    return 0 if e <= a.tol else 1


if __name__ == "__main__":
    sys.exit(main())
