#!/usr/bin/env python3
"""Puts together the partial images `rt_headless --rank r --world n` wrote (one PFM per rank, rank order).

  samples  every partial holds the SUM of its rank's frames: the image is their sum divided by the total frame count
           (the value a host that links RCCL gets from ncclReduce(sum) + one scale);
  tiles    every partial holds its rank's interleaved band (stripes of 16 pixel rows dealt round-robin) and zeros
           elsewhere: the image takes each row from the rank that owns it (a gather), bit for bit.

usage: combine_partials.py samples <total frames> out.pfm part0.pfm part1.pfm ...
       combine_partials.py tiles out.pfm part0.pfm part1.pfm ...
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import partition  # noqa: E402


def read_pfm(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError(f"{path}: not a colour PFM")
        w, h = [int(x) for x in f.readline().split()]
        if float(f.readline()) >= 0:
            raise ValueError(f"{path}: big-endian PFM not supported")
        return np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3).copy()


def write_pfm(path, img):
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(img, dtype="<f4").tobytes())


def combine_samples(parts, total_frames):
    acc = parts[0].copy()
    for p in parts[1:]:
        acc += p            # fp32, rank order: what a ring / tree reduce does up to the order of the additions
    return acc * np.float32(1.0 / float(total_frames))


def combine_tiles(parts):
    h = parts[0].shape[0]
    out = np.zeros_like(parts[0])
    for rank, p in enumerate(parts):
        for r0, r1 in partition.band_row_ranges(rank, len(parts), h, True):
            out[r0:r1] = p[r0:r1]
    return out


def main():
    mode = sys.argv[1]
    if mode == "samples":
        total, out, files = int(sys.argv[2]), sys.argv[3], sys.argv[4:]
        img = combine_samples([read_pfm(f) for f in files], total)
    elif mode == "tiles":
        out, files = sys.argv[2], sys.argv[3:]
        img = combine_tiles([read_pfm(f) for f in files])
    else:
        sys.exit(__doc__)
    write_pfm(out, img)
    print(f"wrote {out}: {img.shape[1]}x{img.shape[0]} from {len(files)} partial images")


if __name__ == "__main__":
    main()
