#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh into profiles/pmc_traffic.json, the file
bench.py reads for roofline.traffic.

Units and corrections (MI355X guide, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, i.e. reads exactly half the bytes of wide 16-B-per-lane reads, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  The path kernels issue 16-B-per-lane accesses (scattered,
not streaming), for which the factor is uncalibrated: the raw value is kept next to the corrected one.
Infinity-Cache hits are counted as traffic by these counters, so the number is an upper bound on HBM bytes.

The figures are stored per launch of the profiled run and per frame (bench.py scales the latter to the launches of its
own run); they describe the workload the profile was taken on, named in "_workload".

usage: pmc_traffic.py <gpurun_out/prof_TAG> <frames per launch> [out.json] [workload]
"""
import collections
import csv
import glob
import json
import os
import sys

CLASS = {"k_paths": "paths", "k_accumulate": "accumulate", "k_mega": "mega", "k_extend": "extend", "k_shade": "shade",
         "k_tail": "tail", "k_generate": "generate"}


def main():
    d, frames = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "profiles", "pmc_traffic.json")
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                    continue
                for key, cls in CLASS.items():
                    if key in row["Kernel_Name"]:
                        vals[cls][row["Counter_Name"]].append(float(row["Counter_Value"]))
    workload = sys.argv[4] if len(sys.argv) > 4 else "dragon80k_1080p"
    res = {"_source": os.path.basename(d.rstrip("/")), "_frames_per_launch": frames, "_workload": workload,
           "_formula": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024, largest launches only"}
    for cls, c in vals.items():
        # launches of the warm-up may be shorter: keep the launches with the largest write volume
        f, w = sorted(c.get("FETCH_SIZE", [0.0])), sorted(c.get("WRITE_SIZE", [0.0]))
        fk, wk = f[len(f) // 2:], w[len(w) // 2:]
        fetch, write = sum(fk) / len(fk), sum(wk) / len(wk)
        res[cls] = {"fetch_kib_raw": round(fetch, 1), "write_kib": round(write, 1),
                    "hbm_bytes_per_launch": round((2 * fetch + write) * 1024),
                    "hbm_bytes_per_frame": round((2 * fetch + write) * 1024 / frames)}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
