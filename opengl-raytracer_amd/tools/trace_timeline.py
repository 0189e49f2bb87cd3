#!/usr/bin/env python3
"""Prints the kernels of a rocprofv3 kernel trace in start order with start / end relative to the first one (ms), queue id and
duration: shows which launches overlapped.  usage: trace_timeline.py <dir with *_kernel_trace.csv> [first] [count]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            rows += list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[first:first + count] if first >= 0 else rows[first:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        name = r["Kernel_Name"]
        for key in ("k_primary", "k_paths", "k_accumulate"):
            if key in name:
                name = key + ("<counting>" if "<true" in name or "ILb1" in name else "")
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{name[:40]:40s} queue {r.get('Queue_Id', '?'):>3s}  start {s / 1e6:9.3f}  end {e / 1e6:9.3f}  dur {(e - s) / 1e6:8.3f} ms  grid {r.get('Grid_Size')}")


if __name__ == "__main__":
    main()
