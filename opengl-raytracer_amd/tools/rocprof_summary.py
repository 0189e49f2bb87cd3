#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace and/or --pmc counter collection) into one small text
table per run: per kernel name the launch count, mean/min/max duration, and per counter the mean value
per launch.  Usage: rocprof_summary.py <rocprofv3 output dir> [more dirs...]"""
import collections
import csv
import glob
import os
import sys


def short(name):
    for key in ("k_primary", "k_rebase", "k_paths", "k_accumulate", "k_extend", "k_shade", "k_tail", "k_mega", "k_generate", "k_primary_hits", "k_normals", "k_selftest"):
        if key in name:
            tail = name[name.index(key) + len(key):]
            t = ""
            if tail.startswith("ILb"):
                t = "<" + ",".join(c for c in tail.split("E")[0] if c in "01")[:] + ">"
            return key + t
    return name[:60]


def main():
    for d in sys.argv[1:]:
        print(f"== {d}")
        for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            durs = collections.defaultdict(list)
            extra = {}
            with open(path) as f:
                for row in csv.DictReader(f):
                    n = short(row["Kernel_Name"])
                    durs[n].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
                    extra[n] = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Grid_Size"), row.get("Workgroup_Size"))
            print(f"-- kernel trace {os.path.basename(path)}  (durations in us)")
            print(f"{'kernel':28s} {'calls':>6s} {'mean':>10s} {'min':>10s} {'max':>10s} {'total_ms':>10s}  vgpr sgpr lds grid wg")
            for n, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
                print(f"{n:28s} {len(v):6d} {sum(v)/len(v):10.2f} {min(v):10.2f} {max(v):10.2f} {sum(v)/1e3:10.3f}  {' '.join(str(x) for x in extra[n])}")
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            vals = collections.defaultdict(lambda: collections.defaultdict(list))
            with open(path) as f:
                for row in csv.DictReader(f):
                    vals[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print(f"-- counters {os.path.basename(path)}  (mean per launch)")
            for n, cs in sorted(vals.items()):
                for cname, v in sorted(cs.items()):
                    print(f"{n:28s} {cname:32s} launches {len(v):5d} mean {sum(v)/len(v):18.1f}")


if __name__ == "__main__":
    main()
