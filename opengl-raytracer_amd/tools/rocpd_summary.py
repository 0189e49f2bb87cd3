#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the *_results.db a --kernel-trace run writes):
rocpd_summary.py <results.db> [--levels KERNEL_SUBSTRING ...]   (--levels prints the durations of the last third of
that kernel's launches in launch order, e.g. one value per level of the last BVH build)"""
import collections
import sqlite3
import sys


def main():
    path = sys.argv[1]
    level_keys = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == "--levels" else []
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, vgpr_count, lds_size, grid_x, workgroup_x from kernels order by start").fetchall()
    d = collections.defaultdict(list)
    meta = {}
    for n, s, e, vg, lds, gx, wx in rows:
        key = n.split("(")[0][:44]
        d[key].append((e - s) / 1e3)
        meta[key] = (vg, lds, wx)
    print(f"{'kernel':46s} {'calls':>6s} {'mean_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s}  vgpr lds wg")
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:46s} {len(v):6d} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {sum(v) / 1e3:9.2f}  {meta[n][0]} {meta[n][1]} {meta[n][2]}")
    for key in level_keys:
        v = [(e - s) / 1e3 for n, s, e, *_ in rows if key in n]
        print(key, "last third, us:", [round(x) for x in v[-(len(v) // 3):]])


if __name__ == "__main__":
    main()
