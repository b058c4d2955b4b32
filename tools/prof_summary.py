#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv) into the per-kernel table kept under
profiles/: calls, average / min / max duration, share of GPU time, registers, LDS."""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    c = sqlite3.connect(path)
    q = "select name, duration, vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels"
    for r in c.execute(q):
        yield r


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            yield (r["Kernel_Name"], dur, int(r.get("VGPR_Count", 0) or 0), int(r.get("SGPR_Count", 0) or 0),
                   int(r.get("LDS_Block_Size", 0) or 0), int(r.get("Scratch_Size", 0) or 0),
                   int(r.get("Grid_Size_X", 0) or 0), int(r.get("Workgroup_Size_X", 0) or 0))


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(lambda: [0, 0, 1 << 62, 0, 0, 0, 0, 0, 0, 0])
    for name, dur, vg, sg, lds, scr, gx, wx in rows:
        a = agg[name]
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
        a[4] = max(a[4], vg); a[5] = max(a[5], sg); a[6] = max(a[6], lds); a[7] = max(a[7], scr); a[8] = max(a[8], gx); a[9] = wx
    total = sum(a[1] for a in agg.values()) or 1
    print("%-72s %7s %10s %9s %9s %6s %5s %5s %7s %8s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "grid_x", "wg_x"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %7d %10.2f %9.2f %9.2f %6.2f %5d %5d %7d %8d %6d" % (name[:72], a[0], a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                                       100.0 * a[1] / total, a[4], a[5], a[6], a[8], a[9]))
    print("total GPU kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(a[0] for a in agg.values())))


if __name__ == "__main__":
    main()
