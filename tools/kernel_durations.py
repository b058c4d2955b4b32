#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 --kernel-trace run (the *_kernel_trace.csv under DIR) as JSON for bench.py:

    python tools/kernel_durations.py DIR profiles/kernel_durations.json

Key = the kernel's name without `void`, `dartk::` and its argument list (template arguments kept), with "@batch" appended when the
launch's grid has more than one row (grid.y > 1: the frame / candidate dimension of the batched launches, so that the 1-frame and
the 64-frame launches of one kernel are not averaged together). Value = {calls, median_us, avg_us, min_us, max_us}."""
import csv
import glob
import json
import statistics
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").replace("dartk::", "").strip()


def main():
    import os
    files = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)     # the newest run last
    if not files:
        raise SystemExit("no *_kernel_trace.csv under %s" % sys.argv[1])
    d = defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        if "dartk::" not in r["Kernel_Name"]:
            continue
        wy = int(r.get("Workgroup_Size_Y", 1) or 1)
        rows = int(r.get("Grid_Size_Y", 1) or 1) // max(wy, 1)
        key = short(r["Kernel_Name"]) + ("@batch" if rows > 1 else "")
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {k: {"calls": len(v), "median_us": round(statistics.median(v), 3), "avg_us": round(sum(v) / len(v), 3),
               "min_us": round(min(v), 3), "max_us": round(max(v), 3)} for k, v in d.items()}
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"])[:40]:
        print("%-64s calls %5d median %9.2f avg %9.2f us" % (k[:64], v["calls"], v["median_us"], v["avg_us"]))


if __name__ == "__main__":
    main()
