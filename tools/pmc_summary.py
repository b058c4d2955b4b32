#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as gfx950's TCC slots
require). Units and corrections per MI355X_MICROARCH.md "HBM": counter values are KiB; on gfx950 FETCH_SIZE reports one
half of the bytes of wide coalesced reads, so the read side is doubled (an upper bound for the 4-byte gathers that dominate
these kernels: access widths other than 16 B/lane are uncalibrated). Output: table + profiles/pmc_traffic.json
(bytes per launch = 2*FETCH + WRITE)."""
import csv
import json
import sys
from collections import defaultdict


LARGEST = False     # --largest: per kernel only the launches with at least half its largest grid (the batch pass: a process that
                    # launches a kernel for 1 frame while it builds its inputs and for 64 frames in the timed loop)


def per_kernel(path, counter):
    tot = defaultdict(float)
    disp = defaultdict(set)
    rows = []
    big = defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            rows.append(r)
            big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"]
        if LARGEST and 2 * int(r["Grid_Size"]) < big[k]:
            continue
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: (tot[k] / max(len(disp[k]), 1), len(disp[k])) for k in tot}


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("dartk::", "")
    if n.startswith("noobs_inter_union_kernel"):  # the intersection terms of an iteration in one launch
        return "noobs_terms_kernel<intersection>"
    if n.startswith("noobs_terms_kernel<"):       # one instantiation per term type: keep them apart (bench.py's class names)
        typ = n[n.index("<") + 1:n.index(">")].split(",")[-1].strip()
        return "noobs_terms_kernel<priors>" if typ == "2" else "noobs_terms_kernel<intersection>"
    return n.split("<")[0]


def main():
    global LARGEST
    if "--largest" in sys.argv:
        LARGEST = True
        sys.argv.remove("--largest")
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out, launches = {}, {}
    print("%-40s %8s %16s %16s %18s" % ("kernel", "launches", "FETCH_KiB/launch", "WRITE_KiB/launch", "HBM_bytes/launch"))
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        b = (2.0 * f + w) * 1024.0
        print("%-40s %8d %16.1f %16.1f %18.0f" % (short(k)[:40], n, f, w, b))
        if "dartk::" in k:      # instantiations that share a name: launch-weighted mean
            key = short(k)
            out[key] = out.get(key, 0.0) + b * n
            launches[key] = launches.get(key, 0) + n
    out = {k: round(v / launches[k]) for k, v in out.items()}
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
