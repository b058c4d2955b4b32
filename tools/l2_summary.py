#!/usr/bin/env python3
"""Per-kernel L2 (TCC) hit rate from one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum pass: the second figure
SURVEY 8(d) asks for next to the HBM bytes (the SDF gathers are served by the L2 / MALL, not by HBM)."""
import csv
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dartk::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
import json
rates = {}
print("%-44s %8s %14s %14s %9s" % ("kernel", "launches", "L2 req/launch", "L2 miss/launch", "hit rate"))
for k in sorted(tot, key=lambda k: -tot[k].get("TCC_REQ_sum", 0)):
    n = len(disp[k])
    hit, miss, req = tot[k].get("TCC_HIT_sum", 0), tot[k].get("TCC_MISS_sum", 0), tot[k].get("TCC_REQ_sum", 0)
    if hit + miss == 0:
        continue
    print("%-44s %8d %14.0f %14.0f %8.1f%%" % (k[:44], n, req / n, miss / n, 100.0 * hit / (hit + miss)))
    rates[k] = {"launches": n, "hit_rate": round(hit / (hit + miss), 4), "requests_per_launch": round(req / n)}
if len(sys.argv) > 2:       # JSON for bench.py (profiles/l2_hit_rate.json)
    json.dump(rates, open(sys.argv[2], "w"), indent=1, sort_keys=True)
