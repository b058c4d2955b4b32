import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = {}
k = 0
for r in rows:
    n = r["Kernel_Name"]
    if "noobs_terms_kernel" not in n:
        continue
    t = n[n.index("<"):n.index(">") + 1]
    if t.endswith(" 1>"):
        t += "ab"[k % 2]; k += 1
    d.setdefault(t, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("  ".join("%s %.0f us" % (t, sum(v) / len(v)) for t, v in sorted(d.items())))
