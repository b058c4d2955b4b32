"""Per-dispatch summary of a rocprofv3 --pmc run of tools/time_grasp.py: the noobs term kernels in launch order."""
import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "noobs_terms" not in r["Kernel_Name"]:
        continue
    k = int(r["Dispatch_Id"])
    e = d.setdefault(k, {"name": r["Kernel_Name"].split("(")[0][-26:], "grid": r["Grid_Size"], "lds": r["LDS_Block_Size"], "vgpr": r["VGPR_Count"],
                         "agpr": r["Accum_VGPR_Count"], "scratch": r["Scratch_Size"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for i in ids[len(ids) // 2:len(ids) // 2 + n]:
    v = d[i]
    w = v.get("SQ_WAVES", 1.0)
    print(v["name"], "grid", v["grid"], "lds", v["lds"], "vgpr", v["vgpr"], "+", v["agpr"], "scratch", v["scratch"], "us", v["ns"] / 1e3)
    print("    per wave:", {k[3:]: round(x / w, 1) for k, x in v.items() if k.startswith("SQ_") and k != "SQ_WAVES"}, "waves", w)
