"""Prints one iteration's kernel timeline of the multi-model single-frame loop (dartk_optimize_poses) from a rocprofv3
--kernel-trace csv of tools/time_config5.py: everything between two consecutive multi_solve_kernel launches near the end."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "multi_solve_kernel" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]["End_Timestamp"])
print("from the end of one multi_solve_kernel to the end of the next (us):")
for r in rows[a + 1:b + 1]:
    print("%8.2f %8.2f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"),
                                   r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dartk::", "")[:60]))
