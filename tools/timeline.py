"""Prints one Gauss-Newton iteration's kernel timeline (start/end relative to the first kernel, stream) from a rocprofv3
--kernel-trace csv of bench.py --no-grasps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last-but-two solve kernel and print everything between it and the next one
idx = [i for i, r in enumerate(rows) if "gn_solve_single" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]["End_Timestamp"])
print("from the end of one solve kernel to the end of the next (us):")
for r in rows[a + 1:b + 1]:
    print("%8.2f %8.2f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"),
                                   r["Kernel_Name"].split("(")[0][-40:]))
