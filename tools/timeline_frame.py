"""Prints one frame's device timeline (kernels and copies) of tools/time_single_loop.py from a rocprofv3 --kernel-trace
--memory-copy-trace csv directory: everything from the end of one frame's last multi_solve_kernel to the first solve of the next."""
import csv, glob, os, sys
d = sys.argv[1]
kf = sorted(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
ev = []
for r in csv.DictReader(open(kf)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dartk::", "")[:60])))
mf = glob.glob(os.path.dirname(kf) + "/*_memory_copy_trace.csv")
if mf:
    for r in csv.DictReader(open(mf[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy engine: %s" % r.get("Direction", "?")))
ev.sort()
solves = [i for i, e in enumerate(ev) if "multi_solve_kernel" in e[2]]
# frames are 10 solves apart; take the boundary between the third-last and second-last frame
a = solves[-21]
b = solves[-20]
t0 = ev[a][1]
print("from the end of a frame's last solve to the end of the next frame's first solve (us):")
for e in ev[a + 1:b + 1]:
    print("%8.2f %8.2f  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, e[2]))
