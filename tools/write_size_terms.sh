# WRITE_SIZE (KiB, MI355X_MICROARCH.md "HBM") per launch of every term-kernel instantiation of the C4 batch (4096 candidates)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ws
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/ws/w -- python3 tools/time_grasp.py 4096 > gpurun_out/ws/w.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ws/w/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "WRITE_SIZE" and "noobs" in r["Kernel_Name"]:
        tot[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(tot.items()):
    v.sort()
    print("%-70s grid %-9s launches %4d  median WRITE %.1f KiB" % (k[0][-70:], k[1], len(v), v[len(v) // 2]))
PY
