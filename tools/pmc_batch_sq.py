"""Per-kernel SQ counter means of a rocprofv3 --pmc pass over tools/time_batch.py (batch launches only, Grid_Size large):
python tools/pmc_batch_sq.py DIR"""
import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))[-1]
d = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dartk::", "")[:36]
    if int(r["Grid_Size"]) < 200000:
        continue
    key = name
    d[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = len(cnt[k]); w = max(v.get("SQ_WAVES", 1.0), 1.0); wc = max(v.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    print("%-36s launches %3d waves/launch %8.0f | per wave: VALU %7.0f VMEM_RD %6.0f | of wave cycles: waiting %4.1f%% issue-stall %4.1f%% active %4.1f%% (VALU %4.1f%%)"
          % (k, n, w / n, v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_VMEM_RD", 0) / w, 100 * v.get("SQ_WAIT_ANY", 0) / wc,
             100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc))
