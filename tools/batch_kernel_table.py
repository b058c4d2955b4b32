"""Per-kernel durations (grouped by grid) of the batch loop from a rocprofv3 --kernel-trace csv directory: python tools/batch_kernel_table.py DIR"""
import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
g = collections.defaultdict(list)
for r in rows:
    if r['Grid_Size_Y'] == '1' and 'multi_solve' not in r['Kernel_Name']:
        continue
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dartk::', '')[:40]
    g[(n, r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0.0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:20]:
    print("%-42s grid %8s x %4s calls %4d avg %8.1f us total %8.2f ms" % (k[0], k[1], k[2], len(v), sum(v) / len(v), sum(v) / 1e3))
    tot += sum(v)
print("sum of the listed kernels: %.2f ms" % (tot / 1e3))
