# VALU-active-lane fraction of the batched term kernels: one SQ counter pass over the C4 batch with this
# round's term kernels and one with the round-2 forms switched back on (no tile culling, VALU outer product, one launch per term).
#   active-lane fraction = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)     (rocprof's VALUUtilization)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/valu
for v in now before; do
  if [ $v = before ]; then export DARTK_NOOBS_NO_TILE_CULL=1 DARTK_NOOBS_NO_MFMA=1 DARTK_NOOBS_NO_UNION=1; fi
  rm -rf gpurun_out/valu/$v
  timeout -s KILL 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/valu/$v -- python3 tools/time_grasp.py 1024 > gpurun_out/valu/$v.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for v in ("before", "now"):
    f = glob.glob("gpurun_out/valu/%s/**/*counter_collection.csv" % v, recursive=True)[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dartk::", "")
        if "noobs_terms" in k or "noobs_inter" in k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    print("== %s (%s)" % (v, "tile culling off, VALU outer product, one launch per term" if v == "before" else "this round's kernels"))
    for k in sorted(tot):
        t = tot[k]
        print("%-36s launches %4d  VALU instructions/wave %8.0f  active-lane fraction %.3f" % (
            k, len(n[k]), t["SQ_INSTS_VALU"] / max(t["SQ_WAVES"], 1), t["SQ_THREAD_CYCLES_VALU"] / max(t["SQ_ACTIVE_INST_VALU"] * 64.0, 1)))
PY
