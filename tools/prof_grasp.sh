cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pg
for v in x; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pg/c$v -- python3 tools/time_grasp.py 4096 > gpurun_out/pg/c$v.log 2>&1 || exit 1
echo "== $v"; grep "B=4096" gpurun_out/pg/c$v.log | tail -1; python3 tools/split_inter.py gpurun_out/pg/c$v
done
