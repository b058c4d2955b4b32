# A/B of two builds of the library on one box: per-term kernel time of the C4 batch under rocprofv3, product build vs libdartk_$1.so, twice each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pg
for v in prod $1 prod $1; do
  if [ $v = prod ]; then unset DARTK_LIB; else export DARTK_LIB=$GRAFT_REPO_ROOT/dart_amd/libdartk_$v.so; fi
  rm -rf gpurun_out/pg/c$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pg/c$v -- python3 tools/time_grasp.py 4096 > gpurun_out/pg/c$v.log 2>&1 || exit 1
  echo "== $v"; grep "B=4096" gpurun_out/pg/c$v.log | tail -1; python3 tools/split_inter.py gpurun_out/pg/c$v
done
