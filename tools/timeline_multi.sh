# one iteration's kernel timeline of the config-5 single-frame loop + host enqueue time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tlm && mkdir -p gpurun_out/tlm
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlm -- python3 tools/time_config5.py > gpurun_out/tlm/run.log 2>&1 || exit 1
tail -2 gpurun_out/tlm/run.log
python3 tools/timeline_multi.py gpurun_out/tlm
timeout -k 10 300 python3 tools/time_enqueue_multi.py
