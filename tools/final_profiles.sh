#!/bin/bash
# The round's measurement passes on the GPU box (each step under its own limit; stops at the first failure):
#   kernel stats, HBM read / write traffic (two PMC passes), L2 hit rate (one PMC pass), of the same bench command.
#   The counter passes leave the config-5 batch out: its launches carry the same kernel names as the single-frame loop's
#   (256 frames per launch), and a per-launch mean over both says nothing about either; the batch gets two passes of its own.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --grasp-steps 1 --repeats 0 --multi-steps 1 --no-pose-batch"
# the single-model frame batch alone (its launches share kernel names with the other legs'): a kernel trace and two PMC passes of its own
CMDP="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --repeats 0 --no-tracking --no-grasps --no-multi-model --pose-steps 1"
O=gpurun_out/final
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1
CMD1="$CMD --no-multi-model"
CMD5="$CMD --no-grasps --no-tracking"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD1 > $O/fetch.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD1 > $O/write.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/l2 -- $CMD1 > $O/l2.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch5 -- $CMD5 > $O/fetch5.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write5 -- $CMD5 > $O/write5.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsP -- $CMDP > $O/statsP.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetchP -- $CMDP > $O/fetchP.log 2>&1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/writeP -- $CMDP > $O/writeP.log 2>&1
echo profiles done; set +e
# extras of round 3: WRITE_SIZE of the term-kernel instantiations (scratch check), VALU lane fraction before / after, the config-5
# single-frame iteration's timeline and host enqueue time
bash tools/write_size_terms.sh > $O/write_size_terms.txt 2>&1
bash tools/pmc_valu_lanes.sh > $O/valu_lanes.txt 2>&1
bash tools/timeline_multi.sh > $O/timeline_multi.txt 2>&1
echo extras done
