#!/bin/bash
# Turns the raw rocprofv3 output of tools/final_profiles.sh (gpurun_out/final/) into the summaries kept under profiles/:
#   tools/collect_profiles.sh r03_a      ->  profiles/r03_a_kernel_stats.txt, _rocprofv3_kernel_stats.csv, _pmc_hbm_traffic.txt,
#   _pmc_hbm_traffic_batch.txt, _pmc_l2_hit_rate.txt  and the JSON tables bench.py reads (kernel_durations.json, pmc_traffic.json,
#   pmc_traffic_batch.json, l2_hit_rate.json)
set -e
TAG=$1
O=gpurun_out/final
one() { ls -t $O/$1/*/*_$2.csv | head -1; }      # the newest: gpurun_out/ keeps the files of earlier runs
python3 tools/prof_summary.py "$(one stats kernel_trace)" > profiles/${TAG}_kernel_stats.txt
cp "$(one stats kernel_stats)" profiles/${TAG}_rocprofv3_kernel_stats.csv
python3 tools/kernel_durations.py $O/stats profiles/kernel_durations.json > /dev/null
python3 tools/pmc_summary.py "$(one fetch counter_collection)" "$(one write counter_collection)" profiles/pmc_traffic.json > profiles/${TAG}_pmc_hbm_traffic.txt
python3 tools/pmc_summary.py --largest "$(one fetch5 counter_collection)" "$(one write5 counter_collection)" profiles/pmc_traffic_batch.json > profiles/${TAG}_pmc_hbm_traffic_batch.txt
python3 tools/l2_summary.py "$(one l2 counter_collection)" profiles/l2_hit_rate.json > profiles/${TAG}_pmc_l2_hit_rate.txt
python3 tools/prof_summary.py "$(one statsP kernel_trace)" > profiles/${TAG}_kernel_stats_pose_batch.txt
python3 tools/kernel_durations.py $O/statsP profiles/kernel_durations_pose_batch.json > /dev/null
python3 tools/pmc_summary.py --largest "$(one fetchP counter_collection)" "$(one writeP counter_collection)" profiles/pmc_traffic_pose_batch.json > profiles/${TAG}_pmc_hbm_traffic_pose_batch.txt
echo "profiles/${TAG}_* written"
for f in write_size_terms valu_lanes timeline_multi; do
  if [ -f $O/$f.txt ]; then grep -v "rocprofv3\]\|amdgpu.ids" $O/$f.txt > profiles/${TAG}_$f.txt; fi
done
