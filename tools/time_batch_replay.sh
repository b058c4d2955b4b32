#!/bin/bash
# sweeps on the config-5 batch leg: workgroups per frame of the batched trace (DARTK_TRACE_WGS); budget of the batch replay (DARTK_REPLAY_WGS)
for w in 24 48 96 192 384; do
  DARTK_TRACE_WGS=$w timeout -k 10 300 python bench.py --no-tracking --no-grasps --no-cpu-baseline --steps 1 --warmup 0 --repeats 0 --multi-steps 3 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*, \"iterations_per_step\": 10" | sed "s/^/trace wgs per frame $w /"
done
