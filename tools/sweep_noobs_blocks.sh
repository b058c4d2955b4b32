#!/bin/bash
# diagnostic: batched ContactGrasp throughput vs workgroups per term (DARTK_NOOBS_BLOCKS override) and batch size
for B in 256 1024 4096; do for nb in 2 4 8 16; do
  r=$(DARTK_NOOBS_BLOCKS=$nb timeout -k 10 120 python tools/time_grasp.py $B 2>/dev/null | tail -1 | sed 's/.*-> //; s/,.*//')
  echo "B=$B nb=$nb: $r"
done; done
