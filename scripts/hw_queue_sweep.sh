#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run their kernels in order.
# Same box, alternating: the default bench line for (queues, streams) pairs; subframes/s and ms per step of every run.
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for qs in "4 4" "8 8" "16 8" "16 12"; do
    set -- $qs
    export GPU_MAX_HW_QUEUES=$1
    python bench.py --no-cpu --no-full --stream-batch 0 --steps 40 --streams $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $1 streams $2', d['value'], d['ms_per_step'])"
  done
done
