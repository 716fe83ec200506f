#!/bin/bash
# K = 20 (the driver's command): streams x hardware queues, three rounds, subframes/s and ms per step of python bench.py --steps 20
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for sq in "8 8" "10 10" "20 20" "5 8" "7 7" "4 4" "12 12"; do
    set -- $sq
    GPU_MAX_HW_QUEUES=$2 python bench.py --no-cpu --stream-batch 0 --steps 20 --streams $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $1 queues $2', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['steady_state']['fill_drain_ms_per_region'])"
  done
done
