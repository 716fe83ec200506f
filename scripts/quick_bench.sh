#!/bin/bash
# headline line at 4 streams and at 1 stream, decoder kernel alone; prints value, ms/step, tdec alone ms, full-iter value
cd "$(dirname "$0")/.."
for st in ${STREAMS:-4 1}; do
  python bench.py --no-cpu --steps 20 --stream-batch 0 --streams $st $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('streams $st:', d['value'], d['ms_per_step'], 'tdec alone', d['kernels']['tdec']['ms'], 'full', d['config']['full_iter_value'], 'bler', d['config']['bler'], 'verified', d['config']['pipeline_instances_verified'])"
done
