#!/bin/bash
# headline line at several stream counts; prints value, ms/step, same-input value, tdec alone ms, full-iter value
cd "$(dirname "$0")/.."
for st in ${STREAMS:-4 1}; do
  python bench.py --no-cpu --steps ${STEPS:-20} --stream-batch 0 --streams $st $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; print('streams $st:', d['value'], d['ms_per_step'], 'same-input', c['same_input_value'], 'tdec alone', d['kernels']['tdec']['ms'], 'full', c['full_iter_value'], 'bler', c['bler'], 'passes', c['avg_siso_passes_per_cb'], c['avg_siso_passes_per_wavefront'], 'verified', c['pipeline_instances_verified'])"
done
