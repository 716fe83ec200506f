#!/bin/bash
# Same-box A/B of bench.py FLAG sets (box-to-box spread on the pool is +-5 %): AB_FLAGS="--zero-copy|" alternates the sets (separated by |) over the
# default line, three rounds, and prints subframes/s of every run.
cd "$(dirname "$0")/.."
IFS='|' read -ra SETS <<< "${AB_FLAGS:-|}"
for round in 1 2 3; do
  for f in "${SETS[@]}"; do
    python bench.py --no-cpu --stream-batch 0 --steps 40 $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f]', d['value'], d['ms_per_step'], 'verified', d['config'].get('pipeline_instances_verified'), d['config'].get('gather_verified'))"
  done
done
