#!/bin/bash
# Same-box A/B of two library builds over the three lines a decoder change can move: default, --llr8, --grants-mix (value and ms per step each).
cd "$(dirname "$0")/.."
cp srslte-emane_amd/csrc/libsrslte_phy_hip.so /tmp/lib_keep.so
for round in 1 2; do
  for v in ${AB_VARIANTS:-OLD NEW}; do
    cp gpurun_ab/lib$v.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
    for f in "" "--llr8" "--grants-mix"; do
      python bench.py --no-cpu --stream-batch 0 --steps 40 $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v [$f]', d['value'], d['ms_per_step'])"
    done
  done
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
