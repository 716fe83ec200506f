#!/bin/bash
# scripts/ab_bench.sh on the 8-bit line (bench.py --llr8): gpurun_ab/libA.so against gpurun_ab/libB.so, alternating on one box.
set -e
cd "$(dirname "$0")/.."
cp srslte-emane_amd/csrc/libsrslte_phy_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in ${AB_VARIANTS:-A B}; do
    cp gpurun_ab/lib$v.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
    python bench.py --llr8 --no-cpu --no-full --stream-batch 0 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernels']['tdec']['ms'])"
  done
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
