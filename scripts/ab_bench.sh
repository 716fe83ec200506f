#!/bin/bash
# Same-box A/B of two builds of libsrslte_phy_hip.so (gpurun_ab/libA.so, gpurun_ab/libB.so; box-to-box spread on the pool is +-5 %):
# alternates them over the default bench line and prints subframes/s of every run.
set -e
cd "$(dirname "$0")/.."
cp srslte-emane_amd/csrc/libsrslte_phy_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in ${AB_VARIANTS:-A B}; do
    cp gpurun_ab/lib$v.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
    python bench.py --no-cpu --stream-batch 0 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernels']['tdec']['ms'], 'chest', d['kernels']['chest_dl']['ms'], 'ofdm', d['kernels']['ofdm_rx']['ms'], 'demod', d['kernels']['pdsch_demod']['ms'], 'rm', d['kernels']['rm_rx']['ms'])"
  done
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
