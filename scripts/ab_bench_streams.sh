#!/bin/bash
# Same-box A/B of two builds (gpurun_ab/libA.so, libB.so) over stream counts and batch sizes: prints subframes/s, ms per step, turbo ms alone.
set -e
cd "$(dirname "$0")/.."
cp srslte-emane_amd/csrc/libsrslte_phy_hip.so /tmp/lib_keep.so
for cfg in "--streams 1" "--streams 4" "--streams 8" "--streams 2 --batch 512" "--streams 4 --batch 512"; do
  for v in ${AB_VARIANTS:-A B}; do
    cp gpurun_ab/lib$v.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
    python bench.py --no-cpu --stream-batch 0 --no-full --min-timed-s 0.2 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$cfg', d['value'], d['ms_per_step'], d['kernels']['tdec']['ms'], d['config']['avg_siso_passes_per_cb'])"
  done
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
