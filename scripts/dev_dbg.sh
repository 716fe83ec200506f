#!/bin/bash
# -DTDEC_DEBUG build on the GPU box; decoder time (six passes, random LLRs) for several SRSLTE_HIP_TDEC_DBG values:
# 16 all, +32 staging loads from one row (L1 hits), +64 no element-wise phases, 17 no sweeps
cd "$(dirname "$0")/../srslte-emane_amd/csrc"
cp libsrslte_phy_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -DTDEC_DEBUG $TDEC_FLAGS -c tdec.hip -o /tmp/tdec_dev.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so $(ls build/*.o | grep -v tdec.o) /tmp/tdec_dev.o || exit 1
cd ../..
for n in ${NCBS:-1664 6656}; do
  for d in ${DBGS:-16 48 80 112 17}; do
    SRSLTE_HIP_TDEC_DBG=$d python scripts/tdec_phase_timing.py 16 $n 2>&1 | tail -1
  done
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
