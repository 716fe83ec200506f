#!/bin/bash
# Development loop on the GPU box: rebuild tdec.hip with extra flags ($TDEC_FLAGS, e.g. -DTDEC_PROF), relink, time both 16-window kernels.
cd "$(dirname "$0")/../srslte-emane_amd/csrc"
cp libsrslte_phy_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. $TDEC_FLAGS -c tdec.hip -o /tmp/tdec_dev.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so $(ls build/*.o | grep -v tdec.o) /tmp/tdec_dev.o || exit 1
cd ../..
for n in ${NCBS:-1664 6656}; do
  SRSLTE_HIP_TDEC_PROF=1 python scripts/tdec_phase_timing.py 16 $n 2>&1 | tail -2
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
