#!/bin/bash
# Timing experiments on the GPU box: -DTDEC_DEBUG builds of tdec.hip with extra flags per variant ($VARIANTS = "name:flags;name:flags"),
# decoder time for $DBGS (default 80: six passes, no element-wise phases).
cd "$(dirname "$0")/../srslte-emane_amd/csrc"
cp libsrslte_phy_hip.so /tmp/lib_keep.so
OBJS=$(ls build/*.o | grep -v tdec.o | tr '\n' ' ')
echo "$VARIANTS" | tr ';' '\n' | while read -r v; do
  name="${v%%:*}"; flags="${v#*:}"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -DTDEC_DEBUG $flags -c tdec.hip -o /tmp/tdec_dev.o 2>/dev/null || { echo "build failed: $name"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so $OBJS /tmp/tdec_dev.o || exit 1
  for n in ${NCBS:-1664}; do
    for d in ${DBGS:-80}; do
      echo -n "$name: "; (cd ../.. && SRSLTE_HIP_TDEC_DBG=$d python scripts/tdec_phase_timing.py 16 $n 2>&1 | tail -1)
    done
  done
done
cp /tmp/lib_keep.so libsrslte_phy_hip.so
