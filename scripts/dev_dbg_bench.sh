#!/bin/bash
# -DTDEC_DEBUG build on the GPU box; the bench line (all blocks at 6 passes: SRSLTE_HIP_TDEC_DBG 16) with and without the element-wise phases (+64)
cd "$(dirname "$0")/../srslte-emane_amd/csrc"
cp libsrslte_phy_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -DTDEC_DEBUG -c tdec.hip -o /tmp/tdec_dev.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so $(ls build/*.o | grep -v tdec.o) /tmp/tdec_dev.o || exit 1
cd ../..
for d in ${DBGS:-16 80 16 80}; do
  SRSLTE_HIP_TDEC_DBG=$d python bench.py --no-cpu --no-full --steps 20 --stream-batch 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dbg $d:', d['value'], d['ms_per_step'], 'passes', d['config']['avg_siso_passes_per_cb'])"
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
