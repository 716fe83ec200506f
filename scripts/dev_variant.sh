#!/bin/bash
# Development loop on the GPU box: for every flag set in VARIANTS (separated by ';') rebuild tdec.hip with it, relink, run scripts/quick_bench.sh.
#   VARIANTS="-DTDEC_PAIR_WAVES=3;-DTDEC_PAIR_WAVES=2" bash scripts/dev_variant.sh
cd "$(dirname "$0")/../srslte-emane_amd/csrc"
cp libsrslte_phy_hip.so /tmp/lib_keep.so
IFS=';' read -ra VS <<< "${VARIANTS:- }"
for v in "${VS[@]}"; do
  echo "== variant: $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. $v -c tdec.hip -o /tmp/tdec_dev.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "tdec_pair_kernel" | grep -E "VGPRs|Scratch" | tr '\n' ' '; echo
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so $(ls build/*.o | grep -v tdec.o) /tmp/tdec_dev.o || exit 1
  (cd ../.. && bash scripts/quick_bench.sh)
done
cp /tmp/lib_keep.so libsrslte_phy_hip.so
