#!/bin/bash
# Same-box A/B of library builds (gpurun_ab/lib<NAME>.so, AB_VARIANTS) over an arbitrary command (AB_CMD), two rounds; prints the command's stdout per run.
cd "$(dirname "$0")/.."
cp srslte-emane_amd/csrc/libsrslte_phy_hip.so /tmp/lib_keep.so
for round in 1 2; do
  for v in ${AB_VARIANTS:-A B}; do
    cp gpurun_ab/lib$v.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
    echo "== $v (round $round)"
    $AB_CMD 2>/dev/null
  done
done
cp /tmp/lib_keep.so srslte-emane_amd/csrc/libsrslte_phy_hip.so
