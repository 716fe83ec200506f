#!/bin/bash
# Build the WHOLE product library with extra flags into gpurun_ab/lib<NAME>.so (for macros that several translation units read):
#   bash scripts/build_full_variant.sh "P3:-DSRSLTE_HIP_FE_PRIO=3" "P0:"
set -e
cd "$(dirname "$0")/../srslte-emane_amd/csrc"
mkdir -p ../../gpurun_ab
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  d=/tmp/fullvar_$name; rm -rf $d; mkdir -p $d
  for f in fft.hip demod.hip chest.hip tdec.hip tdec_mix.hip tcod.hip pdsch.hip api.cpp fec_tables.cpp compat.cpp compat_refsignal.cpp; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -I../../include -I. $flags -x hip -c $f -o $d/${f%.*}.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/lib$name.so $d/*.o
  echo "[$name] $flags"
done
ls -la ../../gpurun_ab/
