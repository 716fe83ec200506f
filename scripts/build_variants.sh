#!/bin/bash
# Build variants of the product library HERE (hipcc cross-compiles) into gpurun_ab/lib<NAME>.so for a same-box A/B on the GPU box
# (scripts/ab_bench.sh with AB_VARIANTS="A B ..."). One translation unit is rebuilt with extra flags, the rest comes from csrc/build/.
#   bash scripts/build_variants.sh tdec.hip "A:-DP_XCHG_INT=0 -DP_CK_LDS=0" "B:-DP_CK_LDS=0" "D:"
set -e
cd "$(dirname "$0")/../srslte-emane_amd/csrc"
make -s -j8 libsrslte_phy_hip.so
src=$1; shift
obj=build/$(basename ${src%.*}).o
mkdir -p ../../gpurun_ab
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. $flags -x hip -c $src -o /tmp/var_$name.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A10 "${KERNEL:-tdec_pair_kernel}" | grep -E "VGPRs:|ScratchSize|LDS Size" | sed 's/.*remark: *//' | tr '\n' ' '
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/lib$name.so $(ls build/*.o | grep -v "^$obj$") /tmp/var_$name.o
    echo "[$name] $flags" ) &
done
wait
ls -la ../../gpurun_ab/
