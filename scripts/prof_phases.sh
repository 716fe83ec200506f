# per-phase cycle shares of the windowed turbo kernel (diagnostic -DTDEC_PROF build, run on the GPU box)
set -e
cd $GRAFT_REPO_ROOT/srslte-emane_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -DTDEC_PROF -c tdec.hip -o build/tdec.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so build/*.o
cd $GRAFT_REPO_ROOT
SRSLTE_HIP_TDEC_PROF=1 python scripts/tdec_sat.py --streams 1 --reps 2 2>&1 | tail -3
SRSLTE_HIP_TDEC_PROF=1 python scripts/tdec_sat.py --streams 3 --reps 6 2>&1 | tail -4
