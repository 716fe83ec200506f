set -e
cd $GRAFT_REPO_ROOT/srslte-emane_amd/csrc
for w in 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -DTDEC_WAVES=$w -c tdec.hip -o build/tdec.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so build/*.o
  cd $GRAFT_REPO_ROOT
  for s in 1 3 5; do echo -n "waves $w "; python scripts/tdec_sat.py --streams $s; done
  cd $GRAFT_REPO_ROOT/srslte-emane_amd/csrc
done
