set -e
cd $GRAFT_REPO_ROOT/srslte-emane_amd/csrc
for u in 2 4 6; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -DTDEC_EWU=$u -Rpass-analysis=kernel-resource-usage -c tdec.hip -o build/tdec.o 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - | grep "ILi16ELi0" | sed 's/.*TdecArgsE//'
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so build/*.o
  cd $GRAFT_REPO_ROOT
  for s in 1 3; do echo -n "ewu $u "; python scripts/tdec_sat.py --streams $s; done
  echo -n "ewu $u e2e3: "; python bench.py --no-cpu --stream-batch 0 --streams 3 --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  cd $GRAFT_REPO_ROOT/srslte-emane_amd/csrc
done
