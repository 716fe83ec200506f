set -e
cd $GRAFT_REPO_ROOT/srslte-emane_amd/csrc
cp tdec.hip /tmp/tdec_new.hip
for v in old new old new; do
  if [ $v = old ]; then cp $GRAFT_REPO_ROOT/scripts/tdec_old.hip.txt tdec.hip; else cp /tmp/tdec_new.hip tdec.hip; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -c tdec.hip -o build/tdec.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsrslte_phy_hip.so build/*.o
  cd $GRAFT_REPO_ROOT
  for s in 1 3; do
    echo -n "$v streams $s: "; python bench.py --no-cpu --stream-batch 0 --streams $s --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels']['tdec']['ms'])"
  done
  echo -n "$v "; python scripts/tdec_sat.py --streams 3
  cd $GRAFT_REPO_ROOT/srslte-emane_amd/csrc
done
cp /tmp/tdec_new.hip tdec.hip
