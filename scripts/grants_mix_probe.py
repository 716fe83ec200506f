import sys, time, importlib, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/scripts')
import numpy as np, torch
import bench_grants_mix as m
pkg = importlib.import_module("srslte-emane_amd"); L = pkg.lib()
B=128; sub=m.build(B,4242); iq=m.noisy(sub,np.random.default_rng(1))
dev=torch.device("cuda",0); d_iq=torch.from_numpy(iq.view(np.float32)).to(dev)
hc=pkg.ChestDlCfg(); hc.filter_coef[0],hc.filter_coef[1]=4.0,1.0
grants=(pkg.DlGrant*B)(*[pkg.DlGrant.make(100,s["cfg"].mod,s["cfg"].tbs,0x1234,cfi=1,prb_mask=s["cfg"].prb_mask) for s in sub])
for ns in (1,4,8):
    rxs=[pkg.DlRx(1,100,1,0x1234,3,75376,6,B,True,hc) for _ in range(ns)]
    sts=[torch.cuda.Stream() for _ in range(ns)]
    def call(k):
        s=k%ns
        t0=time.perf_counter()
        rc=L.srslte_hip_dl_rx_batch_grants(rxs[s].h,d_iq.data_ptr(),0,B,grants,rxs[s].d_tb.ptr,rxs[s].tb_stride,rxs[s].d_ok.ptr,sts[s].cuda_stream)
        assert rc==0
        return time.perf_counter()-t0
    for k in range(2*ns): call(k)
    torch.cuda.synchronize()
    K=100; t0=time.perf_counter(); host=0
    for k in range(K): host+=call(k)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("streams %d: %.1f us per step, host time in the call %.1f us" % (ns, 1e6*dt/K, 1e6*host/K))
    for r in rxs: r.free()
