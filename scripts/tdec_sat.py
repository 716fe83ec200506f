#!/usr/bin/env python3
"""Development probe: saturated throughput of ONE pipeline stage (default 4 = turbo decoder) launched round-robin on N streams
over N pipeline instances, after a full pass has filled the buffers. Usage: python scripts/tdec_sat.py [--streams N] [--stage S]"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from lte_sim import DlConfig, make_subframe

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--stage", type=int, default=4)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--snr", type=float, default=18.0)
ap.add_argument("--llr8", action="store_true")
a = ap.parse_args()
pkg = importlib.import_module("srslte-emane_amd")
torch.cuda.set_device(0)
cfg = DlConfig(100, 1, 3, 75376, llr8=a.llr8)
rng = np.random.default_rng(1000)
iq = np.stack([make_subframe(cfg, t, rng, snr_db=a.snr, amp=0.1)[0] for t in range(a.batch)])
d_iq = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
hc = pkg.ChestDlCfg(); hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
rxs = [pkg.DlRx(1, 100, 1, 0x1234, 3, 75376, 6, a.batch, True, hc, llr_8bit=a.llr8) for _ in range(a.streams)]
ts = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(a.streams - 1)]
for i, rx in enumerate(rxs):
    for s in range(6):
        assert rx.stage(s, d_iq.data_ptr(), 0, a.batch, ts[i].cuda_stream) == 0
torch.cuda.synchronize()
for k in range(a.streams * 2):
    rxs[k % a.streams].stage(a.stage, d_iq.data_ptr(), 0, a.batch, ts[k % a.streams].cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(a.reps):
    rxs[k % a.streams].stage(a.stage, d_iq.data_ptr(), 0, a.batch, ts[k % a.streams].cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
print("stage %d streams %d: %.4f ms per launch" % (a.stage, a.streams, dt * 1e3))
