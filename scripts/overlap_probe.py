#!/usr/bin/env python3
"""What the front-end kernels cost the decoder when batches overlap: N pipeline instances on N streams, steady state,
(a) all six stages per step, (b) only the turbo decoder + TB stage (stages 4, 5) per step, (c) only the decoder: back-to-back launches
of the one kernel on every stream, (d) only stages 0-3.
  python scripts/overlap_probe.py [streams] [batch]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
hp = importlib.import_module("srslte-emane_amd")
from lte_sim import DlConfig, make_subframe  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = DlConfig(100, 1, 3, 75376)
rng = np.random.default_rng(0)
base = np.stack([make_subframe(cfg, b, rng, snr_db=18.0, amp=0.1)[0] for b in range(10)])
iq = torch.from_numpy(np.tile(base, ((B + 9) // 10, 1))[:B].copy()).cuda()
hc = hp.ChestDlCfg()
hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
rxs = [hp.DlRx(1, 100, 1, 0x1234, 3, 75376, 6, B, True, hc) for _ in range(ns)]
tst = [torch.cuda.Stream() for _ in range(ns)]
for s in range(ns):
    assert rxs[s].run_device(iq.data_ptr(), 0, B, tst[s].cuda_stream) == 0
torch.cuda.synchronize()
for name, stages in (("all six stages", range(6)), ("all but the TB stage", range(5)), ("decoder + TB only", (4, 5)), ("decoder only (4)", (4,)), ("front end only (0-3)", range(4))):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        for k in range(K):
            s = k % ns
            for st in stages:
                assert rxs[s].stage(st, iq.data_ptr(), 0, B, tst[s].cuda_stream) == 0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%d streams x %d: %-22s %.1f us per step, %.0f subframes/s" % (ns, B, name, 1e6 * dt / K, K * B / dt))
