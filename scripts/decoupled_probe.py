#!/usr/bin/env python3
"""Upper bound for any re-arrangement of streams and events around the pipeline's kernels: per step the same launches as the pipeline
(stages 0-3 of one object, then stages 4 and 5 of another), but the front end on streams of its own and NO dependency between it and the
decoder launches (the decoders re-decode the soft buffers of the warm-up call). Run with GPU_MAX_HW_QUEUES >= the number of streams.
  python scripts/decoupled_probe.py <decoder streams> <front-end streams>      -> profiles/r03/frontend_cost_probe.txt"""
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

B = 128
ND, NF = int(sys.argv[1]), int(sys.argv[2])
cfg = DlConfig(100, 1, 3, 75376)
rng = np.random.default_rng(0)
base = np.stack([make_subframe(cfg, b, rng, snr_db=18.0, amp=0.1)[0] for b in range(10)])
iq = torch.from_numpy(np.tile(base, ((B + 9) // 10, 1))[:B].copy()).cuda()
hc = hp.ChestDlCfg()
hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
rxs = [hp.DlRx(1, 100, 1, 0x1234, 3, 75376, 6, B, True, hc) for _ in range(ND + NF)]
tst = [torch.cuda.Stream() for _ in range(ND + NF)]
for s in range(ND + NF):
    assert rxs[s].run_device(iq.data_ptr(), 0, B, tst[s].cuda_stream) == 0
torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize()
    t0, K = time.perf_counter(), 240
    for k in range(K):
        s = k % ND
        f = ND + k % NF
        for st in (0, 1, 2, 3):
            assert rxs[f].stage(st, iq.data_ptr(), 0, B, tst[f].cuda_stream) == 0
        assert rxs[s].stage(4, iq.data_ptr(), 0, B, tst[s].cuda_stream) == 0
        assert rxs[s].stage(5, iq.data_ptr(), 0, B, tst[s].cuda_stream) == 0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("decoder on %d streams, front end (0-3) on %d others, no dependencies: %.1f us per step" % (ND, NF, 1e6 * dt / K))
