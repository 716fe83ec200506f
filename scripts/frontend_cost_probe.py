#!/usr/bin/env python3
"""What each front-end kernel costs the decoder when it runs beside it: three pipeline objects run ONLY the turbo decoder back to back on
three streams (steady state, the chip's wavefront slots full); a fourth object on a fourth stream runs ONE other stage once per decoder
step. Printed: microseconds per decoder step without company, and with each stage as company (the difference is what one launch of that
stage costs the pipeline, to be compared with its duration alone).
  python scripts/frontend_cost_probe.py [batch]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ND = 3
cfg = DlConfig(100, 1, 3, 75376)
rng = np.random.default_rng(0)
base = np.stack([make_subframe(cfg, b, rng, snr_db=18.0, amp=0.1)[0] for b in range(10)])
iq = torch.from_numpy(np.tile(base, ((B + 9) // 10, 1))[:B].copy()).cuda()
hc = hp.ChestDlCfg()
hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
rxs = [hp.DlRx(1, 100, 1, 0x1234, 3, 75376, 6, B, True, hc) for _ in range(ND + 1)]
tst = [torch.cuda.Stream() for _ in range(ND + 1)]
for s in range(ND + 1):
    assert rxs[s].run_device(iq.data_ptr(), 0, B, tst[s].cuda_stream) == 0
torch.cuda.synchronize()
names = {0: "ofdm_rx", 1: "chest_dl", 2: "pdsch_demod", 3: "rm_rx", 5: "tb_crc"}


def alone(stage, n=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        assert rxs[ND].stage(stage, iq.data_ptr(), 0, B, tst[ND].cuda_stream) == 0
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


def run(company, per_step=1, K=240):
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            s = k % ND
            assert rxs[s].stage(4, iq.data_ptr(), 0, B, tst[s].cuda_stream) == 0
            for st in company:
                for _ in range(per_step):
                    assert rxs[ND].stage(st, iq.data_ptr(), 0, B, tst[ND].cuda_stream) == 0
        torch.cuda.synchronize()
        dt = 1e6 * (time.perf_counter() - t0) / K
        best = dt if best is None else min(best, dt)
    return best


t_base = run(())
print("batch %d, decoder only on %d streams: %.1f us per step" % (B, ND, t_base))
for st in (0, 1, 2, 3, 5):
    t = run((st,))
    print("  + %-12s once per step on a fourth stream: %.1f us per step (+%.1f; the kernel alone, back to back: %.1f us)" % (names[st], t, t - t_base, alone(st)))
t = run((0, 1, 2, 3, 5))
print("  + all five in order once per step:            %.1f us per step (+%.1f)" % (t, t - t_base))
