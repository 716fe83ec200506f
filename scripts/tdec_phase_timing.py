#!/usr/bin/env python3
"""Time the turbo decoder kernels alone on random LLRs (no early stop), for each value of the -DTDEC_DEBUG knob SRSLTE_HIP_TDEC_DBG:
16 = all passes, 17 = without the SISO sweeps, 18 = without the element-wise phases. Needs a library built with -DTDEC_DEBUG.
  SRSLTE_HIP_TDEC_DBG=16 python scripts/tdec_phase_timing.py <force_w> <ncb> [K]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
hp = importlib.import_module("srslte-emane_amd")
fw, ncb = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5824
rng = np.random.default_rng(0)
w = rng.integers(-60, 60, (ncb, 3 * (K + 32) + 12)).astype(np.int16)
dec = hp.Tdec(6144, ncb)
x = np.ascontiguousarray(w)
din, dout = hp.DevBuf.from_host(x), hp.DevBuf(ncb * (K // 8))
dit, dok = hp.DevBuf(4 * ncb), hp.DevBuf(ncb)
L = hp.lib()
ts = []
for rep in range(6):
    hp.sync()
    t0 = time.perf_counter()
    rc = L.srslte_hip_tdec_run_batch_manual(dec.h, din.ptr, x.shape[1], 1, K, fw, ncb, 6, hp.CRC24B, K, dout.ptr, K // 8, dit.ptr, dok.ptr, None)
    hp.sync()
    ts.append(time.perf_counter() - t0)
    assert rc == 0
print("force_w %d ncb %d dbg %s: %.3f ms (min of 5)" % (fw, ncb, os.environ.get("SRSLTE_HIP_TDEC_DBG"), 1e3 * min(ts[1:])))
