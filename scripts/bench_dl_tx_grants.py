#!/usr/bin/env python3
"""srslte_hip_dl_tx_batch_grants on one stream: 128 subframes of a 100-PRB cell, one full-band 64QAM PDSCH each (TBS 75376), payloads resident on the
device; ms per call and subframes/s. (For A/B runs of changes to the transmit side's grants mode: scripts/ab_script.sh with AB_CMD.)"""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    hp = importlib.import_module("srslte-emane_amd")
    L = hp.lib()
    P, B, tbs, steps = 100, 128, 75376, 40
    tx = hp.DlTx(1, P, 1, 0x1234, 3, tbs, B)

    class TxGrant(C.Structure):
        _fields_ = [("sf", C.c_uint32), ("grant", hp.DlGrant)]
    arr = (TxGrant * B)(*[TxGrant(b, hp.DlGrant.make(P, 3, tbs, 0x100 + b, cfi=1)) for b in range(B)])
    stride = (tbs // 8 + 15) & ~15
    din = hp.DevBuf.from_host(np.random.default_rng(1).integers(0, 256, (B, stride), dtype=np.uint8))
    L.srslte_hip_dl_tx_batch_grants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]

    def step():
        assert L.srslte_hip_dl_tx_batch_grants(tx.h, din.ptr, stride, 0, B, arr, B, tx.d_iq.ptr, None) == 0
    for _ in range(5):
        step()
    hp.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        hp.sync()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(json.dumps({"dl_tx_grants_ms_per_call": round(best * 1e3, 4), "subframes_per_s": round(B / best)}))


if __name__ == "__main__":
    main()
