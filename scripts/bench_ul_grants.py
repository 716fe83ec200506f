#!/usr/bin/env python3
"""Throughput of srslte_hip_ul_rx_batch_grants on one MI355X: a batch of 128 subframes of a 100-PRB cell, every subframe carrying 4 PUSCHs
(24 PRB each, 16QAM, TBS 9144, own RNTI / cyclic shift) or 1 PUSCH of 96 PRB, inputs resident in HBM, one HIP stream, noise free (one SISO pass
per block). The stimulus comes from the device's own PUSCH transmit pipeline (one object per UE), summed on the host.

  python scripts/bench_ul_grants.py [--steps 20]
Prints one JSON object. A record for profiles/, not the bench contract's line."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the product has no CPU path")
    pkg = importlib.import_module("srslte-emane_amd")
    prb, B, out = 100, 128, {}
    rng = np.random.default_rng(5)
    for name, ues in (("4 x 24 PRB", [(24, 24 * u, 2, 9144, u) for u in range(4)]), ("1 x 96 PRB", [(96, 2, 2, 36696, 1)])):
        iq = np.zeros((B, 15 * 1536), np.complex64)
        datas = []
        for (L, n0, mod, tbs, nd) in ues:
            tx = pkg.UlTx(3, prb, 0x100 + nd, mod, tbs, L, n0, nd, B)
            d = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
            iq += tx.encode(d, 0)
            datas.append(d)
            tx.free()
        grants = [pkg.UlGrant.make(b, 0x100 + nd, L, n0, mod, tbs, n_dmrs=nd) for b in range(B) for (L, n0, mod, tbs, nd) in ues]
        rx = pkg.UlRx(3, prb, 0x77, 2, max(u[3] for u in ues), 6, 0, 0, 6, B, max_grants=len(grants))
        tb, ok = rx.decode_grants(iq, 0, grants)
        good = bool(ok.all()) and all(np.array_equal(tb[i::len(ues), :ues[i][3] // 8], datas[i]) for i in range(len(ues)))
        din = pkg.DevBuf.from_host(iq)
        arr = (pkg.UlGrant * len(grants))(*grants)
        L_ = pkg.lib()

        def step():
            rc = L_.srslte_hip_ul_rx_batch_grants(rx.h, din.ptr, 0, B, arr, len(grants), rx.d_tb.ptr, rx.tb_stride, rx.d_ok.ptr, None)
            assert rc == 0
        for _ in range(3):
            step()
        pkg.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_host = time.perf_counter() - t0
        pkg.sync()
        dt = (time.perf_counter() - t0) / args.steps
        out[name] = {"pusch_per_s": round(len(grants) / dt), "subframes_per_s": round(B / dt), "ms_per_batch": round(dt * 1e3, 3),
                     "host_ms_per_call": round(t_host / args.steps * 1e3, 3), "all_decoded": good}
        rx.free()
    out["_about"] = "scripts/bench_ul_grants.py: 128 subframes of a 100-PRB cell per batch, 16QAM, noise free, one HIP stream, inputs and results on the device"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
