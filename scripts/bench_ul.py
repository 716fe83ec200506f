#!/usr/bin/env python3
"""cfg3: 20 MHz PUSCH, L_prb = 100, batch 128 - device transmit chain (UE side) and device receive chain (eNB side).

  python scripts/bench_ul.py [--mod 2|3] [--snr dB] [--steps K]

Prints one JSON line: subframes/s of srslte_hip_ul_tx_batch and of srslte_hip_ul_rx_batch (inputs resident in HBM), BLER and SISO
passes of the receive side on the transmit side's own output plus AWGN. Not the headline metric (bench.py is); a side measurement."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mod", type=int, default=2, help="2 = 16QAM MCS 20 (TBS 43816), 3 = 64QAM MCS 28 (TBS 75376)")
    ap.add_argument("--snr", type=float, default=None)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    pkg = importlib.import_module("srslte-emane_amd")
    L = pkg.lib()
    prb, Lp, B = 100, 100, args.batch
    tbs = {2: 43816, 3: 75376}[args.mod]
    snr = args.snr if args.snr is not None else {2: 12.0, 3: 19.5}[args.mod]
    N = pkg.symbol_sz(prb)
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
    d_tb = torch.from_numpy(data).cuda()
    tx = pkg.UlTx(1, prb, 0x1234, args.mod, tbs, Lp, 0, 0, B)
    rx = pkg.UlRx(1, prb, 0x1234, args.mod, tbs, Lp, 0, 0, 6, B)
    d_iq = torch.empty(B, 15 * N * 2, device="cuda", dtype=torch.float32)

    def run_tx():
        rc = L.srslte_hip_ul_tx_batch(tx.h, d_tb.data_ptr(), tbs // 8, 0, B, d_iq.data_ptr(), None)
        assert rc == 0, rc

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    t_tx = timed(run_tx)
    # AWGN on the device: signal power per sample = M_sc / N with the 1/sqrt(N) normalisation
    sigma = float(np.sqrt(12 * Lp / N / 2) * 10 ** (-snr / 20))
    torch.manual_seed(1)
    d_rx = d_iq + sigma * torch.randn_like(d_iq)

    def run_rx():
        rc = L.srslte_hip_ul_rx_batch(rx.h, d_rx.data_ptr(), 0, B, rx.d_tb.ptr, rx.tb_stride, rx.d_ok.ptr, None)
        assert rc == 0, rc

    t_rx = timed(run_rx)
    ok = rx.d_ok.to_host(np.uint8)[:B]
    tb = rx.d_tb.to_host(np.uint8).reshape(B, rx.tb_stride)[:, :tbs // 8]
    good = int(sum(bool(ok[b]) and np.array_equal(tb[b], data[b]) for b in range(B)))
    wrong = int(sum(bool(ok[b]) and not np.array_equal(tb[b], data[b]) for b in range(B)))
    C_ = {43816: 8, 75376: 13}[tbs]
    iters = rx.debug(6, np.uint32, B * C_)
    print(json.dumps({"workload": "cfg3: 100 PRB PUSCH, L_prb 100, mod %d, TBS %d, batch %d" % (args.mod, tbs, B), "snr_db": snr,
                      "ul_tx_subframes_per_s": round(B / t_tx, 1), "ul_tx_ms_per_batch": round(t_tx * 1e3, 3),
                      "ul_rx_subframes_per_s": round(B / t_rx, 1), "ul_rx_ms_per_batch": round(t_rx * 1e3, 3),
                      "bler": round(1 - good / B, 4), "undetected_errors": wrong, "avg_siso_passes_per_cb": round(float(iters.mean()), 3)}))


if __name__ == "__main__":
    main()
