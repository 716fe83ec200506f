#!/usr/bin/env python3
"""Decoder-only throughput of the 8-bit API (srslte_tdec_run_all_8bit path) at block lengths the sse8 back-end serves (800 < K <= 2048) and, for
comparison, one avx8 length: `nblk` noisy code blocks resident on the device, CRC early stop after every pass, `streams` decoder objects on as many
HIP streams. Prints code blocks/s and Mbit/s per (K, streams). A same-box A/B of two library builds: scripts/ab_bench.sh style, see
profiles/r04/ab_sse8_pair.txt.

    python scripts/bench_tdec_8bit.py [--nblk 1664] [--reps 20]"""
import argparse
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (ROOT, os.path.join(ROOT, "tests")):
    if d not in sys.path:
        sys.path.insert(0, d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nblk", type=int, default=1664)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--snr", type=float, default=1.5)
    a = ap.parse_args()
    import torch  # noqa: F401  (device runtime first)
    from _libs import oracle, p
    hp = importlib.import_module("srslte-emane_amd")
    L = hp.lib()
    out = {}
    for K in (1024, 1536, 2048, 4096):
        rng = np.random.default_rng(K)
        base = 16  # distinct code words, tiled over the batch
        w = np.zeros((base, 3 * K + 12), np.int8)
        for i in range(base):
            payload = rng.integers(0, 256, (K - 24) // 8, dtype=np.uint8)
            crc = oracle().orc_crc_bytes(0x1800063, 24, p(payload), K - 24)
            bits = np.unpackbits(np.concatenate([payload, np.array([crc >> 16, (crc >> 8) & 255, crc & 255], np.uint8)]))
            enc = np.zeros(3 * K + 12, np.uint8)
            oracle().orc_tcod_encode_bits(p(bits), p(enc), K)
            w[i] = (20 * ((2.0 * enc - 1) + 10 ** (-(a.snr - 3.0 + 0.3 * (i % 5)) / 20) * rng.standard_normal(3 * K + 12))).clip(-128, 127).astype(np.int8)
        x = np.ascontiguousarray(np.tile(w, (a.nblk // base + 1, 1))[:a.nblk])
        for ns in (1, 4):
            decs = [hp.Tdec(K, a.nblk) for _ in range(ns)]
            din = [hp.DevBuf.from_host(x) for _ in range(ns)]
            dout, dit, dok = [hp.DevBuf(a.nblk * (K // 8)) for _ in range(ns)], [hp.DevBuf(4 * a.nblk) for _ in range(ns)], [hp.DevBuf(a.nblk) for _ in range(ns)]
            streams = [torch.cuda.Stream() for _ in range(ns)]

            def launch(s):
                rc = L.srslte_hip_tdec_run_batch_8bit(decs[s].h, din[s].ptr, x.shape[1], 0, K, a.nblk, 6, hp.CRC24B, K, dout[s].ptr, K // 8, dit[s].ptr, dok[s].ptr,
                                                      C.c_void_p(streams[s].cuda_stream))
                assert rc == 0
            for s in range(ns):
                launch(s)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                for r in range(a.reps):
                    launch(r % ns)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            dt = float(np.median(ts))
            iters = dit[0].to_host(np.uint32)
            okf = dok[0].to_host(np.uint8)
            out[(K, ns)] = (a.reps * a.nblk / dt, float(iters.mean()), float(okf.mean()))
            print("K %d streams %d: %.0f blocks/s, %.1f Mbit/s, %.2f passes per block, %.2f delivered" % (K, ns, out[(K, ns)][0], out[(K, ns)][0] * (K - 24) / 1e6,
                                                                                                      out[(K, ns)][1], out[(K, ns)][2]), flush=True)
            for d_ in decs:
                d_.free()


if __name__ == "__main__":
    main()
