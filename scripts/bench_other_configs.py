#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations other than the headline one (those are parity-test cases, not bench lines; this is a
record for profiles/): device-resident inputs, one HIP stream, whole batches per call, results left on the device.
  cfg3  20 MHz uplink, batch 128: PUSCH transmit pipeline (CRC, turbo encoder, rate matching, interleaver, DFT precoding, DMRS, OFDM TX)
        and the eNB-side receive pipeline on its output
  cfg5  20 MHz downlink, 256QAM, batch 512: receive pipeline (16-bit and 8-bit LLRs)
  TM3   20 MHz downlink, 2 x 64QAM transport blocks (large-delay CDD), 2 antennas, batch 128
  python scripts/bench_other_configs.py > profiles/r02/other_configs.json"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
hp = importlib.import_module("srslte-emane_amd")
L = hp.lib()
rng = np.random.default_rng(0)
out = {}


def timed(fn, reps=30):
    fn()
    hp.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    hp.sync()
    return (time.perf_counter() - t0) / reps


def chest():
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    return hc


# ---- cfg3
prb, Lp, mod, tbs, B = 100, 96, 2, 36696, 128
tx, rx = hp.UlTx(3, prb, 0x77, mod, tbs, Lp, 2, 1, B), hp.UlRx(3, prb, 0x77, mod, tbs, Lp, 2, 1, 6, B)
data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
din = hp.DevBuf.from_host(data)
t_tx = timed(lambda: L.srslte_hip_ul_tx_batch(tx.h, din.ptr, tbs // 8, 3, B, tx.d_iq.ptr, None))
t_rx = timed(lambda: L.srslte_hip_ul_rx_batch(rx.h, tx.d_iq.ptr, 3, B, rx.d_tb.ptr, rx.tb_stride, rx.d_ok.ptr, None))
ok = rx.d_ok.to_host(np.uint8)[:B]
out["cfg3_ul"] = {"workload": "20 MHz uplink, 96-PRB grant, 16QAM, TBS 36696 (6 x K=6144), batch 128, noise free (one SISO pass per block)",
                  "tx_subframes_per_s": round(B / t_tx), "rx_subframes_per_s": round(B / t_rx), "all_decoded": bool(ok.all())}
tx.free()
rx.free()

# ---- cfg5
prb, mod, tbs, B = 100, 4, 97896, 512
txd = hp.DlTx(2, prb, 1, 0x4321, mod, tbs, B)
data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
iq = txd.encode(data, 0)[:, 0, :]
sigma = np.sqrt(np.mean(np.abs(iq) ** 2) / 2) * 10 ** (-24.0 / 20)
d_iq = hp.DevBuf.from_host((iq + sigma * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64))
txd.free()
for llr8 in (False, True):
    rxd = hp.DlRx(2, prb, 1, 0x4321, mod, tbs, 6, B, True, chest(), llr_8bit=llr8)
    t = timed(lambda: rxd.run_device(d_iq.ptr, 0, B), 10)
    ok = rxd.d_ok.to_host(np.uint8)[:B]
    it = rxd.debug(6, np.uint32, B * 16)
    out["cfg5_256qam_llr%d" % (8 if llr8 else 16)] = {"workload": "20 MHz downlink, 256QAM, TBS 97896 (16 x K=6144), batch 512, AWGN 24 dB", "subframes_per_s": round(B / t),
                                                       "bler": round(1 - float(ok.mean()), 4), "avg_siso_passes_per_cb": round(float(it.mean()), 3)}
    rxd.free()

# ---- TM3
from lte_sim import DlConfig, make_subframe_mimo  # noqa: E402

prb, mod, tbs, B = 100, 3, 75376, 128
cfg = DlConfig(prb, 2, mod, tbs, nof_rx=2, nof_ports=2, tx_scheme="cdd", mod2=mod, tbs2=tbs)
base = np.stack([make_subframe_mimo(cfg, b, rng, snr_db=27.0, amp=0.2)[0] for b in range(10)])
d_iq = hp.DevBuf.from_host(np.ascontiguousarray(np.tile(base, (13, 1, 1))[:B]))
rxm = hp.DlRx(2, prb, 1, 0x1234, mod, tbs, 6, B, True, chest(), nof_rx=2, nof_ports=2, tx_scheme=3, mod2=mod, tbs2=tbs)
t = timed(lambda: rxm.run_device(d_iq.ptr, 0, B), 10)
ok = rxm.d_ok.to_host(np.uint8)[:2 * B]
out["tm3_cdd"] = {"workload": "20 MHz downlink, large-delay CDD, 2 x (64QAM, TBS 75376) per subframe, 2 receive antennas, batch 128, AWGN 27 dB",
                  "subframes_per_s": round(B / t), "transport_blocks_per_s": round(2 * B / t), "Mbit_per_s": round(2 * B * tbs / t / 1e6), "bler": round(1 - float(ok.mean()), 4)}
rxm.free()
print(json.dumps(out, indent=1))
