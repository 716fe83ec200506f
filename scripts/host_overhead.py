#!/usr/bin/env python3
"""Is bench.py's timed loop limited by the host? Same workload, three ways of submitting it: per-stage calls with the two event records
(what bench.py times), per-stage calls without events, one srslte_hip_dl_rx_batch call per step. Prints ms per step for each."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    pkg = importlib.import_module("srslte-emane_amd")
    L = pkg.lib()
    from lte_sim import DlConfig, make_subframe
    B, nstreams, steps = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 3, 60
    cfg = DlConfig(100, 1, 3, 75376)
    rng = np.random.default_rng(1000)
    iq = np.stack([make_subframe(cfg, t, rng, snr_db=18.0, amp=0.1)[0] for t in range(B)])
    d_iq = torch.from_numpy(iq.view(np.float32)).cuda()
    hc = pkg.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rxs = [pkg.DlRx(1, 100, 1, 0x1234, 3, 75376, 6, B, True, hc) for _ in range(nstreams)]
    ts = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    st = [t.cuda_stream for t in ts]
    ev = [(L.srslte_hip_event_create(), L.srslte_hip_event_create()) for _ in range(steps)]

    def staged(k, events):
        for s in range(6):
            if events and s == 4:
                L.srslte_hip_event_record(ev[k][0], st[k % nstreams])
            rxs[k % nstreams].stage(s, d_iq.data_ptr(), 0, B, st[k % nstreams])
            if events and s == 4:
                L.srslte_hip_event_record(ev[k][1], st[k % nstreams])

    def whole(k):
        rxs[k % nstreams].run_device(d_iq.data_ptr(), 0, B, st[k % nstreams])

    for name, fn in (("staged+events", lambda k: staged(k, True)), ("staged", lambda k: staged(k, False)), ("one call", whole)):
        for k in range(6):
            fn(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            fn(k)
        t_submit = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print("%-14s %.3f ms/step (%.0f subframes/s), host submission %.3f ms/step" % (name, t / steps * 1e3, B * steps / t, t_submit / steps * 1e3))


if __name__ == "__main__":
    main()
