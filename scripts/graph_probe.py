#!/usr/bin/env python3
"""Development probe: one pipeline object, one stream, batch 128 of the headline workload - the six stage launches issued call by call against
the same chain replayed from a hipGraph (captured once through torch.cuda.graph on the stream the stages are launched on).
  python scripts/graph_probe.py [steps]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("srslte-emane_amd")
from lte_sim import DlConfig, make_subframe

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prb, mod, tbs, B = 100, 3, 75376, 128
cfg = DlConfig(prb, 1, mod, tbs)
rng = np.random.default_rng(1)
base = [make_subframe(cfg, t, rng, snr_db=18.0, amp=0.1)[0] for t in range(16)]
iq = np.stack([base[b % 16] for b in range(B)])  # TTI b of the batch uses TTI (b % 16)'s samples: the pass statistics matter here, not the payload
hc = pkg.ChestDlCfg()
hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
rx = pkg.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, B, True, hc)
d_iq = torch.from_numpy(iq.view(np.float32)).cuda()
s = torch.cuda.Stream()


def chain():
    for st in range(6):
        assert rx.stage(st, d_iq.data_ptr(), 0, B, s.cuda_stream) == 0


for _ in range(5):
    chain()
torch.cuda.synchronize()


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


t_plain = timed(chain)
ok_plain = rx.d_ok.to_host(np.uint8)[:B].copy()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    chain()
t_graph = timed(g.replay)
ok_graph = rx.d_ok.to_host(np.uint8)[:B]
print("call by call: %.1f us per step (%.0f subframes/s); hipGraph replay: %.1f us per step (%.0f subframes/s); CRC flags identical: %s, %d of %d ok" %
      (1e6 * t_plain, B / t_plain, 1e6 * t_graph, B / t_graph, bool(np.array_equal(ok_plain, ok_graph)), int(ok_graph.sum()), B))
