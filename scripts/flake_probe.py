#!/usr/bin/env python3
"""Development probe: the mixed two-layer grants batch of tests/test_gpu_mimo.py decoded REPS times from the same grids on one object; reports
which repetitions differ from the first in LLRs (both codewords), per-block pass counts, CRC flags and bytes."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
hp = importlib.import_module("srslte-emane_amd")
import test_gpu_mimo as T
from lte_sim import DlConfig, make_subframe, make_subframe_mimo

P, cid, tti0, csi = 25, 7, 4, True
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(40 * P + tti0)
items = []
for b, (kind, how, n, mod, tbs, mod2, tbs2, pmi, cfi, snr) in enumerate(T.MIXED[P]):
    mask, rnti = T._mask(P, rng, how, n), 0x200 + 3 * b
    kw = dict(cfi=cfi, rnti=rnti, nof_rx=2, nof_ports=2, csi=csi, prb_mask=mask)
    if kind == "div":
        cfg = DlConfig(P, cid, mod, tbs, **kw)
        iq, data = make_subframe(cfg, tti0 + b, rng, snr_db=snr - 2.0, amp=0.2)
    else:
        cfg = DlConfig(P, cid, mod, tbs, tx_scheme="cdd" if kind == "cdd" else "mux", pmi=pmi, mod2=mod2 or None, tbs2=tbs2, **kw)
        iq, data = make_subframe_mimo(cfg, tti0 + b, rng, snr_db=snr - 2.0, amp=0.2)
    g = hp.DlGrant2(hp.DlGrant.make(P, mod, tbs, rnti, cfi=cfi, prb_mask=mask), {"div": 1, "cdd": 3, "mux": 2, "mux1": 2}[kind], pmi, mod2, tbs2, 0, 1)
    items.append((kind, cfg, iq, g))
n = len(items)
tbs_max = max(max(c.tbss) if c.tx_scheme else c.tbs for _, c, _, _ in items)
rx = hp.DlRx(cid, P, 1, 0, 1, tbs_max, 6, n, True, T._chest(hp), nof_rx=2, nof_ports=2, csi=csi)
max_bits = 16 * ((14 * 12 * P * 8 + 15) // 16)
ref = None
bad = 0
for r in range(reps):
    if r == 0:
        rc, tb, ok = rx.decode_grants2(np.stack([it[2] for it in items]), tti0, [it[3] for it in items])
        grid = rx.debug(0, np.complex64, n * 2 * 14 * 12 * P).reshape(n, -1)
    else:
        rc, tb, ok = rx.decode_grants2(grid, tti0, [it[3] for it in items], from_grid=True)
    e = rx.debug(11, np.int16, 2 * n * max_bits).reshape(2 * n, -1).copy()
    cur = (e, [o.copy() for o in ok], [t.copy() for t in tb])
    if ref is None:
        ref = cur
        continue
    de = np.flatnonzero((cur[0] != ref[0]).any(axis=1))
    dok = [np.flatnonzero(cur[1][c] != ref[1][c]).tolist() for c in range(2)]
    dtb = [np.flatnonzero((cur[2][c] != ref[2][c]).any(axis=1)).tolist() for c in range(2)]
    if len(de) or any(dok) or any(dtb):
        bad += 1
        print("rep %d: LLR rows differing %s, flags differing %s, byte rows differing %s" % (r, de.tolist(), dok, dtb), flush=True)
print("%d of %d repetitions differ from the first; flags of the first: %s" % (bad, reps - 1, [o.tolist() for o in ref[1]]))
