"""Stimulus generator and CPU receive chain built ONLY on the oracle (oracle/liboracle.so).

Test infrastructure (used by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke()): it builds seeded
synthetic DL subframes the way phy_dl_test.c:146-196 does (eNB side: TB -> CRC/segmentation/turbo/rate-matching ->
scrambling -> modulation -> RE mapping + CRS -> OFDM TX) and decodes them with the oracle's restatement of the UE
side (ue_dl.c:369-384 -> pdsch.c:833-997 -> sch.c:507-532). Nothing here is part of the product.
"""
import ctypes as C

import numpy as np

from _libs import OrcCbsegm, OrcCell, OrcChestCfg, OrcChestRes, OrcOfdm, OrcSchCfg, oracle, p

MOD_BITS = {0: 1, 1: 2, 2: 4, 3: 6, 4: 8}


class DlConfig:
    """One PDSCH configuration: single port, full-band grant, rv 0."""

    def __init__(self, nof_prb, cell_id, mod, tbs, cfi=1, rnti=0x1234, max_iter=6, chest=None):
        self.nof_prb, self.cell_id, self.mod, self.tbs, self.cfi, self.rnti, self.max_iter = nof_prb, cell_id, mod, tbs, cfi, rnti, max_iter
        self.Qm = MOD_BITS[mod]
        self.cell = OrcCell(cell_id, nof_prb, 1, True)
        self.nre = 12 * nof_prb
        self.grid_len = 14 * self.nre
        self.lstart = cfi + (1 if nof_prb < 10 else 0)
        self.N = oracle().orc_symbol_sz(nof_prb)
        self.sf_len = 15 * self.N
        self.chest = chest or {"filter_coef": (4.0, 1.0)}  # phy_dl_test.c:587-595
        self.seg = OrcCbsegm()
        assert oracle().orc_cbsegm(C.byref(self.seg), tbs) == 0 and self.seg.F == 0

    def indices(self, sf_idx):
        idx = np.zeros(self.grid_len, np.uint32)
        n = oracle().orc_pdsch_indices(C.byref(self.cell), sf_idx, self.lstart, None, p(idx))
        return idx[:n].copy()

    def orc_chest_cfg(self):
        c = OrcChestCfg()
        for k, v in self.chest.items():
            if k == "filter_coef":
                c.filter_coef[0], c.filter_coef[1] = v
            else:
                setattr(c, k, v)
        return c


def scramble_seq(cfg, sf_idx, nbits):
    c = np.zeros(nbits, np.uint8)
    oracle().orc_gold(C.c_uint32(oracle().orc_pdsch_cinit(cfg.rnti, 0, sf_idx, cfg.cell_id)), nbits, p(c))
    return c


def make_subframe(cfg, tti, rng, snr_db=None, amp=1.0):
    """Returns (iq[sf_len] complex64, payload bytes[tbs/8]) for TTI `tti`."""
    orc = oracle()
    sf_idx = tti % 10
    idx = cfg.indices(sf_idx)
    nbits = len(idx) * cfg.Qm
    data = rng.integers(0, 256, cfg.tbs // 8, dtype=np.uint8)
    sch = OrcSchCfg(cfg.tbs, nbits, cfg.Qm, 0, cfg.max_iter)
    e = np.zeros(nbits, np.uint8)
    assert orc.orc_dlsch_encode(C.byref(sch), p(data), p(e)) == 0
    e ^= scramble_seq(cfg, sf_idx, nbits)
    syms = np.zeros(len(idx), np.complex64)
    orc.orc_modulate(cfg.mod, p(e), p(syms), nbits)
    grid = np.zeros(cfg.grid_len, np.complex64)
    grid[idx] = syms
    orc.orc_crs_put_sf(C.byref(cfg.cell), sf_idx, 0, p(grid))
    q = OrcOfdm()
    orc.orc_ofdm_init(C.byref(q), cfg.nof_prb, True)
    q.normalize = True
    iq = np.zeros(cfg.sf_len, np.complex64)
    orc.orc_ofdm_tx_sf(C.byref(q), p(grid), p(iq))
    iq *= np.float32(amp)
    if snr_db is not None:
        # signal power per time sample with a normalised IFFT: nof_re/N per unit-power RE
        sigma = np.sqrt(amp * amp * cfg.nre / cfg.N / 2) * 10 ** (-snr_db / 20)
        iq = iq + (sigma * (rng.standard_normal(cfg.sf_len) + 1j * rng.standard_normal(cfg.sf_len))).astype(np.complex64)
    return iq.astype(np.complex64), data


def oracle_rx(cfg, iq, tti, keep=False):
    """Oracle UE receive chain for one subframe. Returns dict with tb bytes (tbs/8+3), ok flag and (keep=True) every intermediate."""
    orc = oracle()
    sf_idx = tti % 10
    q = OrcOfdm()
    orc.orc_ofdm_init(C.byref(q), cfg.nof_prb, True)
    grid = np.zeros(cfg.grid_len, np.complex64)
    orc.orc_ofdm_rx_sf(C.byref(q), p(np.ascontiguousarray(iq, np.complex64)), p(grid))
    ce = np.zeros(cfg.grid_len, np.complex64)
    res = OrcChestRes()
    ccfg = cfg.orc_chest_cfg()
    assert orc.orc_chest_dl(C.byref(cfg.cell), sf_idx, C.byref(ccfg), p(grid), p(ce), C.byref(res)) == 0
    idx = cfg.indices(sf_idx)
    y, h = np.ascontiguousarray(grid[idx]), np.ascontiguousarray(ce[idx])
    d = np.zeros(len(idx), np.complex64)
    orc.orc_predecoding_single(p(y), p(h), p(d), len(idx), 1.0, res.noise_estimate)
    nbits = len(idx) * cfg.Qm
    e = np.zeros(nbits, np.int16)
    orc.orc_demod_soft_s(cfg.mod, p(d), p(e), len(idx))
    orc.orc_scramble_s(p(e), p(scramble_seq(cfg, sf_idx, nbits)), nbits)
    sch = OrcSchCfg(cfg.tbs, nbits, cfg.Qm, 0, cfg.max_iter)
    tb = np.zeros(cfg.tbs // 8 + 16, np.uint8)
    iters = np.zeros(cfg.seg.C, np.uint32)
    cbok = np.zeros(cfg.seg.C, np.uint8)
    rc = orc.orc_dlsch_decode(C.byref(sch), p(e), p(tb), p(iters), p(cbok))
    out = {"tb": tb[:cfg.tbs // 8 + 3], "ok": rc == 0, "iters": iters, "cb_ok": cbok}
    if keep:
        out.update(grid=grid, ce=ce, noise=res.noise_estimate, d=d, e=e, res=res)
    return out
