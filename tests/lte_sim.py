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
    """One PDSCH configuration: full-band grant, rv 0; nof_ports = 1: single antenna port (TM1),
    nof_ports = 2: 2-port transmit diversity (TM2)."""

    def __init__(self, nof_prb, cell_id, mod, tbs, cfi=1, rnti=0x1234, max_iter=6, chest=None, llr8=False, nof_rx=1, nof_ports=1, csi=False, p_a=None,
                 prb_mask=None, tx_scheme=None, pmi=0, mod2=None, tbs2=0, cp_ext=False, tdd=None):
        self.nof_prb, self.cell_id, self.mod, self.tbs, self.cfi, self.rnti, self.max_iter = nof_prb, cell_id, mod, tbs, cfi, rnti, max_iter
        # extended-CP cell: 6 symbols per slot, CRS on symbols 0 and 3, PSS / SSS on symbols 5 and 4 of slot 0 (phy_common.h:101-141, pdsch.c:81-206)
        self.cp_ext, self.cp_norm, self.nsym = bool(cp_ext), not cp_ext, 12 if cp_ext else 14
        # two-layer modes of a 2-port cell with 2 receive antennas: tx_scheme "cdd" (TM3, two transport blocks) or "mux"
        # (TM4: two transport blocks with pmi 0/1, or one - tbs2 = 0 - with pmi 0..3); srslte_pdsch_grant_t.tx_scheme / pmi / tb[1]
        self.tx_scheme, self.pmi = tx_scheme, pmi
        self.mods, self.tbss = [mod] + ([mod2 if mod2 is not None else mod] if tbs2 else []), [tbs] + ([tbs2] if tbs2 else [])
        self.nof_tb = len(self.tbss)
        if tx_scheme is not None:
            assert nof_ports == 2 and nof_rx == 2 and tx_scheme in ("cdd", "mux") and (tx_scheme == "mux" or self.nof_tb == 2)
            self.codebook_idx = pmi if self.nof_tb == 1 else pmi + 1  # pdsch.c:914
        # srslte_pdsch_grant_t.prb_idx[s][n] (pdsch_cfg.h:41): None = every PRB in both slots, else [2][nof_prb] of 0/1
        self.prb_mask = None if prb_mask is None else np.ascontiguousarray(prb_mask, np.uint8).reshape(2, nof_prb)
        self.Qm = MOD_BITS[mod]
        # bits per "symbol" in the code-block split of the rate matcher: Qm * N_L, N_L = 2 for transmit diversity (36.212 5.1.4.1.2;
        # srslte_dlsch_decode2 / _encode2, sch.c:507-531,:549-575)
        self.Qm_sch = self.Qm * (2 if nof_ports > 1 and tx_scheme is None else 1)
        self.nof_rx = nof_rx  # receive antennas (single tx port): MRC combining, SURVEY §8f N4
        self.llr8 = llr8  # 8-bit LLR path (pdsch.c q->llr_is_8bit, sch.c:336-338,:354-356), SURVEY §8f N2
        self.nof_ports = nof_ports
        # srslte_pdsch_cfg_t.power_scale / p_a (pdsch.c:518-554,:852-858; p_b chosen so that rho_b = 1, as phy_dl_test.c:176-178): the
        # receiver divides by rho_a = 10^(p_a/20) (x sqrt(2) for 2 ports); None = power_scale off
        self.p_a = p_a
        self.scaling = 1.0 if p_a is None else float(np.float32(10.0) ** np.float32(p_a / 20.0) * (np.float32(np.sqrt(np.float32(2.0))) if nof_ports > 1 else np.float32(1.0)))
        self.csi = csi  # srslte_pdsch_cfg_t.csi_enable: LLRs weighted by the channel gain (pdsch.c:574-690), the srsUE default
        # tdd = (uplink-downlink configuration 0-6, special-subframe configuration 0-9): srslte_cell_t.frame_type = SRSLTE_TDD + srslte_tdd_config_t
        self.tdd = tdd
        self.cell = OrcCell(cell_id, nof_prb, nof_ports, self.cp_norm, 1 if tdd else 0, tdd[0] if tdd else 0, tdd[1] if tdd else 0)
        self.nre = 12 * nof_prb
        self.grid_len = self.nsym * self.nre
        self.lstart = cfi + (1 if nof_prb < 10 else 0)
        self.N = oracle().orc_symbol_sz(nof_prb)
        self.sf_len = 15 * self.N
        self.chest = chest or {"filter_coef": (4.0, 1.0)}  # phy_dl_test.c:587-595
        self.seg = OrcCbsegm()
        assert oracle().orc_cbsegm(C.byref(self.seg), tbs) == 0 and self.seg.F == 0
        self.segs = [self.seg]
        if tbs2:
            self.segs.append(OrcCbsegm())
            assert oracle().orc_cbsegm(C.byref(self.segs[1]), tbs2) == 0 and self.segs[1].F == 0

    def indices(self, sf_idx):
        idx = np.zeros(self.grid_len, np.uint32)
        n = oracle().orc_pdsch_indices(C.byref(self.cell), sf_idx, self.lstart, None if self.prb_mask is None else p(self.prb_mask), p(idx))
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


def make_subframe(cfg, tti, rng, snr_db=None, amp=1.0, rv=0, data=None, keep=None):
    """Returns (iq[sf_len] complex64, payload bytes[tbs/8]) for TTI `tti`; rv / data: a HARQ retransmission of an earlier payload;
    keep: dict that receives the per-port PDSCH symbols before RE mapping ("y": list of arrays) and the RE indices ("idx")."""
    orc = oracle()
    sf_idx = tti % 10
    idx = cfg.indices(sf_idx)
    nbits = len(idx) * cfg.Qm
    if data is None:
        data = rng.integers(0, 256, cfg.tbs // 8, dtype=np.uint8)
    sch = OrcSchCfg(cfg.tbs, nbits, cfg.Qm_sch, rv, cfg.max_iter)
    e = np.zeros(nbits, np.uint8)
    assert orc.orc_dlsch_encode(C.byref(sch), p(data), p(e)) == 0
    e ^= scramble_seq(cfg, sf_idx, nbits)
    syms = np.zeros(len(idx), np.complex64)
    orc.orc_modulate(cfg.mod, p(e), p(syms), nbits)
    q = OrcOfdm()
    orc.orc_ofdm_init(C.byref(q), cfg.nof_prb, cfg.cp_norm)
    q.normalize = True
    if cfg.nof_ports > 1:
        return _make_subframe_2ports(cfg, sf_idx, idx, syms, q, rng, snr_db, amp, keep), data
    if cfg.scaling != 1.0:
        syms = syms * np.float32(cfg.scaling)  # rho_a (pdsch.c:1100-1114,:1166-1170)
    if keep is not None:
        keep.update(y=[syms.copy()], idx=idx)
    grid = np.zeros(cfg.grid_len, np.complex64)
    grid[idx] = syms
    orc.orc_crs_put_sf(C.byref(cfg.cell), sf_idx, 0, p(grid))
    iq = np.zeros(cfg.sf_len, np.complex64)
    orc.orc_ofdm_tx_sf(C.byref(q), p(grid), p(iq))
    iq *= np.float32(amp)
    # signal power per time sample with a normalised IFFT: nof_re/N per unit-power RE
    sigma = 0.0 if snr_db is None else np.sqrt(amp * amp * cfg.nre / cfg.N / 2) * 10 ** (-snr_db / 20)

    def noisy(x):
        return x if snr_db is None else x + (sigma * (rng.standard_normal(cfg.sf_len) + 1j * rng.standard_normal(cfg.sf_len))).astype(np.complex64)

    if cfg.nof_rx == 1:
        return noisy(iq).astype(np.complex64), data
    gains = (1.0, 0.6 * np.exp(1j * 1.0), 0.8 * np.exp(-1j * 2.0), 0.4j)[:cfg.nof_rx]  # flat per-antenna channels, own noise each
    return np.stack([noisy(np.complex64(g) * iq) for g in gains]).astype(np.complex64), data


def _make_subframe_2ports(cfg, sf_idx, idx, syms, q, rng, snr_db, amp, keep=None):
    """eNB side of pdsch.c:1150-1175 for 2- and 4-port transmit diversity: layer mapping + SFBC (+ FSTD) precoding, RE mapping and CRS
    per port; every (antenna, port) path has its own smooth frequency response (a gain and a delay), applied in the frequency domain."""
    orc = oracle()
    npt = cfg.nof_ports
    y = [np.zeros(len(idx), np.complex64) for _ in range(npt)]
    if npt == 2:
        orc.orc_precoding_diversity2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        orc.orc_precoding_diversity2(p(syms), p(y[0]), p(y[1]), len(idx), cfg.scaling)  # rho_a with power allocation (pdsch.c:1100-1114), else 1
    else:
        orc.orc_precoding_diversity4.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        orc.orc_precoding_diversity4(p(syms), (C.c_void_p * 4)(*[v.ctypes.data for v in y]), len(idx), cfg.scaling)
    if keep is not None:
        keep.update(y=[v.copy() for v in y], idx=idx)
    return _ports_to_iq(cfg, sf_idx, idx, y, q, rng, snr_db, amp)


def _ports_to_iq(cfg, sf_idx, idx, y, q, rng, snr_db, amp):
    """Per-port PDSCH symbols -> RE mapping + CRS per port -> every (antenna, port) path through its own smooth frequency response (a gain
    and a delay) -> OFDM TX per receive antenna, plus noise."""
    orc = oracle()
    npt = cfg.nof_ports
    tx = []
    for port in range(npt):
        g = np.zeros(cfg.grid_len, np.complex64)
        g[idx] = y[port]
        orc.orc_crs_put_sf(C.byref(cfg.cell), sf_idx, port, p(g))
        tx.append(g)
    k = (np.arange(cfg.grid_len) % cfg.nre) - cfg.nre / 2
    sigma = 0.0 if snr_db is None else np.sqrt(amp * amp * cfg.nre / cfg.N / 2) * 10 ** (-snr_db / 20)
    out = []
    for a in range(cfg.nof_rx):
        rxg = np.zeros(cfg.grid_len, np.complex128)
        for port in range(npt):
            gain = (1.0, 0.8 * np.exp(0.9j), 0.7 * np.exp(-2.0j), 0.9 * np.exp(2.4j), 0.85 * np.exp(1.7j), 0.75 * np.exp(-0.4j), 0.95 * np.exp(-2.9j),
                    0.65 * np.exp(0.3j))[(2 * a + port) % 8]
            rxg += tx[port] * gain * np.exp(-2j * np.pi * k * (0.6 + 0.9 * (port % 2) + 0.3 * (port // 2) + 0.5 * a) / cfg.N)
        rxg = np.ascontiguousarray(rxg.astype(np.complex64))
        iq = np.zeros(cfg.sf_len, np.complex64)
        orc.orc_ofdm_tx_sf(C.byref(q), p(rxg), p(iq))
        iq *= np.float32(amp)
        if snr_db is not None:
            iq = iq + (sigma * (rng.standard_normal(cfg.sf_len) + 1j * rng.standard_normal(cfg.sf_len))).astype(np.complex64)
        out.append(iq.astype(np.complex64))
    return out[0] if cfg.nof_rx == 1 else np.stack(out)


def make_subframe_mimo(cfg, tti, rng, snr_db=None, amp=1.0, rv=(0, 0), data=None, keep=None):
    """eNB side of pdsch.c:1059-1185 for the two-layer modes: every transport block coded, scrambled with its codeword's sequence and
    modulated on its own (srslte_pdsch_codeword_encode, :1000-1056), no layer mapping (nof_layers == nof_tb), srslte_precoding_type with
    CDD or the codebook, RE mapping + CRS per port, a 2x2 channel. Returns (iq [2][sf_len], [payload per TB])."""
    orc = oracle()
    sf_idx = tti % 10
    idx = cfg.indices(sf_idx)
    nre = len(idx)
    if data is None:
        data = [rng.integers(0, 256, t // 8, dtype=np.uint8) for t in cfg.tbss]
    x = []
    for cw in range(cfg.nof_tb):
        Qm = MOD_BITS[cfg.mods[cw]]
        sch = OrcSchCfg(cfg.tbss[cw], nre * Qm, Qm, rv[cw], cfg.max_iter)
        e = np.zeros(nre * Qm, np.uint8)
        assert orc.orc_dlsch_encode(C.byref(sch), p(np.ascontiguousarray(data[cw])), p(e)) == 0
        c = np.zeros(nre * Qm, np.uint8)
        orc.orc_gold(C.c_uint32(orc.orc_pdsch_cinit(cfg.rnti, cw, sf_idx, cfg.cell_id)), nre * Qm, p(c))
        syms = np.zeros(nre, np.complex64)
        orc.orc_modulate(cfg.mods[cw], p(e ^ c), p(syms), nre * Qm)
        x.append(syms)
    y = [np.zeros(nre, np.complex64) for _ in range(2)]
    if cfg.tx_scheme == "cdd":
        orc.orc_precoding_cdd2.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_float]
        orc.orc_precoding_cdd2(p(x[0]), p(x[1]), p(y[0]), p(y[1]), nre, cfg.scaling)
    else:
        orc.orc_precoding_mux2.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_float]
        assert orc.orc_precoding_mux2(p(x[0]), p(x[1]) if cfg.nof_tb == 2 else None, p(y[0]), p(y[1]), cfg.nof_tb, cfg.codebook_idx, nre, cfg.scaling) == 0
    if keep is not None:
        keep.update(x=x, y=[v.copy() for v in y], idx=idx)
    q = OrcOfdm()
    orc.orc_ofdm_init(C.byref(q), cfg.nof_prb, cfg.cp_norm)
    q.normalize = True
    return _ports_to_iq(cfg, sf_idx, idx, y, q, rng, snr_db, amp), data


def oracle_rx_mimo(cfg, iq, tti, keep=False, grid_in=None, rv=(0, 0)):
    """Oracle UE chain for the two-layer modes (pdsch.c:833-997 with tx_scheme CDD / SPATIALMUX): per-port, per-antenna estimation, the
    MMSE (or MRC) pre-decoder with its per-layer csi, then per codeword demapping, descrambling with that codeword's sequence, CSI
    weighting and DL-SCH decoding. Lists per transport block in the result."""
    orc = oracle()
    sf_idx, nrx, npt = tti % 10, 2, 2
    if grid_in is not None:
        grid = np.ascontiguousarray(grid_in, np.complex64).reshape(nrx, cfg.grid_len)
    else:
        q = OrcOfdm()
        orc.orc_ofdm_init(C.byref(q), cfg.nof_prb, cfg.cp_norm)
        grid = np.zeros((nrx, cfg.grid_len), np.complex64)
        iq2 = np.ascontiguousarray(iq, np.complex64).reshape(nrx, cfg.sf_len)
        for a in range(nrx):
            orc.orc_ofdm_rx_sf(C.byref(q), p(iq2[a]), p(grid[a]))
    res, ccfg = OrcChestRes(), cfg.orc_chest_cfg()
    idx = cfg.indices(sf_idx)
    nre = len(idx)
    ce = np.zeros((npt * nrx, cfg.grid_len), np.complex64)  # [port * nrx + antenna]
    gp, cp = (C.c_void_p * nrx)(*[g.ctypes.data for g in grid]), (C.c_void_p * (npt * nrx))(*[c.ctypes.data for c in ce])
    orc.orc_chest_dl_ports.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert orc.orc_chest_dl_ports(C.byref(cfg.cell), sf_idx, C.byref(ccfg), nrx, gp, cp, C.byref(res), None) == 0
    ys, hs = [np.ascontiguousarray(g[idx]) for g in grid], [np.ascontiguousarray(c[idx]) for c in ce]
    yp, hp = (C.c_void_p * nrx)(*[v.ctypes.data for v in ys]), (C.c_void_p * (npt * nrx))(*[v.ctypes.data for v in hs])
    x, csi = [np.zeros(nre, np.complex64) for _ in range(2)], [np.zeros(nre, np.float32) for _ in range(2)]
    noise = res.noise_estimate
    if cfg.tx_scheme == "cdd":
        orc.orc_predecoding_cdd_2x2.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_float]
        orc.orc_predecoding_cdd_2x2(yp, hp, p(x[0]), p(x[1]), p(csi[0]), p(csi[1]), nre, cfg.scaling, noise)
    elif cfg.nof_tb == 2:
        orc.orc_predecoding_mux_2x2.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_float, C.c_float]
        assert orc.orc_predecoding_mux_2x2(yp, hp, p(x[0]), p(x[1]), p(csi[0]), p(csi[1]), cfg.codebook_idx, nre, cfg.scaling, noise) == 0
    else:
        orc.orc_predecoding_mux_2x1.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float]
        assert orc.orc_predecoding_mux_2x1(yp, hp, p(x[0]), p(csi[0]), cfg.codebook_idx, nre, cfg.scaling) == 0
    out = {"tb": [], "ok": [], "iters": [], "cb_ok": [], "e": [], "e_raw": []}
    for cw in range(cfg.nof_tb):
        mod, Qm, tbs, seg = cfg.mods[cw], MOD_BITS[cfg.mods[cw]], cfg.tbss[cw], cfg.segs[cw]
        llr8 = getattr(cfg, "llr8", False)  # pdsch.c:760-779 takes q->llr_is_8bit with any scheme
        e = np.zeros(nre * Qm, np.int8 if llr8 else np.int16)
        (orc.orc_demod_soft_b if llr8 else orc.orc_demod_soft_s)(mod, p(x[cw]), p(e), nre)
        c = np.zeros(nre * Qm, np.uint8)
        orc.orc_gold(C.c_uint32(orc.orc_pdsch_cinit(cfg.rnti, cw, sf_idx, cfg.cell_id)), nre * Qm, p(c))
        (orc.orc_scramble_b if llr8 else orc.orc_scramble_s)(p(e), p(c), nre * Qm)
        out["e_raw"].append(e.copy())
        if cfg.csi:
            (orc.orc_csi_correction_b if llr8 else orc.orc_csi_correction_s)(p(e), p(csi[cw]), nre, mod)
        sch = OrcSchCfg(tbs, nre * Qm, Qm, rv[cw], cfg.max_iter)
        tb, iters, cbok = np.zeros(tbs // 8 + 16, np.uint8), np.zeros(seg.C, np.uint32), np.zeros(seg.C, np.uint8)
        rc = (orc.orc_dlsch_decode_8bit if llr8 else orc.orc_dlsch_decode)(C.byref(sch), p(e), p(tb), p(iters), p(cbok))
        out["tb"].append(tb[:tbs // 8 + 3])
        out["ok"].append(rc == 0)
        out["iters"].append(iters)
        out["cb_ok"].append(cbok)
        out["e"].append(e)
    if keep:
        out.update(grid=grid, ce=ce, noise=noise, d=x[:cfg.nof_tb], csi=csi[:cfg.nof_tb], res=res)
    return out


def make_grid(cfg, tti, rng, snr_db):
    """SURVEY §8d cfg5 stimulus: frequency-domain grid with CRS + PDSCH symbols through the smooth per-RE channel of
    chest_test_dl.c:159-161, h = (3 + x) e^{jx}, plus AWGN. Returns (grid[14*12*prb] complex64, payload bytes)."""
    orc = oracle()
    sf_idx = tti % 10
    idx = cfg.indices(sf_idx)
    nbits = len(idx) * cfg.Qm
    data = rng.integers(0, 256, cfg.tbs // 8, dtype=np.uint8)
    sch = OrcSchCfg(cfg.tbs, nbits, cfg.Qm_sch, 0, cfg.max_iter)
    e = np.zeros(nbits, np.uint8)
    assert orc.orc_dlsch_encode(C.byref(sch), p(data), p(e)) == 0
    e ^= scramble_seq(cfg, sf_idx, nbits)
    syms = np.zeros(len(idx), np.complex64)
    orc.orc_modulate(cfg.mod, p(e), p(syms), nbits)
    grid = np.zeros(cfg.grid_len, np.complex64)
    grid[idx] = syms
    orc.orc_crs_put_sf(C.byref(cfg.cell), sf_idx, 0, p(grid))
    k, l = np.arange(cfg.grid_len) % cfg.nre, np.arange(cfg.grid_len) // cfg.nre
    x = (k / cfg.nre * 2.0 + 0.0 * l).astype(np.float64)  # static over the subframe: the default estimator averages the pilots in time
    h = (3.0 + x) * np.exp(1j * x)
    sigma = np.sqrt(np.mean(np.abs(h) ** 2) / 2) * 10 ** (-snr_db / 20)
    noise = sigma * (rng.standard_normal(cfg.grid_len) + 1j * rng.standard_normal(cfg.grid_len))
    return (grid * h + noise).astype(np.complex64), data


class OrcHarq:
    """srslte_softbuffer_rx_t of one transport block for the oracle chain (buffer_f, cb_crc, data; softbuffer.c:46-150)."""

    def __init__(self, cfg):
        self.w = np.zeros(cfg.seg.C * oracle().orc_harq_softbuffer_stride(), np.int16)
        self.crc = np.zeros(cfg.seg.C, np.uint8)
        self.data = np.zeros(cfg.seg.C * 768, np.uint8)


def oracle_rx(cfg, iq, tti, keep=False, grid_in=None, harq=None, rv=0, new_data=True):
    """Oracle UE receive chain for one subframe. Returns dict with tb bytes (tbs/8+3), ok flag and (keep=True) every intermediate.
    grid_in: start from a frequency-domain grid instead of time samples. harq: an OrcHarq that persists between the transmissions
    of one transport block (rv, new_data as the MAC would set them)."""
    orc = oracle()
    sf_idx = tti % 10
    nrx = cfg.nof_rx
    if grid_in is not None:
        grid = np.ascontiguousarray(grid_in, np.complex64).reshape(nrx, cfg.grid_len)
    else:
        q = OrcOfdm()
        orc.orc_ofdm_init(C.byref(q), cfg.nof_prb, cfg.cp_norm)
        grid = np.zeros((nrx, cfg.grid_len), np.complex64)
        iq2 = np.ascontiguousarray(iq, np.complex64).reshape(nrx, cfg.sf_len)
        for a in range(nrx):
            orc.orc_ofdm_rx_sf(C.byref(q), p(iq2[a]), p(grid[a]))
    ce = np.zeros((nrx, cfg.grid_len), np.complex64)
    res = OrcChestRes()
    ccfg = cfg.orc_chest_cfg()
    idx = cfg.indices(sf_idx)
    d = np.zeros(len(idx), np.complex64)
    if cfg.nof_ports > 1:  # pdsch.c:890-935 with tx_scheme DIVERSITY: srslte_predecoding_diversity_multi (csi variant) + layer demapping
        npt = cfg.nof_ports
        ce = np.zeros((npt * nrx, cfg.grid_len), np.complex64)  # [port * nrx + antenna]
        gp, cp = (C.c_void_p * nrx)(*[g.ctypes.data for g in grid]), (C.c_void_p * (npt * nrx))(*[c.ctypes.data for c in ce])
        orc.orc_chest_dl_ports.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        assert orc.orc_chest_dl_ports(C.byref(cfg.cell), sf_idx, C.byref(ccfg), nrx, gp, cp, C.byref(res), None) == 0
        ys, hs = [np.ascontiguousarray(g[idx]) for g in grid], [np.ascontiguousarray(c[idx]) for c in ce]
        yp, hp = (C.c_void_p * nrx)(*[v.ctypes.data for v in ys]), (C.c_void_p * (npt * nrx))(*[v.ctypes.data for v in hs])
        fn = orc.orc_predecoding_diversity2 if npt == 2 else orc.orc_predecoding_diversity4
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        csi = np.zeros(len(idx), np.float32)
        fn(yp, hp, p(d), p(csi), nrx, len(idx), cfg.scaling)
    elif nrx == 1:
        assert orc.orc_chest_dl(C.byref(cfg.cell), sf_idx, C.byref(ccfg), p(grid[0]), p(ce[0]), C.byref(res)) == 0
        y, h = np.ascontiguousarray(grid[0][idx]), np.ascontiguousarray(ce[0][idx])
        orc.orc_predecoding_single(p(y), p(h), p(d), len(idx), cfg.scaling, res.noise_estimate)
        hs = [h]
        grid, ce = grid[0], ce[0]
    else:  # pdsch.c:890-935 with nof_rx_antennas > 1: srslte_predecoding_single_multi
        gp, cp = (C.c_void_p * nrx)(*[g.ctypes.data for g in grid]), (C.c_void_p * nrx)(*[c.ctypes.data for c in ce])
        assert orc.orc_chest_dl_multi(C.byref(cfg.cell), sf_idx, C.byref(ccfg), nrx, gp, cp, C.byref(res)) == 0
        ys, hs = [np.ascontiguousarray(g[idx]) for g in grid], [np.ascontiguousarray(c[idx]) for c in ce]
        yp, hp = (C.c_void_p * nrx)(*[v.ctypes.data for v in ys]), (C.c_void_p * nrx)(*[v.ctypes.data for v in hs])
        orc.orc_predecoding_single_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        orc.orc_predecoding_single_multi(yp, hp, p(d), nrx, len(idx), cfg.scaling, res.noise_estimate)
    if cfg.nof_ports == 1:
        csi = np.zeros(len(idx), np.float32)
        orc.orc_predecoding_csi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        orc.orc_predecoding_csi((C.c_void_p * nrx)(*[v.ctypes.data for v in hs]), p(csi), nrx, len(idx), res.noise_estimate)
    nbits = len(idx) * cfg.Qm
    e = np.zeros(nbits, np.int8 if cfg.llr8 else np.int16)
    sch = OrcSchCfg(cfg.tbs, nbits, cfg.Qm_sch, rv, cfg.max_iter)
    tb = np.zeros(cfg.tbs // 8 + 16, np.uint8)
    iters = np.zeros(cfg.seg.C, np.uint32)
    cbok = np.zeros(cfg.seg.C, np.uint8)
    if cfg.llr8:
        orc.orc_demod_soft_b(cfg.mod, p(d), p(e), len(idx))
        orc.orc_scramble_b(p(e), p(scramble_seq(cfg, sf_idx, nbits)), nbits)
        e_raw = e.copy()
        if cfg.csi:
            orc.orc_csi_correction_b(p(e), p(csi), len(idx), cfg.mod)
    else:
        orc.orc_demod_soft_s(cfg.mod, p(d), p(e), len(idx))
        orc.orc_scramble_s(p(e), p(scramble_seq(cfg, sf_idx, nbits)), nbits)
        e_raw = e.copy()
        if cfg.csi:
            orc.orc_csi_correction_s(p(e), p(csi), len(idx), cfg.mod)
    if harq is not None:
        rc = orc.orc_dlsch_decode_harq(C.byref(sch), p(e), 1 if cfg.llr8 else 0, 1 if new_data else 0, p(harq.w), p(harq.crc), p(harq.data),
                                       p(tb), p(iters), p(cbok))
    elif cfg.llr8:
        rc = orc.orc_dlsch_decode_8bit(C.byref(sch), p(e), p(tb), p(iters), p(cbok))
    else:
        rc = orc.orc_dlsch_decode(C.byref(sch), p(e), p(tb), p(iters), p(cbok))
    out = {"tb": tb[:cfg.tbs // 8 + 3], "ok": rc == 0, "iters": iters, "cb_ok": cbok}
    if keep:
        out.update(grid=grid, ce=ce, noise=res.noise_estimate, d=d, e=e, e_raw=e_raw, csi=csi, res=res)
    return out


class RefRx:
    """UE receive chain on the REFERENCE's own compiled code (oracle/_ref/libsrslte_ref.so) for every stage it holds:
    srslte_chest_dl_estimate_cfg, srslte_predecoding_single, srslte_demod_soft_demodulate_s, srslte_rm_turbo_rx_lut,
    srslte_tdec_new_cb/_iteration and srslte_crc_checksum_byte, driven as pdsch.c:833-997 / sch.c:299-500 drive them.
    The FFT is the oracle's (FFTW is absent from this image), RE extraction and descrambling are numpy one-liners.
    Used as bench.py's cpu_baseline (kind "reference") and to pin the oracle chain in tests."""

    def __init__(self, cfg):
        from _libs import RefCell, RefChestCfg, RefChestRes, RefDlSfCfg, aligned, opaque, ref
        self.R = ref()
        assert self.R is not None, "oracle/_ref/libsrslte_ref.so is not built"
        self.cfg = cfg
        self.aligned = aligned
        self.chest = opaque(1 << 20)
        assert self.R.srslte_chest_dl_init(self.chest, cfg.nof_prb, cfg.nof_rx) == 0
        assert self.R.srslte_chest_dl_set_cell(self.chest, RefCell(cfg.nof_prb, cfg.nof_ports, cfg.cell_id, 1 if cfg.cp_ext else 0, 0, 0, 0)) == 0
        self.rc = RefChestCfg()
        for k, v in cfg.chest.items():
            if k == "filter_coef":
                self.rc.filter_coef[0], self.rc.filter_coef[1] = v
            else:
                setattr(self.rc, k, v)
        self.res, self.sf = RefChestRes(), RefDlSfCfg()
        self.ces = [aligned(2 * cfg.grid_len, np.float32) for _ in range(cfg.nof_rx * cfg.nof_ports)]  # [port * nof_rx + antenna]
        for i_, c_ in enumerate(self.ces):
            self.res.ce[i_ // cfg.nof_rx][i_ % cfg.nof_rx] = c_.ctypes.data
        self.ce = self.ces[0]
        self.R.srslte_predecoding_diversity_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        self.R.srslte_layerdemap_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.R.srslte_predecoding_single_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        self.tdec = opaque(1 << 20)
        assert self.R.srslte_tdec_init(self.tdec, 6144) == 0
        self.crc_tb, self.crc_cb = opaque(4096), opaque(4096)
        self.R.srslte_crc_init(self.crc_tb, 0x1864CFB, 24)
        self.R.srslte_crc_init(self.crc_cb, 0x1800063, 24)
        self.R.srslte_crc_checksum_byte.restype = C.c_uint32
        self.R.srslte_cbsegm_cbindex.restype = C.c_int
        self.q = OrcOfdm()
        oracle().orc_ofdm_init(C.byref(self.q), cfg.nof_prb, cfg.cp_norm)
        self.idx = {s: cfg.indices(s) for s in range(10)}
        self.scr = {}

    def run(self, iq, tti):
        cfg, R, orc = self.cfg, self.R, oracle()
        sf_idx = tti % 10
        nrx = cfg.nof_rx
        grids = [self.aligned(2 * cfg.grid_len, np.float32) for _ in range(nrx)]
        iq2 = np.ascontiguousarray(iq, np.complex64).reshape(nrx, cfg.sf_len)
        for a_ in range(nrx):
            orc.orc_ofdm_rx_sf(C.byref(self.q), p(iq2[a_]), p(grids[a_]))
        grid = grids[0]
        self.sf.tti = sf_idx
        inp = (C.c_void_p * 4)(*([g.ctypes.data for g in grids] + [0] * (4 - nrx)))
        assert R.srslte_chest_dl_estimate_cfg(self.chest, C.byref(self.sf), C.byref(self.rc), inp, C.byref(self.res)) == 0
        idx = self.idx[sf_idx]
        n = len(idx)
        d = self.aligned(2 * n, np.float32)
        ys, hs = [self.aligned(2 * n, np.float32) for _ in range(nrx)], [self.aligned(2 * n, np.float32) for _ in range(nrx * cfg.nof_ports)]
        for a_ in range(nrx):
            ys[a_].view(np.complex64)[:] = grids[a_].view(np.complex64)[idx]
        for i_ in range(nrx * cfg.nof_ports):
            hs[i_].view(np.complex64)[:] = self.ces[i_].view(np.complex64)[idx]
        if cfg.nof_ports > 1:
            npt = cfg.nof_ports
            yp = (C.c_void_p * 4)(*([v.ctypes.data for v in ys] + [0] * (4 - nrx)))
            hp = ((C.c_void_p * 4) * 4)()
            for i_ in range(npt * nrx):
                hp[i_ // nrx][i_ % nrx] = hs[i_].ctypes.data
            x = [self.aligned(n, np.float32) for _ in range(npt)]
            xp = (C.c_void_p * 4)(*([v.ctypes.data for v in x] + [0] * (4 - npt)))
            csi = self.aligned(n, np.float32)
            R.srslte_predecoding_diversity_multi(yp, hp, xp, (C.c_void_p * 2)(csi.ctypes.data, 0), nrx, npt, n, 1.0)
            R.srslte_layerdemap_diversity(xp, p(d), npt, n // npt)
        elif nrx == 1:
            R.srslte_predecoding_single(p(ys[0]), p(hs[0]), p(d), None, n, 1.0, self.res.noise_estimate)
        else:
            yp = (C.c_void_p * 4)(*([v.ctypes.data for v in ys] + [0] * (4 - nrx)))
            hp = (C.c_void_p * 4)(*([v.ctypes.data for v in hs] + [0] * (4 - nrx)))
            R.srslte_predecoding_single_multi(yp, hp, p(d), None, nrx, n, 1.0, self.res.noise_estimate)
        nbits = n * cfg.Qm
        lt = np.int8 if cfg.llr8 else np.int16
        e = self.aligned(nbits + 64, lt)
        (R.srslte_demod_soft_demodulate_b if cfg.llr8 else R.srslte_demod_soft_demodulate_s)(cfg.mod, p(d), p(e), n)
        if (sf_idx, nbits) not in self.scr:
            self.scr[(sf_idx, nbits)] = scramble_seq(cfg, sf_idx, nbits).astype(bool)
        ev = e[:nbits]
        ev[self.scr[(sf_idx, nbits)]] *= -1  # wraps -(-128) / -(-32768) like the reference's sign instructions
        return ref_sch_decode(self, e, nbits)


class RefPdsch:
    """The reference's own srslte_pdsch_decode (pdsch.c:833-997) on its compiled code: srslte_pdsch_init_ue / set_cell / set_rnti, a
    hand-filled srslte_pdsch_cfg_t (grant: full band, one codeword, TM1 or TM2) and the srslte_chest_dl_res_t of
    srslte_chest_dl_estimate_cfg. Only the FFT in front of it is the oracle's. Struct offsets come from the reference headers at run
    time (_libs.ref_layout). Pins the stage-by-stage chains (RefRx, oracle_rx) and the CSI weighting of the LLRs."""

    def __init__(self, cfg, csi_enable=False, lib=None):
        from _libs import RefCell, RefChestCfg, RefChestRes, RefDlSfCfg, aligned, opaque, ref, ref_layout
        R = self.R = lib if lib is not None else ref()
        self.cfg, self.aligned = cfg, aligned
        L = self.L = ref_layout({"srslte_pdsch_t": ["llr_is_8bit", "d", "e", "csi", "dl_sch"], "srslte_sch_t": ["llr_is_8bit"],
                                 "srslte_pdsch_cfg_t": ["rnti", "max_nof_iterations", "decoder_type", "csi_enable", "softbuffers", "p_a", "p_b", "power_scale"],
                                 "srslte_pdsch_grant_t": ["tx_scheme", "pmi", "prb_idx", "nof_prb", "nof_re", "nof_symb_slot", "tb", "nof_tb", "nof_layers"],
                                 "srslte_ra_tb_t": ["mod", "tbs", "rv", "nof_bits", "cw_idx", "enabled"],
                                 "srslte_softbuffer_rx_t": [], "srslte_pdsch_res_t": ["payload", "crc"]}, ["srslte/phy/phch/pdsch.h"])
        cell = RefCell(cfg.nof_prb, cfg.nof_ports, cfg.cell_id, 1 if cfg.cp_ext else 0, 0, 0, 1 if cfg.tdd else 0)  # srslte_cp_t NORM 0 / EXT 1; frame type FDD 0 / TDD 1
        self.chest = opaque(1 << 20)
        assert R.srslte_chest_dl_init(self.chest, cfg.nof_prb, cfg.nof_rx) == 0 and R.srslte_chest_dl_set_cell(self.chest, cell) == 0
        self.rc = RefChestCfg()
        for k, v in cfg.chest.items():
            if k == "filter_coef":
                self.rc.filter_coef[0], self.rc.filter_coef[1] = v
            else:
                setattr(self.rc, k, v)
        self.res, self.sf = RefChestRes(), RefDlSfCfg()
        self.ces = [aligned(2 * cfg.grid_len, np.float32) for _ in range(cfg.nof_rx * cfg.nof_ports)]
        for i_, c_ in enumerate(self.ces):
            self.res.ce[i_ // cfg.nof_rx][i_ % cfg.nof_rx] = c_.ctypes.data
        self.q = opaque(L["srslte_pdsch_t"] + 64)
        assert R.srslte_pdsch_init_ue(self.q, cfg.nof_prb, cfg.nof_rx) == 0 and R.srslte_pdsch_set_cell(self.q, cell) == 0
        R.srslte_pdsch_set_rnti.argtypes = [C.c_void_p, C.c_uint16]
        assert R.srslte_pdsch_set_rnti(self.q, cfg.rnti) == 0
        for off in (L["srslte_pdsch_t.llr_is_8bit"], L["srslte_pdsch_t.dl_sch"] + L["srslte_sch_t.llr_is_8bit"]):  # srsue cc_worker.cc:101-102
            self.q[off] = b"\x01" if cfg.llr8 else b"\x00"
        self.sb = opaque(L["srslte_softbuffer_rx_t"] + 64)
        assert R.srslte_softbuffer_rx_init(self.sb, cfg.nof_prb) == 0
        self.pc = np.zeros(L["srslte_pdsch_cfg_t"], np.uint8)
        g = self.pc  # grant at offset 0

        def u32(off, v):
            g[off:off + 4].view(np.uint32)[0] = v
        # srslte_tx_scheme_t (phy_common.h:232-237): PORT0, DIVERSITY, SPATIALMUX, CDD
        u32(L["srslte_pdsch_grant_t.tx_scheme"], {"cdd": 3, "mux": 2}[cfg.tx_scheme] if cfg.tx_scheme else (1 if cfg.nof_ports > 1 else 0))
        u32(L["srslte_pdsch_grant_t.pmi"], cfg.pmi)
        g[L["srslte_pdsch_grant_t.prb_idx"]:L["srslte_pdsch_grant_t.prb_idx"] + 220].reshape(2, 110)[:, :cfg.nof_prb] = 1 if cfg.prb_mask is None else cfg.prb_mask
        u32(L["srslte_pdsch_grant_t.nof_prb"], cfg.nof_prb if cfg.prb_mask is None else int(cfg.prb_mask[0].sum()))
        u32(L["srslte_pdsch_grant_t.nof_symb_slot"], cfg.nsym // 2)
        u32(L["srslte_pdsch_grant_t.nof_symb_slot"] + 4, cfg.nsym // 2)
        u32(L["srslte_pdsch_grant_t.nof_tb"], cfg.nof_tb)
        u32(L["srslte_pdsch_grant_t.nof_layers"], cfg.nof_tb if cfg.tx_scheme else cfg.nof_ports)
        self.tb0 = L["srslte_pdsch_grant_t.tb"]
        for cw in range(cfg.nof_tb):
            o = self.tb0 + cw * L["srslte_ra_tb_t"]
            u32(o + L["srslte_ra_tb_t.mod"], cfg.mods[cw])
            u32(o + L["srslte_ra_tb_t.tbs"], cfg.tbss[cw])
            u32(o + L["srslte_ra_tb_t.cw_idx"], cw)
            g[o + L["srslte_ra_tb_t.enabled"]] = 1
        g[L["srslte_pdsch_cfg_t.rnti"]:L["srslte_pdsch_cfg_t.rnti"] + 2].view(np.uint16)[0] = cfg.rnti
        u32(L["srslte_pdsch_cfg_t.max_nof_iterations"], cfg.max_iter)
        u32(L["srslte_pdsch_cfg_t.decoder_type"], 1)  # SRSLTE_MIMO_DECODER_MMSE
        g[L["srslte_pdsch_cfg_t.csi_enable"]] = 1 if csi_enable else 0
        if cfg.p_a is not None:
            g[L["srslte_pdsch_cfg_t.power_scale"]] = 1
            g[L["srslte_pdsch_cfg_t.p_a"]:L["srslte_pdsch_cfg_t.p_a"] + 4].view(np.float32)[0] = cfg.p_a
            u32(L["srslte_pdsch_cfg_t.p_b"], 1 if cfg.nof_ports > 1 else 0)  # rho_b = 1 (phy_dl_test.c:178)
        g[L["srslte_pdsch_cfg_t.softbuffers"]:L["srslte_pdsch_cfg_t.softbuffers"] + 8].view(np.uint64)[0] = C.addressof(self.sb)
        if cfg.nof_tb == 2:  # softbuffers.rx[1]
            self.sb1 = opaque(L["srslte_softbuffer_rx_t"] + 64)
            assert R.srslte_softbuffer_rx_init(self.sb1, cfg.nof_prb) == 0
            g[L["srslte_pdsch_cfg_t.softbuffers"] + 8:L["srslte_pdsch_cfg_t.softbuffers"] + 16].view(np.uint64)[0] = C.addressof(self.sb1)
        self.u32 = u32
        self.ofdm = OrcOfdm()
        oracle().orc_ofdm_init(C.byref(self.ofdm), cfg.nof_prb, cfg.cp_norm)
        self.nre = {s_: len(cfg.indices(s_)) for s_ in (0, 5, 1)}

    def _ptr(self, name, dtype, count, cw=0):
        addr = np.frombuffer(self.q, np.uint64, 1, self.L["srslte_pdsch_t." + name] + 8 * cw)[0]
        nbytes = count * np.dtype(dtype).itemsize
        return np.frombuffer(C.string_at(int(addr), nbytes), dtype).copy()

    def run_mimo(self, iq, tti, grid_in=None):
        """Two-layer modes: per transport block lists of tb / ok / d / e / csi (rv 0, new data)."""
        cfg, R, L = self.cfg, self.R, self.L
        nrx = cfg.nof_rx
        grids = [self.aligned(2 * cfg.grid_len, np.float32) for _ in range(nrx)]
        if grid_in is not None:
            for a_ in range(nrx):
                grids[a_].view(np.complex64)[:] = np.asarray(grid_in, np.complex64).reshape(nrx, -1)[a_]
        else:
            iq2 = np.ascontiguousarray(iq, np.complex64).reshape(nrx, cfg.sf_len)
            for a_ in range(nrx):
                oracle().orc_ofdm_rx_sf(C.byref(self.ofdm), p(iq2[a_]), p(grids[a_]))
        self.sf.tti, self.sf.cfi = tti, cfg.cfi
        inp = (C.c_void_p * 4)(*([g_.ctypes.data for g_ in grids] + [0] * (4 - nrx)))
        assert R.srslte_chest_dl_estimate_cfg(self.chest, C.byref(self.sf), C.byref(self.rc), inp, C.byref(self.res)) == 0
        nre = self.nre[0 if tti % 10 == 0 else (5 if tti % 10 == 5 else 1)]
        self.u32(L["srslte_pdsch_grant_t.nof_re"], nre)
        R.srslte_softbuffer_rx_reset(self.sb)
        payloads = [np.zeros(t // 8 + 64, np.uint8) for t in cfg.tbss]
        data = np.zeros(2 * L["srslte_pdsch_res_t"], np.uint8)
        for cw in range(cfg.nof_tb):
            o = self.tb0 + cw * L["srslte_ra_tb_t"]
            self.u32(o + L["srslte_ra_tb_t.nof_bits"], nre * MOD_BITS[cfg.mods[cw]])
            self.u32(o + L["srslte_ra_tb_t.rv"], 0)
            data[cw * L["srslte_pdsch_res_t"]:][:8].view(np.uint64)[0] = payloads[cw].ctypes.data
            if cw:
                R.srslte_softbuffer_rx_reset(self.sb1)
        assert R.srslte_pdsch_decode(self.q, C.byref(self.sf), p(self.pc), C.byref(self.res), inp, p(data)) == 0
        out = {"tb": [], "ok": [], "d": [], "e": [], "csi": [], "noise": self.res.noise_estimate}
        for cw in range(cfg.nof_tb):
            out["tb"].append(payloads[cw][:cfg.tbss[cw] // 8 + 3].copy())
            out["ok"].append(bool(data[cw * L["srslte_pdsch_res_t"] + L["srslte_pdsch_res_t.crc"]]))
            out["d"].append(self._ptr("d", np.complex64, nre, cw))
            out["e"].append(self._ptr("e", np.int16, nre * MOD_BITS[cfg.mods[cw]], cw))
            out["csi"].append(self._ptr("csi", np.float32, nre, cw))
        return out

    def run(self, iq, tti, grid_in=None, rv=0, new_data=True):
        cfg, R, L = self.cfg, self.R, self.L
        sf_idx, nrx = tti % 10, cfg.nof_rx
        grids = [self.aligned(2 * cfg.grid_len, np.float32) for _ in range(nrx)]
        if grid_in is not None:
            for a_ in range(nrx):
                grids[a_].view(np.complex64)[:] = np.asarray(grid_in, np.complex64).reshape(nrx, -1)[a_]
        else:
            iq2 = np.ascontiguousarray(iq, np.complex64).reshape(nrx, cfg.sf_len)
            for a_ in range(nrx):
                oracle().orc_ofdm_rx_sf(C.byref(self.ofdm), p(iq2[a_]), p(grids[a_]))
        self.sf.tti, self.sf.cfi = tti, cfg.cfi
        nre = self.nre[0 if sf_idx == 0 else (5 if sf_idx == 5 else 1)]
        if cfg.tdd:  # srslte_dl_sf_cfg_t.tdd_config and what srslte_ra_dl_compute_nof_re derives from it (ra_dl.c:446-460)
            self.sf.tdd_config.sf_config, self.sf.tdd_config.ss_config, self.sf.tdd_config.configured = cfg.tdd[0], cfg.tdd[1], True
            orc_ = oracle()
            orc_.orc_nof_symb_slot.restype = C.c_uint32
            for s_ in (0, 1):
                self.u32(L["srslte_pdsch_grant_t.nof_symb_slot"] + 4 * s_, orc_.orc_nof_symb_slot(C.byref(cfg.cell), sf_idx, s_))
            nre = len(cfg.indices(sf_idx))
        inp = (C.c_void_p * 4)(*([g_.ctypes.data for g_ in grids] + [0] * (4 - nrx)))
        assert R.srslte_chest_dl_estimate_cfg(self.chest, C.byref(self.sf), C.byref(self.rc), inp, C.byref(self.res)) == 0
        self.u32(L["srslte_pdsch_grant_t.nof_re"], nre)
        self.u32(self.tb0 + L["srslte_ra_tb_t.nof_bits"], nre * cfg.Qm)
        self.u32(self.tb0 + L["srslte_ra_tb_t.rv"], rv)
        if new_data:
            R.srslte_softbuffer_rx_reset(self.sb)
        payload = np.zeros(cfg.tbs // 8 + 64, np.uint8)
        data = np.zeros(2 * L["srslte_pdsch_res_t"], np.uint8)
        data[:8].view(np.uint64)[0] = payload.ctypes.data
        assert R.srslte_pdsch_decode(self.q, C.byref(self.sf), p(self.pc), C.byref(self.res), inp, p(data)) == 0
        ok = bool(data[L["srslte_pdsch_res_t.crc"]])
        e = self._ptr("e", np.int8 if cfg.llr8 else np.int16, nre * cfg.Qm)
        return {"tb": payload[:cfg.tbs // 8 + 3].copy(), "ok": ok, "d": self._ptr("d", np.complex64, nre), "e": e,
                "csi": self._ptr("csi", np.float32, nre), "noise": self.res.noise_estimate}


# smoothing filters of the three configurations of tests/golden/pmch.npz: "b" the applications' (triangle 0.1), the others phy_dl_test's Gauss filter
PMCH_GOLDEN_CHEST = {"a": None, "b": {"filter_type": 1, "filter_coef": (0.1, 0.0)}, "c": None}


class PmchConfig:
    """One PMCH configuration (pmch.c): an MBSFN subframe of MBSFN area `area_id` on a single-port cell - 12 extended-CP symbols
    behind a non-MBSFN region of `non_mbsfn_region` symbols, all PRBs, rv 0, scrambled with the area's sequence. cp_ext is the CELL's
    cyclic prefix: it only selects the CRS sequence of symbol 0 (N_CP in c_init)."""

    def __init__(self, nof_prb, cell_id, area_id, mod, tbs, cfi=2, non_mbsfn_region=2, max_iter=6, chest=None, nof_rx=1, cp_ext=True):
        self.nof_prb, self.cell_id, self.area_id, self.mod, self.tbs, self.cfi, self.max_iter = nof_prb, cell_id, area_id, mod, tbs, cfi, max_iter
        self.non_mbsfn_region, self.nof_rx, self.cp_ext = non_mbsfn_region, nof_rx, bool(cp_ext)
        self.Qm = MOD_BITS[mod]
        self.cell = OrcCell(cell_id, nof_prb, 1, not cp_ext, 0, 0, 0)
        self.nre = 12 * nof_prb
        self.grid_len = 12 * self.nre
        self.lstart = cfi + (1 if nof_prb < 10 else 0)  # SRSLTE_NOF_CTRL_SYMBOLS (pmch.c:322)
        self.N = oracle().orc_symbol_sz(nof_prb)
        self.sf_len = 15 * self.N
        self.chest = chest or {"filter_coef": (4.0, 1.0)}
        self.chest = dict(self.chest, interpolate_subframe=True)  # the MBSFN estimate is only defined with it (chest_dl.c:430-478)
        self.seg = OrcCbsegm()
        assert oracle().orc_cbsegm(C.byref(self.seg), tbs) == 0 and self.seg.F == 0
        idx = np.zeros(self.grid_len, np.uint32)
        self.idx = idx[:oracle().orc_pmch_indices(nof_prb, self.lstart, p(idx))].copy()
        self.nof_re, self.nbits = len(self.idx), len(self.idx) * self.Qm

    def orc_chest_cfg(self):
        return DlConfig.orc_chest_cfg(self)

    def ofdm(self, normalize):
        q = OrcOfdm()
        oracle().orc_ofdm_init(C.byref(q), self.nof_prb, False)
        q.non_mbsfn_region, q.normalize = self.non_mbsfn_region, normalize
        return q

    def scramble(self, sf_idx):
        c = np.zeros(self.nbits, np.uint8)
        orc = oracle()
        orc.orc_pmch_cinit.restype = C.c_uint32
        orc.orc_gold(C.c_uint32(orc.orc_pmch_cinit(sf_idx, self.area_id)), self.nbits, p(c))
        return c


def make_pmch_subframe(cfg, tti, rng, snr_db=None, amp=1.0, data=None, keep=None):
    """eNB side of an MBSFN subframe: srslte_pmch_encode (pmch.c:423-483) + srslte_refsignal_mbsfn_put_sf (enb_dl.c put_mbsfn_base_signals) +
    srslte_ofdm_tx_sf on an MBSFN OFDM object (ofdm.c:558-574) -> (iq [nof_rx][sf_len] or [sf_len], payload bytes)."""
    orc = oracle()
    sf_idx = tti % 10
    if data is None:
        data = rng.integers(0, 256, cfg.tbs // 8, dtype=np.uint8)
    sch = OrcSchCfg(cfg.tbs, cfg.nbits, cfg.Qm, 0, cfg.max_iter)
    e = np.zeros(cfg.nbits, np.uint8)
    assert orc.orc_dlsch_encode(C.byref(sch), p(data), p(e)) == 0
    e ^= cfg.scramble(sf_idx)
    syms = np.zeros(cfg.nof_re, np.complex64)
    orc.orc_modulate(cfg.mod, p(e), p(syms), cfg.nbits)
    grid = np.zeros(cfg.grid_len, np.complex64)
    grid[cfg.idx] = syms
    assert orc.orc_mbsfn_put_sf(C.byref(cfg.cell), sf_idx, 0, cfg.area_id, p(grid)) == 0
    if keep is not None:
        keep.update(d=syms.copy(), grid=grid.copy(), e=e.copy())
    q = cfg.ofdm(True)
    iq = np.zeros(cfg.sf_len, np.complex64)
    orc.orc_ofdm_tx_sf(C.byref(q), p(grid), p(iq))
    iq *= np.float32(amp)
    sigma = 0.0 if snr_db is None else np.sqrt(amp * amp * cfg.nre / cfg.N / 2) * 10 ** (-snr_db / 20)

    def noisy(x):
        return x if snr_db is None else x + (sigma * (rng.standard_normal(cfg.sf_len) + 1j * rng.standard_normal(cfg.sf_len))).astype(np.complex64)

    if cfg.nof_rx == 1:
        return noisy(iq).astype(np.complex64), data
    gains = (1.0, 0.6 * np.exp(1j * 1.0), 0.8 * np.exp(-1j * 2.0), 0.4j)[:cfg.nof_rx]
    return np.stack([noisy(np.complex64(g) * iq) for g in gains]).astype(np.complex64), data


def oracle_pmch_rx(cfg, iq, tti, keep=False):
    """UE side of an MBSFN subframe: srslte_ofdm_rx_sf on the MBSFN object (ofdm.c:424-437), srslte_chest_dl_estimate_cfg with sf_type MBSFN
    (chest_dl.c:718-745; noise = the REFS estimate averaged over the antennas, :757-764) and srslte_pmch_decode (pmch.c:291-394)."""
    orc = oracle()
    sf_idx, nrx = tti % 10, cfg.nof_rx
    q = cfg.ofdm(False)
    grid, ce = np.zeros((nrx, cfg.grid_len), np.complex64), np.zeros((nrx, cfg.grid_len), np.complex64)
    iq2 = np.ascontiguousarray(iq, np.complex64).reshape(nrx, cfg.sf_len)
    ccfg = cfg.orc_chest_cfg()
    noise = np.zeros(nrx, np.float32)
    orc.orc_chest_dl_mbsfn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    for a in range(nrx):
        orc.orc_ofdm_rx_sf(C.byref(q), p(iq2[a]), p(grid[a]))
        n_ = C.c_float(0)
        assert orc.orc_chest_dl_mbsfn(C.byref(cfg.cell), sf_idx, C.byref(ccfg), cfg.area_id, 0, p(grid[a]), p(ce[a]), C.byref(n_)) == 0
        noise[a] = n_.value
    n0 = np.float32(0)
    for a in range(nrx):  # get_noise (chest_dl.c:747-758): float sums in antenna order
        n0 = np.float32(n0 + noise[a])
    n0 = np.float32(n0 / np.float32(nrx)) if nrx > 1 else noise[0]
    d = np.zeros(cfg.nof_re, np.complex64)
    ys, hs = [np.ascontiguousarray(g[cfg.idx]) for g in grid], [np.ascontiguousarray(c[cfg.idx]) for c in ce]
    if nrx == 1:
        orc.orc_predecoding_single(p(ys[0]), p(hs[0]), p(d), cfg.nof_re, 1.0, float(n0))
    else:
        yp, hp = (C.c_void_p * nrx)(*[v.ctypes.data for v in ys]), (C.c_void_p * nrx)(*[v.ctypes.data for v in hs])
        orc.orc_predecoding_single_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        orc.orc_predecoding_single_multi(yp, hp, p(d), nrx, cfg.nof_re, 1.0, float(n0))
    e = np.zeros(cfg.nbits, np.int16)
    orc.orc_demod_soft_s(cfg.mod, p(d), p(e), cfg.nof_re)
    orc.orc_scramble_s(p(e), p(cfg.scramble(sf_idx)), cfg.nbits)
    sch = OrcSchCfg(cfg.tbs, cfg.nbits, cfg.Qm, 0, cfg.max_iter)
    tb, iters, cbok = np.zeros(cfg.tbs // 8 + 16, np.uint8), np.zeros(cfg.seg.C, np.uint32), np.zeros(cfg.seg.C, np.uint8)
    rc = orc.orc_dlsch_decode(C.byref(sch), p(e), p(tb), p(iters), p(cbok))
    out = {"tb": tb[:cfg.tbs // 8 + 3], "ok": rc == 0, "iters": iters, "cb_ok": cbok}
    if keep:
        out.update(grid=grid, ce=ce, noise=float(n0), d=d, e=e)
    return out


class RefPmch:
    """The reference's own srslte_pmch_encode / srslte_pmch_decode (pmch.c:291-483) and its MBSFN channel estimate on the compiled code, with a
    hand-filled srslte_pmch_cfg_t; only the FFT around them is the oracle's (FFTW is absent). Pins orc_pmch_indices, orc_pmch_cinit and the
    PMCH chains of make_pmch_subframe / oracle_pmch_rx."""

    def __init__(self, cfg):
        from _libs import RefCell, RefChestCfg, RefChestRes, RefDlSfCfg, aligned, opaque, ref, ref_layout
        R = self.R = ref()
        self.cfg, self.aligned = cfg, aligned
        L = self.L = ref_layout({"srslte_pmch_t": ["d", "e"], "srslte_pmch_cfg_t": ["pdsch_cfg", "area_id"],
                                 "srslte_pdsch_cfg_t": ["max_nof_iterations", "softbuffers"],
                                 "srslte_pdsch_grant_t": ["prb_idx", "nof_prb", "nof_re", "tb", "nof_tb", "nof_layers"],
                                 "srslte_ra_tb_t": ["mod", "tbs", "rv", "nof_bits", "cw_idx", "enabled"],
                                 "srslte_softbuffer_rx_t": [], "srslte_softbuffer_tx_t": [], "srslte_pdsch_res_t": ["payload", "crc"]},
                                ["srslte/phy/phch/pmch.h"])
        cell = RefCell(cfg.nof_prb, 1, cfg.cell_id, 1 if cfg.cp_ext else 0, 0, 0, 0)
        self.chest = opaque(1 << 20)
        assert R.srslte_chest_dl_init(self.chest, cfg.nof_prb, cfg.nof_rx) == 0 and R.srslte_chest_dl_set_cell(self.chest, cell) == 0
        R.srslte_chest_dl_set_mbsfn_area_id.argtypes = [C.c_void_p, C.c_uint16]
        assert R.srslte_chest_dl_set_mbsfn_area_id(self.chest, cfg.area_id) == 0
        self.rc = RefChestCfg()
        for k, v in cfg.chest.items():
            if k == "filter_coef":
                self.rc.filter_coef[0], self.rc.filter_coef[1] = v
            else:
                setattr(self.rc, k, v)
        self.rc.mbsfn_area_id = cfg.area_id
        self.res, self.sf = RefChestRes(), RefDlSfCfg()
        self.sf.sf_type, self.sf.cfi, self.sf.non_mbsfn_region = 1, cfg.cfi, cfg.non_mbsfn_region  # SRSLTE_SF_MBSFN
        self.glen = 14 * cfg.nre  # the reference's buffers hold a normal-CP subframe's worth; an MBSFN subframe uses the first 12 symbols
        self.ces = [aligned(2 * self.glen, np.float32) for _ in range(cfg.nof_rx)]
        for a_, c_ in enumerate(self.ces):
            self.res.ce[0][a_] = c_.ctypes.data
        self.q = opaque(L["srslte_pmch_t"] + 64)
        assert R.srslte_pmch_init(self.q, cfg.nof_prb, cfg.nof_rx) == 0 and R.srslte_pmch_set_cell(self.q, cell) == 0
        R.srslte_pmch_set_area_id.argtypes = [C.c_void_p, C.c_uint16]
        assert R.srslte_pmch_set_area_id(self.q, cfg.area_id) == 0
        self.sb_rx, self.sb_tx = opaque(L["srslte_softbuffer_rx_t"] + 64), opaque(L["srslte_softbuffer_tx_t"] + 64)
        assert R.srslte_softbuffer_rx_init(self.sb_rx, cfg.nof_prb) == 0 and R.srslte_softbuffer_tx_init(self.sb_tx, cfg.nof_prb) == 0
        self.pc = np.zeros(L["srslte_pmch_cfg_t"], np.uint8)
        g = self.pc

        def u32(off, v):
            g[off:off + 4].view(np.uint32)[0] = v
        g[L["srslte_pdsch_grant_t.prb_idx"]:L["srslte_pdsch_grant_t.prb_idx"] + 220].reshape(2, 110)[:, :cfg.nof_prb] = 1
        u32(L["srslte_pdsch_grant_t.nof_prb"], cfg.nof_prb)
        u32(L["srslte_pdsch_grant_t.nof_re"], cfg.nof_re)
        u32(L["srslte_pdsch_grant_t.nof_tb"], 1)
        u32(L["srslte_pdsch_grant_t.nof_layers"], 1)
        o = L["srslte_pdsch_grant_t.tb"]
        u32(o + L["srslte_ra_tb_t.mod"], cfg.mod)
        u32(o + L["srslte_ra_tb_t.tbs"], cfg.tbs)
        u32(o + L["srslte_ra_tb_t.nof_bits"], cfg.nbits)
        g[o + L["srslte_ra_tb_t.enabled"]] = 1
        u32(L["srslte_pdsch_cfg_t.max_nof_iterations"], cfg.max_iter)
        g[L["srslte_pmch_cfg_t.area_id"]:L["srslte_pmch_cfg_t.area_id"] + 2].view(np.uint16)[0] = cfg.area_id
        self.sb_off = L["srslte_pdsch_cfg_t.softbuffers"]

    def encode(self, data, tti):
        """payload -> the resource grid after srslte_pmch_encode (the MBSFN reference signals are not part of it)"""
        cfg, R = self.cfg, self.R
        self.sf.tti = tti
        self.pc[self.sb_off:self.sb_off + 8].view(np.uint64)[0] = C.addressof(self.sb_tx)  # a union of tx[] and rx[] pointers (pdsch_cfg.h:65-68)
        R.srslte_softbuffer_tx_reset(self.sb_tx)
        d = np.zeros(cfg.tbs // 8 + 64, np.uint8)
        d[:cfg.tbs // 8] = data
        grid = self.aligned(2 * self.glen, np.float32)
        grid[:] = 0
        assert R.srslte_pmch_encode(self.q, C.byref(self.sf), p(self.pc), p(d), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)) == 0
        return grid.view(np.complex64)[:cfg.grid_len].copy()

    def decode(self, iq, tti):
        cfg, R, L = self.cfg, self.R, self.L
        nrx = cfg.nof_rx
        q = cfg.ofdm(False)
        grids = [self.aligned(2 * self.glen, np.float32) for _ in range(nrx)]
        iq2 = np.ascontiguousarray(iq, np.complex64).reshape(nrx, cfg.sf_len)
        for a_ in range(nrx):
            grids[a_][:] = 0
            oracle().orc_ofdm_rx_sf(C.byref(q), p(iq2[a_]), p(grids[a_]))
        self.sf.tti = tti
        inp = (C.c_void_p * 4)(*([g_.ctypes.data for g_ in grids] + [0] * (4 - nrx)))
        assert R.srslte_chest_dl_estimate_cfg(self.chest, C.byref(self.sf), C.byref(self.rc), inp, C.byref(self.res)) == 0
        return self._decode(inp, grids)

    def _decode(self, inp, grids):
        cfg, R, L = self.cfg, self.R, self.L
        self.pc[self.sb_off:self.sb_off + 8].view(np.uint64)[0] = C.addressof(self.sb_rx)
        R.srslte_softbuffer_rx_reset(self.sb_rx)
        payload = np.zeros(cfg.tbs // 8 + 64, np.uint8)
        data = np.zeros(2 * L["srslte_pdsch_res_t"], np.uint8)
        data[:8].view(np.uint64)[0] = payload.ctypes.data
        assert R.srslte_pmch_decode(self.q, C.byref(self.sf), p(self.pc), C.byref(self.res), inp, p(data)) == 0

        def ptr(name, dtype, count):
            addr = np.frombuffer(self.q, np.uint64, 1, L["srslte_pmch_t." + name])[0]
            return np.frombuffer(C.string_at(int(addr), count * np.dtype(dtype).itemsize), dtype).copy()
        return {"tb": payload[:cfg.tbs // 8 + 3].copy(), "ok": bool(data[L["srslte_pdsch_res_t.crc"]]), "d": ptr("d", np.complex64, cfg.nof_re),
                "e": ptr("e", np.int16, cfg.nbits), "noise": self.res.noise_estimate, "ce": [c_.view(np.complex64)[:cfg.grid_len].copy() for c_ in self.ces]}


class RefPdschTx:
    """The reference's own srslte_pdsch_encode (pdsch.c:1059-1185, eNB object): TB -> DL-SCH coding -> scrambling -> modulation -> layer
    mapping / SFBC precoding -> RE mapping, one resource grid per port (without CRS: srslte_enb_dl_put_base adds them). Pins the
    stimulus generator make_subframe builds from oracle pieces. p_a = 0: rho_a = 1 for one port, sqrt(2) for two (pdsch.c:525)."""

    def __init__(self, cfg):
        from _libs import RefCell, RefDlSfCfg, aligned, opaque, ref, ref_layout
        R = self.R = ref()
        self.cfg, self.aligned = cfg, aligned
        L = self.L = ref_layout({"srslte_pdsch_t": [], "srslte_pdsch_cfg_t": ["rnti", "softbuffers"],
                                 "srslte_pdsch_grant_t": ["tx_scheme", "pmi", "prb_idx", "nof_prb", "nof_re", "nof_symb_slot", "tb", "nof_tb", "nof_layers"],
                                 "srslte_ra_tb_t": ["mod", "tbs", "rv", "nof_bits", "cw_idx", "enabled"], "srslte_softbuffer_tx_t": []},
                                ["srslte/phy/phch/pdsch.h"])
        cell = RefCell(cfg.nof_prb, cfg.nof_ports, cfg.cell_id, 1 if cfg.cp_ext else 0, 0, 0, 0)
        self.q = opaque(L["srslte_pdsch_t"] + 64)
        assert R.srslte_pdsch_init_enb(self.q, cfg.nof_prb) == 0 and R.srslte_pdsch_set_cell(self.q, cell) == 0
        R.srslte_pdsch_set_rnti.argtypes = [C.c_void_p, C.c_uint16]
        assert R.srslte_pdsch_set_rnti(self.q, cfg.rnti) == 0
        self.sb = opaque(L["srslte_softbuffer_tx_t"] + 64)
        assert R.srslte_softbuffer_tx_init(self.sb, cfg.nof_prb) == 0
        self.pc = g = np.zeros(L["srslte_pdsch_cfg_t"], np.uint8)

        def u32(off, v):
            g[off:off + 4].view(np.uint32)[0] = v
        u32(L["srslte_pdsch_grant_t.tx_scheme"], {"cdd": 3, "mux": 2}[cfg.tx_scheme] if cfg.tx_scheme else (1 if cfg.nof_ports > 1 else 0))
        u32(L["srslte_pdsch_grant_t.pmi"], cfg.pmi)
        g[L["srslte_pdsch_grant_t.prb_idx"]:L["srslte_pdsch_grant_t.prb_idx"] + 220].reshape(2, 110)[:, :cfg.nof_prb] = 1 if cfg.prb_mask is None else cfg.prb_mask
        u32(L["srslte_pdsch_grant_t.nof_prb"], cfg.nof_prb if cfg.prb_mask is None else int(cfg.prb_mask[0].sum()))
        u32(L["srslte_pdsch_grant_t.nof_symb_slot"], cfg.nsym // 2)
        u32(L["srslte_pdsch_grant_t.nof_symb_slot"] + 4, cfg.nsym // 2)
        u32(L["srslte_pdsch_grant_t.nof_tb"], cfg.nof_tb)
        u32(L["srslte_pdsch_grant_t.nof_layers"], cfg.nof_tb if cfg.tx_scheme else cfg.nof_ports)
        self.tb0 = L["srslte_pdsch_grant_t.tb"]
        for cw in range(cfg.nof_tb):
            o = self.tb0 + cw * L["srslte_ra_tb_t"]
            u32(o + L["srslte_ra_tb_t.mod"], cfg.mods[cw])
            u32(o + L["srslte_ra_tb_t.tbs"], cfg.tbss[cw])
            u32(o + L["srslte_ra_tb_t.cw_idx"], cw)
            g[o + L["srslte_ra_tb_t.enabled"]] = 1
        g[L["srslte_pdsch_cfg_t.rnti"]:L["srslte_pdsch_cfg_t.rnti"] + 2].view(np.uint16)[0] = cfg.rnti
        g[L["srslte_pdsch_cfg_t.softbuffers"]:L["srslte_pdsch_cfg_t.softbuffers"] + 8].view(np.uint64)[0] = C.addressof(self.sb)
        if cfg.nof_tb == 2:  # softbuffers.tx[1]
            self.sb1 = opaque(L["srslte_softbuffer_tx_t"] + 64)
            assert R.srslte_softbuffer_tx_init(self.sb1, cfg.nof_prb) == 0
            g[L["srslte_pdsch_cfg_t.softbuffers"] + 8:L["srslte_pdsch_cfg_t.softbuffers"] + 16].view(np.uint64)[0] = C.addressof(self.sb1)
        self.u32, self.sf = u32, RefDlSfCfg()

    def run_mimo(self, data, tti):
        """Two-layer modes: data = [payload per transport block] -> one resource grid per port (rv 0)."""
        cfg, R, L = self.cfg, self.R, self.L
        nre = len(cfg.indices(tti % 10))
        self.u32(L["srslte_pdsch_grant_t.nof_re"], nre)
        self.sf.tti, self.sf.cfi = tti, cfg.cfi
        bufs = []
        for cw in range(cfg.nof_tb):
            o = self.tb0 + cw * L["srslte_ra_tb_t"]
            self.u32(o + L["srslte_ra_tb_t.nof_bits"], nre * MOD_BITS[cfg.mods[cw]])
            self.u32(o + L["srslte_ra_tb_t.rv"], 0)
            R.srslte_softbuffer_tx_reset(self.sb1 if cw else self.sb)
            d = np.zeros(cfg.tbss[cw] // 8 + 64, np.uint8)
            d[:cfg.tbss[cw] // 8] = data[cw]
            bufs.append(d)
        dp = (C.c_void_p * 2)(*([b_.ctypes.data for b_ in bufs] + [0] * (2 - cfg.nof_tb)))
        grids = [self.aligned(2 * cfg.grid_len, np.float32) for _ in range(2)]
        gp = (C.c_void_p * 4)(grids[0].ctypes.data, grids[1].ctypes.data, 0, 0)
        assert R.srslte_pdsch_encode(self.q, C.byref(self.sf), p(self.pc), dp, gp) == 0
        return [g_.view(np.complex64).copy() for g_ in grids]

    def run(self, data, tti, rv=0):
        cfg, R, L = self.cfg, self.R, self.L
        nre = len(cfg.indices(tti % 10))
        self.u32(L["srslte_pdsch_grant_t.nof_re"], nre)
        self.u32(self.tb0 + L["srslte_ra_tb_t.nof_bits"], nre * cfg.Qm)
        self.u32(self.tb0 + L["srslte_ra_tb_t.rv"], rv)
        self.sf.tti, self.sf.cfi = tti, cfg.cfi
        R.srslte_softbuffer_tx_reset(self.sb)
        d = np.zeros(cfg.tbs // 8 + 64, np.uint8)
        d[:cfg.tbs // 8] = data
        dp = (C.c_void_p * 2)(d.ctypes.data, 0)
        for v in ((0, rv) if rv else (0,)):  # a retransmission reads the circular buffer the rv 0 encode left in the soft buffer (sch.c:183-297)
            self.u32(self.tb0 + L["srslte_ra_tb_t.rv"], v)
            grids = [self.aligned(2 * cfg.grid_len, np.float32) for _ in range(cfg.nof_ports)]
            gp = (C.c_void_p * 4)(*([g_.ctypes.data for g_ in grids] + [0] * (4 - cfg.nof_ports)))
            assert R.srslte_pdsch_encode(self.q, C.byref(self.sf), p(self.pc), dp, gp) == 0
        return [g_.view(np.complex64).copy() for g_ in grids]


def ref_sch_decode(self, e, nbits):
    """decode_tb / decode_tb_cb (sch.c:299-500) on the reference's rm_turbo / tdec / crc objects held by `self` (a RefRx or RefUlRx)."""
    cfg, R = self.cfg, self.R
    llr8 = getattr(cfg, "llr8", False)
    lt = np.int8 if llr8 else np.int16
    if True:
        s = cfg.seg
        tb = np.zeros(cfg.tbs // 8 + 16, np.uint8)
        iters, all_ok = np.zeros(s.C, np.uint32), True
        qm = getattr(cfg, "Qm_sch", cfg.Qm)  # Qm * N_L (sch.c:507-531)
        Gp = nbits // qm
        gamma, n_e = Gp % s.C, qm * (Gp // s.C)
        for cb in range(s.C):
            K = s.K1 if cb < s.C1 else s.K2
            rlen = K if s.C == 1 else K - 24
            rp, n_e2 = cb * n_e, n_e
            if cb > s.C - gamma:
                n_e2 = n_e + qm
                rp = (s.C - gamma) * n_e + (cb - (s.C - gamma)) * n_e2
            w = self.aligned(3 * (K + 32) + 12 + 64, lt)
            ein = self.aligned(n_e2 + 64, lt)
            ein[:n_e2] = e[rp:rp + n_e2]
            assert (R.srslte_rm_turbo_rx_lut_8bit if llr8 else R.srslte_rm_turbo_rx_lut)(p(ein), p(w), n_e2, R.srslte_cbsegm_cbindex(K), 0) == 0
            assert R.srslte_tdec_new_cb(self.tdec, K) == 0
            out = tb[cb * rlen // 8:]
            ok, noi = False, 0
            while noi < cfg.max_iter and not ok:
                (R.srslte_tdec_iteration_8bit if llr8 else R.srslte_tdec_iteration)(self.tdec, p(w), p(out))
                noi += 1
                if s.C > 1:
                    ok = R.srslte_crc_checksum_byte(self.crc_cb, p(out), K) == 0
                else:
                    ok = R.srslte_crc_checksum_byte(self.crc_tb, p(out), cfg.tbs + 24) == 0
            iters[cb] = noi
            all_ok = all_ok and ok
        if all_ok:
            par_rx = R.srslte_crc_checksum_byte(self.crc_tb, p(tb), cfg.tbs)
            par_tx = (int(tb[cfg.tbs // 8]) << 16) | (int(tb[cfg.tbs // 8 + 1]) << 8) | int(tb[cfg.tbs // 8 + 2])
            all_ok = par_rx == par_tx and par_rx != 0
        return {"tb": tb[:cfg.tbs // 8 + 3], "ok": all_ok, "iters": iters}


# ------------------------------------------------------------------------------------------------------------------ UL
class UlConfig:
    """One PUSCH configuration (the UCI goes in per subframe); tbs = 0 is a PUSCH without UL-SCH data (CQI-only, sch.c:943-975); cp_ext: a cell
    with the extended CP (6 symbols per slot, DMRS in symbol 2 of each slot)."""

    def __init__(self, nof_prb, cell_id, mod, tbs, L_prb, n_prb=0, n_dmrs=0, rnti=0x1234, max_iter=6, cyclic_shift=0, delta_ss=0, n_prb_slot1=None,
                 group_hopping=False, sequence_hopping=False, shortened=False, cp_ext=False):
        from _libs import OrcUlDmrs, OrcUlDmrsCfg
        self.nof_prb, self.cell_id, self.mod, self.tbs, self.L_prb, self.n_prb, self.n_dmrs = nof_prb, cell_id, mod, tbs, L_prb, n_prb, n_dmrs
        # srslte_pusch_grant_t.n_prb[2] / n_prb_tilde[2]: the PRB offset of each slot (intra-subframe hopping when they differ)
        self.n_prbs = (n_prb, n_prb if n_prb_slot1 is None else n_prb_slot1)
        self.rnti, self.max_iter = rnti, max_iter
        self.Qm = MOD_BITS[mod]
        self.nre = 12 * nof_prb
        self.cp_ext, self.nsl = cp_ext, 6 if cp_ext else 7
        self.grid_len = 2 * self.nsl * self.nre
        self.M_sc = 12 * L_prb
        self.nsymb = 2 * (self.nsl - 1) - (1 if shortened else 0)  # data symbols: 2 (N_symb - 1) - N_srs (ra_ul.c:234, pusch.c:52-91); the SRS takes the last one
        self.nof_re = self.nsymb * self.M_sc
        self.nbits = self.nof_re * self.Qm
        self.N = oracle().orc_symbol_sz(nof_prb)
        self.sf_len = 15 * self.N
        self.dmrs_cfg = OrcUlDmrsCfg(cyclic_shift, delta_ss, group_hopping, sequence_hopping)
        self.dmrs = OrcUlDmrs()
        assert oracle().orc_ul_dmrs_init_cp(C.byref(self.dmrs), cell_id, self.nsl) == 0
        self.dmrs_syms = (self.nsl - 4, 2 * self.nsl - 4)  # SRSLTE_REFSIGNAL_UL_L (refsignal_ul.h:43): 3, 10 / 2, 8
        self.seg = OrcCbsegm()
        assert oracle().orc_cbsegm(C.byref(self.seg), tbs) == 0 and self.seg.F == 0
        self.data_syms = [l for l in range(2 * self.nsl) if l not in self.dmrs_syms][:self.nsymb]
        # UL channel interleaver without UCI (36.212 5.2.2.8, sch.c:580-598,:891-913): q[(i*R + j)*Qm + k] = g[(j*12 + i)*Qm + k]
        j, i, k = np.meshgrid(np.arange(self.M_sc), np.arange(self.nsymb), np.arange(self.Qm), indexing="ij")
        self.q_of_g = ((i * self.M_sc + j) * self.Qm + k).reshape(-1)  # g index (j, i, k) row-major -> q index

    def r_dmrs(self, sf_idx):
        r = np.zeros(2 * self.M_sc, np.complex64)
        assert oracle().orc_ul_dmrs_pusch_gen(C.byref(self.dmrs), C.byref(self.dmrs_cfg), self.L_prb, sf_idx, self.n_dmrs, p(r)) == 0
        return r

    def scramble(self, sf_idx):
        c = np.zeros(self.nbits, np.uint8)
        oracle().orc_gold(C.c_uint32(oracle().orc_pdsch_cinit(self.rnti, 0, sf_idx, self.cell_id)), self.nbits, p(c))  # sequences.c:65-67
        return c


def ack_type_map(cfg, ack, Qp):
    """Bookkeeping for tests, per q position: -1 no ACK bit, 0 / 1 value bit, 2 repetition of the previous bit, 3 placeholder - the positions of
    uci_ulsch_interleave_ack_gen (uci.c:497-520) and the pattern of encode_ri_ack (:573-602), derived from orc_uci_ack_insert itself by
    inserting into constant streams (scrambling off: value bits show their value, placeholders 1, repetitions their predecessor)."""
    orc = oracle()
    zeros = np.zeros(cfg.nbits, np.uint8)
    O = len(ack)
    a = np.array(list(ack) + [0], np.uint8)[:2]

    def run(fill, av):
        q = np.full(cfg.nbits, fill, np.uint8)
        assert orc.orc_uci_ack_insert(p(q), p(zeros), p(np.ascontiguousarray(av, np.uint8)), O, cfg.Qm, cfg.nof_re, cfg.nsymb, Qp) == 0
        return q
    r0, r1 = run(0, a), run(1, a)
    touched = (r0 != 0) | (r1 != 1) | (run(0, a ^ 1) != 0) | (run(0, a ^ np.array([1, 0], np.uint8)) != 0)
    flips = (r0 != run(0, a ^ np.array([1, 0], np.uint8))) | (r0 != run(0, a ^ np.array([0, 1], np.uint8)))  # follows an ACK value
    types = np.full(cfg.nbits, -1, np.int8)
    types[touched & ~flips] = 3                   # constant 1: placeholder
    for q_i in np.nonzero(touched & flips)[0]:    # a repetition bit directly follows its value bit inside the symbol (1-bit ACK only)
        types[q_i] = r0[q_i] if ((q_i % cfg.Qm) == 0 or O == 2) else 2
    return types


def ul_ack_ri_qprime(cfg, O, I_offset, is_ri, O_cqi=0, I_offset_cqi=0):
    """Q' of a HARQ-ACK / rank indication (0 without one); on a PUSCH without UL-SCH data the CQI report's size and offset enter (uci.c:557-564)"""
    orc = oracle()
    if not O:
        return 0
    if cfg.tbs == 0:
        return orc.orc_uci_ack_ri_qprime_nodata(O, I_offset, 1 if is_ri else 0, O_cqi, I_offset_cqi, cfg.L_prb, cfg.nsymb)
    return (orc.orc_uci_ri_qprime if is_ri else orc.orc_uci_ack_qprime)(O, I_offset, cfg.L_prb, cfg.nsymb, cfg.seg.C * cfg.seg.K1)


def ul_ri_layout(cfg, O_ri, I_offset_ri, O_cqi=0, I_offset_cqi=0):
    """(Q'_ri, lut, RI mask, G): the channel interleaver with the rank-indication symbols left out (ulsch_interleave_gen, sch.c:580-598)."""
    orc = oracle()
    Qp = ul_ack_ri_qprime(cfg, O_ri, I_offset_ri, True, O_cqi, I_offset_cqi)
    assert Qp >= 0
    lut = np.zeros(cfg.nbits, np.uint32)
    G = orc.orc_ulsch_interleaver_lut(cfg.Qm, cfg.nof_re, cfg.nsymb, Qp, p(lut))
    mask = (lut == 0) & (np.arange(cfg.nbits) != 0)
    assert G == cfg.nbits - Qp * cfg.Qm == cfg.nbits - int(mask.sum())
    return Qp, lut, mask, G


def ul_cqi_qprime(cfg, O_cqi, I_offset_cqi, Qp_ri):
    Qp = oracle().orc_uci_cqi_qprime(O_cqi, I_offset_cqi, cfg.L_prb, cfg.nsymb, cfg.seg.C * cfg.seg.K1, Qp_ri) if O_cqi else 0
    assert Qp >= 0
    return Qp


def make_ul_subframe(cfg, tti, rng, snr_db=None, amp=1.0, gain=1.0 + 0j, data=None, keep=None, ack=(), I_offset_ack=0, ri=(), I_offset_ri=0, cqi=(),
                     I_offset_cqi=0, rv=0):
    """UE transmit side of pusch.c:314-421 (UL-SCH, optionally with 1-2 HARQ-ACK bits, a 1-2 bit rank indication and / or a CQI report of
    `cqi` bits multiplexed): returns (iq[sf_len], payload bytes); keep: dict that receives g, d, z, grid (and q_tx, ack_types with an ACK)."""
    orc = oracle()
    sf_idx = tti % 10
    if data is None:
        data = rng.integers(0, 256, cfg.tbs // 8, dtype=np.uint8)
    Qp_ri, lut, ri_mask, G = ul_ri_layout(cfg, len(ri), I_offset_ri, len(cqi), I_offset_cqi)
    # the CQI's coded bits come first in the stream the interleaver reads, the UL-SCH is rate-matched to the rest (sch.c:1133-1160)
    Qp_cqi = ul_cqi_qprime(cfg, len(cqi), I_offset_cqi, Qp_ri)
    n_cqi = Qp_cqi * cfg.Qm
    sch = OrcSchCfg(cfg.tbs, G - n_cqi, cfg.Qm, rv, cfg.max_iter)  # rv: srslte_pusch_grant_t.tb.rv of a HARQ retransmission
    g = np.zeros(G, np.uint8)
    if n_cqi:
        qc = np.zeros(n_cqi, np.uint8)
        assert orc.orc_uci_cqi_encode(p(np.ascontiguousarray(cqi, np.uint8)), len(cqi), p(qc), n_cqi) == 0
        g[:n_cqi] = qc
    if cfg.tbs:
        gs = np.zeros(G - n_cqi, np.uint8)
        assert orc.orc_dlsch_encode(C.byref(sch), p(data), p(gs)) == 0  # UL-SCH data path = segmentation + coder + rate matching (sch.c:1068-1160)
        g[n_cqi:] = gs
    else:  # no UL-SCH (sch.c:1157: cb_segm.tbs == 0): the report fills what the rank indication leaves (Q_prime_cqi with K = 0, uci.c:266-281)
        assert n_cqi == G and len(cqi)
    q = np.zeros(cfg.nbits, np.uint8)
    q[~ri_mask] = g[lut[~ri_mask]]
    if not len(ri):
        assert np.array_equal(np.flatnonzero(~ri_mask)[np.argsort(lut[~ri_mask])], cfg.q_of_g)
    c = cfg.scramble(sf_idx)
    q ^= c
    if len(ri):  # RI symbols in the places the interleaver left free (sch.c:1110-1129)
        r2 = np.array(list(ri) + [0], np.uint8)[:2]
        assert Qp_ri > 0 and orc.orc_uci_ri_insert(p(q), p(c), p(r2), len(ri), cfg.Qm, cfg.nof_re, cfg.nsymb, Qp_ri) == 0
    if len(ack):  # HARQ-ACK symbols overwrite UL-SCH symbols next to the DMRS (36.212 5.2.2.6-5.2.2.8; orc_uci.c)
        Qp = ul_ack_ri_qprime(cfg, len(ack), I_offset_ack, False, len(cqi), I_offset_cqi)
        a2 = np.array(list(ack) + [0], np.uint8)[:2]
        assert Qp > 0 and orc.orc_uci_ack_insert(p(q), p(c), p(a2), len(ack), cfg.Qm, cfg.nof_re, cfg.nsymb, Qp) == 0
        if keep is not None:
            keep.update(q_tx=q.copy(), ack_types=ack_type_map(cfg, ack, Qp))
    d = np.zeros(cfg.nof_re, np.complex64)
    orc.orc_modulate(cfg.mod, p(q), p(d), cfg.nbits)
    z = np.zeros_like(d)
    orc.orc_dft_precoding(p(d), p(z), cfg.L_prb, cfg.nsymb, 1, True)
    grid = np.zeros(cfg.grid_len, np.complex64)
    for n, l in enumerate(cfg.data_syms):  # pusch_cp (pusch.c:52-91): slot l // N_symb at its own offset n_prb_tilde[slot]
        o = l * cfg.nre + 12 * cfg.n_prbs[l // cfg.nsl]
        grid[o:o + cfg.M_sc] = z[n * cfg.M_sc:(n + 1) * cfg.M_sc]
    r = cfg.r_dmrs(sf_idx)
    for s_, l in enumerate(cfg.dmrs_syms):  # srslte_refsignal_dmrs_pusch_put (refsignal_ul.c:316-330)
        o = l * cfg.nre + 12 * cfg.n_prbs[s_]
        grid[o:o + cfg.M_sc] = r[s_ * cfg.M_sc:(s_ + 1) * cfg.M_sc]
    tx = OrcOfdm()
    orc.orc_ofdm_init(C.byref(tx), cfg.nof_prb, not cfg.cp_ext)
    tx.normalize, tx.freq_shift, tx.freq_shift_f = True, True, 0.5  # ue_ul.c:63-64
    iq = np.zeros(cfg.sf_len, np.complex64)
    orc.orc_ofdm_tx_sf(C.byref(tx), p(grid), p(iq))
    if keep is not None:
        keep.update(g=g, d=d, z=z, grid=grid)
    iq = iq * np.complex64(amp * gain)
    if snr_db is not None:
        sigma = np.sqrt(amp * amp * abs(gain) ** 2 * cfg.M_sc / cfg.N / 2) * 10 ** (-snr_db / 20)
        iq = iq + (sigma * (rng.standard_normal(cfg.sf_len) + 1j * rng.standard_normal(cfg.sf_len))).astype(np.complex64)
    return iq.astype(np.complex64), data


def oracle_ul_rx(cfg, iq, tti, keep=False, O_ack=0, I_offset_ack=0, O_ri=0, I_offset_ri=0, O_cqi=0, I_offset_cqi=0, harq=None, rv=0, new_data=True):
    """eNB receive side: enb_ul.c:58-63 OFDM settings, srslte_chest_ul_estimate_pusch, srslte_pusch_decode (pusch.c:423-520) and the
    UL-SCH part of srslte_ulsch_decode (sch.c:991-1066) without UCI."""
    from _libs import OrcChestUlRes
    orc = oracle()
    sf_idx = tti % 10
    rxo = OrcOfdm()
    orc.orc_ofdm_init(C.byref(rxo), cfg.nof_prb, not cfg.cp_ext)
    rxo.normalize, rxo.freq_shift, rxo.freq_shift_f = False, True, -0.5
    grid = np.zeros(cfg.grid_len, np.complex64)
    orc.orc_ofdm_rx_sf(C.byref(rxo), p(np.ascontiguousarray(iq, np.complex64)), p(grid))
    ce, res = np.zeros(cfg.grid_len, np.complex64), OrcChestUlRes()
    assert orc.orc_chest_ul_pusch_hop_cp(p(cfg.r_dmrs(sf_idx)), cfg.nof_prb, cfg.L_prb, cfg.n_prbs[0], cfg.n_prbs[1], cfg.nsl, p(grid), p(ce),
                                         C.byref(res)) == 0
    sel = np.concatenate([np.arange(l * cfg.nre + 12 * cfg.n_prbs[l // cfg.nsl], l * cfg.nre + 12 * cfg.n_prbs[l // cfg.nsl] + cfg.M_sc) for l in cfg.data_syms])
    y, h = np.ascontiguousarray(grid[sel]), np.ascontiguousarray(ce[sel])
    z, d = np.zeros(cfg.nof_re, np.complex64), np.zeros(cfg.nof_re, np.complex64)
    orc.orc_predecoding_single(p(y), p(h), p(z), cfg.nof_re, 1.0, res.noise_estimate)
    orc.orc_dft_precoding(p(z), p(d), cfg.L_prb, cfg.nsymb, 0, True)
    qllr = np.zeros(cfg.nbits, np.int16)
    orc.orc_demod_soft_s(cfg.mod, p(d), p(qllr), cfg.nof_re)
    c_seq = cfg.scramble(sf_idx)
    orc.orc_scramble_s(p(qllr), p(c_seq), cfg.nbits)
    q_before_ack = qllr.copy()
    ack_out = np.zeros(2, np.uint8)
    if O_ack:  # uci_decode_ri_ack (sch.c:929-966): ACK decisions from the interleaved LLRs, then those positions are zeroed
        Qp = ul_ack_ri_qprime(cfg, O_ack, I_offset_ack, False, O_cqi, I_offset_cqi)
        assert Qp > 0 and orc.orc_uci_ack_extract(p(qllr), p(c_seq), p(ack_out), O_ack, cfg.Qm, cfg.nof_re, cfg.nsymb, Qp) == 0
    ri_out = np.zeros(2, np.uint8)
    Qp_ri, lut, ri_mask, G = ul_ri_layout(cfg, O_ri, I_offset_ri, O_cqi, I_offset_cqi)
    if O_ri:  # after the ACK (sch.c:968-979); the RI LLRs stay in q, and the scatter below leaves the last of them in g[0] (sch.c:891-918)
        assert orc.orc_uci_ri_extract(p(qllr), p(c_seq), p(ri_out), O_ri, cfg.Qm, cfg.nof_re, cfg.nsymb, Qp_ri) == 0
    g_full = np.zeros(cfg.nbits, np.int16)
    orc.orc_ulsch_deinterleave(p(qllr), p(lut), p(g_full), cfg.nbits)  # ulsch_deinterleave: g[lut[i]] = q[i]
    g = np.ascontiguousarray(g_full[:G])
    if not O_ri:
        assert np.array_equal(g, qllr[cfg.q_of_g])
    # the CQI report sits in front of the UL-SCH (sch.c:1031-1060)
    n_cqi = ul_cqi_qprime(cfg, O_cqi, I_offset_cqi, Qp_ri) * cfg.Qm
    cqi_out, cqi_ok = np.zeros(64, np.uint8), C.c_uint8(0)
    if O_cqi:
        assert orc.orc_uci_cqi_decode(p(g[:n_cqi].copy()), n_cqi, O_cqi, p(cqi_out), C.byref(cqi_ok)) == 0
    sch = OrcSchCfg(cfg.tbs, G - n_cqi, cfg.Qm, rv, cfg.max_iter)
    tb, iters, cbok = np.zeros(cfg.tbs // 8 + 16, np.uint8), np.zeros(cfg.seg.C, np.uint32), np.zeros(cfg.seg.C, np.uint8)
    if cfg.tbs == 0:  # no UL-SCH to decode (sch.c:1062-1065)
        rc = -1
    elif harq is not None:  # an OrcHarq kept between the transmissions of one transport block (cfg->softbuffers.rx, sch.c:1063)
        rc = orc.orc_dlsch_decode_harq(C.byref(sch), p(np.ascontiguousarray(g[n_cqi:])), 0, 1 if new_data else 0, p(harq.w), p(harq.crc), p(harq.data),
                                       p(tb), p(iters), p(cbok))
    else:
        rc = orc.orc_dlsch_decode(C.byref(sch), p(np.ascontiguousarray(g[n_cqi:])), p(tb), p(iters), p(cbok))
    out = {"tb": tb[:cfg.tbs // 8 + 3], "ok": rc == 0, "iters": iters, "cb_ok": cbok, "cqi": cqi_out[:O_cqi], "cqi_ok": bool(cqi_ok.value)}
    if keep:
        out.update(grid=grid, ce=ce, noise=res.noise_estimate, z=z, d=d, q=qllr, g=g, res=res, q_before_ack=q_before_ack, ack=ack_out, ri=ri_out)
    return out


class RefUlRx:
    """eNB PUSCH receive chain on the REFERENCE's compiled code for every stage it holds (srslte_chest_ul_estimate_pusch,
    srslte_predecoding_single, srslte_demod_soft_demodulate_s, rm_turbo / tdec / crc); the FFT and the transform de-precoding are the
    oracle's (FFTW is absent), RE extraction, descrambling and the UL deinterleaver are numpy one-liners."""

    def __init__(self, cfg):
        from _libs import RefCell, RefChestUlRes, aligned, opaque, ref, ref_pusch_cfg
        self.R = ref()
        assert self.R is not None, "oracle/_ref/libsrslte_ref.so is not built"
        R = self.R
        self.cfg, self.aligned = cfg, aligned
        self.chest = opaque(1 << 16)
        assert R.srslte_chest_ul_init(self.chest, cfg.nof_prb) == 0
        assert R.srslte_chest_ul_set_cell(self.chest, RefCell(cfg.nof_prb, 1, cfg.cell_id, 1 if cfg.cp_ext else 0, 0, 0, 0)) == 0
        R.srslte_chest_ul_pregen(self.chest, C.byref(cfg.dmrs_cfg))
        self.pcfg = ref_pusch_cfg(cfg.L_prb, cfg.n_prb, cfg.n_dmrs, cfg.n_prbs[1])
        self.res = RefChestUlRes()
        self.ce = aligned(2 * cfg.grid_len, np.float32)
        self.res.ce = self.ce.ctypes.data
        self.tdec = opaque(1 << 20)
        assert R.srslte_tdec_init(self.tdec, 6144) == 0
        self.crc_tb, self.crc_cb = opaque(4096), opaque(4096)
        R.srslte_crc_init(self.crc_tb, 0x1864CFB, 24)
        R.srslte_crc_init(self.crc_cb, 0x1800063, 24)
        R.srslte_crc_checksum_byte.restype = C.c_uint32
        R.srslte_cbsegm_cbindex.restype = C.c_int
        self.q = OrcOfdm()
        oracle().orc_ofdm_init(C.byref(self.q), cfg.nof_prb, not cfg.cp_ext)
        self.q.normalize, self.q.freq_shift, self.q.freq_shift_f = False, True, -0.5
        self.sel = np.concatenate([np.arange(l * cfg.nre + 12 * cfg.n_prbs[l // cfg.nsl], l * cfg.nre + 12 * cfg.n_prbs[l // cfg.nsl] + cfg.M_sc) for l in cfg.data_syms])

    def run(self, iq, tti):
        from _libs import ref_ul_sf_cfg
        cfg, R, orc = self.cfg, self.R, oracle()
        grid = self.aligned(2 * cfg.grid_len, np.float32)
        orc.orc_ofdm_rx_sf(C.byref(self.q), p(np.ascontiguousarray(iq, np.complex64)), p(grid))
        self.ce[:] = 0
        assert R.srslte_chest_ul_estimate_pusch(self.chest, ref_ul_sf_cfg(tti), self.pcfg, p(grid), C.byref(self.res)) == 0
        n = cfg.nof_re
        y, h, z = self.aligned(2 * n, np.float32), self.aligned(2 * n, np.float32), self.aligned(2 * n, np.float32)
        y.view(np.complex64)[:] = grid.view(np.complex64)[self.sel]
        h.view(np.complex64)[:] = self.ce.view(np.complex64)[self.sel]
        R.srslte_predecoding_single(p(y), p(h), p(z), None, n, 1.0, self.res.noise_estimate)
        d = self.aligned(2 * n, np.float32)
        orc.orc_dft_precoding(p(z), p(d), cfg.L_prb, cfg.nsymb, 0, True)
        q = self.aligned(cfg.nbits + 64, np.int16)
        R.srslte_demod_soft_demodulate_s(cfg.mod, p(d), p(q), n)
        qv = q[:cfg.nbits]
        qv[cfg.scramble(tti % 10).astype(bool)] *= -1
        g = np.ascontiguousarray(qv[cfg.q_of_g])
        return ref_sch_decode(self, g, cfg.nbits)


class RefUlsch:
    """The reference's own srslte_ulsch_encode / srslte_ulsch_decode (sch.c:991-1160: UL-SCH coding, the UL channel interleaver of
    36.212 5.2.2.8 and its inverse, UCI absent) on its compiled code, with a hand-filled srslte_pusch_cfg_t (offsets from the reference
    headers at run time). Pins the UL side of the oracle chain: orc_dlsch_encode/decode used as UL-SCH coder and UlConfig.q_of_g."""

    def __init__(self, cfg, O_ack=0, I_offset_ack=0, O_ri=0, I_offset_ri=0, cqi_N=None, I_offset_cqi=0):
        """cqi_N: None = no CQI report; 0 = wide-band report (4 bits); N > 0 = higher-layer configured sub-band report, 4 + 2 N bits."""
        from _libs import opaque, ref, ref_layout
        R = self.R = ref()
        self.cfg = cfg
        L = self.L = ref_layout({"srslte_sch_t": [], "srslte_pusch_cfg_t": ["grant", "max_nof_iterations", "softbuffers", "uci_cfg", "uci_offset"],
                                 "srslte_uci_cfg_t": ["ack", "cqi"], "srslte_uci_cfg_ack_t": ["nof_acks"], "srslte_cqi_cfg_t": ["ri_len", "data_enable", "N", "type"],
                                 "srslte_uci_offset_cfg_t": ["I_offset_ack", "I_offset_ri", "I_offset_cqi"],
                                 "srslte_uci_value_t": ["ack", "ri", "cqi"], "srslte_uci_value_ack_t": ["ack_value"],
                                 "srslte_cqi_value_t": ["data_crc", "wideband.wideband_cqi", "subband_hl.wideband_cqi_cw0", "subband_hl.subband_diff_cqi_cw0"],
                                 "srslte_pusch_grant_t": ["L_prb", "nof_re", "nof_symb", "tb"],
                                 "srslte_ra_tb_t": ["mod", "tbs", "rv", "nof_bits", "enabled"],
                                 "srslte_softbuffer_rx_t": [], "srslte_softbuffer_tx_t": []}, ["srslte/phy/ch_estimation/chest_ul.h", "srslte/phy/phch/pusch.h"])
        self.q = opaque(L["srslte_sch_t"] + 64)
        assert R.srslte_sch_init(self.q) == 0
        R.srslte_sch_set_max_noi.argtypes = [C.c_void_p, C.c_uint32]
        R.srslte_sch_set_max_noi(self.q, cfg.max_iter)
        self.sb_rx, self.sb_tx = opaque(L["srslte_softbuffer_rx_t"] + 64), opaque(L["srslte_softbuffer_tx_t"] + 64)
        assert R.srslte_softbuffer_rx_init(self.sb_rx, cfg.nof_prb) == 0 and R.srslte_softbuffer_tx_init(self.sb_tx, cfg.nof_prb) == 0
        self.pc = np.zeros(L["srslte_pusch_cfg_t"], np.uint8)
        g0, t0 = L["srslte_pusch_cfg_t.grant"], L["srslte_pusch_cfg_t.grant"] + L["srslte_pusch_grant_t.tb"]

        def u32(off, v):
            self.pc[off:off + 4].view(np.uint32)[0] = v
        u32(g0 + L["srslte_pusch_grant_t.L_prb"], cfg.L_prb)
        u32(g0 + L["srslte_pusch_grant_t.nof_re"], cfg.nof_re)
        u32(g0 + L["srslte_pusch_grant_t.nof_symb"], cfg.nsymb)
        u32(t0 + L["srslte_ra_tb_t.mod"], cfg.mod)
        u32(t0 + L["srslte_ra_tb_t.tbs"], cfg.tbs)
        u32(t0 + L["srslte_ra_tb_t.nof_bits"], cfg.nbits)
        self.pc[t0 + L["srslte_ra_tb_t.enabled"]] = 1
        u32(L["srslte_pusch_cfg_t.max_nof_iterations"], cfg.max_iter)
        u32(L["srslte_pusch_cfg_t.uci_cfg"] + L["srslte_uci_cfg_t.ack"] + L["srslte_uci_cfg_ack_t.nof_acks"], O_ack)  # HARQ-ACK bits of carrier 0
        u32(L["srslte_pusch_cfg_t.uci_offset"] + L["srslte_uci_offset_cfg_t.I_offset_ack"], I_offset_ack)
        u32(L["srslte_pusch_cfg_t.uci_cfg"] + L["srslte_uci_cfg_t.cqi"] + L["srslte_cqi_cfg_t.ri_len"], O_ri)
        u32(L["srslte_pusch_cfg_t.uci_offset"] + L["srslte_uci_offset_cfg_t.I_offset_ri"], I_offset_ri)
        self.cqi_N = cqi_N
        if cqi_N is not None:
            c0 = L["srslte_pusch_cfg_t.uci_cfg"] + L["srslte_uci_cfg_t.cqi"]
            self.pc[c0 + L["srslte_cqi_cfg_t.data_enable"]] = 1
            u32(c0 + L["srslte_cqi_cfg_t.N"], cqi_N)
            u32(c0 + L["srslte_cqi_cfg_t.type"], 3 if cqi_N else 0)  # SRSLTE_CQI_TYPE_SUBBAND_HL / _WIDEBAND (cqi.h:113-118)
            u32(L["srslte_pusch_cfg_t.uci_offset"] + L["srslte_uci_offset_cfg_t.I_offset_cqi"], I_offset_cqi)
        self.cqi_val = L["srslte_uci_value_t.cqi"]
        self.ri_off = L["srslte_uci_value_t.ri"]
        self.ack_off = L["srslte_uci_value_t.ack"] + L["srslte_uci_value_ack_t.ack_value"]
        self.sb_off = L["srslte_pusch_cfg_t.softbuffers"]
        self.rv_off = t0 + L["srslte_ra_tb_t.rv"]

    def cqi_bits(self, wb, diff=0):
        """the report's bits as srslte_cqi_value_pack lays them out (cqi.c:41-76,:100-134): 4-bit wide-band CQI, then 2 N bits of sub-band
        differentials, MSB first"""
        N = self.cqi_N
        return np.array([(wb >> (3 - i)) & 1 for i in range(4)] + [(diff >> (2 * N - 1 - i)) & 1 for i in range(2 * N)], np.uint8)

    def _put_cqi(self, uci, wb, diff):
        L = self.L
        if self.cqi_N:
            uci[self.cqi_val + L["srslte_cqi_value_t.subband_hl.wideband_cqi_cw0"]] = wb
            uci[self.cqi_val + L["srslte_cqi_value_t.subband_hl.subband_diff_cqi_cw0"]:][:4].view(np.uint32)[0] = diff
        else:
            uci[self.cqi_val + L["srslte_cqi_value_t.wideband.wideband_cqi"]] = wb

    def encode(self, data, ack=(), ri=None, cqi=None, rv=0):
        """payload bytes (+ HARQ-ACK values) -> (g bits, q bits) one per element, as srslte_pusch_encode gets them before scrambling
        (pusch.c:380-395); the ACK positions of q hold the value bits, 0 for repetition / placeholder bits. rv: grant.tb.rv."""
        cfg, R = self.cfg, self.R
        self.pc[self.rv_off:self.rv_off + 4].view(np.uint32)[0] = rv
        self.pc[self.sb_off:self.sb_off + 8].view(np.uint64)[0] = C.addressof(self.sb_tx)
        if rv == 0:  # a retransmission re-reads the coded bits the rv 0 call left in the soft buffer (encode_tb_off, sch.c:228-262)
            R.srslte_softbuffer_tx_reset(self.sb_tx)
        d = np.zeros(cfg.tbs // 8 + 64, np.uint8)
        d[:cfg.tbs // 8] = data
        uci = np.zeros(4096, np.uint8)
        uci[self.ack_off:self.ack_off + len(ack)] = ack
        if ri is not None:
            uci[self.ri_off] = ri
        if cqi is not None:
            self._put_cqi(uci, *cqi)
        g, q = np.zeros(cfg.nbits // 8 + 64, np.uint8), np.zeros(cfg.nbits // 8 + 64, np.uint8)
        assert R.srslte_ulsch_encode(self.q, p(self.pc), p(d), p(uci), p(g), p(q)) >= 0  # returns the number of RI/ACK q-bits
        return np.unpackbits(g)[:cfg.nbits], np.unpackbits(q)[:cfg.nbits]

    def decode(self, q_llr, c_seq, rv=0, new_data=True):
        """descrambled int16 LLRs in received (q) order -> {tb, ok}, as srslte_pusch_decode calls it (pusch.c:497-503). new_data False: a
        retransmission with redundancy version rv into the soft buffer the previous calls left (no srslte_softbuffer_rx_reset)."""
        cfg, R = self.cfg, self.R
        self.pc[self.rv_off:self.rv_off + 4].view(np.uint32)[0] = rv
        self.pc[self.sb_off:self.sb_off + 8].view(np.uint64)[0] = C.addressof(self.sb_rx)
        if new_data:
            R.srslte_softbuffer_rx_reset(self.sb_rx)
        ql = np.zeros(cfg.nbits + 64, np.int16)
        ql[:cfg.nbits] = q_llr
        gl = np.zeros(cfg.nbits + 64, np.int16)
        cs = np.ascontiguousarray(c_seq, np.uint8)
        tb, uci = np.zeros(cfg.tbs // 8 + 64, np.uint8), np.zeros(4096, np.uint8)
        rc = R.srslte_ulsch_decode(self.q, p(self.pc), p(ql), p(gl), p(cs), p(tb), p(uci))
        out = {"tb": tb[:cfg.tbs // 8 + 3].copy(), "ok": rc == 0, "g": gl[:cfg.nbits].copy(), "ack": uci[self.ack_off:self.ack_off + 2].copy(),
               "ri": int(uci[self.ri_off])}
        if self.cqi_N is not None:
            L = self.L
            out["cqi_crc"] = bool(uci[self.cqi_val + L["srslte_cqi_value_t.data_crc"]])
            if self.cqi_N:
                out["cqi"] = (int(uci[self.cqi_val + L["srslte_cqi_value_t.subband_hl.wideband_cqi_cw0"]]),
                              int(uci[self.cqi_val + L["srslte_cqi_value_t.subband_hl.subband_diff_cqi_cw0"]:][:4].view(np.uint32)[0]))
            else:
                out["cqi"] = (int(uci[self.cqi_val + L["srslte_cqi_value_t.wideband.wideband_cqi"]]), 0)
        return out
