#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE (oracle/_ref/libsrslte_ref.so, built by oracle/ref.mk from the
sources under /root/reference) and from the known-answer data the reference's own test holds
(lib/src/phy/fec/test/turbodecoder_test.h:75-160, parsed as data). Run in the authoring container only:

    make -C oracle ref && python tests/gen_golden.py

A fixture is inputs + expected outputs; no reference source text is stored. The seeded inputs are regenerated here, the
expected outputs come from reference code. tests/test_oracle_golden.py pins the oracle to them on any machine and
tests/test_gpu_golden.py checks the HIP path against the same files.
"""
import ctypes as C
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _libs import (RefCell, RefChestCfg, RefChestRes, RefDlSfCfg, OrcCell, acopy, aligned, opaque, oracle, p, ref)  # noqa: E402
from lte_sim import DlConfig, RefRx, make_subframe  # noqa: E402

OUT = os.path.join(HERE, "golden")
REF_TEST_H = "/root/reference/lib/src/phy/fec/test/turbodecoder_test.h"


def kat():
    src = open(REF_TEST_H).read()

    def arr(name):
        m = re.search(name + r"\[[^\]]*\]\s*=\s*\{([^}]*)\}", src)
        return np.array([int(x) for x in m.group(1).split(",") if x.strip()], np.uint8)

    return arr("known_data"), arr("known_data_encoded")


def main():
    R, orc = ref(), oracle()
    assert R is not None, "build oracle/_ref first"
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    R.srslte_cbsegm_cbindex.restype = C.c_int

    # ---------------- turbo encoder: reference KAT + reference outputs on seeded blocks
    kd, kde = kat()
    tcod = opaque(4096)
    R.srslte_tcod_init(tcod, 6144)
    enc = {"kat_in": kd, "kat_out": kde}
    for K in (40, 176, 504, 1008, 5824, 6144):
        bits = rng.integers(0, 2, K).astype(np.uint8)
        out = np.zeros(3 * K + 12, np.uint8)
        R.srslte_tcod_encode(tcod, p(bits), p(out), K)
        enc["in_%d" % K], enc["out_%d" % K] = bits, out
    chk = np.zeros(3 * 504 + 12, np.uint8)
    R.srslte_tcod_encode(tcod, p(kd), p(chk), 504)
    # The header's codeword is only ever DECODED upstream (turbodecoder_test.c:236-240). The reference's own encoder
    # reproduces it except for the first tail bit (index 3K): recorded here as a fact about the reference.
    assert list(np.nonzero(chk != kde)[0]) == [1512], "reference encoder vs its KAT: unexpected difference set"
    np.savez_compressed(os.path.join(OUT, "tcod.npz"), **enc)

    # ---------------- turbo decoder: hard decisions after each of 6 passes (srslte_tdec_iteration)
    dec = {}
    for K in (40, 176, 504, 1008, 5824, 6144):
        tdec = opaque(1 << 20)
        assert R.srslte_tdec_init(tdec, 6144) == 0
        R.srslte_tdec_force_not_sb(tdec)
        tx = 2.0 * enc["out_%d" % K].astype(np.float64) - 1.0
        llr = acopy((100 * (tx + 10 ** (-1.2 / 20) * rng.standard_normal(tx.shape))).astype(np.int16))
        hard = np.zeros((6, K // 8), np.uint8)
        assert R.srslte_tdec_new_cb(tdec, K) == 0
        for it in range(6):
            R.srslte_tdec_iteration(tdec, p(llr), p(hard[it]))
        dec["llr_%d" % K], dec["hard_%d" % K] = np.array(llr), hard
        R.srslte_tdec_free(tdec)
    # SB layout through the reference's rate de-matcher (sch.c:336-346 -> turbodecoder_iter.h:84-91)
    for K in (816, 5824):
        tdec = opaque(1 << 20)
        assert R.srslte_tdec_init(tdec, 6144) == 0
        n_e = 6726 if K == 5824 else 2040
        bits = rng.integers(0, 2, K).astype(np.uint8)
        e_bits, d = np.zeros(n_e, np.uint8), np.zeros(3 * K + 12, np.uint8)
        R.srslte_tcod_encode(tcod, p(bits), p(d), K)
        orc.orc_rm_turbo_tx_bits(p(d), p(e_bits), n_e, K, 0)
        e = acopy((100 * ((2.0 * e_bits - 1) + 0.8 * rng.standard_normal(n_e))).astype(np.int16))
        w = aligned(3 * (K + 32) + 12 + 64, np.int16)
        assert R.srslte_rm_turbo_rx_lut(p(e), p(w), n_e, R.srslte_cbsegm_cbindex(K), 0) == 0
        w_before = np.array(w)  # the decoder parks the tail LLRs in the pad of this buffer (turbodecoder_iter.h:58-68)
        hard = np.zeros((6, K // 8), np.uint8)
        assert R.srslte_tdec_new_cb(tdec, K) == 0
        for it in range(6):
            R.srslte_tdec_iteration(tdec, p(w), p(hard[it]))
        dec["sb_e_%d" % K], dec["sb_w_%d" % K], dec["sb_hard_%d" % K], dec["sb_bits_%d" % K] = np.array(e), w_before, hard, bits
        R.srslte_tdec_free(tdec)
    np.savez_compressed(os.path.join(OUT, "tdec.npz"), **dec)

    # ---------------- soft demapper
    dm = {}
    for mod, qm in ((0, 1), (1, 2), (2, 4), (3, 6), (4, 8)):
        nsym = 203
        x = acopy(((rng.standard_normal(2 * nsym)) * (0.8 if mod else 1.0)).astype(np.float32))
        dm["sym_%d" % mod] = np.array(x)
        for name, dt in (("", np.float32), ("_s", np.int16), ("_b", np.int8)):
            llr = aligned(nsym * qm + 64, dt)
            getattr(R, "srslte_demod_soft_demodulate" + name)(mod, p(x), p(llr), nsym)
            dm["llr%s_%d" % (name, mod)] = np.array(llr[:nsym * qm])
    np.savez_compressed(os.path.join(OUT, "demod.npz"), **dm)

    # ---------------- channel estimator (chest_test_dl.c-style smooth channel), 6 and 25 PRB, two configurations
    ch = {}
    for prb, cid, sf_idx in ((6, 1, 0), (25, 2, 3)):
        cell = OrcCell(cid, prb, 1, True)
        nre, n = 12 * prb, 14 * 12 * prb
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        orc.orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
        k, l = np.arange(n) % nre, np.arange(n) // nre
        h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
        grid = acopy((g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        for ci, kw in enumerate(({"filter_coef": (4.0, 1.0)}, {"interpolate_subframe": True, "filter_coef": (4.0, 2.0), "cfo_estimate_enable": True})):
            q = opaque(1 << 20)
            assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
            rc = RefChestCfg()
            for kk, v in kw.items():
                if kk == "filter_coef":
                    rc.filter_coef[0], rc.filter_coef[1] = v
                else:
                    setattr(rc, kk, v)
            rc.cfo_estimate_sf_mask = 0x3FF
            ce, res, sf = aligned(2 * n, np.float32), RefChestRes(), RefDlSfCfg()
            res.ce[0][0] = ce.ctypes.data
            sf.tti = sf_idx
            inp = (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)
            assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
            tag = "%d_%d" % (prb, ci)
            ch["grid_%d" % prb] = np.array(grid)
            ch["ce_" + tag] = np.array(ce)
            ch["scal_" + tag] = np.array([res.noise_estimate, res.noise_estimate_dbm, res.snr_db, res.rsrp, res.rsrp_dbm, res.rsrq, res.rsrq_db,
                                          res.rssi_dbm, res.cfo], np.float32)
            ch["meta_" + tag] = np.array([prb, cid, sf_idx, ci], np.int32)
            R.srslte_chest_dl_free(q)
    np.savez_compressed(os.path.join(OUT, "chest.npz"), **ch)

    # ---------------- whole receive chain on reference code: cfg1-like (6 PRB QPSK) and one cfg2 subframe (100 PRB 64QAM)
    e2e = {}
    for tag, (prb, mod, tbs, snr, ttis) in {"cfg1": (6, 1, 936, 4.0, (1, 2, 3)), "cfg2": (100, 3, 75376, 18.0, (0,))}.items():
        cfg = DlConfig(prb, 1, mod, tbs)
        chain = RefRx(cfg)
        for t in ttis:
            iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
            r = chain.run(iq, t)
            e2e["%s_iq_%d" % (tag, t)] = iq.astype(np.complex64)
            e2e["%s_tb_%d" % (tag, t)] = r["tb"].copy()
            e2e["%s_ok_%d" % (tag, t)] = np.array([r["ok"]], np.uint8)
            e2e["%s_iters_%d" % (tag, t)] = r["iters"].copy()
            e2e["%s_data_%d" % (tag, t)] = data
    np.savez_compressed(os.path.join(OUT, "dl_chain.npz"), **e2e)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


def extra():
    """Fixtures added after the first set; each draws from its own generator so the files above stay byte-identical."""
    from _libs import make_crc
    R = ref()
    assert R is not None, "build oracle/_ref first"
    # ---------------- byte encoder with fused CRC attachment (srslte_tcod_encode_lut, turbocoder.c:189-367, as sch.c:260 drives it)
    rng = np.random.default_rng(2026100301)
    tcod = opaque(4096)
    R.srslte_tcod_init(tcod, 6144)
    lut = {}
    for n, (idx, with_cb, last, tb_init) in enumerate(((0, False, False, 0), (17, True, False, 0x123456), (59, True, True, 0xabcdef),
                                                       (100, False, True, 0x5a5a5a), (183, True, False, 0), (187, True, True, 0x00ff00))):
        K = R.srslte_cbsegm_cbsize(idx)
        data = rng.integers(0, 256, K // 8 + 1).astype(np.uint8)
        buf, par = data.copy(), np.zeros(K // 4 + 2, np.uint8)
        crc_tb, crc_cb = make_crc(0x1864CFB, 24), make_crc(0x1800063, 24)
        crc_tb.crcinit = tb_init
        assert R.srslte_tcod_encode_lut(tcod, C.byref(crc_tb), C.byref(crc_cb) if with_cb else None, p(buf), p(par), idx, last) == 3 * K + 12
        lut["meta_%d" % n] = np.array([idx, K, with_cb, last, tb_init, crc_tb.crcinit], np.uint64)
        lut["in_%d" % n], lut["sys_%d" % n], lut["par_%d" % n] = data, buf, par[: K // 4 + 1]
    np.savez_compressed(os.path.join(OUT, "tcod_lut.npz"), **lut)
    print("tcod_lut.npz", os.path.getsize(os.path.join(OUT, "tcod_lut.npz")), "bytes")

    # ---------------- 8-bit LLR path: per-pass hard decisions of srslte_tdec_iteration_8bit (sse8: K=1008, avx8: K=2112,
    # 6144; widening fall-back: K=504) and the whole receive chain with q->llr_is_8bit semantics (pdsch.c:760-779, sch.c:336-356)
    rng = np.random.default_rng(2026100302)
    g8 = {}
    for K in (504, 1008, 2112, 6144):
        bits = rng.integers(0, 2, K).astype(np.uint8)
        enc = np.zeros(3 * K + 12, np.uint8)
        R.srslte_tcod_encode(tcod, p(bits), p(enc), K)
        llr = acopy((22 * ((2.0 * enc - 1) + 0.85 * rng.standard_normal(enc.shape))).clip(-128, 127).astype(np.int8))
        tdec = opaque(1 << 20)
        assert R.srslte_tdec_init(tdec, 6144) == 0 and R.srslte_tdec_new_cb(tdec, K) == 0
        R.srslte_tdec_force_not_sb(tdec)
        hard = np.zeros((6, K // 8), np.uint8)
        for it in range(6):
            R.srslte_tdec_iteration_8bit(tdec, p(llr), p(hard[it]))
        g8["llr_%d" % K], g8["hard_%d" % K], g8["bits_%d" % K] = np.array(llr), hard, bits
        R.srslte_tdec_free(tdec)
    for tag, (prb, mod, tbs, snr, ttis) in {"cfg1": (6, 1, 936, 5.0, (1, 2)), "cfg2": (100, 3, 75376, 19.5, (5,))}.items():
        cfg = DlConfig(prb, 1, mod, tbs, llr8=True)
        chain = RefRx(cfg)
        for t in ttis:
            iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
            r = chain.run(iq, t)
            g8["%s_iq_%d" % (tag, t)] = iq.astype(np.complex64)
            g8["%s_tb_%d" % (tag, t)] = r["tb"].copy()
            g8["%s_ok_%d" % (tag, t)] = np.array([r["ok"]], np.uint8)
            g8["%s_iters_%d" % (tag, t)] = r["iters"].copy()
    np.savez_compressed(os.path.join(OUT, "llr8.npz"), **g8)
    print("llr8.npz", os.path.getsize(os.path.join(OUT, "llr8.npz")), "bytes")

    # ---------------- UL: PUSCH DMRS sequences and srslte_chest_ul_estimate_pusch outputs of the reference
    from _libs import OrcUlDmrsCfg, RefChestUlRes, ref_pusch_cfg, ref_ul_sf_cfg
    rng = np.random.default_rng(2026100303)
    ul = {}
    for n, (cell_id, prb, L, n_prb, cs, ds, gh, sh, tti, n_dmrs) in enumerate(((1, 6, 4, 1, 0, 0, 0, 0, 4, 3), (77, 25, 25, 0, 3, 7, 1, 0, 19, 0),
                                                                              (301, 100, 100, 0, 5, 13, 1, 1, 7, 6), (12, 50, 20, 17, 7, 29, 0, 1, 2, 5))):
        q = opaque(1 << 16)
        assert R.srslte_chest_ul_init(q, prb) == 0 and R.srslte_chest_ul_set_cell(q, RefCell(prb, 1, cell_id, 0, 0, 0, 0)) == 0
        dcfg = OrcUlDmrsCfg(cs, ds, bool(gh), bool(sh))
        R.srslte_chest_ul_pregen(q, C.byref(dcfg))
        rs = opaque(1 << 16)
        assert R.srslte_refsignal_ul_init(rs, prb) == 0 and R.srslte_refsignal_ul_set_cell(rs, RefCell(prb, 1, cell_id, 0, 0, 0, 0)) == 0
        r = aligned(2 * 2 * 12 * L, np.float32)
        assert R.srslte_refsignal_dmrs_pusch_gen(rs, C.byref(dcfg), L, tti % 10, n_dmrs, p(r)) == 0
        r = np.array(r).view(np.complex64)
        nre, ng = 12 * prb, 14 * 12 * prb
        grid = (0.5 * (rng.standard_normal(ng) + 1j * rng.standard_normal(ng))).astype(np.complex64)
        k = np.arange(12 * L)
        h = ((1.5 + 0.4 * np.sin(k / 30.0)) * np.exp(1j * (0.4 + k / 150.0))).astype(np.complex64)
        for s_, sym in enumerate((3, 10)):
            grid[sym * nre + 12 * n_prb: sym * nre + 12 * (n_prb + L)] = r[s_ * 12 * L:(s_ + 1) * 12 * L] * h
        grid = acopy((grid + 0.1 * (rng.standard_normal(ng) + 1j * rng.standard_normal(ng))).astype(np.complex64).view(np.float32))
        ce, res = aligned(2 * ng, np.float32), RefChestUlRes()
        ce[:] = 0
        res.ce = ce.ctypes.data
        assert R.srslte_chest_ul_estimate_pusch(q, ref_ul_sf_cfg(tti), ref_pusch_cfg(L, n_prb, n_dmrs), p(grid), C.byref(res)) == 0
        ul["meta_%d" % n] = np.array([cell_id, prb, L, n_prb, cs, ds, gh, sh, tti, n_dmrs], np.int32)
        ul["r_%d" % n], ul["grid_%d" % n] = r.copy(), np.array(grid).view(np.complex64).copy()
        sel = np.concatenate([np.arange(l * nre + 12 * n_prb, l * nre + 12 * (n_prb + L)) for l in range(14)])
        ul["ce_%d" % n] = np.array(ce).view(np.complex64)[sel].copy()  # the granted PRBs of the 14 symbols (everything else stays untouched)
        ul["scal_%d" % n] = np.array([res.noise_estimate, res.noise_estimate_dbm, res.snr, res.snr_db], np.float32)
        R.srslte_chest_ul_free(q)
        R.srslte_refsignal_ul_free(rs)
    np.savez_compressed(os.path.join(OUT, "chest_ul.npz"), **ul)
    print("chest_ul.npz", os.path.getsize(os.path.join(OUT, "chest_ul.npz")), "bytes")

    # ---------------- PUSCH receive chain on reference code (RefUlRx): IQ in, TB / CRC / per-block passes out
    from lte_sim import RefUlRx, UlConfig, make_ul_subframe
    rng = np.random.default_rng(2026100304)
    uc = {}
    for tag, (prb, L, n_prb, mod, tbs, snr, ttis) in {"a": (6, 6, 0, 1, 1000, 4.0, (2, 7)), "b": (25, 10, 5, 2, 4008, 10.0, (9,)),
                                                      "c": (100, 48, 20, 3, 30576, 17.5, (4,))}.items():
        cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True)
        chain = RefUlRx(cfg)
        for t in ttis:
            iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j))
            r = chain.run(iq, t)
            assert r["ok"], "fixture subframes are chosen to decode (a failing block amplifies the reference's 12-bit reciprocal)"
            uc["%s_iq_%d" % (tag, t)], uc["%s_tb_%d" % (tag, t)] = iq.astype(np.complex64), r["tb"].copy()
            uc["%s_iters_%d" % (tag, t)], uc["%s_data_%d" % (tag, t)] = r["iters"].copy(), data
    np.savez_compressed(os.path.join(OUT, "ul_chain.npz"), **uc)
    print("ul_chain.npz", os.path.getsize(os.path.join(OUT, "ul_chain.npz")), "bytes")


def pdsch_function():
    """Outputs of the reference's OWN srslte_pdsch_decode (lte_sim.RefPdsch) for the options added after the first fixtures: 2-port
    transmit diversity, CSI weighting of the LLRs, power scaling, HARQ retransmissions into a kept soft buffer. Small cells only."""
    from lte_sim import DlConfig, RefPdsch, make_subframe
    rng = np.random.default_rng(2026100401)
    out = {}
    cases = {"tm2": dict(prb=6, mod=2, tbs=936, nrx=2, npt=2, csi=True, p_a=None, llr8=False, snr=6.0, seq=((0, 4), (0, 5))),
             "tm1": dict(prb=6, mod=1, tbs=152, nrx=1, npt=1, csi=True, p_a=-3.0, llr8=False, snr=3.0, seq=((0, 0), (0, 7))),
             "harq": dict(prb=6, mod=3, tbs=1736, nrx=1, npt=2, csi=False, p_a=0.0, llr8=False, snr=2.0, seq=((0, 3), (2, 11), (3, 15))),
             "b8": dict(prb=6, mod=2, tbs=936, nrx=1, npt=2, csi=True, p_a=None, llr8=True, snr=9.5, seq=((0, 2), (0, 5)))}
    for tag, c in cases.items():
        cfg = DlConfig(c["prb"], 7, c["mod"], c["tbs"], nof_rx=c["nrx"], nof_ports=c["npt"], llr8=c["llr8"], csi=c["csi"], p_a=c["p_a"])
        chain = RefPdsch(cfg, csi_enable=c["csi"])
        harq = tag == "harq"
        data = None
        for n, (rv, t) in enumerate(c["seq"]):
            iq, data = make_subframe(cfg, t, rng, snr_db=c["snr"], amp=0.1, rv=rv, data=data if harq else None)
            r = chain.run(iq, t, rv=rv, new_data=(n == 0 or not harq))
            out["%s_iq_%d" % (tag, n)] = np.ascontiguousarray(iq, np.complex64)
            out["%s_e_%d" % (tag, n)] = r["e"].copy()        # LLRs after descrambling and CSI weighting, as handed to srslte_dlsch_decode2
            out["%s_ok_%d" % (tag, n)] = np.array([r["ok"]], np.uint8)
            out["%s_tb_%d" % (tag, n)] = r["tb"].copy() if r["ok"] else np.zeros(0, np.uint8)
            out["%s_data_%d" % (tag, n)] = data
        out["%s_meta" % tag] = np.array([c["prb"], c["mod"], c["tbs"], c["nrx"], c["npt"], int(c["csi"]), int(c["llr8"]), int(harq)], np.int32)
        out["%s_pa" % tag] = np.array([np.nan if c["p_a"] is None else c["p_a"]], np.float32)
        out["%s_seq" % tag] = np.array(c["seq"], np.int32)
        print(tag, [int(out["%s_ok_%d" % (tag, n)][0]) for n in range(len(c["seq"]))])
    np.savez_compressed(os.path.join(OUT, "pdsch_function.npz"), **out)
    print("pdsch_function.npz", os.path.getsize(os.path.join(OUT, "pdsch_function.npz")), "bytes")


def chest_mbsfn():
    """Outputs of the reference's srslte_chest_dl_estimate_cfg on MBSFN subframes (after srslte_chest_dl_set_mbsfn_area_id), with the
    applications' configuration (triangle 0.1, PSS noise) and with the REFS noise / no filter: grids in, the 12 estimated symbols and
    the noise estimate out. Stimulus: the oracle's orc_mbsfn_put_sf (itself pinned on srslte_refsignal_mbsfn_put_sf)."""
    R, rng, out = ref(), np.random.default_rng(2026100402), {}
    for tag, (prb, cid, area, sf_idx, ftype, coef, alg) in {"app": (25, 2, 9, 1, 1, 0.1, 1), "refs": (6, 1, 200, 7, 2, 0.0, 0), "tri": (50, 3, 31, 3, 1, 0.2, 0)}.items():
        nre, n = 12 * prb, 14 * 12 * prb
        cell = OrcCell(cid, prb, 1, True)
        q = opaque(1 << 20)
        assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
        assert R.srslte_chest_dl_set_mbsfn_area_id(q, area) == 0
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        assert oracle().orc_mbsfn_put_sf(C.byref(cell), sf_idx, 0, area, p(g)) == 0
        k, l = np.arange(n) % nre, np.arange(n) // nre
        h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
        grid = acopy((g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        rc, res, sf = RefChestCfg(), RefChestRes(), RefDlSfCfg()
        rc.noise_alg, rc.filter_type, rc.interpolate_subframe, rc.mbsfn_area_id = alg, ftype, True, area
        rc.filter_coef[0] = coef
        ce = aligned(2 * n, np.float32)
        res.ce[0][0] = ce.ctypes.data
        sf.tti, sf.sf_type = sf_idx, 1
        assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
        out[tag + "_meta"] = np.array([prb, cid, area, sf_idx, ftype, alg], np.int32)
        out[tag + "_coef"] = np.array([coef], np.float32)
        out[tag + "_grid"] = grid.view(np.complex64).copy()
        out[tag + "_ce"] = ce.view(np.complex64)[:12 * nre].copy()
        out[tag + "_noise"] = np.array([res.noise_estimate], np.float32)
        R.srslte_chest_dl_free(q)
    np.savez_compressed(os.path.join(OUT, "chest_mbsfn.npz"), **out)
    print("chest_mbsfn.npz", os.path.getsize(os.path.join(OUT, "chest_mbsfn.npz")), "bytes")


def pmch():
    """The reference's srslte_pmch_encode (the PMCH's RE of the MBSFN grid) and srslte_pmch_decode with its MBSFN estimate (noise figure, equalised
    symbols, LLRs, transport block, CRC verdict) on time samples made by the oracle's generator (tests/lte_sim.make_pmch_subframe). 12-symbol grids."""
    from lte_sim import PMCH_GOLDEN_CHEST as PMCH_CHEST, PmchConfig, RefPmch, make_pmch_subframe
    rng, out = np.random.default_rng(2026100501), {}
    for tag, (prb, cid, area, mod, tbs, cfi, region, snr, cp_ext, ttis) in {"a": (6, 1, 1, 1, 488, 2, 2, 8.0, True, (1, 8)), "b": (15, 44, 255, 2, 2216, 1, 1, 13.0, False, (3,)),
                                                                           "c": (25, 7, 3, 3, 6200, 2, 2, 20.0, False, (12,))}.items():
        # "b": the applications' smoothing filter (triangle 0.1); the others the Gauss filter of phy_dl_test (upstream warns and computes)
        cfg = PmchConfig(prb, cid, area, mod, tbs, cfi=cfi, non_mbsfn_region=region, cp_ext=cp_ext, chest=PMCH_CHEST[tag])
        chain = RefPmch(cfg)
        out[tag + "_meta"] = np.array([prb, cid, area, mod, tbs, cfi, region, 1 if cp_ext else 0], np.int32)
        out[tag + "_ttis"] = np.array(ttis, np.int32)
        for t in ttis:
            iq, data = make_pmch_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
            g = chain.encode(data, t)
            r = chain.decode(iq, t)
            assert r["ok"], "fixture subframes are chosen to decode"
            out["%s_iq_%d" % (tag, t)], out["%s_data_%d" % (tag, t)] = iq.astype(np.complex64), data
            out["%s_txsym_%d" % (tag, t)] = g[cfg.idx].copy()  # srslte_pmch_encode's symbols, in mapping order
            out["%s_tb_%d" % (tag, t)], out["%s_e_%d" % (tag, t)] = r["tb"].copy(), r["e"].copy()
            out["%s_noise_%d" % (tag, t)] = np.array([r["noise"]], np.float32)
            out["%s_d_%d" % (tag, t)] = r["d"].copy()
    np.savez_compressed(os.path.join(OUT, "pmch.npz"), **out)
    print("pmch.npz", os.path.getsize(os.path.join(OUT, "pmch.npz")), "bytes")


def ul_extcp():
    """Extended-CP cells on the uplink: the reference's DMRS (srslte_refsignal_dmrs_pusch_gen with cell.cp = EXT), srslte_chest_ul_estimate_pusch on
    12-symbol grids, and the PUSCH receive chain on the reference's compiled stages (RefUlRx) on the oracle generator's time samples."""
    from _libs import OrcUlDmrsCfg, RefChestUlRes, ref_pusch_cfg, ref_ul_sf_cfg
    from lte_sim import RefUlRx, UlConfig, make_ul_subframe
    R, rng, out = ref(), np.random.default_rng(2026100502), {}
    for n, (cell_id, prb, L, n0, n1, cs, ds, gh, sh, tti, n_dmrs) in enumerate(((1, 6, 2, 0, 4, 0, 0, 0, 0, 4, 3), (77, 25, 10, 12, 12, 3, 7, 1, 0, 19, 0),
                                                                               (150, 50, 50, 0, 0, 5, 13, 1, 1, 7, 6))):
        cell = RefCell(prb, 1, cell_id, 1, 0, 0, 0)
        q, rs = opaque(1 << 16), opaque(1 << 16)
        assert R.srslte_chest_ul_init(q, prb) == 0 and R.srslte_chest_ul_set_cell(q, cell) == 0
        dcfg = OrcUlDmrsCfg(cs, ds, bool(gh), bool(sh))
        R.srslte_chest_ul_pregen(q, C.byref(dcfg))
        assert R.srslte_refsignal_ul_init(rs, prb) == 0 and R.srslte_refsignal_ul_set_cell(rs, cell) == 0
        r = aligned(2 * 2 * 12 * L, np.float32)
        assert R.srslte_refsignal_dmrs_pusch_gen(rs, C.byref(dcfg), L, tti % 10, n_dmrs, p(r)) == 0
        r = np.array(r).view(np.complex64)
        nre, ng = 12 * prb, 12 * 12 * prb
        grid = (0.5 * (rng.standard_normal(ng) + 1j * rng.standard_normal(ng))).astype(np.complex64)
        k = np.arange(12 * L)
        for s_, (sym, npb) in enumerate(((2, n0), (8, n1))):
            h = ((1.5 - 0.5 * s_ + 0.4 * np.sin(k / 30.0)) * np.exp(1j * (0.4 + s_ + k / 150.0))).astype(np.complex64)
            grid[sym * nre + 12 * npb: sym * nre + 12 * (npb + L)] = r[s_ * 12 * L:(s_ + 1) * 12 * L] * h
        grid = acopy((grid + 0.1 * (rng.standard_normal(ng) + 1j * rng.standard_normal(ng))).astype(np.complex64).view(np.float32))
        ce, res = aligned(2 * ng, np.float32), RefChestUlRes()
        ce[:] = 0
        res.ce = ce.ctypes.data
        assert R.srslte_chest_ul_estimate_pusch(q, ref_ul_sf_cfg(tti), ref_pusch_cfg(L, n0, n_dmrs, n1), p(grid), C.byref(res)) == 0
        out["meta_%d" % n] = np.array([cell_id, prb, L, n0, n1, cs, ds, gh, sh, tti, n_dmrs], np.int32)
        out["r_%d" % n], out["grid_%d" % n] = r.copy(), np.array(grid).view(np.complex64).copy()
        sel = np.concatenate([np.arange(l * nre + 12 * (n0 if l < 6 else n1), l * nre + 12 * ((n0 if l < 6 else n1) + L)) for l in range(12)])
        out["ce_%d" % n] = np.array(ce).view(np.complex64)[sel].copy()
        out["scal_%d" % n] = np.array([res.noise_estimate, res.noise_estimate_dbm, res.snr, res.snr_db], np.float32)
        R.srslte_chest_ul_free(q)
        R.srslte_refsignal_ul_free(rs)
    for tag, (prb, L, n_prb, mod, tbs, snr, short, ttis) in {"a": (6, 6, 0, 1, 808, 5.0, True, (2, 7)), "b": (25, 10, 5, 2, 3240, 10.0, False, (9,))}.items():
        cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, shortened=short, cp_ext=True)
        chain = RefUlRx(cfg)
        out[tag + "_meta"] = np.array([prb, L, n_prb, mod, tbs, 1 if short else 0], np.int32)
        out[tag + "_ttis"] = np.array(ttis, np.int32)
        for t in ttis:
            iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j))
            r = chain.run(iq, t)
            assert r["ok"], "fixture subframes are chosen to decode"
            out["%s_iq_%d" % (tag, t)], out["%s_tb_%d" % (tag, t)] = iq.astype(np.complex64), r["tb"].copy()
            out["%s_iters_%d" % (tag, t)], out["%s_data_%d" % (tag, t)] = r["iters"].copy(), data
    np.savez_compressed(os.path.join(OUT, "ul_extcp.npz"), **out)
    print("ul_extcp.npz", os.path.getsize(os.path.join(OUT, "ul_extcp.npz")), "bytes")


if __name__ == "__main__":
    if "--round4-only" in sys.argv:
        pmch()
        ul_extcp()
        sys.exit(0)
    if "--mbsfn-only" in sys.argv:
        chest_mbsfn()
        sys.exit(0)
    if "--pdsch-only" in sys.argv:
        pdsch_function()
        sys.exit(0)
    if "--extra-only" not in sys.argv:
        main()
    extra()
    pdsch_function()
    chest_mbsfn()
    pmch()
    ul_extcp()
