"""Pins the oracle to the committed golden vectors (generated from the reference build by tests/gen_golden.py, plus the
reference's own known-answer data). Runs anywhere: needs only gcc-built oracle/liboracle.so."""
import ctypes as C
import os

import numpy as np
import pytest

from _libs import OrcCell, OrcChestCfg, OrcChestRes, acopy, oracle, p
from lte_sim import DlConfig, oracle_rx

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_tcod_golden_and_kat():
    g = load("tcod.npz")
    for K in (40, 176, 504, 1008, 5824, 6144):
        out = np.zeros(3 * K + 12, np.uint8)
        assert oracle().orc_tcod_encode_bits(p(g["in_%d" % K]), p(out), K) == 0
        assert np.array_equal(out, g["out_%d" % K])
    # reference KAT (turbodecoder_test.h:75-160): the codeword matches except the first tail bit, as the reference's own encoder
    out = np.zeros(3 * 504 + 12, np.uint8)
    oracle().orc_tcod_encode_bits(p(g["kat_in"]), p(out), 504)
    assert list(np.nonzero(out != g["kat_out"])[0]) == [1512]
    # ... and decoding the KAT codeword returns the KAT data (what turbodecoder_test -k exercises)
    llr = (100 * (2 * g["kat_out"].astype(np.int16) - 1)).astype(np.int16)
    hard = np.zeros(504 // 8, np.uint8)
    assert oracle().orc_tdec_run(p(llr), False, 504, 2, p(hard), None) == 0
    assert np.array_equal(np.unpackbits(hard), g["kat_in"])


def test_tcod_lut_golden():
    """Byte encoder layout + CB CRC of srslte_tcod_encode_lut's outputs (tests/gen_golden.py:extra)."""
    g = load("tcod_lut.npz")
    oracle().orc_crc_bytes.restype = C.c_uint32
    for n in range(6):
        idx, K, with_cb, last, tb_init, tb_final = (int(v) for v in g["meta_%d" % n])
        sys_, par = g["sys_%d" % n].copy(), np.zeros(K // 4 + 2, np.uint8)
        assert oracle().orc_tcod_encode_bytes(p(sys_), p(par), K) == 3 * K + 12
        assert np.array_equal(sys_, g["sys_%d" % n]) and np.array_equal(par[: K // 4 + 1], g["par_%d" % n])
        if with_cb:  # turbocoder.c:247-258: CRC24B over everything before it
            body = np.ascontiguousarray(sys_[: K // 8 - 3])
            crc = oracle().orc_crc_bytes(0x1800063, 24, p(body), K - 24)
            assert [crc >> 16 & 0xff, crc >> 8 & 0xff, crc & 0xff] == list(sys_[K // 8 - 3: K // 8])


def test_ofdm_mbsfn_and_r2hc_definitions():
    """The oracle's MBSFN symbol layout (ofdm.c:424-437,:558-574) and half-complex DFT against numpy on first principles."""
    from _libs import OrcOfdm
    rng = np.random.default_rng(5)
    for prb, region in ((6, 1), (6, 2), (25, 2)):
        q = OrcOfdm()
        assert oracle().orc_ofdm_init(C.byref(q), prb, False) == 0
        q.exact, q.normalize, q.non_mbsfn_region = True, False, region
        N, nre = q.symbol_sz, 12 * prb
        ext, n0, n1 = -(-512 * N // 2048), -(-160 * N // 2048), -(-144 * N // 2048)
        g = (rng.standard_normal(12 * nre) + 1j * rng.standard_normal(12 * nre)).astype(np.complex64)
        t = np.full(15 * N, 7 + 7j, np.complex64)
        oracle().orc_ofdm_tx_sf(C.byref(q), p(g), p(t))
        pos, exp, touched = 0, np.full(15 * N, 7 + 7j, np.complex64), np.zeros(15 * N, bool)
        for s in range(12):
            if s == region:
                pos += (ext - n0) if region == 1 else (2 * ext - n0 - n1)
            cp = ext if (s >= region) else (n0 if s == 0 else n1)
            X = np.zeros(N, np.complex128)
            X[1: nre // 2 + 1], X[N - nre // 2:] = g[s * nre + nre // 2: (s + 1) * nre], g[s * nre: s * nre + nre // 2]
            x = np.fft.ifft(X) * N
            exp[pos: pos + cp], exp[pos + cp: pos + cp + N] = x[N - cp:], x
            touched[pos: pos + cp + N] = True
            pos += cp + N
        assert pos == 15 * N and np.abs(t - exp).max() < 1e-4 * np.abs(exp).max()
        assert np.all(t[~touched] == 7 + 7j) and (~touched).sum() == ((ext - n0) if region == 1 else (2 * ext - n0 - n1))
        back = np.zeros(12 * nre, np.complex64)
        oracle().orc_ofdm_rx_sf(C.byref(q), p(t), p(back))
        assert np.abs(back / N - g).max() < 1e-4
    for N in (8, 15, 128, 300):
        x = rng.standard_normal(N).astype(np.float32)
        hc, back = np.zeros(N, np.float32), np.zeros(N, np.float32)
        oracle().orc_dft_r2hc(p(x), p(hc), N, 1)
        X = np.fft.fft(x.astype(np.float64))
        exp = np.concatenate([X.real[: N // 2 + 1], X.imag[1: (N + 1) // 2][::-1]])
        assert np.abs(hc - exp).max() < 1e-4 * np.abs(exp).max()
        oracle().orc_dft_r2hc(p(hc), p(back), N, 0)
        assert np.abs(back / N - x).max() < 1e-4


def test_ofdm_standard_rate_symbol_sizes_first_principles():
    """The power-of-two symbol sizes srslte_symbol_sz returns after srslte_use_standard_symbol_size(true) (phy_common.c:304-345: 512 / 1024 /
    1536 / 2048 for 25 / 50 / 75 / 100-110 PRB) through the oracle's modulator and demodulator against numpy on first principles: guards
    (N - 12 nof_prb) / 2, DC skipped, CP lengths ceil(160 N / 2048) and ceil(144 N / 2048) (ofdm.c:43-57,:384-393, phy_common.h:93-99)."""
    from _libs import OrcOfdm
    rng = np.random.default_rng(6)
    for prb, N in ((25, 512), (50, 1024), (75, 1536), (100, 2048), (110, 2048), (6, 128), (15, 256)):
        assert oracle().orc_symbol_sz_power2(prb) == N
        q = OrcOfdm()
        assert oracle().orc_ofdm_init_sz(C.byref(q), prb, N, True) == 0 and q.symbol_sz == N and q.sf_sz == 15 * N
        q.exact, q.normalize = True, False
        nre, n0, n1 = 12 * prb, -(-160 * N // 2048), -(-144 * N // 2048)
        g = (rng.standard_normal(14 * nre) + 1j * rng.standard_normal(14 * nre)).astype(np.complex64)
        t = np.zeros(15 * N, np.complex64)
        oracle().orc_ofdm_tx_sf(C.byref(q), p(g), p(t))
        pos, exp = 0, np.zeros(15 * N, np.complex64)
        for s in range(14):
            cp = n0 if s % 7 == 0 else n1
            X = np.zeros(N, np.complex128)
            X[1: nre // 2 + 1], X[N - nre // 2:] = g[s * nre + nre // 2: (s + 1) * nre], g[s * nre: s * nre + nre // 2]
            x = np.fft.ifft(X) * N
            exp[pos: pos + cp], exp[pos + cp: pos + cp + N] = x[N - cp:], x
            pos += cp + N
        assert pos == 15 * N and np.abs(t - exp).max() < 1e-4 * np.abs(exp).max(), (prb, N)
        back = np.zeros(14 * nre, np.complex64)
        oracle().orc_ofdm_rx_sf(C.byref(q), p(t), p(back))
        assert np.abs(back / N - g).max() < 1e-4, (prb, N)
    q = OrcOfdm()
    assert oracle().orc_ofdm_init_sz(C.byref(q), 100, 1024, True) < 0  # the carriers do not fit


@pytest.mark.parametrize("K", [40, 176, 504, 1008, 5824, 6144])
def test_tdec_golden(K):
    g = load("tdec.npz")
    per = np.zeros((6, K // 8), np.uint8)
    out = np.zeros(K // 8, np.uint8)
    assert oracle().orc_tdec_run(p(g["llr_%d" % K]), False, K, 6, p(out), p(per)) == 0
    assert np.array_equal(per, g["hard_%d" % K])


@pytest.mark.parametrize("K", [816, 5824])
def test_rm_and_tdec_sb_golden(K):
    g = load("tdec.npz")
    e = g["sb_e_%d" % K]
    w = np.zeros(3 * (K + 32) + 12 + 64, np.int16)
    W = oracle().orc_tdec_autoimp_subblocks(K)
    assert oracle().orc_rm_turbo_rx(p(e), p(w), len(e), K, 0, W) == 0
    assert np.array_equal(w, g["sb_w_%d" % K])
    per = np.zeros((6, K // 8), np.uint8)
    out = np.zeros(K // 8, np.uint8)
    assert oracle().orc_tdec_run(p(w), True, K, 6, p(out), p(per)) == 0
    assert np.array_equal(per, g["sb_hard_%d" % K])
    if K == 816:  # this one converges; the K=5824 vector deliberately does not (residual errors exercise all 6 passes)
        assert np.array_equal(np.unpackbits(per[-1]), g["sb_bits_%d" % K])


@pytest.mark.parametrize("mod", [0, 1, 2, 3, 4])
def test_demod_golden(mod):
    g = load("demod.npz")
    x = acopy(g["sym_%d" % mod])
    nsym, qm = len(x) // 2, (1 if mod == 0 else 2 * mod)
    for name, fn, dt in (("llr", "orc_demod_soft_f", np.float32), ("llr_s", "orc_demod_soft_s", np.int16), ("llr_b", "orc_demod_soft_b", np.int8)):
        out = np.zeros(nsym * qm, dt)
        assert getattr(oracle(), fn)(mod, p(x), p(out), nsym) == 0
        assert np.array_equal(out, g["%s_%d" % (name, mod)]), (name, mod)


@pytest.mark.parametrize("tag", ["6_0", "6_1", "25_0", "25_1"])
def test_chest_golden(tag):
    g = load("chest.npz")
    prb, cid, sf_idx, ci = [int(v) for v in g["meta_" + tag]]
    cfg = OrcChestCfg()
    if ci == 0:
        cfg.filter_coef[0], cfg.filter_coef[1] = 4.0, 1.0
    else:
        cfg.interpolate_subframe, cfg.cfo_estimate_enable = True, True
        cfg.filter_coef[0], cfg.filter_coef[1] = 4.0, 2.0
    cell = OrcCell(cid, prb, 1, True)
    ce, res = np.zeros(14 * 12 * prb, np.complex64), OrcChestRes()
    assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(cfg), p(g["grid_%d" % prb]), p(ce), C.byref(res)) == 0
    ref = g["ce_" + tag].view(np.complex64)
    assert np.abs(ce - ref).max() <= 1e-4 * max(np.abs(ref).max(), np.sqrt((np.abs(ref) ** 2).mean()))
    scal = np.array([res.noise_estimate, res.noise_estimate_dbm, res.snr_db, res.rsrp, res.rsrp_dbm, res.rsrq, res.rsrq_db, res.rssi_dbm, res.cfo])
    assert np.all(np.abs(scal - g["scal_" + tag]) <= 1e-4 * np.abs(g["scal_" + tag]) + 1e-6)


@pytest.mark.parametrize("tag,prb,mod,tbs,ttis", [("cfg1", 6, 1, 936, (1, 2, 3)), ("cfg2", 100, 3, 75376, (0,))])
def test_dl_chain_golden(tag, prb, mod, tbs, ttis):
    g = load("dl_chain.npz")
    cfg = DlConfig(prb, 1, mod, tbs)
    for t in ttis:
        r = oracle_rx(cfg, g["%s_iq_%d" % (tag, t)], t)
        assert r["ok"] == bool(g["%s_ok_%d" % (tag, t)][0])
        assert np.array_equal(r["iters"], g["%s_iters_%d" % (tag, t)])
        assert np.array_equal(r["tb"], g["%s_tb_%d" % (tag, t)])


def test_llr8_golden():
    """8-bit LLR path against reference outputs: per-pass decisions of srslte_tdec_iteration_8bit and whole chains."""
    g = load("llr8.npz")
    for K in (504, 1008, 2112, 6144):
        per = np.zeros((6, K // 8), np.uint8)
        assert oracle().orc_tdec_run_8bit(p(g["llr_%d" % K]), False, K, 6, None, p(per)) == 0
        assert np.array_equal(per, g["hard_%d" % K]), K
        assert np.array_equal(np.unpackbits(per[5]), g["bits_%d" % K])
    for tag, prb, mod, tbs, ttis in (("cfg1", 6, 1, 936, (1, 2)), ("cfg2", 100, 3, 75376, (5,))):
        cfg = DlConfig(prb, 1, mod, tbs, llr8=True)
        for t in ttis:
            r = oracle_rx(cfg, g["%s_iq_%d" % (tag, t)], t)
            assert r["ok"] == bool(g["%s_ok_%d" % (tag, t)][0]) and np.array_equal(r["iters"], g["%s_iters_%d" % (tag, t)])
            assert np.array_equal(r["tb"], g["%s_tb_%d" % (tag, t)])


def test_chest_ul_golden():
    """UL DMRS and PUSCH channel estimates against reference outputs (tests/gen_golden.py:extra)."""
    from _libs import OrcChestUlRes, OrcUlDmrs, OrcUlDmrsCfg
    g = load("chest_ul.npz")
    for n in range(4):
        cell_id, prb, L, n_prb, cs, ds, gh, sh, tti, n_dmrs = (int(v) for v in g["meta_%d" % n])
        o = OrcUlDmrs()
        assert oracle().orc_ul_dmrs_init(C.byref(o), cell_id) == 0
        cfg = OrcUlDmrsCfg(cs, ds, bool(gh), bool(sh))
        r = np.zeros(2 * 12 * L, np.complex64)
        assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(cfg), L, tti % 10, n_dmrs, p(r)) == 0
        assert np.abs(r - g["r_%d" % n]).max() <= 2e-6
        nre = 12 * prb
        ce, res = np.zeros(14 * nre, np.complex64), OrcChestUlRes()
        assert oracle().orc_chest_ul_pusch(p(r), prb, L, n_prb, p(np.ascontiguousarray(g["grid_%d" % n])), p(ce), C.byref(res)) == 0
        sel = np.concatenate([np.arange(l * nre + 12 * n_prb, l * nre + 12 * (n_prb + L)) for l in range(14)])
        ref = g["ce_%d" % n]
        assert np.abs(ce[sel] - ref).max() <= 1e-4 * np.abs(ref).max()
        mask = np.ones(14 * nre, bool)
        mask[sel] = False
        assert np.all(ce[mask] == 0)  # nothing outside the grant is written
        for x, y in zip((res.noise_estimate, res.noise_estimate_dbm, res.snr, res.snr_db), g["scal_%d" % n]):
            assert abs(x - y) <= 1e-4 * abs(y) + 1e-6
    small = np.zeros(48, np.complex64)  # 2-PRB grant: tabulated QPSK base sequence, unit modulus
    assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(cfg), 2, 0, 0, p(small)) == 0 and np.allclose(np.abs(small), 1.0, atol=1e-6)


UL_CASES = (("a", 6, 6, 0, 1, 1000, (2, 7)), ("b", 25, 10, 5, 2, 4008, (9,)), ("c", 100, 48, 20, 3, 30576, (4,)))


def test_ul_chain_golden():
    """PUSCH receive chain against the reference-code chain's outputs (tests/gen_golden.py:extra)."""
    from lte_sim import UlConfig, oracle_ul_rx
    g = load("ul_chain.npz")
    for tag, prb, L, n_prb, mod, tbs, ttis in UL_CASES:
        cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True)
        for t in ttis:
            r = oracle_ul_rx(cfg, g["%s_iq_%d" % (tag, t)], t)
            assert r["ok"] and np.array_equal(r["iters"], g["%s_iters_%d" % (tag, t)]) and np.array_equal(r["tb"], g["%s_tb_%d" % (tag, t)])
            assert np.array_equal(r["tb"][:tbs // 8], g["%s_data_%d" % (tag, t)])


@pytest.mark.parametrize("tag", ["tm2", "tm1", "harq", "b8"])
def test_pdsch_function_golden(tag):
    """Oracle chain vs outputs of the reference's own srslte_pdsch_decode (tests/gen_golden.py:pdsch_function): 2-port transmit
    diversity, CSI weighting, power scaling, HARQ soft combining, 8-bit LLRs. The TM1 equaliser of the reference multiplies by an
    approximate reciprocal, so its LLRs may differ by one LSB; CRC results and transport blocks are exact."""
    from lte_sim import DlConfig, OrcHarq, oracle_rx
    g = np.load(os.path.join(G, "pdsch_function.npz"))
    prb, mod, tbs, nrx, npt, csi, llr8, harq = [int(v) for v in g[tag + "_meta"]]
    p_a = None if np.isnan(g[tag + "_pa"][0]) else float(g[tag + "_pa"][0])
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, llr8=bool(llr8), csi=bool(csi), p_a=p_a)
    h = OrcHarq(cfg)
    for n, (rv, t) in enumerate(g[tag + "_seq"]):
        r = oracle_rx(cfg, g["%s_iq_%d" % (tag, n)], int(t), keep=True, harq=h, rv=int(rv), new_data=(n == 0 or not harq))
        diff = np.abs(r["e"].astype(np.int32) - g["%s_e_%d" % (tag, n)].astype(np.int32))
        # (the -Ofast build of the 8-bit weighting multiplies by a hoisted reciprocal in its vector body and divides in its epilogue)
        assert diff.max() <= (0 if npt == 2 and nrx == 1 and not llr8 else 1) and (diff != 0).mean() <= 0.05, (n, diff.max(), (diff != 0).mean())
        assert bool(g["%s_ok_%d" % (tag, n)][0]) == r["ok"], n
        if r["ok"]:
            assert np.array_equal(r["tb"], g["%s_tb_%d" % (tag, n)]) and np.array_equal(r["tb"][:tbs // 8], g["%s_data_%d" % (tag, n)])


@pytest.mark.parametrize("tag", ["app", "refs", "tri"])
def test_chest_mbsfn_golden(tag):
    """Oracle vs outputs of the reference's srslte_chest_dl_estimate_cfg on MBSFN subframes (tests/gen_golden.py:chest_mbsfn)."""
    g = np.load(os.path.join(G, "chest_mbsfn.npz"))
    prb, cid, area, sf_idx, ftype, alg = [int(x) for x in g[tag + "_meta"]]
    orc = oracle()
    orc.orc_chest_dl_mbsfn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    cell, oc = OrcCell(cid, prb, 1, True), OrcChestCfg()
    oc.noise_alg, oc.filter_type, oc.interpolate_subframe = alg, ftype, True
    oc.filter_coef[0] = float(g[tag + "_coef"][0])
    grid, ce, nz = np.ascontiguousarray(g[tag + "_grid"]), np.zeros(14 * 12 * prb, np.complex64), C.c_float(0)
    assert orc.orc_chest_dl_mbsfn(C.byref(cell), sf_idx, C.byref(oc), area, 0, p(grid), p(ce), C.byref(nz)) == 0
    want = g[tag + "_ce"]
    assert np.abs(ce[:want.size] - want).max() <= 1e-4 * max(np.abs(want).max(), np.sqrt((np.abs(want) ** 2).mean()))
    if alg == 0:
        assert abs(nz.value - float(g[tag + "_noise"][0])) <= 1e-4 * nz.value


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_pmch_golden(tag):
    """Oracle PMCH chain vs outputs of the reference's srslte_pmch_encode / srslte_pmch_decode with its MBSFN estimate (tests/gen_golden.py:pmch)."""
    from lte_sim import PMCH_GOLDEN_CHEST as PMCH_CHEST, PmchConfig, make_pmch_subframe, oracle_pmch_rx
    g = np.load(os.path.join(G, "pmch.npz"))
    prb, cid, area, mod, tbs, cfi, region, cp_ext = [int(x) for x in g[tag + "_meta"]]
    cfg = PmchConfig(prb, cid, area, mod, tbs, cfi=cfi, non_mbsfn_region=region, cp_ext=bool(cp_ext), chest=PMCH_CHEST[tag])
    for t in [int(x) for x in g[tag + "_ttis"]]:
        k = {}
        make_pmch_subframe(cfg, t, np.random.default_rng(0), data=g["%s_data_%d" % (tag, t)], keep=k)
        assert np.array_equal(k["d"].view(np.float32), g["%s_txsym_%d" % (tag, t)].view(np.float32)), t  # the transmit side, symbol for symbol
        r = oracle_pmch_rx(cfg, g["%s_iq_%d" % (tag, t)], t, keep=True)
        assert abs(r["noise"] - float(g["%s_noise_%d" % (tag, t)][0])) <= 1e-4 * r["noise"]
        want_d, want_e = g["%s_d_%d" % (tag, t)], g["%s_e_%d" % (tag, t)]
        assert np.abs(r["d"] - want_d).max() <= 2e-4 * np.abs(want_d).max()
        diff = np.abs(r["e"].astype(np.int32) - want_e.astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 2e-3 * diff.size + 1
        assert r["ok"] and np.array_equal(r["tb"], g["%s_tb_%d" % (tag, t)]) and np.array_equal(r["tb"][:tbs // 8], g["%s_data_%d" % (tag, t)])


def test_ul_extended_cp_golden():
    """Oracle vs the reference on an extended-CP cell (tests/gen_golden.py:ul_extcp): DMRS, srslte_chest_ul_estimate_pusch on 12-symbol grids, and the
    PUSCH receive chain on the reference's compiled stages."""
    from _libs import OrcChestUlRes, OrcUlDmrs, OrcUlDmrsCfg
    from lte_sim import UlConfig, oracle_ul_rx
    g = np.load(os.path.join(G, "ul_extcp.npz"))
    orc = oracle()
    orc.orc_chest_ul_pusch_hop_cp.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    for n in range(3):
        cell_id, prb, L, n0, n1, cs, ds, gh, sh, tti, n_dmrs = [int(x) for x in g["meta_%d" % n]]
        o, dcfg = OrcUlDmrs(), OrcUlDmrsCfg(cs, ds, bool(gh), bool(sh))
        assert orc.orc_ul_dmrs_init_cp(C.byref(o), cell_id, 6) == 0
        r = np.zeros(2 * 12 * L, np.complex64)
        assert orc.orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(dcfg), L, tti % 10, n_dmrs, p(r)) == 0
        assert np.abs(r - g["r_%d" % n]).max() <= 2e-6
        nre, ng = 12 * prb, 12 * 12 * prb
        ce, res = np.zeros(ng, np.complex64), OrcChestUlRes()
        assert orc.orc_chest_ul_pusch_hop_cp(p(np.ascontiguousarray(g["r_%d" % n])), prb, L, n0, n1, 6, p(np.ascontiguousarray(g["grid_%d" % n])), p(ce), C.byref(res)) == 0
        sel = np.concatenate([np.arange(l * nre + 12 * (n0 if l < 6 else n1), l * nre + 12 * ((n0 if l < 6 else n1) + L)) for l in range(12)])
        want = g["ce_%d" % n]
        assert np.abs(ce[sel] - want).max() <= 1e-4 * np.abs(want).max()
        for j, nm in enumerate(("noise_estimate", "noise_estimate_dbm", "snr", "snr_db")):
            x = float(g["scal_%d" % n][j])
            assert abs(getattr(res, nm) - x) <= 1e-4 * abs(x) + 1e-6, nm
    for tag in ("a", "b"):
        prb, L, n_prb, mod, tbs, short = [int(x) for x in g[tag + "_meta"]]
        cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, shortened=bool(short), cp_ext=True)
        for t in [int(x) for x in g[tag + "_ttis"]]:
            r = oracle_ul_rx(cfg, g["%s_iq_%d" % (tag, t)], t)
            assert r["ok"] and np.array_equal(r["tb"], g["%s_tb_%d" % (tag, t)]) and np.array_equal(r["iters"], g["%s_iters_%d" % (tag, t)])
            assert np.array_equal(r["tb"][:tbs // 8], g["%s_data_%d" % (tag, t)])
