"""HIP path against the committed golden vectors (reference outputs, tests/gen_golden.py)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hp():
    return importlib.import_module("srslte-emane_amd")


def load(name):
    return np.load(os.path.join(G, name))


def test_tcod_golden(hp):
    g = load("tcod.npz")
    for K in (40, 176, 504, 1008, 5824, 6144):
        rc, out = hp.tcod_encode(g["in_%d" % K], K)
        assert rc == 0 and np.array_equal(out[0], g["out_%d" % K])


def test_tdec_golden(hp):
    g = load("tdec.npz")
    dec = hp.Tdec(6144, 4)
    for K in (40, 176, 504, 1008, 5824, 6144):
        for nit in range(1, 7):
            rc, out, _, _ = dec.run_all(g["llr_%d" % K], K, nit)
            assert rc == 0 and np.array_equal(out[0], g["hard_%d" % K][nit - 1]), (K, nit)
    for K in (816, 5824):
        w = g["sb_w_%d" % K][:3 * (K + 32) + 12]
        for nit in range(1, 7):
            rc, out, _, _ = dec.run_all(w, K, nit, sb_layout=True)
            assert rc == 0 and np.array_equal(out[0], g["sb_hard_%d" % K][nit - 1]), (K, nit)
    kat = load("tcod.npz")
    llr = (100 * (2 * kat["kat_out"].astype(np.int16) - 1)).astype(np.int16)
    rc, out, _, _ = dec.run_all(llr, 504, 2)
    assert np.array_equal(np.unpackbits(out[0]), kat["kat_in"])
    dec.free()


def test_demod_golden(hp):
    g = load("demod.npz")
    for mod in range(5):
        x = g["sym_%d" % mod].view(np.complex64)
        for kind, name in (("f", "llr"), ("s", "llr_s"), ("b", "llr_b")):
            rc, llr = hp.demod_soft_demodulate(mod, x, kind)
            ref = g["%s_%d" % (name, mod)]
            if kind == "f":
                assert np.abs(llr[0] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
            else:
                assert np.array_equal(llr[0], ref), (mod, kind)


def test_chest_golden(hp):
    g = load("chest.npz")
    for tag in ("6_0", "6_1", "25_0", "25_1"):
        prb, cid, sf_idx, ci = [int(v) for v in g["meta_" + tag]]
        cfg = hp.ChestDlCfg()
        if ci == 0:
            cfg.filter_coef[0], cfg.filter_coef[1] = 4.0, 1.0
        else:
            cfg.interpolate_subframe, cfg.cfo_estimate_enable = 1, 1
            cfg.filter_coef[0], cfg.filter_coef[1] = 4.0, 2.0
        est = hp.ChestDl(cid, prb)
        ce, res = est.estimate(g["grid_%d" % prb].view(np.complex64), sf_idx, cfg)
        ref = g["ce_" + tag].view(np.complex64)
        assert np.abs(ce[0] - ref).max() <= 1e-4 * max(np.abs(ref).max(), np.sqrt((np.abs(ref) ** 2).mean()))
        names = ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm", "cfo")
        scal = np.array([res[n][0] for n in names])
        assert np.all(np.abs(scal - g["scal_" + tag]) <= 1e-4 * np.abs(g["scal_" + tag]) + 1e-3)
        est.free()


def test_dl_chain_golden(hp):
    g = load("dl_chain.npz")
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    for tag, prb, mod, tbs, ttis in (("cfg1", 6, 1, 936, (1, 2, 3)), ("cfg2", 100, 3, 75376, (0,))):
        rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, len(ttis), True, hc)
        iq = np.stack([g["%s_iq_%d" % (tag, t)] for t in ttis])
        tb, ok = rx.decode(iq, ttis[0])
        C_ = 1 if prb == 6 else 13
        it = rx.debug(6, np.uint32, len(ttis) * C_).reshape(len(ttis), C_)
        for i, t in enumerate(ttis):
            assert bool(ok[i]) == bool(g["%s_ok_%d" % (tag, t)][0])
            assert np.array_equal(it[i], g["%s_iters_%d" % (tag, t)])
            assert np.array_equal(tb[i], g["%s_tb_%d" % (tag, t)])
        rx.free()


def test_llr8_golden(hp):
    """8-bit LLR path on the device against reference outputs (tests/gen_golden.py:extra)."""
    g = load("llr8.npz")
    dec = hp.Tdec(6144, 2)
    for K in (504, 1008, 2112, 6144):
        for nit in range(1, 7):
            rc, out, _, _ = dec.run_all(g["llr_%d" % K], K, nit, llr8=True)
            assert rc == 0 and np.array_equal(out[0], g["hard_%d" % K][nit - 1]), (K, nit)
    dec.free()
    for tag, prb, mod, tbs, ttis in (("cfg1", 6, 1, 936, (1, 2)), ("cfg2", 100, 3, 75376, (5,))):
        hc = hp.ChestDlCfg()
        hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
        rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, 1, True, hc, llr_8bit=True)
        C_ = 13 if prb == 100 else 1
        for t in ttis:
            tb, ok = rx.decode(g["%s_iq_%d" % (tag, t)][None, :], t)
            assert bool(ok[0]) == bool(g["%s_ok_%d" % (tag, t)][0]) and np.array_equal(tb[0], g["%s_tb_%d" % (tag, t)])
            assert np.array_equal(rx.debug(6, np.uint32, C_), g["%s_iters_%d" % (tag, t)])
        rx.free()


def test_chest_ul_golden(hp):
    """UL DMRS (host table) and chest_ul kernel against reference outputs."""
    g = load("chest_ul.npz")
    for n in range(4):
        cell_id, prb, L, n_prb, cs, ds, gh, sh, tti, n_dmrs = (int(v) for v in g["meta_%d" % n])
        q = hp.ChestUl(cell_id, prb, cs, ds, bool(gh), bool(sh))
        rc, r = q.dmrs(L, tti % 10, n_dmrs)
        assert rc == 0 and np.abs(r - g["r_%d" % n]).max() <= 2e-6
        rc, ce, res = q.estimate_pusch(g["grid_%d" % n], tti, L, n_prb, n_dmrs)
        assert rc == 0
        nre = 12 * prb
        sel = np.concatenate([np.arange(l * nre + 12 * n_prb, l * nre + 12 * (n_prb + L)) for l in range(14)])
        ref = g["ce_%d" % n]
        assert np.abs(ce[0][sel] - ref).max() <= 1e-4 * np.abs(ref).max()
        mask = np.ones(14 * nre, bool)
        mask[sel] = False
        assert np.all(ce[0][mask] == 0)
        for x, y in zip(res[0, :4], g["scal_%d" % n]):
            assert abs(x - y) <= 1e-4 * abs(y) + 1e-6
        assert q.dmrs(2, 0, 0)[0] == 0  # tabulated 1-/2-PRB sequences (36.211 Tables 5.5.1.2-1/-2)
        q.free()


UL_CASES = (("a", 6, 6, 0, 1, 1000, (2, 7)), ("b", 25, 10, 5, 2, 4008, (9,)), ("c", 100, 48, 20, 3, 30576, (4,)))


def test_ul_chain_golden(hp):
    """PUSCH receive chain on the device against the reference-code chain's outputs."""
    g = load("ul_chain.npz")
    for tag, prb, L, n_prb, mod, tbs, ttis in UL_CASES:
        rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, 6, 1, 2, 5, True, False)
        rc, seg = hp.cbsegm(tbs)
        for t in ttis:
            tb, ok = rx.decode(g["%s_iq_%d" % (tag, t)][None, :], t)
            assert ok[0] == 1 and np.array_equal(tb[0], g["%s_tb_%d" % (tag, t)])
            assert np.array_equal(rx.debug(6, np.uint32, seg.C), g["%s_iters_%d" % (tag, t)])
        rx.free()


@pytest.mark.parametrize("tag", ["tm2", "tm1", "harq", "b8"])
def test_pdsch_function_golden(hp, tag):
    """Device pipeline vs outputs of the reference's own srslte_pdsch_decode (tests/gen_golden.py:pdsch_function): 2-port transmit
    diversity, CSI weighting, power scaling, HARQ soft combining into kept soft buffers, 8-bit LLRs: CRC result and transport block
    of every transmission."""
    g = load("pdsch_function.npz")
    prb, mod, tbs, nrx, npt, csi, llr8, harq = [int(v) for v in g[tag + "_meta"]]
    p_a = g[tag + "_pa"][0]
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(7, prb, 1, 0x1234, mod, tbs, 6, 1, True, hc, llr_8bit=bool(llr8), nof_rx=nrx, nof_ports=npt, csi=bool(csi),
                 power_scale=not np.isnan(p_a), p_a=0.0 if np.isnan(p_a) else float(p_a))
    for n, (rv, t) in enumerate(g[tag + "_seq"]):
        tb, ok = rx.decode_harq(g["%s_iq_%d" % (tag, n)][None], int(t), int(rv), n == 0 or not harq)
        assert bool(ok[0]) == bool(g["%s_ok_%d" % (tag, n)][0]), n
        if ok[0]:
            assert np.array_equal(tb[0], g["%s_tb_%d" % (tag, n)]) and np.array_equal(tb[0][:tbs // 8], g["%s_data_%d" % (tag, n)])
    rx.free()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["app", "refs", "tri"])
def test_chest_mbsfn_golden(hp, tag):
    """Device MBSFN estimator vs outputs of the reference's srslte_chest_dl_estimate_cfg (tests/gen_golden.py:chest_mbsfn)."""
    g = load("chest_mbsfn.npz")
    prb, cid, area, sf_idx, ftype, alg = [int(x) for x in g[tag + "_meta"]]
    hc = hp.ChestDlCfg()
    hc.noise_alg, hc.filter_type, hc.interpolate_subframe, hc.mbsfn_area_id = alg, ftype, 1, area
    hc.filter_coef[0] = float(g[tag + "_coef"][0])
    est = hp.ChestDl(cid, prb, 1)
    assert est.set_mbsfn_area_id(area) == 0
    rc, ce, noise = est.estimate_mbsfn(g[tag + "_grid"], sf_idx, hc, 1)
    want = g[tag + "_ce"]
    assert rc == 0 and np.abs(ce[0, 0, 0, :want.size] - want).max() <= 1e-4 * max(np.abs(want).max(), np.sqrt((np.abs(want) ** 2).mean()))
    if alg == 0:
        assert abs(noise[0, 0, 0] - float(g[tag + "_noise"][0])) <= 1e-4 * noise[0, 0, 0]
    est.free()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_pmch_golden(hp, tag):
    """The fused PMCH pipelines vs outputs of the reference's srslte_pmch_encode / srslte_pmch_decode (tests/gen_golden.py:pmch): the transmit side's
    symbols bit for bit, the receive side's noise figure, LLRs (one LSB on <= 0.2 %), CRC verdict and transport block."""
    from lte_sim import PMCH_GOLDEN_CHEST
    g = load("pmch.npz")
    prb, cid, area, mod, tbs, cfi, region, cp_ext = [int(x) for x in g[tag + "_meta"]]
    ttis = [int(x) for x in g[tag + "_ttis"]]
    hc = hp.ChestDlCfg()
    ch = PMCH_GOLDEN_CHEST[tag] or {"filter_coef": (4.0, 1.0)}
    hc.filter_type = ch.get("filter_type", 0)
    hc.filter_coef[0], hc.filter_coef[1] = ch["filter_coef"]
    rx = hp.DlRx(cid, prb, cfi, 0, mod, tbs, 6, 1, True, hc, cp_ext=bool(cp_ext), mbsfn=(area, region))
    tx = hp.DlTx(cid, prb, cfi, 0, mod, tbs, 1, 1, 0.0, cp_ext=bool(cp_ext), mbsfn=(area, region))
    nbits = rx.nof_re(1) * {1: 2, 2: 4, 3: 6}[mod]
    for t in ttis:
        tx.encode(g["%s_data_%d" % (tag, t)][None], t, 0)
        y = tx.debug(2, np.complex64, rx.nof_re(1))
        assert np.array_equal(y.view(np.float32), g["%s_txsym_%d" % (tag, t)].view(np.float32)), t
        tb, ok = rx.decode(g["%s_iq_%d" % (tag, t)][None], t)
        res = rx.debug(2, np.float32, 10)
        want = float(g["%s_noise_%d" % (tag, t)][0])
        assert abs(res[0] - want) <= 1e-4 * want, (t, res[0], want)
        e = rx.debug(4, np.int16, nbits)
        diff = np.abs(e.astype(np.int32) - g["%s_e_%d" % (tag, t)].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 2e-3 * diff.size + 1, (t, int(diff.max()), int((diff != 0).sum()))
        assert ok[0] and np.array_equal(tb[0][:tbs // 8 + 3], g["%s_tb_%d" % (tag, t)]) and np.array_equal(tb[0][:tbs // 8], g["%s_data_%d" % (tag, t)])
    rx.free()
    tx.free()


@pytest.mark.gpu
def test_ul_extended_cp_golden(hp):
    """Extended-CP uplink on the device vs the reference's outputs (tests/gen_golden.py:ul_extcp): DMRS, srslte_chest_ul_estimate_pusch with a PRB offset
    per slot on 12-symbol grids, and the PUSCH receive pipeline (transport blocks and per-block pass counts of the reference-code chain)."""
    g = load("ul_extcp.npz")
    for n in range(3):
        cell_id, prb, L, n0, n1, cs, ds, gh, sh, tti, n_dmrs = [int(x) for x in g["meta_%d" % n]]
        q = hp.ChestUl(cell_id, prb, cs, ds, bool(gh), bool(sh), cp_ext=True)
        rc, r = q.dmrs(L, tti % 10, n_dmrs)
        assert rc == 0 and np.abs(r - g["r_%d" % n]).max() <= 2e-6
        if n0 == n1:  # the wrapper's estimate call takes one PRB offset (hopping: through the pipeline below and tests/test_gpu_ul_extcp.py)
            rc, ce, res = q.estimate_pusch(g["grid_%d" % n][None], tti, L, n0, n_dmrs)
            nre = 12 * prb
            sel = np.concatenate([np.arange(l * nre + 12 * n0, l * nre + 12 * (n0 + L)) for l in range(12)])
            want = g["ce_%d" % n]
            assert rc == 0 and np.abs(ce[0][sel] - want).max() <= 1e-4 * np.abs(want).max()
            for j in range(4):
                x = float(g["scal_%d" % n][j])
                assert abs(res[0, j] - x) <= 1e-4 * abs(x) + 1e-5, j
        q.free()
    for tag in ("a", "b"):
        prb, L, n_prb, mod, tbs, short = [int(x) for x in g[tag + "_meta"]]
        rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, 6, 1, 2, 5, True, False, shortened=bool(short), cp_ext=True)
        C_ = -(-(tbs + 24) // 6120) if tbs + 24 > 6144 else 1
        for t in [int(x) for x in g[tag + "_ttis"]]:
            tb, ok = rx.decode(g["%s_iq_%d" % (tag, t)][None], t)
            it = rx.debug(6, np.uint32, C_)
            assert ok[0] and np.array_equal(tb[0][:tbs // 8 + 3], g["%s_tb_%d" % (tag, t)]) and np.array_equal(it, g["%s_iters_%d" % (tag, t)])
        rx.free()
