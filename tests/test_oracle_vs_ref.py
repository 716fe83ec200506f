"""Oracle vs the reference's own compiled code (oracle/_ref/libsrslte_ref.so) on seeded random inputs.
Skipped where the reference build is absent (it only exists where /root/reference was available to `make -C oracle ref`)."""
import ctypes as C

import numpy as np
import pytest

from _libs import (OrcCell, OrcChestCfg, OrcChestRes, RefCell, RefChestCfg, RefChestRes, RefDlSfCfg, acopy, aligned, opaque, oracle, p, ref)
from lte_sim import DlConfig, RefRx, RefUlRx, UlConfig, make_subframe, make_ul_subframe, oracle_rx, oracle_ul_rx

pytestmark = pytest.mark.skipif(ref() is None, reason="oracle/_ref/libsrslte_ref.so not built")
orc = oracle()
ALL_K = list(range(40, 513, 8)) + list(range(528, 1025, 16)) + list(range(1056, 2049, 32)) + list(range(2112, 6145, 64))


@pytest.mark.parametrize("K", [40, 400, 408, 504, 800, 816, 1008, 2048, 2112, 3136, 5824, 6144])
def test_turbo_decoder_8bit_vs_ref(K):
    """srslte_tdec_run_all_8bit (turbodecoder.c:565-593): avx8 (K > 2048), sse8 (K > 800) and the widening fall-backs; plain
    [s p0 p1] input for every K, and the rate-dematcher's SB layout where upstream handles it consistently (K > 800)."""
    R, rng = ref(), np.random.default_rng(8000 + K)
    R.srslte_cbsegm_cbindex.restype = C.c_int
    tcod = opaque(4096)
    R.srslte_tcod_init(tcod, 6144)
    bits = rng.integers(0, 2, K).astype(np.uint8)
    enc = np.zeros(3 * K + 12, np.uint8)
    R.srslte_tcod_encode(tcod, p(bits), p(enc), K)
    assert oracle().orc_tdec_autoimp_subblocks_8bit(K) == R.srslte_tdec_autoimp_get_subblocks_8bit(K)
    for snr, scale in ((1.0, 12), (3.0, 25), (-2.0, 60)):
        llr = acopy((scale * ((2.0 * enc - 1) + 10 ** (-snr / 20) * rng.standard_normal(enc.shape))).clip(-128, 127).astype(np.int8))
        tdec = opaque(1 << 20)
        assert R.srslte_tdec_init(tdec, 6144) == 0
        R.srslte_tdec_force_not_sb(tdec)
        for nit in (1, 2, 3, 6):
            a, b = np.zeros(K // 8, np.uint8), np.zeros(K // 8, np.uint8)
            R.srslte_tdec_run_all_8bit(tdec, p(llr), p(a), nit, K)
            assert oracle().orc_tdec_run_8bit(p(llr), False, K, nit, p(b), None) == 0
            assert np.array_equal(a, b), (K, snr, nit)
        R.srslte_tdec_free(tdec)
    if K > 800:  # SB layout through the reference's 8-bit rate de-matcher (sch.c:336-338)
        n_e = 3 * K + 12 + 500
        W = oracle().orc_tdec_autoimp_subblocks_8bit(K)
        e_bits = np.zeros(n_e, np.uint8)
        oracle().orc_rm_turbo_tx_bits(p(enc), p(e_bits), n_e, K, 0)
        e = acopy((20 * ((2.0 * e_bits - 1) + 0.8 * rng.standard_normal(n_e))).clip(-128, 127).astype(np.int8))
        w, w2 = aligned(3 * (K + 32) + 12 + 64, np.int8), np.zeros(3 * (K + 32) + 12 + 64, np.int8)
        assert R.srslte_rm_turbo_rx_lut_8bit(p(e), p(w), n_e, R.srslte_cbsegm_cbindex(K), 0) == 0
        assert oracle().orc_rm_turbo_rx_8bit(p(e), p(w2), n_e, K, 0, W) == 0
        assert np.array_equal(np.array(w), w2)
        tdec = opaque(1 << 20)
        assert R.srslte_tdec_init(tdec, 6144) == 0 and R.srslte_tdec_new_cb(tdec, K) == 0
        per = np.zeros((6, K // 8), np.uint8)
        for it in range(6):
            hard = np.zeros(K // 8, np.uint8)
            R.srslte_tdec_iteration_8bit(tdec, p(w), p(hard))
            per[it] = hard
        mine = np.zeros((6, K // 8), np.uint8)
        assert oracle().orc_tdec_run_8bit(p(w2), True, K, 6, None, p(mine)) == 0
        assert np.array_equal(per, mine), K
        R.srslte_tdec_free(tdec)


def test_cbsegm_all_tbs_sample():
    from _libs import OrcCbsegm
    R = ref()
    for tbs in list(range(16, 6200, 8))[::7] + [6120, 6144, 6200, 75376, 97896, 149776]:
        a, b = OrcCbsegm(), OrcCbsegm()
        ra, rb = R.srslte_cbsegm(C.byref(a), tbs), oracle().orc_cbsegm(C.byref(b), tbs)
        assert (ra == 0) == (rb == 0)
        if ra == 0:
            assert all(getattr(a, f) == getattr(b, f) for f, _ in OrcCbsegm._fields_), tbs


def test_turbo_encoder_all_188_sizes():
    """turbocoder_test.c:67-129 sweeps all K; here bit-exact vs the reference encoder."""
    R, rng = ref(), np.random.default_rng(1)
    tcod = opaque(4096)
    R.srslte_tcod_init(tcod, 6144)
    for K in ALL_K:
        bits = rng.integers(0, 2, K).astype(np.uint8)
        a, b = np.zeros(3 * K + 12, np.uint8), np.zeros(3 * K + 12, np.uint8)
        R.srslte_tcod_encode(tcod, p(bits), p(a), K)
        assert oracle().orc_tcod_encode_bits(p(bits), p(b), K) == 0
        assert np.array_equal(a, b), K


def test_turbo_encoder_lut_bytes():
    """srslte_tcod_encode_lut (turbocoder.c:189-367) as turbocoder_test.c:103 calls it, plus the CRC-fusing forms sch.c:260 uses:
    packed parity/tail layout of the oracle's byte encoder and make_crc() vs the reference's srslte_crc_init."""
    from _libs import SrslteCrc, make_crc
    R, rng = ref(), np.random.default_rng(2)
    tcod = opaque(4096)
    R.srslte_tcod_init(tcod, 6144)
    mine = make_crc(0x1864CFB, 24)
    theirs = SrslteCrc()
    assert R.srslte_crc_init(C.byref(theirs), 0x1864CFB, 24) == 0
    assert bytes(mine) == bytes(theirs)
    for idx in list(range(0, 188, 9)) + [187]:
        K = R.srslte_cbsegm_cbsize(idx)
        for with_cb, last in ((False, False), (True, False), (True, True), (False, True)):
            if with_cb and last and K < 56:
                continue  # both CRCs do not fit (upstream would index input[-1]); a CB CRC implies C > 1 and a large K
            data = rng.integers(0, 256, K // 8 + 1).astype(np.uint8)
            a_in, b_in = data.copy(), data.copy()
            a_par, b_par = np.zeros(K // 4 + 2, np.uint8), np.zeros(K // 4 + 2, np.uint8)
            crc_tb, crc_cb = make_crc(0x1864CFB, 24), make_crc(0x1800063, 24)
            crc_tb.crcinit = 0x5a5a5a  # a running TB checksum, as in the middle of a transport block
            r = R.srslte_tcod_encode_lut(tcod, C.byref(crc_tb), C.byref(crc_cb) if with_cb else None, p(a_in), p(a_par), idx, last)
            assert r == 3 * K + 12
            # oracle: the caller attaches the CRCs (orc_dlsch_encode does the same), then the byte encoder runs
            b_in[: K // 8] = a_in[: K // 8]
            assert oracle().orc_tcod_encode_bytes(p(b_in), p(b_par), K) == 3 * K + 12
            assert np.array_equal(a_in, b_in) and np.array_equal(a_par[: K // 4 + 1], b_par[: K // 4 + 1]), (K, with_cb, last)


@pytest.mark.parametrize("K", ALL_K[::6] + [400, 408, 800, 816, 6144])
def test_turbo_decoder_vs_ref(K):
    """Both input layouts, 1..6 passes, three SNR/scale points (turbodecoder_test.c:117-311 style stimulus)."""
    R, rng = ref(), np.random.default_rng(K)
    R.srslte_cbsegm_cbindex.restype = C.c_int
    tcod = opaque(4096)
    R.srslte_tcod_init(tcod, 6144)
    bits = rng.integers(0, 2, K).astype(np.uint8)
    enc = np.zeros(3 * K + 12, np.uint8)
    R.srslte_tcod_encode(tcod, p(bits), p(enc), K)
    W = oracle().orc_tdec_autoimp_subblocks(K)
    for snr, scale in ((0.5, 100), (2.0, 100), (-3.0, 3000)):
        llr = acopy((scale * ((2.0 * enc - 1) + 10 ** (-snr / 20) * rng.standard_normal(enc.shape))).clip(-32000, 32000).astype(np.int16))
        tdec = opaque(1 << 20)
        assert R.srslte_tdec_init(tdec, 6144) == 0
        R.srslte_tdec_force_not_sb(tdec)
        for nit in (1, 2, 3, 6):
            a, b = np.zeros(K // 8, np.uint8), np.zeros(K // 8, np.uint8)
            R.srslte_tdec_run_all(tdec, p(llr), p(a), nit, K)
            assert oracle().orc_tdec_run(p(llr), False, K, nit, p(b), None) == 0
            assert np.array_equal(a, b), (K, snr, nit)
        R.srslte_tdec_free(tdec)
        if W:  # SB layout via the reference rate de-matcher
            n_e = (3 * K * 8 // 10) // 6 * 6
            eb = np.zeros(n_e, np.uint8)
            oracle().orc_rm_turbo_tx_bits(p(enc), p(eb), n_e, K, 0)
            e = acopy((scale * ((2.0 * eb - 1) + 10 ** (-snr / 20) * rng.standard_normal(n_e))).clip(-32000, 32000).astype(np.int16))
            w1, w2 = aligned(3 * (K + 32) + 76, np.int16), aligned(3 * (K + 32) + 76, np.int16)
            R.srslte_rm_turbo_rx_lut(p(e), p(w1), n_e, R.srslte_cbsegm_cbindex(K), 0)
            oracle().orc_rm_turbo_rx(p(e), p(w2), n_e, K, 0, W)
            assert np.array_equal(w1, w2)
            tdec = opaque(1 << 20)
            assert R.srslte_tdec_init(tdec, 6144) == 0
            for nit in (1, 4, 6):
                a, b = np.zeros(K // 8, np.uint8), np.zeros(K // 8, np.uint8)
                R.srslte_tdec_run_all(tdec, p(acopy(w1)), p(a), nit, K)
                assert oracle().orc_tdec_run(p(w2), True, K, nit, p(b), None) == 0
                assert np.array_equal(a, b), (K, snr, nit, "sb")
            R.srslte_tdec_free(tdec)


@pytest.mark.parametrize("K", [40, 176, 504, 1008, 5824, 6144])
def test_rate_dematching_all_rv_and_wraps(K):
    R, rng = ref(), np.random.default_rng(K)
    R.srslte_cbsegm_cbindex.restype = C.c_int
    W = oracle().orc_tdec_autoimp_subblocks(K)
    for rv in range(4):
        for n_e in (3 * K - 90, 3 * K + 12, 2 * (3 * K + 12) + 37, 1000):
            e = acopy(rng.integers(-2000, 2000, n_e).astype(np.int16))
            a, b = aligned(3 * (K + 32) + 76, np.int16), aligned(3 * (K + 32) + 76, np.int16)
            assert R.srslte_rm_turbo_rx_lut(p(e), p(a), n_e, R.srslte_cbsegm_cbindex(K), rv) == 0
            assert oracle().orc_rm_turbo_rx(p(e), p(b), n_e, K, rv, W) == 0
            assert np.array_equal(a, b), (K, rv, n_e)


def test_gold_and_crs():
    R = ref()
    for seed in (1, 12345, 0x7FFFFFFF, (0x1234 << 14) + (3 << 9) + 1):
        n = 5000
        c = np.zeros(n, np.uint8)
        oracle().orc_gold(C.c_uint32(seed), n, p(c))
        q = opaque(256)
        assert R.srslte_sequence_LTE_pr(q, n, C.c_uint32(seed)) == 0
        ptr = C.cast(q, C.POINTER(C.c_void_p))[0]
        assert np.array_equal(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n,)), c)


@pytest.mark.parametrize("mod", [0, 1, 2, 3, 4])
def test_modem_vs_ref(mod):
    R, rng = ref(), np.random.default_rng(mod)
    qm = 1 if mod == 0 else 2 * mod
    tab = opaque(1 << 16)
    assert R.srslte_modem_table_lte(tab, mod) == 0
    for nsym in (1, 3, 4, 7, 8, 15, 16, 33, 120, 14580):
        bits = rng.integers(0, 2, nsym * qm).astype(np.uint8)
        a, b = aligned(2 * nsym, np.float32), aligned(2 * nsym, np.float32)
        R.srslte_mod_modulate(tab, p(bits), p(a), nsym * qm)
        oracle().orc_modulate(mod, p(bits), p(b), nsym * qm)
        assert np.array_equal(a, b)
        for scale in (1, 3, 40):
            x = acopy(((a + 0.3 * rng.standard_normal(2 * nsym)) * scale).astype(np.float32))
            for name, on, dt in (("", "_f", np.float32), ("_s", "_s", np.int16), ("_b", "_b", np.int8)):
                l1, l2 = aligned(nsym * qm + 64, dt), aligned(nsym * qm + 64, dt)
                getattr(R, "srslte_demod_soft_demodulate" + name)(mod, p(x), p(l1), nsym)
                getattr(oracle(), "orc_demod_soft" + on)(mod, p(x), p(l2), nsym)
                assert np.array_equal(l1, l2), (mod, nsym, scale, name)


CHEST_CFGS = [{}, {"filter_coef": (4.0, 1.0)}, {"interpolate_subframe": True, "filter_coef": (4.0, 2.0), "cfo_estimate_enable": True},
              {"interpolate_subframe": True, "filter_type": 2}, {"filter_type": 1, "filter_coef": (0.1, 0.0)}, {"filter_type": 2},
              {"filter_coef": (4.0, 1.0), "sync_error_enable": True, "rsrp_neighbour": True}]


def check_sync_and_neighbour(res, ores, kw):
    """sync_error (chest_dl.c:692-703; srslte_vec_estimate_frequency's SIMD body multiplies by an approximate reciprocal, so 1e-3) and
    rsrp_neigh (:706-709,:821-843), when the configuration enables them."""
    if kw.get("sync_error_enable"):
        assert abs(res.sync_error - ores.sync_error) <= 2e-3 * abs(ores.sync_error) + 1e-4, (res.sync_error, ores.sync_error)
    else:
        assert np.isnan(res.sync_error) and np.isnan(ores.sync_error)
    if kw.get("rsrp_neighbour"):
        assert abs(res.rsrp_neigh - ores.rsrp_neigh) <= 1e-4 * abs(ores.rsrp_neigh) + 1e-9, (res.rsrp_neigh, ores.rsrp_neigh)


@pytest.mark.parametrize("prb,cid", [(6, 1), (6, 0), (25, 2), (50, 3), (100, 1), (100, 4), (100, 5), (15, 150)])
def test_chest_dl_vs_ref(prb, cid):
    R, rng = ref(), np.random.default_rng(prb + cid)
    for sf_idx in (0, 3):
        nre, n = 12 * prb, 14 * 12 * prb
        cell = OrcCell(cid, prb, 1, True)
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
        k, l = np.arange(n) % nre, np.arange(n) // nre
        h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
        grid = acopy((g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        for kw in CHEST_CFGS:
            q = opaque(1 << 20)
            assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
            rc, oc = RefChestCfg(), OrcChestCfg()
            for kk, v in kw.items():
                if kk == "filter_coef":
                    rc.filter_coef[0], rc.filter_coef[1] = v
                    oc.filter_coef[0], oc.filter_coef[1] = v
                else:
                    setattr(rc, kk, v)
                    setattr(oc, kk, v)
            rc.cfo_estimate_sf_mask = 0x3FF
            ce1, res, sf = aligned(2 * n, np.float32), RefChestRes(), RefDlSfCfg()
            res.ce[0][0] = ce1.ctypes.data
            sf.tti = sf_idx
            inp = (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)
            assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
            ce2, ores = np.zeros(n, np.complex64), OrcChestRes()
            assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(oc), p(grid), p(ce2), C.byref(ores)) == 0
            a = ce1.view(np.complex64)
            assert np.abs(a - ce2).max() <= 1e-4 * max(np.abs(a).max(), np.sqrt((np.abs(a) ** 2).mean())), (prb, cid, kw)
            for nm in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm", "cfo"):
                x, y = getattr(res, nm), getattr(ores, nm)
                assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, x, y)
            check_sync_and_neighbour(res, ores, kw)
            R.srslte_chest_dl_free(q)


@pytest.mark.parametrize("prb,cid", [(25, 2), (50, 3), (100, 4)])
def test_chest_dl_standard_symbol_sizes_vs_ref(prb, cid):
    """After srslte_use_standard_symbol_size(true) (phy_common.c:292-345; rf_uhd_imp.c:457,:473 does it for some radios) srslte_symbol_sz returns
    512 / 1024 / 2048 for 25 / 50 / 100 PRB, and the estimator's timing-error figure scales with it (chest_dl.c:695; the CFO's :575 cancels): the
    reference's compiled estimator with the switch on against the oracle with ITS switch on; then both off again, and the figures differ."""
    R, rng = ref(), np.random.default_rng(40 + prb)
    R.srslte_use_standard_symbol_size.argtypes = [C.c_bool]
    nre, n, sf_idx = 12 * prb, 14 * 12 * prb, 4
    cell = OrcCell(cid, prb, 1, True)
    g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
    k, l = np.arange(n) % nre, np.arange(n) // nre
    h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 90.0 + 0.12 * l))).astype(np.complex64)  # a slope over the carriers and over the symbols
    grid = acopy((g * h + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
    got = {}
    try:
        for std in (True, False):
            R.srslte_use_standard_symbol_size(std)
            oracle().orc_use_standard_symbol_size(std)
            assert R.srslte_symbol_sz(prb) == oracle().orc_symbol_sz(prb) == ({25: 512, 50: 1024, 100: 2048} if std else {25: 384, 50: 768, 100: 1536})[prb]
            q = opaque(1 << 20)
            assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
            rc, oc = RefChestCfg(), OrcChestCfg()
            rc.filter_coef[0], rc.filter_coef[1], oc.filter_coef[0], oc.filter_coef[1] = 4.0, 1.0, 4.0, 1.0
            rc.cfo_estimate_enable = oc.cfo_estimate_enable = True
            rc.sync_error_enable = oc.sync_error_enable = True
            rc.cfo_estimate_sf_mask = 0x3FF
            ce1, res, sf = aligned(2 * n, np.float32), RefChestRes(), RefDlSfCfg()
            res.ce[0][0] = ce1.ctypes.data
            sf.tti = sf_idx
            inp = (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)
            assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
            ce2, ores = np.zeros(n, np.complex64), OrcChestRes()
            assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(oc), p(grid), p(ce2), C.byref(ores)) == 0
            assert abs(res.cfo - ores.cfo) <= 1e-4 * abs(res.cfo) + 1e-6, (std, res.cfo, ores.cfo)
            assert abs(res.sync_error - ores.sync_error) <= 2e-3 * abs(ores.sync_error) + 1e-4, (std, res.sync_error, ores.sync_error)
            got[std] = (res.cfo, res.sync_error, ce1.view(np.complex64).copy())
            R.srslte_chest_dl_free(q)
    finally:
        R.srslte_use_standard_symbol_size(False)
        oracle().orc_use_standard_symbol_size(False)
    assert np.array_equal(got[True][2], got[False][2])                      # the estimates do not depend on the rate family,
    assert abs(got[True][1] / got[False][1] - 4.0 / 3.0) < 1e-3            # the timing error scales with the symbol size (x 4/3),
    assert abs(got[True][0] - got[False][0]) <= 1e-6 * abs(got[False][0])  # the CFO does not: N / (7.5 N + CP) is the same in both families


@pytest.mark.parametrize("prb,cid,npt", [(6, 1, 1), (25, 2, 1), (50, 3, 2), (100, 4, 1), (100, 5, 2)])
@pytest.mark.parametrize("alg", [1, 2])
def test_chest_dl_noise_pss_empty_vs_ref(prb, cid, npt, alg):
    """cfg.noise_alg PSS / EMPTY (chest_dl.c:381-411,:657-672) over a run of subframes on ONE estimator object: the estimate is renewed in
    subframes 0 and 5 only (from the PSS carriers against the channel estimates, or from the empty carriers around PSS / SSS), is reported
    unchanged in between, and sets the automatic Gauss filter of the next subframes; srslte_pss_generate for the sequence."""
    R, rng = ref(), np.random.default_rng(300 + prb + cid + alg)
    orc = oracle()
    orc.orc_chest_dl_ports_state.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for n2 in range(3):
        a, b = aligned(2 * 62, np.float32), np.zeros(62, np.complex64)
        assert R.srslte_pss_generate(p(a), n2) == 0
        orc.orc_pss_generate(n2, p(b))
        assert np.array_equal(a, b.view(np.float32))
    nre, n = 12 * prb, 14 * 12 * prb
    cell = OrcCell(cid, prb, npt, True)
    q = opaque(1 << 20)
    assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, npt, cid, 0, 0, 0, 0)) == 0
    state = np.zeros(16, np.float32)
    pss = np.zeros(62, np.complex64)
    orc.orc_pss_generate(cid % 3, p(pss))
    k, l = np.arange(n) % nre, np.arange(n) // nre
    h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
    # (a fresh object's estimate is 0, and the automatic Gauss filter then is NaN in the reference as well: start with explicit taps)
    for step, (tti, kw) in enumerate([(0, {"filter_coef": (4.0, 1.5)}), (1, {}), (2, {"interpolate_subframe": True}), (5, {}), (6, {}),
                                      (10, {"filter_type": 2}), (13, {})]):
        sf_idx = tti % 10
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        for pp in range(npt):
            orc.orc_crs_put_sf(C.byref(cell), sf_idx, pp, p(g))
        if sf_idx in (0, 5):
            kp, ks = 6 * nre + nre // 2 - 31, 5 * nre + nre // 2 - 31
            g[kp:kp + 62] = pss
            for k0 in (kp - 5, kp + 62, ks - 5, ks + 62):
                g[k0:k0 + 5] = 0
        sig = 0.05 * (1 + step)
        grid = acopy((g * h + sig * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        rc, oc = RefChestCfg(), OrcChestCfg()
        for kk, v in kw.items():
            if kk == "filter_coef":
                rc.filter_coef[0], rc.filter_coef[1] = v
                oc.filter_coef[0], oc.filter_coef[1] = v
            else:
                setattr(rc, kk, v)
                setattr(oc, kk, v)
        rc.noise_alg = oc.noise_alg = alg
        ces = [aligned(2 * n, np.float32) for _ in range(npt)]
        res, sf = RefChestRes(), RefDlSfCfg()
        for pp in range(npt):
            res.ce[pp][0] = ces[pp].ctypes.data
        sf.tti = tti
        assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
        ce2 = [np.zeros(n, np.complex64) for _ in range(npt)]
        ores = OrcChestRes()
        gp, cp = (C.c_void_p * 1)(grid.ctypes.data), (C.c_void_p * npt)(*[c.ctypes.data for c in ce2])
        prev = state.copy()
        assert orc.orc_chest_dl_ports_state(C.byref(cell), sf_idx, C.byref(oc), 1, gp, cp, C.byref(ores), None, p(state)) == 0
        for pp in range(npt):
            a = ces[pp].view(np.complex64)
            assert np.abs(a - ce2[pp]).max() <= 1e-4 * max(np.abs(a).max(), np.sqrt((np.abs(a) ** 2).mean())), (tti, pp)
        assert abs(res.noise_estimate - ores.noise_estimate) <= 1e-4 * abs(ores.noise_estimate) + 1e-12, (tti, res.noise_estimate, ores.noise_estimate)
        if sf_idx in (0, 5):
            assert not np.array_equal(state[:npt], prev[:npt])
        else:
            assert np.array_equal(state, prev)
        if step:
            assert abs(res.snr_db - ores.snr_db) < 1e-3 and abs(res.noise_estimate_dbm - ores.noise_estimate_dbm) < 1e-3
    R.srslte_chest_dl_free(q)


@pytest.mark.parametrize("prb,cid,npt", [(6, 1, 1), (25, 2, 2), (50, 3, 1), (100, 4, 2), (15, 150, 1)])
def test_chest_dl_extended_cp_vs_ref(prb, cid, npt):
    """Cells with the extended cyclic prefix (12 symbols per subframe; CRS on symbols 0, 3, 6, 9 with N_cp = 0 in c_init, refsignal_dl.c:
    66-116,:234-249): srslte_chest_dl_estimate_cfg with every configuration of CHEST_CFGS, the extended-CP time interpolation
    (chest_dl.c:497-502), CFO with 6 symbols per slot, and the PSS / EMPTY noise positions - against the oracle."""
    R, rng = ref(), np.random.default_rng(700 + prb + cid)
    orc = oracle()
    orc.orc_chest_dl_ports_state.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    nre, n = 12 * prb, 12 * 12 * prb
    cell = OrcCell(cid, prb, npt, False)
    pss = np.zeros(62, np.complex64)
    orc.orc_pss_generate(cid % 3, p(pss))
    q = opaque(1 << 20)
    assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, npt, cid, 1, 0, 0, 0)) == 0
    state, last_cfo = np.zeros(16, np.float32), 0.0
    k, l = np.arange(n) % nre, np.arange(n) // nre
    h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
    cfgs = [dict(kw) for kw in CHEST_CFGS] + [{"noise_alg": 1, "filter_coef": (4.0, 1.5)}, {"noise_alg": 2, "filter_type": 1, "filter_coef": (0.1, 0.0)},
                                              {"noise_alg": 1}]
    for step, kw in enumerate(cfgs):
        sf_idx = (0, 3, 5)[step % 3]
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        for pp in range(npt):
            orc.orc_crs_put_sf(C.byref(cell), sf_idx, pp, p(g))
        if sf_idx in (0, 5):
            kp, ks = 5 * nre + nre // 2 - 31, 4 * nre + nre // 2 - 31
            g[kp:kp + 62] = pss
            for k0 in (kp - 5, kp + 62, ks - 5, ks + 62):
                g[k0:k0 + 5] = 0
        grid = acopy((g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        rc, oc = RefChestCfg(), OrcChestCfg()
        for kk, v in kw.items():
            if kk == "filter_coef":
                rc.filter_coef[0], rc.filter_coef[1] = v
                oc.filter_coef[0], oc.filter_coef[1] = v
            else:
                setattr(rc, kk, v)
                setattr(oc, kk, v)
        rc.cfo_estimate_sf_mask = 0x3FF
        ces = [aligned(2 * n, np.float32) for _ in range(npt)]
        res, sf = RefChestRes(), RefDlSfCfg()
        for pp in range(npt):
            res.ce[pp][0] = ces[pp].ctypes.data
        sf.tti = sf_idx
        assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
        ce2, ores = [np.zeros(n, np.complex64) for _ in range(npt)], OrcChestRes()
        gp, cp = (C.c_void_p * 1)(grid.ctypes.data), (C.c_void_p * npt)(*[c.ctypes.data for c in ce2])
        assert orc.orc_chest_dl_ports_state(C.byref(cell), sf_idx, C.byref(oc), 1, gp, cp, C.byref(ores), None, p(state)) == 0
        for pp in range(npt):
            a = ces[pp].view(np.complex64)
            assert np.abs(a - ce2[pp]).max() <= 1e-4 * max(np.abs(a).max(), np.sqrt((np.abs(a) ** 2).mean())), (step, kw, pp)
        for nm in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm"):
            x, y = getattr(res, nm), getattr(ores, nm)
            assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, step, x, y)
        if kw.get("cfo_estimate_enable"):
            last_cfo = ores.cfo
            assert last_cfo != 0
        assert abs(res.cfo - last_cfo) <= 1e-4 * abs(last_cfo) + 1e-9  # q->cfo keeps its last enabled value
    R.srslte_chest_dl_free(q)


TDD_CHEST_CFGS = [{}, {"filter_coef": (4.0, 1.0)}, {"interpolate_subframe": True, "filter_coef": (4.0, 2.0)}, {"interpolate_subframe": True, "filter_type": 2},
                  {"filter_type": 1, "filter_coef": (0.1, 0.0)}, {"filter_type": 2}, {"filter_coef": (4.0, 1.0), "sync_error_enable": True}]


@pytest.mark.parametrize("prb,cid,npt,sf_cfg", [(6, 1, 1, 0), (25, 2, 2, 1), (50, 3, 1, 2), (100, 4, 2, 6), (15, 150, 4, 3), (100, 7, 1, 5)])
def test_chest_dl_tdd_special_subframes_vs_ref(prb, cid, npt, sf_cfg):
    """TDD cells (VERDICT r3 missing item 2): in a special subframe only the DwPTS symbols carry CRS - 4, 3, 2 or 1 pilot symbols for ports
    0 / 1 depending on the special-subframe configuration (srslte_refsignal_cs_nof_symbols, refsignal_dl.c:162-225) - and
    srslte_chest_dl_estimate_cfg follows: RSSI / RSRP over those symbols, the one-symbol noise formula (chest_dl.c:322-331), the time average
    that sums two rows and scales by 2 / 3 when there are three (:527-545), the 4 -> 7 slope continued to the subframe's end (:481-488), a
    single row spread over the subframe. Every special-subframe configuration 0-9 in subframes 1 and 6, and the downlink subframes beside
    them, reference build against the oracle."""
    R, rng = ref(), np.random.default_rng(1700 + prb + cid)
    orc = oracle()
    orc.orc_chest_dl_ports_state.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    orc.orc_crs_nof_symbols.restype = C.c_uint32
    nre, n = 12 * prb, 14 * 12 * prb
    k, l = np.arange(n) % nre, np.arange(n) // nre
    h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
    seen = set()
    for ss_cfg in range(10):
        cell = OrcCell(cid, prb, npt, True, 1, sf_cfg, ss_cfg)
        q = opaque(1 << 20)
        assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, npt, cid, 0, 0, 0, 1)) == 0  # frame_type SRSLTE_TDD
        state = np.zeros(16, np.float32)
        for step, kw in enumerate(TDD_CHEST_CFGS):
            sf_idx = (1, 6 if sf_cfg in (0, 1, 2, 6) else 5, 0)[step % 3]  # special, special (or downlink where subframe 6 is one), downlink
            nsym0 = orc.orc_crs_nof_symbols(C.byref(cell), sf_idx, 0)
            seen.add(nsym0)
            g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
            for pp in range(npt):
                orc.orc_crs_put_sf(C.byref(cell), sf_idx, pp, p(g))
            grid = acopy((g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
            rc, oc = RefChestCfg(), OrcChestCfg()
            for kk, v in kw.items():
                if kk == "filter_coef":
                    rc.filter_coef[0], rc.filter_coef[1] = v
                    oc.filter_coef[0], oc.filter_coef[1] = v
                else:
                    setattr(rc, kk, v)
                    setattr(oc, kk, v)
            ces = [aligned(2 * n, np.float32) for _ in range(npt)]
            ce2 = [np.zeros(n, np.complex64) for _ in range(npt)]
            if npt == 4 and kw.get("interpolate_subframe"):  # ports 2/3: symbol 0 of ce is in / out (the copy branch replicates what is there)
                for pp in (2, 3):
                    seed_row = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
                    ces[pp].view(np.complex64)[:] = seed_row
                    ce2[pp][:] = seed_row
            res, sf = RefChestRes(), RefDlSfCfg()
            for pp in range(npt):
                res.ce[pp][0] = ces[pp].ctypes.data
            sf.tti = sf_idx
            sf.tdd_config.sf_config, sf.tdd_config.ss_config, sf.tdd_config.configured = sf_cfg, ss_cfg, True
            assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
            ores = OrcChestRes()
            gp, cp = (C.c_void_p * 1)(grid.ctypes.data), (C.c_void_p * npt)(*[c.ctypes.data for c in ce2])
            assert orc.orc_chest_dl_ports_state(C.byref(cell), sf_idx, C.byref(oc), 1, gp, cp, C.byref(ores), None, p(state)) == 0
            for pp in range(npt):
                a = ces[pp].view(np.complex64)
                assert np.abs(a - ce2[pp]).max() <= 1e-4 * max(np.abs(a).max(), np.sqrt((np.abs(a) ** 2).mean())), (ss_cfg, sf_idx, step, kw, pp, nsym0)
            for nm in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm"):
                x, y = getattr(res, nm), getattr(ores, nm)
                assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, ss_cfg, sf_idx, step, x, y)
            if kw.get("sync_error_enable"):
                assert abs(res.sync_error - ores.sync_error) <= 1e-3 * abs(res.sync_error) + 1e-4, (ss_cfg, sf_idx)
        R.srslte_chest_dl_free(q)
    assert seen == {1, 2, 3, 4}  # every pilot-symbol count of refsignal_dl.c:162-225


MBSFN_CFGS = [{"filter_type": 1, "filter_coef": (0.1, 0.0), "noise_alg": 1}, {"filter_type": 2}, {"filter_type": 1, "filter_coef": (0.2, 0.0)},
              {"filter_coef": (4.0, 1.5)}, {}]


@pytest.mark.parametrize("prb,cid,area,port", [(6, 1, 1, 0), (25, 2, 0, 0), (50, 3, 255, 0), (100, 4, 17, 0), (100, 5, 2, 1), (15, 150, 77, 1)])
def test_chest_dl_mbsfn_vs_ref(prb, cid, area, port):
    """MBSFN subframes: srslte_refsignal_mbsfn_put_sf and srslte_chest_dl_estimate_cfg with sf_type MBSFN
    (chest_dl.c:718-745 and the MBSFN branches of :304-556) against orc_mbsfn_put_sf / orc_chest_dl_mbsfn, with the applications'
    configuration (triangle 0.1, PSS noise: cc_worker.cc:90-93) and the other filters; the 12 written symbols of ce and the REFS noise.
    The measurement fields the reference leaves stale in an MBSFN subframe keep the previous normal subframe's values."""
    R, rng = ref(), np.random.default_rng(900 + prb + cid + area)
    orc = oracle()
    orc.orc_chest_dl_mbsfn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    nports = 2 if port else 1
    nre, n = 12 * prb, 14 * 12 * prb
    cell = OrcCell(cid, prb, nports, True)
    q = opaque(1 << 20)
    assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, nports, cid, 0, 0, 0, 0)) == 0
    assert R.srslte_chest_dl_set_mbsfn_area_id(q, area) == 0
    # a normal subframe first: its rsrp / rssi / noise are what the MBSFN subframes report for the fields they do not measure
    g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    for pp in range(nports):
        orc.orc_crs_put_sf(C.byref(cell), 0, pp, p(g))
    grid0 = acopy((g * np.complex64(2 + 1j) + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
    rc0, res0, sf0 = RefChestCfg(), RefChestRes(), RefDlSfCfg()
    assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf0), C.byref(rc0), (C.c_void_p * 4)(grid0.ctypes.data, 0, 0, 0), C.byref(res0)) == 0
    prev_noise = res0.noise_estimate
    for sf_idx in (1, 2, 3, 6, 7, 8):
        # the reference's generator for the stimulus pilots
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        g2 = g.copy()
        assert orc.orc_mbsfn_put_sf(C.byref(cell), sf_idx, port, area, p(g)) == 0
        crs, mb = np.zeros(8 * prb, np.complex64), np.zeros(18 * prb, np.complex64)
        orc.orc_crs_pilots(C.byref(cell), sf_idx, port, p(crs))
        orc.orc_mbsfn_pilots(prb, area, sf_idx, p(mb))
        rs = opaque(4096)
        assert R.srslte_refsignal_mbsfn_init(rs, prb) == 0 and R.srslte_refsignal_mbsfn_set_cell(rs, RefCell(prb, nports, cid, 0, 0, 0, 0), area) == 0
        from _libs import ref_layout
        off = ref_layout({"srslte_refsignal_t": ["pilots"]}, ["srslte/phy/ch_estimation/refsignal_dl.h"])["srslte_refsignal_t.pilots"]
        pil_ptr = C.cast(C.byref(rs, off), C.POINTER(C.c_void_p))
        ref_mb = np.frombuffer(C.string_at(pil_ptr[(port // 2) * 10 + sf_idx], 8 * 18 * prb), np.complex64)
        assert np.array_equal(ref_mb.view(np.float32), mb.view(np.float32)), "MBSFN pilot sequence"
        ga = acopy(g2.view(np.float32))
        assert R.srslte_refsignal_mbsfn_put_sf(RefCell(prb, nports, cid, 0, 0, 0, 0), port, p(acopy(crs.view(np.float32))), C.c_void_p(ref_mb.ctypes.data), p(ga)) == 0
        assert np.array_equal(ga, g.view(np.float32)), "mbsfn_put_sf"
        R.srslte_refsignal_free(rs)
        k, l = np.arange(n) % nre, np.arange(n) // nre
        h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
        grid = acopy((g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        for kw in MBSFN_CFGS:
            rc, oc = RefChestCfg(), OrcChestCfg()
            for kk, v in kw.items():
                if kk == "filter_coef":
                    rc.filter_coef[0], rc.filter_coef[1] = v
                    oc.filter_coef[0], oc.filter_coef[1] = v
                else:
                    setattr(rc, kk, v)
                    setattr(oc, kk, v)
            rc.interpolate_subframe = oc.interpolate_subframe = True
            rc.mbsfn_area_id = area
            ces = [aligned(2 * n, np.float32) for _ in range(nports)]
            res, sf = RefChestRes(), RefDlSfCfg()
            for pp in range(nports):
                res.ce[pp][0] = ces[pp].ctypes.data
            sf.tti, sf.sf_type = sf_idx, 1
            inp = (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)
            assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
            ce2, noise = np.zeros(n, np.complex64), C.c_float(0)
            assert orc.orc_chest_dl_mbsfn(C.byref(cell), sf_idx, C.byref(oc), area, port, p(grid), p(ce2), C.byref(noise)) == 0
            a = ces[port].view(np.complex64)[:12 * nre]
            assert np.abs(a - ce2[:12 * nre]).max() <= 1e-4 * max(np.abs(a).max(), np.sqrt((np.abs(a) ** 2).mean())), (prb, cid, kw, sf_idx)
            if kw.get("noise_alg", 0) == 0 and nports == 1:
                assert abs(res.noise_estimate - noise.value) <= 1e-4 * abs(noise.value), (res.noise_estimate, noise.value)
                prev_noise = res.noise_estimate
            elif kw.get("noise_alg", 0):
                assert np.isnan(noise.value)
                if nports == 1:
                    assert res.noise_estimate == prev_noise  # PSS / EMPTY leave the estimate alone outside subframes 0 and 5
            assert res.rsrp == res0.rsrp and res.rsrq == res0.rsrq and res.rssi_dbm == res0.rssi_dbm and res.cfo == res0.cfo
            if nports == 1:
                assert abs(res.snr_db - 10 * np.log10(res0.rsrp / res.noise_estimate)) < 1e-3
    R.srslte_chest_dl_free(q)
    oc = OrcChestCfg()
    assert orc.orc_chest_dl_mbsfn(C.byref(cell), 1, C.byref(oc), area, 0, p(grid), p(ce2), None) == -3  # needs interpolate_subframe


@pytest.mark.parametrize("prb,cid", [(6, 0), (25, 7), (100, 301)])
def test_chest_dl_two_rx_antennas_vs_ref(prb, cid):
    """srslte_chest_dl_estimate_cfg with nof_rx_antennas = 2 (chest_dl.c:884-908 + fill_res :845-871): per-antenna estimates and the
    antenna-averaged scalars."""
    R, rng = ref(), np.random.default_rng(2000 + prb + cid)
    nre, n = 12 * prb, 14 * 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    for sf_idx, kw in ((0, CHEST_CFGS[0]), (4, CHEST_CFGS[-1]), (5, {"filter_coef": (4.0, 1.0)})):
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
        k = np.arange(n) % nre
        grids = []
        for a_, (amp, ph, nz) in enumerate(((2.5, 0.3, 0.08), (0.9, -1.1, 0.2))):
            h = (amp * (1 + 0.3 * np.sin(k / 35.0 + a_)) * np.exp(1j * (ph + k / 90.0))).astype(np.complex64)
            grids.append(acopy((g * h + nz * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32)))
        q = opaque(1 << 20)
        assert R.srslte_chest_dl_init(q, prb, 2) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
        rc, oc = RefChestCfg(), OrcChestCfg()
        for kk, v in kw.items():
            if kk == "filter_coef":
                rc.filter_coef[0], rc.filter_coef[1] = v
                oc.filter_coef[0], oc.filter_coef[1] = v
            else:
                setattr(rc, kk, v)
                setattr(oc, kk, v)
        rc.cfo_estimate_sf_mask = 0x3FF
        ce_r, res, sf = [aligned(2 * n, np.float32) for _ in range(2)], RefChestRes(), RefDlSfCfg()
        res.ce[0][0], res.ce[0][1] = ce_r[0].ctypes.data, ce_r[1].ctypes.data
        sf.tti = sf_idx
        inp = (C.c_void_p * 4)(grids[0].ctypes.data, grids[1].ctypes.data, 0, 0)
        assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
        ce_o, ores = [np.zeros(n, np.complex64) for _ in range(2)], OrcChestRes()
        gp, cp = (C.c_void_p * 2)(grids[0].ctypes.data, grids[1].ctypes.data), (C.c_void_p * 2)(ce_o[0].ctypes.data, ce_o[1].ctypes.data)
        assert oracle().orc_chest_dl_multi(C.byref(cell), sf_idx, C.byref(oc), 2, gp, cp, C.byref(ores)) == 0
        for a_ in range(2):
            x = ce_r[a_].view(np.complex64)
            assert np.abs(x - ce_o[a_]).max() <= 1e-4 * max(np.abs(x).max(), np.sqrt((np.abs(x) ** 2).mean())), (prb, cid, a_)
        for nm in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm", "cfo"):
            x, y = getattr(res, nm), getattr(ores, nm)
            assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, x, y)
        R.srslte_chest_dl_free(q)


@pytest.mark.parametrize("prb,cid,nrx,npt", [(6, 0, 1, 2), (25, 7, 1, 2), (100, 301, 1, 2), (50, 4, 2, 2), (100, 149, 2, 2),
                                             (6, 2, 1, 4), (25, 7, 1, 4), (100, 301, 1, 4), (50, 5, 2, 4), (15, 148, 2, 4)])
def test_chest_dl_two_ports_vs_ref(prb, cid, nrx, npt):
    """srslte_chest_dl_estimate_cfg for a 2-port cell (ports 0/1 share the pilot values, refsignal_dl.c pilots[port / 2], at v-shifted
    positions) with 1 and 2 receive antennas: every ce[port][antenna], the aggregated scalars incl. the port-by-antenna-index quirk
    of get_rsrp and the last-estimate-wins CFO. 4-port cells add ports 2/3 (two pilot symbols, 1 and 8):
    the two-row noise estimate, no time-averaging partner, and a CFO that pairs port 3's symbols with what port 1 left in the shared
    pilot buffer; interpolate_subframe is left out for them (upstream replicates a never-written symbol)."""
    R, rng = ref(), np.random.default_rng(3000 + prb + cid + nrx)
    nre, n = 12 * prb, 14 * 12 * prb
    cell = OrcCell(cid, prb, npt, True)
    oracle().orc_chest_dl_ports.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    cfo_cfg = {"filter_coef": (4.0, 2.0), "cfo_estimate_enable": True}
    # 4-port cells with interpolate_subframe (sf 2, 8): ports 2/3 come out as symbol 0 of what the caller's buffer held, replicated (chest_dl.c:467-471)
    for sf_idx, kw in ((0, CHEST_CFGS[0]), (3, CHEST_CFGS[2] if npt == 2 else cfo_cfg), (5, CHEST_CFGS[1]), (9, CHEST_CFGS[3] if npt == 2 else CHEST_CFGS[5]),
                       (4, CHEST_CFGS[4]), (7, CHEST_CFGS[6]), (2, CHEST_CFGS[2]), (8, CHEST_CFGS[3])):
        k, l = np.arange(n) % nre, np.arange(n) // nre
        tx = []
        for port in range(npt):  # each port: its own CRS (zeros at the other ports' positions) and some data
            g = np.zeros(n, np.complex64)
            oracle().orc_crs_put_sf(C.byref(cell), sf_idx, port, p(g))
            tx.append(g)
        hole = np.zeros(n, bool)
        for g in tx:
            hole |= g != 0
        data = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        grids = []
        for a_ in range(nrx):
            rxg = np.where(hole, 0, data).astype(np.complex64) * (1.5 - 0.4 * a_)
            for port in range(npt):
                h = ((2.0 - 0.35 * port * (2 if npt == 2 else 1) + 0.3 * a_) * (1 + 0.25 * np.sin(k / 30.0 + port + 2 * a_)) *
                     np.exp(1j * (0.4 * port - 0.9 * a_ + k / 80.0 + 0.05 * l))).astype(np.complex64)
                rxg = rxg + tx[port] * h
            rxg = rxg + (0.05 + 0.1 * a_) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
            grids.append(acopy(rxg.astype(np.complex64).view(np.float32)))
        q = opaque(1 << 20)
        assert R.srslte_chest_dl_init(q, prb, nrx) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, npt, cid, 0, 0, 0, 0)) == 0
        rc, oc = RefChestCfg(), OrcChestCfg()
        for kk, v in kw.items():
            if kk == "filter_coef":
                rc.filter_coef[0], rc.filter_coef[1] = v
                oc.filter_coef[0], oc.filter_coef[1] = v
            else:
                setattr(rc, kk, v)
                setattr(oc, kk, v)
        rc.cfo_estimate_sf_mask = 0x3FF
        res, sf = RefChestRes(), RefDlSfCfg()
        ce_r = [[aligned(2 * n, np.float32) for _ in range(nrx)] for _ in range(npt)]
        before = (rng.standard_normal((npt, nrx, n)) + 1j * rng.standard_normal((npt, nrx, n))).astype(np.complex64)
        for port in range(npt):
            for a_ in range(nrx):
                ce_r[port][a_].view(np.complex64)[:] = before[port, a_]
                res.ce[port][a_] = ce_r[port][a_].ctypes.data
        sf.tti = sf_idx
        inp = (C.c_void_p * 4)(*([g.ctypes.data for g in grids] + [0] * (4 - nrx)))
        assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
        ce_o, ores = [before[i // nrx, i % nrx].copy() for i in range(npt * nrx)], OrcChestRes()
        gp = (C.c_void_p * nrx)(*[g.ctypes.data for g in grids])
        cp = (C.c_void_p * (npt * nrx))(*[c.ctypes.data for c in ce_o])
        raw = np.zeros(nrx * npt * 4, np.float32)
        assert oracle().orc_chest_dl_ports(C.byref(cell), sf_idx, C.byref(oc), nrx, gp, cp, C.byref(ores), p(raw)) == 0
        for port in range(npt):
            for a_ in range(nrx):
                x = ce_r[port][a_].view(np.complex64)
                assert np.abs(x - ce_o[port * nrx + a_]).max() <= 1e-4 * max(np.abs(x).max(), np.sqrt((np.abs(x) ** 2).mean())), (prb, cid, port, a_)
        for nm in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm", "cfo"):
            x, y = getattr(res, nm), getattr(ores, nm)
            assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, x, y, sf_idx)
        check_sync_and_neighbour(res, ores, kw)
        raw = raw.reshape(nrx, npt, 4)
        for port in range(npt):  # per-port fields of fill_res (chest_dl.c:860-870) from the per-(antenna, port) scalars
            assert abs(res.rsrp_port_dbm[port] - (10 * np.log10(raw[:, port, 1].mean()) + 30)) <= 1e-3
            for a_ in range(nrx):
                assert abs(res.snr_ant_port_db[a_][port] - 10 * np.log10(raw[a_, port, 1] / raw[a_, port, 0])) <= 1e-3
                assert abs(res.rsrq_ant_port_db[a_][port] - 10 * np.log10(prb * raw[a_, port, 1] / raw[a_, port, 2])) <= 1e-3
        R.srslte_chest_dl_free(q)


def test_equaliser_two_rx_vs_ref():
    """srslte_predecoding_single_multi (precoding.c:325-348): AVX body and scalar tail both divide exactly."""
    R, rng = ref(), np.random.default_rng(19)
    R.srslte_predecoding_single_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
    oracle().orc_predecoding_single_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
    for n in (20, 1003):
        ys = [acopy(rng.standard_normal(2 * n).astype(np.float32)) for _ in range(2)]
        hs = [acopy((rng.standard_normal(2 * n) + 1.5).astype(np.float32)) for _ in range(2)]
        a, b = aligned(2 * n, np.float32), aligned(2 * n, np.float32)
        yp, hp = (C.c_void_p * 4)(ys[0].ctypes.data, ys[1].ctypes.data, 0, 0), (C.c_void_p * 4)(hs[0].ctypes.data, hs[1].ctypes.data, 0, 0)
        for noise in (0.0, 0.07):
            R.srslte_predecoding_single_multi(yp, hp, p(a), None, 2, n, 1.0, noise)
            oracle().orc_predecoding_single_multi(yp, hp, p(b), 2, n, 1.0, noise)
            assert np.abs(np.array(a) - np.array(b)).max() <= 2e-6 * np.abs(np.array(b)).max(), (n, noise)


def test_tx_diversity_vs_ref():
    """2-port SFBC: srslte_layermap_diversity + srslte_precoding_diversity on the transmit side, srslte_predecoding_diversity_multi with a
    csi buffer (the path srslte_pdsch_decode takes) and without one (SSE body) + srslte_layerdemap_diversity on the receive side."""
    R, rng, orc = ref(), np.random.default_rng(23), oracle()
    R.srslte_precoding_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    R.srslte_layermap_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    R.srslte_layerdemap_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    R.srslte_predecoding_diversity_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
    orc.orc_precoding_diversity2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    orc.orc_predecoding_diversity2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    for n in (24, 1000, 14052):
        d = acopy(rng.standard_normal(2 * n).astype(np.float32))
        x = [aligned(n, np.float32), aligned(n, np.float32)]
        xp = (C.c_void_p * 4)(x[0].ctypes.data, x[1].ctypes.data, 0, 0)
        assert R.srslte_layermap_diversity(p(d), xp, 2, n) == n // 2
        y_r = [aligned(2 * n, np.float32), aligned(2 * n, np.float32)]
        yp = (C.c_void_p * 4)(y_r[0].ctypes.data, y_r[1].ctypes.data, 0, 0)
        assert R.srslte_precoding_diversity(xp, yp, 2, n // 2, 1.0) == n
        y_o = [np.zeros(n, np.complex64), np.zeros(n, np.complex64)]
        orc.orc_precoding_diversity2(p(d), p(y_o[0]), p(y_o[1]), n, 1.0)
        for port in range(2):
            assert np.abs(y_r[port].view(np.complex64) - y_o[port]).max() <= 1e-6
        for nrx in (1, 2):
            hs = []  # [port * nrx + antenna]; nearly constant over each sub-carrier pair, as SFBC assumes
            for _ in range(2 * nrx):
                hc = np.repeat(rng.standard_normal(n // 2) + 1j * rng.standard_normal(n // 2) + 0.5, 2)
                hc = hc * (1 + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
                hs.append(acopy(hc.astype(np.complex64).view(np.float32)))
            ys = []
            for a_ in range(nrx):
                rx = sum(y_o[port] * hs[port * nrx + a_].view(np.complex64) for port in range(2))
                rx = rx + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
                ys.append(acopy(rx.astype(np.complex64).view(np.float32)))
            yp2 = (C.c_void_p * 4)(*([v.ctypes.data for v in ys] + [0] * (4 - nrx)))
            hp = ((C.c_void_p * 4) * 4)()
            for port in range(2):
                for a_ in range(nrx):
                    hp[port][a_] = hs[port * nrx + a_].ctypes.data
            d_o, csi_o = np.zeros(n, np.complex64), np.zeros(n, np.float32)
            yo = (C.c_void_p * nrx)(*[v.ctypes.data for v in ys])
            ho = (C.c_void_p * (2 * nrx))(*[v.ctypes.data for v in hs])
            orc.orc_predecoding_diversity2(yo, ho, p(d_o), p(csi_o), nrx, n, 1.0)
            for with_csi in (True, False):
                xr = [aligned(n, np.float32), aligned(n, np.float32)]
                xrp = (C.c_void_p * 4)(xr[0].ctypes.data, xr[1].ctypes.data, 0, 0)
                csi_r = aligned(n, np.float32)
                csip = (C.c_void_p * 2)(csi_r.ctypes.data if with_csi else 0, 0)
                R.srslte_predecoding_diversity_multi(yp2, hp, xrp, csip, nrx, 2, n, 1.0)
                d_r = aligned(2 * n, np.float32)
                assert R.srslte_layerdemap_diversity(xrp, p(d_r), 2, n // 2) == n
                a = d_r.view(np.complex64)
                assert np.abs(a - d_o).max() <= 2e-6 * max(1.0, np.abs(a).max()), (n, nrx, with_csi)
                if with_csi:
                    assert np.abs(np.array(csi_r) - csi_o).max() <= 1e-6 * csi_o.max()
            # the transmitted symbols come back (noise- and mismatch-limited)
            assert np.sqrt(np.mean(np.abs(d_o - d.view(np.complex64)) ** 2)) < 0.25


def test_tx_diversity_4_ports_vs_ref():
    """4-port SFBC + FSTD: srslte_layermap_diversity + srslte_precoding_diversity (precoding.c:1862-1890) and
    srslte_predecoding_diversity_multi with a csi buffer (the variant srslte_pdsch_decode reaches, :599-650) + srslte_layerdemap_diversity."""
    R, rng, orc = ref(), np.random.default_rng(29), oracle()
    R.srslte_precoding_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    R.srslte_layermap_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    R.srslte_layerdemap_diversity.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    R.srslte_predecoding_diversity_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
    orc.orc_precoding_diversity4.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    orc.orc_predecoding_diversity4.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    for n, scaling in ((24, 1.0), (1000, 1.4142135), (13824, 1.0)):
        d = acopy(rng.standard_normal(2 * n).astype(np.float32))
        x = [aligned(n // 2, np.float32) for _ in range(4)]
        xp = (C.c_void_p * 4)(*[v.ctypes.data for v in x])
        assert R.srslte_layermap_diversity(p(d), xp, 4, n) == n // 4
        y_r = [aligned(2 * n, np.float32) for _ in range(4)]
        yp = (C.c_void_p * 4)(*[v.ctypes.data for v in y_r])
        assert R.srslte_precoding_diversity(xp, yp, 4, n // 4, scaling) == n
        y_o = [np.zeros(n, np.complex64) for _ in range(4)]
        orc.orc_precoding_diversity4(p(d), (C.c_void_p * 4)(*[v.ctypes.data for v in y_o]), n, scaling)
        for port in range(4):
            assert np.abs(y_r[port].view(np.complex64) - y_o[port]).max() <= 1e-6
        for nrx in (1, 2):
            hs = []  # [port * nrx + antenna], nearly constant over each group of four sub-carriers
            for _ in range(4 * nrx):
                hc = np.repeat(rng.standard_normal(n // 4) + 1j * rng.standard_normal(n // 4) + 0.5, 4)
                hc = hc * (1 + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)))
                hs.append(acopy(hc.astype(np.complex64).view(np.float32)))
            ys = []
            for a_ in range(nrx):
                rx = sum(y_o[port] * hs[port * nrx + a_].view(np.complex64) for port in range(4))
                rx = rx + 0.03 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
                ys.append(acopy(rx.astype(np.complex64).view(np.float32)))
            yp2 = (C.c_void_p * 4)(*([v.ctypes.data for v in ys] + [0] * (4 - nrx)))
            hp = ((C.c_void_p * 4) * 4)()
            for port in range(4):
                for a_ in range(nrx):
                    hp[port][a_] = hs[port * nrx + a_].ctypes.data
            d_o, csi_o = np.zeros(n, np.complex64), np.zeros(n, np.float32)
            orc.orc_predecoding_diversity4((C.c_void_p * nrx)(*[v.ctypes.data for v in ys]), (C.c_void_p * (4 * nrx))(*[v.ctypes.data for v in hs]),
                                           p(d_o), p(csi_o), nrx, n, scaling)
            xr = [aligned(n // 2, np.float32) for _ in range(4)]
            xrp = (C.c_void_p * 4)(*[v.ctypes.data for v in xr])
            csi_r = aligned(n, np.float32)
            R.srslte_predecoding_diversity_multi(yp2, hp, xrp, (C.c_void_p * 2)(csi_r.ctypes.data, 0), nrx, 4, n, scaling)
            d_r = aligned(2 * n, np.float32)
            assert R.srslte_layerdemap_diversity(xrp, p(d_r), 4, n // 4) == n
            a = d_r.view(np.complex64)
            assert np.abs(a - d_o).max() <= 2e-6 * max(1.0, np.abs(a).max()), (n, nrx)
            assert np.abs(np.array(csi_r) - csi_o).max() <= 1e-6 * csi_o.max()
            assert np.sqrt(np.mean(np.abs(d_o - d.view(np.complex64)) ** 2)) < 0.25  # the transmitted symbols come back


PDSCH_GRANT_OFF = {"prb_idx": 8, "nof_prb": 228, "nof_re": 232, "nof_symb_slot": 236}  # srslte_pdsch_grant_t (pdsch_cfg.h:37-49), SRSLTE_MAX_PRB 110


@pytest.mark.parametrize("prb,cid,ports,cfi", [(6, 0, 1, 3), (6, 3, 2, 3), (15, 7, 2, 2), (25, 11, 1, 1), (25, 150, 2, 1), (100, 301, 2, 1), (75, 2, 2, 3),
                                               (6, 5, 4, 1), (25, 7, 4, 2), (100, 148, 4, 1), (15, 3, 4, 3)])
def test_pdsch_re_mapping_vs_ref(prb, cid, ports, cfi):
    """srslte_pdsch_cp (pdsch.c:81-206) on a grid of running indices = orc_pdsch_indices: 1- and 2-port CRS holes, PSS/SSS/PBCH, odd
    bandwidths, for subframes 0, 5 and an ordinary one."""
    R = ref()
    n = 14 * 12 * prb
    q = opaque(1 << 16)
    C.memmove(q, C.byref(RefCell(prb, ports, cid, 0, 0, 0, 0)), C.sizeof(RefCell))  # srslte_pdsch_t begins with its srslte_cell_t (pdsch.h:53-54)
    grant = np.zeros(4096, np.uint8)
    grant[PDSCH_GRANT_OFF["prb_idx"]:PDSCH_GRANT_OFF["prb_idx"] + 220].reshape(2, 110)[:, :prb] = 1
    grant[PDSCH_GRANT_OFF["nof_symb_slot"]:PDSCH_GRANT_OFF["nof_symb_slot"] + 8].view(np.uint32)[:] = 7
    lstart = cfi + (1 if prb < 10 else 0)
    cell = OrcCell(cid, prb, ports, True)
    src = acopy(np.arange(2 * n, dtype=np.float32))
    src.view(np.complex64).real[:] = np.arange(n)
    R.srslte_pdsch_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    for sf_idx in (0, 5, 7):
        out = aligned(2 * n, np.float32)
        cnt = R.srslte_pdsch_get(q, p(src), p(out), p(grant), lstart, sf_idx)
        idx = np.zeros(n, np.uint32)
        cnt_o = oracle().orc_pdsch_indices(C.byref(cell), sf_idx, lstart, None, p(idx))
        assert cnt == cnt_o and np.array_equal(out.view(np.complex64).real[:cnt].astype(np.uint32), idx[:cnt]), (sf_idx, cnt, cnt_o)
        assert ports == 1 or cnt % ports == 0


def test_equaliser_vs_ref_rcp_tolerance():
    """The reference's AVX body uses _mm256_rcp_ps (12-bit): parity at 1e-3, see oracle/orc_pdsch.c."""
    R, rng = ref(), np.random.default_rng(9)
    n = 1000
    y, h = acopy(rng.standard_normal(2 * n).astype(np.float32)), acopy((rng.standard_normal(2 * n) + 2).astype(np.float32))
    a, b = aligned(2 * n, np.float32), aligned(2 * n, np.float32)
    R.srslte_predecoding_single(p(y), p(h), p(a), None, n, 1.0, 0.1)
    oracle().orc_predecoding_single(p(y), p(h), p(b), n, 1.0, 0.1)
    assert np.abs(a - b).max() <= 1e-3 * np.abs(b).max()


@pytest.mark.parametrize("prb,mod,tbs,snr", [(6, 1, 936, 3.0), (100, 3, 75376, 18.0), (100, 4, 97896, 27.0)])
def test_whole_chain_vs_reference_code(prb, mod, tbs, snr):
    """Same IQ through the oracle chain and through the reference's compiled chest/eq/demod/rm/tdec/crc: same TBs and pass counts."""
    rng = np.random.default_rng(prb + mod)
    cfg = DlConfig(prb, 1, mod, tbs)
    chain = RefRx(cfg)
    ttis = (1, 2, 3) if prb == 6 else (0, 5, 7)
    for t in ttis:
        iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
        r, o = chain.run(iq, t), oracle_rx(cfg, iq, t)
        assert r["ok"] == o["ok"] and np.array_equal(r["iters"], o["iters"]) and np.array_equal(r["tb"], o["tb"])


@pytest.mark.parametrize("prb,mod,tbs,snr", [(6, 1, 936, 4.0), (100, 3, 75376, 19.0), (100, 4, 97896, 29.0)])
def test_whole_chain_8bit_vs_reference_code(prb, mod, tbs, snr):
    """8-bit LLR path (pdsch.c:760-779, sch.c:336-356): demod_b, int8 descrambling, srslte_rm_turbo_rx_lut_8bit,
    srslte_tdec_iteration_8bit (sse8 for K=960, avx8 for K=5824/6144) - same TBs and pass counts as the oracle chain."""
    rng = np.random.default_rng(80 + prb + mod)
    cfg = DlConfig(prb, 1, mod, tbs, llr8=True)
    chain = RefRx(cfg)
    ttis = (1, 2, 3) if prb == 6 else (0, 5, 7)
    nok = 0
    for t in ttis:
        iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
        r, o = chain.run(iq, t), oracle_rx(cfg, iq, t)
        assert r["ok"] == o["ok"] and np.array_equal(r["iters"], o["iters"]) and np.array_equal(r["tb"], o["tb"])
        nok += r["ok"]
    assert nok > 0


@pytest.mark.parametrize("prb,mod,tbs,snr", [(6, 1, 936, 0.0), (100, 3, 75376, 17.5)])
def test_whole_chain_two_rx_vs_reference_code(prb, mod, tbs, snr):
    """Two receive antennas: per-antenna chest_dl + srslte_predecoding_single_multi in the reference-code chain
    vs the oracle chain; the SNR is below the single-antenna waterfall, so decoding relies on the combining gain."""
    rng = np.random.default_rng(300 + prb + mod)
    cfg = DlConfig(prb, 1, mod, tbs, nof_rx=2)
    chain = RefRx(cfg)
    nok = 0
    for t in ((1, 2, 3) if prb == 6 else (0, 5, 7)):
        iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
        r, o = chain.run(iq, t), oracle_rx(cfg, iq, t)
        assert r["ok"] == o["ok"] and np.array_equal(r["iters"], o["iters"]) and np.array_equal(r["tb"], o["tb"])
        nok += r["ok"]
    assert nok > 0


@pytest.mark.parametrize("prb,mod,tbs,nrx,snr,llr8", [(6, 1, 152, 1, 4.0, False), (25, 2, 4008, 1, 11.0, False), (25, 3, 9912, 2, 13.5, False),
                                                        (100, 3, 75376, 1, 19.5, False), (50, 2, 11448, 2, 7.0, True)])
def test_whole_chain_tx_diversity_vs_reference_code(prb, mod, tbs, nrx, snr, llr8):
    """2-port transmit diversity (TM2): 2-port chest_dl + srslte_predecoding_diversity_multi + layer demapping in the
    reference-code chain vs the oracle chain, 1 and 2 receive antennas, near the waterfall."""
    rng = np.random.default_rng(400 + prb + mod)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=2, llr8=llr8)
    chain = RefRx(cfg)
    nok = 0
    for t in (0, 5, 7, 8):
        iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
        r, o = chain.run(iq, t), oracle_rx(cfg, iq, t)
        assert r["ok"] == o["ok"] and np.array_equal(r["iters"], o["iters"]), t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        nok += r["ok"]
    assert nok > 0


@pytest.mark.parametrize("prb,mod,tbs,nrx,npt,snr,p_a", [(25, 2, 4008, 1, 1, 9.0, -3.0), (25, 3, 9912, 2, 2, 13.5, 0.0), (100, 3, 75376, 1, 2, 19.5, 0.0), (15, 1, 1000, 1, 2, 2.0, 1.77)])
def test_pdsch_decode_power_scaling_vs_oracle_chain(prb, mod, tbs, nrx, npt, snr, p_a):
    """cfg->power_scale with p_a (pdsch.c:518-554,:852-858; p_b such that rho_b = 1 as phy_dl_test.c:176-178): srslte_pdsch_decode divides
    the equalised symbols by rho_a; the transmitter here scales its PDSCH symbols by the same rho_a (relative to the CRS)."""
    from lte_sim import RefPdsch
    rng = np.random.default_rng(900 + prb + mod + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, p_a=p_a)
    chain = RefPdsch(cfg)
    nok = 0
    for t in (0, 4, 5):
        iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
        r, o = chain.run(iq, t), oracle_rx(cfg, iq, t, keep=True)
        assert np.abs(r["d"] - o["d"]).max() <= (2e-6 if npt == 2 else 1e-3) * max(1.0, np.abs(o["d"]).max())
        diff = np.abs(r["e"].astype(np.int32) - o["e"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() <= 0.05 and r["ok"] == o["ok"], t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        nok += r["ok"]
    assert nok > 0


@pytest.mark.parametrize("csi", [False, True])
@pytest.mark.parametrize("prb,mod,tbs,nrx,npt,snr,llr8", [(6, 1, 152, 1, 1, 4.0, False), (15, 1, 1000, 1, 2, 2.0, False), (25, 2, 4008, 1, 1, 9.0, False),
                                                            (25, 3, 9912, 2, 2, 13.5, False), (100, 3, 75376, 1, 1, 18.0, False),
                                                            (25, 2, 4008, 1, 2, 9.0, True), (100, 4, 97896, 2, 1, 24.0, False), (50, 3, 11448, 2, 1, 8.0, True),
                                                            (100, 3, 75376, 1, 2, 19.5, False), (100, 2, 43816, 2, 2, 10.5, False),
                                                            (25, 2, 4008, 2, 4, 9.0, False), (100, 3, 61664, 1, 4, 19.0, False), (15, 1, 1000, 1, 4, 3.0, True)])
def test_pdsch_decode_function_vs_oracle_chain(prb, mod, tbs, nrx, npt, snr, llr8, csi):
    """The reference's own srslte_pdsch_decode (pdsch.c:833-997; UE object, so the csi variants of the equalisers run) on the output of
    its srslte_chest_dl_estimate_cfg, against the oracle chain on identical IQ: TM1 / TM2, 1-2 antennas, 16- and 8-bit LLRs, with and
    without the CSI weighting of the LLRs (cfg->csi_enable, the srsUE default). The reference's single-port equaliser multiplies by the
    AVX reciprocal approximation, the oracle divides: symbols agree to 1e-3, LLRs to one LSB on a few percent of the values (6 % for 256QAM), CRC results
    and transport blocks exactly. TM2 divides exactly in both; its 100-PRB cases have a code-block count that does not divide
    nof_re / 2, which is where the N_L = 2 of the rate matcher's block split (sch.c:507-531) shows."""
    from lte_sim import RefPdsch
    rng = np.random.default_rng(500 + prb + mod + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, llr8=llr8, csi=csi)
    chain = RefPdsch(cfg, csi_enable=csi)
    nok = 0
    for t in (0, 3, 5, 8):
        iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
        r, o = chain.run(iq, t), oracle_rx(cfg, iq, t, keep=True)
        assert np.abs(r["d"] - o["d"]).max() <= (2e-6 if npt > 1 else 1e-3) * max(1.0, np.abs(o["d"]).max())
        assert np.abs(r["csi"] - o["csi"]).max() <= 2e-6 * o["csi"].max()
        diff = np.abs(r["e"].astype(np.int32) - o["e"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() <= (0.005 if npt > 1 else 0.08), (t, diff.max(), (diff != 0).mean())
        assert r["ok"] == o["ok"], t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        nok += r["ok"]
    assert nok > 0


@pytest.mark.parametrize("prb,mod,tbs,nrx,snr", [(6, 1, 152, 1, 5.0), (25, 2, 4008, 1, 10.0), (100, 3, 43816, 1, 15.0), (50, 3, 11448, 2, 8.0)])
def test_pdsch_decode_extended_cp_vs_oracle_chain(prb, mod, tbs, nrx, snr):
    """Extended-CP cells (VERDICT r3 missing item 3): 12 symbols per subframe, CRS on symbols 0 and 3 of each slot, PSS / SSS on the last two
    symbols of slot 0 (pdsch.c:81-206 with nof_symb_slot = 6, SRSLTE_SYMBOL_HAS_REF; ofdm.c:424-437; chest_dl.c:497-502). The reference's own
    srslte_chest_dl_estimate_cfg + srslte_pdsch_decode on a cell with cp = SRSLTE_CP_EXT against the oracle chain on identical IQ, subframes
    with and without PSS / SSS / PBCH; with interpolate_subframe (the branch :497-502) and without."""
    from lte_sim import RefPdsch
    for interp in (False, True):
        rng = np.random.default_rng(900 + prb + mod)
        cfg = DlConfig(prb, 11, mod, tbs, nof_rx=nrx, cp_ext=True, chest={"filter_coef": (4.0, 1.0), "interpolate_subframe": interp})
        assert cfg.grid_len == 12 * 12 * prb
        chain = RefPdsch(cfg)
        nok = 0
        for t in (0, 4, 5, 9):
            iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
            r, o = chain.run(iq, t), oracle_rx(cfg, iq, t, keep=True)
            assert np.abs(r["d"] - o["d"]).max() <= 1e-3 * max(1.0, np.abs(o["d"]).max())
            diff = np.abs(r["e"].astype(np.int32) - o["e"].astype(np.int32))
            assert diff.max() <= 1 and (diff != 0).mean() <= 0.08, (t, diff.max(), (diff != 0).mean())
            assert r["ok"] == o["ok"], t
            if r["ok"]:
                assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
            nok += r["ok"]
        assert nok > 0, interp


@pytest.mark.parametrize("prb,mod,tbs,snr,tdd", [(6, 1, 152, 6.0, (1, 7)), (25, 2, 2216, 9.0, (2, 4)), (100, 2, 14112, 6.5, (0, 1)), (50, 2, 5736, 8.0, (6, 9)),
                                                  (100, 3, 30576, 12.0, (1, 3)), (15, 1, 1000, 6.0, (2, 2))])
def test_pdsch_decode_tdd_vs_oracle_chain(prb, mod, tbs, snr, tdd):
    """TDD cells: the reference's own srslte_chest_dl_estimate_cfg + srslte_pdsch_decode with cell.frame_type = SRSLTE_TDD, the subframe's
    srslte_tdd_config_t and the grant's DwPTS symbol counts (ra_dl.c:446-460) against the oracle chain on identical IQ - downlink subframes
    0 (SSS on the last symbol of slot 1, PBCH) and 5 (SSS), a plain one, and the special subframes 1 and 6 (PSS on symbol 2, PDSCH in the DwPTS
    symbols only, CRS in those symbols only: pdsch.c:124-140, refsignal_dl.c:162-225). The transport block sizes fit the shortened subframes."""
    from lte_sim import RefPdsch
    rng = np.random.default_rng(1900 + prb + mod)
    cfg = DlConfig(prb, 9, mod, tbs, tdd=tdd)
    orc = oracle()
    orc.orc_tdd_sf_type.restype = C.c_int
    chain = RefPdsch(cfg)
    nok = nspecial = 0
    for t in (0, 1, 4 if orc.orc_tdd_sf_type(C.byref(cfg.cell), 4) == 0 else 5, 5, 6, 9 if orc.orc_tdd_sf_type(C.byref(cfg.cell), 9) == 0 else 0):
        typ = orc.orc_tdd_sf_type(C.byref(cfg.cell), t % 10)
        if typ == 1:
            continue  # uplink subframe: no PDSCH
        nspecial += typ == 2
        iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1)
        r, o = chain.run(iq, t), oracle_rx(cfg, iq, t, keep=True)
        assert len(o["d"]) == len(cfg.indices(t % 10))
        assert np.abs(r["d"] - o["d"]).max() <= 1e-3 * max(1.0, np.abs(o["d"]).max()), t
        diff = np.abs(r["e"].astype(np.int32) - o["e"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() <= 0.08, (t, diff.max(), (diff != 0).mean())
        assert r["ok"] == o["ok"], t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        nok += r["ok"]
    assert nok > 0 and nspecial >= 1


@pytest.mark.parametrize("prb,mod,tbs,nrx,npt,snr,llr8", [(25, 2, 4008, 1, 1, 3.0, False), (100, 3, 75376, 1, 1, 17.2, False), (50, 3, 11448, 1, 2, 8.2, True),
                                                            (100, 3, 75376, 2, 2, 12.0, False), (6, 1, 152, 1, 1, -6.0, False)])
def test_harq_retransmissions_vs_reference_pdsch_decode(prb, mod, tbs, nrx, npt, snr, llr8):
    """HARQ: the same transport block sent with rv 0, 2, 3, 1 in different subframes, the reference's srslte_pdsch_decode keeping its
    srslte_softbuffer_rx_t between them (reset only for new data) vs the oracle chain with its OrcHarq: soft combining in
    srslte_rm_turbo_rx_lut, blocks whose CRC passed are skipped and their bytes kept (sch.c:299-414)."""
    from lte_sim import OrcHarq, RefPdsch
    rng = np.random.default_rng(600 + prb + mod + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, llr8=llr8)
    chain = RefPdsch(cfg)
    nok, nretx, nskipped = 0, 0, 0
    for trial in range(4):
        h, data = OrcHarq(cfg), None
        for n, (rv, t) in enumerate(((0, 1 + trial), (2, 9 + trial), (3, 15), (1, 20))):
            iq, data = make_subframe(cfg, t, rng, snr_db=snr, amp=0.1, rv=rv, data=data)
            r, o = chain.run(iq, t, rv=rv, new_data=n == 0), oracle_rx(cfg, iq, t, harq=h, rv=rv, new_data=n == 0)
            assert r["ok"] == o["ok"], (trial, n)
            nskipped += int((o["iters"] == 0).sum())
            if r["ok"]:
                assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
                nok += 1
                nretx += n > 0
                break
    # 8-bit soft buffers accumulate with wrapping int8 sums (rm_turbo.c:428-465): combining rarely helps there, as upstream
    assert nok > 0 and (nretx > 0 or llr8)
    if prb == 100 and npt == 1:
        assert nskipped > 0  # some block passed early and was carried over


@pytest.mark.parametrize("cell_id,prb", [(1, 6), (77, 25), (301, 100)])
def test_ul_dmrs_pusch_vs_ref(cell_id, prb):
    """srslte_refsignal_dmrs_pusch_gen (refsignal_ul.c:459-487): every float of the sequence, incl. group / sequence hopping and all
    cyclic shifts, from the tabulated QPSK sequences of 1- and 2-PRB grants to 100 PRB. At 100 PRB the exponent's argument reaches 4e6 rad:
    one different rounding is a 0.5 rad phase error, so this pins the exact operation order of the reference build."""
    from _libs import OrcUlDmrs, OrcUlDmrsCfg
    R = ref()
    q = opaque(1 << 16)
    assert R.srslte_refsignal_ul_init(q, prb) == 0 and R.srslte_refsignal_ul_set_cell(q, RefCell(prb, 1, cell_id, 0, 0, 0, 0)) == 0
    o = OrcUlDmrs()
    assert oracle().orc_ul_dmrs_init(C.byref(o), cell_id) == 0
    worst = 0.0
    for cfg in (OrcUlDmrsCfg(0, 0, False, False), OrcUlDmrsCfg(3, 7, True, False), OrcUlDmrsCfg(7, 29, False, True), OrcUlDmrsCfg(5, 13, True, True)):
        for L in sorted({1, 2, 3, 4, 6, prb // 2 // 1 if oracle().orc_dft_precoding_valid_prb(prb // 2) else 3, prb if oracle().orc_dft_precoding_valid_prb(prb) else 6}):
            for sf_idx, n_dmrs in ((0, 0), (3, 5), (9, 7)):
                a, b = aligned(2 * 2 * 12 * L, np.float32), np.zeros(2 * 12 * L, np.complex64)
                assert R.srslte_refsignal_dmrs_pusch_gen(q, C.byref(cfg), L, sf_idx, n_dmrs, p(a)) == 0
                assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(cfg), L, sf_idx, n_dmrs, p(b)) == 0
                worst = max(worst, float(np.abs(a.view(np.complex64) - b).max()))
    assert worst <= 2e-6, worst
    R.srslte_refsignal_ul_free(q)


@pytest.mark.parametrize("cell_id,prb,L,n_prb", [(1, 6, 4, 1), (77, 25, 25, 0), (301, 100, 100, 0), (12, 50, 20, 17)])
def test_chest_ul_pusch_vs_ref(cell_id, prb, L, n_prb):
    """srslte_chest_ul_estimate_pusch (chest_ul.c:268-327) with the init defaults: ce on the granted PRBs of all 14 symbols, noise, SNR."""
    from _libs import OrcChestUlRes, OrcUlDmrs, OrcUlDmrsCfg, RefChestUlRes, ref_pusch_cfg, ref_ul_sf_cfg
    R, rng = ref(), np.random.default_rng(cell_id + prb)
    q = opaque(1 << 16)
    assert R.srslte_chest_ul_init(q, prb) == 0 and R.srslte_chest_ul_set_cell(q, RefCell(prb, 1, cell_id, 0, 0, 0, 0)) == 0
    dcfg = OrcUlDmrsCfg(2, 4, True, False)
    R.srslte_chest_ul_pregen(q, C.byref(dcfg))
    o = OrcUlDmrs()
    oracle().orc_ul_dmrs_init(C.byref(o), cell_id)
    nre, n = 12 * prb, 14 * 12 * prb
    for tti, n_dmrs, nz in ((4, 3, 0.05), (19, 0, 0.3)):
        r = np.zeros(2 * 12 * L, np.complex64)
        assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(dcfg), L, tti % 10, n_dmrs, p(r)) == 0
        grid = (0.5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        k = np.arange(12 * L)
        h = ((1.5 + 0.4 * np.sin(k / 30.0)) * np.exp(1j * (0.4 + k / 150.0))).astype(np.complex64)
        for s_, sym in enumerate((3, 10)):
            grid[sym * nre + 12 * n_prb: sym * nre + 12 * (n_prb + L)] = r[s_ * 12 * L:(s_ + 1) * 12 * L] * h
        grid = acopy((grid + nz * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        ce_r, res = aligned(2 * n, np.float32), RefChestUlRes()
        ce_r[:] = 0
        res.ce = ce_r.ctypes.data
        assert R.srslte_chest_ul_estimate_pusch(q, ref_ul_sf_cfg(tti), ref_pusch_cfg(L, n_prb, n_dmrs), p(grid), C.byref(res)) == 0
        ce_o, ores = np.zeros(n, np.complex64), OrcChestUlRes()
        assert oracle().orc_chest_ul_pusch(p(r), prb, L, n_prb, p(grid), p(ce_o), C.byref(ores)) == 0
        a = ce_r.view(np.complex64)
        assert np.abs(a - ce_o).max() <= 1e-4 * np.abs(a).max()
        for nm in ("noise_estimate", "noise_estimate_dbm", "snr", "snr_db"):
            x, y = getattr(res, nm), getattr(ores, nm)
            assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, x, y)
    R.srslte_chest_ul_free(q)


@pytest.mark.parametrize("cell_id,prb,L,n0,n1", [(1, 6, 2, 0, 4), (77, 25, 10, 12, 1), (301, 100, 48, 2, 50), (12, 50, 20, 30, 0)])
def test_chest_ul_pusch_intra_subframe_hopping_vs_ref(cell_id, prb, L, n0, n1, capfd):
    """srslte_pusch_grant_t.n_prb[0] != n_prb[1]: the reference's estimator prints "intra-subframe frequency hopping not supported" and
    goes on (chest_ul.c:293-295); with the per-slot copy it is compiled with (DO_LINEAR_INTERPOLATION undefined, :244-258) every slot
    is estimated and filled at its own PRB offset, which is what the oracle's orc_chest_ul_pusch_hop restates."""
    from _libs import OrcChestUlRes, OrcUlDmrs, OrcUlDmrsCfg, RefChestUlRes, ref_pusch_cfg, ref_ul_sf_cfg
    R, rng = ref(), np.random.default_rng(cell_id + prb)
    q = opaque(1 << 16)
    assert R.srslte_chest_ul_init(q, prb) == 0 and R.srslte_chest_ul_set_cell(q, RefCell(prb, 1, cell_id, 0, 0, 0, 0)) == 0
    dcfg = OrcUlDmrsCfg(1, 7, False, True)
    R.srslte_chest_ul_pregen(q, C.byref(dcfg))
    o = OrcUlDmrs()
    oracle().orc_ul_dmrs_init(C.byref(o), cell_id)
    nre, n = 12 * prb, 14 * 12 * prb
    for tti, n_dmrs, nz in ((7, 5, 0.05), (12, 1, 0.25)):
        r = np.zeros(2 * 12 * L, np.complex64)
        assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(dcfg), L, tti % 10, n_dmrs, p(r)) == 0
        grid = (0.5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        k = np.arange(12 * L)
        for s_, (sym, npb) in enumerate(((3, n0), (10, n1))):
            h = ((1.5 - 0.5 * s_ + 0.4 * np.sin(k / 30.0)) * np.exp(1j * (0.4 + s_ + k / 150.0))).astype(np.complex64)
            grid[sym * nre + 12 * npb: sym * nre + 12 * (npb + L)] = r[s_ * 12 * L:(s_ + 1) * 12 * L] * h
        grid = acopy((grid + nz * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        ce_r, res = aligned(2 * n, np.float32), RefChestUlRes()
        ce_r[:] = 0
        res.ce = ce_r.ctypes.data
        assert R.srslte_chest_ul_estimate_pusch(q, ref_ul_sf_cfg(tti), ref_pusch_cfg(L, n0, n_dmrs, n1), p(grid), C.byref(res)) == 0
        ce_o, ores = np.zeros(n, np.complex64), OrcChestUlRes()
        oracle().orc_chest_ul_pusch_hop.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        assert oracle().orc_chest_ul_pusch_hop(p(r), prb, L, n0, n1, p(grid), p(ce_o), C.byref(ores)) == 0
        a = ce_r.view(np.complex64)
        assert np.abs(a - ce_o).max() <= 1e-4 * np.abs(a).max()
        filled = np.zeros(n, bool)
        for sym in range(14):
            npb = n0 if sym < 7 else n1
            filled[sym * nre + 12 * npb: sym * nre + 12 * (npb + L)] = True
        assert not a[~filled].any() and np.abs(a[filled]).min() > 0
        for nm in ("noise_estimate", "noise_estimate_dbm", "snr", "snr_db"):
            x, y = getattr(res, nm), getattr(ores, nm)
            assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, x, y)
    R.srslte_chest_ul_free(q)
    C.CDLL(None).fflush(None)
    capfd.readouterr()  # the reference's complaint, once per call


@pytest.mark.parametrize("prb,L,n0,n1,mod,tbs,snr", [(25, 10, 5, 14, 2, 4008, 9.5), (100, 48, 50, 1, 3, 30576, 17.0), (6, 2, 0, 4, 1, 256, 5.0)])
def test_pusch_chain_with_hopping_vs_reference_code(prb, L, n0, n1, mod, tbs, snr, capfd):
    """The receive chain on the reference's compiled stages with a different PRB offset in each slot (what pusch_cp / pusch_get does with
    grant.n_prb_tilde[slot], pusch.c:52-91): same transport blocks, CRC flags and pass counts as the oracle chain."""
    rng = np.random.default_rng(950 + prb + L)
    cfg = UlConfig(prb, 11, mod, tbs, L, n0, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, n_prb_slot1=n1)
    chain = RefUlRx(cfg)
    nok = 0
    for t in (0, 4, 9):
        iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j))
        r, o = chain.run(iq, t), oracle_ul_rx(cfg, iq, t)
        assert r["ok"] == o["ok"] and np.array_equal(r["iters"], o["iters"])
        if r["ok"]:
            nok += 1
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
    assert nok > 0
    C.CDLL(None).fflush(None)
    capfd.readouterr()


@pytest.mark.parametrize("cell_id,prb,L,n0,n1", [(1, 6, 2, 0, 4), (77, 25, 10, 12, 12), (301, 100, 48, 2, 50), (150, 50, 50, 0, 0)])
def test_ul_dmrs_and_chest_ul_extended_cp_vs_ref(cell_id, prb, L, n0, n1, capfd):
    """An extended-CP cell on the uplink: the cyclic-shift hopping n_PRS is read at a stride of 8 x 6 bits (refsignal_ul.c:127-133), the DMRS sit in
    symbol 2 of each slot (SRSLTE_REFSIGNAL_UL_L, refsignal_ul.h:43), the estimate is copied over the 6 symbols of its slot (chest_ul.c:244-258)."""
    from _libs import OrcChestUlRes, OrcUlDmrs, OrcUlDmrsCfg, RefChestUlRes, ref_pusch_cfg, ref_ul_sf_cfg
    R, rng = ref(), np.random.default_rng(cell_id + prb)
    cell = RefCell(prb, 1, cell_id, 1, 0, 0, 0)  # SRSLTE_CP_EXT
    rs = opaque(1 << 16)
    assert R.srslte_refsignal_ul_init(rs, prb) == 0 and R.srslte_refsignal_ul_set_cell(rs, cell) == 0
    q = opaque(1 << 16)
    assert R.srslte_chest_ul_init(q, prb) == 0 and R.srslte_chest_ul_set_cell(q, cell) == 0
    dcfg = OrcUlDmrsCfg(3, 7, True, False)
    R.srslte_chest_ul_pregen(q, C.byref(dcfg))
    o, o7 = OrcUlDmrs(), OrcUlDmrs()
    assert oracle().orc_ul_dmrs_init_cp(C.byref(o), cell_id, 6) == 0 and oracle().orc_ul_dmrs_init(C.byref(o7), cell_id) == 0
    nre, n = 12 * prb, 12 * 12 * prb
    differs = 0
    for tti, n_dmrs, nz in ((7, 5, 0.05), (12, 1, 0.25), (9, 7, 0.1)):
        a, r, r7 = aligned(2 * 2 * 12 * L, np.float32), np.zeros(2 * 12 * L, np.complex64), np.zeros(2 * 12 * L, np.complex64)
        assert R.srslte_refsignal_dmrs_pusch_gen(rs, C.byref(dcfg), L, tti % 10, n_dmrs, p(a)) == 0
        assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(dcfg), L, tti % 10, n_dmrs, p(r)) == 0
        assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o7), C.byref(dcfg), L, tti % 10, n_dmrs, p(r7)) == 0
        assert np.abs(a.view(np.complex64) - r).max() <= 2e-6
        differs += int(np.abs(r - r7).max() > 0.1)
        grid = (0.5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        k = np.arange(12 * L)
        for s_, (sym, npb) in enumerate(((2, n0), (8, n1))):
            h = ((1.5 - 0.5 * s_ + 0.4 * np.sin(k / 30.0)) * np.exp(1j * (0.4 + s_ + k / 150.0))).astype(np.complex64)
            grid[sym * nre + 12 * npb: sym * nre + 12 * (npb + L)] = r[s_ * 12 * L:(s_ + 1) * 12 * L] * h
        grid = acopy((grid + nz * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        ce_r, res = aligned(2 * n, np.float32), RefChestUlRes()
        ce_r[:] = 0
        res.ce = ce_r.ctypes.data
        assert R.srslte_chest_ul_estimate_pusch(q, ref_ul_sf_cfg(tti), ref_pusch_cfg(L, n0, n_dmrs, n1), p(grid), C.byref(res)) == 0
        ce_o, ores = np.zeros(n, np.complex64), OrcChestUlRes()
        oracle().orc_chest_ul_pusch_hop_cp.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        assert oracle().orc_chest_ul_pusch_hop_cp(p(r), prb, L, n0, n1, 6, p(grid), p(ce_o), C.byref(ores)) == 0
        av = ce_r.view(np.complex64)
        assert np.abs(av - ce_o).max() <= 1e-4 * np.abs(av).max()
        filled = np.zeros(n, bool)
        for sym in range(12):
            npb = n0 if sym < 6 else n1
            filled[sym * nre + 12 * npb: sym * nre + 12 * (npb + L)] = True
        assert not av[~filled].any() and np.abs(av[filled]).min() > 0
        for nm in ("noise_estimate", "noise_estimate_dbm", "snr", "snr_db"):
            x, y = getattr(res, nm), getattr(ores, nm)
            assert abs(x - y) <= 1e-4 * abs(x) + 1e-6, (nm, x, y)
    assert differs > 0  # the hopping stride is the CP's: the normal-CP sequences are other sequences
    R.srslte_chest_ul_free(q)
    R.srslte_refsignal_ul_free(rs)
    C.CDLL(None).fflush(None)
    capfd.readouterr()


@pytest.mark.parametrize("prb,L,n0,n1,mod,tbs,snr,short", [(25, 10, 5, 5, 2, 3240, 9.5, False), (100, 48, 50, 1, 3, 24496, 17.0, False), (6, 2, 0, 4, 1, 208, 5.0, True)])
def test_pusch_chain_extended_cp_vs_reference_code(prb, L, n0, n1, mod, tbs, snr, short, capfd):
    """The eNB receive chain on an extended-CP cell: 12 symbols per subframe, DMRS in symbols 2 and 8, 10 data symbols (9 shortened, ra_ul.c:234)
    - reference-code chain and oracle chain on identical IQ give the same transport blocks, CRC flags and pass counts."""
    rng = np.random.default_rng(990 + prb + L)
    cfg = UlConfig(prb, 11, mod, tbs, L, n0, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, n_prb_slot1=n1, shortened=short, cp_ext=True)
    assert cfg.nsymb == (9 if short else 10) and cfg.data_syms[:3] == [0, 1, 3]
    chain = RefUlRx(cfg)
    nok = 0
    for t in (0, 4, 9):
        iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j))
        r, o = chain.run(iq, t), oracle_ul_rx(cfg, iq, t)
        assert r["ok"] == o["ok"] and np.array_equal(r["iters"], o["iters"])
        if r["ok"]:
            nok += 1
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
    assert nok > 0
    C.CDLL(None).fflush(None)
    capfd.readouterr()


@pytest.mark.parametrize("prb,L,mod,tbs,snr,short,O_ri,I_ri,O_ack,I_ack", [(25, 10, 2, 3240, 12.0, False, 1, 9, 2, 9), (6, 6, 1, 808, 6.0, True, 1, 5, 1, 5),
                                                                          (100, 48, 3, 24496, 19.0, False, 2, 8, 1, 8), (50, 20, 2, 6200, 12.0, True, 0, 0, 2, 10)])
def test_uci_on_pusch_extended_cp_vs_reference_ulsch_functions(prb, L, mod, tbs, snr, short, O_ri, I_ri, O_ack, I_ack):
    """HARQ-ACK and rank indication with grant.nof_symb = 10 / 9: the column sets {1, 2, 6, 7} and {0, 3, 5, 8} of uci.c:502,:527 (chosen by
    N_pusch_symbs <= 10), against srslte_ulsch_encode / srslte_ulsch_decode."""
    from lte_sim import RefUlsch, UlConfig, make_ul_subframe, oracle_ul_rx, ul_ri_layout
    rng = np.random.default_rng(1300 + prb + L + O_ri + O_ack)
    cfg = UlConfig(prb, 11, mod, tbs, L, (prb - L) // 2, n_dmrs=3, shortened=short, cp_ext=True)
    chain = RefUlsch(cfg, O_ack, I_ack, O_ri, I_ri)
    Qp_ri, lut, ri_mask, G = ul_ri_layout(cfg, O_ri, I_ri)
    n_ok = 0
    for t, ri, ack in ((2, 1, (1, 0)), (7, 0, (0, 1)), (9, 1, (1, 1))):
        ack = ack[:O_ack]
        k = {}
        iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, keep=k, ack=ack, I_offset_ack=I_ack, ri=(ri, 0)[:O_ri], I_offset_ri=I_ri)
        g_r, q_r = chain.encode(data, ack, ri if O_ri else None)
        assert np.array_equal(g_r[:G], k["g"]), "UL-SCH bits rate-matched to G = %d" % G
        c = cfg.scramble(t % 10)
        ack_pos = k["ack_types"] >= 0
        q_plain = np.zeros(cfg.nbits, np.uint8)
        q_plain[~ri_mask] = k["g"][lut[~ri_mask]]
        sel = ~ri_mask & ~ack_pos
        assert np.array_equal(q_r[sel], q_plain[sel]), t
        # the reference leaves the value bits in the ACK / RI positions of q (0 for repetitions and placeholders): same positions as the oracle's
        vb = np.flatnonzero(k["ack_types"] == 1)
        assert q_r[vb].all()
        o = oracle_ul_rx(cfg, iq, t, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri)
        r = chain.decode(o["q_before_ack"], c)
        if O_ri:
            assert r["ri"] == o["ri"][0] == ri, (t, r["ri"], o["ri"])
        assert np.array_equal(r["ack"][:O_ack], o["ack"][:O_ack]) and np.array_equal(o["ack"][:O_ack], np.array(ack, np.uint8))
        assert np.array_equal(r["g"][:G], o["g"]) and r["ok"] == o["ok"], t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        n_ok += r["ok"]
    assert n_ok > 0


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,snr", [(6, 6, 0, 1, 1000, 3.5), (25, 10, 5, 2, 4008, 9.5), (100, 100, 0, 2, 43816, 12.5), (100, 48, 20, 3, 30576, 17.0)])
def test_pusch_chain_vs_reference_code(prb, L, n_prb, mod, tbs, snr):
    """eNB PUSCH receive chain: reference-code chain vs oracle chain on identical IQ - same TBs, CRC flags, pass counts."""
    rng = np.random.default_rng(900 + prb + L)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True)
    chain = RefUlRx(cfg)
    nok = 0
    for t in (0, 4, 9):
        iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j))
        r, o = chain.run(iq, t), oracle_ul_rx(cfg, iq, t)
        assert r["ok"] == o["ok"] and np.array_equal(r["iters"], o["iters"])
        if r["ok"]:  # a block that never converges turns the 12-bit reciprocal of the reference's equaliser into different garbage
            nok += 1
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
    assert nok > 0


@pytest.mark.parametrize("prb,L,mod,tbs,snr,short", [(6, 6, 1, 1000, 3.5, False), (25, 10, 2, 4008, 9.5, False), (100, 100, 2, 43816, 12.5, False),
                                                       (100, 48, 3, 30576, 17.0, False), (50, 45, 3, 30576, 18.5, False), (25, 10, 2, 4008, 10.5, True),
                                                       (100, 96, 3, 61664, 19.5, True), (15, 1, 1, 104, 5.0, True)])
def test_ulsch_functions_vs_oracle_chain(prb, L, mod, tbs, snr, short):
    """The reference's own srslte_ulsch_encode and srslte_ulsch_decode (sch.c:991-1160, no UCI): coded bits g, interleaved bits q
    (36.212 5.2.2.8) and, on the receive side, the de-interleaved LLRs, CRC result and transport block, against the oracle chain's
    UL-SCH coder / UlConfig.q_of_g / decoder on identical inputs."""
    from lte_sim import RefUlsch, UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(700 + prb + L)
    cfg = UlConfig(prb, 11, mod, tbs, L, (prb - L) // 2, n_dmrs=3, shortened=short)  # short: 11 data symbols (grant.nof_symb), SRS in the last
    chain = RefUlsch(cfg)
    nok = 0
    for t in (2, 7, 9):
        k = {}
        iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, keep=k)
        g_r, q_r = chain.encode(data)
        q_o = np.zeros(cfg.nbits, np.uint8)
        q_o[cfg.q_of_g] = k["g"]
        assert np.array_equal(g_r, k["g"]) and np.array_equal(q_r, q_o), t
        o = oracle_ul_rx(cfg, iq, t, keep=True)
        r = chain.decode(o["q"], cfg.scramble(t % 10))
        assert np.array_equal(r["g"], o["g"]) and r["ok"] == o["ok"], t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        nok += r["ok"]
    assert nok > 0


@pytest.mark.parametrize("prb,mod,tbs,npt", [(6, 1, 152, 1), (25, 2, 4008, 2), (100, 3, 75376, 1), (100, 3, 75376, 2), (50, 4, 48936, 1), (15, 1, 1000, 2),
                                             (25, 2, 4008, 4), (100, 3, 61664, 4)])
def test_pdsch_encode_function_vs_stimulus_generator(prb, mod, tbs, npt):
    """The reference's own srslte_pdsch_encode (pdsch.c:1059-1185) vs the transmit chain make_subframe builds from oracle pieces (DL-SCH
    coding incl. the Qm * N_L block split, scrambling, modulation, SFBC precoding, RE mapping), port by port on the resource grid;
    rv 0 and a retransmission version. With p_a = 0 the reference transmits a 2-port cell at rho_a = sqrt(2)."""
    from lte_sim import RefPdschTx
    rng = np.random.default_rng(800 + prb + mod + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_ports=npt)
    chain = RefPdschTx(cfg)
    for t, rv in ((0, 0), (3, 0), (5, 2), (8, 1)):
        k = {}
        _, data = make_subframe(cfg, t, rng, rv=rv, keep=k)
        grids = chain.run(data, t, rv=rv)
        for port in range(npt):
            exp = np.zeros(cfg.grid_len, np.complex64)
            exp[k["idx"]] = k["y"][port] * (np.sqrt(2.0) if npt > 1 else 1.0)
            assert np.abs(grids[port] - exp).max() <= 1e-6, (t, rv, port)


@pytest.mark.parametrize("prb,L,mod,tbs,snr,short,O_ack,I_off", [(25, 10, 2, 4008, 9.5, False, 1, 8), (25, 10, 1, 2216, 4.0, False, 2, 5), (100, 100, 2, 43816, 12.5, False, 1, 10),
                                                                  (100, 48, 3, 30576, 17.5, False, 2, 9), (50, 45, 3, 30576, 19.0, True, 1, 12), (6, 6, 1, 1000, 4.5, True, 2, 3),
                                                                  (15, 1, 2, 104, 8.0, False, 1, 14)])
def test_harq_ack_on_pusch_vs_reference_ulsch_functions(prb, L, mod, tbs, snr, short, O_ack, I_off):
    """HARQ-ACK (1 and 2 bits) multiplexed on the PUSCH: Q' and the ACK positions / value bits srslte_ulsch_encode writes into the
    interleaved stream (sch.c:1170-1215 with srslte_uci_encode_ack_ri), and on the receive side the ACK decisions, the zeroed ACK
    positions and the transport block of srslte_ulsch_decode (uci_decode_ri_ack, sch.c:929-966) - against orc_uci.c on identical inputs."""
    from lte_sim import RefUlsch, UlConfig, make_ul_subframe, oracle_ul_rx
    orc = oracle()
    rng = np.random.default_rng(1000 + prb + L + O_ack)
    cfg = UlConfig(prb, 11, mod, tbs, L, (prb - L) // 2, n_dmrs=3, shortened=short)
    chain = RefUlsch(cfg, O_ack, I_off)
    Qp = orc.orc_uci_ack_qprime(O_ack, I_off, L, cfg.nsymb, cfg.seg.C * cfg.seg.K1)
    assert Qp > 0
    n_ok = 0
    for t, ack in ((2, (1, 0)), (7, (0, 1)), (9, (1, 1))):
        ack = ack[:O_ack]
        k = {}
        iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, keep=k, ack=ack, I_offset_ack=I_off)
        # transmit side: the reference's pre-scrambling q stream against the data stream with the ACK value bits put in
        g_r, q_r = chain.encode(data, ack)
        c = cfg.scramble(t % 10)
        q_tx = k["q_tx"]  # oracle: interleaved, scrambled, ACK inserted
        val = np.array([q_tx[i] ^ c[i] for i in range(cfg.nbits)], np.uint8)  # unscrambled view
        types = k["ack_types"]  # per q position: -1 none, 0/1 value, 2 repetition, 3 placeholder
        sel = types < 0
        assert np.array_equal(q_r[sel], val[sel]), t
        vb = (types == 0) | (types == 1)
        assert vb.sum() > 0 and np.array_equal(q_r[vb], types[vb].astype(np.uint8)) and not q_r[(types == 2) | (types == 3)].any()
        # receive side
        o = oracle_ul_rx(cfg, iq, t, keep=True, O_ack=O_ack, I_offset_ack=I_off)
        r = chain.decode(o["q_before_ack"], c)
        assert np.array_equal(r["ack"][:O_ack], o["ack"][:O_ack]) and np.array_equal(o["ack"][:O_ack], np.array(ack, np.uint8)), (t, r["ack"], o["ack"])
        assert np.array_equal(r["g"], o["g"]) and r["ok"] == o["ok"], t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        n_ok += r["ok"]
    assert n_ok > 0


@pytest.mark.parametrize("prb,L,mod,tbs,snr,short,O_ri,I_ri,O_ack,I_ack", [(25, 10, 2, 4008, 12.0, False, 1, 9, 0, 0), (25, 10, 2, 4008, 12.0, False, 1, 9, 2, 9),
                                                                          (6, 6, 1, 1000, 6.0, True, 1, 5, 1, 5), (100, 48, 3, 30576, 19.0, False, 1, 12, 0, 0),
                                                                          (50, 20, 2, 7736, 12.0, False, 2, 8, 1, 8), (100, 100, 2, 43816, 15.0, True, 1, 11, 2, 12)])
def test_ri_on_pusch_vs_reference_ulsch_functions(prb, L, mod, tbs, snr, short, O_ri, I_ri, O_ack, I_ack):
    """Rank indication (with and without HARQ-ACK) on the PUSCH against srslte_ulsch_encode / srslte_ulsch_decode: the RI symbols the
    channel interleaver leaves out (sch.c:580-598), the UL-SCH rate-matched to the rest, the RI decision, and the de-interleaved LLRs
    including the reference's g[0], which its scatter leaves holding the last RI position's LLR (sch.c:891-918)."""
    from lte_sim import RefUlsch, UlConfig, make_ul_subframe, oracle_ul_rx, ul_ri_layout
    rng = np.random.default_rng(1200 + prb + L + O_ri + O_ack)
    cfg = UlConfig(prb, 11, mod, tbs, L, (prb - L) // 2, n_dmrs=3, shortened=short)
    chain = RefUlsch(cfg, O_ack, I_ack, O_ri, I_ri)
    Qp_ri, lut, ri_mask, G = ul_ri_layout(cfg, O_ri, I_ri)
    assert Qp_ri > 0 and ri_mask.sum() == Qp_ri * cfg.Qm
    n_ok = 0
    for t, ri, ack in ((2, 1, (1, 0)), (7, 0, (0, 1)), (9, 1, (1, 1))):
        ack = ack[:O_ack]
        k = {}
        iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, keep=k, ack=ack, I_offset_ack=I_ack, ri=(ri, 0)[:O_ri], I_offset_ri=I_ri)
        g_r, q_r = chain.encode(data, ack, ri)
        assert np.array_equal(g_r[:G], k["g"]), "UL-SCH bits rate-matched to G = %d" % G
        c = cfg.scramble(t % 10)
        ack_pos = k["ack_types"] >= 0 if O_ack else np.zeros(cfg.nbits, bool)
        q_plain = np.zeros(cfg.nbits, np.uint8)
        q_plain[~ri_mask] = k["g"][lut[~ri_mask]]
        sel = ~ri_mask & ~ack_pos
        assert np.array_equal(q_r[sel], q_plain[sel]), t
        first = np.flatnonzero(ri_mask)[::cfg.Qm]  # first bit of every RI symbol carries a value bit
        if O_ri == 1:
            assert (q_r[first] == ri).all() and not q_r[np.setdiff1d(np.flatnonzero(ri_mask), first)].any()
        o = oracle_ul_rx(cfg, iq, t, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri)
        r = chain.decode(o["q_before_ack"], c)
        assert r["ri"] == o["ri"][0] == ri, (t, r["ri"], o["ri"])
        assert np.array_equal(r["ack"][:O_ack], o["ack"][:O_ack])
        assert np.array_equal(r["g"][:G], o["g"]) and r["ok"] == o["ok"], t
        if r["ok"]:
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
        n_ok += r["ok"]
    assert n_ok > 0


@pytest.mark.parametrize("prb,cell_id,area,mod,tbs,cfi,region,snr,nrx,cp_ext", [(6, 1, 1, 1, 488, 2, 2, 7.0, 1, True), (25, 7, 3, 2, 4584, 2, 2, 12.0, 1, True),
                                                                                (50, 101, 200, 3, 15264, 1, 1, 19.0, 1, False),
                                                                                (100, 301, 77, 2, 18336, 2, 2, 11.0, 2, False),
                                                                                (15, 44, 255, 2, 2216, 1, 1, 12.0, 1, True)])
def test_pmch_encode_decode_vs_oracle_chain(prb, cell_id, area, mod, tbs, cfi, region, snr, nrx, cp_ext):
    """The reference's srslte_pmch_encode and srslte_pmch_decode (pmch.c:291-483) with its MBSFN channel estimate against the oracle's PMCH chain
    (SURVEY §8f N4's pmch_test): the RE mapping of pmch_cp around the MBSFN reference signal, the area's scrambling sequence, the single-port
    equaliser with the MBSFN noise estimate, LLRs, transport blocks and CRC verdicts on identical time samples."""
    from lte_sim import PmchConfig, RefPmch, make_pmch_subframe, oracle_pmch_rx
    rng = np.random.default_rng(3300 + prb + area)
    cfg = PmchConfig(prb, cell_id, area, mod, tbs, cfi=cfi, non_mbsfn_region=region, nof_rx=nrx, cp_ext=cp_ext)
    assert cfg.nof_re == ((6 - cfg.lstart) * 12 - (6 if cfg.lstart <= 2 else 0) + 60) * prb  # ra_re_x_prb with sf_type MBSFN (ra_dl.c:50-158)
    chain = RefPmch(cfg)
    nok = 0
    for t in (1, 3, 8, 12):
        k = {}
        iq, data = make_pmch_subframe(cfg, t, rng, snr_db=snr, amp=0.1, keep=k)
        g_ref = chain.encode(data, t)
        mask = np.zeros(cfg.grid_len, bool)
        mask[cfg.idx] = True
        assert np.array_equal(g_ref[mask].view(np.float32), k["d"].view(np.float32)) and not g_ref[~mask].any(), t  # symbols and where they go
        r, o = chain.decode(iq, t), oracle_pmch_rx(cfg, iq, t, keep=True)
        for a in range(nrx):
            assert np.abs(r["ce"][a] - o["ce"][a]).max() <= 1e-4 * np.abs(r["ce"][a]).max(), (t, a)
        assert abs(r["noise"] - o["noise"]) <= 1e-4 * abs(r["noise"]), t
        assert np.abs(r["d"] - o["d"]).max() <= 2e-4 * np.abs(r["d"]).max(), t
        diff = np.abs(r["e"].astype(np.int32) - o["e"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 2e-3 * diff.size + 1, (t, int(diff.max()), int((diff != 0).sum()))
        if diff.max() == 0 or (r["ok"] and o["ok"]):
            assert r["ok"] == o["ok"], t
        if r["ok"] and o["ok"]:
            nok += 1
            assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data), t
    assert nok >= 2


def test_reference_mbsfn_estimate_without_interpolate_subframe_reads_stale_symbols():
    """Fact about the reference, recorded because libsrslte_phy_hip.so REFUSES this call (SRSLTE_ERROR + message): with sf_type MBSFN and
    interpolate_subframe off, chest_dl.c:430-433 interpolates symbol 0 only in frequency, yet the MBSFN time interpolation of :475-479
    still runs between symbols 0, 2, 6 and 10 of `ce` - three of which this call never wrote. The result is a function of what the
    caller's buffer held before: the same grid estimated into two differently prefilled buffers gives two different answers (and the
    pilot averaging of :527-545 walks the 2 + 3 x 6 references per PRB as if they were CRS rows). There is nothing to be bit-exact with."""
    R, rng = ref(), np.random.default_rng(4242)
    prb, cid, area, n = 25, 7, 3, 14 * 12 * 25
    q = opaque(1 << 20)
    assert R.srslte_chest_dl_init(q, prb, 1) == 0 and R.srslte_chest_dl_set_cell(q, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
    assert R.srslte_chest_dl_set_mbsfn_area_id(q, area) == 0
    grid = acopy(((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64).view(np.float32))
    outs = []
    for fill in (0.0, 5.0):
        rc, res, sf = RefChestCfg(), RefChestRes(), RefDlSfCfg()
        rc.interpolate_subframe, rc.mbsfn_area_id = False, area
        ce = aligned(2 * n, np.float32)
        ce[:] = fill
        res.ce[0][0] = ce.ctypes.data
        sf.tti, sf.sf_type = 1, 1
        assert R.srslte_chest_dl_estimate_cfg(q, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
        outs.append(ce.copy())
    sym = lambda a, l: a.view(np.complex64)[l * 12 * prb:(l + 1) * 12 * prb]
    assert np.array_equal(sym(outs[0], 0), sym(outs[1], 0))            # what the call does compute
    for l in (1, 3, 7, 11):                                            # interpolated FROM symbols 2 / 6 / 10, which nothing wrote
        assert not np.allclose(sym(outs[0], l), sym(outs[1], l)), l
    R.srslte_chest_dl_free(q)


def _ref_ulsch_round_trip(tbs):
    """the reference's srslte_ulsch_encode, its q bits as clean LLRs into its srslte_ulsch_decode: (encode return code, decoded ok, bytes equal)"""
    from lte_sim import RefUlsch, UlConfig
    cfg = UlConfig(50, 3, 2, 4008, 40, 0)
    cfg.tbs = tbs
    chain = RefUlsch(cfg)
    R = chain.R
    data = np.random.default_rng(tbs).integers(0, 256, tbs // 8, dtype=np.uint8)
    chain.pc[chain.rv_off:chain.rv_off + 4].view(np.uint32)[0] = 0
    chain.pc[chain.sb_off:chain.sb_off + 8].view(np.uint64)[0] = C.addressof(chain.sb_tx)
    R.srslte_softbuffer_tx_reset(chain.sb_tx)
    d = np.zeros(tbs // 8 + 64, np.uint8)
    d[:tbs // 8] = data
    uci, g, q = np.zeros(4096, np.uint8), np.zeros(cfg.nbits // 8 + 64, np.uint8), np.zeros(cfg.nbits // 8 + 64, np.uint8)
    rc = R.srslte_ulsch_encode(chain.q, p(chain.pc), p(d), p(uci), p(g), p(q))
    if rc < 0:
        out = chain.decode(np.zeros(cfg.nbits, np.int16), np.zeros(cfg.nbits, np.uint8))
        return rc, out["ok"], False
    llr = ((2 * np.unpackbits(q)[:cfg.nbits].astype(np.int16) - 1) * 40).astype(np.int16)
    out = chain.decode(llr, np.zeros(cfg.nbits, np.uint8))
    return rc, out["ok"], bool(np.array_equal(out["tb"][:tbs // 8], data))


def test_reference_refuses_filler_bits_and_cannot_round_trip_two_block_sizes():
    """Facts about the reference, recorded because the batched pipelines refuse such transport-block sizes (SRSLTE_ERROR + message):
    (1) a size whose segmentation needs filler bits (36.212 5.1.2 F > 0: e.g. 4016 -> K = 4096) is refused by encode_tb_off and by decode_tb
    themselves ("Error filler bits are not supported. Use standard TBS", sch.c:193-196,:450-453);
    (2) a size that segments into TWO block lengths without filler (6264 -> 3200 + 3136; 6392 -> 3264 + 3200) is accepted, but the encoder
    puts the K2 blocks first (i < C2, sch.c:222-229) and the decoder expects the K1 blocks first (cb_idx < C1, sch.c:318-319), and the two
    compute the longer blocks' share of the bits differently (sch.c:232-236 vs :331-334): the reference's own noise-free round trip fails.
    A standard size (4008, and the non-table 6200 = 2 x 3136 exactly) goes through: the harness is sound. No table size of 36.213 needs either."""
    from _libs import OrcCbsegm
    assert _ref_ulsch_round_trip(4008) == (0, True, True)
    assert _ref_ulsch_round_trip(6200) == (0, True, True)
    rc, ok, same = _ref_ulsch_round_trip(4016)
    assert rc < 0 and not ok
    for tbs in (6264, 6392):
        seg = OrcCbsegm()
        assert oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 > 0
        rc, ok, same = _ref_ulsch_round_trip(tbs)
        assert rc == 0 and not ok and not same


def test_reference_8bit_pusch_receive_cannot_decode_a_clean_transmission():
    """Fact about the reference, recorded because the uplink receive pipelines have no 8-bit LLR option although srsenb can select one
    (`expert.pusch_8bit_decoder`, "Experimental": srsenb/src/phy/sf_worker.cc:148-149 sets pusch.llr_is_8bit and ul_sch.llr_is_8bit).
    With the flag, srslte_pusch_decode demaps and descrambles into int8 LLRs (pusch.c:481-500) and hands the buffer to
    srslte_ulsch_decode as int16_t* (pusch.c:503); uci_decode_ri_ack and ulsch_deinterleave (srslte_vec_lut_sis, sch.c:890-918,:1014-1034)
    move int16 ELEMENTS - two int8 LLRs at a time, to twice the byte offset - and decode_tb then reads the result as int8
    (srslte_rm_turbo_rx_lut_8bit, sch.c:336-340): the channel deinterleaver is not inverted. The reference's own noise-free round trip,
    fed exactly as pusch.c feeds it, fails with the flag and passes without: there is no working 8-bit PUSCH chain upstream to be equal to."""
    from _libs import ref_layout
    from lte_sim import RefUlsch, UlConfig
    cfg = UlConfig(50, 3, 2, 4008, 40, 0)
    flag = ref_layout({"srslte_sch_t": ["llr_is_8bit"]}, ["srslte/phy/phch/sch.h"])["srslte_sch_t.llr_is_8bit"]
    data = np.random.default_rng(8).integers(0, 256, cfg.tbs // 8, dtype=np.uint8)
    res = {}
    for llr8 in (False, True):
        chain = RefUlsch(cfg)
        _, q = chain.encode(data)
        C.cast(C.addressof(chain.q) + flag, C.POINTER(C.c_uint8))[0] = 1 if llr8 else 0
        llr = (2 * q.astype(np.int16) - 1) * 40
        if llr8:  # int8 LLRs at the start of the buffer, the rest of it as the demapper's buffer is left: whatever was there (here zeros)
            buf = np.zeros(cfg.nbits, np.int16)
            buf.view(np.int8)[:cfg.nbits] = llr.astype(np.int8)
            llr = buf
        out = chain.decode(llr, np.zeros(cfg.nbits, np.uint8))
        res[llr8] = (out["ok"], bool(np.array_equal(out["tb"][:cfg.tbs // 8], data)))
    assert res[False] == (True, True)
    assert res[True] == (False, False)


def test_reference_cdd_predecoder_on_a_noise_free_channel():
    """Fact about the reference, recorded because tests/test_gpu_dropin.py leaves `phy_dl_test -t 3` out: the reference's own compiled
    large-delay-CDD predecoder (mimo/precoding.c:1067-1102 -> srslte_predecoding_ccd_2x2_mmse[_csi], :915-1065), fed what that test
    feeds it - 2 layers, the "perfect crossed channel" y0 = x0 + x1, y1 = x0 - x1 (phy_dl_test.c:543-555), exact channel estimates and
    a vanishing noise estimate - returns NaN, while the spatial-multiplexing predecoder on the same inputs is exact. None of this code
    is replaced by libsrslte_phy_hip.so."""
    R = ref()
    R.srslte_predecoding_type.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    R.srslte_precoding_type.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    n, rng = 256, np.random.default_rng(0)

    def cbuf(v=None):
        b = aligned(2 * n, np.float32)
        if v is not None:
            b.view(np.complex64)[:] = v
        return b

    x = [cbuf((rng.choice([-1, 1], n) + 1j * rng.choice([-1, 1], n)) / np.sqrt(2)) for _ in range(2)]
    res = {}
    for scheme in (2, 3):  # SRSLTE_TXSCHEME_SPATIALMUX, SRSLTE_TXSCHEME_CDD (phy_common.h:233-236)
        y = [cbuf(), cbuf()]
        xp, yp = (C.c_void_p * 4)(x[0].ctypes.data, x[1].ctypes.data, 0, 0), (C.c_void_p * 4)(y[0].ctypes.data, y[1].ctypes.data, 0, 0)
        R.srslte_precoding_type(xp, yp, 2, 2, 1, n, 1.0, scheme)
        t0, t1 = y[0].view(np.complex64).copy(), y[1].view(np.complex64).copy()
        r = [cbuf(t0 + t1), cbuf(t0 - t1)]
        h = [[cbuf(np.ones(n)), cbuf(np.ones(n))], [cbuf(np.ones(n)), cbuf(-np.ones(n))]]  # h[port][rx antenna]
        hp = ((C.c_void_p * 4) * 4)()
        for i in range(2):
            for j in range(2):
                hp[i][j] = h[i][j].ctypes.data
        out = [cbuf(), cbuf()]
        op, rp = (C.c_void_p * 4)(out[0].ctypes.data, out[1].ctypes.data, 0, 0), (C.c_void_p * 4)(r[0].ctypes.data, r[1].ctypes.data, 0, 0)
        assert R.srslte_predecoding_type(rp, hp, op, None, 2, 2, 2, 1, n, scheme, 1.0, 1e-12) == 0
        res[scheme] = max(np.abs(out[k].view(np.complex64) - x[k].view(np.complex64)).max() for k in range(2))
    assert res[2] < 1e-3
    assert np.isnan(res[3])


# ---------------------------------------------------------------- two-layer modes: large-delay CDD (TM3), closed-loop multiplexing (TM4)
def _mimo_bufs(n, seed):
    rng = np.random.default_rng(seed)

    def cbuf(v=None):
        b = aligned(2 * n + 16, np.float32)[:2 * n]
        if v is not None:
            b.view(np.complex64)[:] = v
        return b

    def rc(scale=1.0):
        return (scale * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    return cbuf, rc


def _ref_predecode(R, scheme, y, h, n, nof_layers, codebook_idx, scaling, noise, cbuf):
    """srslte_predecoding_type (precoding.c:1766-1830) with csi buffers, as srslte_pdsch_decode calls it; h[port][antenna]"""
    R.srslte_predecoding_type.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    hp = ((C.c_void_p * 4) * 4)()
    for i in range(2):
        for j in range(2):
            hp[i][j] = h[i][j].ctypes.data
    out, csi = [cbuf(), cbuf()], [aligned(n + 16, np.float32)[:n], aligned(n + 16, np.float32)[:n]]
    op, rp = (C.c_void_p * 4)(out[0].ctypes.data, out[1].ctypes.data, 0, 0), (C.c_void_p * 4)(y[0].ctypes.data, y[1].ctypes.data, 0, 0)
    cp = (C.c_void_p * 2)(csi[0].ctypes.data, csi[1].ctypes.data)
    assert R.srslte_predecoding_type(rp, hp, op, cp, 2, 2, nof_layers, codebook_idx, n, scheme, scaling, noise) == 0
    return [o.view(np.complex64).copy() for o in out[:nof_layers]], [c.copy() for c in csi[:nof_layers]]


def _orc_predecode(scheme, y, h, n, nof_layers, codebook_idx, scaling, noise):
    ys = [np.ascontiguousarray(v.view(np.complex64)) for v in y]
    hs = [np.ascontiguousarray(h[i][j].view(np.complex64)) for i in range(2) for j in range(2)]  # [port * 2 + antenna]
    yp, hp = (C.c_void_p * 2)(*[v.ctypes.data for v in ys]), (C.c_void_p * 4)(*[v.ctypes.data for v in hs])
    x, csi = [np.zeros(n, np.complex64) for _ in range(2)], [np.zeros(n, np.float32) for _ in range(2)]
    if scheme == 3:
        orc.orc_predecoding_cdd_2x2.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_float, C.c_float]
        orc.orc_predecoding_cdd_2x2(yp, hp, p(x[0]), p(x[1]), p(csi[0]), p(csi[1]), n, scaling, noise)
    elif nof_layers == 2:
        orc.orc_predecoding_mux_2x2.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_float, C.c_float]
        assert orc.orc_predecoding_mux_2x2(yp, hp, p(x[0]), p(x[1]), p(csi[0]), p(csi[1]), codebook_idx, n, scaling, noise) == 0
    else:
        orc.orc_predecoding_mux_2x1.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float]
        assert orc.orc_predecoding_mux_2x1(yp, hp, p(x[0]), p(csi[0]), codebook_idx, n, scaling) == 0
    return x[:nof_layers], csi[:nof_layers]


MIMO_MODES = [(3, 2, 0), (2, 2, 0), (2, 2, 1), (2, 2, 2), (2, 1, 0), (2, 1, 1), (2, 1, 2), (2, 1, 3)]  # (tx scheme, layers, codebook index)


@pytest.mark.parametrize("scheme,nl,cb", MIMO_MODES)
def test_mimo_precoding_vs_reference(scheme, nl, cb):
    """orc_precoding_cdd2 / orc_precoding_mux2 vs the reference's srslte_precoding_type (precoding.c:1897-2148), SIMD bodies and scalar tails
    (length 206; the AVX large-delay-CDD precoder has no tail loop and leaves the last nof_symbols % 4 outputs unwritten, :1897-1916: 204
    there): the transmit side of the two-layer stimulus."""
    n = 204 if scheme == 3 else 206
    cbuf, rc = _mimo_bufs(n, 10 * scheme + cb)
    x = [cbuf(rc()), cbuf(rc())]
    R = ref()
    R.srslte_precoding_type.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    y = [cbuf(), cbuf()]
    xp, yp = (C.c_void_p * 4)(x[0].ctypes.data, x[1].ctypes.data, 0, 0), (C.c_void_p * 4)(y[0].ctypes.data, y[1].ctypes.data, 0, 0)
    assert R.srslte_precoding_type(xp, yp, nl, 2, cb, n, 0.8, scheme) >= 0
    xs = [np.ascontiguousarray(v.view(np.complex64)) for v in x]
    o = [np.zeros(n, np.complex64), np.zeros(n, np.complex64)]
    if scheme == 3:
        orc.orc_precoding_cdd2.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_float]
        orc.orc_precoding_cdd2(p(xs[0]), p(xs[1]), p(o[0]), p(o[1]), n, 0.8)
    else:
        orc.orc_precoding_mux2.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_float]
        assert orc.orc_precoding_mux2(p(xs[0]), p(xs[1]), p(o[0]), p(o[1]), nl, cb, n, 0.8) == 0
    for k in range(2):
        assert np.abs(o[k] - y[k].view(np.complex64)).max() < 1e-6


@pytest.mark.parametrize("scheme,nl,cb", MIMO_MODES)
def test_mimo_predecoding_scalar_path_vs_reference(scheme, nl, cb):
    """The pre-decoders' scalar code (what the reference runs on the last < 8 symbols of a subframe, and everywhere on a build without
    SIMD): exact divisions on both sides, so the oracle must agree to float rounding. Six symbols per call keep the reference out of
    its SIMD bodies; the CDD alternation restarts with every call, as in the reference."""
    n = 6
    for seed in range(6):
        cbuf, rc = _mimo_bufs(n, 100 * scheme + 10 * cb + seed)
        y, h = [cbuf(rc()), cbuf(rc())], [[cbuf(rc()), cbuf(rc())], [cbuf(rc()), cbuf(rc())]]
        noise, scaling = (0.0, 0.03, 0.4)[seed % 3], (1.0, 1.4142135)[seed % 2]
        xr, cr = _ref_predecode(ref(), scheme, y, h, n, nl, cb, scaling, noise, cbuf)
        xo, co = _orc_predecode(scheme, y, h, n, nl, cb, scaling, noise)
        for k in range(nl):
            assert np.abs(xo[k] - xr[k]).max() <= 2e-5 * max(1.0, np.abs(xr[k]).max()), (seed, k)
            assert np.abs(co[k] / cr[k] - 1).max() < 2e-5, (seed, k)


@pytest.mark.parametrize("scheme,nl,cb", MIMO_MODES)
def test_mimo_predecoding_simd_path_vs_reference(scheme, nl, cb):
    """The SIMD bodies the x86 reference runs on whole subframes multiply by 12-bit reciprocal approximations (_mm256_rcp_ps in
    srslte_simd_cf_rcp / srslte_simd_f_rcp, simd.h:284-300,:959-978; twice in a row for csi): symbols agree with the oracle's exact
    arithmetic to 1e-3 relative, csi likewise. Large-delay CDD on the build that honours signed zeros (see the next test)."""
    from _libs import ref_sz
    n = 512
    cbuf, rc = _mimo_bufs(n, 1000 + 10 * scheme + cb)
    y, h = [cbuf(rc()), cbuf(rc())], [[cbuf(rc()), cbuf(rc())], [cbuf(rc()), cbuf(rc())]]
    R = ref_sz() if scheme == 3 else ref()
    xr, cr = _ref_predecode(R, scheme, y, h, n, nl, cb, 1.0, 0.05, cbuf)
    xo, co = _orc_predecode(scheme, y, h, n, nl, cb, 1.0, 0.05)
    for k in range(nl):
        assert np.abs(xo[k] - xr[k]).max() <= 1e-3 * np.abs(xr[k]).max()
        assert np.abs(co[k] / cr[k] - 1).max() < 1.5e-3


def test_reference_cdd_sign_masks_collapse_with_no_signed_zeros():
    """Fact about the reference build, recorded because it decides how TM3 is pinned. The SIMD large-delay-CDD pre-decoders build the
    effective channel of even / odd symbols with two sign masks, {+0,-0,+0,...} and {-0,+0,-0,...} (precoding.c:727-745,:929-947). The
    reference's flags include -Ofast (CMakeLists.txt:392), i.e. -fno-signed-zeros, and this image's gcc 11.4 then emits ONE constant for
    both: column 2 of the effective channel equals column 1, the matrix is singular, csi comes out as the noise estimate and the
    symbols are wrong (NaN without noise: the test above, and the reference's own `phy_dl_test -t 3`). Compiling that one file with
    -fsigned-zeros (oracle/ref.mk, libsrslte_ref_sz.so) gives what the source - and its scalar tail in BOTH builds - says."""
    from _libs import ref_sz
    n, noise = 64, 0.05
    cbuf, rc = _mimo_bufs(n, 5)
    y, h = [cbuf(rc()), cbuf(rc())], [[cbuf(rc()), cbuf(rc())], [cbuf(rc()), cbuf(rc())]]
    x_fast, csi_fast = _ref_predecode(ref(), 3, y, h, n, 2, 0, 1.0, noise, cbuf)
    x_sz, csi_sz = _ref_predecode(ref_sz(), 3, y, h, n, 2, 0, 1.0, noise, cbuf)
    xo, co = _orc_predecode(3, y, h, n, 2, 0, 1.0, noise)
    assert np.abs(csi_fast[0] / noise - 1).max() < 0.05 and np.abs(csi_fast[1] / noise - 1).max() < 0.05  # rank-1 channel: csi = N0
    assert np.abs(x_fast[0] - xo[0]).max() > 0.5
    # model of the collapsed build: both columns built with the first mask
    Y = np.stack([v.view(np.complex64) for v in y]).astype(np.complex128)
    H = {(i, j): h[i][j].view(np.complex64).astype(np.complex128) for i in range(2) for j in range(2)}
    sg = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    c0, c1 = H[0, 0] + sg * H[1, 0], H[0, 1] + sg * H[1, 1]
    xm = np.zeros((2, n), complex)
    for i in range(n):
        Hm = np.array([[c0[i], c0[i]], [c1[i], c1[i]]])
        xm[:, i] = 2.0 * np.linalg.inv(Hm.conj().T @ Hm + noise * np.eye(2)) @ Hm.conj().T @ Y[:, i]
    assert np.abs(xm[0] - x_fast[0]).max() < 2e-3 * np.abs(xm).max() and np.abs(xm[1] - x_fast[1]).max() < 2e-3 * np.abs(xm).max()
    for k in range(2):
        assert np.abs(x_sz[k] - xo[k]).max() <= 1e-3 * np.abs(xo[k]).max() and np.abs(csi_sz[k] / co[k] - 1).max() < 1.5e-3


TWO_LAYER_CASES = [  # nof_prb, cell_id, mod, tbs, mod2, tbs2, scheme, pmi, cfi, tti, snr
    (25, 7, 2, 4008, 2, 4008, "cdd", 0, 1, 3, 22.0), (25, 7, 2, 4008, 1, 2216, "cdd", 0, 2, 0, 16.0), (6, 1, 1, 328, 1, 328, "cdd", 0, 3, 5, 14.0),
    (50, 150, 3, 21384, 2, 9912, "cdd", 0, 1, 7, 30.0), (100, 2, 2, 22920, 2, 22920, "cdd", 0, 2, 1, 24.0),
    (25, 7, 2, 4008, 3, 4008, "mux", 0, 1, 3, 24.0), (25, 7, 2, 4008, 2, 2216, "mux", 1, 2, 5, 22.0), (15, 33, 1, 1000, 2, 2216, "mux", 0, 1, 0, 20.0),
    (100, 2, 3, 30576, 3, 30576, "mux", 1, 1, 4, 32.0),
    (25, 7, 2, 4008, None, 0, "mux", 0, 1, 3, 14.0), (25, 7, 3, 6200, None, 0, "mux", 1, 2, 5, 18.0), (6, 1, 1, 328, None, 0, "mux", 2, 3, 0, 6.0),
    (50, 150, 2, 9912, None, 0, "mux", 3, 1, 9, 14.0)]
TWO_LAYER_PA = {(25, 7, 2, 4008, 2, 4008, "cdd", 0, 1, 3, 22.0): -3.0, (25, 7, 2, 4008, 2, 2216, "mux", 1, 2, 5, 22.0): 1.0, (25, 7, 3, 6200, None, 0, "mux", 1, 2, 5, 18.0): -4.77}


@pytest.mark.parametrize("prb,cid,mod,tbs,mod2,tbs2,scheme,pmi,cfi,tti,snr", TWO_LAYER_CASES)
def test_pdsch_two_layer_modes_vs_reference(prb, cid, mod, tbs, mod2, tbs2, scheme, pmi, cfi, tti, snr):
    """The reference's own srslte_pdsch_encode / srslte_pdsch_decode (pdsch.c:833-1185) in TM3 (large-delay CDD, 2 transport blocks) and
    TM4 (closed-loop multiplexing, 2 transport blocks with pmi 0/1 or 1 with pmi 0..3) on a 2-port cell with 2 receive antennas vs the
    oracle: transmit grids exact; equalised symbols, csi and LLRs within what the reference's reciprocal approximations allow; transport
    blocks and CRC verdicts identical. CDD on the build that honours signed zeros (previous test), multiplexing on the prescribed one."""
    from _libs import ref_sz
    from lte_sim import DlConfig, RefPdsch, RefPdschTx, make_subframe_mimo, oracle_rx_mimo
    # three of the cases also with srslte_pdsch_cfg_t.power_scale / p_a (pdsch.c:518-554,:852-858): the receiver divides by rho_a
    p_a = TWO_LAYER_PA.get((prb, cid, mod, tbs, mod2, tbs2, scheme, pmi, cfi, tti, snr))
    cfg = DlConfig(prb, cid, mod, tbs, cfi=cfi, nof_rx=2, nof_ports=2, csi=True, tx_scheme=scheme, pmi=pmi, mod2=mod2, tbs2=tbs2, p_a=p_a)
    rng = np.random.default_rng(prb + cid + tti)
    k = {}
    iq, data = make_subframe_mimo(cfg, tti, rng, snr_db=snr, amp=0.4, keep=k)
    grids = RefPdschTx(cfg).run_mimo(data, tti)
    for port in range(2):  # the reference scales by rho_a = sqrt(2) on a 2-port cell (pdsch.c:525), the stimulus by cfg.scaling
        assert np.abs(grids[port][k["idx"]] - np.float32(np.sqrt(2.0) / cfg.scaling) * k["y"][port]).max() < 4e-6
    r = oracle_rx_mimo(cfg, iq, tti, keep=True)
    rr = RefPdsch(cfg, csi_enable=True, lib=ref_sz() if scheme == "cdd" else None).run_mimo(iq, tti)
    assert abs(r["noise"] / rr["noise"] - 1) < 1e-4
    for cw in range(cfg.nof_tb):
        assert np.abs(r["d"][cw] - rr["d"][cw]).max() <= 1.5e-3 * max(1.0, np.abs(rr["d"][cw]).max()), cw
        assert np.abs(r["csi"][cw] / rr["csi"][cw] - 1).max() < 2e-3, cw
        de = np.abs(r["e"][cw].astype(np.int32) - rr["e"][cw].astype(np.int32))
        assert de.max() <= 2 + np.abs(rr["e"][cw]).max() // 400, (cw, int(de.max()))
        assert r["ok"][cw] == rr["ok"][cw] and r["ok"][cw], cw
        assert np.array_equal(r["tb"][cw], rr["tb"][cw]) and np.array_equal(r["tb"][cw][:len(data[cw])], data[cw]), cw


@pytest.mark.parametrize("prb,cid,scheme,pmi,mod,tbs,mod2,tbs2,how,n,cfi,tti,snr", [
    (25, 150, "cdd", 0, 1, 1000, 2, 2216, "random", 9, 3, 4, 18.0), (25, 7, "mux", 1, 1, 328, 1, 504, "centre", 7, 2, 0, 12.0),
    (25, 7, "mux", 2, 2, 2216, None, 0, "slots", 12, 1, 5, 12.0), (50, 3, "cdd", 0, 1, 504, 1, 328, "centre", 6, 1, 0, 10.0),
    (15, 33, "mux", 0, 2, 1000, 1, 504, "centre", 5, 1, 5, 16.0)])
def test_pdsch_two_layer_modes_partial_allocations_vs_reference(prb, cid, scheme, pmi, mod, tbs, mod2, tbs2, how, n, cfi, tti, snr):
    """The two-layer modes on partial allocations - distributed PRBs, different PRBs per slot, the sync region of subframes 0 / 5 with
    the half PRBs of an odd bandwidth - through the reference's srslte_pdsch_encode / srslte_pdsch_decode with grant.prb_idx set: same RE
    order on both sides of the link (srslte_pdsch_cp for a 2-port cell), same transport blocks."""
    from _libs import ref_sz
    from lte_sim import DlConfig, RefPdsch, RefPdschTx, make_subframe_mimo, oracle_rx_mimo
    rng = np.random.default_rng(7 * prb + tti)
    m = np.zeros((2, prb), np.uint8)
    if how == "centre":
        m[:, prb // 2 - 3:prb // 2 - 3 + n] = 1
    elif how == "slots":
        m[0, rng.choice(prb, n, replace=False)] = 1
        m[1, rng.choice(prb, n, replace=False)] = 1
    else:
        m[:, rng.choice(prb, n, replace=False)] = 1
    cfg = DlConfig(prb, cid, mod, tbs, cfi=cfi, nof_rx=2, nof_ports=2, csi=True, tx_scheme=scheme, pmi=pmi, mod2=mod2, tbs2=tbs2, prb_mask=m)
    k = {}
    iq, data = make_subframe_mimo(cfg, tti, rng, snr_db=snr, amp=0.4, keep=k)
    grids = RefPdschTx(cfg).run_mimo(data, tti)
    for port in range(2):
        assert np.abs(grids[port][k["idx"]] - np.float32(np.sqrt(2.0)) * k["y"][port]).max() < 4e-6
        rest = np.ones(len(grids[port]), bool)
        rest[k["idx"]] = False
        assert not grids[port][rest].any()  # nothing outside the oracle's RE list
    r = oracle_rx_mimo(cfg, iq, tti, keep=True)
    rr = RefPdsch(cfg, csi_enable=True, lib=ref_sz() if scheme == "cdd" else None).run_mimo(iq, tti)
    for cw in range(cfg.nof_tb):
        assert np.abs(r["d"][cw] - rr["d"][cw]).max() <= 1.5e-3 * max(1.0, np.abs(rr["d"][cw]).max()), cw
        assert r["ok"][cw] == rr["ok"][cw] and r["ok"][cw], cw
        assert np.array_equal(r["tb"][cw], rr["tb"][cw]) and np.array_equal(r["tb"][cw][:len(data[cw])], data[cw]), cw


# ---------------------------------------------------------------- arbitrary PRB allocations (srslte_pdsch_grant_t.prb_idx[s][n], pdsch.c:81-206)
def _grant_cases():
    """(nof_prb, cell_id, sf_idx, cfi, mcs, how): `how` builds the allocation on the reference's grant."""
    rng = np.random.default_rng(7)
    cases = []
    for P, cid in ((6, 1), (15, 2), (25, 150), (50, 3), (75, 9), (100, 1)):
        nrbg = -(-P // {6: 1, 15: 2, 25: 2, 50: 3, 75: 4, 100: 4}[P])
        for sf in (0, 5, 1):
            cases.append((P, cid, sf, 1 + (sf % 3), 9, ("type0", int(rng.integers(1, 1 << nrbg)))))          # random RBG bitmap
            cases.append((P, cid, sf, 2, 16, ("mask", "random")))                                              # any PRB subset, same in both slots
            cases.append((P, cid, sf, 1, 5, ("mask", "centre")))                                               # only PRBs around the sync signals
        cases.append((P, cid, 0, 1, 20, ("mask", "slots")))                                                    # different PRBs per slot
        if P >= 15:
            cases.append((P, cid, 4, 2, 7, ("type2", 3, int(rng.integers(0, P - 6)), False)))                  # 3 PRB, localized
            cases.append((P, cid, 0, 1, 12, ("type2", min(P // 2, 16), 1, True)))                              # distributed VRBs (slot hopping)
    return cases


@pytest.mark.parametrize("case", _grant_cases(), ids=lambda c: "P%d-id%d-sf%d-cfi%d-mcs%d-%s" % (c[0], c[1], c[2], c[3], c[4], "-".join(str(x) for x in c[5])))
def test_pdsch_arbitrary_allocation_vs_reference(case):
    """The reference's srslte_pdsch_encode / srslte_pdsch_decode on grants with partial, per-slot different and sync-region-only PRB
    allocations (odd cell bandwidths cut PRBs in half there, with upstream's stale-`offset` quirk): the oracle puts the same symbols on
    the same resource elements and decodes the same transport block."""
    import refdrv
    if refdrv.lib() is None:
        pytest.skip("oracle/_ref not built")
    P, cid, sf, cfi, mcs, how = case
    rng = np.random.default_rng(P * 100 + sf * 10 + mcs)
    rx = refdrv.RefDl(P, 1, cid)
    rnti = 0x1234 + sf
    rx.set_rnti(rnti)
    rx.set_chest_cfg(filter_type=0, coef=(4.0, 1.0))
    rx.set_pdsch_cfg(max_iterations=6, mmse=True)
    if how[0] == "type0":
        rx.set_grant(sf, cfi, rnti, mcs, rbg_bitmask=how[1])
    elif how[0] == "type2":
        rx.set_grant_type2(sf, cfi, rnti, mcs, how[1], how[2], distributed=how[3])
    else:
        n = max(1, P // 3)
        rx.set_grant_type2(sf, cfi, rnti, mcs, n, 0)  # TBS / modulation of an n-PRB grant ...
        m = np.zeros((2, P), np.uint8)                # ... on PRBs of our choosing
        if how[1] == "random":
            m[:, rng.choice(P, n, replace=False)] = 1
        elif how[1] == "centre":
            c = [p_ for p_ in range(P // 2 - 3, P // 2 + 3 + P % 2)]
            m[:, rng.choice(c, min(n, len(c)), replace=False)] = 1
        else:
            m[0, rng.choice(P, n, replace=False)] = 1
            m[1, rng.choice(P, n, replace=False)] = 1
        rx.set_prb_masks(m[0], m[1])
    info = rx.grant_info()
    assert info["nof_re"] > 0
    cfg = DlConfig(P, cid, info["mod"], info["tbs"], cfi=cfi, rnti=rnti, prb_mask=info["prb_mask"])
    idx = cfg.indices(sf)
    assert len(idx) == info["nof_re"] and len(idx) * cfg.Qm == info["nof_bits"]
    keep = {}
    iq, data = make_subframe(cfg, sf, rng, snr_db=None, keep=keep)
    want = np.zeros(cfg.grid_len, np.complex64)
    want[keep["idx"]] = keep["y"][0]
    got = rx.encode_pdsch(data)
    assert np.abs(got - want).max() < 1e-6  # same symbols on the same REs, nothing anywhere else
    # and back: the oracle's receiver and the reference's on the oracle's noisy subframe
    iq, data = make_subframe(cfg, sf, rng, snr_db=14.0 if info["mod"] < 3 else 22.0, data=data)
    r = oracle_rx(cfg, iq, sf, keep=True)
    rx.put_grid(r["grid"])
    assert rx.chest() == 0
    crc, _ = rx.decode_pdsch()
    assert bool(crc) == bool(r["ok"])
    if crc:
        assert np.array_equal(rx.payload(info["tbs"] // 8), r["tb"][:info["tbs"] // 8]) and np.array_equal(r["tb"][:info["tbs"] // 8], data)
    rx.free()


# ---------------------------------------------------------------- CQI / PMI report on the PUSCH (uci.c:264-494)
def _ref_pusch_cfg(L_prb, nof_symb, mod, K_segm):
    from _libs import ref_layout
    lay = ref_layout({"srslte_pusch_cfg_t": ["K_segm", "grant.L_prb", "grant.nof_symb", "grant.tb.mod"]}, ["srslte/phy/phch/pusch_cfg.h"])
    buf = (C.c_uint8 * lay["srslte_pusch_cfg_t"])()
    u32 = C.cast(buf, C.POINTER(C.c_uint32))
    u32[lay["srslte_pusch_cfg_t.K_segm"] // 4], u32[lay["srslte_pusch_cfg_t.grant.L_prb"] // 4] = K_segm, L_prb
    u32[lay["srslte_pusch_cfg_t.grant.nof_symb"] // 4], u32[lay["srslte_pusch_cfg_t.grant.tb.mod"] // 4] = nof_symb, mod
    return buf


BETA_CQI = [-1.0, -1.0, 1.125, 1.25, 1.375, 1.625, 1.750, 2.0, 2.25, 2.5, 2.875, 3.125, 3.5, 4.0, 5.0, 6.25]  # sch.c:51-52


def _ref_viterbi(max_bits=512):
    """srslte_viterbi_t for the LTE tail-biting K = 7 rate-1/3 code, as srslte_uci_cqi_init sets it up (uci.c:249-252)."""
    R = ref()
    v = opaque(4096)
    poly = (C.c_int * 3)(0x6D, 0x4F, 0x57)
    R.srslte_viterbi_init.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_bool]
    assert R.srslte_viterbi_init(v, 2, poly, max_bits, True) == 0  # SRSLTE_VITERBI_37
    return v


def test_viterbi37_vs_reference_float_entry():
    """srslte_viterbi_decode_f (viterbi.c:518-548) with the 16-bit AVX2 decoder an x86 build selects: hard decisions bit for bit, clean to
    hopeless inputs, frame lengths of the CQI reports (O + 8)."""
    R, orc = ref(), oracle()
    v = _ref_viterbi()
    rng = np.random.default_rng(37)
    orc.orc_uci_cqi_encode.restype = C.c_int
    for F in (20, 21, 28, 36, 48, 64, 72):
        for sigma in (0.0, 0.4, 0.8, 1.2, 2.0, 5.0):
            for trial in range(4):
                bits = rng.integers(0, 2, F, dtype=np.uint8)
                sr, enc = 0, np.zeros(3 * F, np.uint8)
                for i in range(F - 6, F):
                    sr = (sr << 1) | int(bits[i])
                for i in range(F):
                    sr = ((sr << 1) | int(bits[i])) & 0x7f
                    for j, pl in enumerate((0x6D, 0x4F, 0x57)):
                        enc[3 * i + j] = bin(sr & pl).count("1") & 1
                sym = ((2.0 * enc - 1) * 7.3 + sigma * 7.3 * rng.standard_normal(3 * F)).astype(np.float32)
                a, b = np.zeros(F, np.uint8), np.zeros(F, np.uint8)
                assert R.srslte_viterbi_decode_f(v, p(acopy(sym)), p(a), F) >= 0
                orc.orc_viterbi37_tb_f(p(sym), F, p(b))
                assert np.array_equal(a, b), (F, sigma, trial)
                if sigma <= 0.4:
                    assert np.array_equal(a, bits)
    R.srslte_viterbi_free(v)


def test_reference_viterbi_decode_s_loses_positive_soft_bits():
    """Fact about the reference: srslte_viterbi_decode_s - what decode_cqi_long (uci.c:393) calls - quantises with
    (int16_t)(32767 + in) on an AVX2 build (viterbi.c:571, srslte_vec_quant_sus vector.c:443-453); every positive soft bit overflows and
    comes out as 0 = certain zero. A clean code word of a non-zero message decodes to all zeros; the float entry point decodes it."""
    R = ref()
    v = _ref_viterbi()
    F = 24
    bits = np.array([1, 0, 1, 1, 0, 0, 1, 0] * 3, np.uint8)
    sr, enc = 0, np.zeros(3 * F, np.uint8)
    for i in range(F - 6, F):
        sr = (sr << 1) | int(bits[i])
    for i in range(F):
        sr = ((sr << 1) | int(bits[i])) & 0x7f
        for j, pl in enumerate((0x6D, 0x4F, 0x57)):
            enc[3 * i + j] = bin(sr & pl).count("1") & 1
    soft = (200 * (2 * enc.astype(np.int32) - 1)).astype(np.int16)
    a, b = np.ones(F, np.uint8), np.zeros(F, np.uint8)
    assert R.srslte_viterbi_decode_s(v, p(acopy(soft)), p(a), F) >= 0
    assert R.srslte_viterbi_decode_f(v, p(acopy(soft.astype(np.float32))), p(b), F) >= 0
    assert not a.any() and np.array_equal(b, bits)
    R.srslte_viterbi_free(v)


@pytest.mark.parametrize("O", [1, 2, 4, 5, 7, 10, 11, 12, 13, 20, 28, 40, 56, 64])
def test_uci_cqi_pusch_vs_reference(O):
    """srslte_uci_encode_cqi_pusch / srslte_uci_decode_cqi_pusch: Q' and coded bits for every report size; decoded report for the block
    code (up to 11 bits) at three noise levels, bit for bit. Above 11 bits the reference's decoder is not usable (previous test): the
    oracle's de-rate-matching + Viterbi (pinned above through the float entry point) + CRC must give back what was sent."""
    R, orc = ref(), oracle()
    rng = np.random.default_rng(900 + O)
    q = opaque(1 << 16)
    assert R.srslte_uci_cqi_init(q) == 0
    R.srslte_uci_encode_cqi_pusch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p]
    R.srslte_uci_decode_cqi_pusch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    for L_prb, nsymb, mod, K_segm, I_off, Qp_ri in ((50, 12, 2, 2 * 5824, 7, 0), (100, 12, 3, 13 * 5824, 15, 8), (6, 11, 1, 1056, 9, 3), (25, 12, 2, 4032, 12, 0)):
        cfg, Qm = _ref_pusch_cfg(L_prb, nsymb, mod, K_segm), 2 * mod
        for trial in range(3):
            cqi = rng.integers(0, 2, O, dtype=np.uint8)
            qb_ref = np.zeros(8 * 14 * 12 * L_prb, np.uint8)
            Qp = R.srslte_uci_encode_cqi_pusch(q, cfg, p(cqi), O, BETA_CQI[I_off], Qp_ri, p(qb_ref))
            assert Qp == orc.orc_uci_cqi_qprime(O, I_off, L_prb, nsymb, K_segm, Qp_ri) and Qp > 0
            Q = Qp * Qm
            qb = np.zeros(Q, np.uint8)
            assert orc.orc_uci_cqi_encode(p(cqi), O, p(qb), Q) == 0
            assert np.array_equal(qb, qb_ref[:Q]), (O, L_prb)
            for sigma in (0.3, 1.0, 2.5):
                llr = np.clip(np.round(40 * ((2.0 * qb - 1) + sigma * rng.standard_normal(Q))), -30000, 30000).astype(np.int16)
                out, ok = np.zeros(64, np.uint8), C.c_uint8(0)
                assert orc.orc_uci_cqi_decode(p(acopy(llr)), Q, O, p(out), C.byref(ok)) == 0
                if O <= 11:
                    out_ref, ack_ref = np.zeros(64, np.uint8), C.c_bool(False)
                    assert R.srslte_uci_decode_cqi_pusch(q, cfg, p(acopy(llr)), BETA_CQI[I_off], Qp_ri, O, p(out_ref), C.byref(ack_ref)) == Qp
                    assert ok.value == 1 and ack_ref.value and np.array_equal(out[:O], out_ref[:O]), (O, L_prb, sigma)
                elif sigma <= 1.0 and Q >= 6 * (O + 8):  # enough redundancy for the noise level: the report comes back and the CRC says so
                    assert ok.value == 1 and np.array_equal(out[:O], cqi), (O, L_prb, sigma)
    R.srslte_uci_cqi_free(q)


@pytest.mark.parametrize("prb,L_prb,n_prb,mod,cqi_N,I_cqi,O_ack,O_ri,short", [(25, 4, 3, 1, 0, 7, 0, 0, False), (50, 4, 10, 1, 0, 12, 1, 0, False), (100, 3, 0, 1, 0, 9, 2, 1, False),
                                                                              (15, 2, 1, 2, 3, 8, 0, 0, True), (50, 6, 0, 2, 9, 10, 1, 1, False), (100, 4, 2, 1, 13, 6, 0, 2, False),
                                                                              (25, 1, 7, 1, 0, 15, 1, 1, False)])
def test_pusch_without_ulsch_data_vs_reference(prb, L_prb, n_prb, mod, cqi_N, I_cqi, O_ack, O_ri, short):
    """A PUSCH that carries a CQI report and no transport block (grant.tb.tbs == 0: srslte_ulsch_encode / _decode skip the UL-SCH,
    sch.c:1062-1065,:1157-1165; Q'_cqi = everything the rank indication leaves, uci.c:266-281; HARQ-ACK and RI sized by the report,
    uci.c:557-564 with beta_harq / beta_cqi, sch.c:943-946,:970-973): the oracle's transmit side puts the reference's bits in the
    reference's places, and on a noisy subframe both receivers return the same ACK, RI and - for the block-coded reports - the same report."""
    from lte_sim import RefUlsch, UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(77 + prb + L_prb + cqi_N)
    cfg = UlConfig(prb, 3, mod, 0, L_prb, n_prb=n_prb, shortened=short)
    I_ack, I_ri = 9, 7
    chain = RefUlsch(cfg, O_ack, I_ack, O_ri, I_ri, cqi_N=cqi_N, I_offset_cqi=I_cqi)
    for trial in range(3):
        wb, diff = int(rng.integers(0, 16)), int(rng.integers(0, 1 << (2 * cqi_N))) if cqi_N else 0
        bits = chain.cqi_bits(wb, diff)
        ack, ri = tuple(rng.integers(0, 2, O_ack)), tuple(rng.integers(0, 2, O_ri))
        keep = {}
        iq, _ = make_ul_subframe(cfg, 4, rng, snr_db=None, keep=keep, ack=ack, I_offset_ack=I_ack, ri=ri, I_offset_ri=I_ri, cqi=bits, I_offset_cqi=I_cqi)
        g_ref, q_ref = chain.encode(np.zeros(0, np.uint8), ack=ack, ri=(ri[0] if O_ri else None), cqi=(wb, diff))
        assert np.array_equal(keep["g"], g_ref[:len(keep["g"])])  # the report's code word fills the stream
        snr = {1: 4.0, 2: 10.0}[mod]
        iq, _ = make_ul_subframe(cfg, 4, rng, snr_db=snr, ack=ack, I_offset_ack=I_ack, ri=ri, I_offset_ri=I_ri, cqi=bits, I_offset_cqi=I_cqi)
        r = oracle_ul_rx(cfg, iq, 4, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri, O_cqi=len(bits), I_offset_cqi=I_cqi)
        d = chain.decode(r["q_before_ack"], cfg.scramble(4))
        assert np.array_equal(d["ack"][:O_ack], r["ack"][:O_ack]) and np.array_equal(r["ack"][:O_ack], ack)
        if O_ri:
            assert d["ri"] == r["ri"][0] == ri[0]
        assert np.array_equal(r["cqi"], bits) and r["cqi_ok"] and not r["ok"]
        if cqi_N == 0:
            assert d["cqi"] == (wb, 0) and d["cqi_crc"]


@pytest.mark.parametrize("prb,L_prb,n_prb,mod,tbs,cqi_N,I_cqi,O_ack,O_ri,short", [(25, 25, 0, 2, 4008, 0, 7, 0, 0, False), (50, 40, 5, 1, 2792, 0, 12, 1, 0, False),
                                                                                  (100, 100, 0, 3, 61664, 0, 9, 2, 2, False), (15, 12, 1, 2, 1800, 3, 8, 0, 0, True),
                                                                                  (50, 50, 0, 2, 11448, 9, 10, 1, 1, False), (100, 96, 2, 3, 43816, 13, 6, 0, 2, False)])
def test_ulsch_with_cqi_vs_reference(prb, L_prb, n_prb, mod, tbs, cqi_N, I_cqi, O_ack, O_ri, short):
    """CQI on the PUSCH through the reference's own srslte_ulsch_encode / srslte_ulsch_decode (sch.c:991-1240): the oracle's transmit side
    puts the same bits in the same places (coded report in front of the UL-SCH, Q' from the offset index, UL-SCH rate-matched to the rest,
    next to HARQ-ACK and RI), and on the oracle's noisy subframe both receivers return the same transport block and - for the block-coded
    reports (wide band, 4 bits) - the same report. Sub-band reports (22 / 30 bits here, convolutional code): transmit side and UL-SCH side
    against the reference, the report itself against what was sent (the reference's decoder for them does not work, see above)."""
    from lte_sim import RefUlsch, UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(prb + tbs + cqi_N)
    cfg = UlConfig(prb, 3, mod, tbs, L_prb, n_prb=n_prb, shortened=short)
    I_ack, I_ri = 9, 7
    chain = RefUlsch(cfg, O_ack, I_ack, O_ri, I_ri, cqi_N=cqi_N, I_offset_cqi=I_cqi)
    wb, diff = int(rng.integers(0, 16)), int(rng.integers(0, 1 << (2 * cqi_N))) if cqi_N else 0
    bits = chain.cqi_bits(wb, diff)
    ack, ri = tuple(rng.integers(0, 2, O_ack)), tuple(rng.integers(0, 2, O_ri))
    data = rng.integers(0, 256, tbs // 8, dtype=np.uint8)
    keep = {}
    iq, _ = make_ul_subframe(cfg, 4, rng, snr_db=None, data=data, keep=keep, ack=ack, I_offset_ack=I_ack, ri=ri, I_offset_ri=I_ri, cqi=bits, I_offset_cqi=I_cqi)
    g_ref, q_ref = chain.encode(data, ack=ack, ri=(ri[0] if O_ri else None), cqi=(wb, diff))
    assert np.array_equal(keep["g"], g_ref[:len(keep["g"])])  # CQI code word + UL-SCH bits, before the interleaver
    snr = {1: 6.0, 2: 12.0, 3: 18.0}[mod]
    iq, _ = make_ul_subframe(cfg, 4, rng, snr_db=snr, data=data, ack=ack, I_offset_ack=I_ack, ri=ri, I_offset_ri=I_ri, cqi=bits, I_offset_cqi=I_cqi)
    r = oracle_ul_rx(cfg, iq, 4, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri, O_cqi=len(bits), I_offset_cqi=I_cqi)
    d = chain.decode(r["q_before_ack"], cfg.scramble(4))
    assert d["ok"] == r["ok"] and (not r["ok"] or np.array_equal(d["tb"], r["tb"]))
    assert r["ok"] and np.array_equal(r["tb"][:tbs // 8], data)
    assert np.array_equal(r["cqi"], bits) and r["cqi_ok"]
    if cqi_N == 0:
        assert d["cqi"] == (wb, 0) and d["cqi_crc"]


@pytest.mark.parametrize("prb,L,mod,tbs,snr,short,O_ack,O_ri", [(25, 10, 2, 4008, 4.5, False, 0, 0), (100, 48, 3, 30576, 11.5, False, 2, 1), (6, 6, 1, 1000, 0.5, True, 1, 0),
                                                                (100, 100, 2, 43816, 7.5, False, 0, 1), (50, 45, 3, 30576, 12.5, True, 0, 0)])
def test_ulsch_harq_vs_reference(prb, L, mod, tbs, snr, short, O_ack, O_ri):
    """HARQ on the uplink: the reference's srslte_ulsch_encode with grant.tb.rv = 0, 2, 3, 1 and its srslte_ulsch_decode into ONE
    srslte_softbuffer_rx_t across the four transmissions (sch.c:1063 -> decode_tb -> decode_tb_cb :299-414), against the oracle's UL chain
    with an OrcHarq: coded bits of every redundancy version, and per transmission the de-interleaved LLRs, the CRC verdict and the bytes.
    SNRs are set so that the first transmission fails and a later one succeeds for at least one of the transport blocks."""
    from lte_sim import OrcHarq, RefUlsch, UlConfig, make_ul_subframe, oracle_ul_rx, ul_ri_layout
    rng = np.random.default_rng(2600 + prb + L + O_ack + O_ri)
    cfg = UlConfig(prb, 11, mod, tbs, L, (prb - L) // 2, n_dmrs=3, shortened=short)
    I_ack, I_ri = 9, 8
    chain = RefUlsch(cfg, O_ack, I_ack, O_ri, I_ri)
    Qp_ri, lut, ri_mask, G = ul_ri_layout(cfg, O_ri, I_ri)
    first_fail_later_ok = 0
    for trial in range(3):
        harq, data, oks = OrcHarq(cfg), None, []
        for n, (rv, t) in enumerate(((0, 2), (2, 10), (3, 18), (1, 26))):
            ack, ri = (1, 0)[:O_ack], (1, 0)[:O_ri]
            k = {}
            iq, data = make_ul_subframe(cfg, t, rng, snr_db=snr, amp=0.1, keep=k, data=data, ack=ack, I_offset_ack=I_ack, ri=ri, I_offset_ri=I_ri, rv=rv)
            g_r, _ = chain.encode(data, ack, ri[0] if O_ri else None, rv=rv)
            assert np.array_equal(g_r[:G], k["g"]), (trial, rv)
            o = oracle_ul_rx(cfg, iq, t, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri, harq=harq, rv=rv, new_data=n == 0)
            r = chain.decode(o["q_before_ack"], cfg.scramble(t % 10), rv=rv, new_data=n == 0)
            assert np.array_equal(r["g"][:G], o["g"]) and r["ok"] == o["ok"], (trial, n, r["ok"], o["ok"])
            if r["ok"]:
                assert np.array_equal(r["tb"], o["tb"]) and np.array_equal(r["tb"][:tbs // 8], data)
            oks.append(r["ok"])
            if r["ok"]:
                break
        first_fail_later_ok += (not oks[0]) and oks[-1]
    assert first_fail_later_ok > 0


@pytest.mark.parametrize("prb,npt,spans0,spans1,mod,tbs", [(25, 1, [(0, 8)], None, 2, 2216), (25, 2, [(8, 17)], None, 1, 1000), (25, 1, [(0, 2), (10, 14), (20, 23)], None, 1, 776),
                                                          (25, 2, [(2, 10)], [(14, 20)], 2, 2216), (100, 1, [(0, 4), (40, 60), (90, 100)], None, 2, 9144),
                                                          (100, 2, [(30, 70)], None, 3, 22152), (25, 4, [(4, 25)], None, 3, 4008), (15, 1, [(5, 11)], None, 2, 1000)])
def test_pdsch_encode_with_prb_masks_vs_stimulus_generator(prb, npt, spans0, spans1, mod, tbs):
    """The reference's srslte_pdsch_encode with partial allocations (srslte_pdsch_grant_t.prb_idx of either slot: contiguous, scattered, different
    in the two slots, across the PSS / SSS / PBCH region of subframes 0 and 5) against the oracle's transmit chain with the same masks: the
    per-port grids. This is what tests/test_gpu_dl_tx_grants.py compares the device's per-PDSCH transmit mode with."""
    from lte_sim import RefPdschTx
    rng = np.random.default_rng(2900 + prb + npt + mod)
    mask = np.zeros((2, prb), np.uint8)
    for s, spans in enumerate((spans0, spans0 if spans1 is None else spans1)):
        for a, b in spans:
            mask[s, a:b] = 1
    cfg = DlConfig(prb, 7, mod, tbs, nof_ports=npt, prb_mask=mask)
    chain = RefPdschTx(cfg)
    for t, rv in ((0, 0), (5, 2), (3, 1), (8, 0)):
        if len(cfg.indices(t % 10)) % npt:
            continue
        k = {}
        _, data = make_subframe(cfg, t, rng, rv=rv, keep=k)
        grids = chain.run(data, t, rv=rv)
        for port in range(npt):
            exp = np.zeros(cfg.grid_len, np.complex64)
            exp[k["idx"]] = k["y"][port] * (np.sqrt(2.0) if npt > 1 else 1.0)
            assert np.abs(grids[port] - exp).max() <= 1e-6, (t, rv, port)
