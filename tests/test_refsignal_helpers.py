"""The helper prototypes of refsignal_dl.h / chest_common.h and the 25.212 interleaver generator, as exported by libsrslte_phy_hip.so,
against the reference's compiled refsignal_dl.c / chest_common.c / tc_interl_umts.c (oracle/_ref). The index rules, init-time tables
and host-grid put / get are host code in the product too and need no GPU; the two array helpers run on the device (gpu-marked)."""
import ctypes as C

import numpy as np
import pytest

from _libs import RefCell, RefDlSfCfg, aligned, hip, p, ref


class RefSignal(C.Structure):
    """srslte_refsignal_t (refsignal_dl.h:49-54)."""
    _fields_ = [("cell", RefCell), ("pilots", (C.c_void_p * 10) * 2), ("type", C.c_int), ("mbsfn_area_id", C.c_uint16)]


def _both():
    r = ref()
    if r is None:
        pytest.skip("oracle/_ref is not built")
    h = hip()
    for lib in (r, h):
        lib.srslte_refsignal_cs_fidx.argtypes = [RefCell, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.srslte_refsignal_cs_set_cell.argtypes = [C.c_void_p, RefCell]
        lib.srslte_refsignal_mbsfn_set_cell.argtypes = [C.c_void_p, RefCell, C.c_uint16]
        lib.srslte_refsignal_mbsfn_get_sf.argtypes = [RefCell, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.srslte_refsignal_mbsfn_put_sf.argtypes = [RefCell, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.srslte_chest_set_smooth_filter3_coeff.argtypes = [C.c_void_p, C.c_float]
        lib.srslte_chest_set_smooth_filter_gauss.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        lib.srslte_chest_estimate_noise_pilots.restype = C.c_float
    return r, h


def _cell(prb, ports, cid, cp=0, tdd=False):
    return RefCell(prb, ports, cid, cp, 0, 0, 1 if tdd else 0)


def _table(q, grp, sf, n):
    return np.ctypeslib.as_array(C.cast(q.pilots[grp][sf], C.POINTER(C.c_float)), (2 * n,)).view(np.complex64).copy()


def test_index_rules_match_reference():
    r, h = _both()
    for port in range(5):
        for idx in range(6):
            assert h.srslte_refsignal_cs_v(port, idx) == r.srslte_refsignal_cs_v(port, idx)
            for cp in (0, 1):
                assert h.srslte_refsignal_cs_nsymbol(idx, cp, port) == r.srslte_refsignal_cs_nsymbol(idx, cp, port)
            for cid in (0, 1, 5, 150, 503):
                for m in (0, 7, 199):
                    assert h.srslte_refsignal_cs_fidx(_cell(50, 2, cid), idx, port, m) == r.srslte_refsignal_cs_fidx(_cell(50, 2, cid), idx, port, m)
    for l in range(5):
        assert h.srslte_refsignal_mbsfn_nsymbol(l) == r.srslte_refsignal_mbsfn_nsymbol(l)
        assert h.srslte_refsignal_mbsfn_fidx(l) == r.srslte_refsignal_mbsfn_fidx(l)
    assert h.srslte_refsignal_mbsfn_nof_symbols() == r.srslte_refsignal_mbsfn_nof_symbols() == 3


def test_nof_symbols_fdd_and_every_tdd_configuration():
    """srslte_refsignal_cs_nof_symbols / _nof_re over frame type, CP, the 7 + 1 uplink-downlink and 10 + 1 special-subframe configurations,
    every subframe index and port - including the out-of-table configurations the reference maps to 'downlink' / 'no DwPTS'."""
    r, h = _both()
    for tdd in (False, True):
        for cp in (0, 1):
            q = RefSignal()
            q.cell = _cell(25, 4, 3, cp, tdd)
            for configured in (False, True):
                for sfc in range(8):
                    for ssc in range(11):
                        for tti in range(10):
                            sf = RefDlSfCfg()
                            sf.tdd_config.sf_config, sf.tdd_config.ss_config, sf.tdd_config.configured, sf.tti = sfc, ssc, configured, 30 + tti
                            for port in range(4):
                                a = h.srslte_refsignal_cs_nof_symbols(C.byref(q), C.byref(sf), port)
                                assert a == r.srslte_refsignal_cs_nof_symbols(C.byref(q), C.byref(sf), port), (tdd, cp, sfc, ssc, tti, port)
                                assert h.srslte_refsignal_cs_nof_re(C.byref(q), C.byref(sf), port) == a * 50
    assert h.srslte_refsignal_cs_nof_symbols(None, None, 0) == r.srslte_refsignal_cs_nof_symbols(None, None, 0) == 4
    assert h.srslte_refsignal_cs_nof_symbols(None, None, 3) == 2


@pytest.mark.parametrize("prb,ports,cid,cp", [(6, 1, 0, 0), (25, 2, 150, 0), (100, 4, 503, 0), (50, 4, 77, 1), (15, 2, 301, 1)])
def test_cs_tables_put_get_match_reference(prb, ports, cid, cp):
    """srslte_refsignal_cs_init / _set_cell: all 2 x 10 tables equal the reference's (exactly: +-1/sqrt(2) from the same Gold sequence);
    _put_sf / _get_sf of every port and subframe move the same values between the same positions."""
    r, h = _both()
    rng = np.random.default_rng(prb + cid)
    qs = []
    for lib in (r, h):
        q = RefSignal()
        assert lib.srslte_refsignal_cs_init(C.byref(q), 110) == 0
        assert lib.srslte_refsignal_cs_set_cell(C.byref(q), _cell(prb, ports, cid, cp)) == 0
        qs.append(q)
    for sfi in range(10):
        assert np.array_equal(_table(qs[0], 0, sfi, 8 * prb), _table(qs[1], 0, sfi, 8 * prb))
        assert np.array_equal(_table(qs[0], 1, sfi, 4 * prb), _table(qs[1], 1, sfi, 4 * prb))
    nsym = 14 if cp == 0 else 12
    for tti in (0, 3, 19):
        sf = RefDlSfCfg()
        sf.tti = tti
        for port in range(4):
            grids, gets = [], []
            for lib, q in zip((r, h), qs):
                g = aligned(nsym * 12 * prb, np.complex64)
                g[:] = 0
                assert lib.srslte_refsignal_cs_put_sf(C.byref(q), C.byref(sf), port, p(g)) == 0
                grids.append(g.copy())
                src = (rng.standard_normal(nsym * 12 * prb) + 1j * rng.standard_normal(nsym * 12 * prb)).astype(np.complex64) if lib is r else src
                out = aligned(8 * prb, np.complex64)
                out[:] = 0
                gin = aligned(src.size, np.complex64)
                gin[:] = src
                assert lib.srslte_refsignal_cs_get_sf(C.byref(q), C.byref(sf), port, p(gin), p(out)) == 0
                gets.append(out.copy())
            assert np.array_equal(grids[0], grids[1]) and np.count_nonzero(grids[1]) == (8 if port < 2 else 4) * prb
            assert np.array_equal(gets[0], gets[1])
    # same cell id again: nothing is rebuilt, even with another width (refsignal_dl.c:77)
    for lib, q in zip((r, h), qs):
        assert lib.srslte_refsignal_cs_set_cell(C.byref(q), _cell(6, ports, cid, cp)) == 0
        assert q.cell.nof_prb == prb
    assert h.srslte_refsignal_cs_set_cell(C.byref(qs[1]), _cell(5, 1, 0)) == r.srslte_refsignal_cs_set_cell(C.byref(qs[0]), _cell(5, 1, 0)) == -2
    assert h.srslte_refsignal_cs_set_cell(None, _cell(6, 1, 0)) == -2 and h.srslte_refsignal_cs_put_sf(None, None, 0, None) == -2
    assert h.srslte_refsignal_cs_put_sf(C.byref(qs[1]), C.byref(sf), 4, p(grids[1])) == -2
    for lib, q in zip((r, h), qs):
        lib.srslte_refsignal_free(C.byref(q))
        assert q.cell.nof_prb == 0 and not q.pilots[0][0] and not q.pilots[1][9]


@pytest.mark.parametrize("prb,cid,area", [(6, 1, 0), (50, 20, 1), (100, 503, 255), (25, 7, 78)])
def test_mbsfn_tables_put_get_match_reference(prb, cid, area):
    r, h = _both()
    rng = np.random.default_rng(prb + area)
    qs = []
    for lib in (r, h):
        q = RefSignal()
        assert lib.srslte_refsignal_mbsfn_init(C.byref(q), 110) == 0 and q.type == 1
        assert lib.srslte_refsignal_mbsfn_set_cell(C.byref(q), _cell(prb, 1, cid, 1), area) == 0
        assert q.mbsfn_area_id == area
        qs.append(q)
    for grp in range(2):
        for sfi in range(10):
            assert np.array_equal(_table(qs[0], grp, sfi, 18 * prb), _table(qs[1], grp, sfi, 18 * prb))
    cell = _cell(prb, 1, cid, 1)
    cs = (rng.standard_normal(2 * prb) + 1j * rng.standard_normal(2 * prb)).astype(np.complex64)
    mb = (rng.standard_normal(18 * prb) + 1j * rng.standard_normal(18 * prb)).astype(np.complex64)
    src = (rng.standard_normal(12 * 12 * prb) + 1j * rng.standard_normal(12 * 12 * prb)).astype(np.complex64)
    for port in (0, 1, 4):
        grids, gets = [], []
        for lib in (r, h):
            g = aligned(12 * 12 * prb, np.complex64)
            g[:] = 0
            a, b = aligned(cs.size, np.complex64), aligned(mb.size, np.complex64)
            a[:], b[:] = cs, mb
            assert lib.srslte_refsignal_mbsfn_put_sf(cell, port, p(a), p(b), p(g)) == 0
            grids.append(g.copy())
            gin, out = aligned(src.size, np.complex64), aligned(20 * prb, np.complex64)
            gin[:], out[:] = src, 0
            assert lib.srslte_refsignal_mbsfn_get_sf(cell, port, p(gin), p(out)) == 0
            gets.append(out.copy())
        assert np.array_equal(grids[0], grids[1]) and np.count_nonzero(grids[1]) == 20 * prb
        assert np.array_equal(gets[0], gets[1])
    assert h.srslte_refsignal_mbsfn_put_sf(cell, 5, p(cs), p(mb), p(src)) == r.srslte_refsignal_mbsfn_put_sf(cell, 5, p(cs), p(mb), p(src)) == -2
    assert h.srslte_refsignal_mbsfn_get_sf(_cell(5, 1, 0, 1), 0, p(src), p(mb)) == -2
    for lib, q in zip((r, h), qs):
        lib.srslte_refsignal_free(C.byref(q))


def test_filter_taps_match_reference():
    r, h = _both()
    for n in (1, 3, 5, 7, 9, 15):
        a, b = np.zeros(n + 1, np.float32), np.zeros(n + 1, np.float32)
        assert r.srslte_chest_set_triangle_filter(p(a), n) == h.srslte_chest_set_triangle_filter(p(b), n) == n
        assert np.allclose(a[:n], b[:n], rtol=1e-6, atol=0) and b[n] == 0 and abs(b[:n].sum() - 1) < 1e-6
    b = np.full(5, 7, np.float32)  # an even length: the in-bounds taps as upstream, nothing written behind them
    a = np.zeros(6, np.float32)
    r.srslte_chest_set_triangle_filter(p(a), 4)
    assert h.srslte_chest_set_triangle_filter(p(b), 4) == 4 and np.allclose(a[:4], b[:4], rtol=1e-6) and b[4] == 7
    for w in (0.0, 0.1, 0.25, 1 / 3):
        a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
        assert r.srslte_chest_set_smooth_filter3_coeff(p(a), w) == h.srslte_chest_set_smooth_filter3_coeff(p(b), w) == 3
        assert np.array_equal(a, b)
    for order in (0, 2, 4, 8, 12, 15):
        for sd in (0.5, 1.0, 2.5, 10.0):
            a, b = np.zeros(order + 1, np.float32), np.zeros(order + 1, np.float32)
            assert r.srslte_chest_set_smooth_filter_gauss(p(a), order, sd) == h.srslte_chest_set_smooth_filter_gauss(p(b), order, sd) == order + 1
            assert np.allclose(a, b, rtol=2e-6, atol=1e-12), (order, sd)


def test_umts_interleaver_every_block_size_matches_reference():
    """srslte_tc_interl_UMTS_gen for every 25.212 block size 40 .. 5114: forward and reverse tables equal the reference's (the reference
    is the specification here, including where it departs from 25.212's text), and they are permutations inverse to each other."""
    r, h = _both()

    class Interl(C.Structure):
        _fields_ = [("forward", C.c_void_p), ("reverse", C.c_void_p), ("max_long_cb", C.c_uint32)]
    qa, qb = Interl(), Interl()
    assert r.srslte_tc_interl_init(C.byref(qa), 5114) == 0 and h.srslte_tc_interl_init(C.byref(qb), 5114) == 0
    tab = lambda q, f, K: np.ctypeslib.as_array(C.cast(getattr(q, f), C.POINTER(C.c_uint16)), (K,))
    for K in range(40, 5115):
        assert r.srslte_tc_interl_UMTS_gen(C.byref(qa), K) == 0 and h.srslte_tc_interl_UMTS_gen(C.byref(qb), K) == 0
        fa, fb = tab(qa, "forward", K), tab(qb, "forward", K)
        assert np.array_equal(fa, fb), K
        assert np.array_equal(tab(qa, "reverse", K), tab(qb, "reverse", K)), K
        if K % 97 == 0:
            assert np.array_equal(np.sort(fb), np.arange(K)) and np.array_equal(tab(qb, "reverse", K)[fb], np.arange(K))
    qb.max_long_cb = 100
    assert h.srslte_tc_interl_UMTS_gen(C.byref(qb), 101) == -1
    qb.max_long_cb = 6144
    assert h.srslte_tc_interl_UMTS_gen(C.byref(qb), 39) == -1 and h.srslte_tc_interl_UMTS_gen(C.byref(qb), 5115) == -1
    qb.max_long_cb = 5114
    r.srslte_tc_interl_free(C.byref(qa))
    h.srslte_tc_interl_free(C.byref(qb))


@pytest.mark.gpu
@pytest.mark.parametrize("nof_ref,nsym,flen", [(200, 4, 5), (12, 2, 3), (800, 1, 9), (600, 3, 13), (3, 1, 3), (24, 2, 1)])
def test_average_pilots_on_device_vs_reference(nof_ref, nsym, flen):
    r, h = _both()
    rng = np.random.default_rng(nof_ref + flen)
    x = aligned(nof_ref * nsym, np.complex64)
    x[:] = (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)).astype(np.complex64)
    filt = aligned(flen + 8, np.float32)
    h.srslte_chest_set_triangle_filter(p(filt), flen)
    a, b = aligned(x.size, np.complex64), aligned(x.size, np.complex64)
    r.srslte_chest_average_pilots(p(x), p(a), p(filt), nof_ref, nsym, flen)
    h.srslte_chest_average_pilots(p(x), p(b), p(filt), nof_ref, nsym, flen)
    tol = 1e-4 * max(1.0, float(np.sqrt(np.mean(np.abs(a) ** 2))))  # SURVEY §8d float tolerance
    assert np.max(np.abs(a - b)) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 200, 800, 3601])
def test_estimate_noise_pilots_on_device_vs_reference(n):
    r, h = _both()
    rng = np.random.default_rng(n)
    x, y = aligned(n, np.complex64), aligned(n, np.complex64)
    x[:] = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    y[:] = x + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    va, vb = aligned(n, np.complex64), aligned(n, np.complex64)
    pa = r.srslte_chest_estimate_noise_pilots(p(y), p(x), p(va), n)
    pb = h.srslte_chest_estimate_noise_pilots(p(y), p(x), p(vb), n)
    assert np.array_equal(va, vb) and abs(pa - pb) <= 1e-4 * abs(pa)
    assert h.srslte_chest_estimate_noise_pilots(None, p(x), p(vb), n) == 0.0
