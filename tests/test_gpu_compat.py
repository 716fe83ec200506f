"""The reference's own single-call API (include/srslte_hip/srslte_compat.h) served by the HIP library: host pointers in
and out, same names/arguments/error codes. Written the way the reference's unit tests drive these calls
(ofdm_test.c, dft_test.c, turbodecoder_test.c, chest_test_dl.c, soft_demod_test.c), checked against the oracle."""
import ctypes as C

import numpy as np
import pytest

from _libs import (OrcCell, OrcChestCfg, OrcChestRes, OrcOfdm, RefCell, RefChestCfg, RefChestRes, RefDlSfCfg, acopy, aligned, hip, opaque, oracle, p)

pytestmark = pytest.mark.gpu


class DftPlan(C.Structure):
    _fields_ = [("init_size", C.c_int), ("size", C.c_int), ("in_", C.c_void_p), ("out", C.c_void_p), ("p", C.c_void_p), ("is_guru", C.c_bool),
                ("forward", C.c_bool), ("mirror", C.c_bool), ("db", C.c_bool), ("norm", C.c_bool), ("dc", C.c_bool), ("dir", C.c_int), ("mode", C.c_int)]


def close(a, b, tol=1e-4):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), np.sqrt((np.abs(b) ** 2).mean()))


@pytest.mark.parametrize("prb,std", [(6, False), (25, False), (100, False), (25, True), (50, True), (100, True)])
def test_ofdm_objects(prb, std):
    """ofdm_test.c:74-179: tx -> rx round trip through srslte_ofdm_t objects bound to caller buffers. std: after
    srslte_use_standard_symbol_size(true) (phy_common.c:292-345, as rf_uhd_imp.c:457,:473 selects it for some radios) srslte_symbol_sz and
    with it srslte_ofdm_tx_init / rx_init use the power-of-two family: 512 / 1024 / 2048 for 25 / 50 / 100 PRB."""
    L = hip()
    L.srslte_use_standard_symbol_size.argtypes = [C.c_bool]
    L.srslte_use_standard_symbol_size(std)
    try:
        _ofdm_objects(L, prb, std)
    finally:
        L.srslte_use_standard_symbol_size(False)


def _ofdm_objects(L, prb, std):
    rng = np.random.default_rng(prb)
    N = L.srslte_symbol_sz(prb)
    assert N == ({6: 128, 25: 512, 50: 1024, 100: 2048} if std else {6: 128, 25: 384, 100: 1536})[prb]
    nre, sf = 14 * 12 * prb, 15 * N
    grid_in, time_buf, grid_out = aligned(2 * nre, np.float32), aligned(2 * sf, np.float32), aligned(2 * nre, np.float32)
    tx, rx = opaque(4096), opaque(4096)
    assert L.srslte_ofdm_tx_init(tx, 0, p(grid_in), p(time_buf), prb) == 0
    assert L.srslte_ofdm_rx_init(rx, 0, p(time_buf), p(grid_out), prb) == 0
    L.srslte_ofdm_set_normalize(tx, True)
    L.srslte_ofdm_set_normalize(rx, True)
    g = (rng.standard_normal(nre) + 1j * rng.standard_normal(nre)).astype(np.complex64)
    grid_in.view(np.complex64)[:] = g
    L.srslte_ofdm_tx_sf(tx)
    q = OrcOfdm()
    assert oracle().orc_ofdm_init_sz(C.byref(q), prb, N, True) == 0
    q.normalize = True
    ref_t = np.zeros(sf, np.complex64)
    oracle().orc_ofdm_tx_sf(C.byref(q), p(g), p(ref_t))
    assert close(time_buf.view(np.complex64), ref_t)
    L.srslte_ofdm_rx_sf(rx)
    assert np.mean(np.abs(grid_out.view(np.complex64) - g) ** 2) < 1e-9  # ofdm_test.c:155 accepts 0.07
    # srslte_ofdm_init_ takes the symbol size from its caller (ofdm.c:38-57): the other family's size on the same carrier count
    other = {128: 128, 384: 512, 512: 384, 768: 1024, 1024: 768, 1536: 2048, 2048: 1536}[N]
    rx2, t2, g2 = opaque(4096), aligned(2 * 15 * other, np.float32), aligned(2 * nre, np.float32)
    assert L.srslte_ofdm_init_(rx2, 0, p(t2), p(g2), other, prb, 0) == 0
    q2 = OrcOfdm()
    assert oracle().orc_ofdm_init_sz(C.byref(q2), prb, other, True) == 0
    t_in = (rng.standard_normal(15 * other) + 1j * rng.standard_normal(15 * other)).astype(np.complex64)
    t2.view(np.complex64)[:] = t_in
    L.srslte_ofdm_rx_sf(rx2)
    g_ref = np.zeros(nre, np.complex64)
    oracle().orc_ofdm_rx_sf(C.byref(q2), p(t_in), p(g_ref))
    assert close(g2.view(np.complex64), g_ref)
    L.srslte_ofdm_rx_free(rx2)
    assert L.srslte_ofdm_init_(rx2, 0, p(t2), p(g2), 1000, prb, 0) == -1  # not a size of either family
    L.srslte_ofdm_tx_free(tx)
    L.srslte_ofdm_rx_free(rx)
    assert L.srslte_ofdm_rx_init(rx, 0, p(time_buf), p(grid_out), 111) == -1  # ofdm.c:237-240


def test_ofdm_slot_calls():
    """srslte_ofdm_rx_slot/_tx_slot/_rx_slot_ng (ofdm.c:384-422,:488-530) give the same halves as the subframe calls."""
    L, rng, prb = hip(), np.random.default_rng(11), 25
    N = L.srslte_symbol_sz(prb)
    nre, sf = 14 * 12 * prb, 15 * N
    grid_in, time_buf, grid_out = aligned(2 * nre, np.float32), aligned(2 * sf, np.float32), aligned(2 * nre, np.float32)
    tx, rx = opaque(4096), opaque(4096)
    assert L.srslte_ofdm_tx_init(tx, 0, p(grid_in), p(time_buf), prb) == 0 and L.srslte_ofdm_rx_init(rx, 0, p(time_buf), p(grid_out), prb) == 0
    g = (rng.standard_normal(nre) + 1j * rng.standard_normal(nre)).astype(np.complex64)
    grid_in.view(np.complex64)[:] = g
    L.srslte_ofdm_tx_sf(tx)
    whole = time_buf.view(np.complex64).copy()
    time_buf[:] = 0
    L.srslte_ofdm_tx_slot(tx, 1)
    assert np.all(time_buf.view(np.complex64)[: sf // 2] == 0) and np.array_equal(time_buf.view(np.complex64)[sf // 2:], whole[sf // 2:])
    L.srslte_ofdm_tx_slot(tx, 0)
    assert np.array_equal(time_buf.view(np.complex64), whole)
    L.srslte_ofdm_rx_sf(rx)
    full = grid_out.view(np.complex64).copy()
    grid_out[:] = 0
    L.srslte_ofdm_rx_slot(rx, 1)
    assert np.all(grid_out.view(np.complex64)[: nre // 2] == 0) and np.array_equal(grid_out.view(np.complex64)[nre // 2:], full[nre // 2:])
    ng = np.zeros(nre // 2, np.complex64)
    second = np.ascontiguousarray(whole[sf // 2:])
    L.srslte_ofdm_rx_slot_ng(rx, p(second), p(ng))
    assert np.array_equal(ng, full[nre // 2:])
    out2 = np.zeros(nre, np.complex64)
    L.srslte_ofdm_rx_sf_ng(rx, p(whole), p(out2))
    assert np.array_equal(out2, full)
    L.srslte_ofdm_tx_free(tx)
    L.srslte_ofdm_rx_free(rx)


@pytest.mark.parametrize("prb,region", [(6, 1), (25, 2), (100, 2)])
def test_ofdm_mbsfn(prb, region):
    """MBSFN subframe objects (ofdm.c:246-305 init, :424-437 rx slot, :558-574 tx slot, :453-467/:580-594 subframe)."""
    L, rng = hip(), np.random.default_rng(prb + region)
    N = L.srslte_symbol_sz(prb)
    nre, sf = 12 * 12 * prb, 15 * N
    grid_in, time_buf, grid_out = aligned(2 * nre, np.float32), aligned(2 * sf, np.float32), aligned(2 * nre, np.float32)
    tx, rx = opaque(4096), opaque(4096)
    assert L.srslte_ofdm_tx_init_mbsfn(tx, 1, p(grid_in), p(time_buf), prb) == 0
    assert L.srslte_ofdm_rx_init_mbsfn(rx, 1, p(time_buf), p(grid_out), prb) == 0
    for o in (tx, rx):
        L.srslte_ofdm_set_non_mbsfn_region(o, region)
        L.srslte_ofdm_set_normalize(o, True)
    g = (rng.standard_normal(nre) + 1j * rng.standard_normal(nre)).astype(np.complex64)
    grid_in.view(np.complex64)[:] = g
    time_buf.view(np.complex64)[:] = 7 + 7j
    L.srslte_ofdm_tx_sf(tx)
    q = OrcOfdm()
    oracle().orc_ofdm_init(C.byref(q), prb, False)
    q.normalize, q.exact, q.non_mbsfn_region = True, True, region
    ref_t = np.full(sf, 7 + 7j, np.complex64)
    oracle().orc_ofdm_tx_sf(C.byref(q), p(g), p(ref_t))
    t = time_buf.view(np.complex64)
    gap = ref_t == 7 + 7j
    assert gap.sum() > 0 and np.all(t[gap] == 7 + 7j)  # the guard between the regions is left untouched (ofdm.c:570-572)
    assert close(t, ref_t)
    L.srslte_ofdm_rx_sf(rx)
    assert np.mean(np.abs(grid_out.view(np.complex64) - g) ** 2) < 1e-9
    ref_g = np.zeros(nre, np.complex64)
    oracle().orc_ofdm_rx_sf(C.byref(q), p(t.copy()), p(ref_g))
    assert close(grid_out.view(np.complex64), ref_g)
    # slot-level MBSFN calls on caller pointers
    slot0_t, slot0_g = np.full(sf // 2, 7 + 7j, np.complex64), np.zeros(nre // 2, np.complex64)
    L.srslte_ofdm_tx_slot_mbsfn(tx, p(np.ascontiguousarray(g[: nre // 2])), p(slot0_t))
    assert np.array_equal(slot0_t, t[: sf // 2])
    L.srslte_ofdm_rx_slot_mbsfn(rx, p(slot0_t), p(slot0_g))
    assert np.array_equal(slot0_g, grid_out.view(np.complex64)[: nre // 2])
    L.srslte_ofdm_tx_free(tx)
    L.srslte_ofdm_rx_free(rx)


@pytest.mark.parametrize("N", [128, 300, 1536])
def test_dft_plan_r(N):
    """srslte_dft_plan_r / srslte_dft_run_r (dft_fftw.c:209-232,:315-334): FFTW half-complex layout, 1/N norm."""
    L, rng = hip(), np.random.default_rng(N)
    x = rng.standard_normal(N).astype(np.float32)
    fwd, bwd = DftPlan(), DftPlan()
    assert L.srslte_dft_plan(C.byref(fwd), N, 0, 1) == 0 and L.srslte_dft_plan_r(C.byref(bwd), N, 1) == 0
    assert fwd.mode == 1 and bwd.mode == 1
    hc, ref_hc, back = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32)
    L.srslte_dft_run(C.byref(fwd), p(x), p(hc))
    oracle().orc_dft_r2hc(p(x), p(ref_hc), N, 1)
    assert close(hc, ref_hc)
    L.srslte_dft_plan_set_norm(C.byref(bwd), True)
    L.srslte_dft_run_r(C.byref(bwd), p(hc), p(back))
    assert np.abs(back - x).max() < 1e-4 * np.abs(x).max() * 10
    assert L.srslte_dft_replan(C.byref(fwd), N // 2) == 0 and fwd.size == N // 2
    assert L.srslte_dft_replan(C.byref(fwd), 2 * N) == -1
    L.srslte_dft_plan_free(C.byref(fwd))
    L.srslte_dft_plan_free(C.byref(bwd))


def test_tcod_encode_lut_golden():
    """srslte_tcod_encode_lut (turbocoder.c:189-367) incl. its in-place CRC attachment, against reference outputs."""
    import os
    from _libs import make_crc
    L = hip()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tcod_lut.npz"))
    tcod = opaque(64)
    assert L.srslte_tcod_init(tcod, 6144) == 0
    L.srslte_tcod_gentable()
    for n in range(6):
        idx, K, with_cb, last, tb_init, tb_final = (int(v) for v in g["meta_%d" % n])
        buf, par = g["in_%d" % n].copy(), np.zeros(K // 4 + 2, np.uint8)
        crc_tb, crc_cb = make_crc(0x1864CFB, 24), make_crc(0x1800063, 24)
        crc_tb.crcinit = tb_init
        r = L.srslte_tcod_encode_lut(tcod, C.byref(crc_tb), C.byref(crc_cb) if with_cb else None, p(buf), p(par), idx, bool(last))
        assert r == 3 * K + 12
        assert np.array_equal(buf, g["sys_%d" % n]) and np.array_equal(par[: K // 4 + 1], g["par_%d" % n]) and crc_tb.crcinit == tb_final
    assert L.srslte_tcod_encode_lut(tcod, C.byref(crc_tb), None, p(buf), p(par), 188, False) == -1
    L.srslte_tcod_free(tcod)


@pytest.mark.parametrize("N", [128, 1536, 12 * 25])
def test_dft_plan_options(N):
    """dft_test.c:80-130: forward o backward identity for the mirror/dc/norm option combinations + direct check."""
    L, rng = hip(), np.random.default_rng(N)
    x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    for mirror, dc, norm in ((False, False, False), (True, False, True), (True, True, True), (False, False, True)):
        fwd, bwd = DftPlan(), DftPlan()
        assert L.srslte_dft_plan_c(C.byref(fwd), N, 0) == 0 and L.srslte_dft_plan_c(C.byref(bwd), N, 1) == 0
        for pl in (fwd, bwd):
            L.srslte_dft_plan_set_mirror(C.byref(pl), mirror)
            L.srslte_dft_plan_set_dc(C.byref(pl), dc)
            L.srslte_dft_plan_set_norm(C.byref(pl), norm)
        y, z = np.zeros(N, np.complex64), np.zeros(N, np.complex64)
        L.srslte_dft_run_c(C.byref(fwd), p(x), p(y))
        if not mirror:
            ref = np.zeros(N, np.complex64)
            oracle().orc_dft_exact(p(x), p(ref), N, 1)
            assert close(y, ref / (np.sqrt(N) if norm else 1.0))
        L.srslte_dft_run_c(C.byref(bwd), p(y), p(z))
        if dc:
            pass  # the DC bin is dropped on the way: identity does not hold (dft_test.c skips the comparison of bin 0 likewise)
        else:
            scale = 1.0 if norm else float(N)
            assert np.abs(z / scale - x).max() < 1e-4 * np.abs(x).max() * 10
        L.srslte_dft_plan_free(C.byref(fwd))
        L.srslte_dft_plan_free(C.byref(bwd))
    # FFTW plans any length (dft_fftw.c:167-191): so does this library - lengths without a 2/3/5 plan go to the direct-sum kernel
    odd = DftPlan()
    assert L.srslte_dft_plan_c(C.byref(odd), 7 * 12, 0) == 0
    xo, yo, ro = x[:84].copy(), np.zeros(84, np.complex64), np.zeros(84, np.complex64)
    L.srslte_dft_run_c(C.byref(odd), p(xo), p(yo))
    oracle().orc_dft_exact(p(xo), p(ro), 84, 1)
    assert close(yo, ro)
    L.srslte_dft_plan_free(C.byref(odd))
    bad = DftPlan()
    assert L.srslte_dft_plan_c(C.byref(bad), 0, 0) != 0 and L.srslte_dft_plan_c(C.byref(bad), 1 << 20, 0) != 0


@pytest.mark.parametrize("N,M,istride,idist,ostride,odist", [(128, 7, 1, 137, 1, 128), (64, 5, 5, 1, 1, 64), (300, 3, 2, 700, 3, 1), (12, 6, 6, 1, 6, 1)])
def test_dft_guru_plan_layouts(N, M, istride, idist, ostride, odist):
    """srslte_dft_plan_guru_c / srslte_dft_run_guru_c (dft_fftw.c:137-165,:307-313): fftw_plan_many_dft's layout - element j of transform i at
    [i * dist + j * stride] - on caller buffers captured at plan time; elements of the output that the layout does not address stay untouched."""
    L, rng = hip(), np.random.default_rng(N + M)
    nin, nout = (M - 1) * idist + (N - 1) * istride + 1, (M - 1) * odist + (N - 1) * ostride + 1
    x = aligned(nin, np.complex64)
    x[:] = (rng.standard_normal(nin) + 1j * rng.standard_normal(nin)).astype(np.complex64)
    y = aligned(nout, np.complex64)
    y[:] = -7.0
    pl = DftPlan()
    assert L.srslte_dft_plan_guru_c(C.byref(pl), N, 0, p(x), p(y), istride, ostride, M, idist, odist) == 0 and pl.is_guru
    L.srslte_dft_run_guru_c(C.byref(pl))
    touched = np.zeros(nout, bool)
    for i in range(M):
        xi, ref = np.ascontiguousarray(x[i * idist:i * idist + (N - 1) * istride + 1:istride]), np.zeros(N, np.complex64)
        oracle().orc_dft_exact(p(xi), p(ref), N, 1)
        sl = slice(i * odist, i * odist + (N - 1) * ostride + 1, ostride)
        assert close(y[sl], ref), i
        touched[sl] = True
    assert np.all(y[~touched] == -7.0)
    # replan on the same buffers to a shorter transform (dft_fftw.c:93-118), backward this time through the plan's direction
    N2 = N // 2
    assert L.srslte_dft_replan_guru_c(C.byref(pl), N2, p(x), p(y), istride, ostride, M, idist, odist) == 0 and pl.size == N2
    L.srslte_dft_run_guru_c(C.byref(pl))
    xi, ref = np.ascontiguousarray(x[0:(N2 - 1) * istride + 1:istride]), np.zeros(N2, np.complex64)
    oracle().orc_dft_exact(p(xi), p(ref), N2, 1)
    assert close(y[0:(N2 - 1) * ostride + 1:ostride], ref)
    L.srslte_dft_plan_free(C.byref(pl))
    plain = DftPlan()
    assert L.srslte_dft_plan_c(C.byref(plain), N, 0) == 0
    L.srslte_dft_run_guru_c(C.byref(plain))  # "the selected plan is not guru": an error line and no transform
    L.srslte_dft_plan_free(C.byref(plain))
    assert L.srslte_dft_plan_guru_c(C.byref(pl), N, 0, p(x), p(y), 0, 1, M, idist, odist) != 0


def test_dft_precoding_object():
    L, rng = hip(), np.random.default_rng(3)
    q = opaque(1 << 16)
    assert L.srslte_dft_precoding_init_tx(q, 100) == 0
    for nprb in (1, 6, 25, 100):
        x = (rng.standard_normal(12 * 12 * nprb) + 1j * rng.standard_normal(12 * 12 * nprb)).astype(np.complex64)
        y, ref = np.zeros_like(x), np.zeros_like(x)
        assert L.srslte_dft_precoding(q, p(x), p(y), nprb, 12) == 0
        oracle().orc_dft_precoding(p(x), p(ref), nprb, 12, 1, True)
        assert close(y, ref)
    assert L.srslte_dft_precoding(q, p(x), p(y), 7, 12) == -1
    L.srslte_dft_precoding_free(q)


@pytest.mark.parametrize("K", [176, 504, 5824])
def test_tdec_object(K):
    """turbodecoder_test.c:117-311: tcod_encode -> noisy LLRs -> srslte_tdec_run_all, and the per-iteration API."""
    L, rng = hip(), np.random.default_rng(K)
    tcod, tdec = opaque(64), opaque(1 << 16)
    assert L.srslte_tcod_init(tcod, 6144) == 0 and L.srslte_tdec_init(tdec, 6144) == 0
    bits = rng.integers(0, 2, K).astype(np.uint8)
    enc, ref_enc = np.zeros(3 * K + 12, np.uint8), np.zeros(3 * K + 12, np.uint8)
    assert L.srslte_tcod_encode(tcod, p(bits), p(enc), K) == 0
    oracle().orc_tcod_encode_bits(p(bits), p(ref_enc), K)
    assert np.array_equal(enc, ref_enc)
    llr = (100 * ((2.0 * enc - 1) + 0.9 * rng.standard_normal(enc.shape))).astype(np.int16)
    L.srslte_tdec_force_not_sb(tdec)
    out, ref = np.zeros(K // 8, np.uint8), np.zeros(K // 8, np.uint8)
    assert L.srslte_tdec_run_all(tdec, p(llr), p(out), 4, K) == 0
    oracle().orc_tdec_run(p(llr), False, K, 4, p(ref), None)
    assert np.array_equal(out, ref) and L.srslte_tdec_get_nof_iterations(tdec) == 4
    per = np.zeros((6, K // 8), np.uint8)
    oracle().orc_tdec_run(p(llr), False, K, 6, p(ref), p(per))
    assert L.srslte_tdec_new_cb(tdec, K) == 0
    for it in range(6):  # sch.c:353-383 drives it like this
        L.srslte_tdec_iteration(tdec, p(llr), p(out))
        assert np.array_equal(out, per[it]) and L.srslte_tdec_get_nof_iterations(tdec) == it + 1
    assert L.srslte_tdec_new_cb(tdec, 41) == -1 and L.srslte_tdec_new_cb(tdec, 6145) == -1
    L.srslte_tdec_free(tdec)
    L.srslte_tcod_free(tcod)


@pytest.mark.parametrize("K", [504, 1008, 5824])
def test_tdec_object_8bit(K):
    """srslte_tdec_iteration_8bit / srslte_tdec_run_all_8bit (turbodecoder.c:565-593) and the manual 8-bit back-ends
    (SRSLTE_TDEC_SSE8_WINDOW / AVX8_WINDOW, turbodecoder.c:183-186,:209-212) fed through the 16-bit API."""
    L, rng = hip(), np.random.default_rng(K + 8)
    tdec = opaque(1 << 16)
    assert L.srslte_tdec_init(tdec, 6144) == 0
    bits = rng.integers(0, 2, K).astype(np.uint8)
    enc = np.zeros(3 * K + 12, np.uint8)
    oracle().orc_tcod_encode_bits(p(bits), p(enc), K)
    llr = (20 * ((2.0 * enc - 1) + 0.8 * rng.standard_normal(enc.shape))).clip(-128, 127).astype(np.int8)
    assert L.srslte_tdec_autoimp_get_subblocks_8bit(K) == oracle().orc_tdec_autoimp_subblocks_8bit(K)
    L.srslte_tdec_force_not_sb(tdec)
    out, ref, per = np.zeros(K // 8, np.uint8), np.zeros(K // 8, np.uint8), np.zeros((6, K // 8), np.uint8)
    assert L.srslte_tdec_run_all_8bit(tdec, p(llr), p(out), 4, K) == 0
    oracle().orc_tdec_run_8bit(p(llr), False, K, 4, p(ref), None)
    assert np.array_equal(out, ref) and L.srslte_tdec_get_nof_iterations(tdec) == 4
    oracle().orc_tdec_run_8bit(p(llr), False, K, 6, None, p(per))
    assert L.srslte_tdec_new_cb(tdec, K) == 0
    for it in range(6):
        L.srslte_tdec_iteration_8bit(tdec, p(llr), p(out))
        assert np.array_equal(out, per[it])
    L.srslte_tdec_free(tdec)
    if K == 5824:  # manual avx8 back-end behind the 16-bit API: LLRs narrowed with a C cast (convert_16_to_8, :458-463)
        man = opaque(1 << 16)
        assert L.srslte_tdec_init_manual(man, 6144, 7) == 0  # SRSLTE_TDEC_AVX8_WINDOW
        L.srslte_tdec_force_not_sb(man)
        assert L.srslte_tdec_run_all(man, p(llr.astype(np.int16)), p(out), 3, K) == 0
        oracle().orc_tdec_run_8bit(p(llr), False, K, 3, p(ref), None)
        assert np.array_equal(out, ref)
        L.srslte_tdec_free(man)


def test_chest_dl_object():
    """chest_test_dl.c:78-255: init, set_cell, res_init, estimate with the default configuration."""
    L, rng = hip(), np.random.default_rng(4)
    prb, cid = 25, 2
    est, res = opaque(1 << 16), RefChestRes()
    assert L.srslte_chest_dl_init(est, prb, 1) == 0
    assert L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
    assert L.srslte_chest_dl_res_init(C.byref(res), prb) == 0
    n, nre = 14 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    for sf_idx in (0, 4):
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
        k, l = np.arange(n) % nre, np.arange(n) // nre
        grid = acopy((g * ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))) + 0.05 * rng.standard_normal(n)).astype(np.complex64).view(np.float32))
        sf = RefDlSfCfg()
        sf.tti = 10 + sf_idx
        inp = (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)
        assert L.srslte_chest_dl_estimate(est, C.byref(sf), inp, C.byref(res)) == 0
        ref, rres, oc = np.zeros(n, np.complex64), OrcChestRes(), OrcChestCfg()
        assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(oc), p(grid), p(ref), C.byref(rres)) == 0
        ce = np.ctypeslib.as_array(C.cast(res.ce[0][0], C.POINTER(C.c_float)), (2 * n,)).view(np.complex64)
        assert close(ce, ref)
        assert abs(res.noise_estimate - rres.noise_estimate) <= 1e-4 * rres.noise_estimate and abs(res.snr_db - rres.snr_db) < 1e-3
        assert abs(res.rsrp_dbm - rres.rsrp_dbm) < 1e-3 and np.isnan(res.sync_error)
    L.srslte_chest_dl_res_free(C.byref(res))
    L.srslte_chest_dl_free(est)


def test_chest_dl_object_standard_symbol_sizes():
    """srslte_chest_dl_estimate_cfg reads srslte_symbol_sz(cell.nof_prb) at every call for its timing-error figure (chest_dl.c:695): after
    srslte_use_standard_symbol_size(true) the same object on the same grid reports 4/3 of it (512 against 384 points at 25 PRB), as the
    reference's compiled estimator does (tests/test_oracle_vs_ref.py::test_chest_dl_standard_symbol_sizes_vs_ref); estimates and CFO stay."""
    L, rng = hip(), np.random.default_rng(14)
    L.srslte_use_standard_symbol_size.argtypes = [C.c_bool]
    prb, cid, sf_idx = 25, 2, 4
    est, res = opaque(1 << 16), RefChestRes()
    assert L.srslte_chest_dl_init(est, prb, 1) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
    assert L.srslte_chest_dl_res_init(C.byref(res), prb) == 0
    n, nre = 14 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
    k, l = np.arange(n) % nre, np.arange(n) // nre
    grid = acopy((g * ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 90.0 + 0.12 * l))) + 0.05 * rng.standard_normal(n)).astype(np.complex64).view(np.float32))
    got = {}
    try:
        for std in (True, False):
            L.srslte_use_standard_symbol_size(std)
            oracle().orc_use_standard_symbol_size(std)
            sf, rc, oc = RefDlSfCfg(), RefChestCfg(), OrcChestCfg()
            sf.tti = sf_idx
            rc.filter_coef[0], rc.filter_coef[1], oc.filter_coef[0], oc.filter_coef[1] = 4.0, 1.0, 4.0, 1.0
            rc.cfo_estimate_enable = oc.cfo_estimate_enable = True
            rc.sync_error_enable = oc.sync_error_enable = True
            rc.cfo_estimate_sf_mask = 0x3FF
            assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
            ref, rres = np.zeros(n, np.complex64), OrcChestRes()
            assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(oc), p(grid), p(ref), C.byref(rres)) == 0
            ce = np.ctypeslib.as_array(C.cast(res.ce[0][0], C.POINTER(C.c_float)), (2 * n,)).view(np.complex64)
            assert close(ce, ref)
            assert abs(res.cfo - rres.cfo) <= 1e-4 * abs(rres.cfo) + 1e-6 and abs(res.sync_error - rres.sync_error) <= 2e-3 * abs(rres.sync_error) + 1e-4
            got[std] = (res.cfo, res.sync_error)
    finally:
        L.srslte_use_standard_symbol_size(False)
        oracle().orc_use_standard_symbol_size(False)
    assert abs(got[True][1] / got[False][1] - 4.0 / 3.0) < 1e-3 and abs(got[True][0] - got[False][0]) <= 1e-5 * abs(got[False][0])
    L.srslte_chest_dl_res_free(C.byref(res))
    L.srslte_chest_dl_free(est)


def test_chest_dl_object_extended_cp():
    """srslte_chest_dl_* on an extended-CP cell (12-symbol grids, SRSLTE_SF_LEN_RE): default configuration and subframe interpolation."""
    L, rng = hip(), np.random.default_rng(24)
    prb, cid = 25, 3
    est, res = opaque(1 << 16), RefChestRes()
    assert L.srslte_chest_dl_init(est, prb, 1) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, cid, 1, 0, 0, 0)) == 0
    assert L.srslte_chest_dl_res_init(C.byref(res), prb) == 0
    n, nre = 12 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, 1, False)
    for sf_idx, interp in ((0, False), (4, True)):
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
        k, l = np.arange(n) % nre, np.arange(n) // nre
        grid = acopy((g * ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))) + 0.05 * rng.standard_normal(n)).astype(np.complex64).view(np.float32))
        sf, rc, oc = RefDlSfCfg(), RefChestCfg(), OrcChestCfg()
        sf.tti = 10 + sf_idx
        rc.interpolate_subframe = oc.interpolate_subframe = interp
        rc.filter_coef[0], rc.filter_coef[1] = 4.0, 1.0
        oc.filter_coef[0], oc.filter_coef[1] = 4.0, 1.0
        assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
        ref, rres = np.zeros(n, np.complex64), OrcChestRes()
        assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(oc), p(grid), p(ref), C.byref(rres)) == 0
        ce = np.ctypeslib.as_array(C.cast(res.ce[0][0], C.POINTER(C.c_float)), (2 * n,)).view(np.complex64)
        assert close(ce, ref)
        assert abs(res.noise_estimate - rres.noise_estimate) <= 1e-4 * rres.noise_estimate and abs(res.rsrp_dbm - rres.rsrp_dbm) < 1e-3
    L.srslte_chest_dl_res_free(C.byref(res))
    L.srslte_chest_dl_free(est)


@pytest.mark.parametrize("alg", [1, 2])
def test_chest_dl_object_noise_pss_empty(alg):
    """srslte_chest_dl_estimate_cfg with cfg.noise_alg PSS / EMPTY (phy_common.cc:111-118 selects them from snr_estim_alg) over a run of
    subframes on one object, with the automatic Gauss filter fed by the kept estimate, against the oracle's run."""
    L, rng = hip(), np.random.default_rng(40 + alg)
    orc = oracle()
    orc.orc_chest_dl_ports_state.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    prb, cid = 25, 5
    est, res = opaque(1 << 16), RefChestRes()
    assert L.srslte_chest_dl_init(est, prb, 1) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
    assert L.srslte_chest_dl_res_init(C.byref(res), prb) == 0
    n, nre = 14 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    pss = np.zeros(62, np.complex64)
    orc.orc_pss_generate(cid % 3, p(pss))
    k, l = np.arange(n) % nre, np.arange(n) // nre
    h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
    state = np.zeros(16, np.float32)
    ce = np.ctypeslib.as_array(C.cast(res.ce[0][0], C.POINTER(C.c_float)), (2 * n,)).view(np.complex64)
    for step, (tti, coef) in enumerate([(0, (4.0, 1.5)), (1, (0.0, 0.0)), (5, (0.0, 0.0)), (6, (0.0, 0.0)), (7, (3.0, 1.0))]):
        sf_idx = tti % 10
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        orc.orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
        if sf_idx in (0, 5):
            kp, ks = 6 * nre + nre // 2 - 31, 5 * nre + nre // 2 - 31
            g[kp:kp + 62] = pss
            for k0 in (kp - 5, kp + 62, ks - 5, ks + 62):
                g[k0:k0 + 5] = 0
        grid = acopy((g * h + 0.03 * (1 + step) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        sf, rc, oc = RefDlSfCfg(), RefChestCfg(), OrcChestCfg()
        sf.tti = tti
        rc.noise_alg = oc.noise_alg = alg
        rc.filter_coef[0], rc.filter_coef[1] = coef
        oc.filter_coef[0], oc.filter_coef[1] = coef
        assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
        ref, ores = np.zeros(n, np.complex64), OrcChestRes()
        gp, cp = (C.c_void_p * 1)(grid.ctypes.data), (C.c_void_p * 1)(ref.ctypes.data)
        assert orc.orc_chest_dl_ports_state(C.byref(cell), sf_idx, C.byref(oc), 1, gp, cp, C.byref(ores), None, p(state)) == 0
        assert close(ce, ref), step
        assert abs(res.noise_estimate - ores.noise_estimate) <= 1e-4 * ores.noise_estimate and abs(res.snr_db - ores.snr_db) < 1e-3
        assert abs(res.rsrp_dbm - ores.rsrp_dbm) < 1e-3 and abs(res.snr_ant_port_db[0][0] - ores.snr_db) < 1e-3
    L.srslte_chest_dl_res_free(C.byref(res))
    L.srslte_chest_dl_free(est)


def test_chest_dl_object_mbsfn():
    """srslte_chest_dl_set_mbsfn_area_id + srslte_chest_dl_estimate_cfg on an MBSFN subframe (ue_dl.c:374-397 with cc_worker.cc:90-93's
    configuration, and with the REFS noise): the 12 estimated symbols, and a result struct whose rsrp / rssi-derived fields are the last
    normal subframe's while the noise (REFS) is the MBSFN subframe's, as the reference's fill_res reports them."""
    L, rng = hip(), np.random.default_rng(14)
    orc = oracle()
    orc.orc_chest_dl_mbsfn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    prb, cid, area = 50, 7, 33
    est, res = opaque(1 << 16), RefChestRes()
    assert L.srslte_chest_dl_init(est, prb, 1) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
    assert L.srslte_chest_dl_res_init(C.byref(res), prb) == 0
    n, nre = 14 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    k, l = np.arange(n) % nre, np.arange(n) // nre
    h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)
    g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    orc.orc_crs_put_sf(C.byref(cell), 0, 0, p(g))
    grid0 = acopy((g * h + 0.05 * rng.standard_normal(n)).astype(np.complex64).view(np.float32))
    sf, rc = RefDlSfCfg(), RefChestCfg()
    sf.tti = 0
    assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), (C.c_void_p * 4)(grid0.ctypes.data, 0, 0, 0), C.byref(res)) == 0
    r0, oc0 = OrcChestRes(), OrcChestCfg()
    assert orc.orc_chest_dl(C.byref(cell), 0, C.byref(oc0), p(grid0), None, C.byref(r0)) == 0
    prev_noise = res.noise_estimate
    sf.tti, sf.sf_type = 21, 1
    rc.mbsfn_area_id, rc.interpolate_subframe = area, True
    g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    assert orc.orc_mbsfn_put_sf(C.byref(cell), 1, 0, area, p(g)) == 0
    grid = acopy((g * h + 0.05 * rng.standard_normal(n)).astype(np.complex64).view(np.float32))
    inp = (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)
    assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), inp, C.byref(res)) != 0  # area id not set
    assert L.srslte_chest_dl_set_mbsfn_area_id(est, 256) != 0 and L.srslte_chest_dl_set_mbsfn_area_id(est, area) == 0
    ce = np.ctypeslib.as_array(C.cast(res.ce[0][0], C.POINTER(C.c_float)), (2 * n,)).view(np.complex64)
    for alg, ftype, coef in ((1, 1, 0.1), (0, 1, 0.1), (2, 2, 0.0), (0, 0, 0.0)):
        rc.noise_alg, rc.filter_type = alg, ftype
        rc.filter_coef[0] = coef
        ce[12 * nre:] = 7 + 7j
        rcode = L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), inp, C.byref(res))
        oc = OrcChestCfg()
        oc.noise_alg, oc.filter_type, oc.interpolate_subframe = alg, ftype, True
        oc.filter_coef[0] = coef
        ref, nz = np.zeros(n, np.complex64), C.c_float(0)
        assert rcode == 0 and orc.orc_chest_dl_mbsfn(C.byref(cell), 1, C.byref(oc), area, 0, p(grid), p(ref), C.byref(nz)) == 0
        assert close(ce[:12 * nre], ref[:12 * nre]) and (ce[12 * nre:] == 7 + 7j).all()
        if alg == 0:
            assert abs(res.noise_estimate - nz.value) <= 1e-4 * nz.value
            prev_noise = res.noise_estimate
        else:
            assert res.noise_estimate == prev_noise
        assert abs(res.rsrp - r0.rsrp) <= 1e-4 * r0.rsrp and abs(res.rsrq_db - r0.rsrq_db) < 1e-3 and abs(res.rssi_dbm - r0.rssi_dbm) < 1e-3
        assert abs(res.snr_db - 10 * np.log10(res.rsrp / res.noise_estimate)) < 1e-3
        assert abs(res.snr_ant_port_db[0][0] - res.snr_db) < 1e-3 and abs(res.rsrp_port_dbm[0] - r0.rsrp_dbm) < 1e-3
    rc.noise_alg, rc.filter_type = 1, 0
    rc.filter_coef[0] = 0.0  # automatic Gauss from a noise estimate this subframe does not make: refused, not guessed
    assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), inp, C.byref(res)) != 0
    L.srslte_chest_dl_res_free(C.byref(res))
    L.srslte_chest_dl_free(est)


@pytest.mark.parametrize("mod", [1, 2, 3, 4])
def test_demod_calls(mod):
    """soft_demod_test.c:118-249: srslte_demod_soft_demodulate{,_s,_b} on host arrays."""
    L, rng = hip(), np.random.default_rng(mod)
    nsym, qm = 1203, 2 * mod
    x = acopy(rng.standard_normal(2 * nsym).astype(np.float32))
    for fn, on, dt in (("srslte_demod_soft_demodulate", "orc_demod_soft_f", np.float32), ("srslte_demod_soft_demodulate_s", "orc_demod_soft_s", np.int16),
                       ("srslte_demod_soft_demodulate_b", "orc_demod_soft_b", np.int8)):
        a, b = np.zeros(nsym * qm, dt), np.zeros(nsym * qm, dt)
        assert getattr(L, fn)(mod, p(x), p(a), nsym) == 0
        getattr(oracle(), on)(mod, p(x), p(b), nsym)
        assert np.array_equal(a, b) if dt != np.float32 else np.abs(a - b).max() < 1e-6
    assert L.srslte_demod_soft_demodulate_s(9, p(x), p(a), nsym) == -1


def test_chest_dl_object_two_rx_antennas():
    """srslte_chest_dl_init(q, prb, 2) + estimate_cfg with two input grids (chest_dl.c:884-908): per-antenna estimates, antenna-averaged
    scalars and the per-antenna SNR / RSRP / RSRQ fields of fill_res (:860-870)."""
    L, rng = hip(), np.random.default_rng(44)
    prb, cid, sf_idx = 25, 9, 3
    est, res = opaque(1 << 16), RefChestRes()
    assert L.srslte_chest_dl_init(est, prb, 2) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
    n, nre = 14 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
    k = np.arange(n) % nre
    grids = [acopy((g * (amp * np.exp(1j * (ph + k / 80.0))) + nz * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
             for amp, ph, nz in ((2.0, 0.2, 0.1), (0.7, -0.9, 0.25))]
    ce = [aligned(2 * n, np.float32) for _ in range(2)]
    res.ce[0][0], res.ce[0][1] = ce[0].ctypes.data, ce[1].ctypes.data
    sf, rc, oc = RefDlSfCfg(), RefChestCfg(), OrcChestCfg()
    sf.tti = sf_idx
    rc.filter_coef[0], rc.filter_coef[1], oc.filter_coef[0], oc.filter_coef[1] = 4.0, 1.0, 4.0, 1.0
    inp = (C.c_void_p * 4)(grids[0].ctypes.data, grids[1].ctypes.data, 0, 0)
    assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
    ce_o, ores, single = [np.zeros(n, np.complex64) for _ in range(2)], OrcChestRes(), [OrcChestRes(), OrcChestRes()]
    gp, cp = (C.c_void_p * 2)(grids[0].ctypes.data, grids[1].ctypes.data), (C.c_void_p * 2)(ce_o[0].ctypes.data, ce_o[1].ctypes.data)
    assert oracle().orc_chest_dl_multi(C.byref(cell), sf_idx, C.byref(oc), 2, gp, cp, C.byref(ores)) == 0
    for a in range(2):
        assert close(ce[a].view(np.complex64), ce_o[a])
        assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(oc), p(grids[a]), None, C.byref(single[a])) == 0
        assert abs(res.snr_ant_port_db[a][0] - single[a].snr_db) < 1e-3
        assert abs(res.rsrp_ant_port_dbm[a][0] - single[a].rsrp_dbm) < 1e-3
        assert abs(res.rsrq_ant_port_db[a][0] - single[a].rsrq_db) < 1e-3
    for nm in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm"):
        x, y = getattr(res, nm), getattr(ores, nm)
        assert abs(x - y) <= 1e-4 * abs(y) + 1e-5, (nm, x, y)
    L.srslte_chest_dl_free(est)


@pytest.mark.parametrize("prb,cid,nrx,npt", [(25, 9, 1, 2), (100, 304, 1, 2), (50, 5, 2, 2), (6, 0, 2, 2), (25, 9, 1, 4), (100, 305, 1, 4), (50, 2, 2, 4), (6, 4, 2, 4)])
def test_chest_dl_object_two_ports(prb, cid, nrx, npt):
    """A 2-port cell through srslte_chest_dl_init / set_cell / estimate_cfg (chest_dl.c:884-908): ce[port][antenna], the aggregated
    scalars (noise over ports and antennas, get_rsrp's port-by-antenna-index maximum, last-estimate CFO) and the per-port /
    per-antenna fields of fill_res, against the oracle (itself pinned to the reference build for this case)."""
    L, rng = hip(), np.random.default_rng(55 + prb + nrx)
    n, nre = 14 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, npt, True)
    oracle().orc_chest_dl_ports.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    est, last_cfo = opaque(1 << 16), 0.0
    assert L.srslte_chest_dl_init(est, prb, nrx) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, npt, cid, 0, 0, 0, 0)) == 0
    for sf_idx, kw in ((0, {}), (3, {"interpolate_subframe": npt == 2, "filter_coef": (4.0, 2.0), "cfo_estimate_enable": True}),
                       (5, {"filter_coef": (4.0, 1.0)}), (8, {"filter_type": 1, "filter_coef": (0.1, 0.0)}), (9, {"filter_type": 2}),
                       (7, {"filter_coef": (4.0, 1.0), "sync_error_enable": True, "rsrp_neighbour": True})):
        k, l = np.arange(n) % nre, np.arange(n) // nre
        tx = []
        for port in range(npt):
            g = np.zeros(n, np.complex64)
            oracle().orc_crs_put_sf(C.byref(cell), sf_idx, port, p(g))
            tx.append(g)
        hole = np.zeros(n, bool)
        for g in tx:
            hole |= g != 0
        data = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        grids = []
        for a in range(nrx):
            rxg = np.where(hole, 0, data).astype(np.complex64) * (1.5 - 0.4 * a)
            for port in range(npt):
                rxg = rxg + tx[port] * ((2.0 - 0.35 * port + 0.3 * a) * (1 + 0.25 * np.sin(k / 30.0 + port + 2 * a)) *
                                        np.exp(1j * (0.4 * port - 0.9 * a + k / 80.0 + 0.05 * l))).astype(np.complex64)
            rxg = rxg + (0.05 + 0.1 * a) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
            grids.append(acopy(rxg.astype(np.complex64).view(np.float32)))
        res, sf, rc, oc = RefChestRes(), RefDlSfCfg(), RefChestCfg(), OrcChestCfg()
        for kk, v in kw.items():
            if kk == "filter_coef":
                rc.filter_coef[0], rc.filter_coef[1], oc.filter_coef[0], oc.filter_coef[1] = v + v
            else:
                setattr(rc, kk, v)
                setattr(oc, kk, v)
        rc.cfo_estimate_sf_mask = 0x3FF
        ce = [aligned(2 * n, np.float32) for _ in range(npt * nrx)]
        for i, c_ in enumerate(ce):
            res.ce[i // nrx][i % nrx] = c_.ctypes.data
        sf.tti = 20 + sf_idx
        inp = (C.c_void_p * 4)(*([g.ctypes.data for g in grids] + [0] * (4 - nrx)))
        assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
        ce_o, ores, raw = [np.zeros(n, np.complex64) for _ in range(npt * nrx)], OrcChestRes(), np.zeros(nrx * npt * 4, np.float32)
        gp = (C.c_void_p * nrx)(*[g.ctypes.data for g in grids])
        cp = (C.c_void_p * (npt * nrx))(*[c_.ctypes.data for c_ in ce_o])
        assert oracle().orc_chest_dl_ports(C.byref(cell), sf_idx, C.byref(oc), nrx, gp, cp, C.byref(ores), p(raw)) == 0
        for i in range(npt * nrx):
            assert close(ce[i].view(np.complex64), ce_o[i]), (sf_idx, i)
        for nm in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm"):
            x, y = getattr(res, nm), getattr(ores, nm)
            assert abs(x - y) <= 1e-4 * abs(y) + 1e-5, (nm, x, y, sf_idx)
        if kw.get("cfo_estimate_enable"):
            last_cfo = ores.cfo
        assert abs(res.cfo - last_cfo) <= 1e-4 * abs(last_cfo) + 1e-6  # q->cfo keeps the last enabled estimate (chest_dl.c:612-614,:849)
        if kw.get("sync_error_enable"):  # chest_dl.c:692-703,:859 and get_rsrp_neighbour :821-843
            assert abs(res.sync_error - ores.sync_error) <= 1e-4 * abs(ores.sync_error) + 1e-5, (res.sync_error, ores.sync_error)
            assert abs(res.rsrp_neigh - ores.rsrp_neigh) <= 1e-4 * abs(ores.rsrp_neigh) + 1e-9, (res.rsrp_neigh, ores.rsrp_neigh)
        else:
            assert np.isnan(res.sync_error)
        raw = raw.reshape(nrx, npt, 4)
        for port in range(npt):
            assert abs(res.rsrp_port_dbm[port] - (10 * np.log10(raw[:, port, 1].mean()) + 30)) < 1e-3
            for a in range(nrx):
                assert abs(res.snr_ant_port_db[a][port] - 10 * np.log10(raw[a, port, 1] / raw[a, port, 0])) < 1e-3
                assert abs(res.rsrp_ant_port_dbm[a][port] - (10 * np.log10(raw[a, port, 1]) + 30)) < 1e-3
                assert abs(res.rsrq_ant_port_db[a][port] - 10 * np.log10(prb * raw[a, port, 1] / raw[a, port, 2])) < 1e-3
    if npt == 4:  # interpolate_subframe: ports 2/3 take upstream's copy branch (chest_dl.c:467-471) - symbol 0 of what the CALLER's buffer held
        rc.interpolate_subframe = oc.interpolate_subframe = True
        before = [(rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) for _ in range(npt * nrx)]
        for i, c_ in enumerate(ce):
            c_.view(np.complex64)[:] = before[i]
        assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
        ce_o = [b_.copy() for b_ in before]
        cp = (C.c_void_p * (npt * nrx))(*[c_.ctypes.data for c_ in ce_o])
        assert oracle().orc_chest_dl_ports(C.byref(cell), sf_idx, C.byref(oc), nrx, gp, cp, C.byref(ores), p(np.zeros(nrx * npt * 4, np.float32))) == 0
        for i in range(npt * nrx):
            assert close(ce[i].view(np.complex64), ce_o[i]), ("interpolate_subframe", i)
            if i // nrx >= 2:
                assert np.array_equal(ce[i].view(np.complex64).reshape(14, nre), np.tile(before[i][:nre], (14, 1)))
    L.srslte_chest_dl_free(est)


def test_chest_dl_object_sync_error_and_neighbour_rsrp():
    """Single port, single antenna: cfg.sync_error_enable (mean normalised phase slope of the pilot estimates, chest_dl.c:692-703) and
    cfg.rsrp_neighbour (power of the coherent pilot mean, :706-709,:821-843) through the compat API, on a grid with a timing offset."""
    L, rng = hip(), np.random.default_rng(91)
    prb, cid = 50, 33
    est, res = opaque(1 << 16), RefChestRes()
    assert L.srslte_chest_dl_init(est, prb, 1) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, cid, 0, 0, 0, 0)) == 0
    assert L.srslte_chest_dl_res_init(C.byref(res), prb) == 0
    n, nre = 14 * 12 * prb, 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    for sf_idx, delay in ((1, 0.0), (6, 3.5), (9, -2.25)):
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
        k = np.arange(n) % nre
        grid = acopy((g * 1.3 * np.exp(-2j * np.pi * k * delay / 768.0) + 0.03 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
        sf, rc, oc = RefDlSfCfg(), RefChestCfg(), OrcChestCfg()
        sf.tti = sf_idx
        rc.filter_coef[0], rc.filter_coef[1], oc.filter_coef[0], oc.filter_coef[1] = 4.0, 1.0, 4.0, 1.0
        rc.sync_error_enable = oc.sync_error_enable = True
        rc.rsrp_neighbour = oc.rsrp_neighbour = True
        inp = (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0)
        assert L.srslte_chest_dl_estimate_cfg(est, C.byref(sf), C.byref(rc), inp, C.byref(res)) == 0
        ref, rres = np.zeros(n, np.complex64), OrcChestRes()
        assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(oc), p(grid), p(ref), C.byref(rres)) == 0
        assert abs(res.sync_error - rres.sync_error) <= 1e-4 * abs(rres.sync_error) + 1e-5, (res.sync_error, rres.sync_error)
        assert abs(res.rsrp_neigh - rres.rsrp_neigh) <= 1e-4 * abs(rres.rsrp_neigh) + 1e-9
        assert abs(rres.sync_error - delay) < 0.05  # timing error in samples
    L.srslte_chest_dl_res_free(C.byref(res))
    L.srslte_chest_dl_free(est)


def test_concurrent_host_threads():
    """SURVEY §8b "Threading": the single-call API is used from several worker threads at once, each with its own objects; the
    object-less demapper calls must be re-entrant too. Four threads run demapper + OFDM round trip + channel estimator concurrently
    (ctypes releases the GIL inside the calls); every result is checked against the oracle."""
    import threading
    L = hip()
    errors = []

    def worker(tid):
        try:
            rng = np.random.default_rng(500 + tid)
            prb = (6, 15, 25, 50)[tid]
            n = 14 * 12 * prb
            ifft, fft = opaque(1 << 12), opaque(1 << 12)
            N = oracle().orc_symbol_sz(prb)
            a, t, b = aligned(2 * n, np.float32), aligned(2 * 15 * N, np.float32), aligned(2 * n, np.float32)
            assert L.srslte_ofdm_tx_init(ifft, 0, p(a), p(t), prb) == 0 and L.srslte_ofdm_rx_init(fft, 0, p(t), p(b), prb) == 0
            L.srslte_ofdm_set_normalize(ifft, True)
            L.srslte_ofdm_set_normalize(fft, True)
            est, res = opaque(1 << 16), RefChestRes()
            assert L.srslte_chest_dl_init(est, prb, 1) == 0 and L.srslte_chest_dl_set_cell(est, RefCell(prb, 1, 10 + tid, 0, 0, 0, 0)) == 0
            assert L.srslte_chest_dl_res_init(C.byref(res), prb) == 0
            cell = OrcCell(10 + tid, prb, 1, True)
            for it in range(6):
                mod = 1 + (tid + it) % 4
                nsym = 1000 + 37 * tid + it
                x = acopy((rng.standard_normal(2 * nsym) * 0.8).astype(np.float32))
                l1, l2 = aligned(nsym * 2 * mod + 64, np.int16), aligned(nsym * 2 * mod + 64, np.int16)
                assert L.srslte_demod_soft_demodulate_s(mod, p(x), p(l1), nsym) == 0
                oracle().orc_demod_soft_s(mod, p(x), p(l2), nsym)
                assert np.array_equal(l1, l2), ("demod", tid, it)
                a[:] = rng.standard_normal(2 * n).astype(np.float32)
                L.srslte_ofdm_tx_sf(ifft)
                L.srslte_ofdm_rx_sf(fft)
                assert close(b, a), ("ofdm", tid, it)
                g = np.array(a).view(np.complex64).copy()
                oracle().orc_crs_put_sf(C.byref(cell), it, 0, p(g))
                grid = acopy((g * (2.0 + tid) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64).view(np.float32))
                sf = RefDlSfCfg()
                sf.tti = it
                assert L.srslte_chest_dl_estimate(est, C.byref(sf), (C.c_void_p * 4)(grid.ctypes.data, 0, 0, 0), C.byref(res)) == 0
                ref, rres, oc = np.zeros(n, np.complex64), OrcChestRes(), OrcChestCfg()
                assert oracle().orc_chest_dl(C.byref(cell), it, C.byref(oc), p(grid), p(ref), C.byref(rres)) == 0
                ce = np.ctypeslib.as_array(C.cast(res.ce[0][0], C.POINTER(C.c_float)), (2 * n,)).view(np.complex64)
                assert close(ce, ref), ("chest", tid, it)
            L.srslte_chest_dl_res_free(C.byref(res))
            L.srslte_chest_dl_free(est)
            L.srslte_ofdm_tx_free(ifft)
            L.srslte_ofdm_rx_free(fft)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_four_host_threads_overlap():
    """SURVEY 8b 'Threading': the reference's worker threads (three sf_workers in srsue) call the single-call API concurrently on distinct
    objects. Every host thread has its own non-blocking stream and pinned arena in the library: four threads, each decoding its own code
    blocks through srslte_tdec_run_all, must give the single-thread results, on FOUR DISTINCT non-blocking streams (on the null stream they
    serialised), and the device-side intervals of the four threads' work - HIP events recorded on each thread's stream around its calls - must
    share a common instant. Wall-clock ratios are printed, not asserted: a shared or slower box must not turn correct results red."""
    import threading
    import time
    L, K, reps = hip(), 5824, 24
    L.srslte_hip_compat_thread_stream.restype = C.c_void_p
    L.srslte_hip_compat_thread_stream.argtypes = [C.POINTER(C.c_uint)]
    L.srslte_hip_event_create.restype = C.c_void_p
    L.srslte_hip_event_record.argtypes = [C.c_void_p, C.c_void_p]
    L.srslte_hip_event_elapsed_ms.restype = C.c_float
    L.srslte_hip_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    jobs = []
    for t in range(4):
        tdec = opaque(1 << 16)
        assert L.srslte_tdec_init(tdec, 6144) == 0
        bits = rng.integers(0, 2, K).astype(np.uint8)
        enc = np.zeros(3 * K + 12, np.uint8)
        oracle().orc_tcod_encode_bits(p(bits), p(enc), K)
        llr = (100 * ((2.0 * enc - 1) + 0.8 * rng.standard_normal(enc.shape))).astype(np.int16)
        L.srslte_tdec_force_not_sb(tdec)
        ref = np.zeros(K // 8, np.uint8)
        oracle().orc_tdec_run(p(llr), False, K, 4, p(ref), None)
        jobs.append((tdec, llr, ref, np.zeros(K // 8, np.uint8)))

    marks, errors = {}, []
    start = threading.Barrier(4)

    def work(j, tid=None):
        tdec, llr, ref, out = j
        try:
            if tid is not None:
                flags = C.c_uint(0xFFFF)
                stream = L.srslte_hip_compat_thread_stream(C.byref(flags))
                e0, e1 = L.srslte_hip_event_create(), L.srslte_hip_event_create()
                start.wait(timeout=60)
                assert L.srslte_hip_event_record(e0, stream) == 0
            for _ in range(reps):
                assert L.srslte_tdec_run_all(tdec, p(llr), p(out), 4, K) == 0
            if tid is not None:
                assert L.srslte_hip_event_record(e1, stream) == 0
                marks[tid] = (stream, flags.value, e0, e1)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(repr(e))

    work(jobs[0])  # warm up (tables, first-touch)
    base = L.srslte_hip_event_create()
    assert L.srslte_hip_event_record(base, None) == 0 and L.srslte_hip_sync() == 0
    t0 = time.perf_counter()
    work(jobs[0])
    t_one = time.perf_counter() - t0
    th = [threading.Thread(target=work, args=(j, i)) for i, j in enumerate(jobs)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    t_four = time.perf_counter() - t0
    assert not errors and len(marks) == 4, errors
    assert L.srslte_hip_sync() == 0
    for tdec, llr, ref, out in jobs:
        assert np.array_equal(out, ref)
        L.srslte_tdec_free(tdec)
    streams = [m[0] for m in marks.values()]
    assert all(streams) and len(set(streams)) == 4, streams          # a stream per host thread, none of them the null stream
    assert all(m[1] & 1 for m in marks.values()), [m[1] for m in marks.values()]  # hipStreamNonBlocking
    iv = [(L.srslte_hip_event_elapsed_ms(base, m[2]), L.srslte_hip_event_elapsed_ms(base, m[3])) for m in marks.values()]
    assert all(b > a for a, b in iv), iv
    assert max(a for a, b in iv) < min(b for a, b in iv), "the four threads' device intervals share no instant: %s" % iv
    print("four host threads: %.2f x the single-thread time (intervals on their streams, ms after the base event: %s)" % (t_four / t_one, iv))
