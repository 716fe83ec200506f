"""Parity of the HIP path (through the C ABI of libsrslte_phy_hip.so) against the oracle on seeded inputs.

Bars (BASELINE.json north_star / SURVEY §8d): hard bits, decoded blocks, iteration counts, encoder output and integer
LLRs fed identical symbols: bit-exact. Float arrays: |a-b| <= 1e-4 * max(|b|, rms(b)) per element.
"""
import ctypes as C
import importlib

import numpy as np
import pytest

from _libs import OrcCell, OrcChestCfg, OrcChestRes, OrcOfdm, acopy, oracle, p
from lte_sim import DlConfig, UlConfig, make_subframe, make_ul_subframe, oracle_rx, oracle_ul_rx

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def hp():
    return importlib.import_module("srslte-emane_amd")


def assert_close_c(a, b, what, tol=TOL):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    ref = max(np.abs(b).max(), np.sqrt((np.abs(b) ** 2).mean()))
    err = np.abs(a - b).max()
    assert err <= tol * ref, "%s: max err %.3e vs %.3e allowed" % (what, err, tol * ref)


# ---------------------------------------------------------------- DFT / OFDM
DFT_SIZES = [128, 256, 384, 512, 768, 1024, 1536, 2048] + [12 * n for n in (1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 24, 25, 27,
                                                                           30, 32, 36, 40, 45, 48, 50, 54, 60, 64, 72, 75, 80, 81, 90, 96, 100)]


@pytest.mark.parametrize("N", DFT_SIZES)
def test_dft_sizes(hp, N):
    rng = np.random.default_rng(N)
    x = (rng.standard_normal((3, N)) + 1j * rng.standard_normal((3, N))).astype(np.complex64)
    for fwd in (True, False):
        y = hp.dft(x, forward=fwd)
        ref = np.zeros_like(x)
        for i in range(3):
            oracle().orc_dft_exact(p(x[i]), p(ref[i]), N, 1 if fwd else 0)
        assert_close_c(y, ref, "dft N=%d fwd=%s" % (N, fwd))


# symbol size None: srslte_symbol_sz of the default rate family; else the power-of-two family of srslte_use_standard_symbol_size(true)
# (phy_common.c:304-345), handed over as srslte_ofdm_init_ takes it (ofdm.c:38-57)
@pytest.mark.parametrize("prb,N", [(6, None), (15, None), (25, None), (50, None), (75, None), (100, None), (110, None),
                                   (25, 512), (50, 1024), (75, 1536), (100, 2048), (110, 2048)])
@pytest.mark.parametrize("norm,shift", [(False, 0.0), (True, 0.0), (True, 0.5), (False, -0.5)])
def test_ofdm_rx_tx(hp, prb, N, norm, shift):
    rng = np.random.default_rng(prb)
    q = OrcOfdm()
    if N is None:
        assert oracle().orc_ofdm_init(C.byref(q), prb, True) == 0
    else:
        assert oracle().orc_ofdm_init_sz(C.byref(q), prb, N, True) == 0 and q.symbol_sz == N
    q.normalize, q.exact = norm, prb <= 25
    if shift:
        q.freq_shift, q.freq_shift_f = True, shift
    nsf = 3
    grid = (rng.standard_normal((nsf, 14 * 12 * prb)) + 1j * rng.standard_normal((nsf, 14 * 12 * prb))).astype(np.complex64)
    tx = hp.Ofdm(prb, True, rx=False, symbol_sz=N)
    rx = hp.Ofdm(prb, True, rx=True, symbol_sz=N)
    assert tx.sf_len == rx.sf_len == q.sf_sz
    for o in (tx, rx):
        o.set_normalize(norm)
        if shift:
            o.set_freq_shift(shift)
    t_gpu = tx.tx_sf(grid)
    t_ref = np.zeros((nsf, q.sf_sz), np.complex64)
    for i in range(nsf):
        oracle().orc_ofdm_tx_sf(C.byref(q), p(grid[i]), p(t_ref[i]))
    assert_close_c(t_gpu, t_ref, "ofdm_tx prb=%d" % prb)
    time_in = (rng.standard_normal((nsf, q.sf_sz)) + 1j * rng.standard_normal((nsf, q.sf_sz))).astype(np.complex64)
    g_gpu = rx.rx_sf(time_in)
    g_ref = np.zeros((nsf, 14 * 12 * prb), np.complex64)
    for i in range(nsf):
        oracle().orc_ofdm_rx_sf(C.byref(q), p(time_in[i]), p(g_ref[i]))
    assert_close_c(g_gpu, g_ref, "ofdm_rx prb=%d" % prb)
    if not shift:  # ofdm_test.c:74-179 round trip, MSE bound 0.07 upstream (normalised) - ours must be ~0
        back = rx.rx_sf(tx.tx_sf(grid))
        scale = 1.0 if norm else float(q.symbol_sz)
        assert np.mean(np.abs(back / scale - grid) ** 2) < 1e-9
    tx.free()
    rx.free()


@pytest.mark.parametrize("N", [1, 7, 139, 255, 839, 2049, 4096])
def test_dft_any_length(hp, N):
    """FFTW plans every length (dft_fftw.c:93-117) and callers beyond the hot path use that (prach.c: 839 / 139 point sequences,
    dft_test -N 255): lengths without a 2/3/5 plan or above 2048 are served by the direct-sum kernel."""
    rng = np.random.default_rng(N)
    x = (rng.standard_normal((2, N)) + 1j * rng.standard_normal((2, N))).astype(np.complex64)
    for fwd in (True, False):
        y = hp.dft(x, forward=fwd)
        ref = np.zeros_like(x)
        for i in range(2):
            oracle().orc_dft_exact(p(x[i]), p(ref[i]), N, 1 if fwd else 0)
        assert_close_c(y, ref, "dft N=%d fwd=%s" % (N, fwd))


def test_dft_precoding_invalid(hp):
    rc, _ = hp.dft_precoding(np.zeros(12 * 7, np.complex64), 7, 1)
    assert rc == hp.SRSLTE_ERROR  # dft_precoding.c:104-107


@pytest.mark.parametrize("L", [1, 2, 3, 5, 25, 50, 100])
def test_dft_precoding(hp, L):
    rng = np.random.default_rng(L)
    x = (rng.standard_normal((12, 12 * L)) + 1j * rng.standard_normal((12, 12 * L))).astype(np.complex64)
    for fwd in (True, False):
        rc, y = hp.dft_precoding(x, L, 12, forward=fwd)
        assert rc == 0
        ref = np.zeros_like(x)
        assert oracle().orc_dft_precoding(p(x), p(ref), L, 12, 1 if fwd else 0, True) == 0
        assert_close_c(y, ref, "precoding L=%d" % L)


# ---------------------------------------------------------------- demapper
@pytest.mark.parametrize("mod", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("nsym", [1, 3, 4, 7, 8, 15, 33, 14580])
def test_demod_soft(hp, mod, nsym):
    rng = np.random.default_rng(100 * mod + nsym)
    qm = 1 if mod == 0 else 2 * mod
    ncalls = 3
    for scale in (1.0, 40.0):
        s = ((rng.standard_normal((ncalls, nsym)) + 1j * rng.standard_normal((ncalls, nsym))) * scale).astype(np.complex64)
        for kind, dt in (("f", np.float32), ("s", np.int16), ("b", np.int8)):
            rc, llr = hp.demod_soft_demodulate(mod, s, kind, ncalls)
            assert rc == 0
            ref = np.zeros((ncalls, nsym * qm), dt)
            for c in range(ncalls):
                getattr(oracle(), "orc_demod_soft_" + kind)(mod, p(acopy(s[c].view(np.float32))), p(ref[c]), nsym)
            if kind == "f":
                assert_close_c(llr, ref, "demod f mod=%d" % mod, 1e-6)
            else:
                assert np.array_equal(llr, ref), "demod %s mod=%d nsym=%d: %d mismatches" % (kind, mod, nsym, (llr != ref).sum())


def test_demod_invalid_mod(hp):
    rc, _ = hp.demod_soft_demodulate(7, np.zeros(4, np.complex64), "s")
    assert rc == hp.SRSLTE_ERROR  # demod_soft.c:496-498


# ---------------------------------------------------------------- channel estimator
def _chest_case(prb, cid, sf_idx, rng, snr_db=20):
    nre, n = 12 * prb, 14 * 12 * prb
    cell = OrcCell(cid, prb, 1, True)
    g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    oracle().orc_crs_put_sf(C.byref(cell), sf_idx, 0, p(g))
    k, l = np.arange(n) % nre, np.arange(n) // nre
    h = ((3 + np.sin(k / 40.0)) * np.exp(1j * (k / 100.0 + 0.1 * l))).astype(np.complex64)  # as chest_test_dl.c:157-161, smoother
    sig = 10 ** (-snr_db / 20)
    return cell, (g * h + sig * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


CHEST_CFGS = [{}, {"filter_coef": (4.0, 1.0)}, {"interpolate_subframe": True, "filter_coef": (4.0, 2.0), "cfo_estimate_enable": True},
              {"interpolate_subframe": True, "filter_type": 2}, {"filter_type": 1, "filter_coef": (0.1, 0.0)}, {"filter_type": 2}]


@pytest.mark.parametrize("prb,cid", [(6, 1), (6, 0), (25, 2), (50, 3), (100, 1), (100, 4), (100, 5), (15, 150)])
@pytest.mark.parametrize("ci", range(len(CHEST_CFGS)))
def test_chest_dl(hp, prb, cid, ci):
    rng = np.random.default_rng(prb * 1000 + cid)
    tti0, nsf = 8, 4  # covers sf 8, 9, 0, 1
    grids, cells = [], None
    for b in range(nsf):
        cells, g = _chest_case(prb, cid, (tti0 + b) % 10, rng)
        grids.append(g)
    hc, oc = hp.ChestDlCfg(), OrcChestCfg()
    for k, v in CHEST_CFGS[ci].items():
        if k == "filter_coef":
            hc.filter_coef[0], hc.filter_coef[1] = v
            oc.filter_coef[0], oc.filter_coef[1] = v
        else:
            setattr(hc, k, 1 if v is True else v)
            setattr(oc, k, v)
    est = hp.ChestDl(cid, prb)
    ce, res = est.estimate(np.stack(grids), tti0, hc)
    for b in range(nsf):
        ref, rres = np.zeros(14 * 12 * prb, np.complex64), OrcChestRes()
        assert oracle().orc_chest_dl(C.byref(cells), (tti0 + b) % 10, C.byref(oc), p(grids[b]), p(ref), C.byref(rres)) == 0
        assert_close_c(ce[b], ref, "ce prb=%d sf=%d cfg=%d" % (prb, b, ci))
        for name in ("noise_estimate", "rsrp", "rsrq", "cfo"):
            a, r = float(res[name][b]), float(getattr(rres, name))
            assert abs(a - r) <= 1e-4 * abs(r) + 1e-9, (name, a, r)
        for name in ("noise_estimate_dbm", "snr_db", "rsrp_dbm", "rsrq_db", "rssi_dbm"):
            assert abs(float(res[name][b]) - float(getattr(rres, name))) <= 1e-3, name
    est.free()


@pytest.mark.parametrize("prb,cid,npt,nrx", [(6, 1, 1, 1), (25, 2, 1, 2), (50, 3, 2, 1), (100, 4, 1, 1), (100, 5, 2, 2)])
@pytest.mark.parametrize("alg", [1, 2])
def test_chest_dl_noise_pss_empty(hp, prb, cid, npt, nrx, alg):
    """cfg.noise_alg PSS / EMPTY (chest_dl.c:381-411,:657-672) on the device vs the oracle run subframe by subframe with its kept
    estimate: renewed in subframes 0 and 5 from the PSS carriers / the empty carriers, carried over the other subframes of the batch
    and into the next call on the object; estimates, per-(port, antenna) noise, combined result fields."""
    orc = oracle()
    orc.orc_chest_dl_ports_state.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(2100 + prb + cid + alg)
    nre, n = 12 * prb, 14 * 12 * prb
    cell = OrcCell(cid, prb, npt, True)
    pss = np.zeros(62, np.complex64)
    orc.orc_pss_generate(cid % 3, p(pss))
    k, l = np.arange(n) % nre, np.arange(n) // nre
    est = hp.ChestDl(cid, prb, npt)
    state = np.zeros(16, np.float32)  # oracle's, [antenna][port]
    for call, (tti0, nsf, kw) in enumerate([(8, 9, {"filter_coef": (4.0, 1.5)}), (17, 5, {"filter_type": 1, "filter_coef": (0.1, 0.0)}),
                                            (22, 1, {}), (23, 4, {"filter_type": 2, "interpolate_subframe": True})]):
        grids = np.zeros((nsf, nrx, n), np.complex64)
        for b in range(nsf):
            sf_idx = (tti0 + b) % 10
            g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
            for pp in range(npt):
                orc.orc_crs_put_sf(C.byref(cell), sf_idx, pp, p(g))
            if sf_idx in (0, 5):
                kp, ks = 6 * nre + nre // 2 - 31, 5 * nre + nre // 2 - 31
                g[kp:kp + 62] = pss
                for k0 in (kp - 5, kp + 62, ks - 5, ks + 62):
                    g[k0:k0 + 5] = 0
            for a in range(nrx):
                h = ((3 + np.sin(k / 40.0 + a)) * np.exp(1j * (k / 100.0 + 0.1 * l + a))).astype(np.complex64)
                grids[b, a] = (g * h + 0.05 * (1 + b + call) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        hc, oc = hp.ChestDlCfg(), OrcChestCfg()
        for k_, v in kw.items():
            if k_ == "filter_coef":
                hc.filter_coef[0], hc.filter_coef[1] = v
                oc.filter_coef[0], oc.filter_coef[1] = v
            else:
                setattr(hc, k_, 1 if v is True else (0 if v is False else v))
                setattr(oc, k_, v)
        hc.noise_alg = oc.noise_alg = alg
        # what the estimate buffers hold before the call matters for ports 2/3 with interpolate_subframe: upstream replicates their symbol 0
        ce0 = (rng.standard_normal((nsf, npt, nrx, n)) + 1j * rng.standard_normal((nsf, npt, nrx, n))).astype(np.complex64)
        rc, ce, res, raw = est.estimate_multi(grids, tti0, hc, nrx, ce_in=ce0)
        assert rc == 0
        for b in range(nsf):
            ce2 = [ce0[b, i // nrx, i % nrx].copy() for i in range(npt * nrx)]
            ores = OrcChestRes()
            gl = [np.ascontiguousarray(grids[b, a]) for a in range(nrx)]
            gp, cp = (C.c_void_p * nrx)(*[x.ctypes.data for x in gl]), (C.c_void_p * (npt * nrx))(*[c.ctypes.data for c in ce2])
            assert orc.orc_chest_dl_ports_state(C.byref(cell), (tti0 + b) % 10, C.byref(oc), nrx, gp, cp, C.byref(ores), None, p(state)) == 0
            for pt in range(npt):
                for a in range(nrx):
                    assert_close_c(ce[b, pt, a], ce2[pt * nrx + a], "ce call %d sf %d port %d ant %d" % (call, b, pt, a))
                    want = state[a * npt + pt]
                    assert abs(raw[b, pt, a, 0] - want) <= 1e-4 * want + 1e-12, (call, b, pt, a, raw[b, pt, a, 0], want)
            for name in ("noise_estimate", "rsrp", "rsrq"):
                x, y = float(res[name][b]), float(getattr(ores, name))
                assert abs(x - y) <= 1e-4 * abs(y) + 1e-9, (name, call, b, x, y)
            if call or b >= 2:  # a noise estimate exists from the first subframe 0 on
                for name in ("noise_estimate_dbm", "snr_db", "rsrp_dbm", "rsrq_db", "rssi_dbm"):
                    assert abs(float(res[name][b]) - float(getattr(ores, name))) <= 1e-3, name
    hc = hp.ChestDlCfg()
    hc.noise_alg = alg
    assert est.estimate_multi(grids, 0, hc, nrx)[0] == hp.SRSLTE_ERROR  # automatic Gauss over a batch: a sequential chain, one subframe per call
    est.free()


@pytest.mark.parametrize("prb,cid,npt,nrx", [(6, 1, 1, 1), (25, 2, 2, 2), (50, 3, 1, 1), (100, 4, 2, 1), (15, 150, 1, 2)])
def test_chest_dl_extended_cp(hp, prb, cid, npt, nrx):
    """Extended-CP cells (12 symbols per subframe, CRS on symbols 0, 3, 6, 9, N_cp = 0 in the pilot sequences; chest_dl.c:497-502) on the
    device vs the oracle (pinned on the reference): every filter / interpolation configuration, CFO, and the PSS / EMPTY noise positions."""
    orc = oracle()
    orc.orc_chest_dl_ports_state.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(2300 + prb + cid)
    nre, n = 12 * prb, 12 * 12 * prb
    cell = OrcCell(cid, prb, npt, False)
    pss = np.zeros(62, np.complex64)
    orc.orc_pss_generate(cid % 3, p(pss))
    k, l = np.arange(n) % nre, np.arange(n) // nre
    est = hp.ChestDl(cid, prb, npt, cp_norm=False)
    state = np.zeros(16, np.float32)
    cfgs = [dict(kw) for kw in CHEST_CFGS] + [{"noise_alg": 1, "filter_coef": (4.0, 1.5)}, {"noise_alg": 2, "filter_type": 1, "filter_coef": (0.1, 0.0)}]
    for call, kw in enumerate(cfgs):
        tti0, nsf = 8 + call, 4
        grids = np.zeros((nsf, nrx, n), np.complex64)
        for b in range(nsf):
            sf_idx = (tti0 + b) % 10
            g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
            for pp in range(npt):
                orc.orc_crs_put_sf(C.byref(cell), sf_idx, pp, p(g))
            if sf_idx in (0, 5):
                kp, ks = 5 * nre + nre // 2 - 31, 4 * nre + nre // 2 - 31
                g[kp:kp + 62] = pss
                for k0 in (kp - 5, kp + 62, ks - 5, ks + 62):
                    g[k0:k0 + 5] = 0
            for a in range(nrx):
                h = ((3 + np.sin(k / 40.0 + a)) * np.exp(1j * (k / 100.0 + 0.1 * l + a))).astype(np.complex64)
                grids[b, a] = (g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        hc, oc = hp.ChestDlCfg(), OrcChestCfg()
        for k_, v in kw.items():
            if k_ == "filter_coef":
                hc.filter_coef[0], hc.filter_coef[1] = v
                oc.filter_coef[0], oc.filter_coef[1] = v
            else:
                setattr(hc, k_, 1 if v is True else v)
                setattr(oc, k_, v)
        rc, ce, res, raw = est.estimate_multi(grids, tti0, hc, nrx)
        assert rc == 0
        if kw.get("noise_alg") and not cfgs[call - 1].get("noise_alg"):
            state[:] = 0  # the batched object keeps its PSS / EMPTY estimate between PSS / EMPTY calls only (phy_hip.h)
        for b in range(nsf):
            ce2 = [np.zeros(n, np.complex64) for _ in range(npt * nrx)]
            ores = OrcChestRes()
            gl = [np.ascontiguousarray(grids[b, a]) for a in range(nrx)]
            gp, cp = (C.c_void_p * nrx)(*[x.ctypes.data for x in gl]), (C.c_void_p * (npt * nrx))(*[c.ctypes.data for c in ce2])
            assert orc.orc_chest_dl_ports_state(C.byref(cell), (tti0 + b) % 10, C.byref(oc), nrx, gp, cp, C.byref(ores), None, p(state)) == 0
            for pt in range(npt):
                for a in range(nrx):
                    assert_close_c(ce[b, pt, a], ce2[pt * nrx + a], "ce cfg %d sf %d port %d ant %d" % (call, b, pt, a))
            names = ("noise_estimate", "rsrp", "rsrq") + (("cfo",) if kw.get("cfo_estimate_enable") else ())
            if kw.get("noise_alg") and not state[:npt * nrx].all():
                names = ("rsrp", "rsrq")  # no PSS / EMPTY estimate yet
            for name in names:
                x, y = float(res[name][b]), float(getattr(ores, name))
                assert abs(x - y) <= 1e-4 * abs(y) + 1e-9, (name, call, b, x, y)
    est.free()


MBSFN_CFGS = [{"filter_type": 1, "filter_coef": (0.1, 0.0), "noise_alg": 1}, {"filter_type": 2}, {"filter_type": 1, "filter_coef": (0.2, 0.0)},
              {"filter_coef": (4.0, 1.5)}, {}]


@pytest.mark.parametrize("prb,cid,area,nports,nrx", [(6, 1, 1, 1, 1), (25, 2, 0, 1, 2), (50, 3, 255, 1, 1), (100, 4, 17, 1, 1), (100, 5, 2, 2, 2),
                                                     (15, 150, 77, 2, 1), (110, 9, 40, 1, 1)])
@pytest.mark.parametrize("ci", range(len(MBSFN_CFGS)))
def test_chest_dl_mbsfn(hp, prb, cid, area, nports, nrx, ci):
    """MBSFN subframes (chest_dl.c:718-745 and the MBSFN branches of :304-556) on the device vs the oracle (pinned on the
    reference's estimator): the 12 estimated symbols of every (port, antenna) and the REFS noise; with the applications' configuration
    (triangle 0.1, PSS noise: cc_worker.cc:90-93), other filters, 1-2 ports, 1-2 antennas."""
    orc = oracle()
    orc.orc_chest_dl_mbsfn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(prb * 1000 + cid + area)
    nre, n = 12 * prb, 14 * 12 * prb
    cell = OrcCell(cid, prb, nports, True)
    tti0, nsf = 1, 3  # MBSFN subframes 1, 2, 3
    grids = np.zeros((nsf, nrx, n), np.complex64)
    for b in range(nsf):
        g = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
        assert orc.orc_mbsfn_put_sf(C.byref(cell), tti0 + b, 0, area, p(g)) == 0
        k, l = np.arange(n) % nre, np.arange(n) // nre
        for a in range(nrx):
            h = ((3 + np.sin(k / 40.0 + a)) * np.exp(1j * (k / 100.0 + 0.1 * l + a))).astype(np.complex64)
            grids[b, a] = (g * h + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    hc, oc = hp.ChestDlCfg(), OrcChestCfg()
    for k_, v in MBSFN_CFGS[ci].items():
        if k_ == "filter_coef":
            hc.filter_coef[0], hc.filter_coef[1] = v
            oc.filter_coef[0], oc.filter_coef[1] = v
        else:
            setattr(hc, k_, v)
            setattr(oc, k_, v)
    hc.mbsfn_area_id, hc.interpolate_subframe, oc.interpolate_subframe = area, 1, True
    est = hp.ChestDl(cid, prb, nports)
    assert est.estimate_mbsfn(grids, tti0, hc, nrx)[0] == hp.SRSLTE_ERROR  # area id not initialised (chest_dl.c:729-731)
    assert est.set_mbsfn_area_id(area) == 0 and est.set_mbsfn_area_id(area) == 0
    rc, ce, noise = est.estimate_mbsfn(grids, tti0, hc, nrx)
    assert rc == 0
    for b in range(nsf):
        for pt in range(nports):
            for a in range(nrx):
                ref, nz = np.zeros(n, np.complex64), C.c_float(0)
                assert orc.orc_chest_dl_mbsfn(C.byref(cell), tti0 + b, C.byref(oc), area, pt, p(np.ascontiguousarray(grids[b, a])), p(ref), C.byref(nz)) == 0
                assert_close_c(ce[b, pt, a, :12 * nre], ref[:12 * nre], "ce sf %d port %d ant %d" % (b, pt, a))
                assert not ce[b, pt, a, 12 * nre:].any()  # symbols 12, 13 are not part of the subframe
                if hc.noise_alg == 0:
                    assert abs(noise[b, pt, a] - nz.value) <= 1e-4 * nz.value, (noise[b, pt, a], nz.value)
    hc.interpolate_subframe = 0
    assert est.estimate_mbsfn(grids, tti0, hc, nrx)[0] == hp.SRSLTE_ERROR
    est.free()


# ---------------------------------------------------------------- turbo
def _noisy_llr(rng, enc_bits, snr_db, scale=100):
    tx = 2.0 * enc_bits.astype(np.float64) - 1.0
    return (scale * (tx + 10 ** (-snr_db / 20) * rng.standard_normal(tx.shape))).clip(-32000, 32000).astype(np.int16)


@pytest.mark.parametrize("K", [40, 176, 400, 408, 504, 800, 816, 1008, 2048, 5824, 6144])
def test_tcod_encode(hp, K):
    rng = np.random.default_rng(K)
    bits = rng.integers(0, 2, (5, K)).astype(np.uint8)
    rc, out = hp.tcod_encode(bits, K)
    assert rc == 0
    for i in range(5):
        ref = np.zeros(3 * K + 12, np.uint8)
        assert oracle().orc_tcod_encode_bits(p(bits[i]), p(ref), K) == 0
        assert np.array_equal(out[i], ref), "tcod K=%d block %d" % (K, i)


def test_tcod_encode_every_block_length(hp):
    """All 188 block lengths of 36.212 Table 5.1.3-3 (every QPP interleaver) through the device encoder against the oracle's."""
    sizes = list(range(40, 512, 8)) + list(range(512, 1024, 16)) + list(range(1024, 2048, 32)) + list(range(2048, 6145, 64))
    assert len(sizes) == 188
    for K in sizes:
        bits = np.random.default_rng(5000 + K).integers(0, 2, (2, K)).astype(np.uint8)
        rc, out = hp.tcod_encode(bits, K)
        assert rc == 0, K
        for i in range(2):
            ref = np.zeros(3 * K + 12, np.uint8)
            assert oracle().orc_tcod_encode_bits(p(bits[i]), p(ref), K) == 0
            assert np.array_equal(out[i], ref), "tcod K=%d block %d" % (K, i)


def test_tcod_invalid_len(hp):
    rc, _ = hp.tcod_encode(np.zeros(41, np.uint8), 41)
    assert rc == hp.SRSLTE_ERROR  # turbocoder.c:89-93


@pytest.mark.parametrize("K", [40, 176, 400, 408, 504, 800, 816, 1008, 2048, 5824, 6144])
def test_tdec_run_all(hp, K):
    """srslte_tdec_run_all on [s p0 p1] input (force_not_sb), as turbodecoder_test.c:117-311."""
    rng = np.random.default_rng(K)
    ncb = 9
    dec = hp.Tdec(6144, 16)
    bits = rng.integers(0, 2, (ncb, K)).astype(np.uint8)
    enc = np.zeros((ncb, 3 * K + 12), np.uint8)
    for i in range(ncb):
        oracle().orc_tcod_encode_bits(p(bits[i]), p(enc[i]), K)
    for snr, scale in ((0.0, 100), (2.0, 100), (1.0, 400), (-3.0, 2000)):
        llr = _noisy_llr(rng, enc, snr, scale)
        for nit in (1, 2, 3, 6):
            rc, out, _, _ = dec.run_all(llr, K, nit)
            assert rc == 0
            for i in range(ncb):
                ref = np.zeros(K // 8, np.uint8)
                assert oracle().orc_tdec_run(p(llr[i]), False, K, nit, p(ref), None) == 0
                assert np.array_equal(out[i], ref), "tdec K=%d snr=%s nit=%d cb=%d: %d byte mismatches" % (K, snr, nit, i, (out[i] != ref).sum())
    dec.free()


@pytest.mark.parametrize("K,llr8", [(6144, False), (6144, True), (800, False), (400, False), (40, False), (5824, False)])
def test_tdec_full_object_boundaries(hp, K, llr8):
    """The decoder object filled to the last block it was created for, with caller strides that are odd / not multiples of anything:
    every per-wave slab (work arrays, x/y scratch, beta) and every caller buffer is indexed up to its last element. An index that
    runs past a block's share shows as a wrong block, a changed guard word, or a fault here rather than in a benchmark run (round 1
    has one unexplained `Memory access fault` on record from an uncommitted decoder build). max_nof_cb = 61 is neither a
    multiple of 8 (the generic kernel packs 8 blocks per wave) nor of the wave count of a CU."""
    L = hp.lib()
    rng = np.random.default_rng(K + 17)
    ncb, nit = 61, 3
    dec = hp.Tdec(K, ncb)  # exactly this length, exactly this many blocks: no slack in any slab
    bits = rng.integers(0, 2, (ncb, K)).astype(np.uint8)
    enc = np.zeros((ncb, 3 * K + 12), np.uint8)
    for i in range(ncb):
        oracle().orc_tcod_encode_bits(p(bits[i]), p(enc[i]), K)
    llr = _noisy_llr(rng, enc, 1.5, 20 if llr8 else 300)
    if llr8:
        llr = np.clip(llr, -127, 127).astype(np.int8)
    in_stride, out_stride, G = 3 * K + 12 + 5, K // 8 + 3, 64  # odd strides; G guard elements around every caller buffer
    x = np.full(G + ncb * in_stride + G, 77, llr.dtype)
    for i in range(ncb):
        x[G + i * in_stride:G + i * in_stride + 3 * K + 12] = llr[i]
    din = hp.DevBuf.from_host(x)
    dout, dit, dok = hp.DevBuf.from_host(np.full(G + ncb * out_stride + G, 0xA5, np.uint8)), hp.DevBuf.from_host(np.full(ncb + 2 * G, 0xDEADBEEF, np.uint32)), \
        hp.DevBuf.from_host(np.full(ncb + 2 * G, 0x5A, np.uint8))
    isz = x.dtype.itemsize
    fn = L.srslte_hip_tdec_run_batch_8bit if llr8 else L.srslte_hip_tdec_run_batch
    rc = fn(dec.h, din.ptr + G * isz, in_stride, 0, K, ncb, nit, 0, 0, dout.ptr + G, out_stride, dit.ptr + 4 * G, dok.ptr + G, None)
    assert rc == 0
    hp.sync()
    out, it, ok = dout.to_host(np.uint8), dit.to_host(np.uint32), dok.to_host(np.uint8)
    assert np.all(out[:G] == 0xA5) and np.all(out[G + ncb * out_stride:] == 0xA5) and np.all(it[:G] == 0xDEADBEEF) and np.all(it[G + ncb:] == 0xDEADBEEF)
    assert np.all(ok[:G] == 0x5A) and np.all(ok[G + ncb:] == 0x5A) and np.all(it[G:G + ncb] == nit)
    assert np.array_equal(din.to_host(x.dtype), x)  # the input is read-only
    for i in range(ncb):
        row = out[G + i * out_stride:G + (i + 1) * out_stride]
        ref = np.zeros(K // 8, np.uint8)
        if llr8:
            assert oracle().orc_tdec_run_8bit(p(np.ascontiguousarray(llr[i])), False, K, nit, p(ref), None) == 0
        else:
            assert oracle().orc_tdec_run(p(np.ascontiguousarray(llr[i])), False, K, nit, p(ref), None) == 0
        assert np.array_equal(row[:K // 8], ref), (K, i)
        assert np.all(row[K // 8:] == 0xA5)  # the bytes between two blocks' outputs stay untouched
    dec.free()


@pytest.mark.parametrize("K,W", [(504, 0), (504, 16), (1008, 8), (1008, 0), (6144, 8)])
def test_tdec_manual_numerics(hp, K, W):
    """srslte_tdec_init_manual: force generic / sse16 / avx16 numerics on any K (turbodecoder.c:168-215)."""
    if W == 16 and (K % 16 or K // 16 < 40):
        pytest.skip("window shorter than the 40-step overlap")
    rng = np.random.default_rng(K + W)
    dec = hp.Tdec(6144, 8)
    bits = rng.integers(0, 2, (3, K)).astype(np.uint8)
    enc = np.zeros((3, 3 * K + 12), np.uint8)
    for i in range(3):
        oracle().orc_tcod_encode_bits(p(bits[i]), p(enc[i]), K)
    llr = _noisy_llr(rng, enc, 1.0, 300)
    for nit in (1, 4):
        rc, out, _, _ = dec.run_all(llr, K, nit, force_subblocks=W)
        assert rc == 0
        for i in range(3):
            ref = np.zeros(K // 8, np.uint8)
            assert oracle().orc_tdec_run_w(p(llr[i]), False, K, W, nit, p(ref), None) == 0
            assert np.array_equal(out[i], ref)
    dec.free()


@pytest.mark.parametrize("K", [408, 816, 5824, 6144])
def test_tdec_sb_layout_and_early_stop(hp, K):
    """rm_turbo SB-layout input + CRC early stop as sch.c:348-383 (CB CRC24B over K bits)."""
    rng = np.random.default_rng(K + 1)
    ncb, W = 6, oracle().orc_tdec_autoimp_subblocks(K)
    dec = hp.Tdec(6144, 8)
    n_e = (3 * K * 9 // 10) // 6 * 6
    stride = 3 * (K + 32) + 12
    w = np.zeros((ncb, stride), np.int16)
    for i in range(ncb):
        payload = rng.integers(0, 256, (K - 24) // 8, dtype=np.uint8)
        crc = oracle().orc_crc_bytes(0x1800063, 24, p(payload), K - 24)
        cb = np.concatenate([payload, np.array([crc >> 16, (crc >> 8) & 255, crc & 255], np.uint8)])
        bits = np.unpackbits(cb)
        enc = np.zeros(3 * K + 12, np.uint8)
        oracle().orc_tcod_encode_bits(p(bits), p(enc), K)
        e = np.zeros(n_e, np.uint8)
        oracle().orc_rm_turbo_tx_bits(p(enc), p(e), n_e, K, 0)
        snr = [8.0, 3.0, 1.5, 1.0, 0.5, -2.0][i]
        llr = _noisy_llr(rng, e, snr, 100)
        oracle().orc_rm_turbo_rx(p(llr), p(w[i]), n_e, K, 0, W)
    rc, out, iters, ok = dec.run_all(w, K, 6, sb_layout=True, crc_poly=hp.CRC24B, crc_nbits=K)
    assert rc == 0
    for i in range(ncb):
        per = np.zeros((6, K // 8), np.uint8)
        ref = np.zeros(K // 8, np.uint8)
        assert oracle().orc_tdec_run(p(w[i]), True, K, 6, p(ref), p(per)) == 0
        n = 0
        good = False
        while n < 6 and not good:
            good = oracle().orc_crc_bytes(0x1800063, 24, p(per[n]), K) == 0
            n += 1
        assert iters[i] == n and bool(ok[i]) == good, (K, i, iters[i], n, ok[i], good)
        assert np.array_equal(out[i], per[n - 1])
    assert ok[0] == 1 and ok[-1] == 0  # the sweep must exercise both outcomes
    dec.free()


@pytest.mark.parametrize("K,ncb,sb", [(816, 5, True), (832, 19, False), (1008, 8, True), (2048, 11, True), (3136, 16, False), (5824, 29, True), (6144, 9, True),
                                       (5824, 13, False), (6144, 1, False), (4160, 2, True), (1024, 3, True)])
def test_tdec_two_blocks_per_wavefront_mapping(hp, K, ncb, sb):
    """tdec_pair_kernel (a lane = a butterfly pair of states of a window pair, two code blocks per wavefront: what AUTO runs for K > 800) against
    tdec_win_kernel<16, 0> (a lane = one state, a block per wavefront: rounds 1-2, forced with 3016 sub-blocks) and the oracle's avx16
    restatement: decoded bytes, pass counts and CRC flags of every block, the two blocks of a wavefront stopping after different numbers of
    passes, odd block counts (the last wavefront's second slot shadows the first); with and without early stop; plain [s p0 p1] and
    rate-dematcher (SB) input layouts; window lengths with every remainder mod 12 and mod 3."""
    rng = np.random.default_rng(31 * K + ncb)
    dec = hp.Tdec(6144, 32)
    stride = (3 * (K + 32) + 12) if sb else (3 * K + 12)
    w = np.zeros((ncb, stride), np.int16)
    n_e = (3 * K * 9 // 10) // 6 * 6
    for i in range(ncb):
        payload = rng.integers(0, 256, (K - 24) // 8, dtype=np.uint8)
        crc = oracle().orc_crc_bytes(0x1800063, 24, p(payload), K - 24)
        bits = np.unpackbits(np.concatenate([payload, np.array([crc >> 16, (crc >> 8) & 255, crc & 255], np.uint8)]))
        enc = np.zeros(3 * K + 12, np.uint8)
        oracle().orc_tcod_encode_bits(p(bits), p(enc), K)
        snr = (8.0, 3.0, 1.5, 1.0, 0.5, -2.0, 1.2, 0.8)[i % 8]
        if sb:
            e = np.zeros(n_e, np.uint8)
            oracle().orc_rm_turbo_tx_bits(p(enc), p(e), n_e, K, 0)
            oracle().orc_rm_turbo_rx(p(_noisy_llr(rng, e, snr, 100)), p(w[i]), n_e, K, 0, 16)
        else:
            w[i] = _noisy_llr(rng, enc, snr - 4.0, 60)
    for poly, nbits, nit in ((hp.CRC24B, K, 6), (0, 0, 3), (0, 0, 4)):
        rc, out, iters, ok = dec.run_all(w, K, nit, sb_layout=sb, crc_poly=poly, crc_nbits=nbits, force_subblocks=16)
        rc2, out2, iters2, ok2 = dec.run_all(w, K, nit, sb_layout=sb, crc_poly=poly, crc_nbits=nbits, force_subblocks=3016)
        assert rc == 0 and rc2 == 0
        assert np.array_equal(iters, iters2) and np.array_equal(ok, ok2), (iters, iters2)
        assert np.array_equal(out, out2)
        if poly:
            assert len(set(iters.tolist())) > 1 or ncb < 4
    for i in range(0, ncb, 5):  # and against the oracle
        per = np.zeros((6, K // 8), np.uint8)
        ref = np.zeros(K // 8, np.uint8)
        assert oracle().orc_tdec_run(p(w[i]), sb, K, 6, p(ref), p(per)) == 0
        rc, out, iters, ok = dec.run_all(w[i:i + 1], K, 4, sb_layout=sb)
        assert np.array_equal(out[0], per[3])
    dec.free()


def test_tdec_two_blocks_per_wavefront_every_block_length(hp):
    """Every LTE block length that takes the 16-window decoder (36.212 Table 5.1.3-3, K = 816 .. 6144: window lengths 51 .. 384, every
    remainder of the 12-step segments and of the trellis phases, every interleaver) through tdec_pair_kernel and through the state-per-lane
    kernel of rounds 1-2 on the same three noisy code words (an odd count: the last wavefront runs with a shadow slot; SNRs such that
    the blocks stop after different numbers of passes): bytes, pass counts and CRC flags identical."""
    L = hp.lib()
    sizes = [K for K in list(range(40, 512, 8)) + list(range(512, 1024, 16)) + list(range(1024, 2048, 32)) + list(range(2048, 6145, 64))
             if L.srslte_hip_tdec_autoimp_get_subblocks(K) == 16]
    assert len(sizes) > 90 and sizes[0] == 816 and sizes[-1] == 6144
    dec = hp.Tdec(6144, 4)
    spread = set()
    for K in sizes:
        rng = np.random.default_rng(K)
        w = np.zeros((3, 3 * K + 12), np.int16)
        for i in range(3):
            payload = rng.integers(0, 256, (K - 24) // 8, dtype=np.uint8)
            crc = oracle().orc_crc_bytes(0x1800063, 24, p(payload), K - 24)
            bits = np.unpackbits(np.concatenate([payload, np.array([crc >> 16, (crc >> 8) & 255, crc & 255], np.uint8)]))
            enc = np.zeros(3 * K + 12, np.uint8)
            oracle().orc_tcod_encode_bits(p(bits), p(enc), K)
            w[i] = _noisy_llr(rng, enc, (4.0, -1.0, -3.5)[i], 60)
        rc, out, iters, ok = dec.run_all(w, K, 6, sb_layout=False, crc_poly=hp.CRC24B, crc_nbits=K, force_subblocks=16)
        rc2, out2, iters2, ok2 = dec.run_all(w, K, 6, sb_layout=False, crc_poly=hp.CRC24B, crc_nbits=K, force_subblocks=3016)
        assert rc == 0 and rc2 == 0
        assert np.array_equal(iters, iters2) and np.array_equal(ok, ok2) and np.array_equal(out, out2), (K, iters, iters2)
        spread.update(iters.tolist())
    assert len(spread) >= 4  # early and late stops both occurred
    dec.free()


def test_tdec_every_block_length_both_widths_vs_oracle(hp):
    """All 188 LTE block lengths (36.212 Table 5.1.3-3) through srslte_tdec_run_all and srslte_tdec_run_all_8bit with the back-end the
    reference's AUTO selection takes for each (turbodecoder.c:421-487: generic / 8 / 16 windows in 16 bits; widening fall-back / sse8 / avx8 in 8
    bits) against the oracle decoders on the same three noisy code words and pass counts 1 .. 6 drawn per length: bytes identical."""
    sizes = list(range(40, 512, 8)) + list(range(512, 1024, 16)) + list(range(1024, 2048, 32)) + list(range(2048, 6145, 64))
    assert len(sizes) == 188
    dec = hp.Tdec(6144, 4)
    for K in sizes:
        rng = np.random.default_rng(9000 + K)
        bits = rng.integers(0, 2, (3, K)).astype(np.uint8)
        enc = np.zeros((3, 3 * K + 12), np.uint8)
        for i in range(3):
            oracle().orc_tcod_encode_bits(p(bits[i]), p(enc[i]), K)
        nit = int(rng.integers(1, 7))
        llr = _noisy_llr(rng, enc, float(rng.uniform(-3.0, 3.0)), int(rng.choice([60, 200, 1500])))
        llr8 = (int(rng.choice([10, 25, 60])) * ((2.0 * enc - 1) + 10 ** (-float(rng.uniform(-2.0, 3.0)) / 20) * rng.standard_normal(enc.shape))).clip(-128, 127).astype(np.int8)
        rc, out, _, _ = dec.run_all(llr, K, nit)
        rc8, out8, _, _ = dec.run_all(llr8, K, nit, llr8=True)
        assert rc == 0 and rc8 == 0, K
        for i in range(3):
            ref, ref8 = np.zeros(K // 8, np.uint8), np.zeros(K // 8, np.uint8)
            assert oracle().orc_tdec_run(p(llr[i]), False, K, nit, p(ref), None) == 0 and oracle().orc_tdec_run_8bit(p(llr8[i]), False, K, nit, p(ref8), None) == 0
            assert np.array_equal(out[i], ref), "16-bit K=%d nit=%d cb=%d: %d byte mismatches" % (K, nit, i, (out[i] != ref).sum())
            assert np.array_equal(out8[i], ref8), "8-bit K=%d nit=%d cb=%d: %d byte mismatches" % (K, nit, i, (out8[i] != ref8).sum())
    dec.free()


@pytest.mark.parametrize("K", [40, 408, 800, 816, 1008, 2048, 2112, 3136, 5824, 6144])
def test_tdec_run_all_8bit(hp, K):
    """srslte_tdec_run_all_8bit (turbodecoder.c:573-588): avx8 for K > 2048, sse8 for K > 800, widening fall-backs below."""
    rng = np.random.default_rng(8 * K)
    ncb = 9
    dec = hp.Tdec(6144, 16)
    bits = rng.integers(0, 2, (ncb, K)).astype(np.uint8)
    enc = np.zeros((ncb, 3 * K + 12), np.uint8)
    for i in range(ncb):
        oracle().orc_tcod_encode_bits(p(bits[i]), p(enc[i]), K)
    for snr, scale in ((1.0, 12), (3.0, 25), (0.0, 40), (-2.0, 90)):
        llr = (scale * ((2.0 * enc - 1) + 10 ** (-snr / 20) * rng.standard_normal(enc.shape))).clip(-128, 127).astype(np.int8)
        for nit in (1, 2, 3, 6):
            rc, out, _, _ = dec.run_all(llr, K, nit, llr8=True)
            assert rc == 0
            for i in range(ncb):
                ref = np.zeros(K // 8, np.uint8)
                assert oracle().orc_tdec_run_8bit(p(llr[i]), False, K, nit, p(ref), None) == 0
                assert np.array_equal(out[i], ref), "tdec8 K=%d snr=%s nit=%d cb=%d: %d byte mismatches" % (K, snr, nit, i, (out[i] != ref).sum())
    dec.free()


@pytest.mark.parametrize("K", [504, 816, 2112, 5824, 6144])
def test_tdec_8bit_sb_layout_and_early_stop(hp, K):
    """srslte_rm_turbo_rx_lut_8bit layout in, CRC early stop as sch.c:348-383 with srslte_tdec_iteration_8bit."""
    rng = np.random.default_rng(K + 8)
    ncb, W = 6, oracle().orc_tdec_autoimp_subblocks_8bit(K)
    dec = hp.Tdec(6144, 8)
    n_e = (3 * K * 9 // 10) // 6 * 6
    stride = 3 * (K + 32) + 12
    w = np.zeros((ncb, stride), np.int8)
    for i in range(ncb):
        payload = rng.integers(0, 256, (K - 24) // 8, dtype=np.uint8)
        crc = oracle().orc_crc_bytes(0x1800063, 24, p(payload), K - 24)
        cb = np.concatenate([payload, np.array([crc >> 16, (crc >> 8) & 255, crc & 255], np.uint8)])
        enc = np.zeros(3 * K + 12, np.uint8)
        oracle().orc_tcod_encode_bits(p(np.unpackbits(cb)), p(enc), K)
        e = np.zeros(n_e, np.uint8)
        oracle().orc_rm_turbo_tx_bits(p(enc), p(e), n_e, K, 0)
        snr = [8.0, 3.0, 1.5, 1.0, 0.5, -2.0][i]
        llr = (20 * ((2.0 * e - 1) + 10 ** (-snr / 20) * rng.standard_normal(n_e))).clip(-128, 127).astype(np.int8)
        oracle().orc_rm_turbo_rx_8bit(p(llr), p(w[i]), n_e, K, 0, W)
    rc, out, iters, ok = dec.run_all(w, K, 6, sb_layout=True, crc_poly=hp.CRC24B, crc_nbits=K, llr8=True)
    assert rc == 0
    for i in range(ncb):
        per = np.zeros((6, K // 8), np.uint8)
        assert oracle().orc_tdec_run_8bit(p(w[i]), True, K, 6, None, p(per)) == 0
        n, good = 0, False
        while n < 6 and not good:
            good = oracle().orc_crc_bytes(0x1800063, 24, p(per[n]), K) == 0
            n += 1
        assert iters[i] == n and bool(ok[i]) == good, (K, i, iters[i], n, ok[i], good)
        assert np.array_equal(out[i], per[n - 1])
    assert ok[0] == 1 and ok[-1] == 0
    dec.free()


def test_tdec_sse8_two_blocks_per_wavefront_every_block_length(hp):
    """tdec_ar16_kernel - the 8-bit API's 16-window back-end (sse8 numerics, 800 < K <= 2048) with two code blocks per wavefront around the
    pair-mapped sweeps - on every block length it serves, against the oracle's sse8 restatement: three blocks (an odd count: the last wavefront
    has an empty slot) at SNRs that make them stop after different numbers of passes (a wavefront's slots stop at different times), rate-dematcher
    (SB) and plain layouts alternating; bytes of the stopping pass, pass counts and CRC flags of every block."""
    L = hp.lib()
    sizes = [K for K in list(range(40, 512, 8)) + list(range(512, 1024, 16)) + list(range(1024, 2048, 32)) + list(range(2048, 6145, 64))
             if L.srslte_hip_tdec_autoimp_get_subblocks_8bit(K) == 16]
    assert len(sizes) > 40 and sizes[0] == 816 and sizes[-1] == 2048
    dec = hp.Tdec(2048, 4)
    spread = set()
    for n_, K in enumerate(sizes):
        rng = np.random.default_rng(K + 1)
        sb = bool(n_ & 1)
        n_e = (3 * K * 9 // 10) // 6 * 6
        w = np.zeros((3, (3 * (K + 32) + 12) if sb else (3 * K + 12)), np.int8)
        for i in range(3):
            payload = rng.integers(0, 256, (K - 24) // 8, dtype=np.uint8)
            crc = oracle().orc_crc_bytes(0x1800063, 24, p(payload), K - 24)
            bits = np.unpackbits(np.concatenate([payload, np.array([crc >> 16, (crc >> 8) & 255, crc & 255], np.uint8)]))
            enc = np.zeros(3 * K + 12, np.uint8)
            oracle().orc_tcod_encode_bits(p(bits), p(enc), K)
            snr = (6.0, 1.0, -3.0)[(i + n_) % 3]
            if sb:
                e = np.zeros(n_e, np.uint8)
                oracle().orc_rm_turbo_tx_bits(p(enc), p(e), n_e, K, 0)
                llr = (20 * ((2.0 * e - 1) + 10 ** (-(snr + 1.0) / 20) * rng.standard_normal(n_e))).clip(-128, 127).astype(np.int8)
                oracle().orc_rm_turbo_rx_8bit(p(llr), p(w[i]), n_e, K, 0, 16)
            else:
                w[i] = (20 * ((2.0 * enc - 1) + 10 ** (-(snr - 3.0) / 20) * rng.standard_normal(3 * K + 12))).clip(-128, 127).astype(np.int8)
        rc, out, iters, ok = dec.run_all(w, K, 6, sb_layout=sb, crc_poly=hp.CRC24B, crc_nbits=K, llr8=True)
        assert rc == 0
        for i in range(3):
            per = np.zeros((6, K // 8), np.uint8)
            assert oracle().orc_tdec_run_8bit(p(w[i]), sb, K, 6, None, p(per)) == 0
            n, good = 0, False
            while n < 6 and not good:
                good = oracle().orc_crc_bytes(0x1800063, 24, p(per[n]), K) == 0
                n += 1
            assert iters[i] == n and bool(ok[i]) == good, (K, i, iters[i], n, ok[i], good)
            assert np.array_equal(out[i], per[n - 1]), (K, i)
        spread.update(iters.tolist())
        # no CRC: a fixed number of passes for every block, two blocks (both slots run to the end together) and one (a lone slot)
        for cnt, nit in ((2, 3), (1, 2)):
            rc, out, iters, ok = dec.run_all(w[:cnt], K, nit, sb_layout=sb, llr8=True)
            assert rc == 0 and (iters == nit).all()
            for i in range(cnt):
                per = np.zeros((6, K // 8), np.uint8)
                assert oracle().orc_tdec_run_8bit(p(w[i]), sb, K, 6, None, p(per)) == 0
                assert np.array_equal(out[i], per[nit - 1]), (K, i, nit)
    assert len(spread) >= 4
    dec.free()


def test_tdec_errors(hp):
    dec = hp.Tdec(1024, 4)
    rc, *_ = dec.run_all(np.zeros((1, 3 * 2048 + 12), np.int16), 2048, 1)
    assert rc == hp.SRSLTE_ERROR  # turbodecoder.c:524-527: exceeds max_long_cb
    rc, *_ = dec.run_all(np.zeros((1, 3 * 41 + 12), np.int16), 41, 1)
    assert rc == hp.SRSLTE_ERROR  # turbodecoder.c:531-534: invalid CB length
    dec.free()


def test_cbsegm_and_interleaver(hp):
    for tbs in (16, 152, 936, 6120, 6144, 6200, 75376, 97896):
        rc, s = hp.cbsegm(tbs)
        from _libs import OrcCbsegm
        r = OrcCbsegm()
        assert oracle().orc_cbsegm(C.byref(r), tbs) == rc == 0
        assert all(getattr(s, f) == getattr(r, f) for f, _ in OrcCbsegm._fields_)
    for K, W in ((40, 1), (504, 8), (5824, 16), (6144, 16), (6144, 1)):
        rc, f, r = hp.tc_interl(K, W)
        rf, rr = np.zeros(K, np.uint16), np.zeros(K, np.uint16)
        assert oracle().orc_qpp(K, W, p(rf), p(rr)) == rc == 0
        assert np.array_equal(f, rf) and np.array_equal(r, rr)
    assert hp.tc_interl(41, 1)[0] == hp.SRSLTE_ERROR


# ---------------------------------------------------------------- end to end
@pytest.mark.parametrize("prb,mod,tbs,snr,tti0,nsf", [(6, 1, 936, 12.0, 1, 4), (6, 1, 936, 2.5, 1, 4), (100, 3, 75376, 30.0, 8, 4),
                                                       (100, 3, 75376, 19.0, 9, 3), (100, 4, 97896, 35.0, 4, 3),
                                                       (25, 2, 6200, 11.0, 3, 4)])  # 6200: not a 36.213 table size; 2 x 3136, no filler (cbsegm.c:77-107)
def test_dl_rx_chain(hp, prb, mod, tbs, snr, tti0, nsf):
    """IQ -> TB on the device vs the oracle chain on identical IQ: TB bytes, CRC flags and per-CB iteration counts equal."""
    rng = np.random.default_rng(prb + mod + int(snr * 10))
    cfg = DlConfig(prb, 1, mod, tbs)
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.05 / np.sqrt(prb) * 20) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc)
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    max_re = max(rx.nof_re(s) for s in (0, 1, 5))
    grid = rx.debug(0, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    ce = rx.debug(1, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    e_all = rx.debug(4, np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    n_diff = n_tot = 0
    for b in range(nsf):
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        nre = rx.nof_re((tti0 + b) % 10)
        assert nre == len(r["d"])
        assert_close_c(grid[b], r["grid"], "grid sf %d" % b)
        assert_close_c(ce[b], r["ce"], "ce sf %d" % b)
        e = e_all[b, :nre * cfg.Qm].astype(np.int32)
        diff = np.abs(e - r["e"].astype(np.int32))
        assert diff.max() <= 1, "LLR differs by more than 1 LSB (sf %d: %d)" % (b, diff.max())
        n_diff += int((diff != 0).sum())
        n_tot += diff.size
        assert bool(ok[b]) == r["ok"], "tb_ok sf %d" % b
        assert np.array_equal(it[b], r["iters"]), "iterations sf %d: %s vs %s" % (b, it[b], r["iters"])
        assert np.array_equal(tb[b], r["tb"]), "TB bytes sf %d" % b
        if r["ok"]:
            assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_diff <= 1e-3 * n_tot, "LLR LSB differences on %d of %d" % (n_diff, n_tot)
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,snr,tti0,nsf,interp,nrx,llr8", [(6, 1, 152, 6.0, 9, 4, False, 1, False), (25, 2, 4008, 11.0, 4, 4, False, 1, False),
                                                                       (100, 3, 43816, 15.0, 8, 4, False, 1, False), (100, 3, 43816, 15.5, 3, 3, True, 1, False),
                                                                       (50, 3, 11448, 8.5, 0, 6, False, 2, False), (25, 2, 4008, 11.5, 5, 4, True, 1, True)])
def test_dl_rx_chain_extended_cp(hp, prb, mod, tbs, snr, tti0, nsf, interp, nrx, llr8):
    """Extended-CP cells in the fused receive pipeline (cfg.cp_ext; VERDICT r3 missing item 3): 12 symbols per subframe - grids, estimates and RE
    lists [12][12 nof_prb] -, CRS on symbols 0 and 3 of each slot, PSS / SSS on symbols 5 and 4 of slot 0. IQ -> TB on the device against the
    oracle chain on identical IQ (pinned to the reference's srslte_pdsch_decode on an extended-CP cell by
    tests/test_oracle_vs_ref.py::test_pdsch_decode_extended_cp_vs_oracle_chain): grid, estimates, LLRs within one LSB, CRC flags, pass counts
    and transport blocks equal; subframes with PSS / SSS / PBCH among them, with and without interpolate_subframe (chest_dl.c:497-502)."""
    rng = np.random.default_rng(7000 + prb + mod + int(snr * 10))
    cfg = DlConfig(prb, 11, mod, tbs, nof_rx=nrx, llr8=llr8, cp_ext=True, chest={"filter_coef": (4.0, 1.0), "interpolate_subframe": interp})
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.05 / np.sqrt(prb) * 20) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1], hc.interpolate_subframe = 4.0, 1.0, 1 if interp else 0
    rx = hp.DlRx(11, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, llr_8bit=llr8, nof_rx=nrx, cp_ext=True)
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    grid = rx.debug(0, np.complex64, nsf * nrx * cfg.grid_len).reshape(nsf, nrx, -1)
    ce = rx.debug(1, np.complex64, nsf * nrx * cfg.grid_len).reshape(nsf, nrx, -1)
    e_all = rx.debug(4, np.int8 if llr8 else np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    n_diff = n_tot = n_ok = 0
    for b in range(nsf):
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        nre = rx.nof_re((tti0 + b) % 10)
        assert nre == len(r["d"]) == len(cfg.indices((tti0 + b) % 10))
        assert_close_c(grid[b].ravel(), np.asarray(r["grid"]).ravel(), "grid sf %d" % b)
        assert_close_c(ce[b].ravel(), np.asarray(r["ce"]).ravel(), "ce sf %d" % b)
        diff = np.abs(e_all[b, :nre * cfg.Qm].astype(np.int32) - r["e"].astype(np.int32))
        assert diff.max() <= 1, "LLR differs by more than 1 LSB (sf %d: %d)" % (b, diff.max())
        n_diff += int((diff != 0).sum())
        n_tot += diff.size
        assert bool(ok[b]) == r["ok"], "tb_ok sf %d" % b
        assert np.array_equal(it[b], r["iters"]), "iterations sf %d: %s vs %s" % (b, it[b], r["iters"])
        assert np.array_equal(tb[b], r["tb"]), "TB bytes sf %d" % b
        if r["ok"]:
            assert np.array_equal(tb[b][:tbs // 8], data[b])
            n_ok += 1
    assert n_diff <= 2e-3 * n_tot and n_ok > 0, (n_diff, n_tot, n_ok)
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,snr,tti0,nsf", [(6, 1, 936, 12.0, 1, 4), (6, 1, 936, 4.0, 1, 4), (100, 3, 75376, 30.0, 8, 4),
                                                       (100, 3, 75376, 19.5, 9, 3), (100, 4, 97896, 35.0, 4, 3)])
def test_dl_rx_chain_8bit(hp, prb, mod, tbs, snr, tti0, nsf):
    """8-bit LLR path (pdsch.c:760-779, sch.c:336-356) on the device vs the oracle's 8-bit chain."""
    rng = np.random.default_rng(80 + prb + mod + int(snr * 10))
    cfg = DlConfig(prb, 1, mod, tbs, llr8=True)
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.05 / np.sqrt(prb) * 20) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, llr_8bit=True)
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    max_re = max(rx.nof_re(s) for s in (0, 1, 5))
    e_all = rx.debug(4, np.int8, nsf * rx.e_stride).reshape(nsf, -1)
    n_diff = n_tot = n_ok = 0
    for b in range(nsf):
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        nre = rx.nof_re((tti0 + b) % 10)
        diff = np.abs(e_all[b, :nre * cfg.Qm].astype(np.int32) - r["e"].astype(np.int32))
        assert diff.max() <= 1, "LLR differs by more than 1 LSB (sf %d: %d)" % (b, diff.max())
        n_diff += int((diff != 0).sum())
        n_tot += diff.size
        assert bool(ok[b]) == r["ok"], "tb_ok sf %d" % b
        assert np.array_equal(it[b], r["iters"]), "iterations sf %d: %s vs %s" % (b, it[b], r["iters"])
        assert np.array_equal(tb[b], r["tb"]), "TB bytes sf %d" % b
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_diff <= 1e-3 * n_tot, "LLR LSB differences on %d of %d" % (n_diff, n_tot)
    assert n_ok > 0 or snr < 10
    rx.free()


def test_smallest_and_largest_transport_blocks(hp):
    """The ends of the size range through the fused receive pipelines, both directions: transport blocks of 16 / 24 / 40 bits (one code block
    of K = 40 / 48 / 64: the generic decoder, and for 8-bit LLRs the widening fall-back of turbodecoder.c:438-469) on a 6-PRB cell, and
    the largest a 110-PRB cell carries at 256QAM (117 256 bits in 20 code blocks; uplink 83 864 bits in 14) - CRC flags, pass counts and
    the bytes of delivered blocks equal the oracle chain's on identical samples."""
    from _libs import OrcCbsegm
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(1)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0

    def check(rx, cfg, iq, tti0, orc):
        tb, ok = rx.decode(np.stack(iq), tti0)
        it = rx.debug(6, np.uint32, len(iq) * cfg.seg.C).reshape(len(iq), -1)
        for b in range(len(iq)):
            r = orc(cfg, iq[b], tti0 + b)
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), (cfg.tbs, b, ok[b], r["ok"], it[b], r["iters"])
            if r["ok"]:
                assert np.array_equal(tb[b], r["tb"]), (cfg.tbs, b)
        rx.free()
        return int(ok.sum())

    n_ok = 0
    for tbs in (16, 24, 40):
        for llr8 in (False, True):
            cfg = DlConfig(6, 9, 1, tbs, llr8=llr8)
            iq = [make_subframe(cfg, 3 + b, rng, snr_db=-2.0, amp=0.1)[0] for b in range(3)]
            n_ok += check(hp.DlRx(9, 6, 1, 0x1234, 1, tbs, 6, 3, True, hc, llr_8bit=llr8), cfg, iq, 3, oracle_rx)
        cfg = UlConfig(6, 9, 1, tbs, 1, 2, n_dmrs=1)
        iq = [make_ul_subframe(cfg, 3 + b, rng, snr_db=3.0, amp=0.1)[0] for b in range(3)]
        n_ok += check(hp.UlRx(9, 6, 0x1234, 1, tbs, 1, 2, 1, 6, 3), cfg, iq, 3, oracle_ul_rx)

    def largest(nbits, rate):
        tbs = int(rate * nbits) // 8 * 8
        while True:
            seg = OrcCbsegm()
            if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
                return tbs
            tbs -= 8

    probe = DlConfig(110, 5, 4, 16)
    tbs = largest(min(len(probe.indices(sf)) for sf in (0, 1, 5)) * probe.Qm, 0.92)
    for llr8 in (False, True):
        cfg = DlConfig(110, 5, 4, tbs, llr8=llr8)
        assert cfg.seg.C == 20
        iq = [make_subframe(cfg, 1 + b, rng, snr_db=34.0, amp=0.1)[0] for b in range(2)]
        n_ok += check(hp.DlRx(5, 110, 1, 0x1234, 4, tbs, 6, 2, True, hc, llr_8bit=llr8), cfg, iq, 1, oracle_rx)
    tbs = largest(UlConfig(110, 5, 3, 16, 108, 0).nbits, 0.9)
    cfg = UlConfig(110, 5, 3, tbs, 108, 0, n_dmrs=1)
    iq = [make_ul_subframe(cfg, 1 + b, rng, snr_db=30.0, amp=0.1)[0] for b in range(2)]
    n_ok += check(hp.UlRx(5, 110, 0x1234, 3, tbs, 108, 0, 1, 6, 2), cfg, iq, 1, oracle_ul_rx)
    assert n_ok >= 25


@pytest.mark.parametrize("prb", [7, 20, 33, 64, 91, 110])
def test_dl_rx_chain_any_bandwidth(hp, prb):
    """srslte_cell_isvalid takes any 6 .. 110 PRB (phy_common.c:43-52), not only the six of 36.101: the drawn-configuration test at bandwidths
    between them (symbol size of the next one up, odd and even carrier counts, the half-PRB rule of srslte_pdsch_cp for odd counts)."""
    test_dl_rx_chain_drawn_configurations(hp, 100 + prb, prb)


@pytest.mark.parametrize("seed", range(6))
def test_dl_rx_chain_drawn_configurations_256qam(hp, seed):
    """The drawn-configuration test with 256QAM (srslte_mod_t 4; the 8-bit demapper's and the 16-bit one's 256QAM branches, Qm = 8)."""
    test_dl_rx_chain_drawn_configurations(hp, 200 + seed, None, 4)


@pytest.mark.parametrize("seed", range(12))
def test_dl_rx_chain_drawn_configurations(hp, seed, force_prb=None, force_mod=None):
    """test_dl_rx_chain on configurations DRAWN from the space the pipeline accepts instead of listed: bandwidth, cell id, RNTI, CFI-independent
    modulation, a transport-block size that is not taken from a table (any multiple of 8 that segments into one block length without filler,
    cbsegm.c:77-107) at a drawn code rate, a drawn first TTI and an SNR a few dB either side of the waterfall; 16- and 8-bit LLRs, one and
    two receive antennas. TB bytes, CRC flags and per-block pass counts equal the oracle chain's on identical samples."""
    from _libs import OrcCbsegm, OrcSchCfg
    rng = np.random.default_rng(7000 + seed)
    prb = int(rng.choice([6, 15, 25, 50]))
    if force_prb:
        prb = force_prb
    mod = force_mod or int(rng.choice([1, 2, 3]))
    llr8, nrx = bool(seed % 3 == 2), 1 + int(seed % 4 == 1)
    cell_id, rnti = int(rng.integers(0, 504)), int(rng.integers(1, 0xFFF0))
    probe = DlConfig(prb, cell_id, mod, 16, rnti=rnti)
    nbits = min(len(probe.indices(s)) for s in (0, 1, 5)) * probe.Qm
    rate = float(rng.uniform(0.25, 0.8))
    tbs = max(40, int(rate * nbits) // 8 * 8)
    while True:  # the next size below that needs neither filler bits nor two block lengths
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs -= 8
    cfg = DlConfig(prb, cell_id, mod, tbs, rnti=rnti, nof_rx=nrx, llr8=llr8)
    tti0, nsf = int(rng.integers(0, 10240)), 3
    # rough waterfall of a rate-r code at this modulation, +- a few dB: some blocks fail, some pass, either is fine - equality is the test
    snr = {1: 1.0, 2: 7.0, 3: 12.0, 4: 18.0}[mod] + 10.0 * (tbs / nbits - 0.4) + float(rng.uniform(-2.0, 4.0)) - (3.0 if nrx == 2 else 0.0)
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(cell_id, prb, 1, rnti, mod, tbs, 6, nsf, True, hc, llr_8bit=llr8, nof_rx=nrx)
    tb, ok = rx.decode(np.stack(iq), tti0)
    it = rx.debug(6, np.uint32, nsf * cfg.seg.C).reshape(nsf, -1)
    e_all = rx.debug(4, np.int8 if llr8 else np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    n_diff = n_tot = 0
    for b in range(nsf):
        what = (prb, mod, tbs, nrx, llr8, tti0 + b, snr)
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        # floating-point part (OFDM, estimator, equaliser, demapper): LLRs within one LSB of the oracle's, on at most one in a thousand
        nb = rx.nof_re((tti0 + b) % 10) * cfg.Qm
        e = np.ascontiguousarray(e_all[b, :nb])
        diff = np.abs(e.astype(np.int32) - r["e"].astype(np.int32))
        assert len(r["e"]) == nb and diff.max() <= 1, what
        n_diff += int((diff != 0).sum())
        n_tot += nb
        # integer part (rate de-matching, turbo decoder, CRCs, assembly): the oracle's back end on the DEVICE's LLRs - exact, also for a
        # transport block that fails (one LSB on one LLR changes the bytes of a block that does not converge)
        sch = OrcSchCfg(tbs, nb, cfg.Qm_sch, 0, cfg.max_iter)
        otb, oit, ocb = np.zeros(tbs // 8 + 16, np.uint8), np.zeros(cfg.seg.C, np.uint32), np.zeros(cfg.seg.C, np.uint8)
        rc = (oracle().orc_dlsch_decode_8bit if llr8 else oracle().orc_dlsch_decode)(C.byref(sch), p(e), p(otb), p(oit), p(ocb))
        assert bool(ok[b]) == (rc == 0) and np.array_equal(it[b], oit) and np.array_equal(tb[b], otb[:tbs // 8 + 3]), what
        if ok[b]:
            assert np.array_equal(tb[b][:tbs // 8], data[b]), what
        if not diff.any():  # identical LLRs: the oracle chain end to end says the same
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]) and np.array_equal(tb[b], r["tb"]), what
    assert n_diff <= 1e-3 * n_tot, "LLR LSB differences on %d of %d" % (n_diff, n_tot)
    rx.free()


def test_cfg4_multi_ue(hp):
    """several UEs, each with its own cell id and RNTI (CRS position/sequence, scrambling, RE map differ)."""
    for u, (prb, mod, tbs, snr) in enumerate([(100, 3, 75376, 24.0), (100, 3, 75376, 21.0), (6, 1, 936, 6.0), (25, 2, 11448, 16.0)]):
        cell_id, rnti = 1 + 5 * u + (u == 3) * 97, 0x1234 + u
        rng = np.random.default_rng(400 + u)
        cfg = DlConfig(prb, cell_id, mod, tbs, rnti=rnti)
        ttis = [0, 5, 6] if prb != 6 else [1, 6, 9]
        iq, data = zip(*[make_subframe(cfg, t, rng, snr_db=snr, amp=0.1) for t in ttis])
        hc = hp.ChestDlCfg()
        hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
        rx = hp.DlRx(cell_id, prb, 1, rnti, mod, tbs, 6, 1, True, hc)
        for b, t in enumerate(ttis):
            tb, ok = rx.decode(iq[b][None, :], t)
            r = oracle_rx(cfg, iq[b], t)
            assert bool(ok[0]) == r["ok"] and np.array_equal(tb[0], r["tb"]), (u, t)
            assert np.array_equal(rx.debug(6, np.uint32, cfg.seg.C), r["iters"])
            assert r["ok"] and np.array_equal(tb[0][:tbs // 8], data[b])
        rx.free()


@pytest.mark.parametrize("llr8", [False, True])
def test_cfg5_256qam_grid_snr_sweep(hp, llr8):
    """100 PRB 256QAM (TBS 97896, 16 x K=6144) from frequency-domain grids through the chest_test_dl channel,
    SNR sweep; ce / noise / LLR parity and identical block decisions (so identical BLER) at every point."""
    from lte_sim import make_grid
    prb, mod, tbs = 100, 4, 97896
    cfg = DlConfig(prb, 1, mod, tbs, llr8=llr8)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, 2, True, hc, llr_8bit=llr8)
    outcomes = []
    for si, snr in enumerate((10, 15, 20, 25, 30, 35)):
        rng = np.random.default_rng(500 + si)
        ttis = (3, 4)
        grids, data = zip(*[make_grid(cfg, t, rng, snr) for t in ttis])
        tb, ok = rx.decode_grid(np.stack(grids), ttis[0])
        ce = rx.debug(1, np.complex64, 2 * cfg.grid_len).reshape(2, -1)
        res = rx.debug(2, np.float32, 2 * 10).reshape(2, 10)
        it = rx.debug(6, np.uint32, 2 * cfg.seg.C).reshape(2, -1)
        max_re = max(rx.nof_re(s_) for s_ in (0, 1, 5))
        e_all = rx.debug(4, np.int8 if llr8 else np.int16, 2 * rx.e_stride).reshape(2, -1)
        for b, t in enumerate(ttis):
            r = oracle_rx(cfg, None, t, keep=True, grid_in=grids[b])
            assert_close_c(ce[b], r["ce"], "ce snr %d sf %d" % (snr, b))
            assert abs(res[b, 0] - r["noise"]) <= 1e-4 * abs(r["noise"])
            nre = rx.nof_re(t % 10)
            e = e_all[b, :nre * cfg.Qm].astype(np.int32)
            diff = np.abs(e - r["e"].astype(np.int32))
            assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), (snr, b)
            # A block that never converges amplifies a 1-LSB LLR difference (float chest upstream) into different garbage:
            # byte equality is required where the LLRs are identical or the block decoded
            if r["ok"] or diff.max() == 0:
                assert np.array_equal(tb[b], r["tb"]), (snr, b)
            outcomes.append((snr, r["ok"]))
            if r["ok"]:
                assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert not outcomes[0][1] and outcomes[-1][1]  # the sweep crosses the waterfall
    rx.free()


def test_cfg3_ul_tx_chain(hp):
    """SURVEY §8d cfg3 (UL transmit, 100 PRB 16QAM): TB -> CB CRC -> turbo encode (device, byte API) -> rate matching + modulation
    (host, out of scope: oracle) -> 12 x 1200-point transform precoding (device) -> SC-FDMA grid -> OFDM TX with the half-carrier
    shift and 1/sqrt(N) (device); every device stage and the final time signal against the oracle chain."""
    from _libs import OrcOfdm, OrcSchCfg
    rng = np.random.default_rng(33)
    prb, Qm, mod, tbs = 100, 4, 2, 43816  # 16QAM, I_TBS 20 at 100 PRB: 8 x K = 5504
    rc, seg = hp.cbsegm(tbs)
    assert rc == 0 and seg.C == 8 and seg.K1 == 5504 and seg.C2 == 0
    K, Cn = seg.K1, seg.C
    nof_re = 12 * 12 * prb
    nbits = nof_re * Qm
    data = rng.integers(0, 256, tbs // 8, dtype=np.uint8)
    # oracle chain (orc_dlsch_encode is the UL-SCH data path without UCI: same segmentation, coder and rate matching, sch.c:580-650)
    sch = OrcSchCfg(tbs, nbits, Qm, 0, 6)
    e_ref = np.zeros(nbits, np.uint8)
    assert oracle().orc_dlsch_encode(C.byref(sch), p(data), p(e_ref)) == 0
    # device encoder on the same code blocks
    crc = oracle().orc_crc_bytes(0x1864CFB, 24, p(data), tbs)
    tbb = np.concatenate([data, np.array([crc >> 16, (crc >> 8) & 255, crc & 255], np.uint8)])
    rlen = K - 24
    cbs = np.zeros((Cn, K // 8), np.uint8)
    for i in range(Cn):
        body = tbb[i * rlen // 8:(i + 1) * rlen // 8]
        c = oracle().orc_crc_bytes(0x1800063, 24, p(np.ascontiguousarray(body)), rlen)
        cbs[i] = np.concatenate([body, np.array([c >> 16, (c >> 8) & 255, c & 255], np.uint8)])
    L = hp.lib()
    din, dpar, dtail = hp.DevBuf.from_host(cbs), hp.DevBuf(Cn * (K // 4 + 1)), hp.DevBuf(Cn)
    assert L.srslte_hip_tcod_encode_bytes_batch(din.ptr, K // 8, dpar.ptr, K // 4 + 1, dtail.ptr, K, Cn, None) == 0
    hp.sync()
    par, tail = dpar.to_host(np.uint8).reshape(Cn, -1), dtail.to_host(np.uint8)
    e_dev, wp = np.zeros(nbits, np.uint8), 0
    Gp = nbits // Qm
    gamma = Gp % Cn
    for i in range(Cn):
        sysb = np.concatenate([cbs[i], tail[i:i + 1]])
        ref_sys, ref_par = sysb.copy(), np.zeros(K // 4 + 2, np.uint8)
        oracle().orc_tcod_encode_bytes(p(ref_sys), p(ref_par), K)
        assert np.array_equal(ref_sys, sysb) and np.array_equal(ref_par[:K // 4 + 1], par[i])
        pb = np.unpackbits(par[i])
        d = np.zeros(3 * K + 12, np.uint8)
        sb = np.unpackbits(sysb)
        d[0:3 * K:3], d[1:3 * K:3], d[2:3 * K:3] = sb[:K], pb[:K], pb[K + 4:2 * K + 4]
        t0, t1, t2 = sb[K:K + 4], pb[K:K + 4], pb[2 * K + 4:2 * K + 8]
        for j in range(4):
            d[3 * K + 3 * j: 3 * K + 3 * j + 3] = (t0[j], t1[j], t2[j])
        n_e = Qm * (Gp // Cn) if i <= Cn - gamma - 1 else Qm * -(-Gp // Cn)
        ee = np.zeros(n_e, np.uint8)
        oracle().orc_rm_turbo_tx_bits(p(d), p(ee), n_e, K, 0)
        e_dev[wp:wp + n_e] = ee
        wp += n_e
    assert wp == nbits and np.array_equal(e_dev, e_ref)
    syms = np.zeros(nof_re, np.complex64)
    oracle().orc_modulate(mod, p(e_ref), p(syms), nbits)
    rc, z = hp.dft_precoding(syms, prb, 12, True)
    z_ref = np.zeros_like(syms)
    oracle().orc_dft_precoding(p(syms), p(z_ref), prb, 12, 1, True)
    assert rc == 0
    assert_close_c(z, z_ref, "transform precoding")
    grid = np.zeros(14 * 12 * prb, np.complex64)
    data_syms = [l for l in range(14) if l not in (3, 10)]  # DMRS symbols left empty here (refsignal_ul is §8f N3)
    for j, l in enumerate(data_syms):
        grid[l * 12 * prb:(l + 1) * 12 * prb] = z_ref[j * 12 * prb:(j + 1) * 12 * prb]
    tx = hp.Ofdm(prb, True, False)
    tx.set_normalize(True)
    tx.set_freq_shift(0.5)
    t = tx.tx_sf(grid)
    q = OrcOfdm()
    oracle().orc_ofdm_init(C.byref(q), prb, True)
    q.normalize, q.freq_shift, q.freq_shift_f, q.exact = True, True, 0.5, True
    t_ref = np.zeros(15 * 1536, np.complex64)
    oracle().orc_ofdm_tx_sf(C.byref(q), p(grid), p(t_ref))
    assert_close_c(t.ravel(), t_ref, "SC-FDMA time signal")


@pytest.mark.parametrize("prb,mod,tbs,snr,tti0,nsf,llr8", [(6, 1, 936, 0.0, 1, 4, False), (100, 3, 75376, 17.5, 9, 3, False), (100, 3, 75376, 19.0, 4, 2, True)])
def test_dl_rx_chain_two_rx_antennas(hp, prb, mod, tbs, snr, tti0, nsf, llr8):
    """SURVEY §8f N4, two receive antennas: per-antenna chest_dl, antenna-averaged noise, srslte_predecoding_single_multi (MRC), then the
    usual chain - vs the oracle on identical IQ [nsf][2][sf_len]."""
    rng = np.random.default_rng(700 + prb + mod + int(snr * 10))
    cfg = DlConfig(prb, 1, mod, tbs, nof_rx=2, llr8=llr8)
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.05 / np.sqrt(prb) * 20) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, llr_8bit=llr8, nof_rx=2)
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    max_re = max(rx.nof_re(s) for s in (0, 1, 5))
    ce = rx.debug(1, np.complex64, nsf * 2 * cfg.grid_len).reshape(nsf, 2, -1)
    res = rx.debug(2, np.float32, nsf * 10).reshape(nsf, 10)
    e_all = rx.debug(4, np.int8 if llr8 else np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    n_ok = 0
    for b in range(nsf):
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        nre = rx.nof_re((tti0 + b) % 10)
        for a in range(2):
            assert_close_c(ce[b, a], r["ce"][a], "ce sf %d antenna %d" % (b, a))
        for j, nm in enumerate(("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm")):
            x = getattr(r["res"], nm)
            assert abs(res[b, j] - x) <= 1e-4 * abs(x) + 1e-5, (nm, res[b, j], x)
        diff = np.abs(e_all[b, :nre * cfg.Qm].astype(np.int32) - r["e"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size
        assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
        if r["ok"] or diff.max() == 0:
            assert np.array_equal(tb[b], r["tb"])
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_ok > 0
    rx.free()


@pytest.mark.parametrize("cell_id,prb,L,n_prb", [(3, 6, 6, 0), (150, 50, 45, 2), (1, 100, 100, 0), (9, 100, 3, 60)])
def test_chest_ul_pusch_batch(hp, cell_id, prb, L, n_prb):
    """srslte_chest_ul_estimate_pusch on a batch of subframes vs the oracle: DMRS of every subframe index, ce, noise, SNR."""
    from _libs import OrcChestUlRes, OrcUlDmrs, OrcUlDmrsCfg
    rng = np.random.default_rng(cell_id + prb + L)
    cs, ds, gh, sh, n_dmrs, tti0, nsf = 4, 11, True, L >= 6, 2, 7, 5
    q = hp.ChestUl(cell_id, prb, cs, ds, gh, sh)
    o, cfg = OrcUlDmrs(), OrcUlDmrsCfg(cs, ds, gh, sh)
    oracle().orc_ul_dmrs_init(C.byref(o), cell_id)
    nre, ng = 12 * prb, 14 * 12 * prb
    grids, refs = np.zeros((nsf, ng), np.complex64), []
    for b in range(nsf):
        r = np.zeros(2 * 12 * L, np.complex64)
        assert oracle().orc_ul_dmrs_pusch_gen(C.byref(o), C.byref(cfg), L, (tti0 + b) % 10, n_dmrs, p(r)) == 0
        rc, r_dev = q.dmrs(L, (tti0 + b) % 10, n_dmrs)
        assert rc == 0 and np.array_equal(r, r_dev)  # same host arithmetic: identical floats
        g = (0.5 * (rng.standard_normal(ng) + 1j * rng.standard_normal(ng))).astype(np.complex64)
        k = np.arange(12 * L)
        h = ((1.0 + 0.5 * np.cos(k / 25.0 + b)) * np.exp(1j * (b + k / 120.0))).astype(np.complex64)
        for s_, sym in enumerate((3, 10)):
            g[sym * nre + 12 * n_prb: sym * nre + 12 * (n_prb + L)] = r[s_ * 12 * L:(s_ + 1) * 12 * L] * h
        grids[b] = g + (0.02 + 0.05 * b) * (rng.standard_normal(ng) + 1j * rng.standard_normal(ng))
        refs.append(r)
    rc, ce, res = q.estimate_pusch(grids, tti0, L, n_prb, n_dmrs)
    assert rc == 0
    for b in range(nsf):
        ce_o, ores = np.zeros(ng, np.complex64), OrcChestUlRes()
        assert oracle().orc_chest_ul_pusch(p(refs[b]), prb, L, n_prb, p(np.ascontiguousarray(grids[b])), p(ce_o), C.byref(ores)) == 0
        assert_close_c(ce[b], ce_o, "ce sf %d" % b)
        for j, nm in enumerate(("noise_estimate", "noise_estimate_dbm", "snr", "snr_db")):
            x = getattr(ores, nm)
            assert abs(res[b, j] - x) <= 1e-4 * abs(x) + 1e-5, (nm, res[b, j], x)
    assert q.estimate_pusch(grids, tti0, 7, 0, 0)[0] == hp.SRSLTE_ERROR_INVALID_INPUTS  # 7 PRB is not a valid SC-FDMA size (chest_ul.c:278-281)
    q.free()


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,snr,tti0,nsf", [(6, 6, 0, 1, 1000, 3.5, 2, 4), (25, 10, 5, 2, 4008, 9.5, 8, 4), (100, 100, 0, 2, 43816, 12.5, 0, 3),
                                                                (100, 48, 20, 3, 30576, 17.0, 7, 3), (100, 100, 0, 2, 43816, 9.0, 5, 2),
                                                                (25, 1, 7, 1, 104, 4.0, 3, 6), (50, 2, 31, 2, 328, 10.0, 0, 4),
                                                                (50, 30, 4, 2, 6200, 6.0, 1, 3)])  # a non-table size without filler bits
@pytest.mark.parametrize("short", [False, True])
def test_ul_rx_chain(hp, prb, L, n_prb, mod, tbs, snr, tti0, nsf, short):
    """eNB PUSCH receive chain on the device (cfg3's receive side) vs the oracle chain on identical IQ: grid, ce, noise,
    equalised and de-precoded symbols, de-interleaved LLRs, per-block pass counts, CRC flags and TB bytes."""
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(1100 + prb + L + int(snr * 10))
    low_snr_case = snr < 9.5
    snr = snr + (0.8 if short else 0.0)  # one data symbol less for the same transport block
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6, shortened=short)
    iq, data = zip(*[make_ul_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j)) for b in range(nsf)])
    rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, 6, nsf, 2, 5, True, L >= 6, shortened=short)
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    grid = rx.debug(0, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    ce = rx.debug(1, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    res = rx.debug(2, np.float32, nsf * 5).reshape(nsf, 5)
    d = rx.debug(3, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    g = rx.debug(4, np.int16, nsf * cfg.nbits).reshape(nsf, -1)
    n_ok = 0
    for b in range(nsf):
        r = oracle_ul_rx(cfg, iq[b], tti0 + b, keep=True)
        assert_close_c(grid[b], r["grid"], "grid sf %d" % b)
        assert_close_c(ce[b], r["ce"], "ce sf %d" % b)
        assert abs(res[b, 0] - r["noise"]) <= 1e-4 * abs(r["noise"])
        assert_close_c(d[b], r["d"], "d sf %d" % b)
        diff = np.abs(g[b].astype(np.int32) - r["g"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size
        assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
        if r["ok"] or diff.max() == 0:
            assert np.array_equal(tb[b], r["tb"])
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_ok > 0 or low_snr_case
    rx.free()


@pytest.mark.parametrize("prb", [7, 20, 33, 64, 91, 110])
def test_ul_rx_chain_any_bandwidth(hp, prb):
    """The PUSCH receive chain at cell bandwidths between the six of 36.101 (any 6 .. 110 PRB is a valid cell, phy_common.c:43-52)."""
    test_ul_rx_chain_drawn_configurations(hp, 100 + prb, prb)


@pytest.mark.parametrize("seed", range(10))
def test_ul_rx_chain_drawn_configurations(hp, seed, force_prb=None):
    """test_ul_rx_chain on configurations DRAWN from what the pipeline accepts: bandwidth, cell id, RNTI, an allocation of 2^a 3^b 5^c PRBs at a
    drawn offset (with or without hopping between the slots), modulation, DMRS cyclic shifts and group / sequence hopping, a transport-block
    size not taken from a table at a drawn code rate, shortened subframes, a drawn first TTI, an SNR around the waterfall. The float stages
    (OFDM, estimator, equaliser, transform de-precoding, demapper) stay within one LSB of the oracle's LLRs on at most 1 in 1000; the integer
    back end (rate de-matching, turbo decoder, CRCs) equals the oracle's run on the DEVICE's LLRs exactly, failing blocks included."""
    from _libs import OrcCbsegm, OrcSchCfg
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(7100 + seed)
    prb = int(rng.choice([6, 15, 25, 50, 100]))
    if force_prb:
        prb = force_prb
    sizes = [n for n in range(1, prb + 1) if _is_235(n)]
    L = int(rng.choice(sizes))
    n_prb = int(rng.integers(0, prb - L + 1))
    hop = None if seed % 3 else int(rng.integers(0, prb - L + 1))
    mod = int(rng.choice([1, 2, 3]))
    short, ghop, shop = bool(seed % 2), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and L >= 6
    cell_id, rnti, n_dmrs, cs, dss = int(rng.integers(0, 504)), int(rng.integers(1, 0xFFF0)), int(rng.integers(0, 8)), int(rng.integers(0, 8)), int(rng.integers(0, 30))
    probe = UlConfig(prb, cell_id, mod, 16, L, n_prb, n_dmrs=n_dmrs, rnti=rnti, shortened=short)
    tbs = max(16, int(float(rng.uniform(0.2, 0.75)) * probe.nbits) // 8 * 8)
    while True:  # the next size below that needs neither filler bits nor two block lengths
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs -= 8
    cfg = UlConfig(prb, cell_id, mod, tbs, L, n_prb, n_dmrs=n_dmrs, rnti=rnti, cyclic_shift=cs, delta_ss=dss, n_prb_slot1=hop, group_hopping=ghop,
                   sequence_hopping=shop, shortened=short)
    tti0, nsf = int(rng.integers(0, 10240)), 3
    snr = {1: 1.0, 2: 7.0, 3: 12.0}[mod] + 10.0 * (tbs / cfg.nbits - 0.4) + float(rng.uniform(-2.0, 4.0)) + (3.0 if L < 3 else 0.0)
    iq, data = zip(*[make_ul_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, gain=0.9 * np.exp(-0.4j)) for b in range(nsf)])
    rx = hp.UlRx(cell_id, prb, rnti, mod, tbs, L, n_prb, n_dmrs, 6, nsf, cs, dss, ghop, shop, shortened=short, n_prb_slot1=hop)
    tb, ok = rx.decode(np.stack(iq), tti0)
    it = rx.debug(6, np.uint32, nsf * cfg.seg.C).reshape(nsf, -1)
    g = rx.debug(4, np.int16, nsf * cfg.nbits).reshape(nsf, -1)
    n_diff = 0
    for b in range(nsf):
        what = (prb, L, n_prb, hop, mod, tbs, short, ghop, shop, tti0 + b, snr)
        r = oracle_ul_rx(cfg, iq[b], tti0 + b, keep=True)
        diff = np.abs(g[b].astype(np.int32) - r["g"].astype(np.int32))
        assert diff.max() <= 1, what
        n_diff += int((diff != 0).sum())
        sch = OrcSchCfg(tbs, cfg.nbits, cfg.Qm, 0, cfg.max_iter)
        otb, oit, ocb = np.zeros(tbs // 8 + 16, np.uint8), np.zeros(cfg.seg.C, np.uint32), np.zeros(cfg.seg.C, np.uint8)
        rc = oracle().orc_dlsch_decode(C.byref(sch), p(np.ascontiguousarray(g[b])), p(otb), p(oit), p(ocb))
        assert bool(ok[b]) == (rc == 0) and np.array_equal(it[b], oit) and np.array_equal(tb[b], otb[:tbs // 8 + 3]), what
        if ok[b]:
            assert np.array_equal(tb[b][:tbs // 8], data[b]), what
        if not diff.any():
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]) and np.array_equal(tb[b], r["tb"]), what
    assert n_diff <= max(4, 1e-3 * nsf * cfg.nbits)  # a one-PRB allocation has 792 LLRs per subframe: a floor of four
    rx.free()


def _is_235(n):
    for f in (2, 3, 5):
        while n % f == 0:
            n //= f
    return n == 1


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,tti0,nsf", [(6, 6, 0, 1, 1000, 2, 4), (25, 10, 5, 2, 4008, 8, 11), (100, 100, 0, 2, 43816, 0, 3),
                                                            (100, 100, 0, 3, 75376, 4, 3), (100, 48, 20, 3, 30576, 7, 3), (15, 3, 12, 1, 328, 9, 2),
                                                            (25, 1, 7, 1, 104, 3, 6), (50, 2, 31, 2, 328, 0, 4), (50, 30, 4, 2, 6200, 1, 3)])
@pytest.mark.parametrize("short", [False, True])
def test_ul_tx_chain(hp, prb, L, n_prb, mod, tbs, tti0, nsf, short):
    """UE PUSCH transmit chain on the device vs the oracle's: code blocks with both CRCs, modulated symbols (exact: the
    bits are exact and the levels are table values), transform-precoded symbols, resource grid with DMRS, time samples."""
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(1300 + prb + L + mod)
    hop = dict(n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, shortened=short, **hop)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.UlTx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, nsf, 2, 5, True, L >= 6, shortened=short)
    iq = tx.encode(data, tti0)
    d = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    z = tx.debug(3, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    grid = tx.debug(4, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    for b in range(nsf):
        k = {}
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, data=data[b], keep=k)
        assert np.array_equal(d[b].view(np.float32), k["d"].view(np.float32)), "modulated symbols sf %d" % b
        assert_close_c(z[b], k["z"], "z sf %d" % b)
        assert_close_c(grid[b], k["grid"], "grid sf %d" % b)
        assert_close_c(iq[b], iq_o, "iq sf %d" % b)
    tx.free()


@pytest.mark.parametrize("seed", range(10))
def test_ul_tx_chain_drawn_configurations(hp, seed, force_prb=None):
    """test_ul_tx_chain on configurations drawn from what the transmit pipeline accepts (the draw of test_ul_rx_chain_drawn_configurations):
    modulated symbols exactly, transform-precoded symbols, grid with DMRS and time samples to the float tolerance of the other tests."""
    from _libs import OrcCbsegm
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(7300 + seed)
    prb = force_prb or int(rng.choice([6, 15, 25, 50, 100]))
    L = int(rng.choice([n for n in range(1, prb + 1) if _is_235(n)]))
    n_prb = int(rng.integers(0, prb - L + 1))
    hop = None if seed % 3 else int(rng.integers(0, prb - L + 1))
    mod = int(rng.choice([1, 2, 3]))
    short, ghop, shop = bool(seed % 2), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and L >= 6
    cell_id, rnti, n_dmrs, cs, dss = int(rng.integers(0, 504)), int(rng.integers(1, 0xFFF0)), int(rng.integers(0, 8)), int(rng.integers(0, 8)), int(rng.integers(0, 30))
    probe = UlConfig(prb, cell_id, mod, 16, L, n_prb, n_dmrs=n_dmrs, rnti=rnti, shortened=short)
    tbs = max(16, int(float(rng.uniform(0.2, 0.85)) * probe.nbits) // 8 * 8)
    while True:
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs -= 8
    cfg = UlConfig(prb, cell_id, mod, tbs, L, n_prb, n_dmrs=n_dmrs, rnti=rnti, cyclic_shift=cs, delta_ss=dss, n_prb_slot1=hop, group_hopping=ghop,
                   sequence_hopping=shop, shortened=short)
    tti0, nsf = int(rng.integers(0, 10240)), 3
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.UlTx(cell_id, prb, rnti, mod, tbs, L, n_prb, n_dmrs, nsf, cs, dss, ghop, shop, shortened=short, n_prb_slot1=hop)
    iq = tx.encode(data, tti0)
    d = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    z = tx.debug(3, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    grid = tx.debug(4, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    for b in range(nsf):
        k, what = {}, (prb, L, n_prb, hop, mod, tbs, short, ghop, shop, cell_id, tti0 + b)
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, data=data[b], keep=k)
        assert np.array_equal(d[b].view(np.float32), k["d"].view(np.float32)), what
        assert_close_c(z[b], k["z"], "z %s" % (what,))
        assert_close_c(grid[b], k["grid"], "grid %s" % (what,))
        assert_close_c(iq[b], iq_o, "iq %s" % (what,))
    tx.free()


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,tti0,nsf,O,Ioff,short", [(25, 10, 5, 2, 4008, 8, 6, 1, 9, False), (25, 10, 5, 2, 4008, 1, 6, 2, 9, False),
                                                                        (6, 6, 0, 1, 1000, 2, 4, 2, 5, True), (100, 48, 20, 3, 30576, 7, 4, 1, 12, False),
                                                                        (100, 100, 0, 3, 75376, 4, 4, 2, 14, True), (25, 1, 7, 1, 104, 3, 4, 2, 10, False),
                                                                        (50, 2, 31, 2, 328, 0, 4, 1, 0, False)])
def test_ul_tx_chain_harq_ack(hp, prb, L, n_prb, mod, tbs, tti0, nsf, O, Ioff, short):
    """PUSCH transmit chain with 1-2 HARQ-ACK bits multiplexed next to the DMRS (sch.c:1168-1215, uci.c:497-602, the placeholder /
    repetition handling of pusch.c:386-400) vs the oracle's (pinned on srslte_ulsch_encode): modulated symbols exact, samples 1e-4."""
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(1700 + prb + L + mod + O)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, shortened=short, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    acks = np.array([[(b >> j) & 1 for j in range(O)] for b in range(nsf)], np.uint8)  # every combination
    tx = hp.UlTx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, nsf, 2, 5, True, L >= 6, shortened=short, ack_len=O, I_offset_ack=Ioff)
    assert hp.lib().srslte_hip_ul_tx_batch(tx.h, tx.d_iq.ptr, tbs // 8, 0, 1, tx.d_iq.ptr, None) == hp.SRSLTE_ERROR_INVALID_INPUTS  # no ACK values
    iq = tx.encode(data, tti0, ack=acks)
    d = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    n_diff = 0
    for b in range(nsf):
        k, k0 = {}, {}
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, data=data[b], keep=k, ack=tuple(acks[b]), I_offset_ack=Ioff)
        make_ul_subframe(cfg, tti0 + b, rng, data=data[b], keep=k0)
        n_diff += int((k["d"] != k0["d"]).sum())
        assert np.array_equal(d[b].view(np.float32), k["d"].view(np.float32)), "modulated symbols sf %d" % b
        assert_close_c(iq[b], iq_o, "iq sf %d" % b)
    assert n_diff > 0  # the ACK did change symbols
    tx.free()


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,snr,tti0,nsf,O,Ioff,short", [(25, 10, 5, 2, 4008, 10.5, 8, 8, 1, 9, False), (25, 10, 5, 2, 4008, 10.5, 1, 8, 2, 9, False),
                                                                            (6, 6, 0, 1, 1000, 5.0, 2, 4, 2, 5, True), (100, 48, 20, 3, 30576, 18.0, 7, 4, 1, 12, False),
                                                                            (100, 100, 0, 2, 43816, 13.5, 4, 4, 2, 14, True), (25, 1, 7, 1, 104, 6.0, 3, 4, 2, 10, False),
                                                                            (25, 10, 5, 1, 1000, -3.0, 0, 8, 2, 2, False)])
def test_ul_rx_chain_harq_ack(hp, prb, L, n_prb, mod, tbs, snr, tti0, nsf, O, Ioff, short):
    """PUSCH receive chain with HARQ-ACK on the PUSCH (uci_decode_ri_ack sch.c:929-966, uci.c:755-790) vs the oracle chain (pinned on
    srslte_ulsch_decode) on identical IQ: ACK decisions, de-interleaved LLRs with the ACK positions zeroed, pass counts, CRC, TB."""
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(1800 + prb + L + mod + O)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6, shortened=short)
    acks = np.array([[(b >> j) & 1 for j in range(O)] for b in range(nsf)], np.uint8)
    iq, data = zip(*[make_ul_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j), ack=tuple(acks[b]), I_offset_ack=Ioff)
                     for b in range(nsf)])
    rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, 6, nsf, 2, 5, True, L >= 6, shortened=short, ack_len=O, I_offset_ack=Ioff)
    for rep in range(2):  # the second call checks that the accumulators are cleared per call
        tb, ok = rx.decode(np.stack(iq), tti0)
        ack = rx.ack()
        C_ = cfg.seg.C
        it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
        g = rx.debug(4, np.int16, nsf * cfg.nbits).reshape(nsf, -1)
        n_ok = n_zero = 0
        for b in range(nsf):
            r = oracle_ul_rx(cfg, iq[b], tti0 + b, keep=True, O_ack=O, I_offset_ack=Ioff)
            diff = np.abs(g[b].astype(np.int32) - r["g"].astype(np.int32))
            assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size
            zeroed = (r["q"] == 0) & (r["q_before_ack"] != 0)
            n_zero += int(zeroed.sum())
            assert np.array_equal(ack[b], r["ack"][:O]), "ack sf %d" % b
            if snr > 0:
                assert np.array_equal(ack[b], acks[b])
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
            if r["ok"] or diff.max() == 0:
                assert np.array_equal(tb[b], r["tb"])
            if r["ok"]:
                n_ok += 1
                assert np.array_equal(tb[b][:tbs // 8], data[b])
        assert n_zero > 0 and (n_ok > 0 or snr < 0)
    rx.free()


UL_RI_CASES = [(25, 10, 5, 2, 4008, 12.0, 8, 6, 1, 9, 0, 0, False), (25, 10, 5, 2, 4008, 12.0, 1, 6, 1, 9, 2, 9, False), (6, 6, 0, 1, 1000, 6.0, 2, 4, 1, 5, 1, 5, True),
               (100, 48, 20, 3, 30576, 19.0, 7, 4, 1, 12, 0, 0, False), (50, 20, 3, 2, 7736, 12.0, 4, 4, 2, 8, 1, 8, False),
               (100, 100, 0, 2, 43816, 15.0, 4, 4, 1, 11, 2, 12, True), (25, 1, 7, 1, 104, 8.0, 3, 4, 1, 10, 0, 0, False), (15, 3, 12, 3, 1800, 19.0, 9, 3, 2, 12, 2, 3, False)]


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,snr,tti0,nsf,O_ri,I_ri,O_ack,I_ack,short", UL_RI_CASES)
def test_ul_tx_chain_rank_indication(hp, prb, L, n_prb, mod, tbs, snr, tti0, nsf, O_ri, I_ri, O_ack, I_ack, short):
    """PUSCH transmit chain with a rank indication (and HARQ-ACK): RI symbols on their own columns, left out by the channel interleaver,
    UL-SCH rate-matched to the rest (sch.c:580-598,:1110-1160) vs the oracle's (pinned on srslte_ulsch_encode): symbols exact."""
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(1900 + prb + L + mod + O_ri)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, shortened=short, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    ris = np.array([[(b >> j) & 1 for j in range(O_ri)] for b in range(nsf)], np.uint8)
    acks = np.array([[((b + 1) >> j) & 1 for j in range(O_ack)] for b in range(nsf)], np.uint8) if O_ack else None
    tx = hp.UlTx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, nsf, 2, 5, True, L >= 6, shortened=short, ack_len=O_ack, I_offset_ack=I_ack, ri_len=O_ri,
                 I_offset_ri=I_ri)
    iq = tx.encode(data, tti0, ack=acks, ri=ris)
    d = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    for b in range(nsf):
        k = {}
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, data=data[b], keep=k, ack=tuple(acks[b]) if O_ack else (), I_offset_ack=I_ack,
                                   ri=tuple(ris[b]), I_offset_ri=I_ri)
        assert np.array_equal(d[b].view(np.float32), k["d"].view(np.float32)), "modulated symbols sf %d" % b
        assert_close_c(iq[b], iq_o, "iq sf %d" % b)
    tx.free()


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,snr,tti0,nsf,O_ri,I_ri,O_ack,I_ack,short", UL_RI_CASES)
def test_ul_rx_chain_rank_indication(hp, prb, L, n_prb, mod, tbs, snr, tti0, nsf, O_ri, I_ri, O_ack, I_ack, short):
    """PUSCH receive chain with a rank indication (and HARQ-ACK) vs the oracle chain (pinned on srslte_ulsch_decode) on identical IQ:
    RI and ACK decisions, the de-interleaved UL-SCH LLRs - first element included, which the reference's scatter leaves holding the last
    RI position's LLR (sch.c:891-918) -, pass counts, CRC, TB."""
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx, ul_ri_layout
    rng = np.random.default_rng(2000 + prb + L + mod + O_ri)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6, shortened=short)
    G = ul_ri_layout(cfg, O_ri, I_ri)[3]
    ris = np.array([[(b >> j) & 1 for j in range(O_ri)] for b in range(nsf)], np.uint8)
    acks = np.array([[((b + 1) >> j) & 1 for j in range(O_ack)] for b in range(nsf)], np.uint8) if O_ack else None
    iq, data = zip(*[make_ul_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j), ack=tuple(acks[b]) if O_ack else (),
                                      I_offset_ack=I_ack, ri=tuple(ris[b]), I_offset_ri=I_ri) for b in range(nsf)])
    rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, 6, nsf, 2, 5, True, L >= 6, shortened=short, ack_len=O_ack, I_offset_ack=I_ack, ri_len=O_ri,
                 I_offset_ri=I_ri)
    for rep in range(2):
        tb, ok = rx.decode(np.stack(iq), tti0)
        ri, ack = rx.ri(), rx.ack()
        C_ = cfg.seg.C
        it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
        g = rx.debug(4, np.int16, nsf * cfg.nbits).reshape(nsf, -1)
        n_ok = 0
        for b in range(nsf):
            r = oracle_ul_rx(cfg, iq[b], tti0 + b, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri)
            diff = np.abs(g[b, :G].astype(np.int32) - r["g"].astype(np.int32))
            assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size + 1, (b, int(diff.max()), int((diff != 0).sum()))
            assert np.array_equal(ri[b], r["ri"][:O_ri]) and np.array_equal(ri[b], ris[b]), "ri sf %d" % b
            if O_ack:
                assert np.array_equal(ack[b], r["ack"][:O_ack])  # (with Q' <= 3 the 2-bit decoder never combines a triplet: all-zero decisions, uci.c:776-777)
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
            if r["ok"] or diff.max() == 0:
                assert np.array_equal(tb[b], r["tb"])
            if r["ok"]:
                n_ok += 1
                assert np.array_equal(tb[b][:tbs // 8], data[b])
        assert n_ok > 0
    rx.free()


# prb, L, n_prb, mod, tbs, snr, tti0, nsf, O_cqi, I_cqi, O_ri, I_ri, O_ack, I_ack, shortened
UL_CQI_CASES = [(25, 10, 5, 2, 4008, 12.0, 8, 6, 4, 7, 0, 0, 0, 0, False), (25, 10, 5, 2, 4008, 12.0, 1, 6, 11, 9, 1, 9, 2, 9, False),
                (6, 6, 0, 1, 1000, 6.0, 2, 4, 20, 6, 1, 5, 1, 5, True), (100, 48, 20, 3, 30576, 19.0, 7, 4, 64, 12, 0, 0, 0, 0, False),
                (50, 20, 3, 2, 7736, 12.0, 4, 4, 28, 8, 2, 8, 1, 8, False), (100, 100, 0, 2, 43816, 15.0, 4, 4, 1, 15, 1, 11, 2, 12, True),
                (25, 2, 7, 1, 256, 8.0, 3, 4, 12, 2, 0, 0, 0, 0, False), (15, 3, 12, 3, 1800, 19.0, 9, 3, 36, 10, 2, 12, 2, 3, False),
                (50, 25, 0, 2, 9912, 3.0, 0, 4, 48, 4, 1, 6, 2, 6, False)]


def _cqi_bits(nsf, O):
    r = np.random.default_rng(4242 + O)
    return r.integers(0, 2, (nsf, O), dtype=np.uint8)


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,snr,tti0,nsf,O_cqi,I_cqi,O_ri,I_ri,O_ack,I_ack,short", UL_CQI_CASES)
def test_ul_tx_chain_cqi(hp, prb, L, n_prb, mod, tbs, snr, tti0, nsf, O_cqi, I_cqi, O_ri, I_ri, O_ack, I_ack, short):
    """PUSCH transmit chain with a CQI report (block code up to 11 bits, CRC-8 + tail-biting convolutional code + rate matching above),
    alone and with RI / HARQ-ACK: the report's Q' symbols lead the interleaved stream, the UL-SCH is rate-matched to the rest
    (sch.c:1133-1160, uci.c:264-302,:470-494) vs the oracle's (pinned on srslte_ulsch_encode): symbols exact."""
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(2100 + prb + L + mod + O_cqi)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, shortened=short, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    cqis = _cqi_bits(nsf, O_cqi)
    ris = np.array([[(b >> j) & 1 for j in range(O_ri)] for b in range(nsf)], np.uint8) if O_ri else None
    acks = np.array([[((b + 1) >> j) & 1 for j in range(O_ack)] for b in range(nsf)], np.uint8) if O_ack else None
    tx = hp.UlTx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, nsf, 2, 5, True, L >= 6, shortened=short, ack_len=O_ack, I_offset_ack=I_ack, ri_len=O_ri,
                 I_offset_ri=I_ri, cqi_len=O_cqi, I_offset_cqi=I_cqi)
    iq = tx.encode(data, tti0, ack=acks, ri=ris, cqi=cqis)
    d = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    for b in range(nsf):
        k = {}
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, data=data[b], keep=k, ack=tuple(acks[b]) if O_ack else (), I_offset_ack=I_ack,
                                   ri=tuple(ris[b]) if O_ri else (), I_offset_ri=I_ri, cqi=tuple(cqis[b]), I_offset_cqi=I_cqi)
        assert np.array_equal(d[b].view(np.float32), k["d"].view(np.float32)), "modulated symbols sf %d" % b
        assert_close_c(iq[b], iq_o, "iq sf %d" % b)
    tx.free()


@pytest.mark.parametrize("seed", range(16))
def test_ul_chains_uci_drawn_configurations(hp, seed):
    """PUSCH with control information DRAWN from the space the pipelines accept: a CQI report of 0 .. 64 bits (block code / convolutional code),
    0-2 rank-indication and 0-2 HARQ-ACK bits, every beta-offset index of 36.213 Tables 8.6.3-1/2/3, allocation, modulation, a transport-block
    size not taken from a table, shortened subframes. Transmit side: modulated symbols equal the oracle's exactly (the Q' of every field,
    the placement around the DMRS, the interleaver with RI symbols left out, the rate matching to what remains). Receive side on the same
    noise-free samples: every field and the transport block come back."""
    from _libs import OrcCbsegm
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(7400 + seed)
    prb = int(rng.choice([15, 25, 50, 100]))
    L = int(rng.choice([n for n in range(3, prb + 1) if _is_235(n)]))
    n_prb, mod, short = int(rng.integers(0, prb - L + 1)), int(rng.choice([1, 2, 3])), bool(seed % 2)
    O_cqi = int(rng.choice([0, int(rng.integers(1, 12)), int(rng.integers(12, 65))]))
    O_ri, O_ack = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    I_cqi, I_ri, I_ack = int(rng.integers(2, 16)), int(rng.integers(0, 13)), int(rng.integers(0, 15))
    cell_id, rnti, n_dmrs, cs, dss = int(rng.integers(0, 504)), int(rng.integers(1, 0xFFF0)), int(rng.integers(0, 8)), int(rng.integers(0, 8)), int(rng.integers(0, 30))
    probe = UlConfig(prb, cell_id, mod, 16, L, n_prb, n_dmrs=n_dmrs, rnti=rnti, shortened=short)
    tbs = max(40, int(float(rng.uniform(0.15, 0.45)) * probe.nbits) // 8 * 8)  # low enough that the control information still fits
    while True:
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs -= 8
    cfg = UlConfig(prb, cell_id, mod, tbs, L, n_prb, n_dmrs=n_dmrs, rnti=rnti, cyclic_shift=cs, delta_ss=dss, group_hopping=bool(seed & 4),
                   sequence_hopping=bool(seed & 8) and L >= 6, shortened=short)
    tti0, nsf = int(rng.integers(0, 10240)), 4
    what = (prb, L, n_prb, mod, tbs, short, O_cqi, I_cqi, O_ri, I_ri, O_ack, I_ack, cell_id, tti0)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    cqis = rng.integers(0, 2, (nsf, O_cqi), dtype=np.uint8)
    ris = rng.integers(0, 2, (nsf, O_ri), dtype=np.uint8) if O_ri else None
    acks = rng.integers(0, 2, (nsf, O_ack), dtype=np.uint8) if O_ack else None
    uci = dict(ack_len=O_ack, I_offset_ack=I_ack, ri_len=O_ri, I_offset_ri=I_ri, cqi_len=O_cqi, I_offset_cqi=I_cqi)
    tx = hp.UlTx(cell_id, prb, rnti, mod, tbs, L, n_prb, n_dmrs, nsf, cs, dss, bool(seed & 4), bool(seed & 8) and L >= 6, shortened=short, **uci)
    iq = tx.encode(data, tti0, ack=acks, ri=ris, cqi=cqis if O_cqi else None)
    d = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    for b in range(nsf):
        k = {}
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, data=data[b], keep=k, ack=tuple(acks[b]) if O_ack else (), I_offset_ack=I_ack,
                                   ri=tuple(ris[b]) if O_ri else (), I_offset_ri=I_ri, cqi=tuple(cqis[b]), I_offset_cqi=I_cqi)
        assert np.array_equal(d[b].view(np.float32), k["d"].view(np.float32)), what + (b,)
        assert_close_c(iq[b], iq_o, "iq %s" % (what,))
    rx = hp.UlRx(cell_id, prb, rnti, mod, tbs, L, n_prb, n_dmrs, 6, nsf, cs, dss, bool(seed & 4), bool(seed & 8) and L >= 6, shortened=short, **uci)
    tb, ok = rx.decode(iq, tti0)
    assert ok.all() and np.array_equal(tb[:, :tbs // 8], data), what
    if O_ri:
        assert np.array_equal(rx.ri()[:nsf], ris), what
    if O_ack:
        assert np.array_equal(rx.ack()[:nsf], acks), what
    if O_cqi:
        cq, cq_ok = rx.cqi()
        assert np.array_equal(cq[:nsf, :O_cqi], cqis) and (O_cqi <= 11 or cq_ok[:nsf].all()), what
    tx.free()
    rx.free()


@pytest.mark.parametrize("prb,L,n_prb,mod,tbs,snr,tti0,nsf,O_cqi,I_cqi,O_ri,I_ri,O_ack,I_ack,short", UL_CQI_CASES)
def test_ul_rx_chain_cqi(hp, prb, L, n_prb, mod, tbs, snr, tti0, nsf, O_cqi, I_cqi, O_ri, I_ri, O_ack, I_ack, short):
    """PUSCH receive chain with a CQI report vs the oracle chain (pinned on srslte_ulsch_decode / srslte_uci_decode_cqi_pusch, the long
    report through srslte_viterbi_decode_f) on identical IQ: report bits and CRC flag, RI / ACK decisions, de-interleaved LLRs, pass
    counts, CRC, TB. The last case is noisy enough for wrong reports: same wrong bits, same CRC verdicts."""
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx, ul_ri_layout
    rng = np.random.default_rng(2200 + prb + L + mod + O_cqi)
    cfg = UlConfig(prb, 11, mod, tbs, L, n_prb, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6, shortened=short)
    G = ul_ri_layout(cfg, O_ri, I_ri)[3]
    cqis = _cqi_bits(nsf, O_cqi)
    ris = np.array([[(b >> j) & 1 for j in range(O_ri)] for b in range(nsf)], np.uint8) if O_ri else None
    acks = np.array([[((b + 1) >> j) & 1 for j in range(O_ack)] for b in range(nsf)], np.uint8) if O_ack else None
    iq, data = zip(*[make_ul_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j), ack=tuple(acks[b]) if O_ack else (),
                                      I_offset_ack=I_ack, ri=tuple(ris[b]) if O_ri else (), I_offset_ri=I_ri, cqi=tuple(cqis[b]),
                                      I_offset_cqi=I_cqi) for b in range(nsf)])
    rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n_prb, 3, 6, nsf, 2, 5, True, L >= 6, shortened=short, ack_len=O_ack, I_offset_ack=I_ack, ri_len=O_ri,
                 I_offset_ri=I_ri, cqi_len=O_cqi, I_offset_cqi=I_cqi)
    for rep in range(2):
        tb, ok = rx.decode(np.stack(iq), tti0)
        ri, ack = rx.ri(), rx.ack()
        cqi, cqi_ok = rx.cqi()
        C_ = cfg.seg.C
        it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
        g = rx.debug(4, np.int16, nsf * cfg.nbits).reshape(nsf, -1)
        n_ok = n_cqi_ok = 0
        for b in range(nsf):
            r = oracle_ul_rx(cfg, iq[b], tti0 + b, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri, O_cqi=O_cqi,
                             I_offset_cqi=I_cqi)
            diff = np.abs(g[b, :G].astype(np.int32) - r["g"].astype(np.int32))
            assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size + 1, (b, int(diff.max()), int((diff != 0).sum()))
            exact = diff.max() == 0
            if O_ri:
                assert np.array_equal(ri[b], r["ri"][:O_ri]) and np.array_equal(ri[b], ris[b]), "ri sf %d" % b
            if O_ack:
                assert np.array_equal(ack[b], r["ack"][:O_ack])
            if exact or snr > 5:
                assert bool(cqi_ok[b]) == r["cqi_ok"], "cqi crc sf %d" % b
                if r["cqi_ok"]:
                    assert np.array_equal(cqi[b], r["cqi"]), "cqi sf %d" % b
            if r["cqi_ok"] and np.array_equal(r["cqi"], cqis[b]):
                n_cqi_ok += 1
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
            if r["ok"] or exact:
                assert np.array_equal(tb[b], r["tb"])
            if r["ok"]:
                n_ok += 1
                assert np.array_equal(tb[b][:tbs // 8], data[b])
        assert (n_ok > 0 and n_cqi_ok > 0) or snr < 5
    rx.free()


def test_ul_tx_rx_loop_cqi(hp):
    """Device transmit chain with CQI report, HARQ-ACK and rank indication into the device receive chain (noise-free): everything comes
    back, for a block-coded and a convolutionally coded report."""
    prb, L, n_prb, mod, tbs, nsf = 50, 40, 4, 2, 17568, 12
    rng = np.random.default_rng(80)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    acks, ris = rng.integers(0, 2, (nsf, 2), dtype=np.uint8), rng.integers(0, 2, (nsf, 1), dtype=np.uint8)
    for O in (4, 11, 12, 40, 64):
        cqis = rng.integers(0, 2, (nsf, O), dtype=np.uint8)
        kw = dict(ack_len=2, I_offset_ack=8, ri_len=1, I_offset_ri=7, cqi_len=O, I_offset_cqi=9)
        tx = hp.UlTx(3, prb, 0x77, mod, tbs, L, n_prb, 1, nsf, **kw)
        rx = hp.UlRx(3, prb, 0x77, mod, tbs, L, n_prb, 1, 6, nsf, **kw)
        tb, ok = rx.decode(tx.encode(data, 5, ack=acks, ri=ris, cqi=cqis), 5)
        cqi, cqi_ok = rx.cqi()
        assert ok.all() and np.array_equal(tb[:, :tbs // 8], data) and np.array_equal(rx.ack(), acks) and np.array_equal(rx.ri(), ris)
        assert cqi_ok.all() and np.array_equal(cqi, cqis), O
        tx.free()
        rx.free()


UL_NODATA_CASES = [  # prb, L, n_prb, mod, snr, tti0, nsf, O_cqi, I_cqi, O_ri, I_ri, O_ack, I_ack, short
    (25, 4, 3, 1, 6.0, 2, 5, 4, 7, 0, 0, 0, 0, False), (50, 4, 10, 1, 5.0, 7, 4, 11, 12, 0, 0, 1, 9, False), (100, 3, 0, 1, 6.0, 0, 6, 20, 9, 1, 7, 2, 9, False),
    (15, 2, 1, 2, 11.0, 5, 3, 10, 8, 0, 0, 0, 0, True), (50, 6, 0, 2, 12.0, 9, 4, 40, 10, 2, 6, 1, 8, False), (100, 4, 2, 1, 2.0, 3, 6, 64, 6, 2, 5, 0, 0, False),
    (25, 1, 7, 1, 7.0, 8, 3, 5, 15, 1, 7, 1, 9, False)]


@pytest.mark.parametrize("prb,L,n_prb,mod,snr,tti0,nsf,O_cqi,I_cqi,O_ri,I_ri,O_ack,I_ack,short", UL_NODATA_CASES)
def test_ul_chains_without_ulsch_data(hp, prb, L, n_prb, mod, snr, tti0, nsf, O_cqi, I_cqi, O_ri, I_ri, O_ack, I_ack, short):
    """A PUSCH that carries a CQI report and no transport block (tbs = 0; srslte_ulsch_encode / _decode with cb_segm.tbs == 0, sch.c:1062-1065,
    :1157-1165): the report fills what the rank indication leaves (uci.c:266-281), HARQ-ACK and RI are sized by the report (uci.c:557-564).
    Transmit pipeline vs the oracle's stimulus (pinned on srslte_ulsch_encode, test_pusch_without_ulsch_data_vs_reference): modulated symbols
    exact; receive pipeline vs the oracle chain on the same noisy subframes: report bits + CRC flag, RI / ACK decisions, LLRs; no transport
    block is delivered."""
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx, ul_ri_layout
    rng = np.random.default_rng(3100 + prb + L + mod + O_cqi)
    cfg = UlConfig(prb, 11, mod, 0, L, n_prb, shortened=short, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6)
    cqis = _cqi_bits(nsf, O_cqi)
    ris = np.array([[(b >> j) & 1 for j in range(O_ri)] for b in range(nsf)], np.uint8) if O_ri else None
    acks = np.array([[((b + 1) >> j) & 1 for j in range(O_ack)] for b in range(nsf)], np.uint8) if O_ack else None
    kw = dict(shortened=short, ack_len=O_ack, I_offset_ack=I_ack, ri_len=O_ri, I_offset_ri=I_ri, cqi_len=O_cqi, I_offset_cqi=I_cqi)
    tx = hp.UlTx(11, prb, 0x1234, mod, 0, L, n_prb, 3, nsf, 2, 5, True, L >= 6, **kw)
    iq = tx.encode(np.zeros((nsf, 1), np.uint8), tti0, ack=acks, ri=ris, cqi=cqis)
    d = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    okw = lambda b: dict(ack=tuple(acks[b]) if O_ack else (), I_offset_ack=I_ack, ri=tuple(ris[b]) if O_ri else (), I_offset_ri=I_ri, cqi=tuple(cqis[b]),
                         I_offset_cqi=I_cqi)
    for b in range(nsf):
        k = {}
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, keep=k, **okw(b))
        assert np.array_equal(d[b].view(np.float32), k["d"].view(np.float32)), "modulated symbols sf %d" % b
        assert_close_c(iq[b], iq_o, "iq sf %d" % b)
    tx.free()
    G = ul_ri_layout(cfg, O_ri, I_ri, O_cqi, I_cqi)[3]
    iqn = [make_ul_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j), **okw(b))[0] for b in range(nsf)]
    rx = hp.UlRx(11, prb, 0x1234, mod, 0, L, n_prb, 3, 6, nsf, 2, 5, True, L >= 6, **kw)
    tb, ok = rx.decode(np.stack(iqn), tti0)
    assert not ok.any()
    ri, ack = rx.ri(), rx.ack()
    cqi, cqi_ok = rx.cqi()
    g = rx.debug(4, np.int16, nsf * cfg.nbits).reshape(nsf, -1)
    n_good = 0
    for b in range(nsf):
        r = oracle_ul_rx(cfg, iqn[b], tti0 + b, keep=True, O_ack=O_ack, I_offset_ack=I_ack, O_ri=O_ri, I_offset_ri=I_ri, O_cqi=O_cqi, I_offset_cqi=I_cqi)
        diff = np.abs(g[b, :G].astype(np.int32) - r["g"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size + 1, (b, int(diff.max()), int((diff != 0).sum()))
        if O_ri:
            assert np.array_equal(ri[b], r["ri"][:O_ri]), "ri sf %d" % b
        if O_ack:
            assert np.array_equal(ack[b], r["ack"][:O_ack]), "ack sf %d" % b
        if diff.max() == 0 or snr > 5:
            assert bool(cqi_ok[b]) == r["cqi_ok"], "cqi crc sf %d" % b
            if r["cqi_ok"]:
                assert np.array_equal(cqi[b], r["cqi"]), "cqi sf %d" % b
        n_good += int(r["cqi_ok"] and np.array_equal(r["cqi"], cqis[b]))
    assert n_good >= nsf - 1 or snr < 5
    rx.free()


def test_ul_tx_rx_loop_without_ulsch_data(hp):
    """Device transmit chain into device receive chain for a CQI-only PUSCH (noise-free), block-coded and convolutionally coded reports, with
    HARQ-ACK and rank indication beside them; and what such an object refuses: grants mode, and creation without a report."""
    prb, L, n_prb, mod, nsf = 50, 4, 20, 1, 10
    rng = np.random.default_rng(81)
    acks, ris = rng.integers(0, 2, (nsf, 2), dtype=np.uint8), rng.integers(0, 2, (nsf, 1), dtype=np.uint8)
    for O in (4, 11, 12, 30, 64):
        cqis = rng.integers(0, 2, (nsf, O), dtype=np.uint8)
        kw = dict(ack_len=2, I_offset_ack=8, ri_len=1, I_offset_ri=7, cqi_len=O, I_offset_cqi=9)
        tx = hp.UlTx(3, prb, 0x77, mod, 0, L, n_prb, 1, nsf, **kw)
        rx = hp.UlRx(3, prb, 0x77, mod, 0, L, n_prb, 1, 6, nsf, **kw)
        tb, ok = rx.decode(tx.encode(np.zeros((nsf, 1), np.uint8), 5, ack=acks, ri=ris, cqi=cqis), 5)
        cqi, cqi_ok = rx.cqi()
        assert not ok.any() and np.array_equal(rx.ack(), acks) and np.array_equal(rx.ri(), ris)
        assert cqi_ok.all() and np.array_equal(cqi, cqis), O
        with pytest.raises(RuntimeError):
            rx.decode_grants(np.zeros((1, rx.sf_len), np.complex64), 0, [hp.UlGrant.make(0, 0x77, L, n_prb, mod, 0, cqi_len=O, I_offset_cqi=9)])
        tx.free()
        rx.free()
    for cls, args in ((hp.UlRx, (3, prb, 0x77, mod, 0, L, n_prb, 1, 6, nsf)), (hp.UlTx, (3, prb, 0x77, mod, 0, L, n_prb, 1, nsf))):
        with pytest.raises(RuntimeError):
            cls(*args)


def test_ul_cqi_config_errors(hp):
    """Creation fails for a reserved CQI offset index (beta < 0, sch.c:51-52) or more than 64 report bits; the plain entry points refuse
    a pipeline with a configured report."""
    for kw in (dict(cqi_len=4, I_offset_cqi=0), dict(cqi_len=4, I_offset_cqi=16), dict(cqi_len=65, I_offset_cqi=5)):
        with pytest.raises(RuntimeError):
            hp.UlRx(3, 25, 0x77, 2, 4008, 10, 2, 1, 6, 2, **kw)
        with pytest.raises(RuntimeError):
            hp.UlTx(3, 25, 0x77, 2, 4008, 10, 2, 1, 2, **kw)
    tx = hp.UlTx(3, 25, 0x77, 2, 4008, 10, 2, 1, 2, cqi_len=4, I_offset_cqi=5)
    with pytest.raises(RuntimeError):
        tx.encode(np.zeros((2, 501), np.uint8), 0)
    tx.free()


UL_HOP_CASES = [(25, 10, 5, 14, 2, 4008, 9.5, 8, 4, False), (100, 48, 50, 1, 3, 30576, 17.0, 3, 3, False), (6, 2, 0, 4, 1, 256, 5.0, 0, 6, True),
                (50, 25, 0, 25, 2, 9912, 9.0, 5, 3, False)]


@pytest.mark.parametrize("prb,L,n0,n1,mod,tbs,snr,tti0,nsf,short", UL_HOP_CASES)
def test_ul_chains_intra_subframe_hopping(hp, prb, L, n0, n1, mod, tbs, snr, tti0, nsf, short):
    """srslte_pusch_grant_t.n_prb[0] != n_prb[1] (cfg.hopping / n_prb_slot1): the transmit pipeline maps each slot's data and DMRS at
    its own PRB offset (symbols exact vs the oracle's), and the receive pipeline estimates, equalises and decodes them there - channel
    estimates, LLRs, pass counts, CRC and bytes vs the oracle chain (pinned on the reference's estimator with hopping)."""
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(2300 + prb + L + n1)
    cfg = UlConfig(prb, 11, mod, tbs, L, n0, shortened=short, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=L >= 6, n_prb_slot1=n1)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.UlTx(11, prb, 0x1234, mod, tbs, L, n0, 3, nsf, 2, 5, True, L >= 6, shortened=short, n_prb_slot1=n1)
    iq_dev = tx.encode(data, tti0)
    for b in range(nsf):
        iq_o, _ = make_ul_subframe(cfg, tti0 + b, rng, data=data[b])
        assert_close_c(iq_dev[b], iq_o, "iq sf %d" % b)
    tx.free()
    iq, _ = zip(*[make_ul_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, gain=0.8 * np.exp(0.7j), data=data[b]) for b in range(nsf)])
    rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n0, 3, 6, nsf, 2, 5, True, L >= 6, shortened=short, n_prb_slot1=n1)
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    g = rx.debug(4, np.int16, nsf * cfg.nbits).reshape(nsf, -1)
    ce = rx.debug(1, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    n_ok = 0
    for b in range(nsf):
        r = oracle_ul_rx(cfg, iq[b], tti0 + b, keep=True)
        for sym in range(14):  # estimates only where the slot's grant is
            o = sym * cfg.nre + 12 * (n0 if sym < 7 else n1)
            assert_close_c(ce[b][o:o + cfg.M_sc], r["ce"][o:o + cfg.M_sc], "ce sf %d symbol %d" % (b, sym))
        diff = np.abs(g[b].astype(np.int32) - r["g"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size + 1, (b, int(diff.max()))
        assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
        if r["ok"] or diff.max() == 0:
            assert np.array_equal(tb[b], r["tb"])
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_ok > 0
    rx.free()


def test_ul_tx_rx_loop_uci(hp):
    """Device transmit chain with HARQ-ACK and rank indication into the device receive chain (noise-free): everything comes back."""
    prb, L, n_prb, mod, tbs, nsf = 50, 40, 4, 2, 17568, 12
    rng = np.random.default_rng(79)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    acks, ris = rng.integers(0, 2, (nsf, 2), dtype=np.uint8), rng.integers(0, 2, (nsf, 1), dtype=np.uint8)
    kw = dict(ack_len=2, I_offset_ack=8, ri_len=1, I_offset_ri=7)
    tx = hp.UlTx(3, prb, 0x77, mod, tbs, L, n_prb, 1, nsf, **kw)
    rx = hp.UlRx(3, prb, 0x77, mod, tbs, L, n_prb, 1, 6, nsf, **kw)
    tb, ok = rx.decode(tx.encode(data, 5, ack=acks, ri=ris), 5)
    assert ok.all() and np.array_equal(tb[:, :tbs // 8], data) and np.array_equal(rx.ack(), acks) and np.array_equal(rx.ri(), ris)
    tx.free()
    rx.free()


def test_ul_tx_rx_loop_harq_ack(hp):
    """Device transmit chain with HARQ-ACK into the device receive chain (noise-free): transport blocks and ACK values come back."""
    prb, L, n_prb, mod, tbs, nsf = 50, 40, 4, 2, 17568, 12
    rng = np.random.default_rng(78)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    acks = rng.integers(0, 2, (nsf, 2), dtype=np.uint8)
    tx = hp.UlTx(3, prb, 0x77, mod, tbs, L, n_prb, 1, nsf, ack_len=2, I_offset_ack=8)
    rx = hp.UlRx(3, prb, 0x77, mod, tbs, L, n_prb, 1, 6, nsf, ack_len=2, I_offset_ack=8)
    tb, ok = rx.decode(tx.encode(data, 5, ack=acks), 5)
    assert ok.all() and np.array_equal(tb[:, :tbs // 8], data) and np.array_equal(rx.ack(), acks)
    tx.free()
    rx.free()


def test_ul_tx_rx_loop(hp):
    """Device transmit chain into the device receive chain (noise-free, flat gain): every transport block comes back, one pass per block."""
    prb, L, n_prb, mod, tbs, nsf = 50, 40, 4, 2, 17568, 12
    rng = np.random.default_rng(77)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.UlTx(3, prb, 0x77, mod, tbs, L, n_prb, 1, nsf)
    rx = hp.UlRx(3, prb, 0x77, mod, tbs, L, n_prb, 1, 6, nsf)
    tb, ok = rx.decode(tx.encode(data, 5), 5)
    assert ok.all() and np.array_equal(tb[:, :tbs // 8], data)
    tx.free()
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,nrx,snr,llr8,tti0,nsf,npt", [(6, 1, 152, 1, 4.0, False, 0, 10, 2), (25, 2, 4008, 1, 11.0, False, 8, 4, 2), (25, 3, 9912, 2, 13.5, False, 3, 4, 2),
                                                                  (100, 3, 75376, 1, 19.5, False, 4, 3, 2), (50, 2, 11448, 2, 7.0, True, 9, 3, 2),
                                                                  (100, 4, 97896, 2, 24.0, False, 5, 2, 2), (6, 1, 152, 1, 5.0, False, 0, 10, 4),
                                                                  (25, 2, 4008, 2, 9.0, False, 8, 4, 4), (100, 3, 61664, 1, 19.0, False, 4, 3, 4),
                                                                  (50, 2, 11448, 2, 8.0, True, 9, 3, 4)])
def test_dl_rx_chain_tx_diversity(hp, prb, mod, tbs, nrx, snr, llr8, tti0, nsf, npt):
    """TM2 (2-port transmit diversity) receive chain on the device vs the oracle chain on identical IQ: estimates of
    both ports on every antenna, noise, SFBC-combined + layer-demapped symbols, LLRs, pass counts, CRC flags, TB bytes."""
    from lte_sim import DlConfig, make_subframe, oracle_rx
    rng = np.random.default_rng(1500 + prb + mod + nrx + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, llr8=llr8)
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(7, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, llr_8bit=llr8, nof_rx=nrx, nof_ports=npt)
    rx.keep_symbols()
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    ce = rx.debug(1, np.complex64, nsf * npt * nrx * cfg.grid_len).reshape(nsf, npt * nrx, -1)
    res = rx.debug(2, np.float32, nsf * 10).reshape(nsf, 10)
    max_re = max(rx.nof_re(s) for s in (0, 1, 5))
    d_all = rx.debug(3, np.complex64, nsf * max_re).reshape(nsf, -1)
    e_all = rx.debug(4, np.int8 if llr8 else np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    n_ok = 0
    for b in range(nsf):
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        nre = rx.nof_re((tti0 + b) % 10)
        assert nre == len(r["d"]) and nre % npt == 0
        for i in range(npt * nrx):
            assert_close_c(ce[b, i], r["ce"][i], "ce[port %d][antenna %d] sf %d" % (i // nrx, i % nrx, b))
        assert abs(res[b, 0] - r["noise"]) <= 1e-4 * abs(r["noise"])
        assert_close_c(d_all[b, :nre], r["d"], "d sf %d" % b)
        diff = np.abs(e_all[b, :nre * cfg.Qm].astype(np.int32) - r["e"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size
        assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
        if r["ok"] or diff.max() == 0:
            assert np.array_equal(tb[b], r["tb"])
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_ok > 0
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,nrx,npt,snr,llr8,tti0,nsf", [(6, 1, 152, 1, 1, 3.0, False, 0, 10), (15, 1, 1000, 1, 2, 1.5, False, 4, 4), (25, 2, 4008, 1, 1, 9.0, False, 8, 4),
                                                                  (25, 3, 9912, 2, 2, 12.5, False, 3, 4), (100, 3, 75376, 1, 1, 18.0, False, 4, 3),
                                                                  (25, 2, 4008, 1, 2, 8.5, True, 9, 3), (100, 4, 97896, 2, 1, 23.0, False, 5, 2),
                                                                  (50, 3, 11448, 2, 1, 7.5, True, 0, 3), (25, 1, 1000, 1, 1, -1.0, False, 2, 3),
                                                                  (25, 3, 7992, 2, 4, 12.5, False, 3, 4), (15, 2, 2216, 1, 4, 9.0, True, 0, 3)])
def test_dl_rx_chain_csi_weighting(hp, prb, mod, tbs, nrx, npt, snr, llr8, tti0, nsf):
    """cfg.csi_enable (the srsUE default; csi_correction, pdsch.c:574-690): per-RE channel gains and their subframe maximum from the
    equaliser kernels, the weighting applied inside the rate de-matching kernels; pass counts, CRC flags and TB bytes vs the oracle
    chain with the same switch (itself checked against the reference's srslte_pdsch_decode)."""
    from lte_sim import DlConfig, make_subframe, oracle_rx
    rng = np.random.default_rng(1700 + prb + mod + nrx + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, llr8=llr8, csi=True)
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(7, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, llr_8bit=llr8, nof_rx=nrx, nof_ports=npt, csi=True)
    tb, ok = rx.decode(np.stack(iq), tti0)
    C_ = cfg.seg.C
    it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
    max_re = max(rx.nof_re(s) for s in (0, 1, 5))
    csi = rx.debug(9, np.float32, nsf * max_re).reshape(nsf, -1)
    cmax = rx.debug(10, np.float32, nsf)
    e_all = rx.debug(4, np.int8 if llr8 else np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    n_ok = 0
    for b in range(nsf):
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        nre = rx.nof_re((tti0 + b) % 10)
        assert np.abs(csi[b, :nre] - r["csi"]).max() <= 1e-4 * r["csi"].max() and abs(cmax[b] - r["csi"].max()) <= 1e-4 * cmax[b]
        diff = np.abs(e_all[b, :nre * cfg.Qm].astype(np.int32) - r["e_raw"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size
        assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
        if r["ok"] or diff.max() == 0:
            assert np.array_equal(tb[b], r["tb"])
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_ok > 0
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,nrx,npt,snr,llr8", [(25, 2, 4008, 1, 1, 3.0, False), (100, 3, 75376, 1, 1, 17.2, False), (50, 3, 11448, 1, 2, 8.2, True),
                                                            (100, 3, 75376, 2, 2, 12.0, False), (6, 1, 152, 1, 1, -6.0, False), (100, 2, 43816, 1, 2, 7.5, False),
                                                            (50, 3, 30576, 1, 4, 12.5, False),
                                                            (15, 2, 1544, 1, 1, -2.5, True), (25, 1, 1800, 1, 1, -4.5, True)])  # K = 1568 / 1824, 8 bit: the two-block sse8 kernel with skipped blocks
def test_dl_rx_harq(hp, prb, mod, tbs, nrx, npt, snr, llr8):
    """HARQ on the device (srslte_hip_dl_rx_batch_harq): four slots, each its own transport block, sent with rv 0, 2, 3, 1 in different
    subframes with fresh noise; soft buffers, per-block CRC flags and bytes persist in the object. Per transmission and slot: CRC
    flag, per-block pass counts (0 = block carried over from an earlier transmission) and TB bytes vs the oracle's OrcHarq chain,
    which is checked against the reference's srslte_pdsch_decode + srslte_softbuffer_rx_t."""
    from lte_sim import DlConfig, OrcHarq, make_subframe, oracle_rx
    rng = np.random.default_rng(1900 + prb + mod + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, llr8=llr8)
    nsf, C_ = 4, cfg.seg.C
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(7, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, llr_8bit=llr8, nof_rx=nrx, nof_ports=npt)
    harq = [OrcHarq(cfg) for _ in range(nsf)]
    data = [None] * nsf
    done = [False] * nsf
    n_ok_first, n_ok_retx, n_carried = 0, 0, 0
    for n, (rv, tti0) in enumerate(((0, 1), (2, 8), (3, 14), (1, 23))):  # tti0 + b: 8..11 crosses subframe 0 (other nof_re)
        iq = []
        for b in range(nsf):
            x, data[b] = make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, rv=rv, data=data[b])
            iq.append(x)
        tb, ok = rx.decode_harq(np.stack(iq), tti0, rv, n == 0)
        it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
        for b in range(nsf):
            if done[b]:
                continue  # the MAC would not schedule a retransmission of an acknowledged block
            r = oracle_rx(cfg, iq[b], tti0 + b, harq=harq[b], rv=rv, new_data=n == 0)
            assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), (n, b, it[b], r["iters"])
            n_carried += int((r["iters"] == 0).sum())
            if r["ok"]:
                assert np.array_equal(tb[b], r["tb"]) and np.array_equal(tb[b][:tbs // 8], data[b])
                done[b] = True
                n_ok_first += n == 0
                n_ok_retx += n > 0
    assert n_ok_first + n_ok_retx > 0 and (n_ok_retx > 0 or llr8)
    if prb == 100 and npt == 1:
        assert n_carried > 0
    rx.free()


@pytest.mark.parametrize("seed", range(14))
def test_dl_rx_ports_antennas_csi_drawn_configurations(hp, seed):
    """The single-layer receive pipeline over its option space, drawn: 1 / 2 / 4 transmit ports (transmit diversity), 1 / 2 receive antennas,
    CSI weighting on or off (csi_correction, pdsch.c:574-690), 16- / 8-bit LLRs, ZF or MMSE, CFI, bandwidth, cell id, RNTI, modulation, a
    non-table transport-block size. LLRs (before the weighting) within one LSB of the oracle's on at most 2 in 1000; then the oracle's
    csi_correction applied to the DEVICE's LLRs with the DEVICE's per-RE gains, and its integer back end on the result: CRC flags, pass
    counts and bytes equal the device's exactly - so the weighting's rounding, its group handling and the decoder are checked bit for bit."""
    from _libs import OrcCbsegm, OrcSchCfg
    rng = np.random.default_rng(7900 + seed)
    prb, mod = int(rng.choice([6, 15, 25, 50])), int(rng.choice([1, 2, 3]))
    npt, nrx, csi, llr8, mmse = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2])), bool(seed % 3 != 1), bool(seed % 4 == 3), bool(seed % 5 != 4)
    cell_id, rnti, cfi = int(rng.integers(0, 504)), int(rng.integers(1, 0xFFF0)), int(rng.integers(1, 4))
    probe = DlConfig(prb, cell_id, mod, 16, cfi=cfi, rnti=rnti, nof_ports=npt)
    nbits = min(len(probe.indices(sf)) for sf in (0, 1, 5)) * probe.Qm
    tbs = max(40, int(float(rng.uniform(0.25, 0.75)) * nbits) // 8 * 8)
    while True:
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs -= 8
    cfg = DlConfig(prb, cell_id, mod, tbs, cfi=cfi, rnti=rnti, nof_rx=nrx, nof_ports=npt, llr8=llr8, csi=csi)
    tti0, nsf = int(rng.integers(0, 10240)), 3
    snr = {1: 1.0, 2: 7.0, 3: 12.0}[mod] + 10.0 * (tbs / nbits - 0.4) + float(rng.uniform(-2.0, 4.0)) - (3.0 if nrx == 2 else 0.0)
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(cell_id, prb, cfi, rnti, mod, tbs, 6, nsf, mmse, hc, llr_8bit=llr8, nof_rx=nrx, nof_ports=npt, csi=csi)
    tb, ok = rx.decode(np.stack(iq), tti0)
    it = rx.debug(6, np.uint32, nsf * cfg.seg.C).reshape(nsf, -1)
    max_re = max(rx.nof_re(sf) for sf in (0, 1, 5))
    e_all = rx.debug(4, np.int8 if llr8 else np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    gains = rx.debug(9, np.float32, nsf * max_re).reshape(nsf, -1) if csi else None
    n_diff = n_tot = 0
    for b in range(nsf):
        what = (seed, prb, mod, npt, nrx, csi, llr8, mmse, cfi, tbs, tti0 + b, snr)
        nre = rx.nof_re((tti0 + b) % 10)
        nb = nre * cfg.Qm
        e = np.ascontiguousarray(e_all[b, :nb])
        if mmse:  # the oracle chain always equalises with the noise estimate
            r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
            diff = np.abs(e.astype(np.int32) - r["e_raw"].astype(np.int32))
            assert diff.max() <= 1, what
            n_diff += int((diff != 0).sum())
            n_tot += nb
        if csi:
            g = np.ascontiguousarray(gains[b, :nre])
            (oracle().orc_csi_correction_b if llr8 else oracle().orc_csi_correction_s)(p(e), p(g), nre, mod)
        sch = OrcSchCfg(tbs, nb, cfg.Qm_sch, 0, cfg.max_iter)
        otb, oit, ocb = np.zeros(tbs // 8 + 16, np.uint8), np.zeros(cfg.seg.C, np.uint32), np.zeros(cfg.seg.C, np.uint8)
        rc = (oracle().orc_dlsch_decode_8bit if llr8 else oracle().orc_dlsch_decode)(C.byref(sch), p(e), p(otb), p(oit), p(ocb))
        assert bool(ok[b]) == (rc == 0) and np.array_equal(it[b], oit) and np.array_equal(tb[b], otb[:tbs // 8 + 3]), what + (bool(ok[b]), rc, it[b], oit)
        if ok[b]:
            assert np.array_equal(tb[b][:tbs // 8], data[b]), what
    assert n_diff <= 2e-3 * n_tot + 2, "LLR LSB differences on %d of %d" % (n_diff, n_tot)
    rx.free()


@pytest.mark.parametrize("seed", range(10))
def test_dl_rx_harq_drawn_sequences(hp, seed):
    """HARQ on the device with redundancy-version SEQUENCES drawn at random - any start version, repeats, up to five transmissions, a new
    transport block in the middle of the run - on drawn configurations (bandwidth, modulation, a non-table size, 16- / 8-bit LLRs). After
    every transmission, per slot: the oracle's soft-combining back end (orc_dlsch_decode_harq: rate de-matching into the kept soft buffer,
    blocks with a passed CRC skipped, sch.c:299-414) fed the DEVICE's LLRs of that transmission gives the device's CRC flag, per-block pass
    counts and bytes exactly."""
    from _libs import OrcCbsegm, OrcSchCfg
    from lte_sim import OrcHarq
    rng = np.random.default_rng(7600 + seed)
    prb, mod, llr8 = int(rng.choice([6, 15, 25, 50])), int(rng.choice([1, 2, 3])), bool(seed % 3 == 1)
    cell_id, rnti = int(rng.integers(0, 504)), int(rng.integers(1, 0xFFF0))
    probe = DlConfig(prb, cell_id, mod, 16, rnti=rnti)
    nbits = min(len(probe.indices(sf)) for sf in (0, 1, 5)) * probe.Qm
    tbs = max(40, int(float(rng.uniform(0.5, 0.95)) * nbits) // 8 * 8)  # high rates: the first transmission often fails
    while True:
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs -= 8
    cfg = DlConfig(prb, cell_id, mod, tbs, rnti=rnti, llr8=llr8)
    nsf, C_ = 3, cfg.seg.C
    snr = {1: 1.0, 2: 7.0, 3: 12.0}[mod] + 10.0 * (tbs / nbits - 0.4) - float(rng.uniform(1.0, 4.0))  # below the waterfall of one transmission
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(cell_id, prb, 1, rnti, mod, tbs, 6, nsf, True, hc, llr_8bit=llr8)
    harq = [OrcHarq(cfg) for _ in range(nsf)]
    data = [None] * nsf
    n_tx = int(rng.integers(3, 6))
    restart = int(rng.integers(1, n_tx))  # this transmission carries new data
    n_ok = n_carried = 0
    for n in range(n_tx):
        new = n == 0 or n == restart
        rv, tti0 = int(rng.integers(0, 4)), int(rng.integers(0, 10240))
        if new:
            data = [None] * nsf
        iq = []
        for b in range(nsf):
            x, data[b] = make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1, rv=rv, data=data[b])
            iq.append(x)
        tb, ok = rx.decode_harq(np.stack(iq), tti0, rv, new)
        it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
        e_all = rx.debug(4, np.int8 if llr8 else np.int16, nsf * rx.e_stride).reshape(nsf, -1)
        for b in range(nsf):
            what = (prb, mod, tbs, llr8, n, rv, new, tti0 + b, snr)
            nb = rx.nof_re((tti0 + b) % 10) * cfg.Qm
            e = np.ascontiguousarray(e_all[b, :nb])
            sch = OrcSchCfg(tbs, nb, cfg.Qm_sch, rv, cfg.max_iter)
            otb, oit, ocb = np.zeros(tbs // 8 + 16, np.uint8), np.zeros(C_, np.uint32), np.zeros(C_, np.uint8)
            rc = oracle().orc_dlsch_decode_harq(C.byref(sch), p(e), 1 if llr8 else 0, 1 if new else 0, p(harq[b].w), p(harq[b].crc), p(harq[b].data),
                                                p(otb), p(oit), p(ocb))
            assert bool(ok[b]) == (rc == 0) and np.array_equal(it[b], oit), what + (it[b], oit)
            n_carried += int((oit == 0).sum())
            if ok[b]:
                n_ok += 1
                assert np.array_equal(tb[b], otb[:tbs // 8 + 3]) and np.array_equal(tb[b][:tbs // 8], data[b]), what
    assert n_ok > 0 or llr8
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,npt,tti0,nsf,rv,p_a", [(6, 1, 152, 1, 0, 10, 0, 0.0), (25, 2, 4008, 2, 8, 4, 0, 0.0), (100, 3, 75376, 1, 4, 3, 0, -3.0),
                                                             (100, 3, 75376, 2, 9, 3, 2, 0.0), (50, 4, 48936, 1, 5, 2, 1, 0.0), (15, 1, 1000, 2, 0, 6, 3, 1.77),
                                                             (25, 2, 4008, 4, 8, 4, 0, 0.0), (100, 3, 61664, 4, 4, 3, 2, 0.0), (6, 1, 152, 4, 0, 10, 1, -1.0)])
def test_dl_tx_chain(hp, prb, mod, tbs, npt, tti0, nsf, rv, p_a, cp_ext=False):
    """eNB PDSCH transmit chain on the device vs the oracle's stimulus generator (pinned to the reference's
    srslte_pdsch_encode): per-port symbol streams exactly (bits exact, levels are table values), resource grids with CRS, time samples."""
    from lte_sim import DlConfig, make_subframe
    from _libs import OrcOfdm
    rng = np.random.default_rng(2100 + prb + mod + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_ports=npt, p_a=p_a, cp_ext=cp_ext)  # rho_a = 10^(p_a/20), x sqrt(2) for a 2-port cell (pdsch.c:525)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.DlTx(7, prb, 1, 0x1234, mod, tbs, nsf, npt, p_a, cp_ext=cp_ext)
    iq = tx.encode(data, tti0, rv)
    max_re = max(len(cfg.indices(s)) for s in (0, 1, 5))
    y = tx.debug(2, np.complex64, nsf * npt * max_re).reshape(nsf, npt, -1)
    grid = tx.debug(3, np.complex64, nsf * npt * cfg.grid_len).reshape(nsf, npt, -1)
    q = OrcOfdm()
    oracle().orc_ofdm_init(C.byref(q), prb, cfg.cp_norm)
    q.normalize = True
    for b in range(nsf):
        k = {}
        make_subframe(cfg, tti0 + b, rng, rv=rv, data=data[b], keep=k)
        for port in range(npt):
            ye = k["y"][port]
            assert np.abs(y[b, port, :len(ye)] - ye).max() <= 3e-7 * max(1.0, cfg.scaling), (b, port)
            exp = np.zeros(cfg.grid_len, np.complex64)
            exp[k["idx"]] = ye
            oracle().orc_crs_put_sf(C.byref(cfg.cell), (tti0 + b) % 10, port, p(exp))
            assert np.abs(grid[b, port] - exp).max() <= 3e-7 * max(1.0, cfg.scaling), (b, port)
            iq_o = np.zeros(cfg.sf_len, np.complex64)
            oracle().orc_ofdm_tx_sf(C.byref(q), p(exp), p(iq_o))
            assert_close_c(iq[b, port], iq_o, "iq sf %d port %d" % (b, port))
    tx.free()


@pytest.mark.parametrize("prb,mod,tbs,npt,tti0,nsf,rv", [(6, 1, 152, 1, 0, 10, 0), (25, 2, 4008, 1, 8, 4, 2), (100, 3, 43816, 1, 4, 3, 0), (50, 3, 11448, 2, 5, 2, 1),
                                                         (15, 1, 1000, 2, 0, 6, 3)])
def test_dl_tx_chain_extended_cp(hp, prb, mod, tbs, npt, tti0, nsf, rv):
    """The transmit pipeline on an extended-CP cell (cfg.cp_ext): [12][12 nof_prb] grids, CRS with the extended-CP sequences on symbols 0 and 3
    of each slot, the long cyclic prefix (ofdm.c:558-574), against the oracle's generator - whose signal the reference's own
    srslte_pdsch_decode takes on an extended-CP cell (test_pdsch_decode_extended_cp_vs_oracle_chain); and back through the receive pipeline."""
    test_dl_tx_chain(hp, prb, mod, tbs, npt, tti0, nsf, rv, 0.0, cp_ext=True)
    if npt == 1:
        rng = np.random.default_rng(prb)
        data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
        tx = hp.DlTx(7, prb, 1, 0x1234, mod, tbs, nsf, 1, 0.0, cp_ext=True)
        iq = tx.encode(data, tti0, 0)
        hc = hp.ChestDlCfg()
        hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
        rx = hp.DlRx(7, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, cp_ext=True)
        tb, ok = rx.decode(iq[:, 0], tti0)
        assert ok.all() and np.array_equal(tb[:, :tbs // 8], data)
        tx.free()
        rx.free()


@pytest.mark.parametrize("prb", [7, 20, 33, 64, 91, 110])
def test_tx_chains_any_bandwidth(hp, prb):
    """Both transmit chains at cell bandwidths between the six of 36.101."""
    test_dl_tx_chain_drawn_configurations(hp, 100 + prb, prb)
    test_ul_tx_chain_drawn_configurations(hp, 100 + prb, prb)


@pytest.mark.parametrize("seed", range(10))
def test_dl_tx_chain_drawn_configurations(hp, seed, force_prb=None):
    """test_dl_tx_chain on configurations drawn from what the transmit pipeline accepts: bandwidth, cell id, RNTI, 1 / 2 / 4 ports, modulation, a
    transport-block size not taken from a table (one block length, no filler) at a drawn code rate, redundancy version, power offset, first TTI."""
    from lte_sim import DlConfig, make_subframe
    from _libs import OrcCbsegm, OrcOfdm
    rng = np.random.default_rng(7200 + seed)
    prb, mod, npt = int(rng.choice([6, 15, 25, 50, 100])), int(rng.choice([1, 2, 3, 4])), int(rng.choice([1, 2, 4]))
    prb = force_prb or prb
    cell_id, rnti, rv = int(rng.integers(0, 504)), int(rng.integers(1, 0xFFF0)), int(rng.integers(0, 4))
    p_a = float(rng.choice([0.0, -3.0, 1.77, -1.0]))
    probe = DlConfig(prb, cell_id, mod, 16, rnti=rnti, nof_ports=npt)
    nbits = min(len(probe.indices(s)) for s in (0, 1, 5)) * probe.Qm
    tbs = max(40, int(float(rng.uniform(0.2, 0.85)) * nbits) // 8 * 8)
    while True:
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs -= 8
    cfg = DlConfig(prb, cell_id, mod, tbs, rnti=rnti, nof_ports=npt, p_a=p_a)
    tti0, nsf = int(rng.integers(0, 10240)), 3
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.DlTx(cell_id, prb, 1, rnti, mod, tbs, nsf, npt, p_a)
    iq = tx.encode(data, tti0, rv)
    max_re = max(len(cfg.indices(s)) for s in (0, 1, 5))
    y = tx.debug(2, np.complex64, nsf * npt * max_re).reshape(nsf, npt, -1)
    grid = tx.debug(3, np.complex64, nsf * npt * cfg.grid_len).reshape(nsf, npt, -1)
    q = OrcOfdm()
    oracle().orc_ofdm_init(C.byref(q), prb, True)
    q.normalize = True
    for b in range(nsf):
        k = {}
        make_subframe(cfg, tti0 + b, rng, rv=rv, data=data[b], keep=k)
        for port in range(npt):
            what, ye = (prb, mod, npt, cell_id, tbs, rv, p_a, tti0 + b, port), k["y"][port]
            assert np.abs(y[b, port, :len(ye)] - ye).max() <= 3e-7 * max(1.0, cfg.scaling), what
            exp = np.zeros(cfg.grid_len, np.complex64)
            exp[k["idx"]] = ye
            oracle().orc_crs_put_sf(C.byref(cfg.cell), (tti0 + b) % 10, port, p(exp))
            assert np.abs(grid[b, port] - exp).max() <= 3e-7 * max(1.0, cfg.scaling), what
            iq_o = np.zeros(cfg.sf_len, np.complex64)
            oracle().orc_ofdm_tx_sf(C.byref(q), p(exp), p(iq_o))
            assert_close_c(iq[b, port], iq_o, "iq %s" % (what,))
    tx.free()


@pytest.mark.parametrize("npt,nrx", [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2)])
def test_dl_tx_rx_loop(hp, npt, nrx):
    """Device transmit chain into the device receive chain (noise-free; the ports of a 2-port cell reach the antennas with different
    flat gains): every transport block comes back; then new blocks under noise that no block survives, their retransmission (rv 2,
    noise-free) into the kept soft buffers, which delivers them all; and a duplicate retransmission of the delivered blocks, which is
    refused as upstream's decode_tb_cb refuses it (sch.c:399-410: CRC flag 0, see the HARQ section of DESIGN.md)."""
    prb, mod, tbs, nsf = 50, 3, 30576, 12
    rng = np.random.default_rng(78)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.DlTx(5, prb, 1, 0x4321, mod, tbs, nsf, npt)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(5, prb, 1, 0x4321, mod, tbs, 6, nsf, True, hc, nof_rx=nrx, nof_ports=npt, power_scale=True, p_a=0.0)  # phy_dl_test.c:176-178,:219-221
    gains = np.array([[1.0, 0.7 * np.exp(1.1j), 0.9 * np.exp(-2.0j), 0.6 * np.exp(0.4j)],
                      [0.8 * np.exp(-0.6j), 0.9 * np.exp(2.2j), 0.7 * np.exp(1.5j), 1.1 * np.exp(-1.2j)]], np.complex64)  # [antenna][port]
    delivered = np.zeros(nsf, bool)
    for step, (rv, new, noisy) in enumerate(((0, True, False), (0, True, True), (2, False, False), (3, False, False))):
        iq = tx.encode(data, 3, rv)  # [nsf][npt][sf_len]
        ant = np.stack([sum(gains[a, port] * iq[:, port] for port in range(npt)) for a in range(nrx)], axis=1)  # [nsf][nrx][sf_len]
        if noisy:  # about 10 dB: below the waterfall of this rate, the soft buffers still hold something to combine with
            ant = ant + np.std(ant) * 0.3 * ((rng.standard_normal(ant.shape) + 1j * rng.standard_normal(ant.shape)) / np.sqrt(2))
        tb, ok = rx.decode_harq(np.ascontiguousarray(ant, np.complex64), 3, rv, new)
        ok = ok.astype(bool)
        if new:
            delivered[:] = False
        assert not ok[delivered].any(), (step, rv, ok)  # duplicates of delivered blocks are refused
        if not noisy:
            assert ok[~delivered].all(), (step, rv, ok)  # a clean (re)transmission delivers what was still missing
        else:
            assert not ok.all(), (step, rv, ok)
        assert np.array_equal(tb[ok, :tbs // 8], data[ok]), (step, rv)
        delivered |= ok
    tx.free()
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,nrx,npt,snr,p_a", [(25, 2, 4008, 1, 1, 9.0, -3.0), (25, 3, 9912, 2, 2, 13.5, 0.0), (100, 3, 75376, 1, 2, 19.5, 0.0)])
def test_dl_rx_chain_power_scaling(hp, prb, mod, tbs, nrx, npt, snr, p_a):
    """cfg.power_scale / p_a (pdsch.c:518-554,:852-858 with rho_b = 1) vs the oracle chain with the same setting (checked against the
    reference's srslte_pdsch_decode): equalised symbols, LLRs, pass counts, CRC flags, TB bytes."""
    from lte_sim import DlConfig, make_subframe, oracle_rx
    rng = np.random.default_rng(2300 + prb + mod + npt)
    cfg = DlConfig(prb, 7, mod, tbs, nof_rx=nrx, nof_ports=npt, p_a=p_a)
    nsf, tti0 = 3, 4
    iq, data = zip(*[make_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.1) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(7, prb, 1, 0x1234, mod, tbs, 6, nsf, True, hc, nof_rx=nrx, nof_ports=npt, power_scale=True, p_a=p_a)
    rx.keep_symbols()
    tb, ok = rx.decode(np.stack(iq), tti0)
    it = rx.debug(6, np.uint32, nsf * cfg.seg.C).reshape(nsf, -1)
    max_re = max(rx.nof_re(s) for s in (0, 1, 5))
    d_all = rx.debug(3, np.complex64, nsf * max_re).reshape(nsf, -1)
    e_all = rx.debug(4, np.int16, nsf * rx.e_stride).reshape(nsf, -1)
    n_ok = 0
    for b in range(nsf):
        r = oracle_rx(cfg, iq[b], tti0 + b, keep=True)
        nre = rx.nof_re((tti0 + b) % 10)
        assert_close_c(d_all[b, :nre], r["d"], "d sf %d" % b)
        diff = np.abs(e_all[b, :nre * cfg.Qm].astype(np.int32) - r["e"].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size
        assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b], r["tb"]) and np.array_equal(tb[b][:tbs // 8], data[b])
    assert n_ok > 0
    rx.free()
