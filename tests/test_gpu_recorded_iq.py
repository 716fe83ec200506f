"""Rows a2/a3 on the device, pinned to reference-held data: the reference's recorded-IQ captures (tests/golden/iq/) demodulated by the
HIP OFDM kernels - through the reference's own single-call API srslte_ofdm_rx_init[_mbsfn] / srslte_ofdm_rx_sf and through the batched
API - give (i) the oracle's grid to 1e-4 and (ii), handed to the reference's compiled channel decoders (oracle/_ref, which travels to
the GPU box prebuilt), exactly what the reference's CTests assert (lib/src/phy/phch/test/CMakeLists.txt:233-238). The device channel
estimator is checked on the same real signals against the reference's srslte_chest_dl_estimate_cfg output."""
import ctypes as C
import importlib

import numpy as np
import pytest

import recorded_iq
import refdrv
from _libs import aligned, hip, opaque, p
from test_recorded_iq import orc_ofdm_rx

pytestmark = pytest.mark.gpu
need_ref = pytest.mark.skipif(refdrv.lib() is None, reason="oracle/_ref did not travel with the repo")


def close(a, b, tol=1e-4):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), np.sqrt((np.abs(b) ** 2).mean()))


def hip_ofdm_rx_compat(nof_prb, cp_norm, iq, region):
    """srslte_ofdm_rx_init / _init_mbsfn + srslte_ofdm_rx_sf on caller buffers, as ue_dl.c:88-104,369-384 uses them."""
    L = hip()
    nsym = 14 if cp_norm else 12
    tbuf, gbuf = aligned(2 * len(iq), np.float32), aligned(2 * nsym * 12 * nof_prb, np.float32)
    q = opaque(4096)
    if region:
        assert L.srslte_ofdm_rx_init_mbsfn(q, 1, p(tbuf), p(gbuf), nof_prb) == 0
        L.srslte_ofdm_set_non_mbsfn_region(q, region)
    else:
        assert L.srslte_ofdm_rx_init(q, 0 if cp_norm else 1, p(tbuf), p(gbuf), nof_prb) == 0
    tbuf.view(np.complex64)[:] = iq
    L.srslte_ofdm_rx_sf(q)
    out = gbuf.view(np.complex64).copy()
    L.srslte_ofdm_rx_free(q)
    return out


def hip_ofdm_rx_batched(nof_prb, cp_norm, iq, region):
    assert region == 0
    hp = importlib.import_module("srslte-emane_amd")
    o = hp.Ofdm(nof_prb, cp_norm, rx=True)
    out = o.rx_sf(iq)[0]
    o.free()
    return out


@pytest.mark.parametrize("name,prb,cp_norm,region,nsf", [("signal.1.92M.amar.dat", 6, True, 0, 10), ("signal.1.92M.dat", 6, True, 0, 5),
                                                         ("signal.10M.dat", 50, True, 0, 1), ("pmch_100prbs_MCS2_SR0.bin", 100, False, 2, 1),
                                                         ("pmch_100prbs_MCS2_SR0.bin", 100, False, 0, 1)])
def test_grid_equals_oracle(name, prb, cp_norm, region, nsf):
    N = hip().srslte_symbol_sz(prb)
    for sf in range(nsf):
        iq = refdrv.read_iq(name, 15 * N, sf * 15 * N)
        want = orc_ofdm_rx(prb, cp_norm, iq, region, exact=True)
        assert close(hip_ofdm_rx_compat(prb, cp_norm, iq, region), want)
        if not region:
            assert close(hip_ofdm_rx_batched(prb, cp_norm, iq, region), want)


@need_ref
@pytest.mark.parametrize("rx", [hip_ofdm_rx_compat, hip_ofdm_rx_batched])
def test_pdsch_pdcch_file(rx):
    res, want = recorded_iq.pdsch_pdcch_file(rx), recorded_iq.pdsch_pdcch_file(orc_ofdm_rx)
    hits = [r for r in res if r["dci"]]
    assert [r["sf"] for r in hits] == [2, 5] and all(r["crc"] for r in hits) and all(r["cfi"] == 3 for r in res)
    for a, b in zip(res, want):  # and the same transport blocks as on the oracle's grid
        assert a["grant"] == b["grant"] and a["crc"] == b["crc"] and (a["tb"] is None or np.array_equal(a["tb"], b["tb"]))
        assert abs(a["cfi_corr"] - b["cfi_corr"]) < 1e-2


@need_ref
def test_pcfich_pbch_pmch_files():
    n, cfi, corr, _ = recorded_iq.pcfich_file(hip_ofdm_rx_compat)
    assert n == 1 and cfi == 2 and corr > 2.8
    n, ports, off, bch = recorded_iq.pbch_file(hip_ofdm_rx_batched)
    assert n == 1 and ports == 2 and off == 0 and list(bch) == recorded_iq.BCH_PAYLOAD_FILE
    r, w = recorded_iq.pmch_file(hip_ofdm_rx_compat), recorded_iq.pmch_file(orc_ofdm_rx)
    assert r["crc"] == 1 and r["tbs"] == 4584 and r["cfi"] == 2 and np.array_equal(r["tb"], w["tb"])


@need_ref
def test_chest_dl_on_the_captures():
    """Device srslte_chest_dl_estimate_cfg on real signals (the synthetic channels of the other tests are smooth by construction):
    default configuration = automatic Gauss filter from the previous noise estimate, time-averaged pilots (chest_dl.c:598-673)."""
    hp = importlib.import_module("srslte-emane_amd")
    ref = recorded_iq.pdsch_pdcch_file(orc_ofdm_rx)
    est = hp.ChestDl(1, 6)
    for sf in range(10):  # one subframe per call: the automatic filter depends on the estimate the previous call left in the object
        grid = orc_ofdm_rx(6, True, refdrv.read_iq("signal.1.92M.amar.dat", 1920, sf * 1920), 0)
        ce, res = est.estimate(grid, sf, hp.ChestDlCfg())
        assert close(ce[0], ref[sf]["ce"], 2e-4), sf
        assert abs(res["noise_estimate"][0] - ref[sf]["noise"]) <= 1e-4 * ref[sf]["noise"]
    # 2-port cell, 50 PRB, capture shorter than a subframe
    _, _, _, ce_ref = recorded_iq.pcfich_file(orc_ofdm_rx)
    est2 = hp.ChestDl(150, 50, nof_ports=2)
    grid = orc_ofdm_rx(50, True, refdrv.read_iq("signal.10M.dat", 15 * 768), 0)
    rc, ce, _, _ = est2.estimate_multi(grid, 0, hp.ChestDlCfg())
    assert rc == 0
    for port in range(2):
        assert close(ce[0, port, 0], ce_ref[port], 2e-4), port
    # MBSFN subframe of the 100-PRB capture: triangle filter, interpolate_subframe (pmch_file_test.c:170-178); the reference's ce of
    # symbols 0..11 (the 12 symbols of the extended-CP subframe)
    w = recorded_iq.pmch_file(orc_ofdm_rx)
    est3 = hp.ChestDl(1, 100, cp_norm=False)
    assert est3.set_mbsfn_area_id(1) == 0
    cfg = hp.ChestDlCfg()
    cfg.noise_alg, cfg.filter_type, cfg.interpolate_subframe, cfg.mbsfn_area_id = 1, 1, 1, 1
    cfg.filter_coef[0] = 0.1
    grid = orc_ofdm_rx(100, False, refdrv.read_iq("pmch_100prbs_MCS2_SR0.bin", 23040), 2)
    rc, ce, _ = est3.estimate_mbsfn(grid, 1, cfg)
    assert rc == 0 and close(ce[0, 0, 0][:12 * 1200], w["ce"][:12 * 1200], 2e-4)


@pytest.mark.parametrize("prb", [6, 15, 25, 50, 75, 100])
def test_ofdm_extended_cp(prb):
    """`ofdm_extended` (dft/test/CMakeLists.txt:29,32: ofdm_test -e): normal subframes of an extended-CP cell, 12 symbols, CP 512·N/2048,
    modulator and demodulator against the oracle and the round trip of ofdm_test.c:74-179."""
    from _libs import OrcOfdm, oracle
    hp = importlib.import_module("srslte-emane_amd")
    rng = np.random.default_rng(prb)
    q = OrcOfdm()
    assert oracle().orc_ofdm_init(C.byref(q), prb, False) == 0
    q.normalize, q.exact = True, True
    nsf, glen = 2, 12 * 12 * prb
    grid = (rng.standard_normal((nsf, glen)) + 1j * rng.standard_normal((nsf, glen))).astype(np.complex64)
    tx, rx = hp.Ofdm(prb, False, rx=False), hp.Ofdm(prb, False, rx=True)
    assert tx.sf_len == q.sf_sz and tx.grid_len == glen
    tx.set_normalize(True)
    rx.set_normalize(True)
    t_gpu, t_ref = tx.tx_sf(grid), np.zeros((nsf, q.sf_sz), np.complex64)
    for i in range(nsf):
        oracle().orc_ofdm_tx_sf(C.byref(q), p(grid[i]), p(t_ref[i]))
    assert close(t_gpu, t_ref)
    time_in = (rng.standard_normal((nsf, q.sf_sz)) + 1j * rng.standard_normal((nsf, q.sf_sz))).astype(np.complex64)
    g_gpu, g_ref = rx.rx_sf(time_in), np.zeros((nsf, glen), np.complex64)
    for i in range(nsf):
        oracle().orc_ofdm_rx_sf(C.byref(q), p(time_in[i]), p(g_ref[i]))
    assert close(g_gpu, g_ref)
    assert np.mean(np.abs(rx.rx_sf(t_gpu) - grid) ** 2) < 1e-9
    tx.free()
    rx.free()
