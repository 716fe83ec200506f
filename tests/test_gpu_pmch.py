"""The PMCH chain as ONE fused pipeline in each direction (SURVEY §8f N4's pmch_test; VERDICT r3 missing item 4): cfg.mbsfn of srslte_hip_dl_rx_* /
srslte_hip_dl_tx_* - MBSFN OFDM layout (ofdm.c:424-437,:558-574), the MBSFN channel estimate (chest_dl.c:718-745), pmch_cp's RE mapping
(pmch.c:44-99), the area's scrambling sequence (sequences.c:76-80) and the DL-SCH coder - against the oracle chain that
tests/test_oracle_vs_ref.py::test_pmch_encode_decode_vs_oracle_chain pins on the reference's srslte_pmch_encode / srslte_pmch_decode."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    return importlib.import_module("srslte-emane_amd")


def close_c(a, b, what, tol=1e-4):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    ref = max(np.abs(b).max(), np.sqrt((np.abs(b) ** 2).mean()))
    assert np.abs(a - b).max() <= tol * ref, what


# prb, cell_id, area, mod, tbs, cfi, region, snr, nrx, cp_ext, tti0, nsf
CASES = [(6, 1, 1, 1, 488, 2, 2, 7.0, 1, True, 1, 4), (25, 7, 3, 2, 4584, 2, 2, 12.0, 1, True, 6, 4), (50, 101, 200, 3, 15264, 1, 1, 19.0, 1, False, 2, 4),
         (100, 301, 77, 2, 18336, 2, 2, 11.0, 2, False, 7, 3), (15, 44, 255, 2, 2216, 1, 1, 12.0, 1, True, 11, 4), (100, 12, 5, 3, 36696, 2, 2, 19.5, 1, False, 1, 3),
         (75, 500, 0, 1, 6200, 2, 1, 3.0, 4, False, 3, 3)]


@pytest.mark.parametrize("prb,cell_id,area,mod,tbs,cfi,region,snr,nrx,cp_ext,tti0,nsf", CASES)
def test_pmch_rx_pipeline(hp, prb, cell_id, area, mod, tbs, cfi, region, snr, nrx, cp_ext, tti0, nsf):
    """IQ of MBSFN subframes -> transport blocks on the device vs the oracle chain on identical noisy IQ: grid, MBSFN estimate, noise figure, LLRs
    within one LSB (<= 0.2 %), per-block pass counts, CRC verdicts and bytes."""
    from lte_sim import PmchConfig, make_pmch_subframe, oracle_pmch_rx
    rng = np.random.default_rng(6100 + prb + area)
    cfg = PmchConfig(prb, cell_id, area, mod, tbs, cfi=cfi, non_mbsfn_region=region, nof_rx=nrx, cp_ext=cp_ext)
    iq, data = zip(*[make_pmch_subframe(cfg, tti0 + b, rng, snr_db=snr, amp=0.05 / np.sqrt(prb) * 20) for b in range(nsf)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0  # interpolate_subframe is implied by cfg.mbsfn
    rx = hp.DlRx(cell_id, prb, cfi, 0, mod, tbs, 6, nsf, True, hc, nof_rx=nrx, cp_ext=cp_ext, mbsfn=(area, region))
    assert rx.nof_re(1) == rx.nof_re(0) == cfg.nof_re
    for rep in range(2):  # the object is reusable
        tb, ok = rx.decode(np.stack(iq), tti0)
        C_ = cfg.seg.C
        it = rx.debug(6, np.uint32, nsf * C_).reshape(nsf, C_)
        grid = rx.debug(0, np.complex64, nsf * nrx * cfg.grid_len).reshape(nsf, nrx, -1)
        ce = rx.debug(1, np.complex64, nsf * nrx * cfg.grid_len).reshape(nsf, nrx, -1)
        res = rx.debug(2, np.float32, nsf * 10).reshape(nsf, 10)
        e_all = rx.debug(4, np.int16, nsf * rx.e_stride).reshape(nsf, -1)
        n_diff = n_tot = n_ok = 0
        for b in range(nsf):
            r = oracle_pmch_rx(cfg, iq[b], tti0 + b, keep=True)
            close_c(grid[b], r["grid"], "grid sf %d" % b)
            close_c(ce[b], r["ce"], "ce sf %d" % b)
            assert abs(res[b, 0] - r["noise"]) <= 1e-4 * abs(r["noise"]), (b, res[b, 0], r["noise"])
            diff = np.abs(e_all[b, :cfg.nbits].astype(np.int32) - r["e"].astype(np.int32))
            assert diff.max() <= 1, "LLR differs by more than 1 LSB (sf %d: %d)" % (b, diff.max())
            n_diff += int((diff != 0).sum())
            n_tot += diff.size
            if diff.max() == 0 or r["ok"]:
                assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), "sf %d" % b
            if r["ok"]:
                assert np.array_equal(tb[b][:tbs // 8 + 3], r["tb"]) and np.array_equal(tb[b][:tbs // 8], data[b])
                n_ok += 1
        assert n_diff <= 2e-3 * n_tot + 1 and n_ok > 0, (n_diff, n_tot, n_ok)
    rx.free()


@pytest.mark.parametrize("prb,cell_id,area,mod,tbs,cfi,region,snr,nrx,cp_ext,tti0,nsf", CASES)
def test_pmch_tx_pipeline(hp, prb, cell_id, area, mod, tbs, cfi, region, snr, nrx, cp_ext, tti0, nsf):
    """Transport blocks -> IQ of MBSFN subframes on the device vs the oracle's generator (srslte_pmch_encode + srslte_refsignal_mbsfn_put_sf +
    the MBSFN OFDM layout): modulated symbols bit for bit, the grid with both reference signals, the time samples; and back through the receive
    pipeline without noise."""
    from lte_sim import PmchConfig, make_pmch_subframe
    rng = np.random.default_rng(6300 + prb + area)
    cfg = PmchConfig(prb, cell_id, area, mod, tbs, cfi=cfi, non_mbsfn_region=region, cp_ext=cp_ext)
    data = rng.integers(0, 256, (nsf, tbs // 8), dtype=np.uint8)
    tx = hp.DlTx(cell_id, prb, cfi, 0, mod, tbs, nsf, 1, 0.0, cp_ext=cp_ext, mbsfn=(area, region))
    iq = tx.encode(data, tti0, 0)[:, 0]
    y = tx.debug(2, np.complex64, nsf * cfg.nof_re).reshape(nsf, -1)
    grid = tx.debug(3, np.complex64, nsf * cfg.grid_len).reshape(nsf, -1)
    for b in range(nsf):
        k = {}
        iq_o, _ = make_pmch_subframe(cfg, tti0 + b, rng, data=data[b], keep=k)
        assert np.array_equal(y[b].view(np.float32), k["d"].view(np.float32)), "symbols sf %d" % b
        close_c(grid[b], k["grid"], "grid sf %d" % b, 1e-6)
        close_c(iq[b], iq_o, "iq sf %d" % b)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(cell_id, prb, cfi, 0, mod, tbs, 6, nsf, True, hc, cp_ext=cp_ext, mbsfn=(area, region))
    tb, ok = rx.decode(iq, tti0)
    assert ok.all() and np.array_equal(tb[:, :tbs // 8], data)
    tx.free()
    rx.free()


def test_pmch_pipeline_configuration_errors(hp):
    """What srslte_pmch_* cannot be asked either: more than one port (pmch.c:159), 8-bit LLRs, CSI weighting, a TDD cell, an area id above 255,
    a non-MBSFN region other than 1 or 2 symbols, per-subframe grants on a PMCH object."""
    hc = hp.ChestDlCfg()
    ok = dict(cell_id=1, nof_prb=25, cfi=2, rnti=0, mod=2, tbs=4584, max_iterations=6, max_batch=2, mmse=True, chest_cfg=hc)
    for bad in (dict(nof_ports=2), dict(llr_8bit=True), dict(csi=True), dict(tdd=(1, 4)), dict(mbsfn=(256, 2)), dict(mbsfn=(1, 0)), dict(mbsfn=(1, 3))):
        kw = dict(ok, mbsfn=(1, 2))
        kw.update(bad)
        with pytest.raises(RuntimeError):
            hp.DlRx(**kw)
    with pytest.raises(RuntimeError):
        hp.DlTx(1, 25, 2, 0, 2, 4584, 2, 2, 0.0, mbsfn=(1, 2))
    rx = hp.DlRx(**dict(ok, mbsfn=(1, 2)))
    g = hp.DlGrant.make(25, 2, 4584, 0x1234, cfi=2)
    assert rx.decode_grants(np.zeros((1, rx.sf_len), np.complex64), 0, [g])[0] == hp.SRSLTE_ERROR
    rx.free()
