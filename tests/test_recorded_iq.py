"""Pins the ORACLE's OFDM demodulator (oracle/orc_sig.c, the restatement of lib/src/phy/dft/ofdm.c) to data the reference holds: its
recorded-IQ captures, decoded by the reference's own compiled channel decoders, reach what the reference's CTests assert. The
reference's ofdm.c cannot be built here (FFTW absent), so this is the pin of rows a2/a3 that does not go through the oracle's own
modulator. Needs oracle/_ref (built where /root/reference exists; the prebuilt libraries travel to the GPU box)."""
import ctypes as C

import numpy as np
import pytest

import recorded_iq
import refdrv
from _libs import OrcOfdm, oracle, p

pytestmark = pytest.mark.skipif(refdrv.lib() is None, reason="oracle/_ref not built (no /root/reference on this machine)")


def orc_ofdm_rx(nof_prb, cp_norm, iq, region, exact=False):
    q = OrcOfdm()
    assert oracle().orc_ofdm_init(C.byref(q), nof_prb, cp_norm) == 0
    q.non_mbsfn_region, q.exact = region, exact
    out = np.zeros((14 if cp_norm else 12) * 12 * nof_prb, np.complex64)
    oracle().orc_ofdm_rx_sf(C.byref(q), p(np.ascontiguousarray(iq, np.complex64)), p(out))
    return out


def test_pdsch_pdcch_file():
    res = recorded_iq.pdsch_pdcch_file(orc_ofdm_rx)
    assert all(r["cfi"] == 3 and r["cfi_corr"] > 30 for r in res)  # the capture's CFI (the CTest passes -f 3)
    hits = [r for r in res if r["dci"]]
    assert [r["sf"] for r in hits] == [2, 5]  # the reference's loop exits at subframe 2 with ret = 1
    assert all(r["crc"] for r in hits)
    assert hits[0]["grant"] == {"mcs": 6, "tbs": 256, "nof_prb": 6, "rv": 3} and hits[1]["grant"] == {"mcs": 2, "tbs": 144, "nof_prb": 6, "rv": 0}


def test_pdsch_pdcch_file_detects_a_wrong_demodulator():
    """What the pin is worth: the same chain on grids with the faults an OFDM restatement could have."""
    nre = 72

    def half_swap(g):  # low and high half of the band exchanged (ofdm.c:411-412 the wrong way round)
        g = g.reshape(14, nre)
        return np.concatenate([g[:, nre // 2:], g[:, :nre // 2]], axis=1).ravel()

    def dc_not_skipped(g):  # high half read from bin 0 instead of bin 1
        g = g.reshape(14, nre).copy()
        g[:, nre // 2 + 1:] = g[:, nre // 2:-1]
        return g.ravel()

    def late_by_one_symbol_cp(nof_prb, cp_norm, iq, region):  # every symbol taken a CP length late: inter-symbol interference
        return orc_ofdm_rx(nof_prb, cp_norm, np.roll(iq, -9), region)

    assert not any(r["dci"] for r in recorded_iq.pdsch_pdcch_file(orc_ofdm_rx, half_swap))
    assert not any(r["dci"] for r in recorded_iq.pdsch_pdcch_file(late_by_one_symbol_cp))
    res = recorded_iq.pdsch_pdcch_file(orc_ofdm_rx, dc_not_skipped)  # one bin off in half the band: the PDCCH of subframe 2 survives, no PDSCH does
    assert not any(r["dci"] and r["crc"] for r in res) and all(r["cfi_corr"] < 30 for r in res)


def test_oracle_rx_equals_the_3gpp_definition():
    """Third, reference-independent view of the same demodulator: 36.211 6.12 written out with numpy's FFT (symbol l starts after the
    CPs and symbols before it, sub-carrier k of the grid is bin k - N_re/2 for the lower half and k - N_re/2 + 1 for the upper half)."""
    for prb, name, cp_norm in ((6, "signal.1.92M.amar.dat", True), (100, "pmch_100prbs_MCS2_SR0.bin", False)):
        N = oracle().orc_symbol_sz(prb)
        iq = refdrv.read_iq(name, 15 * N)
        got = orc_ofdm_rx(prb, cp_norm, iq, 0, exact=True).reshape(-1, 12 * prb)
        nre, pos = 12 * prb, 0
        for l in range(14 if cp_norm else 12):
            cpl = -(-(160 if l % 7 == 0 else 144) * N // 2048) if cp_norm else 512 * N // 2048
            X = np.fft.fft(iq[pos + cpl:pos + cpl + N].astype(np.complex128))
            want = np.concatenate([X[N - nre // 2:], X[1:nre // 2 + 1]])
            assert np.abs(got[l] - want).max() <= 1e-5 * np.abs(want).max()
            pos += cpl + N


def test_pcfich_file():
    n, cfi, corr, _ = recorded_iq.pcfich_file(orc_ofdm_rx)
    assert n == 1 and cfi == 2 and corr > 2.8  # pcfich_file_test.c:252-256


def test_pbch_file():
    n, ports, off, bch = recorded_iq.pbch_file(orc_ofdm_rx)
    assert n == 1 and ports == 2 and off == 0 and list(bch) == recorded_iq.BCH_PAYLOAD_FILE  # pbch_file_test.c:229-233


@pytest.mark.parametrize("exact", [False, True])
def test_pmch_file(exact):
    r = recorded_iq.pmch_file(lambda *a: orc_ofdm_rx(*a, exact=exact))
    assert r["crc"] == 1 and r["tbs"] == 4584  # "PMCH Decoded OK!"
    assert r["cfi"] == 2 and r["cfi_corr"] > 10  # the PCFICH of the non-MBSFN region (normal-CP symbol 0)


def test_pmch_file_detects_a_wrong_mbsfn_layout():
    """The PMCH occupies symbols 2.. of the subframe, which sit on the extended-CP raster in every layout; what the non-MBSFN region of
    ofdm.c:424-437 moves are symbols 0 and 1 (normal CP, then the guard). A plain extended-CP demodulator loses the PCFICH of symbol 0,
    and one that starts the extended-CP part a guard early or late loses the PMCH."""
    r = recorded_iq.pmch_file(lambda prb, cpn, iq, region: orc_ofdm_rx(prb, cpn, iq, 0))
    assert r["cfi"] != 2 and r["cfi_corr"] < 8
    r = recorded_iq.pmch_file(lambda prb, cpn, iq, region: orc_ofdm_rx(prb, cpn, np.roll(iq, -540), region))  # no guard
    assert r["crc"] == 0


@pytest.mark.parametrize("prb", [6, 15, 25, 50, 75, 100])
@pytest.mark.parametrize("cp_norm", [True, False])
def test_ofdm_tx_against_the_pinned_rx(prb, cp_norm):
    """a3: with the demodulator pinned above, the modulator is right iff (i) demodulating its output returns the grid - ofdm_test.c:74-179,
    normal and extended CP (`ofdm_normal`, `ofdm_extended`, dft/test/CMakeLists.txt:28-32) - and (ii) every CP is the copy of its
    symbol's tail (ofdm.c:527), which the demodulator never looks at."""
    rng = np.random.default_rng(prb)
    q = OrcOfdm()
    assert oracle().orc_ofdm_init(C.byref(q), prb, cp_norm) == 0
    q.normalize = True
    nsym = 14 if cp_norm else 12
    g = (rng.standard_normal(nsym * 12 * prb) + 1j * rng.standard_normal(nsym * 12 * prb)).astype(np.complex64)
    t = np.zeros(q.sf_sz, np.complex64)
    oracle().orc_ofdm_tx_sf(C.byref(q), p(g), p(t))
    back = np.zeros_like(g)
    oracle().orc_ofdm_rx_sf(C.byref(q), p(t), p(back))
    assert np.mean(np.abs(back - g) ** 2) < 1e-9  # ofdm_test.c:155 accepts 0.07
    N, pos = q.symbol_sz, 0
    for s in range(nsym):
        cpl = oracle().orc_cp_len_norm(s % 7, N) if cp_norm else oracle().orc_cp_len_ext(N)
        assert np.array_equal(t[pos:pos + cpl], t[pos + N:pos + N + cpl])
        pos += cpl + N
    assert pos == q.sf_sz
