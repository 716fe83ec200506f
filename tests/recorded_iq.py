"""The reference's recorded-IQ CTests (lib/src/phy/phch/test/CMakeLists.txt:233-238) re-run on a caller-supplied OFDM demodulator.

Each case reads a capture the reference ships with its own tests (tests/golden/iq/, data files copied as they are), demodulates it with
`ofdm_rx(nof_prb, cp_norm, iq, region)` - the oracle's orc_ofdm_rx_sf on the CPU, the HIP kernel on the GPU - and hands the grid to the
reference's own compiled estimator and channel decoders (oracle/_ref via oracle/refdrv.c). The assertion of each case is the one the
reference's test program makes. None of this involves an OFDM modulator: a wrong CP offset, bin order, DC skip or MBSFN slot layout in the
demodulator under test makes these fail (tests/test_recorded_iq.py also shows that by breaking the grid on purpose)."""
import numpy as np

from refdrv import RefDl, read_iq

BCH_PAYLOAD_FILE = [0, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]  # pbch_file_test.c:46


def pdsch_pdcch_file(ofdm_rx, mangle=None):
    """pdsch_pdcch_file_test -c 1 -f 3 -n 6 -p 1 -i signal.1.92M.amar.dat (pdsch_pdcch_file_test.c:170-235): subframes 0.. until a
    DCI with the SI-RNTI is found; exit 0 iff one is found and srslte_pdsch_decode raises no error. Returns what was decoded in every
    one of the 10 subframes of the capture (the reference's loop stops at the first hit, subframe 2)."""
    rx = RefDl(6, 1, 1)
    rx.set_rnti(0xFFFF)
    rx.set_chest_cfg()  # ZERO_OBJECT(ue_dl_cfg)
    out = []
    for sf in range(10):
        grid = ofdm_rx(6, True, read_iq("signal.1.92M.amar.dat", 1920, sf * 1920), 0)
        rx.put_grid(mangle(grid) if mangle else grid)
        rc, cfi, corr = rx.estimate(sf)
        assert rc == 0
        found, grant = rx.find_dci(0xFFFF)
        assert found >= 0
        r = {"sf": sf, "cfi": cfi, "cfi_corr": corr, "dci": found == 1, "grant": grant, "crc": None, "tb": None, "ce": rx.ce(), "noise": rx.chest_res().noise_estimate}
        if found == 1:
            crc, _ = rx.decode_pdsch()
            assert crc >= 0
            r["crc"], r["tb"] = bool(crc), rx.payload(grant["tbs"] // 8)
        out.append(r)
    rx.free()
    return out


def pcfich_file(ofdm_rx):
    """pcfich_file_test -c 150 -n 50 -p 2 -i signal.10M.dat (pcfich_file_test.c:205-260): one read of a subframe's worth of samples (the
    capture holds 7681, the rest of the zeroed buffer stays zero), srslte_chest_dl_estimate, srslte_pcfich_decode; exit 0 iff
    cfi == 2 and the correlation exceeds 2.8."""
    rx = RefDl(50, 2, 150)
    rx.put_grid(ofdm_rx(50, True, read_iq("signal.10M.dat", 15 * 768), 0))
    n, cfi, corr = rx.pcfich()
    ce = [rx.ce(0), rx.ce(1)]
    rx.free()
    return n, cfi, corr, ce


def pbch_file(ofdm_rx):
    """pbch_file_test -i signal.1.92M.dat (pbch_file_test.c:182-240): first subframe, srslte_chest_dl_estimate, srslte_pbch_decode; exit 0
    iff 2 ports, SFN offset 0 and the payload equals bch_payload_file."""
    rx = RefDl(6, 2, 150)
    rx.put_grid(ofdm_rx(6, True, read_iq("signal.1.92M.dat", 1920), 0))
    n, ports, off, bch = rx.pbch_decode()
    rx.free()
    return n, ports, off, bch


def pmch_file(ofdm_rx, region=2):
    """pmch_file_test -i pmch_100prbs_MCS2_SR0.bin (pmch_file_test.c:150-228): one MBSFN subframe (tti 1) of a 100-PRB extended-CP cell,
    non-MBSFN region 2, area id 1, triangle filter 0.1 + interpolate_subframe + PSS noise algorithm, forced grant MCS 2; the test prints
    "PMCH Decoded OK!" iff the CRC passes."""
    rx = RefDl(100, 1, 1, cp_ext=True, phich_resources=0)
    assert rx.L.refdrv_dl_set_mbsfn_area_id(rx.h, 1) == 0
    rx.set_chest_cfg(noise_alg=1, filter_type=1, coef=(0.1, 0.0), interpolate_subframe=True, mbsfn_area_id=1)
    rx.put_grid(ofdm_rx(100, False, read_iq("pmch_100prbs_MCS2_SR0.bin", 23040), region))
    rc, cfi, cfi_corr = rx.estimate(1, mbsfn=True, cfi_in=2)  # srslte_ue_dl_decode_fft_estimate also decodes the PCFICH of symbol 0
    assert rc == 0
    crc, tbs = rx.pmch_decode(2, 1, 2)
    tb, ce = rx.payload(tbs // 8), rx.ce()
    rx.free()
    return {"crc": crc, "tbs": tbs, "tb": tb, "ce": ce, "cfi": cfi, "cfi_corr": cfi_corr}
