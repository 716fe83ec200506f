"""srslte_hip_dl_rx_batch_grants: a new grant every subframe (arbitrary PRB allocation per slot, modulation, transport block size,
RNTI, CFI, redundancy version), as srslte_pdsch_decode takes them (pdsch.c:833-997, srslte_pdsch_cp :81-206). Checked against the oracle
chain subframe by subframe - which tests/test_oracle_vs_ref.py::test_pdsch_arbitrary_allocation_vs_reference pins to the reference's
srslte_pdsch_encode / srslte_pdsch_decode on such grants - and, where oracle/_ref travelled, against srslte_pdsch_decode directly."""
import importlib

import numpy as np
import pytest

import refdrv
from lte_sim import DlConfig, OrcHarq, make_subframe, oracle_rx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    return importlib.import_module("srslte-emane_amd")


def ref_grant(rx, P, sf, how, mcs, rnti, cfi, rng, tm=0):
    """Let the reference turn a DCI-like description into a grant (TBS / modulation of the MCS and PRB count, type 0 / type 2 PRB maps)."""
    if how[0] == "type0":
        rx.set_grant(sf, cfi, rnti, mcs, rbg_bitmask=how[1], tm=tm)
    elif how[0] == "type2":
        rx.set_grant_type2(sf, cfi, rnti, mcs, how[1], how[2], distributed=how[3])
    else:
        n = how[2]
        rx.set_grant_type2(sf, cfi, rnti, mcs, n, 0)
        m = np.zeros((2, P), np.uint8)
        if how[1] == "centre":
            m[:, list(range(P // 2 - 3, P // 2 - 3 + n))] = 1
        elif how[1] == "slots":
            m[0, rng.choice(P, n, replace=False)] = 1
            m[1, rng.choice(P, n, replace=False)] = 1
        else:
            m[:, rng.choice(P, n, replace=False)] = 1
        rx.set_prb_masks(m[0], m[1])
    return rx.grant_info()


# (how, mcs, cfi, snr): a mixed TTI stream
MIX = {
    100: [(("type0", 0x1ffffff), 28, 1, 19.0), (("type2", 3, 40, False), 5, 2, 8.0), (("type2", 16, 3, True), 14, 1, 14.0), (("type0", 0x0a5a5a5), 22, 3, 18.0),
          (("mask", "random", 30), 9, 1, 9.0), (("mask", "centre", 6), 12, 1, 12.0), (("type0", 0x1ffffff), 20, 2, 13.0), (("mask", "slots", 25), 17, 1, 16.0),
          (("type2", 100, 0, False), 1, 1, 4.0), (("mask", "random", 7), 27, 2, 24.0)],
    25: [(("type0", 0x1fff), 21, 1, 17.0), (("type2", 3, 10, False), 7, 1, 9.0), (("mask", "centre", 7), 10, 2, 11.0), (("type2", 8, 1, True), 15, 1, 14.0),
         (("mask", "centre", 1), 3, 1, 7.0), (("mask", "slots", 9), 24, 3, 21.0), (("type0", 0x0aaa), 12, 2, 12.0), (("mask", "random", 5), 28, 1, 26.0)],
    15: [(("type0", 0xff), 18, 1, 15.0), (("mask", "centre", 7), 6, 1, 8.0), (("mask", "centre", 2), 9, 3, 10.0), (("type2", 3, 2, False), 25, 1, 22.0),
         (("mask", "slots", 4), 13, 2, 13.0), (("type2", 6, 0, True), 4, 1, 7.0)],
}


need_ref_later = pytest.mark.skipif(refdrv.lib() is None, reason="oracle/_ref did not travel with the repo")


def build_stream(P, cell_id, tti0, rng, csi=False, llr8=False, npt=1, nrx=1, mix=None, cp_ext=False):
    """Grants from the reference where it is there (it is on the GPU box: oracle/_ref travels), subframes from the oracle's transmitter."""
    rx = refdrv.RefDl(P, npt, cell_id, cp_ext=cp_ext)
    out = []
    for b, (how, mcs, cfi, snr) in enumerate(mix or MIX[P]):
        sf, rnti = (tti0 + b) % 10, 0x100 + 7 * b
        info = ref_grant(rx, P, sf, how, mcs, rnti, cfi, rng, tm=1 if npt > 1 else 0)  # srslte_tm_t: SRSLTE_TM2 = 1
        cfg = DlConfig(P, cell_id, info["mod"], info["tbs"], cfi=cfi, rnti=rnti, prb_mask=info["prb_mask"], csi=csi, llr8=llr8, nof_ports=npt, nof_rx=nrx,
                       cp_ext=cp_ext)
        iq, data = make_subframe(cfg, tti0 + b, rng, snr_db=snr)
        out.append({"cfg": cfg, "iq": iq, "data": data, "info": info})
    rx.free()
    return out


@need_ref_later
@pytest.mark.parametrize("P,cell_id,tti0", [(100, 1, 0), (25, 150, 5), (15, 2, 0)])
def test_mixed_grants_on_an_extended_cp_cell(hp, P, cell_id, tti0):
    """The per-subframe-grant entry point on an extended-CP cell (cfg.cp_ext): the RE lists the device makes from the grants' PRB masks
    (pdsch_relist_kernel with six symbols per slot: CRS symbols 0 and 3, PSS / SSS symbols 5 and 4, the stale-offset rule of the odd
    bandwidth's half PRBs) equal srslte_pdsch_cp's order, LLRs within one LSB of the oracle chain, verdicts and transport blocks equal."""
    rng = np.random.default_rng(31 * P + tti0)
    mix = [(how, mcs, cfi, snr + 1.5) for how, mcs, cfi, snr in MIX[P] if mcs <= 24]  # fewer REs per PRB than with the normal CP: the top MCS would not fit
    stream = build_stream(P, cell_id, tti0, rng, mix=mix, cp_ext=True)
    tbs_max = max(s["cfg"].tbs for s in stream)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rxg = hp.DlRx(cell_id, P, 1, 0, 1, tbs_max, 6, len(stream), True, hc, cp_ext=True)
    grants = [hp.DlGrant.make(P, s["cfg"].mod, s["cfg"].tbs, s["cfg"].rnti, cfi=s["cfg"].cfi, prb_mask=s["cfg"].prb_mask) for s in stream]
    rc, tb, ok = rxg.decode_grants(np.stack([s["iq"] for s in stream]), tti0, grants)
    assert rc == 0
    n = len(stream)
    e = rxg.debug(11, np.int16, n * 16 * ((14 * 12 * P * 8 + 15) // 16)).reshape(n, -1)
    relist = rxg.debug(15, np.uint32, n * 14 * 12 * P).reshape(n, -1)
    nok = 0
    for b, s in enumerate(stream):
        cfg = s["cfg"]
        r = oracle_rx(cfg, s["iq"], tti0 + b, keep=True)
        idx = cfg.indices((tti0 + b) % 10)
        assert len(idx) == s["info"]["nof_re"], b                                   # the reference's own count for its grant (srslte_ra_dl_grant_nof_re)
        assert np.array_equal(relist[b, :len(idx)], idx), b
        diff = np.abs(e[b, :len(r["e_raw"])].astype(int) - r["e_raw"].astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (b, diff.max(), (diff > 0).mean())
        assert bool(ok[b]) == bool(r["ok"]), b
        if ok[b]:
            assert np.array_equal(tb[b, :cfg.tbs // 8 + 3], r["tb"]) and np.array_equal(tb[b, :cfg.tbs // 8], s["data"]), b
            nok += 1
    assert nok >= n - 2
    rxg.free()


need_ref = pytest.mark.skipif(refdrv.lib() is None, reason="oracle/_ref did not travel with the repo")


@need_ref
@pytest.mark.parametrize("P,cell_id,tti0,csi,llr8", [(100, 1, 0, False, False), (25, 150, 5, False, False), (15, 2, 0, False, False), (25, 3, 8, False, False),
                                                    (100, 1, 0, True, False), (15, 2, 5, True, False), (25, 150, 0, True, False), (100, 1, 0, True, True),
                                                    (25, 150, 5, False, True), (15, 2, 0, True, True)])
def test_mixed_grants_vs_oracle_and_reference(hp, P, cell_id, tti0, csi, llr8):
    """csi: cfg.csi_enable (the srsUE default) - the per-RE gains of every subframe's own allocation weigh its LLRs (pdsch.c:574-690).
    llr8: cfg.llr_8bit (the srsUE default): int8 demapper, de-matcher and the 8-bit decoder back-end of every block length in the stream
    (avx8 / sse8 / the widening fall-backs, turbodecoder.c:421-487)."""
    rng = np.random.default_rng(P + tti0)
    stream = build_stream(P, cell_id, tti0, rng, csi, llr8)
    assert len({(s["info"]["mod"], s["info"]["tbs"], s["info"]["nof_re"]) for s in stream}) >= 4  # >= 4 different (allocation, MCS) pairs
    tbs_max = max(s["cfg"].tbs for s in stream)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rxg = hp.DlRx(cell_id, P, 1, 0, 1, tbs_max, 6, len(stream), True, hc, csi=csi, llr_8bit=llr8)
    grants = [hp.DlGrant.make(P, s["cfg"].mod, s["cfg"].tbs, s["cfg"].rnti, cfi=s["cfg"].cfi, prb_mask=s["cfg"].prb_mask) for s in stream]
    rc, tb, ok = rxg.decode_grants(np.stack([s["iq"] for s in stream]), tti0, grants)
    assert rc == 0
    n = len(stream)
    e = rxg.debug(11, np.int8 if llr8 else np.int16, n * 16 * ((14 * 12 * P * 8 + 15) // 16)).reshape(n, -1)
    relist = rxg.debug(15, np.uint32, n * 14 * 12 * P).reshape(n, -1)
    ref = refdrv.RefDl(P, 1, cell_id)
    ref.set_chest_cfg(filter_type=0, coef=(4.0, 1.0))
    ref.set_pdsch_cfg(max_iterations=6, mmse=True, csi=csi, llr8=llr8)
    nok = 0
    for b, s in enumerate(stream):
        cfg = s["cfg"]
        r = oracle_rx(cfg, s["iq"], tti0 + b, keep=True)
        idx = cfg.indices((tti0 + b) % 10)
        assert np.array_equal(relist[b, :len(idx)], idx), b                     # RE list made on the device = srslte_pdsch_cp's order
        diff = np.abs(e[b, :len(r["e_raw"])].astype(int) - r["e_raw"].astype(int))  # LLRs (before the CSI weighting): own float stages upstream -> 1 LSB on a few
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (b, diff.max(), (diff > 0).mean())
        assert bool(ok[b]) == bool(r["ok"]), b
        if ok[b]:
            assert np.array_equal(tb[b, :cfg.tbs // 8 + 3], r["tb"]) and np.array_equal(tb[b, :cfg.tbs // 8], s["data"]), b
            nok += 1
        # and the reference's own srslte_pdsch_decode with the reference's own grant, on the same subframe
        ref.set_rnti(cfg.rnti)
        ref_grant(ref, P, (tti0 + b) % 10, MIX[P][b][0], MIX[P][b][1], cfg.rnti, cfg.cfi, np.random.default_rng(0))
        ref.set_prb_masks(cfg.prb_mask[0], cfg.prb_mask[1])
        ref.put_grid(r["grid"])
        assert ref.chest() == 0
        crc, _ = ref.decode_pdsch()
        if bool(crc) == bool(ok[b]) and crc:  # CRC flags can differ on a marginal block (the reference equaliser's 12-bit reciprocal)
            assert np.array_equal(ref.payload(cfg.tbs // 8), tb[b, :cfg.tbs // 8]), b
    assert nok >= n - (4 if llr8 else 2)
    ref.free()
    rxg.free()


MIX_DIV = {
    25: [(("type0", 0x1fff), 14, 1, 12.0), (("type2", 4, 10, False), 7, 1, 8.0), (("mask", "centre", 7), 10, 2, 10.0), (("type2", 8, 1, True), 15, 1, 13.0),
         (("mask", "slots", 9), 20, 3, 18.0), (("type0", 0x0aaa), 12, 2, 11.0), (("mask", "random", 5), 24, 1, 24.0)],
    50: [(("type0", 0x1ffff), 16, 1, 13.0), (("type2", 6, 20, False), 5, 2, 8.0), (("mask", "random", 12), 22, 1, 20.0), (("mask", "slots", 20), 9, 3, 9.0),
         (("type2", 50, 0, False), 12, 1, 11.0)],
    100: [(("type0", 0x1ffffff), 25, 1, 24.0), (("type2", 16, 3, True), 14, 1, 13.0), (("mask", "random", 30), 9, 2, 9.0), (("type0", 0x0a5a5a5), 18, 3, 15.0)],
}


@need_ref
@pytest.mark.parametrize("P,cell_id,tti0,npt,nrx,csi,llr8", [(25, 150, 0, 2, 1, False, False), (25, 7, 4, 2, 2, True, False), (50, 3, 0, 4, 1, False, False),
                                                            (100, 1, 5, 2, 1, True, True), (50, 11, 5, 4, 2, True, False)])
def test_mixed_grants_transmit_diversity(hp, P, cell_id, tti0, npt, nrx, csi, llr8):
    """The same on 2- and 4-port cells (transmit diversity, TM2): RE lists that leave every port's CRS out - with upstream's stale-offset
    rule on the half PRBs of an odd bandwidth -, SFBC (+ FSTD) pre-decoding per subframe's own allocation, the code-block split in units of
    Qm * 2 bits (sch.c:507-531), against the oracle chain and the reference's own srslte_pdsch_decode."""
    rng = np.random.default_rng(10 * P + tti0 + npt)
    stream = build_stream(P, cell_id, tti0, rng, csi, llr8, npt, nrx, MIX_DIV[P])
    tbs_max = max(s["cfg"].tbs for s in stream)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rxg = hp.DlRx(cell_id, P, 1, 0, 1, tbs_max, 6, len(stream), True, hc, csi=csi, llr_8bit=llr8, nof_ports=npt, nof_rx=nrx)
    grants = [hp.DlGrant.make(P, s["cfg"].mod, s["cfg"].tbs, s["cfg"].rnti, cfi=s["cfg"].cfi, prb_mask=s["cfg"].prb_mask) for s in stream]
    rc, tb, ok = rxg.decode_grants(np.stack([s["iq"] for s in stream]), tti0, grants)
    assert rc == 0
    n = len(stream)
    e = rxg.debug(11, np.int8 if llr8 else np.int16, n * 16 * ((14 * 12 * P * 8 + 15) // 16)).reshape(n, -1)
    relist = rxg.debug(15, np.uint32, n * 14 * 12 * P).reshape(n, -1)
    ref = refdrv.RefDl(P, npt, cell_id, nof_rx=nrx)
    ref.set_chest_cfg(filter_type=0, coef=(4.0, 1.0))
    ref.set_pdsch_cfg(max_iterations=6, mmse=True, csi=csi, llr8=llr8)
    nok = 0
    for b, s in enumerate(stream):
        cfg = s["cfg"]
        r = oracle_rx(cfg, s["iq"], tti0 + b, keep=True)
        idx = cfg.indices((tti0 + b) % 10)
        assert np.array_equal(relist[b, :len(idx)], idx), b
        diff = np.abs(e[b, :len(r["e_raw"])].astype(int) - r["e_raw"].astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (b, diff.max(), (diff > 0).mean())
        assert bool(ok[b]) == bool(r["ok"]), b
        if ok[b]:
            assert np.array_equal(tb[b, :cfg.tbs // 8 + 3], r["tb"]) and np.array_equal(tb[b, :cfg.tbs // 8], s["data"]), b
            nok += 1
        ref.set_rnti(cfg.rnti)
        ref_grant(ref, P, (tti0 + b) % 10, MIX_DIV[P][b][0], MIX_DIV[P][b][1], cfg.rnti, cfg.cfi, np.random.default_rng(0), tm=1)
        ref.set_prb_masks(cfg.prb_mask[0], cfg.prb_mask[1])
        for a in range(nrx):
            ref.put_grid(np.asarray(r["grid"]).reshape(nrx, -1)[a], a)
        assert ref.chest() == 0
        crc, _ = ref.decode_pdsch()
        if bool(crc) == bool(ok[b]) and crc:
            assert np.array_equal(ref.payload(cfg.tbs // 8), tb[b, :cfg.tbs // 8]), b
    assert nok >= n - 2
    ref.free()
    rxg.free()


def test_grants_full_band_equals_fixed_pipeline(hp):
    """cfg2's grant through the grants entry point = srslte_hip_dl_rx_batch, byte for byte; subframes without a transport block are skipped."""
    rng = np.random.default_rng(5)
    cfg = DlConfig(100, 1, 3, 75376, cfi=1, rnti=0x1234)
    iq = np.stack([make_subframe(cfg, t, rng, snr_db=18.5, amp=0.1)[0] for t in range(6)])
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(1, 100, 1, 0x1234, 3, 75376, 6, 6, True, hc)
    tb0, ok0 = rx.decode(iq, 0)
    tb0, ok0 = tb0.copy(), ok0.copy()
    grants = [hp.DlGrant.make(100, 3, 75376 if b != 2 else 0, 0x1234) for b in range(6)]
    rc, tb1, ok1 = rx.decode_grants(iq, 0, grants)
    assert rc == 0
    for b in range(6):
        if b == 2:
            assert ok1[b] == 0
        else:
            assert ok1[b] == ok0[b] and (not ok0[b] or np.array_equal(tb1[b], tb0[b])), b
    assert ok0.sum() >= 3
    # argument checks: a transport block larger than the object was made for, an empty allocation, an invalid modulation
    for bad in (hp.DlGrant.make(100, 3, 75376 + 8, 1), hp.DlGrant.make(100, 3, 1000, 1, prb_mask=np.zeros((2, 100))), hp.DlGrant.make(100, 7, 1000, 1)):
        rc, _, _ = rx.decode_grants(iq[:1], 0, [bad])
        assert rc == hp.SRSLTE_ERROR_INVALID_INPUTS
    rx.free()


def test_grants_harq(hp):
    """Per-subframe rv / new_data: slot b keeps its soft buffers between calls (decode_tb_cb, sch.c:299-414). The oracle's HARQ chain is
    fed the device's own LLRs of every transmission (debug buffer 11), so that combining, skipping of decoded blocks and decoding are
    compared exactly (at these SNRs blocks are marginal: an LSB of difference between two float front ends would decide them)."""
    import ctypes as C
    from _libs import OrcSchCfg, oracle, p
    rng = np.random.default_rng(11)
    P, cell_id = 50, 4
    m = np.zeros((2, P), np.uint8)
    m[:, 5:35] = 1
    cfgs = [DlConfig(P, cell_id, 2, 11448, cfi=2, rnti=0x77, prb_mask=m), DlConfig(P, cell_id, 3, 36696, cfi=1, rnti=0x78)]
    snr = [6.0, 12.5]  # too low for one transmission
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    rx = hp.DlRx(cell_id, P, 1, 0, 1, 36696, 6, 2, True, hc)
    harq = [OrcHarq(c) for c in cfgs]
    datas = [rng.integers(0, 256, c.tbs // 8, dtype=np.uint8) for c in cfgs]
    e_stride = 16 * ((14 * 12 * P * 8 + 15) // 16)
    outcomes, done = [], [False, False]
    for t, rv in enumerate((0, 2, 3, 1)):
        iq = [make_subframe(c, 0 + b, rng, snr_db=snr[b], rv=rv, data=datas[b])[0] for b, c in enumerate(cfgs)]
        # an acknowledged transport block is not scheduled again (the MAC's job): its subframe carries no grant for this UE any more
        grants = [hp.DlGrant.make(P, c.mod, 0 if done[b] else c.tbs, c.rnti, cfi=c.cfi, rv=rv, new_data=(t == 0), prb_mask=c.prb_mask) for b, c in enumerate(cfgs)]
        rc, tb, ok = rx.decode_grants(np.stack(iq), 0, grants)
        assert rc == 0
        e = rx.debug(11, np.int16, 2 * e_stride).reshape(2, e_stride)
        iters = rx.debug(13, np.uint32, 2 * cfgs[1].seg.C).reshape(2, -1)
        for b, c in enumerate(cfgs):
            if done[b]:
                assert ok[b] == 0
                continue
            nbits = len(c.indices(b)) * c.Qm
            sch = OrcSchCfg(c.tbs, nbits, c.Qm_sch, rv, c.max_iter)
            otb, oit, ocb = np.zeros(c.tbs // 8 + 16, np.uint8), np.zeros(c.seg.C, np.uint32), np.zeros(c.seg.C, np.uint8)
            orc = oracle().orc_dlsch_decode_harq(C.byref(sch), p(np.ascontiguousarray(e[b, :nbits])), 0, 1 if t == 0 else 0, p(harq[b].w), p(harq[b].crc),
                                                 p(harq[b].data), p(otb), p(oit), p(ocb))
            assert bool(ok[b]) == (orc == 0), (t, b)
            assert np.array_equal(iters[b, :c.seg.C], oit), (t, b, iters[b], oit)  # passes per block; 0 = skipped, decoded in an earlier transmission
            if ok[b]:
                assert np.array_equal(tb[b, :c.tbs // 8 + 3], otb[:c.tbs // 8 + 3]) and np.array_equal(tb[b, :c.tbs // 8], datas[b])
            outcomes.append(bool(ok[b]))
            done[b] = bool(ok[b])
    assert not all(outcomes) and any(outcomes)  # the retransmissions were needed, and helped
    rx.free()


@pytest.mark.parametrize("seed", range(8))
def test_grants_harq_drawn_sequences(hp, seed):
    """test_grants_harq with everything drawn: per call and HARQ slot either a retransmission of the slot's transport block - with a drawn
    redundancy version, a NEW allocation (other PRBs in each slot of the subframe, so another number of LLRs), possibly another modulation,
    another CFI - or a new transport block of a drawn size; now and then a retransmission into a slot that has already passed (answered with
    CRC 0, as decode_tb_cb does, sch.c:399-410). The oracle's soft-combining back end on the DEVICE's LLRs of every transmission gives the
    device's CRC flags, pass counts per block (0 = carried over) and bytes exactly."""
    import ctypes as C
    from _libs import OrcCbsegm, OrcSchCfg, oracle, p
    rng = np.random.default_rng(7700 + seed)
    P, cell_id, nsf = int(rng.choice([15, 25, 50])), int(rng.integers(0, 504)), 3
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0

    def draw_mask():
        m = np.zeros((2, P), np.uint8)
        n = int(rng.integers(max(2, P // 5), P + 1))
        if rng.integers(0, 2):
            m[:, rng.choice(P, n, replace=False)] = 1
        else:  # another set of PRBs in the second slot (distributed allocations)
            m[0, rng.choice(P, n, replace=False)] = 1
            m[1, rng.choice(P, n, replace=False)] = 1
        return m

    def draw_tb(slot):
        mod = int(rng.choice([1, 2, 3]))
        m = draw_mask()
        nre = min(len(DlConfig(P, cell_id, mod, 16, cfi=3, rnti=0x50 + slot, prb_mask=m).indices(sf)) for sf in (0, 1, 5))
        tbs = max(40, int(float(rng.uniform(0.55, 0.95)) * nre * 2 * mod) // 8 * 8)
        while True:
            seg = OrcCbsegm()
            if oracle().orc_cbsegm(C.byref(seg), tbs) == 0 and seg.F == 0 and seg.C2 == 0:
                return {"tbs": tbs, "mod": mod, "mask": m, "data": rng.integers(0, 256, tbs // 8, dtype=np.uint8), "done": False, "new": True}
            tbs -= 8

    tbs_max = 12 * P * 11 * 6  # more than any drawn block
    while True:
        seg = OrcCbsegm()
        if oracle().orc_cbsegm(C.byref(seg), tbs_max) == 0 and seg.F == 0 and seg.C2 == 0:
            break
        tbs_max -= 8
    Cmax = seg.C
    rx = hp.DlRx(cell_id, P, 1, 0, 3, tbs_max, 6, nsf, True, hc)
    e_stride = 16 * ((14 * 12 * P * 8 + 15) // 16)
    slots = [draw_tb(b) for b in range(nsf)]
    harq = [None] * nsf
    n_ok = n_retx_ok = n_after = 0
    for call in range(6):
        tti0 = int(rng.integers(0, 10240))
        cfgs, grants, iq = [], [], []
        for b in range(nsf):
            sl = slots[b]
            if not sl["new"]:
                if sl["done"] and rng.integers(0, 3):  # acknowledged: usually a new block next, sometimes a duplicate retransmission
                    slots[b] = sl = draw_tb(b)
                else:  # retransmission: other allocation, maybe another modulation (the soft buffer holds coded bits)
                    sl["mask"] = draw_mask()
                    if rng.integers(0, 3) == 0:
                        sl["mod"] = int(rng.choice([2, 3]))
            rv, cfi = (0 if sl["new"] and rng.integers(0, 2) else int(rng.integers(0, 4))), int(rng.integers(1, 4))
            c = DlConfig(P, cell_id, sl["mod"], sl["tbs"], cfi=cfi, rnti=0x50 + b, prb_mask=sl["mask"])
            if len(c.indices((tti0 + b) % 10)) * c.Qm < 64:  # too few LLRs to mean anything: take the whole band
                sl["mask"] = np.ones((2, P), np.uint8)
                c = DlConfig(P, cell_id, sl["mod"], sl["tbs"], cfi=cfi, rnti=0x50 + b, prb_mask=sl["mask"])
            snr = {1: 1.0, 2: 7.0, 3: 12.0}[sl["mod"]] + 10.0 * (sl["tbs"] / (len(c.indices((tti0 + b) % 10)) * c.Qm) - 0.4) - float(rng.uniform(0.0, 3.0))
            iq.append(make_subframe(c, tti0 + b, rng, snr_db=snr, rv=rv, data=sl["data"])[0])
            grants.append(hp.DlGrant.make(P, sl["mod"], sl["tbs"], 0x50 + b, cfi=cfi, rv=rv, new_data=sl["new"], prb_mask=sl["mask"]))
            cfgs.append((c, rv))
            if sl["new"]:
                harq[b] = OrcHarq(c)
        rc, tb, ok = rx.decode_grants(np.stack(iq), tti0, grants)
        assert rc == 0
        e = rx.debug(11, np.int16, nsf * e_stride).reshape(nsf, e_stride)
        iters = rx.debug(13, np.uint32, nsf * Cmax).reshape(nsf, -1)
        for b in range(nsf):
            (c, rv), sl = cfgs[b], slots[b]
            what = (seed, call, b, P, sl["mod"], sl["tbs"], rv, sl["new"], sl["done"])
            nbits = len(c.indices((tti0 + b) % 10)) * c.Qm
            sch = OrcSchCfg(c.tbs, nbits, c.Qm_sch, rv, c.max_iter)
            otb, oit, ocb = np.zeros(c.tbs // 8 + 16, np.uint8), np.zeros(c.seg.C, np.uint32), np.zeros(c.seg.C, np.uint8)
            orc = oracle().orc_dlsch_decode_harq(C.byref(sch), p(np.ascontiguousarray(e[b, :nbits])), 0, 1 if sl["new"] else 0, p(harq[b].w), p(harq[b].crc),
                                                 p(harq[b].data), p(otb), p(oit), p(ocb))
            assert bool(ok[b]) == (orc == 0) and np.array_equal(iters[b, :c.seg.C], oit), what + (bool(ok[b]), orc, iters[b, :c.seg.C], oit)
            if sl["done"] and not sl["new"]:
                assert not ok[b] and not oit.any()  # the duplicate retransmission
                n_after += 1
            if ok[b]:
                assert np.array_equal(tb[b, :c.tbs // 8 + 3], otb[:c.tbs // 8 + 3]) and np.array_equal(tb[b, :c.tbs // 8], sl["data"]), what
                n_ok += 1
                n_retx_ok += not sl["new"]
                sl["done"] = True
            sl["new"] = False
    print("drawn grants HARQ: %d blocks delivered, %d of them by a retransmission, %d duplicate retransmissions refused" % (n_ok, n_retx_ok, n_after))
    assert n_ok > 0
    rx.free()



@pytest.mark.parametrize("llr8", [False, True])
def test_every_decoder_kind_in_one_call(hp, llr8):
    """One grants call whose transport blocks need every decoder kind at once - unwindowed (K = 320), 8 windows (K = 640), two blocks per wavefront
    (K = 1568, 4032, and 6144 x 2 as a pair of one transport block), each kind with an odd number of blocks - so that tdec_run_groups takes the
    mixed launch (tdec_mix_kernel; with 8-bit LLRs: avx8 / sse8 launches beside the widened mix of the two short kinds). SNRs around each
    block's threshold: early stops at different passes and failures in one wavefront. Verdicts, transport blocks and the pass count of every
    code block against the oracle chain."""
    P, cell_id, tti0 = 50, 21, 3
    rng = np.random.default_rng(77 + llr8)
    # (first PRB, PRBs, mod, tbs, snr): K = 320 x3, 640 x3, 1568 x3, 4032 x1, 6144 x2 (one TB)
    plan = [(0, 3, 1, 296, 1.0), (5, 3, 1, 296, -1.0), (9, 4, 1, 296, 6.0), (0, 5, 1, 616, 2.0), (10, 5, 1, 616, 0.2), (20, 6, 1, 616, 5.0),
            (0, 12, 1, 1544, 1.5), (14, 12, 1, 1544, 0.3), (30, 14, 1, 1544, 5.0), (0, 25, 2, 4008, 6.5), (0, 50, 2, 12216, 8.5)]
    if llr8:  # the 8-bit waterfalls sit a little higher
        plan = [(a, n, m, t, s + 1.0) for a, n, m, t, s in plan]
    stream = []
    for b, (first, n, mod, tbs, snr) in enumerate(plan):
        mask = np.zeros((2, P), np.uint8)
        mask[:, first:first + n] = 1
        cfg = DlConfig(P, cell_id, mod, tbs, cfi=1 + b % 3, rnti=0x200 + b, prb_mask=mask, llr8=llr8)
        iq, data = make_subframe(cfg, tti0 + b, rng, snr_db=snr)
        stream.append((cfg, iq, data))
    Ks = sorted({int(c.seg.K1) for c, _, _ in stream})
    assert Ks == [320, 640, 1568, 4032, 6144]
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    tbs_max = max(c.tbs for c, _, _ in stream)
    rxg = hp.DlRx(cell_id, P, 1, 0, 1, tbs_max, 6, len(stream), True, hc, llr_8bit=llr8)
    grants = [hp.DlGrant.make(P, c.mod, c.tbs, c.rnti, cfi=c.cfi, prb_mask=c.prb_mask) for c, _, _ in stream]
    rc, tb, ok = rxg.decode_grants(np.stack([iq for _, iq, _ in stream]), tti0, grants)
    assert rc == 0
    Cmax = -(-(tbs_max + 24) // 6120) if tbs_max + 24 > 6144 else 1
    iters = rxg.debug(13, np.uint32, len(stream) * Cmax).reshape(len(stream), Cmax)
    spread, n_ok = set(), 0
    for b, (cfg, iq, data) in enumerate(stream):
        r = oracle_rx(cfg, iq, tti0 + b, keep=True)
        assert bool(ok[b]) == bool(r["ok"]), (b, cfg.seg.K1)
        assert np.array_equal(iters[b, :cfg.seg.C], r["iters"]), (b, cfg.seg.K1, iters[b, :cfg.seg.C], r["iters"])
        spread.update(int(x) for x in r["iters"])
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[b, :cfg.tbs // 8 + 3], r["tb"]) and np.array_equal(tb[b, :cfg.tbs // 8], data), b
    assert n_ok >= 5 and len(spread) >= 3, (n_ok, spread)  # decoded and failed blocks, early and late stops
    rxg.free()


def test_transport_blocks_assembled_by_the_decoders_equal_the_assembly_kernel(hp):
    """A 16-bit grants call has its transport blocks assembled and judged by the decoders themselves (tdec_set_tb_ragged: per-slot block counts, CRC24A
    shares with the factor of the block's position, the last block to arrive gives the verdict; a block kept from an earlier transmission contributes
    its stored bytes); with SRSLTE_HIP_GRANTS_TB_DIRECT=0 in the environment an object uses the assembly kernel (tb_crc_bytes_kernel) instead.
    Same subframes both ways - blocks of every decoder kind, transport blocks of one, two and three blocks, SNRs around the thresholds so that
    transport blocks fail as well, a subframe without a transport block, a dirty result buffer - then a second call that retransmits every slot
    (failed blocks combine and pass, delivered transport blocks are refused as duplicates, some slots start a new block): rows and verdicts must be
    identical call by call, failed blocks' bytes included, and the first call's equal the oracle's."""
    import os
    P, cell_id, tti0 = 50, 33, 6
    rng = np.random.default_rng(2024)
    plan = [(0, 3, 1, 296, 1.0), (5, 3, 1, 296, -1.5), (0, 5, 1, 616, 2.0), (10, 5, 1, 616, -0.5), (0, 12, 1, 1544, 1.5), (14, 12, 1, 1544, -0.5),
            (0, 25, 2, 4008, 6.5), (0, 50, 2, 12216, 8.5), (0, 50, 2, 15264, 10.5), (0, 50, 2, 15264, 8.3), (0, 50, 2, 12216, 7.0), None, (3, 30, 2, 6200, 9.0)]
    stream, grants, grants2, iq2 = [], [], [], []
    for b, pl in enumerate(plan):
        if pl is None:  # no transport block in this subframe
            cfg = DlConfig(P, cell_id, 1, 296, cfi=1, rnti=0x300)
            stream.append((None, make_subframe(cfg, tti0 + b, rng, snr_db=3.0)[0], None))
            grants.append(hp.DlGrant.make(P, 1, 0, 0x300))
            grants2.append(hp.DlGrant.make(P, 1, 0, 0x300))
            iq2.append(stream[-1][1])
            continue
        first, n, mod, tbs, snr = pl
        mask = np.zeros((2, P), np.uint8)
        mask[:, first:first + n] = 1
        cfg = DlConfig(P, cell_id, mod, tbs, cfi=1 + b % 3, rnti=0x300 + b, prb_mask=mask)
        iq, data = make_subframe(cfg, tti0 + b, rng, snr_db=snr)
        stream.append((cfg, iq, data))
        grants.append(hp.DlGrant.make(P, mod, tbs, cfg.rnti, cfi=cfg.cfi, prb_mask=mask))
        # the second call, ten subframes later (same subframe index): rv 2 of the same data in most slots, a new block in slots 2 and 7
        fresh = b in (2, 7)
        d2 = rng.integers(0, 256, tbs // 8, dtype=np.uint8) if fresh else data
        iq2.append(make_subframe(cfg, tti0 + 10 + b, rng, snr_db=snr + 1.0, rv=0 if fresh else 2, data=d2)[0])
        grants2.append(hp.DlGrant.make(P, mod, tbs, cfg.rnti, cfi=cfg.cfi, rv=0 if fresh else 2, new_data=fresh, prb_mask=mask))
    assert [int(c.seg.C) for c, _, _ in stream if c] == [1, 1, 1, 1, 1, 1, 1, 2, 3, 3, 2, 2]
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    tbs_max, nsf = 15264, len(stream)
    iqs, iqs2 = np.stack([iq for _, iq, _ in stream]), np.stack(iq2)
    results = []
    for assembly_kernel in (False, True):
        if assembly_kernel:
            os.environ["SRSLTE_HIP_GRANTS_TB_DIRECT"] = "0"  # read when the object's grants state is made (first grants call)
        try:
            rx = hp.DlRx(cell_id, P, 1, 0, 1, tbs_max, 6, nsf, True, hc)
            calls = []
            for x, gr, t0 in ((iqs, grants, tti0), (iqs2, grants2, tti0 + 10)):
                hp.lib().srslte_hip_memset(rx.d_tb.ptr, 0xA5, rx.d_tb.nbytes)
                hp.lib().srslte_hip_memset(rx.d_ok.ptr, 0xA5, rx.d_ok.nbytes)
                rc, tb, ok = rx.decode_grants(x, t0, gr)
                assert rc == 0
                calls.append((tb.copy(), ok.copy()))
            results.append(calls)
            rx.free()
        finally:
            os.environ.pop("SRSLTE_HIP_GRANTS_TB_DIRECT", None)
    for call in range(2):
        (tb_d, ok_d), (tb_k, ok_k) = results[0][call], results[1][call]
        assert np.array_equal(ok_d, ok_k), (call, ok_d, ok_k)
        for b, (cfg, iq, data) in enumerate(stream):
            if cfg is None:
                assert ok_d[b] == 0
                continue
            nb = cfg.tbs // 8 + 3
            assert np.array_equal(tb_d[b, :nb], tb_k[b, :nb]), (call, b, cfg.tbs, np.flatnonzero(tb_d[b, :nb] != tb_k[b, :nb])[:8])
    (tb_d, ok_d), (tb_2, ok_2) = results[0]
    n_ok = n_fail = n_retx = n_dup = 0
    for b, (cfg, iq, data) in enumerate(stream):
        if cfg is None:
            continue
        nb = cfg.tbs // 8 + 3
        r = oracle_rx(cfg, iq, tti0 + b, keep=True)
        assert bool(ok_d[b]) == bool(r["ok"]), (b, cfg.tbs)
        assert np.array_equal(tb_d[b, :nb], r["tb"][:nb]), (b, cfg.tbs, bool(r["ok"]))
        n_ok += bool(r["ok"])
        n_fail += not r["ok"]
        if b in (2, 7):
            continue
        if ok_d[b]:  # delivered by the first call: the retransmission is a duplicate, nothing is decoded, no second delivery; the row holds the block
            assert not ok_2[b] and np.array_equal(tb_2[b, :cfg.tbs // 8], data), b
            n_dup += 1
        elif ok_2[b]:
            assert np.array_equal(tb_2[b, :cfg.tbs // 8], data), b
            n_retx += 1
    assert n_ok >= 5 and n_fail >= 3 and n_retx >= 2 and n_dup >= 4, (n_ok, n_fail, n_retx, n_dup)
