"""Per-PDSCH grants on the transmit side (srslte_hip_dl_tx_batch_grants): a run of TTIs in which every subframe carries the PDSCHs of several
UEs - own PRB masks (contiguous, distributed, across the PSS / SSS / PBCH region), RNTI, modulation, transport block, redundancy version -
against the oracle's stimulus generator run once per PDSCH (pinned to the reference's srslte_pdsch_encode, including masks:
tests/test_oracle_vs_ref.py), and a round trip through the receive side's grants mode."""
import ctypes as C
import importlib

import numpy as np
import pytest

from _libs import OrcOfdm, oracle, p

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    return importlib.import_module("srslte-emane_amd")


def _mask(prb, spans0, spans1=None):
    m = np.zeros((2, prb), np.uint8)
    for s, spans in enumerate((spans0, spans0 if spans1 is None else spans1)):
        for a, b in spans:
            m[s, a:b] = 1
    return m


def _ue_sets(prb):
    if prb == 25:  # (mask, mod, tbs, rv)
        return [
            [(_mask(25, [(0, 8)]), 2, 2216, 0), (_mask(25, [(8, 17)]), 1, 1000, 0), (_mask(25, [(17, 25)]), 3, 4008, 2)],        # sf 0: centre PRBs lose REs
            [(_mask(25, [(0, 25)]), 2, 4008, 0)],
            [(_mask(25, [(0, 2), (10, 14), (20, 23)]), 1, 776, 1), (_mask(25, [(2, 10)], [(14, 20)]), 2, 2216, 0), (_mask(25, [(14, 20)], [(2, 10)]), 2, 1544, 3)],
            [],                                                                                                             # an empty subframe: CRS only
            [(_mask(25, [(3, 4)]), 1, 104, 0), (_mask(25, [(4, 25)]), 4, 7992, 0)],
        ]
    return [
        [(_mask(100, [(0, 50)]), 3, 30576, 0), (_mask(100, [(50, 100)]), 2, 15264, 0)],
        [(_mask(100, [(0, 100)]), 3, 75376, 0)],
        [(_mask(100, [(0, 4), (40, 60), (90, 100)]), 2, 9144, 2), (_mask(100, [(4, 40)]), 4, 30576, 0), (_mask(100, [(60, 90)]), 1, 4584, 1)],
    ]


@pytest.mark.parametrize("prb,npt,tti0,p_a", [(25, 1, 8, 0.0), (25, 2, 3, -3.0), (100, 1, 9, 0.0), (100, 2, 4, 0.0), (25, 4, 8, 0.0)])
def test_dl_tx_grants_vs_oracle(hp, prb, npt, tti0, p_a):
    """Every port's resource grid = the sum of the oracle's per-PDSCH grids with the CRS counted once (levels are table values: 3e-7), and the
    time samples of every port."""
    from lte_sim import DlConfig, make_subframe
    rng = np.random.default_rng(5100 + prb + npt)
    sets = _ue_sets(prb)
    nsf = len(sets)
    grants, datas, exp = [], [], np.zeros((nsf, npt, 14 * 12 * prb), np.complex64)
    for b, lst in enumerate(sets):
        for port in range(npt):  # the CRS of an empty grid
            oracle().orc_crs_put_sf(C.byref(DlConfig(prb, 7, 1, 1000, nof_ports=npt).cell), (tti0 + b) % 10, port, p(exp[b, port]))
        for u, (mask, mod, tbs, rv) in enumerate(lst):
            rnti = 0x200 + 8 * b + u
            cfg = DlConfig(prb, 7, mod, tbs, nof_ports=npt, p_a=p_a, rnti=rnti, prb_mask=mask)
            if len(cfg.indices((tti0 + b) % 10)) % npt:
                continue
            k = {}
            _, data = make_subframe(cfg, tti0 + b, rng, rv=rv, keep=k)
            for port in range(npt):
                exp[b, port][k["idx"]] = k["y"][port]
            grants.append((b, hp.DlGrant.make(prb, mod, tbs, rnti, cfi=1, rv=rv, prb_mask=mask)))
            datas.append(data)
    tx = hp.DlTx(7, prb, 1, 0x1234, 1, max(g.tbs for _, g in grants), nsf, npt, p_a, max_grants=len(grants))
    iq = tx.encode_grants(datas, tti0, nsf, grants)
    grid = tx.debug(3, np.complex64, nsf * npt * 14 * 12 * prb).reshape(nsf, npt, -1)
    q = OrcOfdm()
    oracle().orc_ofdm_init(C.byref(q), prb, True)
    q.normalize = True
    scale = max(1.0, 10 ** (p_a / 20) * (2 ** 0.5 if npt > 1 else 1.0))
    for b in range(nsf):
        for port in range(npt):
            assert np.abs(grid[b, port] - exp[b, port]).max() <= 3e-7 * scale, (b, port)
            iq_o = np.zeros(15 * q.symbol_sz, np.complex64)
            oracle().orc_ofdm_tx_sf(C.byref(q), p(np.ascontiguousarray(exp[b, port])), p(iq_o))
            ref = max(np.abs(iq_o).max(), 1e-9)
            assert np.abs(iq[b, port] - iq_o).max() <= 1e-4 * ref, (b, port)
    tx.free()


@pytest.mark.parametrize("npt", [1, 2])
def test_dl_tx_grants_round_trip_through_rx_grants(hp, npt):
    """A 100-PRB cell, 32 subframes, the band split between three UEs per subframe with changing sizes and modulations: one transmit call for all
    96 PDSCHs; each UE's receiver (its own RNTI) then decodes its PDSCH of every subframe with srslte_hip_dl_rx_batch_grants, noise free: every
    transport block comes back."""
    prb, nsf = 100, 32
    rng = np.random.default_rng(5300 + npt)
    shapes = [((0, 30), 2, 9144), ((30, 70), 3, 22152), ((70, 100), 1, 4584)]
    grants, datas, per_ue = [], [], [[], [], []]
    for b in range(nsf):
        for u in range(3):
            (a, z), mod, tbs = shapes[(u + b) % 3]
            mask = _mask(prb, [(a, z)])
            g = hp.DlGrant.make(prb, mod, tbs, 0x300 + u, cfi=2, prb_mask=mask)
            d = rng.integers(0, 256, tbs // 8, dtype=np.uint8)
            grants.append((b, g))
            datas.append(d)
            per_ue[u].append((g, d))
    tx = hp.DlTx(9, prb, 2, 0x1234, 1, 22152, nsf, npt, max_grants=len(grants))
    iq = tx.encode_grants(datas, 6, nsf, grants)
    rx_iq = iq[:, 0, :] if npt == 1 else (iq[:, 0, :] + 0.7j * iq[:, 1, :]).astype(np.complex64)  # two flat paths into one antenna
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    for u in range(3):
        # a 2-port cell transmits at rho_a = sqrt(2) 10^(p_a/20) (pdsch.c:525): the receiver is told (srslte_pdsch_cfg_t.power_scale / p_a)
        rx = hp.DlRx(9, prb, 2, 0x300 + u, 1, 22152, 6, nsf, True, hc, nof_ports=npt, power_scale=npt > 1, p_a=0.0)
        rc, tb, ok = rx.decode_grants(rx_iq, 6, [g for g, _ in per_ue[u]])
        assert rc == 0 and ok.all(), (u, rc, None if ok is None else int(ok.sum()))
        for b, (g, d) in enumerate(per_ue[u]):
            assert np.array_equal(tb[b][:g.tbs // 8], d), (u, b)
        rx.free()
    tx.free()


def test_dl_tx_grants_argument_errors(hp):
    prb = 25
    tx = hp.DlTx(1, prb, 1, 0x1234, 2, 4008, 2, 1, max_grants=2)
    G = lambda **kw: hp.DlGrant.make(prb, kw.pop("mod", 2), kw.pop("tbs", 4008), 1, **kw)
    d = [np.zeros(501, np.uint8)]
    for bad in ([(2, G())], [(0, G(mod=5))], [(0, G(tbs=4016))], [(0, G(tbs=6200))], [(0, G(rv=4))], [(0, G(cfi=0))], [(0, G(tbs=0))],
                [(0, G(prb_mask=_mask(prb, [])))], [(0, G())] * 3):
        with pytest.raises(RuntimeError):
            tx.encode_grants(d * len(bad), 0, 2, bad)
    iq = tx.encode_grants([], 0, 2, [])  # no PDSCH at all: CRS-only subframes
    assert np.abs(iq).max() > 0
    tx.free()


def _valid_tbs(hp, limit):
    """Transport block sizes the device pipelines take (multiples of 8, no filler bits, one code-block size), largest first, up to `limit` bits."""
    out = []
    for tbs in range(limit - limit % 8, 39, -8):
        rc, s = hp.cbsegm(tbs)
        if rc == 0 and s.F == 0 and s.C2 == 0:
            out.append(tbs)
            if len(out) >= 8:
                break
    return out


@pytest.mark.parametrize("prb,npt,cell_id", [(15, 1, 3), (25, 2, 10), (50, 1, 77), (75, 2, 150), (100, 1, 501), (6, 1, 1), (33, 1, 44), (110, 2, 301)])
def test_dl_grants_fuzz_round_trip(hp, prb, npt, cell_id):
    """Random schedules on even and odd bandwidths: every subframe of a 20-TTI run (both sync subframes included, CFI 1..3) is split between 1..4 UEs
    at random PRB boundaries, some with the two slots' PRBs swapped between UEs, random modulations, the largest transport block that keeps the
    code rate under ~0.6 and the pipelines accept. One transmit call for the whole run; per UE slot one receive call: every block comes back.
    (Not a 7-PRB cell: there EVERY PRB lies in the PSS / SSS / PBCH region, so in subframe 0 an allocation always "starts inside the region" and
    srslte_pdsch_cp's half-PRB branch reads the CRS offset of the previous reference symbol (pdsch.c:172-190): the eNB - upstream's
    and this one, which reproduces it - then writes one data symbol onto a CRS position after the CRS, the UE's estimate of a cell with that
    few pilots suffers, and a 64QAM block on the one affected PRB does not come back. Device and oracle agree RE for RE on such grants,
    test_dl_tx_grants_vs_oracle; it is the round trip that upstream's rule breaks.)"""
    from lte_sim import DlConfig
    rng = np.random.default_rng(5500 + prb + npt)
    nsf, tti0, max_ue = 20, int(rng.integers(0, 10)), 4
    grants, datas, per_ue = [], [], [[] for _ in range(max_ue)]
    tbs_max = 0
    for b in range(nsf):
        cfi = int(rng.integers(1, 4))
        nue = int(rng.integers(1, min(max_ue, prb // 2) + 1))
        cuts = np.sort(rng.choice(np.arange(1, prb), nue - 1, replace=False)) if nue > 1 else np.array([], int)
        bounds = [0] + [int(c) for c in cuts] + [prb]
        spans = [(bounds[i], bounds[i + 1]) for i in range(nue)]
        swap = nue >= 2 and rng.random() < 0.3  # the first two UEs trade PRBs in slot 1 (distributed allocation)
        for u in range(max_ue):
            if u >= nue:
                per_ue[u].append((hp.DlGrant.make(prb, 1, 0, 0x500 + u, cfi=cfi, prb_mask=_mask(prb, [])), None))  # tbs 0: nothing for this UE
                continue
            s1 = spans[1 - u] if (swap and u < 2) else spans[u]
            mask = _mask(prb, [spans[u]], [s1])
            mod = int(rng.integers(1, 5))
            cfg = DlConfig(prb, cell_id, mod, 1000, nof_ports=npt, prb_mask=mask, cfi=cfi)
            nre = len(cfg.indices((tti0 + b) % 10))
            cand = [t for t in _valid_tbs(hp, min(int(0.6 * nre * 2 * mod), 75376)) if nre >= 2 * npt * hp.cbsegm(t)[1].C] if nre >= 8 * npt and nre % npt == 0 else []
            if not cand:
                per_ue[u].append((hp.DlGrant.make(prb, 1, 0, 0x500 + u, cfi=cfi, prb_mask=_mask(prb, [])), None))
                continue
            tbs = cand[0]
            tbs_max = max(tbs_max, tbs)
            g = hp.DlGrant.make(prb, mod, tbs, 0x500 + u, cfi=cfi, prb_mask=mask)
            d = rng.integers(0, 256, tbs // 8, dtype=np.uint8)
            grants.append((b, g))
            datas.append(d)
            per_ue[u].append((g, d))
    assert len(grants) >= nsf
    tx = hp.DlTx(cell_id, prb, 1, 0x1234, 1, tbs_max, nsf, npt, max_grants=len(grants))
    iq = tx.encode_grants(datas, tti0, nsf, grants)
    tx.free()
    rx_iq = iq[:, 0, :] if npt == 1 else (iq[:, 0, :] + 0.6j * iq[:, 1, :]).astype(np.complex64)
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    n_tb = 0
    for u in range(max_ue):
        if not any(d is not None for _, d in per_ue[u]):
            continue
        rx = hp.DlRx(cell_id, prb, 1, 0x500 + u, 1, tbs_max, 6, nsf, True, hc, nof_ports=npt, power_scale=npt > 1, p_a=0.0)
        rc, tb, ok = rx.decode_grants(rx_iq, tti0, [g for g, _ in per_ue[u]])
        assert rc == 0
        for b, (g, d) in enumerate(per_ue[u]):
            if d is None:
                assert not ok[b]
            else:
                n_tb += 1
                assert ok[b] and np.array_equal(tb[b][:g.tbs // 8], d), (u, b, g.mod, g.tbs)
        rx.free()
    assert n_tb == len(grants)
