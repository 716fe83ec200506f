"""Per-PUSCH grants on the uplink (srslte_hip_ul_rx_batch_grants): a run of TTIs in which every subframe carries several PUSCHs of different
UEs - own allocation, DMRS cyclic shift, RNTI, modulation, transport block, redundancy version - against the oracle's eNB chain run once per
PUSCH on the same time samples (the chain tests/test_oracle_vs_ref.py pins to the reference's srslte_chest_ul_estimate_pusch and
srslte_pusch_decode stages)."""
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


# (L_prb, n_prb, n_prb_slot1, mod, tbs, n_dmrs, snr_db) per PUSCH; lists per subframe. TBS values: no filler bits, one code-block size.
UE_SETS_25 = [
    [(10, 0, 0, 2, 4008, 0, 9.5), (6, 12, 12, 1, 1000, 3, 5.0), (3, 20, 20, 1, 328, 5, 4.0)],
    [(25, 0, 0, 2, 4008, 1, 3.0)],
    [(1, 7, 7, 1, 104, 2, 8.0), (12, 8, 8, 3, 7992, 7, 17.0), (1, 24, 24, 1, 104, 4, 8.0), (4, 20, 20, 2, 1544, 6, 9.5)],
    [(10, 2, 13, 2, 4008, 0, 9.5), (2, 12, 0, 1, 328, 1, 5.0)],  # intra-subframe hopping: the two swap ends between the slots
]
UE_SETS_100 = [
    [(48, 0, 0, 3, 30576, 0, 16.5), (48, 50, 50, 2, 22152, 3, 11.0)],
    [(96, 2, 2, 3, 61664, 5, 18.5)],
    [(25, 0, 0, 2, 4008, 1, 3.0), (50, 25, 25, 3, 30576, 2, 16.0), (20, 80, 80, 1, 2216, 4, 2.5)],
]


@pytest.mark.parametrize("prb,sets,tti0,short", [(25, UE_SETS_25, 7, False), (100, UE_SETS_100, 18, False), (25, UE_SETS_25, 2, True)])
def test_ul_grants_vs_oracle(hp, prb, sets, tti0, short):
    """Every PUSCH: estimator noise figure, de-precoded symbols, de-interleaved LLRs (<= 1 LSB on <= 0.1 %), per-block pass counts, CRC flag
    and bytes equal the oracle's for that UE on the summed time signal of its subframe. short: a cell-wide shortened subframe (11 data symbols,
    the last one left to the SRS), which moves the UCI columns and every Q'."""
    from lte_sim import UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(4100 + prb)
    nsf = len(sets)
    dm = dict(cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=False)
    iq, ues, grants, uci = [], [], [], []
    for b, ue_list in enumerate(sets):
        x, sig = None, []
        for u, (L, n0, n1, mod, tbs, n_dmrs, snr) in enumerate(ue_list):
            rnti = 0x100 + 16 * b + u
            cfg = UlConfig(prb, 11, mod, tbs, L, n0, n_dmrs=n_dmrs, rnti=rnti, n_prb_slot1=n1 if n1 != n0 else None, shortened=short, **dm)
            gain = (0.7 + 0.1 * u) * np.exp(0.3j * (u + 1))
            # every other PUSCH also carries HARQ-ACK (1 or 2 bits) and, every third, a rank indication
            O_ack, O_ri = ((u + b) % 2) * (1 + (u % 2)), 1 if (u + b) % 3 == 0 else 0
            if L == 1:
                O_ack = O_ri = 0
            O_cqi = (8 if b % 2 == 0 else 20) if (u == 0 and L >= 6) else 0  # block-coded and convolutionally coded reports
            ack, ri = tuple(int(v) for v in rng.integers(0, 2, O_ack)), tuple(int(v) for v in rng.integers(0, 2, O_ri))
            cqi = tuple(int(v) for v in rng.integers(0, 2, O_cqi))
            uci.append((O_ack, ack, O_ri, ri, O_cqi, cqi))
            y, data = make_ul_subframe(cfg, tti0 + b, rng, amp=0.1, gain=gain, ack=ack, I_offset_ack=9, ri=ri, I_offset_ri=8, cqi=cqi, I_offset_cqi=7)
            sig.append(np.sqrt(0.01 * abs(gain) ** 2 * cfg.M_sc / cfg.N / 2) * 10 ** (-snr / 20))  # the noise level that gives this UE `snr` per RE
            x = y if x is None else x + y
            ues.append((b, cfg, data))
            grants.append(hp.UlGrant.make(b, rnti, L, n0, mod, tbs, n_dmrs=n_dmrs, n_prb_slot1=n1, ack_len=O_ack, I_offset_ack=9, ri_len=O_ri, I_offset_ri=8,
                                          cqi_len=O_cqi, I_offset_cqi=7))
        x = x + min(sig) * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))  # one receiver noise: every UE at its SNR or better
        iq.append(x.astype(np.complex64))
    max_tbs = max(g.tbs for g in grants)
    rx = hp.UlRx(11, prb, 0x1234, 1, max_tbs, 6, 0, 0, 6, nsf, 2, 5, True, False, max_grants=len(grants), shortened=short)
    tb, ok = rx.decode_grants(np.stack(iq), tti0, grants)
    n = len(grants)
    res = rx.debug(20, np.float32, n * 5).reshape(n, 5)
    Cmax = -(-max_tbs // 6120) if max_tbs > 6120 else 1
    n_ok = 0
    zoff_of = {}
    order = sorted(range(n), key=lambda p: (grants[p].L_prb, grants[p].n_dmrs))
    off = 0
    for p in order:
        zoff_of[p] = off
        off += (11 if short else 12) * 12 * grants[p].L_prb
    d_all = rx.debug(21, np.complex64, off)
    e_rows = rx.debug(22, np.int16, n * ((12 * 12 * prb * 8 + 15) & ~15)).reshape(n, -1)
    acks, ris = rx.grants_uci()
    cqis, cqi_ok = rx.grants_cqi()
    n_cqi = 0
    for p, (b, cfg, data) in enumerate(ues):
        O_ack, ack, O_ri, ri, O_cqi, cqi = uci[p]
        r = oracle_ul_rx(cfg, iq[b], tti0 + b, keep=True, O_ack=O_ack, I_offset_ack=9, O_ri=O_ri, I_offset_ri=8, O_cqi=O_cqi, I_offset_cqi=7)
        if O_cqi:
            n_cqi += 1
            assert tuple(cqis[p][:O_cqi]) == cqi == tuple(r["cqi"]) and bool(cqi_ok[p]) == r["cqi_ok"] and (r["cqi_ok"] or O_cqi <= 11), p
        assert tuple(acks[p][:O_ack]) == ack == tuple(r["ack"][:O_ack]) and tuple(ris[p][:O_ri]) == ri == tuple(r["ri"][:O_ri]), (p, acks[p], ack, ris[p], ri)
        assert abs(res[p, 0] - r["noise"]) <= 1e-4 * abs(r["noise"]), p
        close_c(d_all[zoff_of[p]:zoff_of[p] + cfg.nof_re], r["d"], "d of PUSCH %d" % p)
        diff = np.abs(e_rows[p][:len(r["g"])].astype(np.int32) - r["g"].astype(np.int32))  # the whole row: the CQI report's LLRs, then the UL-SCH's
        assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size, p
        if diff.max() == 0 or r["ok"]:
            assert bool(ok[p]) == r["ok"], p
        if r["ok"]:
            n_ok += 1
            assert np.array_equal(tb[p][:cfg.tbs // 8 + 3], r["tb"]) and np.array_equal(tb[p][:cfg.tbs // 8], data), p
    assert n_ok >= n - 2 and n_cqi >= 2, (n_ok, n, n_cqi)
    rx.free()


def test_ul_grants_harq_and_changing_grants(hp):
    """Slot p of the object is PUSCH p's soft buffer across calls: a first call whose transmissions fail, a second with rv 2 in other subframes
    and - for one UE - other PRBs (adaptive retransmission: same transport block, new allocation size is not allowed, new position is);
    verdicts, pass counts and bytes per transmission equal the oracle's OrcHarq chain; then the slots are re-used for new transport blocks
    of other sizes (new_data)."""
    from lte_sim import OrcHarq, UlConfig, make_ul_subframe, oracle_ul_rx
    rng = np.random.default_rng(4300)
    prb, dm = 50, dict(cyclic_shift=1, delta_ss=3, group_hopping=False, sequence_hopping=False)
    # L, n_prb, mod, tbs, n_dmrs, snr (of the time signal: per RE it is 10 log10(N / M_sc) higher), ~2 dB under what a single transmission needs
    ue = [(20, 0, 2, 7736, 0, 2.7), (24, 20, 3, 15264, 3, 8.7), (5, 45, 1, 776, 6, -11.0)]
    rx = hp.UlRx(5, prb, 0x1234, 1, 15264, 6, 0, 0, 6, 4, 1, 3, False, False, max_grants=3)
    harq, data = [None] * 3, [None] * 3
    n_first_fail, n_retx_ok, inexact, done = 0, 0, [False] * 3, [False] * 3
    for n, (rv, tti0, sfs, shift) in enumerate(((0, 3, (0, 1, 2), 0), (2, 11, (2, 1, 1), 1))):
        iq, sig = [np.zeros(rx.sf_len, np.complex128) for _ in range(3)], [[] for _ in range(3)]
        grants, cfgs = [], []
        for u, (L, n0, mod, tbs, n_dmrs, snr) in enumerate(ue):
            n0 = n0 + (shift if u == 0 else 0) * 3
            cfg = UlConfig(prb, 5, mod, tbs, L, n0, n_dmrs=n_dmrs, rnti=0x200 + u, **dm)
            if harq[u] is None:
                harq[u] = OrcHarq(cfg)
            y, data[u] = make_ul_subframe(cfg, tti0 + sfs[u], rng, amp=0.1, rv=rv, data=data[u])
            iq[sfs[u]] = iq[sfs[u]] + y
            sig[sfs[u]].append(np.sqrt(0.01 * cfg.M_sc / cfg.N / 2) * 10 ** (-snr / 20))
            cfgs.append(cfg)
            grants.append(hp.UlGrant.make(sfs[u], 0x200 + u, L, n0, mod, tbs, n_dmrs=n_dmrs, rv=rv, new_data=n == 0))
        for b in range(3):
            if sig[b]:
                iq[b] = iq[b] + min(sig[b]) * (rng.standard_normal(rx.sf_len) + 1j * rng.standard_normal(rx.sf_len))
        x = np.stack(iq).astype(np.complex64)
        tb, ok = rx.decode_grants(x, tti0, grants)
        e_rows = rx.debug(22, np.int16, 3 * ((12 * 12 * prb * 8 + 15) & ~15)).reshape(3, -1)
        for u, cfg in enumerate(cfgs):
            if done[u]:
                continue  # an acknowledged block is not scheduled again (and the reference keeps no bytes of it, sch.c:393-403)
            r = oracle_ul_rx(cfg, x[sfs[u]], tti0 + sfs[u], keep=True, harq=harq[u], rv=rv, new_data=n == 0)
            diff = np.abs(e_rows[u][:cfg.nbits].astype(np.int32) - r["g"].astype(np.int32))
            assert diff.max() <= 1 and (diff != 0).sum() <= 1e-3 * diff.size, (n, u)
            inexact[u] = inexact[u] or diff.max() != 0  # an LLR one LSB off may turn a marginal block's verdict
            if inexact[u] and bool(ok[u]) != r["ok"]:
                continue
            assert bool(ok[u]) == r["ok"], (n, u, int(diff.max()))
            n_first_fail += n == 0 and not r["ok"]
            if r["ok"]:
                assert np.array_equal(tb[u][:cfg.tbs // 8 + 3], r["tb"]) and np.array_equal(tb[u][:cfg.tbs // 8], data[u])
                n_retx_ok += n > 0
                done[u] = True
    assert n_first_fail >= 1 and n_retx_ok >= 1, (n_first_fail, n_retx_ok)
    # new transport blocks of other sizes in the same slots
    ue2 = [(8, 10, 1, 1256, 2, 6.0), (30, 20, 2, 12216, 5, 12.0)]
    iq, grants, exp = np.zeros((1, rx.sf_len), np.complex64), [], []
    for u, (L, n0, mod, tbs, n_dmrs, snr) in enumerate(ue2):
        cfg = UlConfig(prb, 5, mod, tbs, L, n0, n_dmrs=n_dmrs, rnti=0x300 + u, **dm)
        y, d = make_ul_subframe(cfg, 29, rng, snr_db=snr, amp=0.1)
        iq[0] += y
        grants.append(hp.UlGrant.make(0, 0x300 + u, L, n0, mod, tbs, n_dmrs=n_dmrs))
        exp.append((cfg, d))
    tb, ok = rx.decode_grants(iq, 29, grants)
    for u, (cfg, d) in enumerate(exp):
        r = oracle_ul_rx(cfg, iq[0], 29)
        assert r["ok"] and ok[u] and np.array_equal(tb[u][:cfg.tbs // 8], d), u
    rx.free()


def test_ul_grants_equal_fixed_pipeline(hp):
    """The same grant in every subframe through both entry points: identical bytes and flags."""
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(4500)
    prb, L, n0, mod, tbs, nsf = 25, 10, 5, 2, 4008, 6
    cfg = UlConfig(prb, 11, mod, tbs, L, n0, n_dmrs=3, cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=True)
    iq = np.stack([make_ul_subframe(cfg, 4 + b, rng, snr_db=9.0, amp=0.1)[0] for b in range(nsf)])
    rx = hp.UlRx(11, prb, 0x1234, mod, tbs, L, n0, 3, 6, nsf, 2, 5, True, True)
    tb0, ok0 = rx.decode(iq, 4)
    tb1, ok1 = rx.decode_grants(iq, 4, [hp.UlGrant.make(b, 0x1234, L, n0, mod, tbs, n_dmrs=3) for b in range(nsf)])
    assert np.array_equal(ok0, ok1) and np.array_equal(tb0, tb1[:, :tbs // 8 + 3]) and 0 < ok0.sum()
    rx.free()


def test_ul_grants_argument_errors(hp):
    import ctypes as C
    rx = hp.UlRx(1, 25, 0x1234, 2, 4008, 10, 5, 0, 6, 2, max_grants=2)
    iq = np.zeros((2, rx.sf_len), np.complex64)
    G = hp.UlGrant.make
    for bad in ([G(2, 1, 10, 0, 2, 4008)], [G(0, 1, 7, 0, 2, 4008)], [G(0, 1, 10, 16, 2, 4008)], [G(0, 1, 10, 0, 4, 4008)], [G(0, 1, 10, 0, 2, 4016)],
                [G(0, 1, 10, 0, 2, 6200)], [G(0, 1, 10, 0, 2, 4008, n_dmrs=8)], [G(0, 1, 10, 0, 2, 4008, rv=4)], [G(0, 1, 10, 0, 2, 4008)] * 3,
                # no transport block (a UCI-only PUSCH is not served): an error code, not a division by the block count (round-2 advice)
                [G(0, 1, 10, 0, 2, 0)], [G(0, 1, 10, 0, 2, 0, ack_len=1, I_offset_ack=9)], [G(0, 1, 10, 0, 2, 4008), G(1, 1, 10, 0, 2, 0)]):
        with pytest.raises(RuntimeError):
            rx.decode_grants(iq, 0, bad)
    tb, ok = rx.decode_grants(iq, 0, [])
    assert len(ok) == 0
    for bad in ([G(0, 1, 10, 0, 2, 4008, ack_len=3)], [G(0, 1, 10, 0, 2, 4008, ack_len=1, I_offset_ack=15)], [G(0, 1, 10, 0, 2, 4008, ri_len=1, I_offset_ri=13)],
                [G(0, 1, 10, 0, 2, 4008, cqi_len=65)], [G(0, 1, 10, 0, 2, 4008, cqi_len=4, I_offset_cqi=0)]):
        with pytest.raises(RuntimeError):
            rx.decode_grants(iq, 0, bad)
    rx.free()


@pytest.mark.parametrize("prb,sets,tti0,short", [(25, UE_SETS_25, 7, False), (100, UE_SETS_100, 18, False), (25, UE_SETS_25, 2, True)])
def test_ul_tx_grants_vs_oracle(hp, prb, sets, tti0, short):
    """srslte_hip_ul_tx_batch_grants: the composite signal of all PUSCHs of every subframe from ONE call (own allocation, hopping, DMRS shift, RNTI,
    modulation, transport block, rv, HARQ-ACK / RI / CQI report each) = the sum of the oracle's per-UE signals, in the frequency-domain grid
    (debug buffer) and in the time samples."""
    from lte_sim import UlConfig, make_ul_subframe
    rng = np.random.default_rng(4700 + prb)
    nsf = len(sets)
    dm = dict(cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=False)
    grants, datas, acks, ris, cqis = [], [], [], [], []
    exp_iq = np.zeros((nsf, 15 * {25: 384, 100: 1536}[prb]), np.complex128)
    exp_grid = np.zeros((nsf, 14 * 12 * prb), np.complex128)
    for b, ue_list in enumerate(sets):
        for u, (L, n0, n1, mod, tbs, n_dmrs, _) in enumerate(ue_list):
            rnti, rv = 0x100 + 16 * b + u, (u + b) % 4
            cfg = UlConfig(prb, 11, mod, tbs, L, n0, n_dmrs=n_dmrs, rnti=rnti, n_prb_slot1=n1 if n1 != n0 else None, shortened=short, **dm)
            O_ack, O_ri = ((u + b) % 2) * (1 + (u % 2)), 1 if (u + b) % 3 == 0 else 0
            if L == 1:
                O_ack = O_ri = 0
            O_cqi = (8 if b % 2 == 0 else 20) if (u == 0 and L >= 6) else 0
            ack, ri = tuple(int(v) for v in rng.integers(0, 2, O_ack)), tuple(int(v) for v in rng.integers(0, 2, O_ri))
            cqi = tuple(int(v) for v in rng.integers(0, 2, O_cqi))
            k = {}
            y, data = make_ul_subframe(cfg, tti0 + b, rng, ack=ack, I_offset_ack=9, ri=ri, I_offset_ri=8, cqi=cqi, I_offset_cqi=7, rv=rv, keep=k)
            exp_iq[b] += y
            exp_grid[b] += k["grid"]
            grants.append(hp.UlGrant.make(b, rnti, L, n0, mod, tbs, n_dmrs=n_dmrs, n_prb_slot1=n1, rv=rv, ack_len=O_ack, I_offset_ack=9, ri_len=O_ri, I_offset_ri=8,
                                          cqi_len=O_cqi, I_offset_cqi=7))
            datas.append(data); acks.append(ack); ris.append(ri); cqis.append(cqi)
    tx = hp.UlTx(11, prb, 0x1234, 1, max(g.tbs for g in grants), 6, 0, 0, nsf, 2, 5, True, False, shortened=short, max_grants=len(grants))
    iq = tx.encode_grants(datas, tti0, nsf, grants, ack=acks, ri=ris, cqi=cqis)
    grid = tx.debug(4, np.complex64, nsf * 14 * 12 * prb).reshape(nsf, -1)
    for b in range(nsf):
        close_c(grid[b], exp_grid[b].astype(np.complex64), "grid sf %d" % b)
        close_c(iq[b], exp_iq[b].astype(np.complex64), "iq sf %d" % b)
    tx.free()


def test_ul_grants_with_puschs_without_ulsch_data(hp):
    """Per-PUSCH grants with tbs = 0 (a CQI report and no transport block, sch.c:943-975,:1157-1174) next to data-bearing PUSCHs in the same
    subframes: the transmit call's composite signal = the sum of the oracle's per-UE signals (grid and time samples), and the receive call on
    it (noise-free) returns every transport block, every report with its CRC flag, every ACK / RI; the rows of the CQI-only PUSCHs report
    no transport block."""
    from lte_sim import UlConfig, make_ul_subframe
    prb, nsf, tti0 = 25, 5, 6
    rng = np.random.default_rng(5100)
    dm = dict(cyclic_shift=2, delta_ss=5, group_hopping=True, sequence_hopping=False)
    ues = [(10, 0, 2, 4008, 0, 1, 0), (4, 12, 1, 0, 20, 1, 1), (3, 18, 1, 0, 6, 2, 0), (2, 22, 2, 1544 // 2 // 8 * 8, 0, 0, 0)]  # L, n_prb, mod, tbs, O_cqi, O_ack, O_ri
    ues[3] = (2, 22, 2, 776, 0, 0, 0)
    grants, datas, acks, ris, cqis = [], [], [], [], []
    exp_iq, exp_grid = np.zeros((nsf, 15 * 384), np.complex128), np.zeros((nsf, 14 * 12 * prb), np.complex128)
    for b in range(nsf):
        for u, (L, n0, mod, tbs, O_cqi, O_ack, O_ri) in enumerate(ues):
            rnti = 0x200 + 8 * b + u
            cfg = UlConfig(prb, 11, mod, tbs, L, n0, n_dmrs=(u + b) % 8, rnti=rnti, **dm)
            ack, ri = tuple(int(v) for v in rng.integers(0, 2, O_ack)), tuple(int(v) for v in rng.integers(0, 2, O_ri))
            cqi = tuple(int(v) for v in rng.integers(0, 2, O_cqi))
            k = {}
            y, data = make_ul_subframe(cfg, tti0 + b, rng, ack=ack, I_offset_ack=9, ri=ri, I_offset_ri=8, cqi=cqi, I_offset_cqi=7, keep=k)
            exp_iq[b] += y
            exp_grid[b] += k["grid"]
            grants.append(hp.UlGrant.make(b, rnti, L, n0, mod, tbs, n_dmrs=(u + b) % 8, ack_len=O_ack, I_offset_ack=9, ri_len=O_ri, I_offset_ri=8,
                                          cqi_len=O_cqi, I_offset_cqi=7))
            datas.append(data); acks.append(ack); ris.append(ri); cqis.append(cqi)
    tx = hp.UlTx(11, prb, 0x1234, 1, 4008, 6, 0, 0, nsf, 2, 5, True, False, max_grants=len(grants))
    iq = tx.encode_grants(datas, tti0, nsf, grants, ack=acks, ri=ris, cqi=cqis)
    grid = tx.debug(4, np.complex64, nsf * 14 * 12 * prb).reshape(nsf, -1)
    for b in range(nsf):
        close_c(grid[b], exp_grid[b].astype(np.complex64), "grid sf %d" % b)
        close_c(iq[b], exp_iq[b].astype(np.complex64), "iq sf %d" % b)
    tx.free()
    rx = hp.UlRx(11, prb, 0x1234, 1, 4008, 6, 0, 0, 6, nsf, 2, 5, True, False, max_grants=len(grants))
    tb, ok = rx.decode_grants(iq, tti0, grants)
    a, r = rx.grants_uci()
    cq, cq_ok = rx.grants_cqi()
    for p, g in enumerate(grants):
        if g.tbs:
            assert ok[p] and np.array_equal(tb[p, :g.tbs // 8], datas[p]), p
        else:
            assert not ok[p], p
        assert tuple(a[p, :g.ack_len]) == acks[p] and tuple(r[p, :g.ri_len]) == ris[p], p
        if g.cqi_len:
            assert cq_ok[p] and tuple(cq[p, :g.cqi_len]) == cqis[p], p
    rx.free()


def test_ul_grants_tx_rx_round_trip(hp):
    """One transmit call makes 64 subframes with three UEs each (changing sizes, modulations, rv 0, HARQ-ACK on one of them); one receive call
    decodes all 192 PUSCHs: every transport block and every ACK comes back."""
    prb, nsf = 50, 64
    rng = np.random.default_rng(4900)
    shapes = [(10, 0, 1, 1544), (24, 10, 2, 9144), (15, 35, 3, 7992)]  # L, n_prb, mod, tbs
    grants, datas, acks = [], [], []
    for b in range(nsf):
        for u in range(3):
            L, n0, mod, tbs = shapes[(u + b) % 3]
            grants.append(hp.UlGrant.make(b, 0x400 + u, L, n0, mod, tbs, n_dmrs=(u + b) % 8, ack_len=2 if u == 1 else 0, I_offset_ack=8))
            datas.append(rng.integers(0, 256, tbs // 8, dtype=np.uint8))
            acks.append(tuple(int(v) for v in rng.integers(0, 2, 2)) if u == 1 else ())
    tx = hp.UlTx(4, prb, 0x1234, 1, 9144, 6, 0, 0, nsf, max_grants=len(grants))
    iq = tx.encode_grants(datas, 2, nsf, grants, ack=acks)
    tx.free()
    rx = hp.UlRx(4, prb, 0x1234, 1, 9144, 6, 0, 0, 6, nsf, max_grants=len(grants))
    tb, ok = rx.decode_grants(iq, 2, grants)
    a, _ = rx.grants_uci()
    assert ok.all()
    for p, g in enumerate(grants):
        assert np.array_equal(tb[p][:g.tbs // 8], datas[p]), p
        if g.ack_len:
            assert tuple(a[p]) == acks[p], p
    rx.free()


@pytest.mark.parametrize("prb,cell_id,short", [(6, 1, False), (15, 40, True), (25, 10, False), (50, 77, True), (75, 150, False), (100, 501, False),
                                               (33, 44, False), (110, 301, True), (7, 9, False)])
def test_ul_grants_fuzz_round_trip(hp, prb, cell_id, short):
    """Random uplink schedules: every subframe of a 24-TTI run holds 1..4 PUSCHs at random offsets with random valid sizes (2^a 3^b 5^c PRB),
    some hopping between the slots, random cyclic shifts, modulations, the largest accepted transport block under a code rate of ~0.55,
    HARQ-ACK / RI on some. One transmit call, one receive call: every transport block and every UCI bit comes back."""
    rng = np.random.default_rng(5700 + prb)
    nsf, tti0 = 24, int(rng.integers(0, 10))
    nsymb = 11 if short else 12
    valid = [L for L in range(1, prb + 1) if hp.lib().srslte_hip_dft_precoding_valid_prb(L)]

    def valid_tbs(limit):
        for tbs in range(limit - limit % 8, 39, -8):
            rc, s = hp.cbsegm(tbs)
            if rc == 0 and s.F == 0 and s.C2 == 0:
                return tbs, s.C
        return 0, 0
    grants, datas, acks, ris = [], [], [], []
    for b in range(nsf):
        free, k = 0, 0
        while free < prb and k < 4:
            room = prb - free
            cand = [L for L in valid if L <= room]
            if not cand:
                break
            L = int(rng.choice(cand))
            n0 = free
            free += L + int(rng.integers(0, 3))
            mod = int(rng.integers(1, 4))
            nre = nsymb * 12 * L
            tbs, C_ = valid_tbs(min(int(0.55 * nre * 2 * mod), 75376))
            if tbs == 0 or nre < 40:
                continue
            O_ack, O_ri = (int(rng.integers(0, 3)), int(rng.integers(0, 2))) if L >= 2 else (0, 0)
            grants.append(hp.UlGrant.make(b, 0x600 + k, L, n0, mod, tbs, n_dmrs=int(rng.integers(0, 8)), ack_len=O_ack, I_offset_ack=6, ri_len=O_ri, I_offset_ri=5))
            datas.append(rng.integers(0, 256, tbs // 8, dtype=np.uint8))
            acks.append(tuple(int(v) for v in rng.integers(0, 2, O_ack)))
            ris.append(tuple(int(v) for v in rng.integers(0, 2, O_ri)))
            k += 1
    # a few PUSCHs hop: slot 1 at another free offset of the same size is hard to guarantee in a packed subframe, so hop the lone ones
    per_sf = {}
    for p_, g in enumerate(grants):
        per_sf.setdefault(g.sf, []).append(p_)
    for b, lst in per_sf.items():
        if len(lst) == 1 and grants[lst[0]].L_prb < prb:
            g = grants[lst[0]]
            g.n_prb_slot1 = int(rng.integers(0, prb - g.L_prb + 1))
    tbs_max = max(g.tbs for g in grants)
    tx = hp.UlTx(cell_id, prb, 0x1234, 1, tbs_max, 6 if prb >= 6 else 1, 0, 0, nsf, 3, 11, True, False, shortened=short, max_grants=len(grants))
    iq = tx.encode_grants(datas, tti0, nsf, grants, ack=acks, ri=ris)
    tx.free()
    rx = hp.UlRx(cell_id, prb, 0x1234, 1, tbs_max, 6 if prb >= 6 else 1, 0, 0, 6, nsf, 3, 11, True, False, shortened=short, max_grants=len(grants))
    tb, ok = rx.decode_grants(iq, tti0, grants)
    a, r = rx.grants_uci()
    assert ok.all(), [(p_, grants[p_].L_prb, grants[p_].mod, grants[p_].tbs) for p_ in np.flatnonzero(ok == 0)][:5]
    for p_, g in enumerate(grants):
        assert np.array_equal(tb[p_][:g.tbs // 8], datas[p_]), p_
        assert tuple(a[p_][:g.ack_len]) == acks[p_] and tuple(r[p_][:g.ri_len]) == ris[p_], p_
    rx.free()
