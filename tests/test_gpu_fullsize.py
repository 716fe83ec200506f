"""BASELINE.json's configurations at their FULL sizes (batch 128 / 512 of 20 MHz subframes), where the CPU oracle is too slow to check every
subframe: size-independent properties instead - transmit -> receive round trips on the device, the gain of HARQ combining, no undetected
errors, agreement between entry points and between the 16- and 8-bit LLR paths - plus the oracle on a sample of the batch."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hp():
    return importlib.import_module("srslte-emane_amd")


def _chest(hp):
    hc = hp.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    return hc


def test_cfg2_full_batch_round_trip_and_oracle_sample(hp):
    """cfg2: 128 subframes of 100 PRB, 64QAM, TBS 75376 (13 blocks of 5824). Device PDSCH transmit pipeline -> device receive pipeline, noise
    free: all 128 transport blocks come back; with noise added on the host, CRC verdicts, pass counts and bytes of a sample of the batch
    equal the oracle's, the rest decode to what was sent wherever the CRC passes, and no wrong block passes its CRC."""
    from lte_sim import DlConfig, oracle_rx
    prb, mod, tbs, B = 100, 3, 75376, 128
    rng = np.random.default_rng(2024)
    data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
    tx = hp.DlTx(1, prb, 1, 0x1234, mod, tbs, B)
    rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, B, True, _chest(hp))
    iq = tx.encode(data, 0)[:, 0, :]
    tb, ok = rx.decode(iq, 0)
    assert ok.all() and np.array_equal(tb[:, :tbs // 8], data)
    sigma = np.sqrt(np.mean(np.abs(iq) ** 2) / 2) * 10 ** (-18.0 / 20)  # about the bench's working point: some blocks need all 6 passes
    noisy = (iq + sigma * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64)
    tb, ok = rx.decode(noisy, 0)
    it = rx.debug(6, np.uint32, B * 13).reshape(B, 13)
    assert 0 < ok.sum() < B or ok.all()
    for b in range(B):
        if ok[b]:
            assert np.array_equal(tb[b][:tbs // 8], data[b]), b  # a passed CRC24A with wrong bytes would be an undetected error
    cfg = DlConfig(prb, 1, mod, tbs)
    for b in (0, 5, 77, 127):
        r = oracle_rx(cfg, noisy[b], b, keep=True)
        assert bool(ok[b]) == r["ok"] and np.array_equal(it[b], r["iters"]), b
        if r["ok"]:
            assert np.array_equal(tb[b], r["tb"])
    # the grid entry point on the grids the pipeline made itself = the IQ entry point, byte for byte
    grid = rx.debug(0, np.complex64, B * 14 * 12 * prb).reshape(B, -1)
    tb2, ok2 = rx.decode_grid(grid, 0)
    assert np.array_equal(ok, ok2) and np.array_equal(tb, tb2)
    tx.free()
    rx.free()


def test_cfg2_full_batch_harq_gain(hp):
    """HARQ at full batch (srslte_hip_dl_rx_batch_harq): at an SNR where a single transmission of MCS 28 almost never decodes, the rv 2
    retransmission combined into the kept soft buffers decodes almost every block that failed; the few that passed in the first round are
    neither combined nor decoded again (sch.c:317-318) and their duplicate retransmission is refused as upstream refuses it; no block that
    passes its CRC is wrong."""
    prb, mod, tbs, B = 100, 3, 75376, 128
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
    tx = hp.DlTx(1, prb, 1, 0x1234, mod, tbs, B)
    rx = hp.DlRx(1, prb, 1, 0x1234, mod, tbs, 6, B, True, _chest(hp))
    oks = []
    for rv, new_data in ((0, True), (2, False)):
        iq = tx.encode(data, 0, rv=rv)[:, 0, :]
        sigma = np.sqrt(np.mean(np.abs(iq) ** 2) / 2) * 10 ** (-15.5 / 20)
        noisy = (iq + sigma * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64)
        tb, ok = rx.decode_harq(noisy, 0, rv, new_data)
        for b in range(B):
            if ok[b]:
                assert np.array_equal(tb[b][:tbs // 8], data[b]), (rv, b)
        oks.append(ok.copy())
    first = oks[0].astype(bool)
    assert first.sum() < B // 4 and oks[1][~first].sum() > 3 * (~first).sum() // 4, (int(first.sum()), int(oks[1].sum()))
    # the MAC would not retransmit a block it has acknowledged; sent anyway, the duplicate is refused as upstream's decode_tb_cb refuses it
    # (sch.c:399-410: the bytes of passed blocks are kept only while the transport block as a whole has failed)
    assert not oks[1][first].any()
    tx.free()
    rx.free()


def test_cfg3_full_batch_uplink_round_trip(hp):
    """cfg3: 128 subframes of 20 MHz uplink, the largest valid SC-FDMA grant (96 of 100 PRB), 16QAM: device PUSCH transmit pipeline (CRC,
    segmentation, turbo encoder, rate matching, channel interleaver, DFT precoding, DMRS, OFDM) -> device receive pipeline: every transport
    block comes back, with HARQ-ACK, rank indication and a CQI report multiplexed in."""
    prb, L, mod, tbs, B = 100, 96, 2, 36696, 128
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
    acks, ris, cqis = rng.integers(0, 2, (B, 2), dtype=np.uint8), rng.integers(0, 2, (B, 1), dtype=np.uint8), rng.integers(0, 2, (B, 20), dtype=np.uint8)
    kw = dict(ack_len=2, I_offset_ack=9, ri_len=1, I_offset_ri=8, cqi_len=20, I_offset_cqi=8)
    tx = hp.UlTx(3, prb, 0x77, mod, tbs, L, 2, 1, B, **kw)
    rx = hp.UlRx(3, prb, 0x77, mod, tbs, L, 2, 1, 6, B, **kw)
    tb, ok = rx.decode(tx.encode(data, 3, ack=acks, ri=ris, cqi=cqis), 3)
    cqi, cqi_ok = rx.cqi()
    assert ok.all() and np.array_equal(tb[:, :tbs // 8], data)
    assert np.array_equal(rx.ack(), acks) and np.array_equal(rx.ri(), ris) and cqi_ok.all() and np.array_equal(cqi, cqis)
    tx.free()
    rx.free()


def test_cfg3_full_batch_uplink_harq_gain(hp):
    """Uplink HARQ at full batch (srslte_hip_ul_tx_batch_rv -> noise -> srslte_hip_ul_rx_batch_harq): at an SNR where a single transmission of
    this grant almost never decodes, the rv 2 retransmission combined into the kept soft buffers decodes almost every block; blocks that
    passed in the first round keep their bytes; no transport block that passes its CRC is wrong; the HARQ-ACK bits multiplexed into each
    transmission are read from that transmission alone."""
    prb, L, mod, tbs, B = 100, 96, 2, 36696, 128
    rng = np.random.default_rng(17)
    data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
    kw = dict(ack_len=2, I_offset_ack=9)
    tx = hp.UlTx(3, prb, 0x77, mod, tbs, L, 2, 1, B, **kw)
    rx = hp.UlRx(3, prb, 0x77, mod, tbs, L, 2, 1, 6, B, **kw)
    oks = []
    for rv, new_data in ((0, True), (2, False)):
        acks = rng.integers(0, 2, (B, 2), dtype=np.uint8)
        iq = tx.encode(data, 3, ack=acks, rv=rv)
        sigma = np.sqrt(np.mean(np.abs(iq) ** 2) / 2) * 10 ** (-7.5 / 20)
        noisy = (iq + sigma * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64)
        tb, ok = rx.decode_harq(noisy, 3, rv, new_data)
        assert np.array_equal(rx.ack(), acks), rv
        for b in range(B):
            if ok[b]:
                assert np.array_equal(tb[b][:tbs // 8], data[b]), (rv, b)
        oks.append(ok.copy())
    first = oks[0].astype(bool)
    assert first.sum() < B // 4 and oks[1][~first].sum() > 3 * (~first).sum() // 4, (int(first.sum()), int(oks[1].sum()))
    # the MAC would not retransmit a block it has acknowledged; sent anyway, the duplicate is refused as upstream's decode_tb_cb refuses it
    # (sch.c:399-410: the bytes of passed blocks are kept only while the transport block as a whole has failed)
    assert not oks[1][first].any()
    tx.free()
    rx.free()


def test_full_batch_uplink_grants_four_ues_per_subframe(hp):
    """Per-PUSCH grants at full batch: 128 subframes of a 100-PRB cell, four UEs per subframe (24 PRB each, own RNTI, cyclic shift, modulation
    and transport block), their signals made by four device transmit pipelines and summed: all 512 transport blocks come back through
    one srslte_hip_ul_rx_batch_grants call, each in the row of its grant; then the same batch with the grants listed in another order
    gives the same blocks in the permuted rows."""
    prb, B = 100, 128
    rng = np.random.default_rng(23)
    ues = [(24, 0, 1, 4584, 0), (24, 24, 2, 9144, 3), (24, 48, 3, 15264, 5), (24, 72, 2, 11064, 6)]  # L, n_prb, mod, tbs, n_dmrs
    iq, datas = None, []
    for u, (L, n0, mod, tbs, nd) in enumerate(ues):
        tx = hp.UlTx(3, prb, 0x100 + u, mod, tbs, L, n0, nd, B)
        d = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
        y = tx.encode(d, 5)
        iq = y.copy() if iq is None else iq + y
        datas.append(d)
        tx.free()
    grants = [hp.UlGrant.make(b, 0x100 + u, L, n0, mod, tbs, n_dmrs=nd) for b in range(B) for u, (L, n0, mod, tbs, nd) in enumerate(ues)]
    rx = hp.UlRx(3, prb, 0x77, 2, max(u[3] for u in ues), 6, 0, 0, 6, B, max_grants=len(grants))
    tb, ok = rx.decode_grants(iq, 5, grants)
    assert ok.all()
    for u, (L, n0, mod, tbs, nd) in enumerate(ues):
        assert np.array_equal(tb[u::4, :tbs // 8], datas[u]), u
    perm = rng.permutation(len(grants))
    tb2, ok2 = rx.decode_grants(iq, 5, [grants[i] for i in perm])
    assert ok2.all()
    for r, i in enumerate(perm):  # the bytes of each row's own transport block (behind them a row keeps what an earlier, longer block left)
        nb = grants[i].tbs // 8 + 3
        assert np.array_equal(tb2[r, :nb], tb[i, :nb]), (r, i)
    rx.free()


def test_cfg5_full_batch_256qam(hp):
    """cfg5: 512 subframes, 256QAM (TBS 97896), through transmit -> receive on the device, noise free; then the 8-bit LLR path of the same
    batch agrees with the 16-bit one on every transport block."""
    prb, mod, tbs, B = 100, 4, 97896, 512
    rng = np.random.default_rng(13)
    data = rng.integers(0, 256, (B, tbs // 8), dtype=np.uint8)
    tx = hp.DlTx(2, prb, 1, 0x4321, mod, tbs, B)
    iq = tx.encode(data, 0)[:, 0, :].copy()
    tx.free()
    for llr8 in (False, True):
        rx = hp.DlRx(2, prb, 1, 0x4321, mod, tbs, 6, B, True, _chest(hp), llr_8bit=llr8)
        tb, ok = rx.decode(iq, 0)
        assert ok.all() and np.array_equal(tb[:, :tbs // 8], data), llr8
        rx.free()


def test_cfg4_eight_ues_on_one_device():
    """BASELINE cfg4 as far as a one-GPU box goes (a rehearsal, not a scaling figure): the eight UE identities of sharding.ue_for_rank as eight
    pipeline objects on eight streams of one device, 20 MHz / 64QAM MCS 28 / 13 code blocks each; every UE's delivered transport blocks equal
    what was sent, none is wrong, and a sample of every UE's subframes agrees with the oracle chain (verdict and bytes)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cfg4_one_device", os.path.join(ROOT, "scripts", "cfg4_one_device.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = m.run(ues=8, batch=16, steps=2, snr=18.0, oracle_sample=2, quiet=True, warm_s=0.0, timed_s=0.0)
    c = out["config"]
    assert out["rehearsal"] and c["undetected_errors"] == 0 and c["oracle_sample"] == 16 and c["oracle_sample_agrees"] == 16, c
    assert len({u["rnti"] for u in c["per_ue"]}) == 8 and len({u["cell_id"] for u in c["per_ue"]}) == 8
    assert sum(u["delivered"] for u in c["per_ue"]) >= 8 * 16 * 0.5, c["per_ue"]


def test_pool_one_call_per_batch(hp):
    """srslte_hip_dl_rx_pool_*: one host thread, ONE submission call per batch; the pool round-robins its objects and streams. Eight batches of a
    25-PRB cell through a pool of three: every batch's transport blocks (device buffers and the pinned host record the pool fills on the batch's
    own stream) equal those of a plain object, tickets complete in any order of waiting, and a submission re-uses an object only after its
    previous batch."""
    import ctypes as C
    from lte_sim import DlConfig, make_subframe
    L = hp.lib()
    L.srslte_hip_dl_rx_pool_create.restype = C.c_void_p
    L.srslte_hip_dl_rx_pool_create.argtypes = [C.c_void_p, C.c_uint32]
    L.srslte_hip_dl_rx_pool_submit.restype = C.c_int64
    L.srslte_hip_dl_rx_pool_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.srslte_hip_dl_rx_pool_wait.argtypes = [C.c_void_p, C.c_int64]
    L.srslte_hip_dl_rx_pool_destroy.argtypes = [C.c_void_p]
    prb, mod, tbs, B, nb = 25, 2, 4008, 6, 8
    rng = np.random.default_rng(4)
    cfg = DlConfig(prb, 3, mod, tbs)
    batches = [[make_subframe(cfg, 10 * n + b, rng, snr_db=12.0, amp=0.1) for b in range(B)] for n in range(nb)]
    rx = hp.DlRx(3, prb, 1, 0x1234, mod, tbs, 6, B, True, _chest(hp))
    want = [rx.decode(np.stack([s[0] for s in bt]), 10 * n) for n, bt in enumerate(batches)]
    pool = L.srslte_hip_dl_rx_pool_create(C.byref(rx.cfg), 3)
    assert pool
    stride = rx.tb_stride
    d_iq = [hp.DevBuf.from_host(np.stack([s[0] for s in bt])) for bt in batches]
    d_tb, d_ok = [hp.DevBuf(stride * B) for _ in range(nb)], [hp.DevBuf(B) for _ in range(nb)]
    tickets = [L.srslte_hip_dl_rx_pool_submit(pool, d_iq[n].ptr, 10 * n, B, None, d_tb[n].ptr, stride, d_ok[n].ptr, None) for n in range(nb)]
    assert tickets == list(range(nb))
    for n in (5, 0, 7, 3, 1, 2, 4, 6):
        assert L.srslte_hip_dl_rx_pool_wait(pool, tickets[n]) == 0
        tb = d_tb[n].to_host(np.uint8).reshape(B, stride)[:, :tbs // 8 + 3]
        ok = d_ok[n].to_host(np.uint8)[:B]
        assert np.array_equal(ok, want[n][1]) and np.array_equal(tb[:, :tbs // 8 + 3], want[n][0][:, :tbs // 8 + 3]), n
        for b in range(B):
            if ok[b]:
                assert np.array_equal(tb[b, :tbs // 8], batches[n][b][1])
    assert L.srslte_hip_dl_rx_pool_wait(pool, 99) != 0 and L.srslte_hip_dl_rx_pool_submit(pool, None, 0, B, None, d_tb[0].ptr, stride, d_ok[0].ptr, None) < 0
    L.srslte_hip_dl_rx_pool_destroy(pool)
    rx.free()


@pytest.mark.parametrize("prb,mod,tbs,snr,llr8", [(100, 3, 75376, 18.0, False), (25, 2, 4008, 10.0, False), (6, 1, 152, 4.0, False), (100, 3, 75376, 30.0, True)])
def test_results_straight_into_pinned_host_memory(hp, prb, mod, tbs, snr, llr8):
    """d_tb / d_tb_ok of the C-ABI may point into device-visible (pinned) host memory: the pipeline's last phase - the decoder itself for blocks of
    more than 800 bits (tdec_set_tb_direct), tb_asm / tb_crc otherwise - stores the transport blocks and CRC flags there, no copy after the batch
    (bench.py's N = 1 line, +2.3 %). Same bytes and flags as with a device record, for decodable and undecodable blocks."""
    import torch
    from lte_sim import DlConfig, make_subframe
    B = 8
    rng = np.random.default_rng(prb + mod)
    cfg = DlConfig(prb, 5, mod, tbs, llr8=llr8)
    sub = [make_subframe(cfg, t, rng, snr_db=snr, amp=0.1) for t in range(B)]
    iq = np.stack([s[0] for s in sub])
    rx = hp.DlRx(5, prb, 1, 0x1234, mod, tbs, 6, B, True, _chest(hp), llr_8bit=llr8)
    tb_d, ok_d = rx.decode(iq, 0)
    stride = rx.tb_stride
    rec = torch.full((stride * B + B,), 0xA5, dtype=torch.uint8).pin_memory()
    rx2 = hp.DlRx(5, prb, 1, 0x1234, mod, tbs, 6, B, True, _chest(hp), llr_8bit=llr8, out_ptrs=(rec.data_ptr(), rec.data_ptr() + stride * B))
    d_iq = hp.DevBuf.from_host(iq)
    for stage in range(6):
        assert rx2.stage(stage, d_iq.ptr, 0, B, None) == 0
    hp.sync()
    r = rec.numpy()
    tb_h, ok_h = r[:stride * B].reshape(B, stride), r[stride * B:]
    assert np.array_equal(ok_h, ok_d) and 0 < int(ok_d.sum())
    assert np.array_equal(tb_h[:, :tbs // 8 + 3], tb_d[:, :tbs // 8 + 3])
    for b in range(B):
        if ok_d[b]:
            assert np.array_equal(tb_h[b, :tbs // 8], sub[b][1])
    rx.free()
    rx2.free()
