"""Pins the oracle to the committed golden vectors (generated from the reference build by tests/gen_golden.py, plus the
reference's own known-answer data). Runs anywhere: needs only gcc-built oracle/liboracle.so."""
import ctypes as C
import os

import numpy as np
import pytest

from _libs import OrcCell, OrcChestCfg, OrcChestRes, acopy, oracle, p
from lte_sim import DlConfig, oracle_rx

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_tcod_golden_and_kat():
    g = load("tcod.npz")
    for K in (40, 176, 504, 1008, 5824, 6144):
        out = np.zeros(3 * K + 12, np.uint8)
        assert oracle().orc_tcod_encode_bits(p(g["in_%d" % K]), p(out), K) == 0
        assert np.array_equal(out, g["out_%d" % K])
    # reference KAT (turbodecoder_test.h:75-160): the codeword matches except the first tail bit, as the reference's own encoder
    out = np.zeros(3 * 504 + 12, np.uint8)
    oracle().orc_tcod_encode_bits(p(g["kat_in"]), p(out), 504)
    assert list(np.nonzero(out != g["kat_out"])[0]) == [1512]
    # ... and decoding the KAT codeword returns the KAT data (what turbodecoder_test -k exercises)
    llr = (100 * (2 * g["kat_out"].astype(np.int16) - 1)).astype(np.int16)
    hard = np.zeros(504 // 8, np.uint8)
    assert oracle().orc_tdec_run(p(llr), False, 504, 2, p(hard), None) == 0
    assert np.array_equal(np.unpackbits(hard), g["kat_in"])


@pytest.mark.parametrize("K", [40, 176, 504, 1008, 5824, 6144])
def test_tdec_golden(K):
    g = load("tdec.npz")
    per = np.zeros((6, K // 8), np.uint8)
    out = np.zeros(K // 8, np.uint8)
    assert oracle().orc_tdec_run(p(g["llr_%d" % K]), False, K, 6, p(out), p(per)) == 0
    assert np.array_equal(per, g["hard_%d" % K])


@pytest.mark.parametrize("K", [816, 5824])
def test_rm_and_tdec_sb_golden(K):
    g = load("tdec.npz")
    e = g["sb_e_%d" % K]
    w = np.zeros(3 * (K + 32) + 12 + 64, np.int16)
    W = oracle().orc_tdec_autoimp_subblocks(K)
    assert oracle().orc_rm_turbo_rx(p(e), p(w), len(e), K, 0, W) == 0
    assert np.array_equal(w, g["sb_w_%d" % K])
    per = np.zeros((6, K // 8), np.uint8)
    out = np.zeros(K // 8, np.uint8)
    assert oracle().orc_tdec_run(p(w), True, K, 6, p(out), p(per)) == 0
    assert np.array_equal(per, g["sb_hard_%d" % K])
    if K == 816:  # this one converges; the K=5824 vector deliberately does not (residual errors exercise all 6 passes)
        assert np.array_equal(np.unpackbits(per[-1]), g["sb_bits_%d" % K])


@pytest.mark.parametrize("mod", [0, 1, 2, 3, 4])
def test_demod_golden(mod):
    g = load("demod.npz")
    x = acopy(g["sym_%d" % mod])
    nsym, qm = len(x) // 2, (1 if mod == 0 else 2 * mod)
    for name, fn, dt in (("llr", "orc_demod_soft_f", np.float32), ("llr_s", "orc_demod_soft_s", np.int16), ("llr_b", "orc_demod_soft_b", np.int8)):
        out = np.zeros(nsym * qm, dt)
        assert getattr(oracle(), fn)(mod, p(x), p(out), nsym) == 0
        assert np.array_equal(out, g["%s_%d" % (name, mod)]), (name, mod)


@pytest.mark.parametrize("tag", ["6_0", "6_1", "25_0", "25_1"])
def test_chest_golden(tag):
    g = load("chest.npz")
    prb, cid, sf_idx, ci = [int(v) for v in g["meta_" + tag]]
    cfg = OrcChestCfg()
    if ci == 0:
        cfg.filter_coef[0], cfg.filter_coef[1] = 4.0, 1.0
    else:
        cfg.interpolate_subframe, cfg.cfo_estimate_enable = True, True
        cfg.filter_coef[0], cfg.filter_coef[1] = 4.0, 2.0
    cell = OrcCell(cid, prb, 1, True)
    ce, res = np.zeros(14 * 12 * prb, np.complex64), OrcChestRes()
    assert oracle().orc_chest_dl(C.byref(cell), sf_idx, C.byref(cfg), p(g["grid_%d" % prb]), p(ce), C.byref(res)) == 0
    ref = g["ce_" + tag].view(np.complex64)
    assert np.abs(ce - ref).max() <= 1e-4 * max(np.abs(ref).max(), np.sqrt((np.abs(ref) ** 2).mean()))
    scal = np.array([res.noise_estimate, res.noise_estimate_dbm, res.snr_db, res.rsrp, res.rsrp_dbm, res.rsrq, res.rsrq_db, res.rssi_dbm, res.cfo])
    assert np.all(np.abs(scal - g["scal_" + tag]) <= 1e-4 * np.abs(g["scal_" + tag]) + 1e-6)


@pytest.mark.parametrize("tag,prb,mod,tbs,ttis", [("cfg1", 6, 1, 936, (1, 2, 3)), ("cfg2", 100, 3, 75376, (0,))])
def test_dl_chain_golden(tag, prb, mod, tbs, ttis):
    g = load("dl_chain.npz")
    cfg = DlConfig(prb, 1, mod, tbs)
    for t in ttis:
        r = oracle_rx(cfg, g["%s_iq_%d" % (tag, t)], t)
        assert r["ok"] == bool(g["%s_ok_%d" % (tag, t)][0])
        assert np.array_equal(r["iters"], g["%s_iters_%d" % (tag, t)])
        assert np.array_equal(r["tb"], g["%s_tb_%d" % (tag, t)])
