"""srslte_dlsch_decode2 served by the library (one device call per transport block, include/srslte_hip/srslte_compat.h) against the reference's
own srslte_dlsch_decode2 (oracle/_ref/libsrslte_ref.so, sch.c:507-531 -> decode_tb -> decode_tb_cb) on the same LLRs: return code, transport
block bytes, the soft buffer's CRC flags, its kept bytes and - for the blocks that failed - its soft values, through a HARQ sequence on ONE
srslte_softbuffer_rx_t (rv 0, 2, 3, 1: combining, skipping of blocks already decoded), 16- and 8-bit LLRs, one and several code blocks,
N_L = 1 and 2. Struct layouts come from this repository's compat header, which tests/test_abi_layout.py holds against the reference's."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from _libs import ROOT, hip, opaque, oracle, p, ref

pytestmark = pytest.mark.gpu
_cache = {}
SEQ = {}  # (rv, return code, blocks decoded so far, C) of every transmission of every case: the HARQ paths must have been walked


def layout():
    if not _cache:
        structs = {"srslte_sch_t": ["max_iterations", "avg_iterations", "llr_is_8bit", "decoder", "crc_tb", "crc_cb"],
                   "srslte_pdsch_cfg_t": ["grant", "softbuffers"], "srslte_pdsch_grant_t": ["tb", "nof_tb"],
                   "srslte_ra_tb_t": ["mod", "tbs", "rv", "nof_bits", "cw_idx", "enabled"], "srslte_softbuffer_rx_t": ["max_cb", "buffer_f", "data", "cb_crc", "tb_crc"]}
        body = "".join('  printf("%s %%zu\\n", sizeof(%s));\n' % (s, s) + "".join('  printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (s, f, s, f) for f in fs)
                       for s, fs in structs.items())
        src = '#include <stdio.h>\n#include <stddef.h>\n#include "srslte_hip/srslte_compat.h"\nint main(void) {\n' + body + "  return 0;\n}\n"
        with tempfile.TemporaryDirectory() as d:
            c, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
            open(c, "w").write(src)
            subprocess.check_call(["gcc", "-std=c99", "-D_GNU_SOURCE", "-w", "-I" + os.path.join(ROOT, "include"), c, "-o", exe])
            _cache.update({a: int(b) for a, b in (line.split() for line in subprocess.check_output([exe]).decode().splitlines())})
    return _cache


class Side:
    """one srslte_sch_t + srslte_softbuffer_rx_t + srslte_pdsch_cfg_t; `mine`: decoder and decode entry are this library's"""

    def __init__(self, mine, prb, tbs, mod, nbits, llr8, max_it):
        Ly, R, H = layout(), ref(), hip()
        self.mine, self.R, self.H, self.Ly = mine, R, H, Ly
        self.q = opaque(Ly["srslte_sch_t"] + 256)
        if mine:  # what srslte_sch_init does with the replaced translation units: the library's decoder object, the reference's CRC tables
            assert H.srslte_tdec_init(C.c_void_p(C.addressof(self.q) + Ly["srslte_sch_t.decoder"]), 6144) == 0
            assert R.srslte_crc_init(C.c_void_p(C.addressof(self.q) + Ly["srslte_sch_t.crc_tb"]), 0x1864CFB, 24) == 0
            assert R.srslte_crc_init(C.c_void_p(C.addressof(self.q) + Ly["srslte_sch_t.crc_cb"]), 0x1800063, 24) == 0
        else:
            assert R.srslte_sch_init(self.q) == 0
        qb = np.frombuffer(self.q, np.uint8)
        qb[Ly["srslte_sch_t.max_iterations"]:Ly["srslte_sch_t.max_iterations"] + 4].view(np.uint32)[0] = max_it
        qb[Ly["srslte_sch_t.llr_is_8bit"]] = 1 if llr8 else 0
        self.sb = opaque(Ly["srslte_softbuffer_rx_t"] + 64)
        assert R.srslte_softbuffer_rx_init(self.sb, prb) == 0
        R.srslte_softbuffer_rx_reset(self.sb)  # the init leaves the soft buffers as malloc gave them (softbuffer.c:84); the MAC resets per new transport block
        self.cfg = opaque(Ly["srslte_pdsch_cfg_t"] + 64)
        g = np.frombuffer(self.cfg, np.uint8)
        tb0 = Ly["srslte_pdsch_cfg_t.grant"] + Ly["srslte_pdsch_grant_t.tb"]
        for f, v in (("mod", mod), ("tbs", tbs), ("rv", 0), ("nof_bits", nbits), ("cw_idx", 0), ("enabled", 1)):
            o = tb0 + Ly["srslte_ra_tb_t." + f]
            if f == "enabled":
                g[o] = v
            else:
                g[o:o + 4].view(np.uint32)[0] = v
        self.tb0 = tb0
        o = Ly["srslte_pdsch_cfg_t.softbuffers"]
        g[o:o + 8].view(np.uint64)[0] = C.addressof(self.sb)
        self.g = g

    def set(self, rv, nof_tb):
        o = self.tb0 + self.Ly["srslte_ra_tb_t.rv"]
        self.g[o:o + 4].view(np.uint32)[0] = rv
        o = self.Ly["srslte_pdsch_cfg_t.grant"] + self.Ly["srslte_pdsch_grant_t.nof_tb"]
        self.g[o:o + 4].view(np.uint32)[0] = nof_tb

    def decode(self, e, data, nof_layers):
        fn = self.H.srslte_dlsch_decode2 if self.mine else self.R.srslte_dlsch_decode2
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32]
        return fn(self.q, self.cfg, p(e), p(data), 0, nof_layers)

    def soft(self, C_, n):
        """(cb_crc [C], buffer_f [C][n] int16 view, data [C][768], tb_crc)"""
        sbb = np.frombuffer(self.sb, np.uint8)
        L = self.Ly

        def ptr(f):
            return int(sbb[L["srslte_softbuffer_rx_t." + f]:L["srslte_softbuffer_rx_t." + f] + 8].view(np.uint64)[0])
        crc = np.ctypeslib.as_array((C.c_uint8 * C_).from_address(ptr("cb_crc"))).copy()
        bf = [np.ctypeslib.as_array((C.c_int16 * n).from_address(int(np.ctypeslib.as_array((C.c_uint64 * C_).from_address(ptr("buffer_f")))[c]))).copy() for c in range(C_)]
        dt = [np.ctypeslib.as_array((C.c_uint8 * 768).from_address(int(np.ctypeslib.as_array((C.c_uint64 * C_).from_address(ptr("data")))[c]))).copy() for c in range(C_)]
        return crc, bf, dt, int(sbb[L["srslte_softbuffer_rx_t.tb_crc"]])


@pytest.mark.skipif(ref() is None, reason="oracle/_ref/libsrslte_ref.so not built")
@pytest.mark.parametrize("prb,tbs,mod,nre,Nl,llr8,snr", [(25, 4008, 2, 3000, 1, False, -5.0), (100, 75376, 3, 14580, 1, False, 3.5), (100, 30576, 2, 14580, 2, False, -1.5),
                                                         (6, 328, 1, 600, 1, False, -5.5), (50, 21384, 3, 6600, 1, True, 1.5), (100, 75376, 3, 14580, 1, True, 4.0)])
def test_dlsch_decode2_vs_reference_with_harq(prb, tbs, mod, nre, Nl, llr8, snr):
    from lte_sim import OrcSchCfg
    orc, rng = oracle(), np.random.default_rng(tbs + nre)
    Qm = 2 * mod
    nbits = nre * Qm
    data = rng.integers(0, 256, tbs // 8, dtype=np.uint8)
    seg = hip_seg(tbs)
    K, C_ = seg[0], seg[1]
    n_soft = 3 * (K + 32) + 12
    mine, theirs = Side(True, prb, tbs, mod, nbits, llr8, 4), Side(False, prb, tbs, mod, nbits, llr8, 4)
    outcomes = set()
    for rv in (0, 2, 3, 1):
        sch = OrcSchCfg(tbs, nbits, Qm * Nl, rv, 4)
        bits = np.zeros(nbits, np.uint8)
        assert orc.orc_dlsch_encode(C.byref(sch), p(data), p(bits)) == 0
        scale = 20.0 if llr8 else 100.0
        llr = scale * ((2.0 * bits - 1) + 10 ** (-snr / 20) * rng.standard_normal(nbits))
        e = np.clip(np.round(llr), -127, 127).astype(np.int8) if llr8 else np.clip(np.round(llr), -32000, 32000).astype(np.int16)
        res = []
        for s in (mine, theirs):
            s.set(rv, 2 if Nl == 2 else 1)
            out = np.zeros(tbs // 8 + 64, np.uint8)
            rc = s.decode(e.copy(), out, 1)
            res.append((rc, out, s.soft(C_, n_soft if not llr8 else (n_soft + 1) // 2)))
        (rc_m, out_m, (crc_m, bf_m, dt_m, tbc_m)), (rc_r, out_r, (crc_r, bf_r, dt_r, tbc_r)) = res
        assert rc_m == rc_r, (rv, rc_m, rc_r)
        assert np.array_equal(crc_m, crc_r) and tbc_m == tbc_r, (rv, crc_m, crc_r)
        assert np.array_equal(out_m[:tbs // 8 + 3], out_r[:tbs // 8 + 3]), rv
        for c in range(C_):
            if crc_r[c]:
                if not tbc_r:
                    assert np.array_equal(dt_m[c][:(K - (24 if C_ > 1 else 0)) // 8], dt_r[c][:(K - (24 if C_ > 1 else 0)) // 8]), (rv, c)
            else:  # a block that failed: its soft buffer is what the next transmission adds to
                d = np.nonzero(bf_m[c] != bf_r[c])[0]
                assert d.size == 0, (rv, c, d.size, d[:12].tolist(), bf_m[c][d[:12]].tolist(), bf_r[c][d[:12]].tolist(), K)
        outcomes.add(rc_r)
        SEQ.setdefault((prb, tbs, llr8), []).append((rv, rc_r, int(crc_r.sum()), C_))
        if rc_r == 0:
            assert np.array_equal(out_m[:tbs // 8], data)
            break
    assert 0 in outcomes, "the sequence never decoded: raise the SNR of this case"


def hip_seg(tbs):
    class Seg(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("F", "C", "K1", "K2", "K1_idx", "K2_idx", "C1", "C2", "tbs")]
    s = Seg()
    assert hip().srslte_cbsegm(C.byref(s), tbs) == 0
    return s.K1, s.C


def test_the_harq_paths_were_walked():
    """at least three of the cases above needed a retransmission, one of them with some blocks already decoded and others not"""
    if not SEQ:
        pytest.skip("runs after the cases above")
    retx = [k for k, v in SEQ.items() if v[0][1] != 0]
    partial = [k for k, v in SEQ.items() if any(rc != 0 and 0 < ok < C_ for _, rc, ok, C_ in v)]
    assert len(retx) >= 3 and partial, SEQ
