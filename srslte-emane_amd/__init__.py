"""srslte-emane_amd — MI355X-native drop-in for the sample-level hot path of srsLTE's lib/src/phy.

The product is ``csrc/libsrslte_phy_hip.so`` (hand-written HIP for gfx950 behind a C ABI, see
``include/srslte_hip/phy_hip.h``). This module is only the host-side mirror of the reference's operator
interface used by the tests, ``bench.py`` and ``__graft_entry__``: same names, argument meaning and error
behaviour as the ``srslte_*`` calls, numpy arrays in and out, device buffers managed through the C ABI.

There is NO CPU fallback: importing the native handle without a built library raises, and every call needs a GPU.
(The directory name carries a hyphen; import it with ``importlib.import_module("srslte-emane_amd")``.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsrslte_phy_hip.so")

SRSLTE_SUCCESS, SRSLTE_ERROR, SRSLTE_ERROR_INVALID_INPUTS = 0, -1, -2
MOD_BPSK, MOD_QPSK, MOD_16QAM, MOD_64QAM, MOD_256QAM = range(5)
CRC24A, CRC24B = 0x1864CFB, 0x1800063

_lib = None


class ChestDlCfg(C.Structure):
    """srslte_chest_dl_cfg_t (chest_dl.h:116-130)."""
    _fields_ = [("noise_alg", C.c_int), ("filter_type", C.c_int), ("filter_coef", C.c_float * 2), ("mbsfn_area_id", C.c_uint16),
                ("interpolate_subframe", C.c_uint8), ("rsrp_neighbour", C.c_uint8), ("cfo_estimate_enable", C.c_uint8),
                ("cfo_estimate_sf_mask", C.c_uint32), ("sync_error_enable", C.c_uint8)]


CHEST_RES_FIELDS = ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db", "rssi_dbm", "cfo", "sync_error")


class Cbsegm(C.Structure):
    """srslte_cbsegm_t (cbsegm.h:33-44)."""
    _fields_ = [(n, C.c_uint32) for n in ("F", "C", "K1", "K2", "K1_idx", "K2_idx", "C1", "C2", "tbs")]


class DlRxCfg(C.Structure):
    _fields_ = [("cell_id", C.c_uint32), ("nof_prb", C.c_uint32), ("cfi", C.c_uint32), ("rnti", C.c_uint16), ("mod", C.c_int),
                ("tbs", C.c_uint32), ("max_iterations", C.c_uint32), ("max_batch", C.c_uint32), ("mmse", C.c_int), ("chest_cfg", ChestDlCfg),
                ("llr_8bit", C.c_int), ("nof_rx_antennas", C.c_uint32), ("nof_ports", C.c_uint32), ("csi_enable", C.c_int), ("power_scale", C.c_int), ("p_a", C.c_float),
                ("tx_scheme", C.c_int), ("pmi", C.c_uint32), ("mod2", C.c_int), ("tbs2", C.c_uint32), ("cp_ext", C.c_int),
                ("tdd", C.c_int), ("tdd_sf_config", C.c_uint32), ("tdd_ss_config", C.c_uint32),
                ("mbsfn", C.c_int), ("mbsfn_area_id", C.c_uint32), ("non_mbsfn_region", C.c_uint32)]


class DlGrant(C.Structure):
    """srslte_hip_dl_grant_t (phy_hip.h): the per-subframe part of srslte_pdsch_cfg_t / srslte_pdsch_grant_t."""
    _fields_ = [("prb_mask", (C.c_uint32 * 4) * 2), ("mod", C.c_int), ("tbs", C.c_uint32), ("rv", C.c_uint32), ("cfi", C.c_uint32), ("rnti", C.c_uint16),
                ("new_data", C.c_int)]

    @classmethod
    def make(cls, nof_prb, mod, tbs, rnti, cfi=1, rv=0, new_data=True, prb_mask=None):
        """prb_mask: None = every PRB in both slots, else [2][nof_prb] of 0/1 (srslte_pdsch_grant_t.prb_idx)."""
        g = cls()
        g.mod, g.tbs, g.rv, g.cfi, g.rnti, g.new_data = mod, tbs, rv, cfi, rnti, 1 if new_data else 0
        for s in range(2):
            for n in range(nof_prb):
                if prb_mask is None or prb_mask[s][n]:
                    g.prb_mask[s][n >> 5] |= 1 << (n & 31)
        return g


def lib():
    """The native library; raises if it was not built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.srslte_hip_malloc.restype = vp
        L.srslte_hip_malloc.argtypes = [C.c_size_t]
        L.srslte_hip_free.argtypes = [vp]
        L.srslte_hip_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
        L.srslte_hip_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
        L.srslte_hip_memset.argtypes = [vp, C.c_int, C.c_size_t]
        L.srslte_hip_stream_create.restype = vp
        L.srslte_hip_stream_destroy.argtypes = [vp]
        L.srslte_hip_stream_sync.argtypes = [vp]
        L.srslte_hip_event_create.restype = vp
        L.srslte_hip_event_record.argtypes = [vp, vp]
        L.srslte_hip_event_elapsed_ms.restype = C.c_float
        L.srslte_hip_event_elapsed_ms.argtypes = [vp, vp]
        L.srslte_hip_event_destroy.argtypes = [vp]
        L.srslte_hip_ofdm_create.restype = vp
        L.srslte_hip_ofdm_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.srslte_hip_ofdm_create_sz.restype = vp
        L.srslte_hip_ofdm_create_sz.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.srslte_hip_ofdm_destroy.argtypes = [vp]
        L.srslte_hip_ofdm_set_normalize.argtypes = [vp, C.c_int]
        L.srslte_hip_ofdm_set_freq_shift.argtypes = [vp, C.c_float]
        L.srslte_hip_ofdm_symbol_sz.argtypes = [vp]
        L.srslte_hip_ofdm_sf_len.argtypes = [vp]
        L.srslte_hip_ofdm_rx_sf_batch.argtypes = [vp, vp, vp, C.c_int, vp]
        L.srslte_hip_ofdm_tx_sf_batch.argtypes = [vp, vp, vp, C.c_int, vp]
        L.srslte_hip_dft_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]
        L.srslte_hip_dft_precoding_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp]
        L.srslte_hip_chest_dl_create.restype = vp
        L.srslte_hip_chest_dl_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.srslte_hip_chest_dl_destroy.argtypes = [vp]
        L.srslte_hip_chest_dl_estimate_batch.argtypes = [vp, C.POINTER(ChestDlCfg), C.c_uint32, vp, vp, vp, C.c_int, vp]
        L.srslte_hip_chest_dl_set_mbsfn_area_id.argtypes = [vp, C.c_uint16]
        L.srslte_hip_chest_dl_mbsfn_pilots.restype = vp
        L.srslte_hip_chest_dl_mbsfn_pilots.argtypes = [vp, C.c_uint16]
        L.srslte_hip_chest_dl_estimate_mbsfn_batch.argtypes = [vp, C.POINTER(ChestDlCfg), C.c_uint32, vp, vp, vp, C.c_int, C.c_int, vp]
        for n in ("", "_s", "_b"):
            getattr(L, "srslte_hip_demod_soft_demodulate%s_batch" % n).argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, vp]
        L.srslte_hip_tdec_create.restype = vp
        L.srslte_hip_tdec_create.argtypes = [C.c_uint32, C.c_uint32]
        L.srslte_hip_tdec_destroy.argtypes = [vp]
        L.srslte_hip_tdec_autoimp_get_subblocks.restype = C.c_uint32
        L.srslte_hip_tdec_input_len.restype = C.c_uint32
        L.srslte_hip_tdec_run_batch.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                vp, C.c_uint32, vp, vp, vp]
        L.srslte_hip_tdec_run_batch_manual.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                       C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, vp, vp]
        L.srslte_hip_tdec_run_batch_8bit.argtypes = L.srslte_hip_tdec_run_batch.argtypes
        L.srslte_hip_tdec_autoimp_get_subblocks_8bit.restype = C.c_uint32
        L.srslte_hip_tcod_encode_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
        L.srslte_hip_tcod_encode_bytes_batch.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp]
        L.srslte_hip_cbsegm.argtypes = [C.POINTER(Cbsegm), C.c_uint32]
        L.srslte_hip_tc_interl_LTE_gen_interl.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
        L.srslte_hip_dl_rx_create.restype = vp
        L.srslte_hip_dl_rx_create.argtypes = [C.POINTER(DlRxCfg)]
        L.srslte_hip_dl_rx_destroy.argtypes = [vp]
        L.srslte_hip_dl_rx_nof_re.restype = C.c_uint32
        L.srslte_hip_dl_rx_nof_re.argtypes = [vp, C.c_uint32]
        L.srslte_hip_dl_rx_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, vp]
        L.srslte_hip_dl_rx_grid_batch.argtypes = L.srslte_hip_dl_rx_batch.argtypes
        L.srslte_hip_dl_rx_stage.argtypes = [vp, C.c_int, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, vp]
        L.srslte_hip_dl_rx_batch_grants.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(DlGrant), vp, C.c_uint32, vp, vp]
        L.srslte_hip_dl_rx_debug_buffer.restype = vp
        L.srslte_hip_dl_rx_debug_buffer.argtypes = [vp, C.c_int]
        L.srslte_hip_dl_rx_keep_symbols.argtypes = [vp, C.c_int]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != SRSLTE_SUCCESS:
        raise RuntimeError("%s failed with %d" % (what, rc))


class DevBuf:
    """A device allocation owned through the C ABI."""

    _poison = 0

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().srslte_hip_malloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("srslte_hip_malloc(%d)" % nbytes)
        if os.environ.get("SRSLTE_HIP_TEST_POISON"):  # tests: every allocation starts with its own byte pattern, so that comparing or
            DevBuf._poison = (DevBuf._poison * 37 + 11) & 0xFF  # reading bytes nobody wrote fails every time, not once in a while
            fill = np.full(self.nbytes, DevBuf._poison, np.uint8)
            _check(lib().srslte_hip_memcpy_h2d(self.ptr, fill.ctypes.data, fill.nbytes), "memcpy_h2d")

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(max(arr.nbytes, 1))
        _check(lib().srslte_hip_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes), "memcpy_h2d")
        return b

    def to_host(self, dtype, count=None):
        dtype = np.dtype(dtype)
        n = self.nbytes // dtype.itemsize if count is None else int(count)
        out = np.empty(n, dtype)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib().srslte_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DevView(DevBuf):
    """Device memory owned by somebody else (e.g. a torch tensor), seen through the same interface."""

    def __init__(self, ptr, nbytes):
        self.nbytes, self.ptr = int(nbytes), int(ptr)

    def free(self):
        self.ptr = None


def sync():
    _check(lib().srslte_hip_sync(), "sync")


def symbol_sz(nof_prb):
    """srslte_symbol_sz (phy_common.c:322-345)."""
    for lim, n in ((6, 128), (15, 256), (25, 384), (50, 768), (75, 1024), (110, 1536)):
        if 0 < nof_prb <= lim:
            return n
    return -1


class Ofdm:
    """srslte_ofdm_t: srslte_ofdm_rx_init/tx_init + set_normalize/set_freq_shift + rx_sf/tx_sf (ofdm.h), batched."""

    def __init__(self, nof_prb, cp_norm=True, rx=True, symbol_sz=None):
        """symbol_sz: None = srslte_symbol_sz(nof_prb) of the default rate family; else as srslte_ofdm_init_ takes it (e.g. 2048 for 100 PRB
        after srslte_use_standard_symbol_size(true))."""
        if symbol_sz is None:
            self.h = lib().srslte_hip_ofdm_create(nof_prb, 1 if cp_norm else 0, 1 if rx else 0)
        else:
            self.h = lib().srslte_hip_ofdm_create_sz(nof_prb, symbol_sz, 1 if cp_norm else 0, 1 if rx else 0)
        if not self.h:
            raise RuntimeError("srslte_hip_ofdm_create failed")
        self.nof_prb, self.rx = nof_prb, rx
        self.nsym = 14 if cp_norm else 12
        self.sf_len = lib().srslte_hip_ofdm_sf_len(self.h)
        self.grid_len = self.nsym * 12 * nof_prb

    def set_normalize(self, en):
        _check(lib().srslte_hip_ofdm_set_normalize(self.h, 1 if en else 0), "set_normalize")

    def set_freq_shift(self, f):
        _check(lib().srslte_hip_ofdm_set_freq_shift(self.h, f), "set_freq_shift")

    def rx_sf(self, time_samples):
        x = np.ascontiguousarray(time_samples, np.complex64).reshape(-1, self.sf_len)
        din, dout = DevBuf.from_host(x), DevBuf(x.shape[0] * self.grid_len * 8)
        _check(lib().srslte_hip_ofdm_rx_sf_batch(self.h, din.ptr, dout.ptr, x.shape[0], None), "ofdm_rx_sf_batch")
        sync()
        return dout.to_host(np.complex64).reshape(x.shape[0], self.grid_len)

    def tx_sf(self, grid):
        x = np.ascontiguousarray(grid, np.complex64).reshape(-1, self.grid_len)
        din, dout = DevBuf.from_host(x), DevBuf(x.shape[0] * self.sf_len * 8)
        _check(lib().srslte_hip_ofdm_tx_sf_batch(self.h, din.ptr, dout.ptr, x.shape[0], None), "ofdm_tx_sf_batch")
        sync()
        return dout.to_host(np.complex64).reshape(x.shape[0], self.sf_len)

    def free(self):
        if self.h:
            lib().srslte_hip_ofdm_destroy(self.h)
            self.h = None


def dft(x, forward=True, scale=1.0):
    """srslte_dft_run_c on each row of x (unnormalised unless scale given)."""
    x = np.ascontiguousarray(x, np.complex64)
    x2 = x.reshape(-1, x.shape[-1])
    din, dout = DevBuf.from_host(x2), DevBuf(x2.nbytes)
    n = x2.shape[1]
    rc = lib().srslte_hip_dft_batch(din.ptr, dout.ptr, n, x2.shape[0], n, n, 1 if forward else 0, scale, None)
    _check(rc, "dft_batch")
    sync()
    return dout.to_host(np.complex64).reshape(x.shape)


def dft_precoding(x, nof_prb, nof_symbols, forward=True):
    """srslte_dft_precoding (dft_precoding.c:100-113)."""
    x = np.ascontiguousarray(x, np.complex64)
    din, dout = DevBuf.from_host(x), DevBuf(x.nbytes)
    rc = lib().srslte_hip_dft_precoding_batch(din.ptr, dout.ptr, nof_prb, nof_symbols, 1 if forward else 0, None)
    if rc != SRSLTE_SUCCESS:
        return rc, None
    sync()
    return rc, dout.to_host(np.complex64).reshape(x.shape)


class ChestDl:
    """srslte_chest_dl_t: init + set_cell + estimate_cfg (chest_dl.h:132-156), batched over subframes tti0, tti0+1, ..."""

    def __init__(self, cell_id, nof_prb, nof_ports=1, cp_norm=True):
        self.h = lib().srslte_hip_chest_dl_create(cell_id, nof_prb, nof_ports, 1 if cp_norm else 0)
        if not self.h:
            raise RuntimeError("srslte_hip_chest_dl_create failed")
        self.grid_len = (14 if cp_norm else 12) * 12 * nof_prb
        self.nof_ports = nof_ports

    def set_tdd(self, sf_config, ss_config):
        """TDD cell: srslte_tdd_config_t of the subframes (sf_config < 0: FDD again)."""
        L = lib()
        L.srslte_hip_chest_dl_set_tdd.argtypes = [C.c_void_p, C.c_int, C.c_int]
        return L.srslte_hip_chest_dl_set_tdd(self.h, sf_config, ss_config)

    def set_mbsfn_area_id(self, area_id):
        """srslte_chest_dl_set_mbsfn_area_id (chest_dl.c:244-262)."""
        return lib().srslte_hip_chest_dl_set_mbsfn_area_id(self.h, area_id)

    def estimate_mbsfn(self, grid, tti0, cfg, nof_rx=1, want_ce=True):
        """MBSFN subframes (cfg.mbsfn_area_id): grid [nof_sf][nof_rx][14*12*prb] -> (rc, ce [nof_sf][nof_ports][nof_rx][...], noise
        [nof_sf][nof_ports][nof_rx]); symbols 12, 13 of ce are not written (returned as zeros)."""
        g = np.ascontiguousarray(grid, np.complex64).reshape(-1, nof_rx, self.grid_len)
        n = g.shape[0]
        dg, dce, dn = DevBuf.from_host(g), DevBuf(g.nbytes * self.nof_ports), DevBuf(4 * n * nof_rx * self.nof_ports)
        _check(lib().srslte_hip_memset(dce.ptr, 0, g.nbytes * self.nof_ports), "memset")
        rc = lib().srslte_hip_chest_dl_estimate_mbsfn_batch(self.h, C.byref(cfg), tti0, dg.ptr, dce.ptr if want_ce else None, dn.ptr, n, nof_rx, None)
        if rc != SRSLTE_SUCCESS:
            return rc, None, None
        sync()
        return rc, dce.to_host(np.complex64).reshape(n, self.nof_ports, nof_rx, self.grid_len), dn.to_host(np.float32).reshape(n, self.nof_ports, nof_rx)

    def estimate_multi(self, grid, tti0, cfg, nof_rx=1, ce_in=None):
        """grid [nof_sf][nof_rx][14*12*prb] -> (rc, ce [nof_sf][nof_ports][nof_rx][...], res dict, raw [nof_sf][nof_ports][nof_rx][6]).
        ce_in: what the estimate buffer holds before the call (4-port cells with interpolate_subframe keep symbol 0 of ports 2/3)."""
        L = lib()
        L.srslte_hip_chest_dl_estimate_batch_multi.argtypes = [C.c_void_p, C.POINTER(ChestDlCfg), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                               C.c_int, C.c_int, C.c_void_p]
        L.srslte_hip_chest_dl_last_raw.restype = C.c_void_p
        L.srslte_hip_chest_dl_last_raw.argtypes = [C.c_void_p]
        g = np.ascontiguousarray(grid, np.complex64).reshape(-1, nof_rx, self.grid_len)
        n = g.shape[0]
        dg, dres = DevBuf.from_host(g), DevBuf(n * 40)
        dce = DevBuf(g.nbytes * self.nof_ports) if ce_in is None else DevBuf.from_host(np.ascontiguousarray(ce_in, np.complex64))
        rc = L.srslte_hip_chest_dl_estimate_batch_multi(self.h, C.byref(cfg), tti0, dg.ptr, dce.ptr, dres.ptr, n, nof_rx, None)
        if rc != SRSLTE_SUCCESS:
            return rc, None, None, None
        sync()
        res = dres.to_host(np.float32).reshape(n, 10)
        raw = np.empty(n * self.nof_ports * nof_rx * 6, np.float32)
        _check(L.srslte_hip_memcpy_d2h(raw.ctypes.data, L.srslte_hip_chest_dl_last_raw(self.h), raw.nbytes), "memcpy_d2h")
        return (rc, dce.to_host(np.complex64).reshape(n, self.nof_ports, nof_rx, self.grid_len), {k: res[:, i] for i, k in enumerate(CHEST_RES_FIELDS)},
                raw.reshape(n, self.nof_ports, nof_rx, 6))

    def estimate(self, grid, tti0=0, cfg=None, want_ce=True):
        cfg = cfg or ChestDlCfg()
        g = np.ascontiguousarray(grid, np.complex64).reshape(-1, self.grid_len)
        n = g.shape[0]
        dg, dce, dres = DevBuf.from_host(g), DevBuf(g.nbytes), DevBuf(n * 40)
        rc = lib().srslte_hip_chest_dl_estimate_batch(self.h, C.byref(cfg), tti0, dg.ptr, dce.ptr if want_ce else None, dres.ptr, n, None)
        _check(rc, "chest_dl_estimate_batch")
        sync()
        res = dres.to_host(np.float32).reshape(n, 10)
        return (dce.to_host(np.complex64).reshape(n, self.grid_len) if want_ce else None), {k: res[:, i] for i, k in enumerate(CHEST_RES_FIELDS)}

    def free(self):
        if self.h:
            lib().srslte_hip_chest_dl_destroy(self.h)
            self.h = None


_LLR_DT = {"f": np.float32, "s": np.int16, "b": np.int8}


def demod_soft_demodulate(mod, symbols, kind="s", ncalls=1):
    """srslte_demod_soft_demodulate / _s / _b (demod_soft.h:39-53); kind in 'f','s','b'. Returns (rc, llr)."""
    s = np.ascontiguousarray(symbols, np.complex64).reshape(ncalls, -1)
    nsym = s.shape[1]
    qm = 1 if mod == MOD_BPSK else 2 * mod
    dt = np.dtype(_LLR_DT[kind])
    ds, dl = DevBuf.from_host(s), DevBuf(max(1, s.size * qm * dt.itemsize))
    fn = getattr(lib(), "srslte_hip_demod_soft_demodulate%s_batch" % ("" if kind == "f" else "_" + kind))
    rc = fn(mod, ds.ptr, dl.ptr, nsym, ncalls, None)
    if rc != SRSLTE_SUCCESS:
        return rc, None
    sync()
    return rc, dl.to_host(dt, s.size * qm).reshape(ncalls, nsym * qm)


class Tdec:
    """srslte_tdec_t: srslte_tdec_init + srslte_tdec_run_all / iteration-with-CRC (turbodecoder.h:63-135), batched."""

    def __init__(self, max_long_cb=6144, max_nof_cb=64):
        self.h = lib().srslte_hip_tdec_create(max_long_cb, max_nof_cb)
        if not self.h:
            raise RuntimeError("srslte_hip_tdec_create failed")

    def run_all(self, llr, long_cb, nof_iterations, sb_layout=False, crc_poly=0, crc_nbits=0, force_subblocks=None, llr8=False):
        """llr8: int8 LLRs through srslte_hip_tdec_run_batch_8bit (srslte_tdec_run_all_8bit, turbodecoder.c:573-588)."""
        x = np.ascontiguousarray(llr, np.int8 if llr8 else np.int16)
        x = x.reshape(-1, x.shape[-1])
        ncb = x.shape[0]
        din, dout = DevBuf.from_host(x), DevBuf(ncb * (long_cb // 8))
        dit, dok = DevBuf(4 * ncb), DevBuf(ncb)
        if llr8:
            rc = lib().srslte_hip_tdec_run_batch_8bit(self.h, din.ptr, x.shape[1], 1 if sb_layout else 0, long_cb, ncb, nof_iterations, crc_poly,
                                                      crc_nbits, dout.ptr, long_cb // 8, dit.ptr, dok.ptr, None)
        elif force_subblocks is None:
            rc = lib().srslte_hip_tdec_run_batch(self.h, din.ptr, x.shape[1], 1 if sb_layout else 0, long_cb, ncb, nof_iterations, crc_poly,
                                                 crc_nbits, dout.ptr, long_cb // 8, dit.ptr, dok.ptr, None)
        else:
            rc = lib().srslte_hip_tdec_run_batch_manual(self.h, din.ptr, x.shape[1], 1 if sb_layout else 0, long_cb, force_subblocks, ncb,
                                                        nof_iterations, crc_poly, crc_nbits, dout.ptr, long_cb // 8, dit.ptr, dok.ptr, None)
        if rc != SRSLTE_SUCCESS:
            return rc, None, None, None
        sync()
        return rc, dout.to_host(np.uint8).reshape(ncb, long_cb // 8), dit.to_host(np.uint32), dok.to_host(np.uint8)

    def free(self):
        if self.h:
            lib().srslte_hip_tdec_destroy(self.h)
            self.h = None


def tcod_encode(bits, long_cb):
    """srslte_tcod_encode (turbocoder.c:76-186): [ncb][K] bits -> [ncb][3K+12]. Returns (rc, out)."""
    x = np.ascontiguousarray(bits, np.uint8).reshape(-1, long_cb)
    din, dout = DevBuf.from_host(x), DevBuf(x.shape[0] * (3 * long_cb + 12))
    rc = lib().srslte_hip_tcod_encode_batch(din.ptr, dout.ptr, long_cb, x.shape[0], None)
    if rc != SRSLTE_SUCCESS:
        return rc, None
    sync()
    return rc, dout.to_host(np.uint8).reshape(x.shape[0], 3 * long_cb + 12)


def cbsegm(tbs):
    s = Cbsegm()
    rc = lib().srslte_hip_cbsegm(C.byref(s), tbs)
    return rc, s


def tc_interl(long_cb, win=1):
    f, r = np.zeros(long_cb, np.uint16), np.zeros(long_cb, np.uint16)
    rc = lib().srslte_hip_tc_interl_LTE_gen_interl(f.ctypes.data, r.ctypes.data, long_cb, win)
    return rc, f, r


class DmrsPuschCfg(C.Structure):
    _fields_ = [("cyclic_shift", C.c_uint32), ("delta_ss", C.c_uint32), ("group_hopping_en", C.c_int), ("sequence_hopping_en", C.c_int)]


class ChestUlRes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("noise_estimate", "noise_estimate_dbm", "snr", "snr_db", "cfo")]


class ChestUl:
    """srslte_chest_ul_t: init + set_cell + pregen + estimate_pusch (chest_ul.h:78-104), batched over subframes tti0, tti0+1, ..."""

    def __init__(self, cell_id, nof_prb, cyclic_shift=0, delta_ss=0, group_hopping=False, sequence_hopping=False, cp_ext=False):
        self.cfg = DmrsPuschCfg(cyclic_shift, delta_ss, 1 if group_hopping else 0, 1 if sequence_hopping else 0)
        lib().srslte_hip_chest_ul_create.restype = C.c_void_p
        lib().srslte_hip_chest_ul_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(DmrsPuschCfg)]
        self.h = lib().srslte_hip_chest_ul_create(cell_id, nof_prb, 0 if cp_ext else 1, C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("srslte_hip_chest_ul_create failed")
        self.nof_prb, self.nof_symb = nof_prb, 12 if cp_ext else 14

    def dmrs(self, L_prb, sf_idx, n_dmrs):
        r = np.zeros(2 * 12 * L_prb, np.complex64)
        lib().srslte_hip_refsignal_dmrs_pusch_gen.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        rc = lib().srslte_hip_refsignal_dmrs_pusch_gen(self.h, L_prb, sf_idx, n_dmrs, r.ctypes.data)
        return rc, r

    def estimate_pusch(self, grid, tti0, L_prb, n_prb, n_dmrs, ce_init=None):
        x = np.ascontiguousarray(grid, np.complex64).reshape(-1, self.nof_symb * 12 * self.nof_prb)
        n = x.shape[0]
        dg = DevBuf.from_host(x)
        dce = DevBuf.from_host(np.zeros_like(x) if ce_init is None else np.ascontiguousarray(ce_init, np.complex64))
        dres = DevBuf(n * C.sizeof(ChestUlRes))
        f = lib().srslte_hip_chest_ul_estimate_pusch_batch
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        rc = f(self.h, tti0, L_prb, n_prb, n_dmrs, dg.ptr, dce.ptr, dres.ptr, n, None)
        if rc != SRSLTE_SUCCESS:
            return rc, None, None
        sync()
        res = np.frombuffer(dres.to_host(np.uint8).tobytes(), dtype=np.float32).reshape(n, 5)
        return rc, dce.to_host(np.complex64).reshape(x.shape), res

    def free(self):
        if self.h:
            lib().srslte_hip_chest_ul_destroy.argtypes = [C.c_void_p]
            lib().srslte_hip_chest_ul_destroy(self.h)
            self.h = None


class DlGrant2(C.Structure):
    """srslte_hip_dl_grant2_t: a grant with its transmission scheme, pmi and second transport block."""
    _fields_ = [("tb0", DlGrant), ("tx_scheme", C.c_int), ("pmi", C.c_uint32), ("mod2", C.c_int), ("tbs2", C.c_uint32), ("rv2", C.c_uint32), ("new_data2", C.c_int)]


class DlRx:
    """Batched PDSCH receive chain (ue_dl.c:369-384 + pdsch.c:833-997 + sch.c:507-532 for one codeword)."""

    def __init__(self, cell_id, nof_prb, cfi, rnti, mod, tbs, max_iterations, max_batch, mmse=True, chest_cfg=None, llr_8bit=False, nof_rx=1,
                 nof_ports=1, csi=False, power_scale=False, p_a=0.0, out_ptrs=None, tx_scheme=0, pmi=0, mod2=0, tbs2=0, cp_ext=False, tdd=None,
                 mbsfn=None):
        """mbsfn = (area id, non-MBSFN region length): a PMCH pipeline (MBSFN subframes; rnti unused).
        tx_scheme 3 (large-delay CDD) / 2 (closed-loop multiplexing) with pmi, and mod2 / tbs2 for a second transport block: the two-layer
        modes; decode() then returns lists [transport block 0, transport block 1] of tb and ok arrays."""
        self.cfg = DlRxCfg(cell_id, nof_prb, cfi, rnti, mod, tbs, max_iterations, max_batch, 1 if mmse else 0, chest_cfg or ChestDlCfg(),
                           1 if llr_8bit else 0, nof_rx, nof_ports, 1 if csi else 0, 1 if power_scale else 0, p_a, tx_scheme, pmi, mod2, tbs2, 1 if cp_ext else 0,
                           1 if tdd else 0, tdd[0] if tdd else 0, tdd[1] if tdd else 0, 1 if mbsfn else 0, mbsfn[0] if mbsfn else 0, mbsfn[1] if mbsfn else 0)
        self.nof_rx = nof_rx
        self.h = lib().srslte_hip_dl_rx_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("srslte_hip_dl_rx_create failed")
        self.tbs, self.max_batch, self.tbs2 = tbs, max_batch, tbs2
        self.tb_stride = (max(tbs, tbs2) // 8 + 6 + 15) & ~15
        self.sf_len = 15 * symbol_sz(nof_prb)
        if out_ptrs is None:
            self.d_tb, self.d_ok = DevBuf(self.tb_stride * max_batch * (2 if tbs2 else 1)), DevBuf(max_batch * (2 if tbs2 else 1))
        else:  # caller-owned device memory (e.g. a torch tensor that a collective reads): (tb pointer, ok pointer)
            self.d_tb, self.d_ok = DevView(out_ptrs[0], self.tb_stride * max_batch), DevView(out_ptrs[1], max_batch)
        # per-subframe stride of the LLR buffer e (debug buffer 4)
        self.e_stride = (max(self.nof_re(s) for s in (0, 1, 5)) * {1: 2, 2: 4, 3: 6, 4: 8}[mod] + 15) & ~15

    def nof_re(self, sf_idx):
        return lib().srslte_hip_dl_rx_nof_re(self.h, sf_idx)

    def keep_symbols(self, enable=True):
        """Also write the equalised symbols d (debug buffer 3); off by default: the fused kernel never stores them."""
        _check(lib().srslte_hip_dl_rx_keep_symbols(self.h, 1 if enable else 0), "dl_rx_keep_symbols")

    def run_device(self, d_iq_ptr, tti0, nof_sf, stream=None):
        return lib().srslte_hip_dl_rx_batch(self.h, d_iq_ptr, tti0, nof_sf, self.d_tb.ptr, self.tb_stride, self.d_ok.ptr, stream)

    def stage(self, stage, d_iq_ptr, tti0, nof_sf, stream=None):
        return lib().srslte_hip_dl_rx_stage(self.h, stage, d_iq_ptr, tti0, nof_sf, self.d_tb.ptr, self.tb_stride, self.d_ok.ptr, stream)

    def decode(self, iq, tti0=0):
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.nof_rx * self.sf_len)  # [nsf][nof_rx][sf_len]
        din = DevBuf.from_host(x)
        _check(self.run_device(din.ptr, tti0, x.shape[0]), "dl_rx_batch")
        sync()
        if self.tbs2:  # rows b and nof_sf + b: the two transport blocks of subframe b
            n = x.shape[0]
            tb, ok = self.d_tb.to_host(np.uint8).reshape(-1, self.tb_stride), self.d_ok.to_host(np.uint8)
            return [tb[:n, :self.tbs // 8 + 3], tb[n:2 * n, :self.tbs2 // 8 + 3]], [ok[:n], ok[n:2 * n]]
        tb = self.d_tb.to_host(np.uint8).reshape(self.max_batch, self.tb_stride)[:x.shape[0], :self.tbs // 8 + 3]
        return tb, self.d_ok.to_host(np.uint8)[:x.shape[0]]

    def decode_harq2(self, iq, tti0, rv, new_data):
        """srslte_hip_dl_rx_batch_harq2: rv / new_data per transport block (two-layer modes)."""
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.nof_rx * self.sf_len)
        din = DevBuf.from_host(x)
        n = x.shape[0]
        lib().srslte_hip_dl_rx_batch_harq2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                       C.c_void_p, C.c_void_p]
        _check(lib().srslte_hip_dl_rx_batch_harq2(self.h, din.ptr, tti0, n, (C.c_uint32 * 2)(*rv), (C.c_int * 2)(*[1 if v else 0 for v in new_data]),
                                                  self.d_tb.ptr, self.tb_stride, self.d_ok.ptr, None), "dl_rx_batch_harq2")
        sync()
        tb, ok = self.d_tb.to_host(np.uint8).reshape(-1, self.tb_stride), self.d_ok.to_host(np.uint8)
        return [tb[:n, :self.tbs // 8 + 3], tb[n:2 * n, :self.tbs2 // 8 + 3]], [ok[:n], ok[n:2 * n]]

    def decode_grants(self, iq, tti0, grants):
        """srslte_hip_dl_rx_batch_grants: subframe b with grants[b] (DlGrant). Returns (rc, tb [nsf][tbs_max/8+3], ok [nsf])."""
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.nof_rx * self.sf_len)
        assert len(grants) == x.shape[0]
        arr = (DlGrant * len(grants))(*grants)
        din = DevBuf.from_host(x)
        rc = lib().srslte_hip_dl_rx_batch_grants(self.h, din.ptr, tti0, x.shape[0], arr, self.d_tb.ptr, self.tb_stride, self.d_ok.ptr, None)
        if rc != SRSLTE_SUCCESS:
            return rc, None, None
        sync()
        tb = self.d_tb.to_host(np.uint8).reshape(self.max_batch, self.tb_stride)[:x.shape[0], :self.tbs // 8 + 3]
        return rc, tb, self.d_ok.to_host(np.uint8)[:x.shape[0]]

    def decode_grants2(self, iq, tti0, grants, from_grid=False):
        """srslte_hip_dl_rx_batch_grants2: subframe b with grants[b] (DlGrant2: scheme, pmi and a second transport block per subframe).
        Returns (rc, [tb0 rows, tb1 rows], [ok0, ok1]); on a cell without two-layer grants the second entries are None.
        from_grid: iq holds frequency-domain grids [nsf][nof_rx][14 * 12 * nof_prb] (srslte_hip_dl_rx_grid_batch_grants2)."""
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.nof_rx * (14 * 12 * self.cfg.nof_prb if from_grid else self.sf_len))
        n = x.shape[0]
        assert len(grants) == n
        two = self.cfg.nof_ports == 2 and self.cfg.nof_rx_antennas == 2
        arr = (DlGrant2 * n)(*grants)
        din, dtb, dok = DevBuf.from_host(x), DevBuf(self.tb_stride * n * 2), DevBuf(2 * n)
        L = lib()
        fn = L.srslte_hip_dl_rx_grid_batch_grants2 if from_grid else L.srslte_hip_dl_rx_batch_grants2
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        rc = fn(self.h, din.ptr, tti0, n, arr, dtb.ptr, self.tb_stride, dok.ptr, None)
        if rc != SRSLTE_SUCCESS:
            return rc, None, None
        sync()
        tb, ok = dtb.to_host(np.uint8).reshape(2 * n, self.tb_stride), dok.to_host(np.uint8)
        return rc, [tb[:n], tb[n:] if two else None], [ok[:n], ok[n:2 * n] if two else None]

    def decode_harq(self, iq, tti0, rv, new_data):
        """srslte_hip_dl_rx_batch_harq: slot b keeps its soft buffers / CRC flags / bytes between calls."""
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.nof_rx * self.sf_len)
        din = DevBuf.from_host(x)
        lib().srslte_hip_dl_rx_batch_harq.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32,
                                                      C.c_void_p, C.c_void_p]
        _check(lib().srslte_hip_dl_rx_batch_harq(self.h, din.ptr, tti0, x.shape[0], rv, 1 if new_data else 0, self.d_tb.ptr, self.tb_stride,
                                                 self.d_ok.ptr, None), "dl_rx_batch_harq")
        sync()
        tb = self.d_tb.to_host(np.uint8).reshape(self.max_batch, self.tb_stride)[:x.shape[0], :self.tbs // 8 + 3]
        return tb, self.d_ok.to_host(np.uint8)[:x.shape[0]]

    def decode_grid(self, grid, tti0=0):
        """Frequency-domain input [nsf][14*12*nof_prb] (srslte_hip_dl_rx_grid_batch)."""
        x = np.ascontiguousarray(grid, np.complex64)
        x = x.reshape(-1, x.shape[-1])
        din = DevBuf.from_host(x)
        _check(lib().srslte_hip_dl_rx_grid_batch(self.h, din.ptr, tti0, x.shape[0], self.d_tb.ptr, self.tb_stride, self.d_ok.ptr, None), "dl_rx_grid_batch")
        sync()
        tb = self.d_tb.to_host(np.uint8).reshape(self.max_batch, self.tb_stride)[:x.shape[0], :self.tbs // 8 + 3]
        return tb, self.d_ok.to_host(np.uint8)[:x.shape[0]]

    def debug(self, which, dtype, count):
        ptr = lib().srslte_hip_dl_rx_debug_buffer(self.h, which)
        out = np.empty(count, dtype)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.h:
            lib().srslte_hip_dl_rx_destroy(self.h)
            self.h = None


class UlRxCfg(C.Structure):
    _fields_ = [("cell_id", C.c_uint32), ("nof_prb", C.c_uint32), ("rnti", C.c_uint16), ("mod", C.c_int), ("tbs", C.c_uint32), ("L_prb", C.c_uint32),
                ("n_prb", C.c_uint32), ("n_dmrs", C.c_uint32), ("max_iterations", C.c_uint32), ("max_batch", C.c_uint32), ("mmse", C.c_int),
                ("dmrs_cfg", DmrsPuschCfg), ("shortened", C.c_int), ("ack_len", C.c_uint32), ("I_offset_ack", C.c_uint32),
                ("ri_len", C.c_uint32), ("I_offset_ri", C.c_uint32), ("cqi_len", C.c_uint32), ("I_offset_cqi", C.c_uint32),
                ("hopping", C.c_uint32), ("n_prb_slot1", C.c_uint32), ("max_grants", C.c_uint32), ("cp_ext", C.c_int)]


class UlGrant(C.Structure):
    """srslte_hip_ul_grant_t: one PUSCH of a srslte_hip_ul_rx_batch_grants call."""
    _fields_ = [("sf", C.c_uint32), ("rnti", C.c_uint16), ("L_prb", C.c_uint32), ("n_prb", C.c_uint32), ("n_prb_slot1", C.c_uint32), ("n_dmrs", C.c_uint32),
                ("mod", C.c_int), ("tbs", C.c_uint32), ("rv", C.c_uint32), ("new_data", C.c_int), ("ack_len", C.c_uint32), ("I_offset_ack", C.c_uint32),
                ("ri_len", C.c_uint32), ("I_offset_ri", C.c_uint32), ("cqi_len", C.c_uint32), ("I_offset_cqi", C.c_uint32)]

    @classmethod
    def make(cls, sf, rnti, L_prb, n_prb, mod, tbs, n_dmrs=0, n_prb_slot1=None, rv=0, new_data=True, ack_len=0, I_offset_ack=0, ri_len=0, I_offset_ri=0,
             cqi_len=0, I_offset_cqi=0):
        return cls(sf, rnti, L_prb, n_prb, n_prb if n_prb_slot1 is None else n_prb_slot1, n_dmrs, mod, tbs, rv, 1 if new_data else 0, ack_len, I_offset_ack,
                   ri_len, I_offset_ri, cqi_len, I_offset_cqi)


class UlRx:
    """Batched PUSCH receive chain (enb_ul.c + pusch.c:423-520 + the UL-SCH part of sch.c:991-1066)."""

    def __init__(self, cell_id, nof_prb, rnti, mod, tbs, L_prb, n_prb, n_dmrs, max_iterations, max_batch, cyclic_shift=0, delta_ss=0,
                 group_hopping=False, sequence_hopping=False, mmse=True, shortened=False, ack_len=0, I_offset_ack=0, ri_len=0, I_offset_ri=0,
                 cqi_len=0, I_offset_cqi=0, n_prb_slot1=None, max_grants=0, cp_ext=False):
        self.cfg = UlRxCfg(cell_id, nof_prb, rnti, mod, tbs, L_prb, n_prb, n_dmrs, max_iterations, max_batch, 1 if mmse else 0,
                           DmrsPuschCfg(cyclic_shift, delta_ss, 1 if group_hopping else 0, 1 if sequence_hopping else 0), 1 if shortened else 0,
                           ack_len, I_offset_ack, ri_len, I_offset_ri, cqi_len, I_offset_cqi, 0 if n_prb_slot1 is None else 1, n_prb_slot1 or 0,
                           max_grants, 1 if cp_ext else 0)
        L = lib()
        L.srslte_hip_ul_rx_ri.restype = C.c_void_p
        L.srslte_hip_ul_rx_ri.argtypes = [C.c_void_p]
        L.srslte_hip_ul_rx_cqi.restype = C.c_void_p
        L.srslte_hip_ul_rx_cqi.argtypes = [C.c_void_p]
        L.srslte_hip_ul_rx_create.restype = C.c_void_p
        L.srslte_hip_ul_rx_create.argtypes = [C.POINTER(UlRxCfg)]
        L.srslte_hip_ul_rx_destroy.argtypes = [C.c_void_p]
        L.srslte_hip_ul_rx_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.srslte_hip_ul_rx_batch_harq.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p,
                                                  C.c_void_p]
        L.srslte_hip_ul_rx_debug_buffer.restype = C.c_void_p
        L.srslte_hip_ul_rx_debug_buffer.argtypes = [C.c_void_p, C.c_int]
        L.srslte_hip_ul_rx_ack.restype = C.c_void_p
        L.srslte_hip_ul_rx_ack.argtypes = [C.c_void_p]
        self.h = L.srslte_hip_ul_rx_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("srslte_hip_ul_rx_create failed")
        self.tbs, self.max_batch = tbs, max_batch
        self.tb_stride = (tbs // 8 + 6 + 15) & ~15
        self.sf_len = 15 * symbol_sz(nof_prb)
        self.rows = max(max_batch, max_grants)
        self.rows_grants = max_grants or max_batch
        self.d_tb, self.d_ok = DevBuf(self.tb_stride * self.rows), DevBuf(self.rows)

    def decode_grants(self, iq, tti0, grants):
        """srslte_hip_ul_rx_batch_grants: grants = list of UlGrant; row p of the result belongs to grants[p]."""
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.sf_len)
        din = DevBuf.from_host(x)
        arr = (UlGrant * len(grants))(*grants)
        L = lib()
        L.srslte_hip_ul_rx_batch_grants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                                    C.c_void_p]
        _check(L.srslte_hip_ul_rx_batch_grants(self.h, din.ptr, tti0, x.shape[0], arr, len(grants), self.d_tb.ptr, self.tb_stride, self.d_ok.ptr, None),
               "ul_rx_batch_grants")
        sync()
        tb = self.d_tb.to_host(np.uint8).reshape(self.rows, self.tb_stride)[:len(grants)]
        self.last_nof_grants = len(grants)
        return tb, self.d_ok.to_host(np.uint8)[:len(grants)]

    def grants_uci(self):
        """(HARQ-ACK decisions [nof_grants][2], rank indications [nof_grants][2]) of the last decode_grants()."""
        L = lib()
        out = []
        for fn in (L.srslte_hip_ul_rx_grants_ack, L.srslte_hip_ul_rx_grants_ri):
            fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p]
            a = np.empty(2 * self.last_nof_grants, np.uint8)
            _check(L.srslte_hip_memcpy_d2h(a.ctypes.data, fn(self.h), a.nbytes), "memcpy_d2h")
            out.append(a.reshape(-1, 2))
        return out

    def grants_cqi(self):
        """(report bits [nof_grants][64], CRC flags [nof_grants]) of the last decode_grants()."""
        L = lib()
        L.srslte_hip_ul_rx_grants_cqi.restype, L.srslte_hip_ul_rx_grants_cqi.argtypes = C.c_void_p, [C.c_void_p]
        out = np.empty(65 * self.rows_grants, np.uint8)
        _check(L.srslte_hip_memcpy_d2h(out.ctypes.data, L.srslte_hip_ul_rx_grants_cqi(self.h), out.nbytes), "memcpy_d2h")
        n = self.last_nof_grants
        return out[:64 * self.rows_grants].reshape(-1, 64)[:n], out[64 * self.rows_grants:][:n]

    def decode(self, iq, tti0=0):
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.sf_len)
        din = DevBuf.from_host(x)
        _check(lib().srslte_hip_ul_rx_batch(self.h, din.ptr, tti0, x.shape[0], self.d_tb.ptr, self.tb_stride, self.d_ok.ptr, None), "ul_rx_batch")
        sync()
        tb = self.d_tb.to_host(np.uint8).reshape(self.rows, self.tb_stride)[:x.shape[0], :self.tbs // 8 + 3]
        self.last_nof_sf = x.shape[0]
        return tb, self.d_ok.to_host(np.uint8)[:x.shape[0]]

    def decode_harq(self, iq, tti0, rv, new_data):
        """srslte_hip_ul_rx_batch_harq: slot b keeps its soft buffers between calls; new_data starts new transport blocks."""
        x = np.ascontiguousarray(iq, np.complex64).reshape(-1, self.sf_len)
        din = DevBuf.from_host(x)
        _check(lib().srslte_hip_ul_rx_batch_harq(self.h, din.ptr, tti0, x.shape[0], rv, 1 if new_data else 0, self.d_tb.ptr, self.tb_stride,
                                                 self.d_ok.ptr, None), "ul_rx_batch_harq")
        sync()
        tb = self.d_tb.to_host(np.uint8).reshape(self.rows, self.tb_stride)[:x.shape[0], :self.tbs // 8 + 3]
        self.last_nof_sf = x.shape[0]
        return tb, self.d_ok.to_host(np.uint8)[:x.shape[0]]

    def ack(self):
        """HARQ-ACK decisions [nof_sf][2] of the last decode() (srslte_uci_value_t.ack.ack_value of srslte_pusch_decode)."""
        out = np.empty(2 * self.max_batch, np.uint8)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, lib().srslte_hip_ul_rx_ack(self.h), out.nbytes), "memcpy_d2h")
        return out.reshape(-1, 2)[:self.last_nof_sf, :max(self.cfg.ack_len, 1)]

    def ri(self):
        """Rank-indication decisions [nof_sf][ri_len] of the last decode() (srslte_uci_value_t.ri)."""
        out = np.empty(2 * self.max_batch, np.uint8)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, lib().srslte_hip_ul_rx_ri(self.h), out.nbytes), "memcpy_d2h")
        return out.reshape(-1, 2)[:self.last_nof_sf, :max(self.cfg.ri_len, 1)]

    def cqi(self):
        """CQI reports of the last decode(): bits [nof_sf][cqi_len] and the CRC flags [nof_sf] (srslte_uci_value_t.cqi, .cqi.data_crc)."""
        out = np.empty(65 * self.max_batch, np.uint8)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, lib().srslte_hip_ul_rx_cqi(self.h), out.nbytes), "memcpy_d2h")
        n = self.last_nof_sf
        return out[:64 * self.max_batch].reshape(-1, 64)[:n, :self.cfg.cqi_len], out[64 * self.max_batch:][:n]

    def debug(self, which, dtype, count):
        ptr = lib().srslte_hip_ul_rx_debug_buffer(self.h, which)
        out = np.empty(count, dtype)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.h:
            lib().srslte_hip_ul_rx_destroy(self.h)
            self.h = None


class UlTxCfg(C.Structure):
    _fields_ = [("cell_id", C.c_uint32), ("nof_prb", C.c_uint32), ("rnti", C.c_uint16), ("mod", C.c_int), ("tbs", C.c_uint32), ("L_prb", C.c_uint32),
                ("n_prb", C.c_uint32), ("n_dmrs", C.c_uint32), ("max_batch", C.c_uint32), ("dmrs_cfg", DmrsPuschCfg), ("shortened", C.c_int),
                ("ack_len", C.c_uint32), ("I_offset_ack", C.c_uint32), ("ri_len", C.c_uint32), ("I_offset_ri", C.c_uint32),
                ("cqi_len", C.c_uint32), ("I_offset_cqi", C.c_uint32), ("hopping", C.c_uint32), ("n_prb_slot1", C.c_uint32), ("max_grants", C.c_uint32),
                ("cp_ext", C.c_int)]


class UlTx:
    """Batched PUSCH transmit chain (srslte_ue_ul_encode ue_ul.c:300-340: srslte_pusch_encode pusch.c:314-421 with the UL-SCH part of
    srslte_ulsch_encode sch.c:1068-1160, DMRS, srslte_ofdm_tx_sf with ue_ul.c:59-64 settings)."""

    def __init__(self, cell_id, nof_prb, rnti, mod, tbs, L_prb, n_prb, n_dmrs, max_batch, cyclic_shift=0, delta_ss=0, group_hopping=False,
                 sequence_hopping=False, shortened=False, ack_len=0, I_offset_ack=0, ri_len=0, I_offset_ri=0, cqi_len=0, I_offset_cqi=0,
                 n_prb_slot1=None, max_grants=0, cp_ext=False):
        self.cfg = UlTxCfg(cell_id, nof_prb, rnti, mod, tbs, L_prb, n_prb, n_dmrs, max_batch,
                           DmrsPuschCfg(cyclic_shift, delta_ss, 1 if group_hopping else 0, 1 if sequence_hopping else 0), 1 if shortened else 0,
                           ack_len, I_offset_ack, ri_len, I_offset_ri, cqi_len, I_offset_cqi, 0 if n_prb_slot1 is None else 1, n_prb_slot1 or 0,
                           max_grants, 1 if cp_ext else 0)
        L = lib()
        L.srslte_hip_ul_tx_batch_uci_cqi.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                     C.c_void_p, C.c_void_p]
        L.srslte_hip_ul_tx_batch_uci.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.srslte_hip_ul_tx_create.restype = C.c_void_p
        L.srslte_hip_ul_tx_create.argtypes = [C.POINTER(UlTxCfg)]
        L.srslte_hip_ul_tx_destroy.argtypes = [C.c_void_p]
        L.srslte_hip_ul_tx_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.srslte_hip_ul_tx_batch_ack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.srslte_hip_ul_tx_debug_buffer.restype = C.c_void_p
        L.srslte_hip_ul_tx_debug_buffer.argtypes = [C.c_void_p, C.c_int]
        self.h = L.srslte_hip_ul_tx_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("srslte_hip_ul_tx_create failed")
        self.tbs, self.max_batch = tbs, max_batch
        self.sf_len = 15 * symbol_sz(nof_prb)
        self.d_iq = DevBuf(8 * self.sf_len * max_batch)

    def encode_grants(self, tbs_bytes, tti0, nof_sf, grants, ack=None, ri=None, cqi=None):
        """srslte_hip_ul_tx_batch_grants: grants = list of UlGrant, tbs_bytes[p] the payload of grants[p]; ack / ri: [nof_grants][<= 2] values,
        cqi: [nof_grants][<= 64] report bits (rows of grants without that UCI are ignored) -> iq [nof_sf][sf_len]."""
        n = len(grants)
        stride = (self.tbs // 8 + 15) & ~15
        x = np.zeros((max(n, 1), stride), np.uint8)
        for p_, b in enumerate(tbs_bytes):
            x[p_, :len(b)] = b
        din = DevBuf.from_host(x)
        bufs = []
        for v, w in ((ack, 2), (ri, 2), (cqi, 64)):
            if v is None:
                bufs.append(None)
                continue
            a = np.zeros((max(n, 1), w), np.uint8)
            for p_, row in enumerate(v):
                a[p_, :len(row)] = row
            bufs.append(DevBuf.from_host(a))
        arr = (UlGrant * max(n, 1))(*grants)
        L = lib()
        L.srslte_hip_ul_tx_batch_grants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                    C.c_uint32, C.c_void_p, C.c_void_p]
        _check(L.srslte_hip_ul_tx_batch_grants(self.h, din.ptr, stride, *[b.ptr if b else None for b in bufs], tti0, nof_sf, arr, n, self.d_iq.ptr, None),
               "ul_tx_batch_grants")
        sync()
        return self.d_iq.to_host(np.complex64).reshape(self.max_batch, self.sf_len)[:nof_sf]

    def encode(self, tb, tti0=0, ack=None, ri=None, cqi=None, rv=None):
        """tb: [nof_sf][tbs/8] payload bytes (ack: [nof_sf][ack_len] HARQ-ACK values, ri: [nof_sf][ri_len] rank-indication bits,
        cqi: [nof_sf][cqi_len] report bits; rv: redundancy version through srslte_hip_ul_tx_batch_rv) -> iq [nof_sf][sf_len] (left on the
        device in self.d_iq). tbs = 0 (a PUSCH without UL-SCH data): tb is ignored, one subframe per row of cqi."""
        if self.tbs == 0:
            x = np.zeros((len(cqi), 1), np.uint8)
        else:
            x = np.ascontiguousarray(tb, np.uint8).reshape(-1, self.tbs // 8)
        din = DevBuf.from_host(x)
        if rv is not None:
            bufs = []
            for v, n, w in ((ack, self.cfg.ack_len, 2), (ri, self.cfg.ri_len, 2), (cqi, self.cfg.cqi_len, 64)):
                a = np.zeros((x.shape[0], w), np.uint8)
                if v is not None:
                    a[:, :n] = np.asarray(v, np.uint8).reshape(x.shape[0], -1)[:, :n]
                bufs.append(DevBuf.from_host(a) if v is not None else None)
            L = lib()
            L.srslte_hip_ul_tx_batch_rv.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                    C.c_uint32, C.c_void_p, C.c_void_p]
            _check(L.srslte_hip_ul_tx_batch_rv(self.h, din.ptr, self.tbs // 8, *[b.ptr if b else None for b in bufs], rv, tti0, x.shape[0],
                                               self.d_iq.ptr, None), "ul_tx_batch_rv")
        elif cqi is not None:
            bufs = []
            for v, n, w in ((ack, self.cfg.ack_len, 2), (ri, self.cfg.ri_len, 2), (cqi, self.cfg.cqi_len, 64)):
                a = np.zeros((x.shape[0], w), np.uint8)
                if v is not None:
                    a[:, :n] = np.asarray(v, np.uint8).reshape(x.shape[0], -1)[:, :n]
                bufs.append(DevBuf.from_host(a) if v is not None else None)
            _check(lib().srslte_hip_ul_tx_batch_uci_cqi(self.h, din.ptr, self.tbs // 8, bufs[0].ptr if bufs[0] else None,
                                                        bufs[1].ptr if bufs[1] else None, bufs[2].ptr, tti0, x.shape[0], self.d_iq.ptr, None),
                   "ul_tx_batch_uci_cqi")
        elif ri is not None:
            bufs = []
            for v, n in ((ack, self.cfg.ack_len), (ri, self.cfg.ri_len)):
                a = np.zeros((x.shape[0], 2), np.uint8)
                if v is not None:
                    a[:, :n] = np.asarray(v, np.uint8).reshape(x.shape[0], -1)[:, :n]
                bufs.append(DevBuf.from_host(a) if v is not None else None)
            _check(lib().srslte_hip_ul_tx_batch_uci(self.h, din.ptr, self.tbs // 8, bufs[0].ptr if bufs[0] else None, bufs[1].ptr, tti0, x.shape[0],
                                                    self.d_iq.ptr, None), "ul_tx_batch_uci")
        elif ack is not None:
            a = np.zeros((x.shape[0], 2), np.uint8)
            a[:, :self.cfg.ack_len] = np.asarray(ack, np.uint8).reshape(x.shape[0], -1)[:, :self.cfg.ack_len]
            dack = DevBuf.from_host(a)
            _check(lib().srslte_hip_ul_tx_batch_ack(self.h, din.ptr, self.tbs // 8, dack.ptr, tti0, x.shape[0], self.d_iq.ptr, None), "ul_tx_batch_ack")
        else:
            _check(lib().srslte_hip_ul_tx_batch(self.h, din.ptr, self.tbs // 8, tti0, x.shape[0], self.d_iq.ptr, None), "ul_tx_batch")
        sync()
        return self.d_iq.to_host(np.complex64).reshape(self.max_batch, self.sf_len)[:x.shape[0]]

    def debug(self, which, dtype, count):
        ptr = lib().srslte_hip_ul_tx_debug_buffer(self.h, which)
        out = np.empty(count, dtype)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.h:
            lib().srslte_hip_ul_tx_destroy(self.h)
            self.h = None


class DlTxCfg(C.Structure):
    _fields_ = [("cell_id", C.c_uint32), ("nof_prb", C.c_uint32), ("cfi", C.c_uint32), ("rnti", C.c_uint16), ("mod", C.c_int), ("tbs", C.c_uint32),
                ("max_batch", C.c_uint32), ("nof_ports", C.c_uint32), ("p_a", C.c_float), ("max_grants", C.c_uint32), ("cp_ext", C.c_int),
                ("tdd", C.c_int), ("tdd_sf_config", C.c_uint32), ("tdd_ss_config", C.c_uint32),
                ("mbsfn", C.c_int), ("mbsfn_area_id", C.c_uint32), ("non_mbsfn_region", C.c_uint32)]


class DlTx:
    """Batched PDSCH transmit chain (srslte_pdsch_encode pdsch.c:1059-1185 + CRS + srslte_ofdm_tx_sf, enb_dl.c)."""

    def __init__(self, cell_id, nof_prb, cfi, rnti, mod, tbs, max_batch, nof_ports=1, p_a=0.0, max_grants=0, cp_ext=False, tdd=None, mbsfn=None):
        self.cfg = DlTxCfg(cell_id, nof_prb, cfi, rnti, mod, tbs, max_batch, nof_ports, p_a, max_grants, 1 if cp_ext else 0,
                           1 if tdd else 0, tdd[0] if tdd else 0, tdd[1] if tdd else 0, 1 if mbsfn else 0, mbsfn[0] if mbsfn else 0, mbsfn[1] if mbsfn else 0)
        L = lib()
        L.srslte_hip_dl_tx_create.restype = C.c_void_p
        L.srslte_hip_dl_tx_create.argtypes = [C.POINTER(DlTxCfg)]
        L.srslte_hip_dl_tx_destroy.argtypes = [C.c_void_p]
        L.srslte_hip_dl_tx_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.srslte_hip_dl_tx_debug_buffer.restype = C.c_void_p
        L.srslte_hip_dl_tx_debug_buffer.argtypes = [C.c_void_p, C.c_int]
        self.h = L.srslte_hip_dl_tx_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("srslte_hip_dl_tx_create failed")
        self.tbs, self.max_batch, self.nof_ports = tbs, max_batch, max(1, nof_ports)
        self.sf_len = 15 * symbol_sz(nof_prb)
        self.d_iq = DevBuf(8 * self.sf_len * max_batch * self.nof_ports)

    def encode_grants(self, tbs_bytes, tti0, nof_sf, grants):
        """srslte_hip_dl_tx_batch_grants: grants = list of (sf, DlGrant); tbs_bytes[p] = the payload of grants[p] -> iq [nof_sf][nof_ports][sf_len]."""
        class TxGrant(C.Structure):
            _fields_ = [("sf", C.c_uint32), ("grant", DlGrant)]
        stride = (self.tbs // 8 + 15) & ~15
        x = np.zeros((len(grants), stride), np.uint8)
        for p_, b in enumerate(tbs_bytes):
            x[p_, :len(b)] = b
        din = DevBuf.from_host(x)
        arr = (TxGrant * max(1, len(grants)))(*[TxGrant(sf, g) for sf, g in grants])
        L = lib()
        L.srslte_hip_dl_tx_batch_grants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _check(L.srslte_hip_dl_tx_batch_grants(self.h, din.ptr, stride, tti0, nof_sf, arr, len(grants), self.d_iq.ptr, None), "dl_tx_batch_grants")
        sync()
        return self.d_iq.to_host(np.complex64).reshape(self.max_batch, self.nof_ports, self.sf_len)[:nof_sf]

    def encode(self, tb, tti0=0, rv=0):
        """tb: [nof_sf][tbs/8] payload bytes -> iq [nof_sf][nof_ports][sf_len]."""
        x = np.ascontiguousarray(tb, np.uint8).reshape(-1, self.tbs // 8)
        din = DevBuf.from_host(x)
        _check(lib().srslte_hip_dl_tx_batch(self.h, din.ptr, self.tbs // 8, tti0, x.shape[0], rv, self.d_iq.ptr, None), "dl_tx_batch")
        sync()
        return self.d_iq.to_host(np.complex64).reshape(self.max_batch, self.nof_ports, self.sf_len)[:x.shape[0]]

    def debug(self, which, dtype, count):
        ptr = lib().srslte_hip_dl_tx_debug_buffer(self.h, which)
        out = np.empty(count, dtype)
        _check(lib().srslte_hip_memcpy_d2h(out.ctypes.data, ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.h:
            lib().srslte_hip_dl_tx_destroy(self.h)
            self.h = None
