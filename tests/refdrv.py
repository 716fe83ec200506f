"""ctypes front of oracle/_ref/librefdrv.so (oracle/refdrv.c): the reference's own compiled receive functions driven the way its
callers drive them, from a frequency-domain subframe on. Test infrastructure only; present only where oracle/ref.mk has run."""
import ctypes as C
import os

import numpy as np

from _libs import ORACLE_DIR, RefChestRes, ref

REFDRV_SO = os.path.join(ORACLE_DIR, "_ref", "librefdrv.so")
IQ_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "iq")

_lib = None


def lib():
    global _lib
    if _lib is None and os.path.exists(REFDRV_SO) and ref() is not None:
        L = C.CDLL(REFDRV_SO)
        L.refdrv_dl_new.restype = C.c_void_p
        L.refdrv_dl_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int]
        for f in ("refdrv_dl_grid", "refdrv_dl_ce", "refdrv_dl_payload", "refdrv_dl_chest_res"):
            getattr(L, f).restype = C.c_void_p
        L.refdrv_dl_grid.argtypes = [C.c_void_p, C.c_uint32]
        L.refdrv_dl_ce.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.refdrv_dl_payload.argtypes = [C.c_void_p]
        L.refdrv_dl_chest_res.argtypes = [C.c_void_p]
        L.refdrv_dl_free.argtypes = [C.c_void_p]
        L.refdrv_dl_set_rnti.argtypes = [C.c_void_p, C.c_uint16]
        L.refdrv_dl_set_mbsfn_area_id.argtypes = [C.c_void_p, C.c_uint16]
        L.refdrv_dl_set_chest_cfg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_uint16]
        L.refdrv_dl_set_pdsch_cfg.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.refdrv_dl_estimate.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.refdrv_dl_pcfich.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.refdrv_dl_find_dci.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.refdrv_dl_set_grant.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint16, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.refdrv_dl_chest.argtypes = [C.c_void_p]
        L.refdrv_dl_set_prb_masks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.refdrv_dl_grant_info.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.refdrv_dl_set_grant_type2.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                                C.c_void_p, C.c_void_p]
        L.refdrv_dl_encode_pdsch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.refdrv_dl_decode_pdsch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.refdrv_dl_pmch_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint32, C.c_void_p]
        L.refdrv_dl_pbch_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.refdrv_dl_rx_loop.restype = C.c_double
        L.refdrv_dl_rx_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint16, C.c_uint32,
                                        C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class RefDl:
    """One UE-side receiver of the reference (chest_dl, PCFICH, PDCCH, PDSCH, PMCH, PBCH objects of one cell)."""

    def __init__(self, nof_prb, nof_ports, cell_id, cp_ext=False, nof_rx=1, phich_resources=2, phich_ext=False):
        self.L = lib()
        assert self.L is not None, "oracle/_ref/librefdrv.so is not built"
        self.h = self.L.refdrv_dl_new(nof_prb, nof_ports, cell_id, int(cp_ext), nof_rx, phich_resources, int(phich_ext))
        assert self.h, "reference receiver init failed"
        self.nof_prb, self.nsym = nof_prb, 12 if cp_ext else 14
        self.grid_len = self.nsym * 12 * nof_prb

    def free(self):
        if self.h:
            self.L.refdrv_dl_free(self.h)
            self.h = None

    def put_grid(self, grid, ant=0):
        g = np.ascontiguousarray(grid, np.complex64).ravel()
        assert g.size == self.grid_len
        C.memmove(self.L.refdrv_dl_grid(self.h, ant), g.ctypes.data, g.nbytes)

    def ce(self, port=0, ant=0):
        out = np.empty(self.grid_len, np.complex64)
        C.memmove(out.ctypes.data, self.L.refdrv_dl_ce(self.h, port, ant), out.nbytes)
        return out

    def chest_res(self):
        return RefChestRes.from_address(self.L.refdrv_dl_chest_res(self.h))

    def payload(self, nbytes):
        out = np.empty(nbytes, np.uint8)
        C.memmove(out.ctypes.data, self.L.refdrv_dl_payload(self.h), nbytes)
        return out

    def set_rnti(self, rnti):
        self.L.refdrv_dl_set_rnti(self.h, rnti)

    def set_chest_cfg(self, noise_alg=0, filter_type=0, coef=(0.0, 0.0), interpolate_subframe=False, mbsfn_area_id=0):
        self.L.refdrv_dl_set_chest_cfg(self.h, noise_alg, filter_type, coef[0], coef[1], int(interpolate_subframe), mbsfn_area_id)

    def set_pdsch_cfg(self, max_iterations=0, mmse=False, csi=False, llr8=False):
        self.L.refdrv_dl_set_pdsch_cfg(self.h, max_iterations, int(mmse), int(csi), int(llr8))

    def estimate(self, tti, mbsfn=False, cfi_in=0):
        cfi, corr = C.c_uint32(0), C.c_float(0)
        rc = self.L.refdrv_dl_estimate(self.h, tti, int(mbsfn), cfi_in, C.byref(cfi), C.byref(corr))
        return rc, cfi.value, corr.value

    def pcfich(self, tti=0):
        cfi, corr = C.c_uint32(0), C.c_float(0)
        n = self.L.refdrv_dl_pcfich(self.h, tti, C.byref(cfi), C.byref(corr))
        return n, cfi.value, corr.value

    def find_dci(self, rnti, tm=0):
        mcs, tbs, nprb, rv = C.c_uint32(0), C.c_int(0), C.c_uint32(0), C.c_int(0)
        rc = self.L.refdrv_dl_find_dci(self.h, rnti, tm, C.byref(mcs), C.byref(tbs), C.byref(nprb), C.byref(rv))
        return rc, {"mcs": mcs.value, "tbs": tbs.value, "nof_prb": nprb.value, "rv": rv.value}

    def set_grant(self, tti, cfi, rnti, mcs, rbg_bitmask=0xffffffff, tm=0, rv=0, alt=False):
        tbs, nre = C.c_int(0), C.c_uint32(0)
        rc = self.L.refdrv_dl_set_grant(self.h, tti, cfi, rnti, tm, rbg_bitmask, mcs, rv, int(alt), C.byref(tbs), C.byref(nre))
        assert rc == 0
        return tbs.value, nre.value

    def chest(self):
        return self.L.refdrv_dl_chest(self.h)

    def set_grant_type2(self, tti, cfi, rnti, mcs, L_crb, RB_start, distributed=False, n_gap2=False, rv=0):
        tbs, nre = C.c_int(0), C.c_uint32(0)
        rc = self.L.refdrv_dl_set_grant_type2(self.h, tti, cfi, rnti, L_crb, RB_start, int(distributed), int(n_gap2), mcs, rv, C.byref(tbs), C.byref(nre))
        assert rc == 0
        return tbs.value, nre.value

    def set_prb_masks(self, slot0, slot1):
        a, b = np.ascontiguousarray(slot0, np.uint8), np.ascontiguousarray(slot1, np.uint8)
        assert a.size == self.nof_prb and b.size == self.nof_prb
        return self.L.refdrv_dl_set_prb_masks(self.h, a.ctypes.data, b.ctypes.data)

    def grant_info(self):
        m0, m1 = np.zeros(self.nof_prb, np.uint8), np.zeros(self.nof_prb, np.uint8)
        mod, tbs, nre, nbits, rv = C.c_int(0), C.c_int(0), C.c_uint32(0), C.c_uint32(0), C.c_int(0)
        self.L.refdrv_dl_grant_info(self.h, m0.ctypes.data, m1.ctypes.data, C.byref(mod), C.byref(tbs), C.byref(nre), C.byref(nbits), C.byref(rv))
        return {"prb_mask": np.stack([m0, m1]), "mod": mod.value, "tbs": tbs.value, "nof_re": nre.value, "nof_bits": nbits.value, "rv": rv.value}

    def encode_pdsch(self, data):
        grid = np.zeros(self.grid_len, np.complex64)
        d = np.ascontiguousarray(data, np.uint8)
        assert self.L.refdrv_dl_encode_pdsch(self.h, d.ctypes.data, grid.ctypes.data) == 0
        return grid

    def decode_pdsch(self, new_data=True):
        it = C.c_float(0)
        crc = self.L.refdrv_dl_decode_pdsch(self.h, int(new_data), C.byref(it))
        return crc, it.value

    def pmch_decode(self, cfi, area_id, mcs):
        tbs = C.c_int(0)
        crc = self.L.refdrv_dl_pmch_decode(self.h, cfi, area_id, mcs, C.byref(tbs))
        return crc, tbs.value

    def pbch_decode(self):
        ports, off = C.c_uint32(0), C.c_int(0)
        bch = np.zeros(24, np.uint8)
        n = self.L.refdrv_dl_pbch_decode(self.h, C.byref(ports), C.byref(off), bch.ctypes.data_as(C.c_void_p))
        return n, ports.value, off.value, bch


def read_iq(name, nsamp, offset=0):
    """`nsamp` complex samples of a recorded capture, zero-padded past the end of the file as the reference's tests leave their zeroed
    input buffer when srslte_filesource_read comes back short."""
    a = np.fromfile(os.path.join(IQ_DIR, name), np.complex64)
    out = np.zeros(nsamp, np.complex64)
    seg = a[offset:offset + nsamp]
    out[:seg.size] = seg
    return out
