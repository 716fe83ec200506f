"""Test-side loaders for the three native libraries.

* ``oracle()``  - oracle/liboracle.so, the CPU restatement (test infrastructure; built on demand with gcc).
* ``ref()``     - oracle/_ref/libsrslte_ref.so, the reference's own sources compiled where they lie
                  (only exists where /root/reference was present at build time; ``None`` otherwise).
* ``hip()``     - srslte-emane_amd/csrc/libsrslte_phy_hip.so, the product. Never built here implicitly
                  except through __graft_entry__.build().

ctypes mirrors of the reference structs used by the parity harness follow the reference headers
(lib/include/srslte/phy/common/phy_common.h:195-212, ch_estimation/chest_dl.h:49-130).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsrslte_ref.so")
PKG_DIR = os.path.join(ROOT, "srslte-emane_amd")
HIP_SO = os.path.join(PKG_DIR, "csrc", "libsrslte_phy_hip.so")

_cache = {}


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def aligned(n, dtype, al=64):
    """numpy array whose data pointer is `al`-byte aligned (the reference uses aligned SIMD loads)."""
    isz = np.dtype(dtype).itemsize
    raw = np.zeros(n * isz + al, np.uint8)
    off = (-raw.ctypes.data) % al
    return raw[off:off + n * isz].view(dtype)


def acopy(a):
    b = aligned(a.size, a.dtype)
    b[:] = a.ravel()
    return b


def oracle():
    if "orc" not in _cache:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.startswith("orc") and f[-2:] in (".c", ".h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
        lib = C.CDLL(so)
        lib.orc_crc_bytes.restype = C.c_uint32
        lib.orc_pdsch_cinit.restype = C.c_uint32
        lib.orc_dft_precoding_valid_prb.restype = C.c_bool
        lib.orc_predecoding_single.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
        _cache["orc"] = lib
    return _cache["orc"]


def ref():
    if "ref" not in _cache:
        lib = None
        if os.path.exists(REF_SO):
            lib = C.CDLL(REF_SO)
            lib.srslte_rm_turbo_gentables()
            lib.srslte_predecoding_single.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
            lib.srslte_chest_set_smooth_filter_gauss.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        _cache["ref"] = lib
    return _cache["ref"]


def ref_sz():
    """oracle/_ref/libsrslte_ref_sz.so: the reference with mimo/precoding.c compiled with -fsigned-zeros (oracle/ref.mk says why): the
    build whose large-delay-CDD receiver does what its source says. Only the two-layer tests use it."""
    if "ref_sz" not in _cache:
        path = REF_SO.replace("libsrslte_ref.so", "libsrslte_ref_sz.so")
        lib = None
        if os.path.exists(path):
            lib = C.CDLL(path)
            lib.srslte_rm_turbo_gentables()
        _cache["ref_sz"] = lib
    return _cache["ref_sz"]


def hip():
    """The product library. Raises if it has not been built: tests must fail loudly, never fall back."""
    if "hip" not in _cache:
        if not os.path.exists(HIP_SO):
            raise RuntimeError("libsrslte_phy_hip.so is not built; run python -c 'import __graft_entry__ as g; g.build()'")
        _cache["hip"] = C.CDLL(HIP_SO)
    return _cache["hip"]


# ---------------------------------------------------------------- oracle structs
class OrcCell(C.Structure):
    # frame_type 1 = TDD with its uplink-downlink and special-subframe configurations (all zero: FDD, as before the fields existed)
    _fields_ = [("id", C.c_uint32), ("nof_prb", C.c_uint32), ("nof_ports", C.c_uint32), ("cp_norm", C.c_bool), ("frame_type", C.c_uint32),
                ("tdd_sf_config", C.c_uint32), ("tdd_ss_config", C.c_uint32)]


class OrcChestCfg(C.Structure):
    _fields_ = [("noise_alg", C.c_int), ("filter_type", C.c_int), ("filter_coef", C.c_float * 2),
                ("interpolate_subframe", C.c_bool), ("rsrp_neighbour", C.c_bool), ("cfo_estimate_enable", C.c_bool),
                ("sync_error_enable", C.c_bool)]


class OrcChestRes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("noise_estimate", "noise_estimate_dbm", "snr_db", "rsrp", "rsrp_dbm", "rsrq", "rsrq_db",
                                         "rssi_dbm", "cfo", "sync_error", "rsrp_neigh")]


class OrcOfdm(C.Structure):
    _fields_ = [("nof_prb", C.c_int), ("symbol_sz", C.c_int), ("nof_re", C.c_int), ("nof_symbols", C.c_int), ("sf_sz", C.c_int),
                ("slot_sz", C.c_int), ("cp_norm", C.c_int), ("normalize", C.c_bool), ("freq_shift", C.c_bool),
                ("freq_shift_f", C.c_float), ("exact", C.c_bool), ("non_mbsfn_region", C.c_int)]


class OrcSchCfg(C.Structure):
    _fields_ = [("tbs", C.c_uint32), ("nof_bits", C.c_uint32), ("Qm", C.c_uint32), ("rv", C.c_uint32), ("max_iter", C.c_uint32)]


class OrcCbsegm(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("F", "C", "K1", "K2", "K1_idx", "K2_idx", "C1", "C2", "tbs")]


# ---------------------------------------------------------------- reference structs
class RefCell(C.Structure):
    _fields_ = [("nof_prb", C.c_uint32), ("nof_ports", C.c_uint32), ("id", C.c_uint32), ("cp", C.c_int), ("phich_length", C.c_int),
                ("phich_resources", C.c_int), ("frame_type", C.c_int)]


class RefTddCfg(C.Structure):
    _fields_ = [("sf_config", C.c_uint32), ("ss_config", C.c_uint32), ("configured", C.c_bool)]


class RefDlSfCfg(C.Structure):
    _fields_ = [("tdd_config", RefTddCfg), ("tti", C.c_uint32), ("cfi", C.c_uint32), ("sf_type", C.c_int), ("non_mbsfn_region", C.c_uint32)]


class RefChestRes(C.Structure):
    _fields_ = [("ce", (C.c_void_p * 4) * 4), ("nof_re", C.c_uint32), ("noise_estimate", C.c_float), ("noise_estimate_dbm", C.c_float),
                ("snr_db", C.c_float), ("snr_ant_port_db", (C.c_float * 4) * 4), ("rsrp", C.c_float), ("rsrp_dbm", C.c_float),
                ("rsrp_neigh", C.c_float), ("rsrp_port_dbm", C.c_float * 4), ("rsrp_ant_port_dbm", (C.c_float * 4) * 4),
                ("rsrq", C.c_float), ("rsrq_db", C.c_float), ("rsrq_ant_port_db", (C.c_float * 4) * 4), ("rssi_dbm", C.c_float),
                ("cfo", C.c_float), ("sync_error", C.c_float)]


class RefChestCfg(C.Structure):
    _fields_ = [("noise_alg", C.c_int), ("filter_type", C.c_int), ("filter_coef", C.c_float * 2), ("mbsfn_area_id", C.c_uint16),
                ("interpolate_subframe", C.c_bool), ("rsrp_neighbour", C.c_bool), ("cfo_estimate_enable", C.c_bool),
                ("cfo_estimate_sf_mask", C.c_uint32), ("sync_error_enable", C.c_bool)]


def opaque(nbytes=1 << 20):
    """Zeroed storage for a reference object struct the harness never inspects (srslte_tdec_t, srslte_chest_dl_t ...)."""
    return C.create_string_buffer(nbytes)


class SrslteCrc(C.Structure):
    """srslte_crc_t (fec/crc.h:38-46)."""
    _fields_ = [("table", C.c_uint64 * 256), ("polynom", C.c_int), ("order", C.c_int), ("crcinit", C.c_uint64), ("crcmask", C.c_uint64),
                ("crchighbit", C.c_uint64), ("srslte_crc_out", C.c_uint32)]


def make_crc(poly, order):
    """What srslte_crc_init leaves in the struct (crc.c:30-46,:70-96), for calls that take a caller-initialised srslte_crc_t."""
    h = SrslteCrc()
    h.polynom, h.order, h.crcinit = poly, order, 0
    h.crcmask, h.crchighbit = (1 << order) - 1, 1 << (order - 1)
    for i in range(256):
        crc = i << (order - 8)
        for _ in range(8):
            bit = crc & h.crchighbit
            crc <<= 1
            if bit:
                crc ^= poly
        h.table[i] = crc & h.crcmask
    return h


class OrcUlDmrs(C.Structure):
    _fields_ = [("cell_id", C.c_uint32), ("n_prs", (C.c_uint32 * 20) * 30), ("f_gh", C.c_uint32 * 20), ("v", (C.c_uint32 * 30) * 20)]


class OrcUlDmrsCfg(C.Structure):
    """Also the layout of srslte_refsignal_dmrs_pusch_cfg_t (refsignal_ul.h:46-51)."""
    _fields_ = [("cyclic_shift", C.c_uint32), ("delta_ss", C.c_uint32), ("group_hopping_en", C.c_bool), ("sequence_hopping_en", C.c_bool)]


class OrcChestUlRes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("noise_estimate", "noise_estimate_dbm", "snr", "snr_db", "cfo")]


class RefChestUlRes(C.Structure):
    """srslte_chest_ul_res_t (chest_ul.h:47-55)."""
    _fields_ = [("ce", C.c_void_p), ("nof_re", C.c_uint32), ("noise_estimate", C.c_float), ("noise_estimate_dbm", C.c_float), ("snr", C.c_float),
                ("snr_db", C.c_float), ("cfo", C.c_float)]


def ref_pusch_cfg(L_prb, n_prb, n_dmrs, n_prb_slot1=None):
    """A zeroed srslte_pusch_cfg_t (pusch_cfg.h:62-86) with the grant fields the UL estimator reads; offsets from the reference headers
    (checked by tests/test_abi_layout.py where those headers are available)."""
    buf = (C.c_uint8 * 520)()
    u32 = C.cast(buf, C.POINTER(C.c_uint32))
    u32[388 // 4] = L_prb
    n1 = n_prb if n_prb_slot1 is None else n_prb_slot1
    u32[392 // 4], u32[396 // 4] = n_prb, n1             # n_prb[2]
    u32[400 // 4], u32[404 // 4] = n_prb, n1             # n_prb_tilde[2]
    u32[476 // 4] = n_dmrs
    return buf


def ref_ul_sf_cfg(tti):
    buf = (C.c_uint8 * 20)()
    C.cast(buf, C.POINTER(C.c_uint32))[12 // 4] = tti
    return buf


_ref_layout_cache = {}


def ref_layout(structs, includes):
    """{"struct": [fields]} -> {"struct": sizeof, "struct.field": offsetof} from the REFERENCE headers (compiled on the fly with gcc;
    only where /root/reference exists). For driving reference functions that take configuration structs by pointer."""
    import subprocess
    import tempfile
    key = repr((sorted((k, tuple(v)) for k, v in structs.items()), tuple(includes)))
    if key in _ref_layout_cache:
        return _ref_layout_cache[key]
    body = "".join('  printf("%s %%zu\\n", sizeof(%s));\n' % (st, st) + "".join('  printf("%s.%s %%zu\\n", offsetof(%s, %s));\n' % (st, f, st, f) for f in fs)
                   for st, fs in structs.items())
    src = "#include <stdio.h>\n#include <stddef.h>\n" + "".join('#include "%s"\n' % i for i in includes) + "int main(void) {\n" + body + "  return 0;\n}\n"
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        with open(c, "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-std=c99", "-D_GNU_SOURCE", "-w", "-I/root/reference/lib/include", c, "-o", exe])
        out = {a: int(b) for a, b in (line.split() for line in subprocess.check_output([exe]).decode().splitlines())}
    _ref_layout_cache[key] = out
    return out
