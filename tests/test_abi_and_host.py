"""CPU-side checks of the product library: it loads, exports every symbol include/*.h declares, and its host-only
entry points (segmentation, interleaver tables, argument validation) agree with the oracle. No kernel is launched."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from _libs import HIP_SO, ROOT, OrcCbsegm, oracle, p

pytestmark = pytest.mark.skipif(not os.path.exists(HIP_SO), reason="libsrslte_phy_hip.so not built (run __graft_entry__.build())")


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include", "srslte_hip")
    for f in os.listdir(inc):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, f)).read(), flags=re.S)
        names.update(re.findall(r"\b(srslte_[a-zA-Z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(HIP_SO)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, "declared in include/ but not exported: %s" % missing
    assert len(declared_symbols()) > 40


def test_host_mirror_binds():
    pkg = importlib.import_module("srslte-emane_amd")
    assert pkg.lib() is not None


def test_cbsegm_matches_oracle():
    pkg = importlib.import_module("srslte-emane_amd")
    for tbs in list(range(16, 6200, 8))[::11] + [6120, 6144, 6200, 75376, 97896, 149776, 0]:
        rc, s = pkg.cbsegm(tbs)
        r = OrcCbsegm()
        assert oracle().orc_cbsegm(C.byref(r), tbs) == rc
        assert all(getattr(s, f) == getattr(r, f) for f, _ in OrcCbsegm._fields_), tbs
    lib = pkg.lib()
    assert lib.srslte_hip_cbsegm_cbindex(6145) == -1 and lib.srslte_hip_cbsegm_cbsize(188) == -1
    assert lib.srslte_hip_cbsegm_cbindex(41) == 1 and lib.srslte_hip_cbsegm_cbsize(187) == 6144


def test_interleaver_tables_match_oracle():
    pkg = importlib.import_module("srslte-emane_amd")
    for K, W in ((40, 1), (176, 1), (504, 8), (1008, 16), (5824, 16), (6144, 16), (6144, 8), (6144, 32)):
        rc, f, r = pkg.tc_interl(K, W)
        rf, rr = np.zeros(K, np.uint16), np.zeros(K, np.uint16)
        assert oracle().orc_qpp(K, W, p(rf), p(rr)) == rc == 0
        assert np.array_equal(f, rf) and np.array_equal(r, rr)
        assert np.array_equal(np.sort(f), np.arange(K))
    assert pkg.tc_interl(41, 1)[0] == pkg.SRSLTE_ERROR


def test_argument_validation_without_gpu():
    lib = importlib.import_module("srslte-emane_amd").lib()
    assert lib.srslte_hip_dft_precoding_valid_prb(7) == 0 and lib.srslte_hip_dft_precoding_valid_prb(100) == 1
    assert lib.srslte_hip_tdec_autoimp_get_subblocks(400) == 0 and lib.srslte_hip_tdec_autoimp_get_subblocks(408) == 8
    assert lib.srslte_hip_tdec_autoimp_get_subblocks(800) == 8 and lib.srslte_hip_tdec_autoimp_get_subblocks(816) == 16
    assert lib.srslte_hip_tdec_input_len(5824, 1) == 3 * (5824 + 32) + 12 and lib.srslte_hip_tdec_input_len(40, 0) == 132
    assert lib.srslte_hip_ofdm_rx_sf_batch(None, None, None, 1, None) == -2
    assert lib.srslte_hip_demod_soft_demodulate_s_batch(9, None, None, 1, 1, None) == -1


def test_pipeline_entry_points_reject_bad_arguments_without_gpu():
    """The batched pipelines validate handles and pointers before touching the device: SRSLTE_ERROR_INVALID_INPUTS (-2), no crash,
    and create() refuses configurations the kernels do not cover (NULL back, message on stderr)."""
    import ctypes as C
    pkg = importlib.import_module("srslte-emane_amd")
    lib = pkg.lib()
    vp = C.c_void_p
    for name in ("srslte_hip_dl_rx_batch", "srslte_hip_dl_rx_grid_batch", "srslte_hip_ul_rx_batch"):
        getattr(lib, name).argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, vp]
        assert getattr(lib, name)(None, None, 0, 1, None, 0, None, None) == -2
    lib.srslte_hip_dl_rx_batch_harq.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, C.c_uint32, vp, vp]
    assert lib.srslte_hip_dl_rx_batch_harq(None, None, 0, 1, 0, 1, None, 0, None, None) == -2
    lib.srslte_hip_ul_tx_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    assert lib.srslte_hip_ul_tx_batch(None, None, 0, 0, 1, None, None) == -2
    lib.srslte_hip_dl_tx_batch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    assert lib.srslte_hip_dl_tx_batch(None, None, 0, 0, 1, 0, None, None) == -2
    lib.srslte_hip_dl_rx_keep_symbols.argtypes = [vp, C.c_int]
    assert lib.srslte_hip_dl_rx_keep_symbols(None, 1) == -2
    for fn, cfg in ((lib.srslte_hip_dl_rx_create, pkg.DlRxCfg(1, 100, 1, 1, 3, 75376, 6, 4, 1, pkg.ChestDlCfg(), 0, 1, 3, 0, 0, 0.0)),   # 3 ports
                    (lib.srslte_hip_dl_rx_create, pkg.DlRxCfg(1, 100, 1, 1, 7, 75376, 6, 4, 1, pkg.ChestDlCfg(), 0, 1, 1, 0, 0, 0.0))):  # bad modulation
        fn.restype = vp
        assert fn(C.byref(cfg)) is None
    for L in (lib.srslte_hip_dl_rx_debug_buffer, lib.srslte_hip_ul_rx_debug_buffer, lib.srslte_hip_ul_tx_debug_buffer, lib.srslte_hip_dl_tx_debug_buffer):
        L.restype, L.argtypes = vp, [vp, C.c_int]
        assert L(None, 0) is None


def test_round_additions_reject_bad_arguments_without_gpu():
    """Entry points added for UCI on the PUSCH, MBSFN estimation and the noise algorithms: the same contract (-2 / NULL, no crash)."""
    import ctypes as C
    pkg = importlib.import_module("srslte-emane_amd")
    lib = pkg.lib()
    vp = C.c_void_p
    lib.srslte_hip_ul_tx_batch_ack.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp]
    assert lib.srslte_hip_ul_tx_batch_ack(None, None, 0, None, 0, 1, None, None) == -2
    lib.srslte_hip_ul_tx_batch_uci.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    assert lib.srslte_hip_ul_tx_batch_uci(None, None, 0, None, None, 0, 1, None, None) == -2
    for fn in (lib.srslte_hip_ul_rx_ack, lib.srslte_hip_ul_rx_ri):
        fn.restype, fn.argtypes = vp, [vp]
        assert fn(None) is None
    lib.srslte_hip_chest_dl_set_mbsfn_area_id.argtypes = [vp, C.c_uint16]
    assert lib.srslte_hip_chest_dl_set_mbsfn_area_id(None, 3) == -2
    lib.srslte_hip_chest_dl_mbsfn_pilots.restype, lib.srslte_hip_chest_dl_mbsfn_pilots.argtypes = vp, [vp, C.c_uint16]
    assert lib.srslte_hip_chest_dl_mbsfn_pilots(None, 3) is None
    cfg = pkg.ChestDlCfg()
    lib.srslte_hip_chest_dl_estimate_mbsfn_batch.argtypes = [vp, C.POINTER(pkg.ChestDlCfg), C.c_uint32, vp, vp, vp, C.c_int, C.c_int, vp]
    assert lib.srslte_hip_chest_dl_estimate_mbsfn_batch(None, C.byref(cfg), 0, None, None, None, 1, 1, None) == -2
    lib.srslte_hip_chest_dl_estimate_batch_multi.argtypes = [vp, C.POINTER(pkg.ChestDlCfg), C.c_uint32, vp, vp, vp, C.c_int, C.c_int, vp]
    assert lib.srslte_hip_chest_dl_estimate_batch_multi(None, C.byref(cfg), 0, None, None, None, 1, 1, None) == -2
    # UCI configurations the tables reserve (36.213 Table 8.6.3-1 index 15, Table 8.6.3-2 index 13) or more than 2 bits: create() refuses
    lib.srslte_hip_ul_tx_create.restype = vp
    base = (1, 25, 0x1234, 2, 4008, 10, 5, 0, 4, pkg.DmrsPuschCfg(0, 0, 0, 0), 0)
    for uci in ((1, 15, 0, 0), (3, 0, 0, 0), (0, 0, 1, 13), (0, 0, 3, 0)):
        assert lib.srslte_hip_ul_tx_create(C.byref(pkg.UlTxCfg(*base, *uci))) is None


def test_round2_additions_reject_bad_arguments_without_gpu():
    """Entry points added in round 2 (per-subframe grants, two-layer modes, CQI reports, hopping): -2 / NULL on missing handles and pointers,
    and configurations the reference refuses are refused at creation, all before anything touches a device."""
    import ctypes as C
    pkg = importlib.import_module("srslte-emane_amd")
    lib = pkg.lib()
    vp, u32 = C.c_void_p, C.c_uint32
    lib.srslte_hip_dl_rx_batch_grants.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, vp]
    assert lib.srslte_hip_dl_rx_batch_grants(None, None, 0, 1, None, None, 0, None, None) == -2
    lib.srslte_hip_dl_rx_batch_harq2.argtypes = [vp, vp, u32, u32, vp, vp, vp, u32, vp, vp]
    assert lib.srslte_hip_dl_rx_batch_harq2(None, None, 0, 1, None, None, None, 0, None, None) == -2
    lib.srslte_hip_ul_tx_batch_uci_cqi.argtypes = [vp, vp, u32, vp, vp, vp, u32, u32, vp, vp]
    assert lib.srslte_hip_ul_tx_batch_uci_cqi(None, None, 0, None, None, None, 0, 1, None, None) == -2
    lib.srslte_hip_ul_rx_batch_harq.argtypes = [vp, vp, u32, u32, u32, C.c_int, vp, u32, vp, vp]
    assert lib.srslte_hip_ul_rx_batch_harq(None, None, 0, 1, 0, 1, None, 0, None, None) == -2
    lib.srslte_hip_ul_tx_batch_rv.argtypes = [vp, vp, u32, vp, vp, vp, u32, u32, u32, vp, vp]
    assert lib.srslte_hip_ul_tx_batch_rv(None, None, 0, None, None, None, 0, 0, 1, None, None) == -2
    lib.srslte_hip_ul_tx_batch_grants.argtypes = [vp, vp, u32, vp, vp, vp, u32, u32, vp, u32, vp, vp]
    assert lib.srslte_hip_ul_tx_batch_grants(None, None, 0, None, None, None, 0, 1, None, 1, None, None) == -2
    lib.srslte_hip_dl_rx_grid_batch_grants2.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, vp]
    assert lib.srslte_hip_dl_rx_grid_batch_grants2(None, None, 0, 1, None, None, 0, None, None) == -2
    lib.srslte_hip_dl_tx_batch_grants.argtypes = [vp, vp, u32, u32, u32, vp, u32, vp, vp]
    assert lib.srslte_hip_dl_tx_batch_grants(None, None, 0, 0, 1, None, 1, None, None) == -2
    lib.srslte_hip_ul_rx_batch_grants.argtypes = [vp, vp, u32, u32, vp, u32, vp, u32, vp, vp]
    assert lib.srslte_hip_ul_rx_batch_grants(None, None, 0, 1, None, 1, None, 0, None, None) == -2
    lib.srslte_hip_ul_rx_cqi.restype, lib.srslte_hip_ul_rx_cqi.argtypes = vp, [vp]
    assert lib.srslte_hip_ul_rx_cqi(None) is None
    lib.srslte_hip_chest_ul_estimate_pusch_batch_hop.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp, vp, C.c_int, vp]
    assert lib.srslte_hip_chest_ul_estimate_pusch_batch_hop(None, 0, 1, 0, 0, 0, None, None, None, 1, None) == -2
    lib.srslte_hip_dl_rx_create.restype = vp
    ok2 = dict(nof_rx=2, nof_ports=2)
    for kw in (dict(tx_scheme=3, mod2=2, tbs2=4008, nof_rx=1, nof_ports=2), dict(tx_scheme=3, tbs2=0, **ok2), dict(tx_scheme=2, pmi=2, mod2=2, tbs2=4008, **ok2),
               dict(tx_scheme=2, pmi=4, **ok2), dict(tx_scheme=5, mod2=2, tbs2=4008, **ok2), dict(tx_scheme=3, mod2=7, tbs2=4008, **ok2)):
        cfg = pkg.DlRxCfg(7, 25, 1, 0x1234, 2, 4008, 6, 2, 1, pkg.ChestDlCfg(), 0, kw.get("nof_rx", 1), kw.get("nof_ports", 1), 0, 0, 0.0, kw.get("tx_scheme", 0),
                          kw.get("pmi", 0), kw.get("mod2", 0), kw.get("tbs2", 0))
        assert lib.srslte_hip_dl_rx_create(C.byref(cfg)) is None, kw
    lib.srslte_hip_ul_rx_create.restype = vp
    lib.srslte_hip_ul_tx_create.restype = vp
    dm = pkg.DmrsPuschCfg(0, 0, 0, 0)
    for tail in ((4, 0, 0, 0), (4, 16, 0, 0), (65, 5, 0, 0), (0, 0, 1, 20)):  # cqi_len, I_offset_cqi, hopping, n_prb_slot1 (20 + 10 PRB > 25)
        assert lib.srslte_hip_ul_tx_create(C.byref(pkg.UlTxCfg(1, 25, 0x1234, 2, 4008, 10, 5, 0, 4, dm, 0, 0, 0, 0, 0, *tail))) is None, tail
        assert lib.srslte_hip_ul_rx_create(C.byref(pkg.UlRxCfg(1, 25, 0x1234, 2, 4008, 10, 5, 0, 6, 4, 1, dm, 0, 0, 0, 0, 0, *tail))) is None, tail


def test_symbol_size_families_without_gpu():
    """srslte_symbol_sz / srslte_symbol_sz_power2 / srslte_use_standard_symbol_size as phy_common.c:292-345 defines them (the library's own
    definitions serve where the reference's phy_common.c is not linked in), against the oracle's; and the argument checks of the entry points
    that take a symbol size (no kernel is launched: they fail before any device work)."""
    lib = C.CDLL(HIP_SO)
    lib.srslte_use_standard_symbol_size.argtypes = [C.c_bool]
    lib.srslte_hip_ofdm_create_sz.restype = C.c_void_p
    try:
        for std in (False, True):
            lib.srslte_use_standard_symbol_size(std)
            oracle().orc_use_standard_symbol_size(std)
            for prb in range(0, 112):
                want = oracle().orc_symbol_sz(prb) if 0 < prb <= 110 else -1
                assert lib.srslte_symbol_sz(prb) == want, (std, prb)
                if prb > 0:
                    assert lib.srslte_symbol_sz_power2(prb) == (oracle().orc_symbol_sz_power2(prb) if prb <= 110 else -1), prb
    finally:
        lib.srslte_use_standard_symbol_size(False)
        oracle().orc_use_standard_symbol_size(False)
    assert lib.srslte_symbol_sz(100) == 1536 and lib.srslte_symbol_sz_power2(100) == 2048 and lib.srslte_symbol_sz_power2(25) == 512
    for prb, n in ((100, 1024), (25, 300), (50, 1000), (0, 128), (111, 2048), (6, 64)):  # carriers do not fit / not a size of either family / no such cell
        assert not lib.srslte_hip_ofdm_create_sz(prb, n, 1, 1), (prb, n)
    assert lib.srslte_hip_chest_dl_set_symbol_sz(None, 2048) == -2  # SRSLTE_ERROR_INVALID_INPUTS
