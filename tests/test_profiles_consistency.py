"""bench.py prices the turbo decoder's roofline with counters taken by scripts/profile_r03.sh on a given decoder source (tdec.hip +
tdec_pair.inc) and drops them when that source has changed since (the figures would describe another kernel): this test fails first, so
that the passes are re-taken before a round ends with `roofline.frac` null."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decoder_counters_belong_to_the_current_source():
    c = json.load(open(os.path.join(ROOT, "profiles", "r03", "tdec_counters.json")))
    h = hashlib.sha256()
    for fn in ("tdec.hip", "tdec_pair.inc"):
        h.update(open(os.path.join(ROOT, "srslte-emane_amd", "csrc", fn), "rb").read())
    assert c["tdec_src_sha"] == h.hexdigest()[:16], "re-run scripts/profile_r03.sh on the GPU box and scripts/profile_r03_post.py (see scripts/README.md)"
    assert c["kernel"] == "tdec_pair_kernel" and c["waves_per_launch"] > 0 and c["valu_instr_per_wave_per_pass"] > 0
    u = json.load(open(os.path.join(ROOT, "profiles", "r02", "ubench_issue.json")))
    assert u  # the measured packed-int16 issue rate bench.py reports beside the guide's peak


def test_streaming_kernel_fractions_have_a_rocprof_record():
    """every HBM fraction bench.py prints for the streaming kernels (batch 128 and batch 2048) can be recomputed from profiles/r03/kernels_by_grid.json"""
    k = json.load(open(os.path.join(ROOT, "profiles", "r03", "kernels_by_grid.json")))
    names = {r["kernel"] for r in k["rows"]}
    for want in ("ofdm_rx_kernel", "chest_dl", "pdsch_demod_kernel", "rm_rx_lds_kernel", "tdec_pair_kernel"):
        assert any(want in n for n in names), (want, sorted(names))
    grids = {}
    for r in k["rows"]:
        grids.setdefault(r["kernel"], set()).add(r["grid_size"])
    assert any(len(g) >= 2 for n, g in grids.items() if "ofdm_rx_kernel" in n), "batch-128 and batch-2048 launches of the OFDM demodulator"
    assert all(r["rocprof_avg_us_default"] > 0 for r in k["rows"])
