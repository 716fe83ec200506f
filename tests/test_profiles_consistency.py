"""bench.py prices the turbo decoder's roofline with counters taken by scripts/profile_r04.sh on a given decoder source (tdec.hip +
tdec_pair.inc) and drops them when that source has changed since (the figures would describe another kernel): this test fails first, so
that the passes are re-taken before a round ends with `roofline.frac` null."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decoder_counters_belong_to_the_current_source():
    c = json.load(open(os.path.join(ROOT, "profiles", "r04", "tdec_counters.json")))
    h = hashlib.sha256()
    for fn in ("tdec.hip", "tdec_pair.inc"):
        h.update(open(os.path.join(ROOT, "srslte-emane_amd", "csrc", fn), "rb").read())
    assert c["tdec_src_sha"] == h.hexdigest()[:16], "re-run scripts/profile_r04.sh on the GPU box and scripts/profile_r04_post.py (see scripts/README.md)"
    assert c["kernel"] == "tdec_pair_kernel" and c["waves_per_launch"] > 0 and c["valu_instr_per_wave_per_pass"] > 0
    u = json.load(open(os.path.join(ROOT, "profiles", "r02", "ubench_issue.json")))
    assert u  # the measured packed-int16 issue rate bench.py reports beside the guide's peak


def test_streaming_kernel_fractions_have_a_rocprof_record():
    """every HBM fraction bench.py prints for the streaming kernels (batch 128 and batch 2048) can be recomputed from profiles/r04/kernels_by_grid.json"""
    k = json.load(open(os.path.join(ROOT, "profiles", "r04", "kernels_by_grid.json")))
    names = {r["kernel"] for r in k["rows"]}
    for want in ("ofdm_rx_kernel", "chest_dl", "pdsch_demod_kernel", "rm_rx_lds_kernel", "tdec_pair_kernel"):
        assert any(want in n for n in names), (want, sorted(names))
    grids = {}
    for r in k["rows"]:
        grids.setdefault(r["kernel"], set()).add(r["grid_size"])
    assert any(len(g) >= 2 for n, g in grids.items() if "ofdm_rx_kernel" in n), "batch-128 and batch-2048 launches of the OFDM demodulator"
    assert all(r["rocprof_avg_us_default"] > 0 for r in k["rows"])


def test_committed_bench_line_follows_the_contract():
    """The line `python bench.py` printed on the final source (profiles/r04/final_bench.json): the contract's fields, BASELINE.json's metric
    and workload, fractions that can be recomputed from the numbers beside them, and the rocprof record they rest on."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r04", "final_bench.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "subframes/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert "subframes/s" in json.dumps(base) and "batch=128" in d["config"]["workload"] and "model" not in d["config"]
    # whole-job throughput = subframes of K steps / their time
    assert abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    # roofline.frac is SURVEY 8(d)'s ALGORITHMIC work per step: 40 K packed lane-instructions per SISO pass and code block, x the passes the
    # batch's blocks needed, / the step time / the guide's peak - recomputed here from the numbers beside it
    cfg = d["config"]
    alg = 40.0 * 5824 * cfg["avg_siso_passes_per_cb"] * 13 * 128
    assert abs(r["algorithmic"]["lane_instr_per_launch"] - alg) / alg < 2e-3
    assert abs(r["achieved"] - alg / (d["ms_per_step"] * 1e-3) / 1e12) / r["achieved"] < 0.01
    # the named extras: what the kernel ISSUES (SQ counters of this decoder source) per its own launch duration
    iss = r["issued"]
    assert abs(iss["achieved_launch"] - iss["valu"]["lane_instr_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 0.02
    assert iss["issued_over_algorithmic"] > 1 and 0 < iss["frac_launch"] <= 1 and 0 < iss["frac_step"] <= 1
    c = json.load(open(os.path.join(ROOT, "profiles", "r04", "tdec_counters.json")))
    assert iss["counters_source"]["tdec_src_sha"] == c["tdec_src_sha"] and r["traffic"] == c["traffic_bytes_per_launch"]
    # the event-timed launch duration of the line and rocprof's average for the same kernel in the same command agree
    assert abs(r["avg_launch_ms"] * 1e6 - c["rocprof_avg_ns_default"]) / c["rocprof_avg_ns_default"] < 0.1
    p = r["pipeline_hbm"]
    assert p["bound"] == "hbm" and abs(p["frac"] - p["achieved"] / p["peak"]) < 2e-3 and p["frac"] <= 1
    assert abs(p["achieved"] - p["traffic_bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) / p["achieved"] < 0.01
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["unit"] == d["unit"] and cb["cores"] >= 1 and cb["value"] > 0
    assert cfg["undetected_errors"] == 0 and cfg["results_on_host_verified"] is True and cfg["pipeline_instances_verified"] == cfg["streams"]
    # the input batches rotate (more bytes than the Infinity Cache holds); the same loop fed ONE batch stands beside the value
    assert cfg["input_batches"] >= 16 and cfg["input_MB"] > 256 and cfg["same_input_value"] > 0
    assert sum(cfg["siso_passes_histogram_0_to_6"]) == cfg["input_batches"] * 128 * 13 and cfg["avg_siso_passes_per_wavefront"] >= cfg["avg_siso_passes_per_cb"]


def test_mixed_grant_line_is_on_record():
    """a bench line for a realistic TBS mix through the per-subframe-grant entry point, with roofline and cpu_baseline."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r04", "final_bench_grants_mix.json")))
    c = d["config"]
    assert d["unit"] == "subframes/s" and "srslte_hip_dl_rx_batch_grants" in c["workload"] and c["undetected_errors"] == 0
    small = sum(m["subframes"] for m in c["mix"] if m["nof_prb"] <= 25 and m["mod"] <= 2)
    assert small >= 0.4 * 128 and min(m["K"] for m in c["mix"]) <= 400 and any(400 < m["K"] <= 800 for m in c["mix"])  # every decoder back-end in the batch
    assert 0 < d["roofline"]["frac"] <= 1 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("reference", "port")
