#!/usr/bin/env python3
"""bench.py --grants-mix: the receive chain on a batch of 128 subframes of ONE 20 MHz cell in which every subframe carries a different grant,
through srslte_hip_dl_rx_batch_grants (what srsue's cc_worker does TTI after TTI: srslte_ue_dl_decode_fft_estimate + srslte_pdsch_decode with
the TTI's own srslte_pdsch_grant_t). The mix: 40 % small allocations (2 ... 25 PRB, QPSK / 16QAM, transport blocks of 296
... 6456 bits: block lengths 320 - the unwindowed decoder -, 640 and 704 - the 8-window one -, 960 ... 3904), 30 % medium (25 / 50 PRB), 30 %
large (50 / 75 / 100 PRB 64QAM, up to 13 blocks of 5824); contiguous PRB ranges at drawn offsets, per-grant SNR near the working point of its
MCS. The reference's own CTest matrix walks the same axes one case at a time (phy_dl_test, lib/test/phy/CMakeLists.txt:33-54).

Same contract as the headline line: inputs resident in HBM (--inputs distinct noisy batches in rotation), K steps between barriers repeated
until --min-timed-s, results copied to host inside the timed region, ONE JSON line. roofline.frac = SURVEY 8(d)'s algorithmic decoder work of
the batch (40 K packed lane-instructions per SISO pass and block, summed over the blocks with their own K and pass counts) / step time / the
VALU peak; cpu_baseline = the reference's compiled chain (oracle/_ref) driven grant by grant on the same subframes, else the oracle chain."""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (ROOT, os.path.join(ROOT, "tests")):
    if d not in sys.path:
        sys.path.insert(0, d)

NOF_PRB, CFI, MAX_ITER, AMP, CELL_ID, RNTI = 100, 1, 6, 0.1, 1, 0x1234
VALU_PEAK_LANE = 256 * 4 * 32 * 2.4e9
# (subframes of 128, PRBs, modulation 1 QPSK / 2 16QAM / 3 64QAM, TBS of 36.213 Table 7.1.7.2.1-1 for (I_TBS, N_PRB), I_MCS, SNR dB)
MIX = [
    (8, 2, 1, 296, 9, 6.0), (8, 4, 1, 616, 9, 6.0), (8, 8, 1, 680, 5, 3.0), (8, 6, 1, 936, 9, 6.0), (6, 10, 1, 1544, 9, 6.0),
    (7, 15, 2, 3880, 14, 11.5), (7, 25, 2, 6456, 14, 11.5),                                               # 52 small: 40.6 %
    (13, 25, 3, 9912, 20, 15.0), (13, 50, 2, 15264, 16, 13.0), (12, 50, 3, 22920, 22, 16.5),              # 38 medium
    (13, 100, 3, 75376, 28, 18.5), (13, 75, 3, 45352, 26, 18.0), (12, 50, 3, 27376, 24, 17.5),            # 38 large
]
assert sum(m[0] for m in MIX) == 128


def draw_grants(B, seed):
    """B subframes: (class index, PRB start) in a seeded shuffle of the mix."""
    rng = np.random.default_rng(seed)
    cls = np.concatenate([np.full(m[0], i) for i, m in enumerate(MIX)])
    cls = np.resize(rng.permutation(cls), B)
    return [(int(c), int(rng.integers(0, NOF_PRB - MIX[c][1] + 1))) for c in cls]


def build(B, seed):
    from lte_sim import DlConfig, make_subframe
    rng = np.random.default_rng(seed)
    sub = []
    for b, (c, start) in enumerate(draw_grants(B, seed)):
        n, nprb, mod, tbs, mcs, snr = MIX[c]
        mask = np.zeros((2, NOF_PRB), np.uint8)
        mask[:, start:start + nprb] = 1
        cfg = DlConfig(NOF_PRB, CELL_ID, mod, tbs, cfi=CFI, rnti=RNTI, max_iter=MAX_ITER, prb_mask=mask)
        iq, data = make_subframe(cfg, b, rng, snr_db=None, amp=AMP)
        sub.append({"cls": c, "start": start, "cfg": cfg, "clean": iq, "data": data, "snr": snr,
                    "sigma": float(np.sqrt(AMP * AMP * cfg.nre / cfg.N / 2) * 10 ** (-snr / 20))})
    return sub


def noisy(sub, rng):
    return np.stack([(s["clean"] + (s["sigma"] * (rng.standard_normal(s["clean"].shape) + 1j * rng.standard_normal(s["clean"].shape))).astype(np.complex64))
                     .astype(np.complex64) for s in sub])


# ------------------------------------------------------------------------------------------------ CPU chain (no torch, no GPU)
def cpu_run(sub, iq, lo, hi, seconds, llr8=False):
    """Whole passes over subframes [lo, hi) until `seconds` are used: the reference's srslte_chest_dl_estimate_cfg + srslte_pdsch_decode
    (oracle/_ref) with each subframe's own grant behind the oracle's FFT, else the oracle chain. Returns (n, dt, kind, ok flags, tbs)."""
    import refdrv
    from _libs import OrcOfdm, oracle, p
    from lte_sim import oracle_rx
    kind = "reference" if refdrv.lib() is not None else "port"
    oks, tbs_out = {}, {}
    if kind == "reference":
        O = oracle()
        rx = refdrv.RefDl(NOF_PRB, 1, CELL_ID)
        rx.set_rnti(RNTI)
        rx.set_chest_cfg(filter_type=0, coef=(4.0, 1.0))
        rx.set_pdsch_cfg(max_iterations=MAX_ITER, mmse=True, llr8=llr8)
        q = OrcOfdm()
        assert O.orc_ofdm_init(C.byref(q), NOF_PRB, True) == 0
        grid = np.zeros(14 * 12 * NOF_PRB, np.complex64)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for b in range(lo, hi):
            s = sub[b]
            _, nprb, mod, tbs, mcs, _ = MIX[s["cls"]]
            if kind == "reference":
                g = rx.set_grant_type2(b % 10, CFI, RNTI, mcs, nprb, s["start"])
                info = rx.grant_info()
                assert info["tbs"] == tbs and info["mod"] == mod, (info["tbs"], tbs, info["mod"], mod)
                O.orc_ofdm_rx_sf(C.byref(q), p(iq[b]), p(grid))
                rx.put_grid(grid)
                assert rx.chest() == 0
                crc, _ = rx.decode_pdsch()
                oks[b], tbs_out[b] = bool(crc), rx.payload(tbs // 8)
            else:
                r = oracle_rx(s["cfg"], iq[b], b)  # (16-bit chain; the 8-bit line needs the reference build)
                oks[b], tbs_out[b] = bool(r["ok"]), r["tb"][:tbs // 8]
            n += 1
    return n, time.perf_counter() - t0, kind, oks, tbs_out


def cpu_worker(path, lo, hi, seconds, B, seed, llr8=False):
    sub = build(B, seed)
    iq = np.load(path, mmap_mode="r")
    n, dt, kind, _, _ = cpu_run(sub, np.ascontiguousarray(iq), lo, hi, seconds, llr8)
    print(json.dumps({"n": n, "dt": dt, "kind": kind}))


def main(args):
    B, seed = args.batch, 4242
    sharding = importlib.import_module("srslte-emane_amd.sharding")
    sub = build(B, seed)
    rng = np.random.default_rng(seed + 1)
    iq_host = noisy(sub, rng)
    cpu_multi = None
    if not args.no_cpu:  # all cores first, from a GPU-free parent
        ncores = max(1, min(16, len(os.sched_getaffinity(0))))
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "iq.npy")
            np.save(path, iq_host)
            spans = [sharding.split_contiguous(B, ncores, r) for r in range(ncores)]
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(lo), str(hi), str(args.cpu_seconds), str(B), str(seed), str(int(args.llr8))],
                                      stdout=subprocess.PIPE) for lo, hi in spans if hi > lo]
            outs = [json.loads(p_.communicate(timeout=300 + 10 * args.cpu_seconds)[0].decode().strip().splitlines()[-1]) for p_ in procs]
            cpu_multi = {"value": round(sum(o["n"] / o["dt"] for o in outs), 1), "cores": len(procs), "kind": outs[0]["kind"]}

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    pkg = importlib.import_module("srslte-emane_amd")
    L = pkg.lib()
    tbs_max = max(m[3] for m in MIX)
    hc = pkg.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0
    nstreams, n_inputs = max(1, args.streams), max(1, args.inputs)
    tb_stride = (tbs_max // 8 + 6 + 15) & ~15
    res_bytes, ok_off = sharding.result_layout(tb_stride, B)
    zero_copy = not getattr(args, "no_zero_copy", False)  # the pipelines store into the pinned host record itself (bench.py --no-zero-copy: a device record + a copy)
    t_res = [(torch.zeros(res_bytes, dtype=torch.uint8).pin_memory() if zero_copy else torch.zeros(res_bytes, dtype=torch.uint8, device=dev)) for _ in range(nstreams)]
    rxs = [pkg.DlRx(CELL_ID, NOF_PRB, CFI, RNTI, 3, tbs_max, MAX_ITER, B, True, hc, llr_8bit=args.llr8, out_ptrs=(t_res[s].data_ptr(), t_res[s].data_ptr() + ok_off)) for s in range(nstreams)]
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    streams = [t.cuda_stream for t in tstreams]
    h_out = t_res if zero_copy else [torch.zeros(res_bytes, dtype=torch.uint8).pin_memory() for _ in range(nstreams)]
    grants = (pkg.DlGrant * B)(*[pkg.DlGrant.make(NOF_PRB, s["cfg"].mod, s["cfg"].tbs, RNTI, cfi=CFI, prb_mask=s["cfg"].prb_mask) for s in sub])
    d_iq = torch.from_numpy(iq_host.view(np.float32)).to(dev)
    d_clean = torch.from_numpy(np.stack([s["clean"] for s in sub]).view(np.float32)).to(dev)
    d_sigma = torch.tensor([s["sigma"] for s in sub], dtype=torch.float32, device=dev).reshape(B, 1)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 2)
    d_inputs = [d_iq] + [d_clean + d_sigma * torch.randn(d_clean.shape, generator=gen, device=dev, dtype=torch.float32) for _ in range(n_inputs - 1)]
    del d_clean

    def step(k, src):
        s = k % nstreams
        rc = L.srslte_hip_dl_rx_batch_grants(rxs[s].h, src.data_ptr(), 0, B, grants, rxs[s].d_tb.ptr, rxs[s].tb_stride, rxs[s].d_ok.ptr, streams[s])
        if rc:
            raise RuntimeError("dl_rx_batch_grants failed: %d" % rc)
        if not zero_copy:
            with torch.cuda.stream(tstreams[s]):
                h_out[s].copy_(t_res[s], non_blocking=True)

    issue_s = []

    def repeats(min_s):
        ts, k0 = [], 0
        while not ts or sum(ts) < min_s:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step(k0 + k, d_inputs[(k0 + k) % n_inputs])
            issue_s.append((time.perf_counter() - t0) / args.steps)  # host time to submit a call (descriptors of 128 grants + the launches)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            k0 += args.steps
        return ts

    for k in range(max(args.warmup, nstreams)):
        step(k, d_inputs[k % n_inputs])
    repeats(0.15)
    del issue_s[:]
    times = repeats(args.min_timed_s)
    t_med = float(np.median(times))
    host_issue_ms = float(np.median(issue_s)) * 1e3

    # ---- every input batch once more through instance 0: what was delivered, passes per block, the decoder's algorithmic work
    Cmax = pkg.cbsegm(tbs_max)[1].C
    segs = [pkg.cbsegm(s["cfg"].tbs)[1] for s in sub]
    good = wrong = 0
    per_cls = {i: [0, 0] for i in range(len(MIX))}
    alg_lane, blocks, passes_sum = 0.0, 0, 0
    rec0 = None
    for i in reversed(range(n_inputs)):
        step(0, d_inputs[i])
        torch.cuda.synchronize()
        rec = t_res[0].cpu().numpy()
        ok, tb = rec[ok_off:ok_off + B], rec[:ok_off].reshape(B, tb_stride)
        iters = rxs[0].debug(13, np.uint32, B * Cmax).reshape(B, Cmax)
        for b, s in enumerate(sub):
            nb, same = s["cfg"].tbs // 8, False
            if ok[b]:
                same = np.array_equal(tb[b, :nb], s["data"])
                good, wrong = good + int(same), wrong + int(not same)
            per_cls[s["cls"]][0] += int(bool(ok[b]) and same)
            per_cls[s["cls"]][1] += 1
            it = iters[b, :segs[b].C]
            alg_lane += 40.0 * segs[b].K1 * float(it.sum())
            blocks, passes_sum = blocks + segs[b].C, passes_sum + int(it.sum())
        rec0 = rec
    for k in range(1, nstreams):
        step(k, d_iq)
    torch.cuda.synchronize()
    agree = all(np.array_equal(t_res[s].cpu().numpy(), rec0) for s in range(1, nstreams))
    host_ok = all(np.array_equal(h_out[s].numpy(), rec0) for s in range(nstreams))
    alg_lane /= n_inputs
    ok0, tb0 = rec0[ok_off:ok_off + B], rec0[:ok_off].reshape(B, tb_stride)

    cpu = None
    if not args.no_cpu:
        n1, dt1, kind, coks, ctbs = cpu_run(sub, iq_host, 0, B, args.cpu_seconds, args.llr8)
        both = [b for b in range(B) if ok0[b] and coks.get(b)]
        mism = int(sum(not np.array_equal(ctbs[b], tb0[b, :sub[b]["cfg"].tbs // 8]) for b in both))
        flags = int(sum(bool(ok0[b]) != bool(coks.get(b, False)) for b in range(B) if b in coks))
        cpu = {"value": cpu_multi["value"], "unit": "subframes/s", "cores": cpu_multi["cores"], "kind": kind, "single_core_value": round(n1 / dt1, 2),
               "sample": "one core: %d subframe decodes over the %d benchmark subframes with their own grants, %.1f s; %d processes over disjoint subframes, %.1f s each; "
                         "%s; %d subframes delivered by both CPU and GPU, %d of those differ in a byte; CRC flag differs on %d"
                         % (n1, B, dt1, cpu_multi["cores"], args.cpu_seconds,
                            "reference's compiled srslte_chest_dl_estimate_cfg + srslte_pdsch_decode per grant (oracle/_ref, AVX2) behind the oracle's FFT" if kind == "reference"
                            else "oracle restatement (scalar C)", len(both), mism, flags),
               "tb_mismatches": mism, "crc_flag_mismatches": flags}

    ms_per_step = t_med / args.steps * 1e3
    ks = sorted({(sg.K1, sg.C) for sg in segs})
    alg_achieved = alg_lane / (ms_per_step * 1e-3) / 1e12
    out = {
        "metric": "DL subframes/s (20 MHz, turbo 6-iter)", "value": round(B * args.steps / t_med, 1), "unit": "subframes/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (OFDM/chest/eq) + %s (LLR/turbo)" % ("i8" if args.llr8 else "i16"), "data": "synthetic",
        "config": {"workload": "20 MHz (100 PRB) DL subframe batch=%d, a different grant per subframe (srslte_hip_dl_rx_batch_grants): 40 %% small (2-25 PRB QPSK/16QAM, "
                               "TBS 296-6456), 30 %% medium (25/50 PRB), 30 %% large (50/75/100 PRB 64QAM, TBS up to 75376); OFDM RX + chest_dl + RE lists and scrambling "
                               "sequences from the grants + MMSE + soft demap + ragged rate dematch + one turbo launch per block length + TB CRC + results to host; "
                               "%d distinct input batches in rotation" % (B, n_inputs),
                   "mix": [{"subframes": m[0], "nof_prb": m[1], "mod": m[2], "tbs": m[3], "snr_db": m[5], "K": pkg.cbsegm(m[3])[1].K1, "C": pkg.cbsegm(m[3])[1].C,
                            "delivered": per_cls[i][0], "of": per_cls[i][1]} for i, m in enumerate(MIX)],
                   "block_lengths": [k for k, _ in ks], "decoder_launches_per_step": len({sg.K1 for sg in segs}),
                   "code_blocks_per_step": blocks // n_inputs, "avg_siso_passes_per_cb": round(passes_sum / max(blocks, 1), 3),
                   "bler": round(1 - good / (B * n_inputs), 4), "undetected_errors": wrong, "streams": nstreams, "host_submit_ms_per_step": round(host_issue_ms, 4), "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                   "pipeline_instances_verified": nstreams if agree else 0, "results_on_host_verified": bool(host_ok), "results_to_host": "zero-copy" if zero_copy else "copy", "input_batches": n_inputs,
                   "repeats": len(times), "timed_s": round(sum(times), 3), "repeat_min_value": round(B * args.steps / max(times), 1),
                   "repeat_max_value": round(B * args.steps / min(times), 1)},
        "roofline": {"kernel": "tdec_pair_kernel / tdec_win_kernel<8, 0> / tdec_gen_kernel (one launch per block length)", "bound": "valu",
                     "achieved": round(alg_achieved, 4), "peak": round(VALU_PEAK_LANE / 1e12, 2), "unit": "T lane-instr/s",
                     "frac": round(alg_achieved * 1e12 / VALU_PEAK_LANE, 4), "traffic": None,
                     "definition": "SURVEY 8(d): 40 K packed lane-instructions per SISO pass and block, summed over the batch's blocks (own K, own passes) / step time / peak",
                     "algorithmic_lane_instr_per_step": int(alg_lane)},
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        w = sys.argv[2:]
        cpu_worker(w[0], int(w[1]), int(w[2]), float(w[3]), int(w[4]), int(w[5]), bool(int(w[6])) if len(w) > 6 else False)
    else:
        sys.exit("run through bench.py --grants-mix")
