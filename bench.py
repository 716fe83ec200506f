#!/usr/bin/env python3
"""bench.py — DL 20 MHz subframes/s of the MI355X hot path (OFDM RX -> chest_dl -> soft demap -> turbo decode, max 6
SISO passes with CRC early stop as sch.c:353-383) on synthetic subframes, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.
A "step" is one pass of the whole receive chain over one batch of 128 subframes whose IQ samples are already
resident in HBM. The path shards by subframe/UE with no data-path collective: every rank decodes its
own UE (RNTI 0x1234+rank, cell id 1+rank), so scaling is weak; one all_reduce of the CRC counters after the timed
region does the BLER accounting. Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel (turbo decoder), algorithmic bytes / live HIP-event duration vs the 8 TB/s HBM peak
  kernels       every kernel of the chain timed in isolation (HIP events) with its algorithmic bytes
  cpu_baseline  the same chain on the host CPU, one core: the reference's own compiled code (oracle/_ref) when that
                library travelled with the repo (kind "reference"; its FFT is the oracle's, FFTW being absent), otherwise
                the oracle restatement (kind "port"); timed on a bounded sample of the same subframes
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for d in (ROOT, os.path.join(ROOT, "tests")):
    if d not in sys.path:
        sys.path.insert(0, d)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# 100 PRB, 64QAM MCS 28, TBS 75376 -> 13 x K=5824
NOF_PRB, MOD, TBS, CFI, MAX_ITER, BATCH = 100, 3, 75376, 1, 6, 128


def algorithmic_bytes(pkg_cfg, nof_re_by_sf, ttis):
    """Per-batch algorithmic bytes of each kernel (SURVEY §8d per-unit figures x units per launch)."""
    n = len(ttis)
    N, nre, K, C, Qm = 1536, 1200, 5824, 13, 6
    L = 1 if pkg_cfg.llr8 else 2  # bytes per LLR
    re = sum(nof_re_by_sf[t % 10] for t in ttis)
    return {
        "ofdm_rx": n * (15 * N * 8 + 14 * nre * 8),                       # 318 720 B / subframe
        "chest_dl": n * (4 * nre * 8 + 800 * 8 + 14 * nre * 8),            # 179 200 B / subframe
        "pdsch_demod": re * (16 + L * Qm),                                 # gather y,h + write LLRs
        "rm_rx": re * Qm * L + n * C * (3 * K + 12) * L,                   # read e, write w
        "tdec": n * C * ((3 * K + 12) * L + K // 8),                       # 35 696 B / code block (16-bit LLRs)
        "tb_crc": n * (C * K // 8 + TBS // 8 + 6),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--snr", type=float, default=18.0, help="AWGN SNR in dB (18 dB ~ 10-20 %% BLER for MCS 28)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget (whole passes over the batch)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--llr8", action="store_true", help="8-bit LLR path (SURVEY §8f N2: demod_b, rm_turbo_rx_lut_8bit, avx8 decoder) instead of the 16-bit one")
    ap.add_argument("--streams", type=int, default=4, help="pipeline instances / HIP streams that consecutive steps alternate over")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --force-device 0 rehearses N>1 on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=-1, help="use this GPU for every rank (rehearsal only)")
    ap.add_argument("--stream-batch", type=int, default=2048, help="subframes for the isolated large-batch streaming-kernel timings (0 = skip)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=args.backend)  # "nccl" is RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dev_index = args.force_device if args.force_device >= 0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collective operands live

    pkg = importlib.import_module("srslte-emane_amd")
    sharding = importlib.import_module("srslte-emane_amd.sharding")
    L = pkg.lib()
    from lte_sim import DlConfig, RefRx, make_subframe, oracle_rx
    from _libs import ref as ref_lib

    # ---- synthetic input: `batch` subframes of this rank's UE, TTIs 0..batch-1 (sf 0/5 carry PSS/SSS/PBCH holes)
    rng = np.random.default_rng(1000 + rank)
    ue = sharding.ue_for_rank(rank)  # cfg4: UE u on GPU u, distinct RNTI / cell id
    cfg = DlConfig(NOF_PRB, ue["cell_id"], MOD, TBS, cfi=CFI, rnti=ue["rnti"], max_iter=MAX_ITER, llr8=args.llr8)
    B = args.batch
    ttis = list(range(B))
    iq_list, data_list = [], []
    for t in ttis:
        iq, data = make_subframe(cfg, t, rng, snr_db=args.snr, amp=0.1)
        iq_list.append(iq)
        data_list.append(data)
    iq_host = np.stack(iq_list)
    d_iq = torch.from_numpy(iq_host.view(np.float32)).to(dev)  # resident in HBM before the timed region

    hc = pkg.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0  # phy_dl_test.c:587-595
    # Several pipeline instances (--streams, default 4) on as many HIP streams: consecutive steps (independent batches) alternate between them, so the
    # next batch's kernels fill the SIMDs that the previous batch's turbo-decoder tail (blocks needing all 6 passes) leaves idle.
    nstreams = max(1, args.streams)
    rxs = [pkg.DlRx(ue["cell_id"], NOF_PRB, CFI, ue["rnti"], MOD, TBS, MAX_ITER, B, True, hc, llr_8bit=args.llr8) for _ in range(nstreams)]
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    streams = [t.cuda_stream for t in tstreams]
    rx, stream = rxs[0], streams[0]

    def step(k=0):
        for s in range(6):
            rc = rxs[k % nstreams].stage(s, d_iq.data_ptr(), 0, B, streams[k % nstreams])
            if rc:
                raise RuntimeError("stage %d failed: %d" % (s, rc))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(max(args.warmup, nstreams)):
        step(k)
    barrier()
    # HIP events around the dominant kernel, on the stream it is launched on
    ev = [(L.srslte_hip_event_create(), L.srslte_hip_event_create()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        for s in range(6):
            if s == 4:
                L.srslte_hip_event_record(ev[k][0], streams[k % nstreams])
            rc = rxs[k % nstreams].stage(s, d_iq.data_ptr(), 0, B, streams[k % nstreams])
            if s == 4:
                L.srslte_hip_event_record(ev[k][1], streams[k % nstreams])
            if rc:
                raise RuntimeError("stage %d failed: %d" % (s, rc))
    barrier()
    elapsed = time.perf_counter() - t0
    t_max = elapsed
    if world > 1:
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_max = float(t.item())
    tdec_ms = float(np.mean([L.srslte_hip_event_elapsed_ms(a, b) for a, b in ev]))

    # ---- results of the last step: BLER and turbo passes (bookkeeping, outside the timed region)
    ok = rx.d_ok.to_host(np.uint8)[:B]
    tb = rx.d_tb.to_host(np.uint8).reshape(B, rx.tb_stride)
    iters = rx.debug(6, np.uint32, B * 13)
    good = int(sum(bool(ok[b]) and np.array_equal(tb[b, :TBS // 8], data_list[b]) for b in range(B)))
    wrong = int(sum(bool(ok[b]) and not np.array_equal(tb[b, :TBS // 8], data_list[b]) for b in range(B)))
    # the one collective of a run: BLER accounting over all UEs (srslte-emane_amd/sharding.py)
    good_all, wrong_all, n_all, it_all = sharding.reduce_counts([good, wrong, B, int(iters.sum())], dist if world > 1 else None, cdev)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- per-kernel isolation timing (rank 0)
    nof_re = {s: rx.nof_re(s) for s in range(10)}
    alg = algorithmic_bytes(cfg, nof_re, ttis)
    names = ["ofdm_rx", "chest_dl", "pdsch_demod", "rm_rx", "tdec", "tb_crc"]
    kernels = {}
    reps = 10
    for s, name in enumerate(names):
        a, b = L.srslte_hip_event_create(), L.srslte_hip_event_create()
        rx.stage(s, d_iq.data_ptr(), 0, B, stream)
        L.srslte_hip_event_record(a, stream)
        for _ in range(reps):
            rx.stage(s, d_iq.data_ptr(), 0, B, stream)
        L.srslte_hip_event_record(b, stream)
        ms = L.srslte_hip_event_elapsed_ms(a, b) / reps
        gbs = alg[name] / (ms * 1e-3) / 1e9
        kernels[name] = {"ms": round(ms, 4), "algorithmic_MB": round(alg[name] / 1e6, 3), "GBps": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
    torch.cuda.synchronize()

    # ---- streaming kernels on a batch large enough to leave the launch/latency regime ("per streaming kernel
    #      measured in isolation on a large batch"); same kernels, same per-subframe algorithmic bytes
    big = {}
    if args.stream_batch > 0:
        nb = args.stream_batch
        N, nre, Qm = 1536, 1200, 6
        t_iq = torch.randn(nb, 15 * N * 2, device=dev, dtype=torch.float32)
        t_grid = torch.empty(nb, 14 * nre * 2, device=dev, dtype=torch.float32)
        t_ce = torch.empty_like(t_grid)
        t_res = torch.empty(nb, 10, device=dev, dtype=torch.float32)
        t_llr = torch.empty(nb, 14 * nre * Qm, device=dev, dtype=torch.int16)
        ofdm = pkg.Ofdm(NOF_PRB, True, rx=True)
        est = pkg.ChestDl(ue["cell_id"], NOF_PRB)

        def timed(fn, nbytes):
            a, b = L.srslte_hip_event_create(), L.srslte_hip_event_create()
            fn()
            L.srslte_hip_event_record(a, stream)
            for _ in range(5):
                fn()
            L.srslte_hip_event_record(b, stream)
            ms = L.srslte_hip_event_elapsed_ms(a, b) / 5
            gbs = nbytes / (ms * 1e-3) / 1e9
            return {"ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}

        big["ofdm_rx"] = timed(lambda: L.srslte_hip_ofdm_rx_sf_batch(ofdm.h, t_iq.data_ptr(), t_grid.data_ptr(), nb, stream), nb * 318720)
        big["chest_dl"] = timed(lambda: L.srslte_hip_chest_dl_estimate_batch(est.h, ctypes.byref(hc), 0, t_grid.data_ptr(), t_ce.data_ptr(),
                                                                            t_res.data_ptr(), nb, stream), nb * 179200)
        big["demod_soft_s_64qam"] = timed(lambda: L.srslte_hip_demod_soft_demodulate_s_batch(MOD, t_grid.data_ptr(), t_llr.data_ptr(), 14 * nre, nb, stream),
                                          nb * 14 * nre * (8 + 2 * Qm))
        # the two fused glue kernels of the pipeline on the same large batch, through the pipeline's own stages
        rxb = pkg.DlRx(ue["cell_id"], NOF_PRB, CFI, ue["rnti"], MOD, TBS, MAX_ITER, nb, True, hc, llr_8bit=args.llr8)
        algb = algorithmic_bytes(cfg, nof_re, list(range(nb)))
        for s in (0, 1):
            rxb.stage(s, t_iq.data_ptr(), 0, nb, stream)
        big["pdsch_demod"] = timed(lambda: rxb.stage(2, t_iq.data_ptr(), 0, nb, stream), algb["pdsch_demod"])
        big["rm_rx"] = timed(lambda: rxb.stage(3, t_iq.data_ptr(), 0, nb, stream), algb["rm_rx"])
        big["batch"] = nb
        torch.cuda.synchronize()
        rxb.free()
        del t_iq, t_grid, t_ce, t_res, t_llr

    # ---- CPU baseline on a bounded sample of the same subframes, one core
    cpu = None
    if not args.no_cpu:
        have_ref = ref_lib() is not None
        chain = RefRx(cfg) if have_ref else None
        tc = time.perf_counter()
        same, nsf = True, 0
        while time.perf_counter() - tc < args.cpu_seconds:  # bounded sample: whole passes over the batch, ~12 s
            for b in range(B):
                r = chain.run(iq_list[b], ttis[b]) if have_ref else oracle_rx(cfg, iq_list[b], ttis[b])
                same = same and bool(ok[b]) == bool(r["ok"]) and np.array_equal(tb[b, :TBS // 8 + 3], r["tb"])
                nsf += 1
                if time.perf_counter() - tc >= 2.5 * args.cpu_seconds:
                    break
        dt = time.perf_counter() - tc
        cpu = {"value": round(nsf / dt, 2), "unit": "subframes/s", "cores": 1, "kind": "reference" if have_ref else "port",
               "sample": "%d subframe decodes cycling over the %d benchmark subframes, %.1f s; %s; decoded TBs identical to the GPU's: %s" %
                         (nsf, B, dt, "reference's compiled chest_dl/equaliser/demod/rm_turbo/tdec/crc (oracle/_ref, AVX2) + oracle FFT (no FFTW in image)"
                          if have_ref else "oracle restatement (scalar C)", same)}

    ms_per_step = t_max / args.steps * 1e3
    value = world * B * args.steps / t_max
    tdec_alg = alg["tdec"]
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so this is the value of
    # the committed rocprofv3 --pmc passes of this same command (profiles/r01_pmc/final_traffic.json), valid for B=128.
    traffic = None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc", "final_traffic.json")
    if os.path.exists(pmc) and B == 128 and not args.llr8:
        with open(pmc) as f:
            traffic = json.load(f)["kernels"].get("tdec_win_kernel<16, 0>", {}).get("traffic_bytes")
    out = {
        "metric": "DL subframes/s (20 MHz, turbo 6-iter)", "value": round(value, 1), "unit": "subframes/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (OFDM/chest/eq) + %s (LLR/turbo)" % ("i8" if args.llr8 else "i16"), "data": "synthetic",
        "config": {"workload": "20 MHz (100 PRB) DL subframe batch=%d per GPU, 64QAM MCS 28 (TBS 75376, 13 x K=5824), OFDM RX + chest_dl + MMSE + "
                               "soft demap + rate dematch + turbo max 6 SISO passes with CRC early stop + TB CRC" % B,
                   "snr_db": args.snr, "bler": round(1 - good_all / n_all, 4), "undetected_errors": wrong_all,
                   "avg_siso_passes_per_cb": round(it_all / (n_all * 13), 3), "sharding": "one UE per GPU, no data-path collective",
                   "streams": nstreams},
        "roofline": {"kernel": "tdec_win_kernel<32, 1>" if args.llr8 else "tdec_win_kernel<16, 0>", "bound": "hbm", "achieved": round(tdec_alg / (tdec_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(tdec_alg / (tdec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "avg_launch_ms": round(tdec_ms, 4), "algorithmic_bytes_per_launch": tdec_alg,
                     # the events bracket the launch on its own stream: with several streams they include the time its workgroups queue
                     # behind the other streams' kernels (rocprof's kernel duration starts at the first wave). Alone on the device:
                     "avg_launch_ms_alone": kernels["tdec"]["ms"],
                     "note": "serial-trellis integer kernel: not HBM-bound by construction (SURVEY §8d); streaming kernels are in 'kernels'",
                     # what does bound it: VALU issue. Instructions per wave from the committed SQ counters (profiles/r01_pmc/final7_tdec_sq_*,
                     # 65.9 k at 4.23 passes per block; scaled to this run's pass count), one wave per code block, 64 lanes; peak = 256 CUs x
                     # 4 SIMDs x 16 lanes per clock at the 2.4 GHz boost clock. Launches overlap on the three streams, so the per-launch
                     # duration understates the device-wide rate: 'valu_frac_step' uses the whole step time instead.
                     "valu": None if args.llr8 else {
                         "instr_per_wave": int(65950 * (it_all / (n_all * 13)) / 4.23), "waves": B * 13,
                         "lane_instr_per_s": round(65950 * (it_all / (n_all * 13)) / 4.23 * 64 * B * 13 / (tdec_ms * 1e-3) / 1e12, 2),
                         "peak_lane_instr_per_s": round(256 * 4 * 16 * 2.4e9 / 1e12, 2), "unit": "T lane-instr/s",
                         "valu_frac_launch": round(65950 * (it_all / (n_all * 13)) / 4.23 * 64 * B * 13 / (tdec_ms * 1e-3) / (256 * 4 * 16 * 2.4e9), 3),
                         "valu_frac_step": round(65950 * (it_all / (n_all * 13)) / 4.23 * 64 * B * 13 / (ms_per_step * 1e-3) / (256 * 4 * 16 * 2.4e9), 3)}},
        "kernels": kernels,
        "kernels_large_batch": big,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
