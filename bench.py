#!/usr/bin/env python3
"""bench.py — DL 20 MHz subframes/s of the MI355X hot path (OFDM RX -> chest_dl -> soft demap -> turbo decode, max 6
SISO passes with CRC early stop as sch.c:353-383) on synthetic subframes, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; a plain
`python bench.py --gpus N` starts those N ranks itself as child processes (before anything here touches a GPU) and relays rank 0's
line. A world size that differs from --gpus is an error, never a silently smaller run.
A "step" is one pass of the whole receive chain over one batch of 128 subframes whose IQ samples are already resident in
HBM - consecutive steps take DIFFERENT batches: --inputs (default 16) distinct 23.6 MB batches rotate through the timed loop, 377 MB,
more than the 256 MB Infinity Cache, so no step finds its input on the die (`config.same_input_value` is the same loop fed one batch,
round 3's figure) - INCLUDING the hand-over of the results: the decoded transport blocks + CRC flags of the batch are copied to host
memory (N = 1) or gathered to rank 0 with ONE collective per batch (N > 1: RCCL gather over xGMI; the
reference's sf_worker pool handing TBs to the one MAC, srsenb/src/phy/phy.cc:113-148) and copied to rank 0's host memory.
The decode itself shards by UE with no collective: rank r decodes UE r (RNTI 0x1234+r, cell id 1+r), so scaling is weak.
The timed region of the contract (K steps between barriers) is a few milliseconds, so it is REPEATED until >= 0.5 s have
been timed; `value` / `ms_per_step` are the median repeat, the spread is in `config.repeat_*`. Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel (turbo decoder). It is a serial-trellis integer kernel: bound by VALU issue, not by HBM
                (SURVEY §8d). `bound` says so. `achieved` = ALGORITHMIC work of one launch - SURVEY §8d's 80 K int16 operations
                per SISO pass and code block = 40 K packed lane-instructions, x the passes the blocks of the batch needed - / the
                STEP time (launches of different batches overlap on the streams, so a launch's own duration is not a chip figure);
                `frac` = that / the guide's VALU peak (78.6 T lane-instr/s). Named extras in `roofline.issued`: what the kernel
                actually ISSUES (SQ-counter pass, profiles/r04/tdec_counters.json, tagged with the sha of the decoder source) per
                launch duration (`frac_launch`, HIP events on its stream inside the timed region), per step (`frac_step`), alone
                (`frac_alone`), against the packed-int16 issue rate measured on this chip (`frac_vs_measured_issue`), and
                `issued_over_algorithmic`. The HBM figures (algorithmic bytes, PMC traffic) are in `roofline.hbm`;
                `roofline.pipeline_hbm` prices the WHOLE step (decoder + front end, PMC bytes) against 8 TB/s: the pipeline's
                phases add up in time because they share the memory system (profiles/r03/overlap_probe.txt).
  kernels       every kernel of the chain timed in isolation (HIP events) with its algorithmic bytes
  cpu_baseline  the same chain on the host CPU: the reference's own compiled code (oracle/_ref) driven from a C loop
                (oracle/refdrv.c; its FFT is the oracle's, FFTW being absent) on one core and on all cores of the box's
                CPU share; otherwise the oracle restatement (kind "port"); a bounded sample of the same subframes
"""
import argparse
import ctypes
import hashlib
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for d in (ROOT, os.path.join(ROOT, "tests")):
    if d not in sys.path:
        sys.path.insert(0, d)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# 100 PRB, 64QAM MCS 28, TBS 75376 -> 13 x K=5824
NOF_PRB, MOD, MCS, TBS, CFI, MAX_ITER, BATCH = 100, 3, 28, 75376, 1, 6, 128
AMP = 0.1
PROFILE_DIR = os.path.join(ROOT, "profiles", "r04")
VALU_PEAK_LANE = 256 * 4 * 32 * 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-instructions/s (v_fma_f32 class)


def algorithmic_bytes(llr8, nof_re_by_sf, ttis):
    """Per-batch algorithmic bytes of each kernel (SURVEY §8d per-unit figures x units per launch)."""
    n = len(ttis)
    N, nre, K, C, Qm = 1536, 1200, 5824, 13, 6
    L = 1 if llr8 else 2  # bytes per LLR
    re = sum(nof_re_by_sf[t % 10] for t in ttis)
    return {
        "ofdm_rx": n * (15 * N * 8 + 14 * nre * 8),                       # 318 720 B / subframe
        "chest_dl": n * (4 * nre * 8 + 800 * 8 + 14 * nre * 8),            # 179 200 B / subframe
        "pdsch_demod": re * (16 + L * Qm),                                 # gather y,h + write LLRs
        "rm_rx": re * Qm * L + n * C * (3 * K + 12) * L,                   # read e, write w
        "tdec": n * C * ((3 * K + 12) * L + K // 8),                       # 35 696 B / code block (16-bit LLRs)
        "tb_crc": n * (C * K // 8 + TBS // 8 + 6),
    }


def file_sha(*paths):
    h = hashlib.sha256()
    for path in paths:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def tdec_source_sha():
    c = os.path.join(ROOT, "srslte-emane_amd", "csrc")
    return file_sha(os.path.join(c, "tdec.hip"), os.path.join(c, "tdec_pair.inc"))


def fftw_probe(n_fft=1536, per_sf=14, seconds=0.5):
    """if libfftw3f is on this box, time the OFDM demodulator's FFTs with it (per_sf transforms of n_fft points per subframe).
    Returns seconds per subframe, or None when the library is absent (it is in this image)."""
    import ctypes.util
    name = ctypes.util.find_library("fftw3f")
    if not name:
        return None
    try:
        F = ctypes.CDLL(name)
        F.fftwf_plan_dft_1d.restype = ctypes.c_void_p
        F.fftwf_plan_dft_1d.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
        F.fftwf_execute_dft.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        a = np.random.default_rng(0).standard_normal(2 * n_fft * per_sf).astype(np.float32)
        b = np.zeros_like(a)
        plan = F.fftwf_plan_dft_1d(n_fft, a.ctypes.data, b.ctypes.data, -1, 0)  # FFTW_FORWARD, FFTW_MEASURE
        if not plan:
            return None
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for i in range(per_sf):
                F.fftwf_execute_dft(plan, a.ctypes.data + 8 * n_fft * i, b.ctypes.data + 8 * n_fft * i)
            n += 1
        return (time.perf_counter() - t0) / n
    except (OSError, AttributeError):
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------ CPU baseline (no torch, no GPU)
def cpu_chain(cell_id, rnti, llr8):
    """(run(iq [n][sf_len] complex64, tti0) -> (tb [n][TBS/8] bytes, ok [n], seconds, ofdm seconds), kind)."""
    import refdrv
    from _libs import OrcOfdm, oracle
    O = oracle()
    if refdrv.lib() is not None:
        rx = refdrv.RefDl(NOF_PRB, 1, cell_id)
        rx.set_rnti(rnti)
        rx.set_chest_cfg(filter_type=0, coef=(4.0, 1.0))  # phy_dl_test.c:587-595
        rx.set_pdsch_cfg(max_iterations=MAX_ITER, mmse=True, llr8=llr8)
        q = OrcOfdm()
        assert O.orc_ofdm_init(ctypes.byref(q), NOF_PRB, True) == 0
        fn = ctypes.cast(O.orc_ofdm_rx_sf, ctypes.c_void_p)

        def run(iq, tti0):
            n = iq.shape[0]
            tb, ok, t_ofdm = np.zeros((n, TBS // 8), np.uint8), np.zeros(n, np.uint8), ctypes.c_double(0)
            dt = rx.L.refdrv_dl_rx_loop(rx.h, fn, ctypes.byref(q), iq.ctypes.data, iq.shape[1], n, tti0, CFI, rnti, MCS, 0, tb.ctypes.data, TBS // 8,
                                        ok.ctypes.data, ctypes.byref(t_ofdm))
            assert dt >= 0, "reference chain failed"
            return tb, ok, dt, t_ofdm.value
        return run, "reference"
    from lte_sim import DlConfig, oracle_rx
    cfg = DlConfig(NOF_PRB, cell_id, MOD, TBS, cfi=CFI, rnti=rnti, max_iter=MAX_ITER, llr8=llr8)

    def run_port(iq, tti0):
        t0 = time.perf_counter()
        rs = [oracle_rx(cfg, iq[b], tti0 + b) for b in range(iq.shape[0])]
        return np.stack([r["tb"][:TBS // 8] for r in rs]), np.array([r["ok"] for r in rs], np.uint8), time.perf_counter() - t0, 0.0
    return run_port, "port"


def cpu_worker(path, lo, hi, seconds, cell_id, rnti, llr8):
    """One process of the multi-core CPU baseline: whole passes over subframes [lo, hi) of the saved batch until `seconds` are used."""
    iq = np.load(path, mmap_mode="r")
    mine = np.ascontiguousarray(iq[lo:hi])
    run, kind = cpu_chain(cell_id, rnti, llr8)
    n, t_ofdm, t0 = 0, 0.0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        t_ofdm += run(mine, lo)[3]
        n += hi - lo
    print(json.dumps({"n": n, "dt": time.perf_counter() - t0, "t_ofdm": t_ofdm, "kind": kind}))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as CHILD processes (torch.distributed.run, rendezvous on
    127.0.0.1) before this process has touched a GPU, let rank 0's line through, return the launcher's exit code."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--snr", type=float, default=18.0, help="AWGN SNR in dB (18 dB ~ 20 %% BLER for MCS 28)")
    ap.add_argument("--snr-full", type=float, default=14.0, help="SNR of the companion run in which every code block needs all 6 passes")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--min-timed-s", type=float, default=0.5, help="repeat the K-step timed region until this much has been timed")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="CPU-baseline time budget per leg (one core; all cores)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-full", action="store_true", help="skip the all-6-passes companion run")
    ap.add_argument("--llr8", action="store_true", help="8-bit LLR path (SURVEY §8f N2: demod_b, rm_turbo_rx_lut_8bit, avx8 decoder) instead of the 16-bit one")
    ap.add_argument("--inputs", type=int, default=16, help="distinct input batches (same transmission, independent noise) rotating through the timed loop; "
                    "16 x 23.6 MB exceed the 256 MB Infinity Cache")
    ap.add_argument("--streams", type=int, default=8, help="pipeline instances / HIP streams that consecutive steps alternate over (more than 4: with "
                    "GPU_MAX_HW_QUEUES raised to match, unless the environment already sets it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' + --force-device 0 rehearses N>1 on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=-1, help="use this GPU for every rank (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and take the N > 1 code path (gather, all_reduce, barrier) even "
                    "with one rank: the RCCL branch on a one-GPU box")
    ap.add_argument("--stream-batch", type=int, default=2048, help="subframes for the isolated large-batch streaming-kernel timings (0 = skip)")
    ap.add_argument("--grants", action="store_true", help="run the same workload through srslte_hip_dl_rx_batch_grants: one grant per subframe "
                    "(here 128 equal full-band MCS-28 grants), RE lists and scrambling sequences made on the device from the grants on every call")
    ap.add_argument("--no-zero-copy", action="store_true", help="N = 1 writes transport blocks and CRC flags straight into rank 0's pinned host record (d_tb / "
                    "d_tb_ok of the C-ABI may be device-visible host memory; +2.3 %%, profiles/r04/ab_frontend_priority_zero_copy.txt); with this flag: "
                    "a device record that a hipMemcpyAsync brings out after every batch, as N > 1 needs for the gather")
    ap.add_argument("--pool", action="store_true", help="submit the batches through srslte_hip_dl_rx_pool_* (ONE submission call per batch from this one host "
                    "thread; the library owns --streams pipeline objects and their streams) instead of round-robining objects here")
    ap.add_argument("--grants-mix", action="store_true", help="the mixed-grant workload (scripts/bench_grants_mix.py): 128 subframes of one cell, a different grant "
                    "per subframe - 40 %% small allocations (2-25 PRB QPSK/16QAM), 30 %% medium, 30 %% large - through srslte_hip_dl_rx_batch_grants")
    ap.add_argument("--cpu-worker", nargs=6, metavar=("NPY", "LO", "HI", "SECONDS", "CELL", "RNTI"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        w = args.cpu_worker
        return cpu_worker(w[0], int(w[1]), int(w[2]), float(w[3]), int(w[4]), int(w[5]), args.llr8)

    # HIP maps its streams onto FOUR hardware queues unless told otherwise, and kernels of one queue start in order: a fifth pipeline object then
    # queues behind another one's decoder launch (5 streams: 396 k). With a hardware queue per object eight objects beat four by 3.9 % on this
    # line and by 22 % on the mixed-grant one (profiles/r04/ab_hw_queues.txt). The runtime reads the variable when it starts: set before torch
    # is imported, here and (inherited) in the ranks of a self-launched run; an application sets it in its own environment
    if args.streams > 4:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(args.streams, 16)))
    if args.grants_mix:
        if args.gpus != 1:
            raise SystemExit("--grants-mix is a one-GPU side workload")
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        return importlib.import_module("bench_grants_mix").main(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; a run on fewer ranks than asked for would carry the wrong label" % (args.gpus, world))
    sharding = importlib.import_module("srslte-emane_amd.sharding")
    from lte_sim import DlConfig, make_subframe

    # ---- synthetic input: `batch` subframes of this rank's UE, TTIs 0..batch-1 (sf 0/5 carry PSS/SSS/PBCH holes); one clean
    #      transmission, two noise levels: the headline SNR and one at which no code block stops early
    rng = np.random.default_rng(1000 + rank)
    ue = sharding.ue_for_rank(rank)  # cfg4: UE u on GPU u, distinct RNTI / cell id
    cfg = DlConfig(NOF_PRB, ue["cell_id"], MOD, TBS, cfi=CFI, rnti=ue["rnti"], max_iter=MAX_ITER, llr8=args.llr8)
    B = args.batch
    ttis = list(range(B))
    clean, data_list = [], []
    for t in ttis:
        iq, data = make_subframe(cfg, t, rng, snr_db=None, amp=AMP)
        clean.append(iq)
        data_list.append(data)
    clean = np.stack(clean)

    def noisy(snr_db):  # as lte_sim.make_subframe: per-sample noise for the given SNR per resource element
        sigma = np.sqrt(AMP * AMP * cfg.nre / cfg.N / 2) * 10 ** (-snr_db / 20)
        return (clean + (sigma * (rng.standard_normal(clean.shape) + 1j * rng.standard_normal(clean.shape))).astype(np.complex64)).astype(np.complex64)

    iq_host = noisy(args.snr)
    iq_full_host = None if args.no_full else noisy(args.snr_full)

    # ---- CPU baseline, all-cores leg: one process per core of this box's CPU share over disjoint subframes of the batch. Run BEFORE this
    #      process touches the GPU (child processes are started from a GPU-free parent, and 16 busy cores do not disturb the GPU timing)
    cpu_multi = None
    if rank == 0 and world == 1 and not args.no_cpu:  # the CPU baseline is a rank-0, N = 1 figure
        ncores = max(1, min(16, len(os.sched_getaffinity(0))))  # a one-GPU box shares its host: 16 cores per GPU
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "iq.npy")
            np.save(path, iq_host)
            spans = [sharding.split_contiguous(B, ncores, r) for r in range(ncores)]
            t0 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(lo), str(hi), str(args.cpu_seconds),
                                       str(ue["cell_id"]), str(ue["rnti"])] + (["--llr8"] if args.llr8 else []), stdout=subprocess.PIPE) for lo, hi in spans if hi > lo]
            outs = [json.loads(p_.communicate(timeout=120 + 10 * args.cpu_seconds)[0].decode().strip().splitlines()[-1]) for p_ in procs]
            cpu_multi = {"value": round(sum(o["n"] / o["dt"] for o in outs), 1), "cores": len(procs), "wall_s": round(time.perf_counter() - t0, 1),
                         "value_without_fft": round(sum(o["n"] / max(o["dt"] - o.get("t_ofdm", 0.0), 1e-9) for o in outs), 1)}

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dev_index = args.force_device if args.force_device >= 0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    use_dist = world > 1 or args.force_dist  # the collective path (one rank with --force-dist: same calls, one-row gather)
    if use_dist:
        import torch.distributed as dist
        if "RANK" not in os.environ:  # --force-dist from a plain invocation: a one-rank group on the loopback
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":  # RCCL on ROCm; the group is bound to this rank's GPU, so barrier() needs no guess
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    on_device = args.backend == "nccl"
    cdev = dev if on_device else torch.device("cpu")  # where collective operands live

    pkg = importlib.import_module("srslte-emane_amd")
    L = pkg.lib()
    d_iq = torch.from_numpy(iq_host.view(np.float32)).to(dev)  # resident in HBM before the timed region
    d_iq_full = None if iq_full_host is None else torch.from_numpy(iq_full_host.view(np.float32)).to(dev)
    # the batches that rotate through the timed loop: batch 0 is the CPU sample's; the others carry the same transmission with independent
    # noise of the same level, drawn on the device (every batch is its own 23.6 MB of HBM, its own LLRs and its own pass counts)
    n_inputs = max(1, args.inputs)
    sigma = float(np.sqrt(AMP * AMP * cfg.nre / cfg.N / 2) * 10 ** (-args.snr / 20))
    gen = torch.Generator(device=dev)
    gen.manual_seed(77000 + rank)
    d_clean = torch.from_numpy(clean.view(np.float32)).to(dev)
    d_inputs = [d_iq] + [d_clean + sigma * torch.randn(d_clean.shape, generator=gen, device=dev, dtype=torch.float32) for _ in range(n_inputs - 1)]
    del d_clean

    hc = pkg.ChestDlCfg()
    hc.filter_coef[0], hc.filter_coef[1] = 4.0, 1.0  # phy_dl_test.c:587-595
    # Several pipeline instances (--streams, default 4) on as many HIP streams: consecutive steps (independent batches) alternate between them, so the
    # next batch's kernels fill the SIMDs that the previous batch's turbo-decoder tail (blocks needing all 6 passes) leaves idle.
    nstreams = max(1, args.streams)
    tb_stride = (TBS // 8 + 6 + 15) & ~15
    res_bytes, ok_off = sharding.result_layout(tb_stride, B)
    zero_copy = (not args.no_zero_copy) and world == 1 and not args.force_dist and not args.grants
    # this rank's record: TBs, then CRC flags - in HBM, or (--zero-copy) in pinned host memory the kernels write through the same pointer
    t_res = [(torch.zeros(res_bytes, dtype=torch.uint8).pin_memory() if zero_copy else torch.zeros(res_bytes, dtype=torch.uint8, device=dev)) for _ in range(nstreams)]
    rxs = [pkg.DlRx(ue["cell_id"], NOF_PRB, CFI, ue["rnti"], MOD, TBS, MAX_ITER, B, True, hc, llr_8bit=args.llr8,
                    out_ptrs=(t_res[s].data_ptr(), t_res[s].data_ptr() + ok_off)) for s in range(nstreams)]
    tstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    streams = [t.cuda_stream for t in tstreams]
    rx, stream = rxs[0], streams[0]
    pool = None
    if args.pool:  # the library's own pool: its objects replace rxs[] in the timed loop (rxs[0] stays for the isolated kernel timings and the checks)
        if world > 1 or args.grants:
            raise SystemExit("--pool is the single-GPU, fixed-grant submission path")
        L.srslte_hip_dl_rx_pool_create.restype = ctypes.c_void_p
        L.srslte_hip_dl_rx_pool_create.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.srslte_hip_dl_rx_pool_submit.restype = ctypes.c_int64
        L.srslte_hip_dl_rx_pool_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                   ctypes.c_void_p, ctypes.c_void_p]
        L.srslte_hip_dl_rx_pool_wait.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.srslte_hip_dl_rx_pool_destroy.argtypes = [ctypes.c_void_p]
        pool = L.srslte_hip_dl_rx_pool_create(ctypes.byref(rxs[0].cfg), nstreams)
        if not pool:
            raise RuntimeError("srslte_hip_dl_rx_pool_create failed")
        pool_tickets = [None] * nstreams
    # where the results go: rank 0's host memory (pinned), directly (N = 1) or through the gather
    t_gath = [torch.zeros((world, res_bytes), dtype=torch.uint8, device=cdev) for _ in range(nstreams)] if (rank == 0 and use_dist) else None
    h_stage = [torch.zeros(res_bytes, dtype=torch.uint8).pin_memory() for _ in range(nstreams)] if (use_dist and not on_device) else None
    h_out = [torch.zeros((world, res_bytes), dtype=torch.uint8).pin_memory() for _ in range(nstreams)] if rank == 0 else None
    if zero_copy:
        h_out = [t.view(1, -1) for t in t_res]

    grant_arr = None
    if args.grants:
        grant_arr = (pkg.DlGrant * B)(*[pkg.DlGrant.make(NOF_PRB, MOD, TBS, ue["rnti"], cfi=CFI) for _ in range(B)])

    def step(k, src, ev=None, plain=False):
        s = k % nstreams
        if isinstance(src, list):  # rotating inputs: step k takes batch k mod n
            src = src[k % len(src)]
        if pool is not None and ev is None and not plain:  # one call per batch; the results go to the pinned host record on the batch's own stream inside the pool
            t = L.srslte_hip_dl_rx_pool_submit(pool, src.data_ptr(), 0, B, None, t_res[s].data_ptr(), tb_stride, t_res[s].data_ptr() + ok_off, None if zero_copy else h_out[s][0].data_ptr())
            if t < 0:
                raise RuntimeError("pool submit failed: %d" % t)
            pool_tickets[s] = t
            return
        if grant_arr is not None:  # one call runs every stage: no decoder-only duration in this mode (tdec_ms stays None, roofline fields null)
            rc = L.srslte_hip_dl_rx_batch_grants(rxs[s].h, src.data_ptr(), 0, B, grant_arr, rxs[s].d_tb.ptr, rxs[s].tb_stride, rxs[s].d_ok.ptr, streams[s])
            if rc:
                raise RuntimeError("dl_rx_batch_grants failed: %d" % rc)
        for stage in range(6 if grant_arr is None else 0):
            if ev is not None and stage == 4:
                L.srslte_hip_event_record(ev[0], streams[s])
            rc = rxs[s].stage(stage, src.data_ptr(), 0, B, streams[s])
            if ev is not None and stage == 4:
                L.srslte_hip_event_record(ev[1], streams[s])
            if rc:
                raise RuntimeError("stage %d failed: %d" % (stage, rc))
        with torch.cuda.stream(tstreams[s]):
            if zero_copy:
                pass  # the record IS host memory
            elif not use_dist:
                h_out[s][0].copy_(t_res[s], non_blocking=True)
            elif on_device:  # ONE collective per batch, device tensors, ordered after the batch's kernels on this stream
                sharding.gather_results(t_res[s], t_gath[s] if rank == 0 else None, dist)
                if rank == 0:
                    h_out[s].copy_(t_gath[s], non_blocking=True)
            else:  # gloo rehearsal on a one-GPU box: the same gather on host tensors
                h_stage[s].copy_(t_res[s], non_blocking=True)
                tstreams[s].synchronize()
                sharding.gather_results(h_stage[s], t_gath[s] if rank == 0 else None, dist)
                if rank == 0:
                    h_out[s].copy_(t_gath[s])

    def barrier():
        if pool is not None:
            for t in pool_tickets:
                if t is not None and L.srslte_hip_dl_rx_pool_wait(pool, t):
                    raise RuntimeError("pool wait failed")
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    issue_s = []

    def timed_repeats(src, min_s, with_events, nsteps=None):
        """Repeats of the contract's timed region: K steps between barriers. Returns (per-repeat max-over-ranks seconds, mean tdec ms).
        nsteps: a region of another length (the steady-state figure)."""
        nsteps = nsteps or args.steps
        evs = [(L.srslte_hip_event_create(), L.srslte_hip_event_create()) for _ in range(nsteps)] if (with_events and grant_arr is None) else None
        times, tdec = [], []
        k0 = 0  # the rotation over streams and input batches goes on across repeats (nstreams and n_inputs need not divide K)
        while True:
            barrier()
            t0 = time.perf_counter()
            for k in range(nsteps):
                step(k0 + k, src, evs[k] if evs else None)
            issue_s.append((time.perf_counter() - t0) / nsteps)  # host time to submit a step (the region is host-bound when this nears ms_per_step)
            barrier()
            k0 += nsteps
            el = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([el, float(sum(times) + el >= min_s)], device=cdev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el, done = float(t[0].item()), bool(t[1].item() > 0)  # every rank stops on the same repeat
                times.append(el)
            else:
                times.append(el)
                done = sum(times) >= min_s
            if evs:
                tdec.append(float(np.mean([L.srslte_hip_event_elapsed_ms(a, b) for a, b in evs])))
            if done or len(times) >= 2000:
                break
        return times, (float(np.mean(tdec)) if tdec else None)

    for k in range(max(args.warmup, nstreams)):
        step(k, d_inputs)
    # the contract's W warm-up steps are a few milliseconds; clocks and first-touch effects last longer (round 2: the first repeats of the
    # timed region ran at half speed), so whole untimed repeats follow until 0.15 s have passed
    timed_repeats(d_inputs, 0.15, False)
    del issue_s[:]
    times, tdec_ms = timed_repeats(d_inputs, args.min_timed_s, pool is None)  # pool mode: one call per batch, no per-stage events
    t_med = float(np.median(times))
    host_issue_ms = float(np.median(issue_s)) * 1e3
    # What the K-step region costs beyond K steps of a pipeline that never drains: the same loop over 4 K steps per region; the slope between the
    # two region lengths is the steady-state step, the rest of the K-step region is filling and draining the streams around the two barriers.
    long_times, _ = timed_repeats(d_inputs, args.min_timed_s / 2, False, nsteps=4 * args.steps)
    t_long = float(np.median(long_times))
    steady_step = max((t_long - t_med) / (3 * args.steps), 1e-9)
    # round 3's loop for comparison: every step and every pipeline instance fed the SAME batch (it then stays in the Infinity Cache)
    same_times, _ = timed_repeats(d_iq, args.min_timed_s / 2, False) if n_inputs > 1 else (times, None)

    # ---- results of the last steps: every pipeline instance's TBs against what was sent; BLER and turbo passes (outside the timed region)
    def check_instance(s):
        rec = t_res[s].cpu().numpy()
        ok, tb = rec[ok_off:ok_off + B], rec[:ok_off].reshape(B, tb_stride)
        good = int(sum(bool(ok[b]) and np.array_equal(tb[b, :TBS // 8], data_list[b]) for b in range(B)))
        wrong = int(sum(bool(ok[b]) and not np.array_equal(tb[b, :TBS // 8], data_list[b]) for b in range(B)))
        return good, wrong, rec

    # every input batch once more through instance 0 (what was sent is the same in all of them): BLER, undetected errors, SISO passes
    good = wrong = it_sum = 0
    pass_hist, wf_passes = [0] * 7, []
    for i in reversed(range(n_inputs)):  # batch 0 last: its record is the one compared with the CPU chain and across the instances
        step(0, d_inputs[i], plain=True)  # on the object whose per-block pass counts can be read back (a pool keeps its objects to itself)
        barrier()
        g_, w_, rec0 = check_instance(0)
        iters = rx.debug(13 if args.grants else 6, np.uint32, B * 13)
        good, wrong, it_sum = good + g_, wrong + w_, it_sum + int(iters.sum())
        for n in range(7):
            pass_hist[n] += int((iters == n).sum())
        # the decoder runs two code blocks per wavefront, in lockstep: a wavefront runs as many passes as the slower of its two blocks
        if not args.llr8 and iters.size % 2 == 0:
            wf_passes.append(float(np.maximum(iters[0::2], iters[1::2]).mean()))
    passes_per_wavefront = float(np.mean(wf_passes)) if wf_passes else None
    for k in range(1, nstreams):
        step(k, d_iq)
    barrier()
    instances_agree = all(np.array_equal(check_instance(s)[2], rec0) for s in range(1, nstreams))  # same input on every instance: same records
    good_all, wrong_all, n_all, it_all, agree_all = sharding.reduce_counts([good, wrong, B * n_inputs, it_sum, int(instances_agree)], dist if use_dist else None, cdev)
    # rank 0: the host copy of the gathered records holds every rank's record in rank order
    gather_ok = None
    digest = hashlib.sha256(rec0.tobytes()).hexdigest()
    if use_dist:
        digests = [None] * world
        dist.all_gather_object(digests, digest)
        if rank == 0:
            gather_ok = all(hashlib.sha256(h_out[s][r].numpy().tobytes()).hexdigest() == digests[r] for s in range(nstreams) for r in range(world))
    elif rank == 0:
        gather_ok = all(hashlib.sha256(h_out[s][0].numpy().tobytes()).hexdigest() == digest for s in range(nstreams))

    # ---- companion run: every code block needs all 6 passes (no early stop helps)
    full = None
    if d_iq_full is not None:
        for k in range(nstreams):
            step(k, d_iq_full)
        ftimes, _ = timed_repeats(d_iq_full, args.min_timed_s / 2, False)
        barrier()
        it_full = sharding.reduce_counts([int(rx.debug(13 if args.grants else 6, np.uint32, B * 13).sum())], dist if use_dist else None, cdev)[0]
        full = {"snr_db": args.snr_full, "value": round(world * B * args.steps / float(np.median(ftimes)), 1),
                "avg_siso_passes_per_cb": round(it_full / (world * B * 13), 3), "repeats": len(ftimes)}
        for k in range(nstreams):  # back to the headline input for what follows
            step(k, d_iq)
        barrier()

    if rank != 0:
        dist.destroy_process_group()
        return

    ok = rec0[ok_off:ok_off + B]
    tb = rec0[:ok_off].reshape(B, tb_stride)

    # ---- per-kernel isolation timing (rank 0)
    nof_re = {s: rx.nof_re(s) for s in range(10)}
    alg = algorithmic_bytes(args.llr8, nof_re, ttis)
    names = ["ofdm_rx", "chest_dl", "pdsch_demod", "rm_rx", "tdec", "tb_crc"]
    kernels = {}
    reps = 20
    for s, name in enumerate(names):
        a, b = L.srslte_hip_event_create(), L.srslte_hip_event_create()
        rx.stage(s, d_iq.data_ptr(), 0, B, stream)
        L.srslte_hip_event_record(a, stream)
        for _ in range(reps):
            rx.stage(s, d_iq.data_ptr(), 0, B, stream)
        L.srslte_hip_event_record(b, stream)
        ms = L.srslte_hip_event_elapsed_ms(a, b) / reps
        if name == "tb_crc" and ms < 0.002:  # no launch behind this stage: the 16-bit decoder assembles the transport blocks and gives the verdicts itself
            kernels[name] = {"ms": round(ms, 4), "algorithmic_MB": round(alg[name] / 1e6, 3), "GBps": None, "frac_hbm": None, "note": "folded into the decoder's last phase"}
            continue
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
        t_resb = torch.empty(nb, 10, device=dev, dtype=torch.float32)
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
                                                                            t_resb.data_ptr(), nb, stream), nb * 179200)
        big["demod_soft_s_64qam"] = timed(lambda: L.srslte_hip_demod_soft_demodulate_s_batch(MOD, t_grid.data_ptr(), t_llr.data_ptr(), 14 * nre, nb, stream),
                                          nb * 14 * nre * (8 + 2 * Qm))
        # the two fused glue kernels of the pipeline on the same large batch, through the pipeline's own stages
        rxb = pkg.DlRx(ue["cell_id"], NOF_PRB, CFI, ue["rnti"], MOD, TBS, MAX_ITER, nb, True, hc, llr_8bit=args.llr8)
        algb = algorithmic_bytes(args.llr8, nof_re, list(range(nb)))
        for s in (0, 1):
            rxb.stage(s, t_iq.data_ptr(), 0, nb, stream)
        big["pdsch_demod"] = timed(lambda: rxb.stage(2, t_iq.data_ptr(), 0, nb, stream), algb["pdsch_demod"])
        big["rm_rx"] = timed(lambda: rxb.stage(3, t_iq.data_ptr(), 0, nb, stream), algb["rm_rx"])
        big["batch"] = nb
        torch.cuda.synchronize()
        rxb.free()
        del t_iq, t_grid, t_ce, t_resb, t_llr

    # ---- CPU baseline on a bounded sample of the same subframes: one core, then one process per core of this box's CPU share
    cpu = None
    if world == 1 and not args.no_cpu:
        run, kind = cpu_chain(ue["cell_id"], ue["rnti"], args.llr8)
        nsf, dt, t_ofdm = 0, 0.0, 0.0
        while dt < args.cpu_seconds:  # bounded sample: whole passes over the batch
            ctb, cok, d1, d2 = run(iq_host, 0)
            nsf, dt, t_ofdm = nsf + B, dt + d1, t_ofdm + d2
        # agreement with the GPU on this batch: every transport block both sides deliver must be byte-identical; the CRC flag itself can differ
        # on a marginal block (one that passes on the last allowed pass): srslte_pdsch_decode's equaliser multiplies by _mm256_rcp_ps, a 12-bit
        # reciprocal whose value depends on the CPU vendor (mimo/precoding.c:262-290), the device divides exactly, so LLRs differ by an LSB
        both = [b for b in range(B) if ok[b] and cok[b]]
        tb_mismatch = int(sum(not np.array_equal(ctb[b], tb[b, :TBS // 8]) for b in both))
        flag_mismatch = int(np.sum(cok.astype(bool) != ok.astype(bool)))
        multi = cpu_multi
        src = ("reference's compiled srslte_chest_dl_estimate_cfg + srslte_pdsch_decode (oracle/_ref, AVX2; C loop oracle/refdrv.c:refdrv_dl_rx_loop) + "
               "the oracle's FFT (no FFTW in the image: %.0f %% of the time)" % (100 * t_ofdm / dt)) if kind == "reference" else "oracle restatement (scalar C)"
        t_fftw = fftw_probe()
        cpu = {"value": multi["value"], "unit": "subframes/s", "cores": multi["cores"], "kind": kind, "cpu_model": cpu_model(),
               "single_core_value": round(nsf / dt, 2),
               # the same runs with the OFDM demodulator's time taken out (the oracle's FFT stands in for FFTW, which this image lacks): the
               # reference's own code only - srslte_chest_dl_estimate_cfg + srslte_pdsch_decode. A real reference build lies between the two
               "value_without_fft": multi.get("value_without_fft"), "single_core_value_without_fft": round(nsf / max(dt - t_ofdm, 1e-9), 2),
               "fftw": ("absent on this box (ctypes.util.find_library('fftw3f') is None)" if t_fftw is None else
                        {"ofdm_s_per_subframe": round(t_fftw, 7), "single_core_value_with_fftw": round(nsf / (dt - t_ofdm + nsf * t_fftw), 2)}),
               "sample": "one core: %d subframe decodes cycling over the %d benchmark subframes, %.1f s; %d processes over disjoint subframes of the same batch, %.1f s "
                         "each; %s; of the %d subframes %d are delivered by both CPU and GPU, %d of those differ in a byte; CRC flag differs on %d (marginal blocks: "
                         "the reference equaliser's 12-bit _mm256_rcp_ps)" % (nsf, B, dt, multi["cores"], args.cpu_seconds, src, B, len(both), tb_mismatch, flag_mismatch),
               "tb_mismatches": tb_mismatch, "crc_flag_mismatches": flag_mismatch}

    ms_per_step = t_med / args.steps * 1e3
    value = world * B * args.steps / t_med
    tdec_alg = alg["tdec"]
    passes = it_all / (n_all * 13)
    # What bounds the decoder: VALU issue. Instruction count per wave from the committed SQ-counter pass of this decoder source
    # (profiles/r02/tdec_counters.json, tagged with the sha of tdec.hip it was taken on; dropped when the source has changed since);
    # issue rate from the microbenchmark of this chip (profiles/r02/ubench_issue.json: packed-int16 and DPP instructions, the
    # decoder's mix, issue one wave-instruction per ~4.4 cycles per SIMD at saturation - 16 lanes per clock, not the 32 of plain
    # 32-bit VALU ops).
    tdec_sha = tdec_source_sha()
    kernel_name = "tdec_win_kernel<32, 1>" if args.llr8 else "tdec_pair_kernel"
    counters, traffic, traffic_src = None, None, None
    cpath = os.path.join(PROFILE_DIR, "tdec_counters.json")
    if os.path.exists(cpath):
        with open(cpath) as f:
            c = json.load(f)
        if c.get("tdec_src_sha") == tdec_sha and not args.llr8 and B == c.get("batch"):
            counters = c
            traffic, traffic_src = c.get("traffic_bytes_per_launch"), {"file": "profiles/r04/tdec_counters.json", "tdec_src_sha": c["tdec_src_sha"], "head": c.get("head")}
    # the packed-int16 / DPP issue rate measured on this chip (scripts/ubench_issue.hip: one wave-instruction per ~4.2-4.5 cycles per SIMD,
    # 16 lanes per clock: half of the guide's SIMD-32 figure, which plain 32-bit VALU operations reach)
    cyc_per_instr, clock_ghz = 4.46, 2.4
    upath = os.path.join(ROOT, "profiles", "r02", "ubench_issue.json")
    if os.path.exists(upath):
        with open(upath) as f:
            u = json.load(f)
        mix = [r for r in u["results"] if r["kernel"] == "ind_mix" and r["waves_per_simd"] == 8]
        if mix:
            cyc_per_instr, clock_ghz = mix[0]["cycles_per_wave_instr"], u["clock_ghz"]
    peak_packed = 256 * 4 * 64 * clock_ghz * 1e9 / cyc_per_instr
    valu = None
    if counters and tdec_ms:
        waves = counters["waves_per_launch"]
        ipw = counters["valu_instr_per_wave_per_pass"] * passes + counters.get("valu_instr_per_wave_fixed", 0)
        lane = ipw * 64 * waves  # lane-instructions of one launch
        valu = {"instr_per_wave": int(ipw), "waves": int(waves), "lane_instr_per_launch": int(lane),
                "achieved_launch": round(lane / (tdec_ms * 1e-3) / 1e12, 2), "achieved_step": round(lane / (ms_per_step * 1e-3) / 1e12, 2),
                "achieved_alone": round(lane / (kernels["tdec"]["ms"] * 1e-3) / 1e12, 2)}
    frac = (lambda x: round(x * 1e12 / VALU_PEAK_LANE, 4))
    # ~80 K int16 operations per SISO pass and code block; a packed instruction does two per lane, so the fewest lane-instructions
    # one launch can do with are 40 K x passes x blocks. 'issued_over_algorithmic' is what the mapping spends on top (window warm-ups, beta
    # recomputation, cross-lane moves, element-wise phases): 4.3 in round 2, see DESIGN.md for the account of this round's.
    K_cb, n_cb = 5824, 13 * B
    alg_lane = 40.0 * K_cb * passes * n_cb
    algorithmic = {"int16_ops_per_launch": int(2 * alg_lane), "lane_instr_per_launch": int(alg_lane),
                   "issued_over_algorithmic": round(valu["lane_instr_per_launch"] / alg_lane, 2) if valu else None,
                   "frac_step": frac(alg_lane / (ms_per_step * 1e-3) / 1e12), "frac_launch": frac(alg_lane / (tdec_ms * 1e-3) / 1e12) if tdec_ms else None,
                   "frac_alone": frac(alg_lane / (kernels["tdec"]["ms"] * 1e-3) / 1e12),
                   "source": "SURVEY.md 8(d): 80 K int16 ops per SISO pass; K = 5824, %d blocks, %.3f passes" % (n_cb, passes)}
    hbm = {"achieved": round(tdec_alg / (tdec_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(tdec_alg / (tdec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
           "frac_alone": round(tdec_alg / (kernels["tdec"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
           "algorithmic_bytes_per_launch": tdec_alg} if tdec_ms else None
    # The contract's figure: ALGORITHMIC work of one launch / the step time / the guide's peak (launches of different batches overlap on the
    # streams, so a launch's own duration is the time it SHARES the chip, not a chip figure). What the kernel actually issues - more than the
    # algorithmic minimum: window warm-ups, beta recomputation, cross-lane moves, element-wise phases - is priced in `issued`.
    issued = {"achieved_launch": valu["achieved_launch"] if valu else None,
              "frac_launch": frac(valu["achieved_launch"]) if valu else None,
              "frac_step": frac(valu["achieved_step"]) if valu else None, "frac_alone": frac(valu["achieved_alone"]) if valu else None,
              "frac_vs_measured_issue": round(valu["achieved_launch"] * 1e12 / peak_packed, 4) if valu else None,
              "frac_step_vs_measured_issue": round(valu["achieved_step"] * 1e12 / peak_packed, 4) if valu else None,
              "issued_over_algorithmic": algorithmic["issued_over_algorithmic"], "valu": valu, "counters_source": traffic_src,
              "measured_issue_source": "profiles/r02/ubench_issue.json, %.2f cycles per packed-int16 / DPP wave-instruction per SIMD (%.1f T lane-instr/s)"
                                       % (cyc_per_instr, peak_packed / 1e12)}
    alg_achieved = alg_lane / (ms_per_step * 1e-3) / 1e12
    roofline = {"kernel": kernel_name, "bound": "valu",
                "achieved": round(alg_achieved, 3), "peak": round(VALU_PEAK_LANE / 1e12, 2), "unit": "T lane-instr/s",
                "frac": frac(alg_achieved),
                "definition": "SURVEY 8(d): 80 K int16 ops = 40 K packed lane-instructions per SISO pass and block, x passes x blocks of one launch / step time / peak",
                "peak_source": "MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz",
                "traffic": traffic, "algorithmic": algorithmic, "issued": issued,
                "avg_launch_ms": round(tdec_ms, 4) if tdec_ms else None, "avg_launch_ms_alone": kernels["tdec"]["ms"], "hbm": hbm,
                "note": "serial-trellis integer kernel: VALU-issue bound, not HBM-bound (SURVEY 8d); the HBM-bound streaming kernels are in 'kernels' / 'kernels_large_batch'"}
    # The pipeline as a whole against the HBM: bytes per step by the committed PMC passes (decoder: tdec_counters.json; the five front-end
    # kernels' batch-128 launches: kernels_by_grid.json, same rocprofv3 runs) / the measured step time. profiles/r03/overlap_probe.txt shows
    # why this is the figure to read: decoder-only and front-end-only steady states ADD up to the pipeline's step time - what they share is
    # the memory system.
    pipeline_hbm = None
    kpath = os.path.join(PROFILE_DIR, "kernels_by_grid.json")
    if traffic and os.path.exists(kpath) and not args.llr8 and not args.grants:
        with open(kpath) as f:
            rows = json.load(f).get("rows", [])
        fe = {}
        for r_ in rows:  # the launches of the timed loop: several hundred per grid size (the batch-2048 launches number a handful)
            if r_.get("launches", 0) >= 100 and r_.get("traffic_MB") and r_["kernel"] != kernel_name and not r_["kernel"].startswith("__amd"):
                fe[r_["kernel"]] = r_["traffic_MB"]
        if fe:
            per_step = traffic + sum(fe.values()) * 1e6
            pipeline_hbm = {"bound": "hbm", "traffic_bytes_per_step": int(per_step), "decoder_bytes": int(traffic), "front_end_MB": {k: round(v, 1) for k, v in fe.items()},
                            "achieved": round(per_step / (ms_per_step * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(per_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "source": "profiles/r04/tdec_counters.json + kernels_by_grid.json (2 x FETCH_SIZE + WRITE_SIZE per launch; fabric-side "
                                      "counters: Infinity-Cache hits are included, rocprofv3 on gfx950 lists no counter that separates DRAM), batch %d" % B}
    roofline["pipeline_hbm"] = pipeline_hbm
    for v in (roofline["frac"], issued["frac_launch"], issued["frac_step"], issued["frac_alone"], hbm["frac"] if hbm else None, pipeline_hbm["frac"] if pipeline_hbm else None):
        assert v is None or 0 <= v <= 1, "a roofline fraction above 1 is a measurement error"
    out = {
        "metric": "DL subframes/s (20 MHz, turbo 6-iter)", "value": round(value, 1), "unit": "subframes/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (OFDM/chest/eq) + %s (LLR/turbo)" % ("i8" if args.llr8 else "i16"), "data": "synthetic",
        "config": {"workload": "20 MHz (100 PRB) DL subframe batch=%d per GPU, 64QAM MCS 28 (TBS 75376, 13 x K=5824), OFDM RX + chest_dl + MMSE + "
                               "soft demap + rate dematch + turbo max 6 SISO passes with CRC early stop + TB CRC + results to rank 0's host memory; "
                               "%d distinct input batches in rotation" % (B, n_inputs),
                   "snr_db": args.snr, "bler": round(1 - good_all / n_all, 4), "undetected_errors": wrong_all,
                   "avg_siso_passes_per_cb": round(passes, 3),
                   "siso_passes_histogram_0_to_6": pass_hist,
                   "avg_siso_passes_per_wavefront": round(passes_per_wavefront, 3) if passes_per_wavefront is not None else None,
                   "sharding": "one UE per GPU; one gather of TBs + CRC flags per batch to rank 0 (%s), inside the timed region" % ("RCCL" if on_device else args.backend)
                   if use_dist else ("one UE per GPU; single GPU: the pipelines write the results into pinned host memory (zero-copy) inside the timed region" if zero_copy
                                     else "one UE per GPU; single GPU: results copied to host inside the timed region"),
                   "results_to_host": "zero-copy" if zero_copy else ("gather + copy" if use_dist else "copy"),
                   "entry_point": "srslte_hip_dl_rx_batch_grants (a grant per subframe)" if args.grants else
                   ("srslte_hip_dl_rx_pool_submit (one call per batch, %d objects inside the library)" % nstreams if pool is not None else "srslte_hip_dl_rx_stage x 6 (one fixed grant)"),
                   "input_batches": n_inputs, "input_MB": round(n_inputs * d_iq.numel() * 4 / 1e6, 1),
                   "same_input_value": round(world * B * args.steps / float(np.median(same_times)), 1),
                   "host_submit_ms_per_step": round(host_issue_ms, 4),
                   "streams": nstreams, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "pipeline_instances_verified": nstreams if agree_all == world else 0, "results_on_host_verified": gather_ok,
                   "repeats": len(times), "timed_s": round(sum(times), 3), "repeat_min_value": round(world * B * args.steps / max(times), 1),
                   "repeat_max_value": round(world * B * args.steps / min(times), 1), "full_iter": full,
                   "full_iter_value": full["value"] if full else None},
        "steady_state": {"value": round(world * B / steady_step, 1), "ms_per_step": round(steady_step * 1e3, 4),
                         "fill_drain_ms_per_region": round((t_med - args.steps * steady_step) * 1e3, 4), "long_region_steps": 4 * args.steps,
                         "long_region_value": round(world * B * 4 * args.steps / t_long, 1),
                         "method": "median region of K steps against median region of 4 K steps (same loop, same barriers): slope = a step of a pipeline that "
                                   "never drains, intercept = filling and draining the streams around the barriers; `value` above is the K-step region as the contract times it"},
        "roofline": roofline,
        "kernels": kernels,
        "kernels_large_batch": big,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if pool is not None:
        L.srslte_hip_dl_rx_pool_destroy(pool)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
