"""Compile-time resource check of every device kernel (hipcc cross-compiles gfx950 without a GPU): none may use scratch memory - a kernel
that spills registers or indexes a private array dynamically silently runs several times slower (it happened: writing to a by-value
argument struct moved it to scratch). Also pins the occupancy the pipelines count on for the turbo decoder."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "srslte-emane_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FILES = ["fft.hip", "chest.hip", "demod.hip", "tdec.hip", "tdec_mix.hip", "tcod.hip", "pdsch.hip"]


def _remarks(name):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "--cuda-device-only", "-c", os.path.join(CSRC, name), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
        for key in ("ScratchSize [bytes/lane]", "VGPRs", "AGPRs", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_kernel_uses_scratch_and_decoder_occupancy():
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = dict(zip(FILES, ex.map(_remarks, FILES)))
    total = 0
    for f, kernels in res.items():
        assert kernels, f
        for k, r in kernels.items():
            total += 1
            if "tdec_pair_kernel" in k or "tdec_mix_kernel" in k:  # the mixed launch of ragged batches runs the pair kernel's body under the same budget
                # the second measured exception: a 184-register budget (the kernel needs 194) leaves room for TWO front-end wavefronts of the other
                # streams beside two decoder wavefronts on a SIMD and is 2.4 % faster in the pipeline (profiles/r04/ab_tdec_bytes.txt, table 6);
                # the 20 bytes of scratch sit outside the sweeps. Bounded: spills in the loops would show up as more
                assert r.get("ScratchSize [bytes/lane]", 0) <= 32 and r["VGPRs"] <= 184, (f, k, r)
                continue
            if "tdec_ar16_kernel" in k:
                # the sse8 twin of the exception below (two blocks per wavefront around the same sweeps, same 216-register budget): 64 bytes of
                # scratch outside the sweeps
                assert r.get("ScratchSize [bytes/lane]", 0) <= 128 and r["VGPRs"] <= 216, (f, k, r)
                continue
            if "tdec_ar32_kernel" in k:
                # the one measured exception: the 8-bit avx8 kernel under a 216-register budget spills around its loops (extraction, exchange and
                # decision phases keep their state there while the pair-mapped sweeps run) and is 3 % FASTER in the four-stream pipeline than
                # without the budget (256 VGPRs, 92 B of scratch): profiles/r04/ab_llr8_pair.txt. Bounded, so that a change that spills inside
                # the sweeps shows up
                assert r.get("ScratchSize [bytes/lane]", 0) <= 256 and r["VGPRs"] <= 216, (f, k, r)
                continue
            assert r.get("ScratchSize [bytes/lane]", 0) == 0 and r.get("VGPRs Spill", 0) == 0, (f, k, r)
    assert total >= 40
    dec = [r for k, r in res["tdec.hip"].items() if "tdec_pair_kernel" in k]
    assert len(dec) == 1 and dec[0]["Occupancy [waves/SIMD]"] == 2 and dec[0]["AGPRs"] == 0 and dec[0]["LDS Size [bytes/block]"] <= 13 * 1024 + 256, dec
