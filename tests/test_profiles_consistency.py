"""bench.py prices the turbo decoder's roofline with counters taken by scripts/profile_r02.sh on a given tdec.hip and drops them when that
source has changed since (the figures would describe another kernel): this test fails first, so that the passes are re-taken before a
round ends with `roofline.frac` null."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decoder_counters_belong_to_the_current_source():
    c = json.load(open(os.path.join(ROOT, "profiles", "r02", "tdec_counters.json")))
    sha = hashlib.sha256(open(os.path.join(ROOT, "srslte-emane_amd", "csrc", "tdec.hip"), "rb").read()).hexdigest()[:16]
    assert c["tdec_hip_sha"] == sha, "re-run scripts/profile_r02.sh on the GPU box and scripts/tdec_counters.py (see scripts/README.md)"
    u = json.load(open(os.path.join(ROOT, "profiles", "r02", "ubench_issue.json")))
    assert u  # the measured issue peak bench.py divides by
