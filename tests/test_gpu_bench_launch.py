"""bench.py's N > 1 path on a one-GPU box: the RCCL (nccl backend) branch with a one-rank group on device tensors, and a
plain `python bench.py --gpus 2` that starts its two ranks itself (gloo collectives, both ranks on device 0: a rehearsal of the code path, not a
scaling figure)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SMALL = ["--no-cpu", "--no-full", "--stream-batch", "0", "--steps", "4", "--warmup", "2", "--min-timed-s", "0.05", "--inputs", "3", "--streams", "2"]


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + SMALL, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, rank 0's
    return json.loads(lines[0])


def test_rccl_branch_with_a_one_rank_group():
    d = _run(["--gpus", "1", "--force-dist", "--backend", "nccl"])
    c = d["config"]
    assert d["n_gpus"] == 1 and "RCCL" in c["sharding"] and c["results_on_host_verified"] is True
    assert c["undetected_errors"] == 0 and c["pipeline_instances_verified"] == c["streams"] == 2 and d["value"] > 0


def test_plain_gpus_2_starts_two_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--force-device", "0"])
    c = d["config"]
    assert d["n_gpus"] == 2 and "gloo" in c["sharding"] and c["results_on_host_verified"] is True
    assert c["undetected_errors"] == 0 and c["pipeline_instances_verified"] == 2 and d["scaling"] == "weak"
    assert d["cpu_baseline"] is None  # a rank-0, N = 1 figure
