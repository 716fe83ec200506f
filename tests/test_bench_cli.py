"""bench.py's launch contract: `--gpus N` means N. The CPU-side half: a world size that differs from --gpus is
an error before any work is done, and a plain `--gpus N > 1` invocation builds the torch.distributed.run command the driver would have used."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_world_size_must_equal_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr, (r.returncode, r.stderr[-400:])
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in r.stderr, (r.returncode, r.stderr[-400:])


def test_plain_invocation_with_gpus_n_starts_n_child_ranks(monkeypatch):
    m = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(m.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("RANK", raising=False)
    try:
        m.main()
        raise AssertionError("main() must exit with the launcher's code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]  # the ranks get the caller's own arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is not None
