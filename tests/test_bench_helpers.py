"""Host logic of bench.py that decides what the cpu_baseline leg reports (no GPU): the cgroup CPU quota parser, the core count,
and one tiny run of the native baseline itself (oracle/frontend.cc through bench.cpu_baseline) -- test infrastructure testing
test infrastructure, so that a box with a CPU quota (the GPU box grants 16 of 256 CPUs) is reported as what it is."""
import builtins
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fake_open(files):
    real = builtins.open

    def f(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise FileNotFoundError(path)
            return io.StringIO(files[path])
        if str(path).startswith("/sys/fs/cgroup"):
            raise FileNotFoundError(path)
        return real(path, *a, **k)
    return f


def test_cpu_quota_parsing(monkeypatch):
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert bench.cpu_quota() == 16.0
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert bench.cpu_quota() is None
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "250000\n",
                                                      "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.cpu_quota() == 2.5
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n",
                                                      "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.cpu_quota() is None
    monkeypatch.setattr(builtins, "open", _fake_open({}))
    assert bench.cpu_quota() is None


def test_physical_cores():
    n = bench.physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_cpu_baseline_runs_natively(oracle, synth):
    """A few frames through every leg: the line names the threads it used, the CPU seconds per wall second that ran them, and a
    parallel efficiency per leg; `value` is the best leg."""
    V = bench._util._load("plslam_amd_vocab", os.path.join(ROOT, "pl-slam_amd", "vocab.py"))
    voc = V.Vocabulary.synthetic(102, k=4, L=3, synth=synth, idf=True)   # (a small tree: the test is about the legs, not the words)
    frames = synth.make_frames(2, 4, 480, 640, unique=4)
    r = bench.cpu_baseline(oracle, V, frames, voc, 1000, 8, 200, bench.TUM1_K, bench.TUM1_D, budget_s=0.2)
    assert r["kind"] == "port" and r["unit"] == "frames/s" and r["value"] > 0
    assert r["legs"]["1"]["threads"] == 1 and r["legs"]["1"]["frames_per_s"] > 0
    assert r["cores"] in [leg["threads"] for leg in r["legs"].values()]
    assert abs(r["value"] - max(leg["frames_per_s"] for leg in r["legs"].values())) < 0.011   # (rounded to 2 digits)
    for name, leg in r["legs"].items():
        if name != "1":
            assert leg["cores_busy"] > 0 and 0 < leg["parallel_efficiency"] <= 1.5 and leg["frames"] == leg["threads"] * (leg["frames"] // leg["threads"])
