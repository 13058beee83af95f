"""bench.py pieces that run without a GPU: the CPU-baseline leg (oracle on the host cores) and the core-count probe."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_effective_cores_respects_limits():
    import bench

    n = bench.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))


def test_cpu_baseline_leg_schema():
    import bench
    from nutpie_amd.gaussian import ar1_gaussian

    out = bench.cpu_baseline(ar1_gaussian(40), seed=3, target_seconds=0.05)
    assert set(out) == {"value", "unit", "cores", "kind", "sample"}
    assert out["kind"] == "port" and out["unit"] == "leapfrog steps/s"
    assert out["cores"] == bench.effective_cores() and out["value"] > 1e4
    assert "tune 400 + draws 100" in out["sample"]


def test_bench_refuses_to_run_without_a_gpu():
    import subprocess

    import torch

    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
