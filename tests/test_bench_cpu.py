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

    tuned, out = bench.cpu_baseline(ar1_gaussian(40), seed=3, target_seconds=0.05)
    assert set(out) == {"value", "unit", "cores", "kind", "sample"}
    assert out["kind"] == "port" and out["unit"] == "leapfrog steps/s"
    assert out["cores"] == bench.effective_cores() and out["value"] > 1e4
    assert "tune 400 + draws 100" in out["sample"]
    # the second, honest CPU number: the same sampler built for speed (never a checker)
    assert set(tuned) == {"value", "unit", "cores", "kind", "sample"} and tuned["kind"] == "tuned"
    assert tuned["value"] > out["value"]


def test_traffic_table_matches_the_committed_pmc_summaries():
    """roofline.traffic comes from profiles/traffic.json, which profiles/make_traffic.py derives from the committed PMC
    summaries: (2 x FETCH_SIZE + WRITE_SIZE) KB per launch / leapfrogs per launch."""
    import json
    import re

    import bench

    table = json.load(open(bench.TRAFFIC_JSON))
    assert "1000:1" in table and "10000:4" in table
    for key, e in table.items():
        txt = open(os.path.join(ROOT, e["source"])).read()
        f = float(re.search(r"FETCH_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt).group(1))
        w = float(re.search(r"WRITE_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt).group(1))
        assert abs((2 * f + w) * 1024 / e["leapfrogs_per_launch"] - e["bytes_per_leapfrog"]) < 1.0, key
    bpl, src = bench.measured_traffic(10000, 4, None)
    assert 1.0 < bpl / (40 * 10000) < 1.3 and src.startswith("profiles/")          # <= 1.3x algorithmic
    assert bench.measured_traffic(777, 1, None) == (None, None)
    assert bench.kernel_name(1000, 1) == "k_advance<fused,W=1,NV=8>" and "lean" in bench.kernel_name(10000, 4)


def test_bench_refuses_to_run_without_a_gpu():
    import subprocess

    import torch

    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
