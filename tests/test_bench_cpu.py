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
    assert 0.9 < bpl / (40 * 10000) < 1.3 and src.startswith("profiles/")          # <= 1.3x algorithmic
    assert bench.measured_traffic(777, 1, None) == (None, None)
    assert bench.kernel_name(1000, 1) == "k_advance<fused,W=1,NV=8>" and "lean" in bench.kernel_name(10000, 4)


def test_roofline_object_is_a_fraction_of_the_binding_resource():
    """`frac` must be a fraction of something.  The register-resident kernel keeps the state on chip (it moves
    about half of the 40 * D "stream everything" bytes) and is bound by instruction issue with one wave per SIMD; the lean kernel
    at D = 10 000 is bound by HBM.  Recomputed here from the committed PMC summaries at round 2's measured kernel times."""
    import bench

    r = bench.roofline(1000, 1, 1024, 1024 * 2048, 9.866e-3)       # BENCH_r02: 9.866 ms per launch of 2 097 152 leapfrogs
    assert r["bound"] == "issue" and r["unit"] == "G wave-instructions/s"
    assert 0.3 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["peak"] == 1024 * 2.4e9 / 4 / 1e9
    assert 0.3 < r["hbm_measured"]["frac_of_peak"] < 1.0 and r["hbm_measured"]["over_algorithmic"] < 1.0
    assert r["stream_equivalent"]["over_hbm_peak"] > 1.0          # the figure round 2 reported as `frac`: not a fraction
    assert r["traffic"] == r["hbm_measured"]["bytes_per_leapfrog"] * 1024 * 2048
    assert abs(r["issue"]["insts_per_leapfrog"]["total"] - sum(v for k, v in r["issue"]["insts_per_leapfrog"].items() if k != "total")) < 1e-6
    h = bench.roofline(10000, 4, 1024, 1024 * 512, 37.02e-3)       # r2_bench_d10000_final: 37.02 ms per launch of 524 288 leapfrogs
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and 0.5 < h["frac"] <= 1.0
    assert h["frac"] == h["hbm_measured"]["frac_of_peak"] and 0.9 < h["hbm_measured"]["over_algorithmic"] < 1.3
    u = bench.roofline(777, 1, 1024, 1024 * 2048, 5e-3)            # no PMC summary for this kernel: labelled, and capped
    assert u["frac"] <= 1.0 and u["traffic"] is None and "no PMC summary" in u["note"]


def test_bench_refuses_to_run_without_a_gpu():
    import subprocess

    import torch

    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
