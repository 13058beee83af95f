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


def _run_bench(*argv, env_extra=None, timeout=240):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None), lines


STUB = os.path.join(ROOT, "tests", "bench_stub_engine.py")


def test_gpus_n_spawns_its_own_ranks_and_gathers_in_global_chain_order():
    """`python bench.py --gpus 2` with no RANK in the environment starts two ranks itself (torch.distributed.run,
    127.0.0.1), shards the chains by global id, runs the config-5 shard leg with ONE gather to rank 0, and rank 0 prints one line.
    The engine is a stand-in (tests/bench_stub_engine.py, gloo): this pins the plumbing, it measures nothing."""
    r, out, lines = _run_bench("--gpus", "2", "--engine-stub", STUB, "--chains", "8", "--dim", "16", "--config5-dim", "32",
                               "--steps", "3", "--warmup", "1", "--config5-launches", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1                                        # ONE JSON line, from rank 0
    assert out["n_gpus"] == 2 and out["data"] == "stub" and "stub" in out
    ranks = out["ranks"]
    assert ranks["world"] == 2 and ranks["collective_ranks"] == 2 and ranks["backend"] == "gloo" and ranks["launched_by"] == "torch.distributed.run"
    assert [p["rank"] for p in ranks["per_rank"]] == [0, 1] and [p["device"] for p in ranks["per_rank"]] == [0, 1]
    # weak scaling: every rank advanced its 8 chains by 3 launches of the stub's 8 leapfrogs; value = the sum over ranks / max time
    assert all(p["leapfrogs"] == 8 * 3 * 8 for p in ranks["per_rank"])
    assert out["config"]["leapfrogs_per_step"] == 2 * 8 * 8 and out["steps"] == 3
    assert abs(out["value"] - 2 * 8 * 3 * 8 / (out["ms_per_step"] * 3 / 1000)) < 1e-6 * out["value"]
    g = out["config5_shard"]["gather"]
    assert g["collective_ranks"] == 2 and g["backend"] == "gloo"
    assert g["check"]["chains_gathered"] == 16 and g["check"]["moments_shape"] == [16, 32]
    assert g["bytes_gathered"] == 2 * g["bytes_local"] and g["gather_s"] >= 0
    assert "job" not in out and "cpu_baseline" not in out and "other_configs" not in out      # N = 1 legs only


def test_stub_gather_is_in_global_chain_order():
    """The same leg in one process (no torch.distributed.run): a one-rank group is created for the gather."""
    r, out, _ = _run_bench("--engine-stub", STUB, "--chains", "4", "--dim", "16", "--config5-dim", "32", "--steps", "2", "--warmup", "1",
                           "--config5-launches", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["n_gpus"] == 1 and out["ranks"]["launched_by"] == "single process"
    c5 = out["config5_shard"]
    assert c5["gather"]["collective_ranks"] == 1 and c5["gather"]["check"]["chains_gathered"] == 4
    assert c5["gather"]["arrays"]["draws"][0] == 4 and c5["gather"]["arrays"]["draw_mean"] == [4, 32]


def test_gpus_n_refuses_when_fewer_gpus_are_visible():
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    r, out, _ = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "1")
    assert r.returncode != 0 and out is None
    assert "GPU(s) visible" in (r.stderr + r.stdout)


def test_world_size_must_agree_with_gpus():
    r, out, _ = _run_bench("--gpus", "2", "--engine-stub", STUB, env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                                          "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_eight_ranks_through_the_spawn_path_with_config5s_layout_scaled_down():
    """`bench.py --gpus 8` through its own spawn path (eight gloo ranks, stand-in engine): config 5's layout — every rank owns
    a contiguous block of chains of a long row — scaled down; ONE gather per array into ONE buffer on the root (root memory = 1 x payload),
    chains in global order, the line says collective_ranks 8, and only rank 0 prints."""
    r, out, lines = _run_bench("--gpus", "8", "--engine-stub", STUB, "--chains", "6", "--dim", "16", "--config5-dim", "200",
                               "--steps", "2", "--warmup", "1", "--config5-launches", "2", timeout=480)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and out["n_gpus"] == 8
    ranks = out["ranks"]
    assert ranks["world"] == 8 and ranks["collective_ranks"] == 8 and [p["rank"] for p in ranks["per_rank"]] == list(range(8))
    g = out["config5_shard"]["gather"]
    assert g["collective_ranks"] == 8 and g["check"]["chains_gathered"] == 48 and g["check"]["moments_shape"] == [48, 200]
    assert g["bytes_gathered"] == 8 * g["bytes_local"]
    assert g["root_bytes_allocated"] == g["payload_bytes"] == g["bytes_gathered"]          # one receive buffer, no concatenated copy
    assert "cpu_baseline" not in out["config5_shard"] and "other_configs" not in out       # CPU legs: N = 1 only
