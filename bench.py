#!/usr/bin/env python
"""bench.py — leapfrog steps/s (all chains) of the HIP NUTS engine on BASELINE.json's headline workload.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` — for N>1 launched by
``torch.distributed.run`` with one rank per GPU.  A *step* is one pass of the hot path over the whole
batch of chains: one launch of the fused leapfrog/tree kernel that advances every chain of this GPU
by ``evals_per_launch`` leapfrogs (one logp+gradient evaluation each).  W untimed steps, then EXACTLY
K timed steps bracketed by barrier + synchronize on both sides; time = max over ranks; rank 0 prints
ONE JSON line.

Workload (``config.workload``): BASELINE.json configs[1] — 1000-dimensional correlated Gaussian
(AR(1) rho = 0.9 with per-dimension scales exp(N(0,1)) from numpy.random.default_rng(20260926),
analytic logp/grad fused in-kernel), 1024 chains per GPU, adaptation "diag", target_accept 0.8,
maxdepth 10, U(-2,2) initial points.  Inputs are generated on the host once and are resident in HBM
before the timed region.  Multi-GPU is weak scaling: 1024 chains per GPU, chain ids offset per rank,
no collective in the data path (only the timing all-reduce).

Extra objects in the JSON line:
  roofline      — algorithmic bytes (40 * D per leapfrog per chain: read q, p, sigma^2; write q', p';
                  SURVEY.md §8d, fused analytic gradient) / mean k_advance duration from HIP events on the
                  engine's stream, against the 8 TB/s HBM peak.
  cpu_baseline  — the CPU oracle (oracle/, "port": the real nuts-rs cannot be built here) timed on this
                  box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
  job           — the complete sampling job (tune 400 + draws 1000) wall time, total leapfrogs, min bulk
                  ESS over a subset of dimensions and ESS/s (the second half of BASELINE.json's metric).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes per leapfrog of the register-resident kernel at D = 1000, from the PMC passes in profiles/r1_v4_pmc.txt:
# (2 x FETCH_SIZE + WRITE_SIZE) KB per launch / 262144 leapfrogs  (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)
MEASURED_HBM_BYTES_PER_LEAPFROG_D1000 = 21499.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=60)
    p.add_argument("--warmup", type=int, default=60)
    p.add_argument("--dim", type=int, default=1000)
    p.add_argument("--chains", type=int, default=1024, help="chains PER GPU")
    p.add_argument("--evals-per-launch", type=int, default=256)
    p.add_argument("--waves", type=int, default=0)
    p.add_argument("--seed", type=int, default=20260926)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-job", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    return p.parse_args()


def total_leapfrogs(smp):
    return sum(p.total_num_steps for p in smp.progress())


def effective_cores():
    """CPU threads this process may really use: min(cpu_count, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(model, seed, target_seconds):
    """Oracle ("port") on the host cores: one chain per thread, min(chains, cores) threads — the reference's
    `cores` model (python/nutpie/sample.py:856-857, 1061-1070)."""
    import oracle

    oracle.build()
    cores = effective_cores()
    # calibration run, then a bounded sample sized for ~target_seconds of wall time
    s = oracle.default_settings(seed=seed, num_chains=cores, num_tune=60, num_draws=10, n_threads=cores)
    cal = oracle.sample_tridiag(s, model.diag, model.offdiag)
    rate = cal.stats["n_steps"].sum() / max(cal.seconds, 1e-6)
    tune, draws = 400, 100
    per_chain = 110_000  # leapfrogs per chain for tune 400 + draws 100 on this target (measured)
    chains = int(min(1024, max(cores, (rate * target_seconds) // per_chain // cores * cores)))
    s = oracle.default_settings(seed=seed, num_chains=chains, num_tune=tune, num_draws=draws, n_threads=cores)
    tr = oracle.sample_tridiag(s, model.diag, model.offdiag)
    n = int(tr.stats["n_steps"].sum())
    return {
        "value": n / tr.seconds, "unit": "leapfrog steps/s", "cores": cores, "kind": "port",
        "sample": f"CPU oracle (C++ restatement of nuts-rs diag-NUTS, oracle/), {chains} of the workload's chains, "
                  f"tune {tune} + draws {draws}, one chain per thread on {cores} threads (host: {os.cpu_count()} logical CPUs, "
                  f"cgroup/affinity limit {cores}): {n} leapfrogs in {tr.seconds:.2f} s",
    }


def run_job(hip, model, args, device, chain_offset, dims_for_ess):
    """The complete job: tune 400 + draws 1000 on this GPU's chains; returns wall seconds, leapfrogs, ESS."""
    from nutpie_amd.ess import ess_bulk

    s = hip.PyNutsSettings.Diag(args.seed)
    s.update(num_tune=400, num_draws=1000, num_chains=args.chains * args.gpus)
    m = hip.TridiagGaussianModel(model.diag, model.offdiag)
    t0 = time.perf_counter()
    smp = hip.PySampler(s, m, device=device, waves_per_chain=args.waves, chain_offset=chain_offset, n_local_chains=args.chains,
                        evals_per_launch=512)
    t_alloc = time.perf_counter() - t0
    smp.wait()
    secs = smp.seconds
    n_steps = smp._copy("n_steps", np.int64)
    div = smp._copy("diverging", np.bool_)
    depth = smp._copy("depth", np.int64)
    step = smp._copy("step_size", np.float64)
    ess = None
    try:
        from nutpie_amd.distributed import device_tensor

        d = device_tensor(smp.device_ptr("draws"), (args.chains, 1400, args.dim), "float64", device)
        import torch

        sub = d[:, 400:, torch.as_tensor(dims_for_ess, device=d.device)].cpu().numpy()
    except Exception:
        sub = smp._copy("draws", np.float64, vec=True)[:, 400:, dims_for_ess]
    ess = [float(ess_bulk(sub[:, :, k])) for k in range(sub.shape[2])]
    smp.close()
    return {
        "seconds": secs, "alloc_seconds": t_alloc, "leapfrogs": int(n_steps.sum()), "leapfrogs_tune": int(n_steps[:, :400].sum()),
        "leapfrogs_sample": int(n_steps[:, 400:].sum()), "leapfrogs_per_s": float(n_steps.sum() / secs),
        "mean_depth_sample": float(depth[:, 400:].mean()), "divergences_sample": int(div[:, 400:].sum()),
        "final_step_size_mean": float(step[:, -1].mean()), "ess_dims": len(dims_for_ess), "ess_min": float(np.min(ess)),
        "ess_min_per_s": float(np.min(ess) / secs), "chains": args.chains, "draws": 1000, "tune": 400,
    }


def main():
    args = parse()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (also with a single rank: same code path)
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    device = local_rank if dist is not None else 0
    torch.cuda.set_device(device)

    from nutpie_amd import _lib as hip
    from nutpie_amd.gaussian import ar1_gaussian

    hip.lib()
    model = ar1_gaussian(args.dim)
    s = hip.PyNutsSettings.Diag(args.seed)
    # enough draws that no chain can finish inside the timed region (>= 1 leapfrog per draw); positions are
    # not stored for this leg, the per-draw statistics are (13 small arrays)
    n_draws = (args.warmup + args.steps + 2) * args.evals_per_launch
    s.update(num_tune=400, num_draws=n_draws, num_chains=args.chains * world)
    m = hip.TridiagGaussianModel(model.diag, model.offdiag)
    smp = hip.PySampler(s, m, device=device, waves_per_chain=args.waves, chain_offset=rank * args.chains, n_local_chains=args.chains,
                        store_draws=False, evals_per_launch=args.evals_per_launch, manual=True)
    W = smp.waves_per_chain

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    smp.step(args.warmup)
    barrier()
    n0 = total_leapfrogs(smp)
    tuning0 = sum(p.tuning for p in smp.progress())
    barrier()
    t0 = time.perf_counter()
    done, launches, kernel_ms = smp.step(args.steps)
    barrier()
    t1 = time.perf_counter()
    n1 = total_leapfrogs(smp)
    tuning1 = sum(p.tuning for p in smp.progress())
    assert launches == args.steps and not done
    elapsed = t1 - t0
    leap = float(n1 - n0)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([leap, kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        leap, kernel_ms_sum = float(c[0].item()), float(c[1].item())
        kernel_ms = kernel_ms_sum / world
    smp.close()

    bytes_per_leapfrog = 40.0 * args.dim
    avg_kernel_s = kernel_ms / 1000.0 / args.steps
    leap_per_launch = leap / world / args.steps
    achieved = bytes_per_leapfrog * leap_per_launch / avg_kernel_s / 1e9
    out = {
        "metric": "leapfrog steps/sec (all chains)", "value": leap / elapsed, "unit": "leapfrog steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.dim}-dim correlated Gaussian (AR(1) rho=0.9, analytic logp/grad fused), {args.chains} chains per GPU "
                               f"(BASELINE.json configs[1])", "dim": args.dim, "chains_per_gpu": args.chains, "waves_per_chain": W,
                   "evals_per_launch": args.evals_per_launch, "leapfrogs_per_step": leap / args.steps,
                   "chains_tuning_at_start": int(tuning0), "chains_tuning_at_end": int(tuning1), "parallelism": f"chains{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (MEASURED_HBM_BYTES_PER_LEAPFROG_D1000 * leap_per_launch if (args.dim == 1000 and W == 1) else None),
                     "traffic_source": "rocprofv3 PMC FETCH_SIZE/WRITE_SIZE passes, profiles/r1_v4_pmc.txt (bytes per launch)",
                     "kernel": "k_advance<fused,W=1,NV=8>", "avg_kernel_ms": 1000.0 * avg_kernel_s,
                     "algorithmic_bytes_per_leapfrog": bytes_per_leapfrog},
    }
    if rank == 0 and world == 1 and not args.no_job:
        dims = sorted(set(np.linspace(0, args.dim - 1, 12).astype(int).tolist() + [int(np.argmax(model.diag)), int(np.argmin(model.diag))]))
        out["job"] = run_job(hip, model, args, device, 0, dims)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, args.seed, args.cpu_seconds)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
