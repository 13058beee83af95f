#!/usr/bin/env python
"""bench.py — leapfrog steps/s (all chains) of the HIP NUTS engine on BASELINE.json's headline workload.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` — for N>1 launched by
``torch.distributed.run`` with one rank per GPU.  A *step* is one pass of the hot path over the whole
batch of chains: one launch of the fused leapfrog/tree kernel that advances every chain of this GPU
by ``evals_per_launch`` leapfrogs (one logp+gradient evaluation each).  Set-up (untimed): inputs to HBM
and the chains' whole warm-up (tune = 400), so that the timed region lies in the SAMPLING phase with
positions stored (``--phase tuning`` times inside warm-up instead, as round 1 did).  Then W untimed
steps, then EXACTLY K timed steps bracketed by barrier + synchronize on both sides; time = max over
ranks; rank 0 prints ONE JSON line.  With no ``--steps`` K is sized for a timed region of ~1.2 s.

Workload (``config.workload``): BASELINE.json configs[1] — 1000-dimensional correlated Gaussian
(AR(1) rho = 0.9 with per-dimension scales exp(N(0,1)) from numpy.random.default_rng(20260926),
analytic logp/grad fused in-kernel), 1024 chains per GPU, adaptation "diag", target_accept 0.8,
maxdepth 10, U(-2,2) initial points.  Inputs are generated on the host once and are resident in HBM
before the timed region.  Multi-GPU is weak scaling: 1024 chains per GPU, chain ids offset per rank,
no collective in the data path (only the timing all-reduce).

Extra objects in the JSON line:
  roofline      — what bounds the dominant kernel and how close it runs to that bound, with `frac` <= 1:
                  * register-resident kernels (D <= 4096: the chain state stays in VGPRs / LDS, so the kernel moves LESS than the
                    SURVEY.md §8d "algorithmic" 40 * D bytes per leapfrog and HBM is not its limit): `bound` = "issue" — one wave
                    per SIMD issues at most one instruction every four cycles; `achieved` = wave-instructions per second
                    (instructions per leapfrog from the committed PMC passes x leapfrogs per launch / mean k_advance duration
                    from HIP events on the engine's stream), `peak` = resident waves x 2.4 GHz / 4;
                  * the lean / memory-resident kernels (D > 4096): `bound` = "hbm", `achieved` = measured HBM bytes per launch
                    (PMC) / the same duration, `peak` = 8 TB/s;
                  in both cases `hbm_measured` is the PMC traffic against the 8 TB/s peak, `stream_equivalent` the §8d figure
                  (40 * D bytes per leapfrog / kernel time: what a stream-everything kernel would have to move to keep up —
                  it can exceed the peak and is NOT a fraction of anything), `traffic` the PMC bytes per launch.
  cpu_baseline  — the CPU oracle (oracle/, "port": the real nuts-rs cannot be built here) timed on this
                  box's host cores on a bounded sample of the same workload (rank 0, N=1 only);
  cpu_baseline_tuned — the same C++ sampler built for speed (AVX2 + FMA, free summation order): the honest
                  CPU number; the port is bit-reproducible scalar code.
  tuning_phase  — leapfrogs and kernel-time rate of the (untimed) warm-up that precedes the timed region.
  job           — the complete sampling job (tune 400 + draws 1000) wall time, total leapfrogs, min bulk
                  ESS over a subset of dimensions and ESS/s (the second half of BASELINE.json's metric).
  ranks         — world size, backend, and per rank: device, leapfrogs and kernel time of the timed region (N > 1: what a
                  scaling record is checked against).
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
CLOCK_HZ = 2.4e9       # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs at 2.4 GHz
N_SIMD = 1024
# One wave per SIMD (the register-resident kernels: the whole register file is one wave's) issues at most one instruction, of
# any kind, per four cycles (MI355X_MICROARCH.md, "one wave per SIMD (512-register kernel)": issue slots of ~4 cycles).
CYCLES_PER_ISSUE = 4.0
# measured per-leapfrog HBM bytes and instruction counts per kernel / dimension: profiles/traffic.json, written by
# profiles/make_traffic.py from the committed rocprofv3 PMC summaries of THIS command (timed configuration: sampling phase,
# positions stored, the engine's default launch length; (2 x FETCH_SIZE + WRITE_SIZE) KB per launch / leapfrogs per launch —
# gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=0, help="timed launches; 0 = enough for a timed region of about 1.2 s")
    p.add_argument("--warmup", type=int, default=30)
    p.add_argument("--dim", type=int, default=1000)
    p.add_argument("--chains", type=int, default=1024, help="chains PER GPU")
    p.add_argument("--evals-per-launch", type=int, default=0, help="leapfrogs per chain per launch; 0 = the engine's default for this dimension")
    p.add_argument("--waves", type=int, default=0)
    p.add_argument("--seed", type=int, default=20260926)
    p.add_argument("--phase", choices=("sampling", "tuning"), default="sampling",
                   help="where the timed region lies: after warm-up (default; positions are stored) or inside it")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-job", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
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
    tuned = None
    try:
        chains_t = int(min(1024, max(cores, chains * 4 // cores * cores)))
        st = oracle.default_settings(seed=seed, num_chains=chains_t, num_tune=tune, num_draws=draws, n_threads=cores)
        tt = oracle.sample_tridiag_tuned(st, model.diag, model.offdiag)
        nt = int(tt.stats["n_steps"].sum())
        tuned = {
            "value": nt / tt.seconds, "unit": "leapfrog steps/s", "cores": cores, "kind": "tuned",
            "sample": f"the same C++ sampler built for speed (oracle/Makefile TUNEDFLAGS: AVX2 + FMA, free summation order, fused "
                      f"passes — rounding differs from the contract, never used as a checker), {chains_t} chains, tune {tune} + draws "
                      f"{draws}, {cores} threads: {nt} leapfrogs in {tt.seconds:.2f} s",
        }
    except Exception as e:  # the tuned build is optional evidence
        tuned = {"error": repr(e)}
    return tuned, {
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
    smp = hip.PySampler(s, m, device=device, waves_per_chain=args.waves, chain_offset=chain_offset, n_local_chains=args.chains)
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


def pmc_entry(dim, W):
    try:
        table = json.load(open(TRAFFIC_JSON))
    except OSError:
        return None
    return table.get(f"{dim}:{W}")


def measured_traffic(dim, W, lean):
    """HBM bytes per leapfrog of the kernel that runs at this (dim, waves) from the committed PMC summaries, or None."""
    e = pmc_entry(dim, W)
    return (e["bytes_per_leapfrog"], e["source"]) if e else (None, None)


def roofline(dim, W, chains, leap_per_launch, avg_kernel_s):
    """The `roofline` object of the JSON line (module docstring).  Live: the kernel duration.  From profiles/: bytes and
    instructions per leapfrog of this kernel at the timed configuration."""
    e = pmc_entry(dim, W)
    kname = kernel_name(dim, W)
    stream_bpl = 40.0 * dim
    stream = {"bytes_per_leapfrog": stream_bpl, "GB/s": stream_bpl * leap_per_launch / avg_kernel_s / 1e9,
              "note": "SURVEY.md 8d 'algorithmic' bytes (read q, p, sigma^2; write q', p' every leapfrog) / kernel time: the rate a "
                      "stream-everything kernel would need to keep up.  Not a fraction of a peak: a kernel that keeps the state on chip moves less"}
    stream["over_hbm_peak"] = stream["GB/s"] / HBM_PEAK_GBS
    out = {"kernel": kname, "avg_kernel_ms": 1000.0 * avg_kernel_s, "stream_equivalent": stream}
    hbm = None
    if e:
        gbs = e["bytes_per_leapfrog"] * leap_per_launch / avg_kernel_s / 1e9
        hbm = {"GB/s": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "bytes_per_leapfrog": e["bytes_per_leapfrog"],
               "over_algorithmic": e["bytes_per_leapfrog"] / stream_bpl, "source": e["source"], "pmc_config": e.get("config")}
        out["traffic"] = e["bytes_per_leapfrog"] * leap_per_launch
    else:
        out["traffic"] = None
    out["hbm_measured"] = hbm
    register_resident = "lean" not in kname and "NV=0" not in kname
    if register_resident and e and e.get("insts_per_leapfrog"):
        ipl = e["insts_per_leapfrog"]["total"]
        waves = min(chains * W, N_SIMD * 8)
        peak = waves * CLOCK_HZ / CYCLES_PER_ISSUE / 1e9
        ach = ipl * leap_per_launch / avg_kernel_s / 1e9
        out.update({"bound": "issue", "achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
                    "issue": {"insts_per_leapfrog": e["insts_per_leapfrog"], "pmc_issuing_fraction_of_wave_cycles": e.get("issuing_fraction"),
                              "pmc_waiting_fraction_of_wave_cycles": e.get("waiting_fraction"), "resident_waves": waves,
                              "note": "one wave per SIMD: at most one instruction per 4 cycles per wave; achieved = PMC instructions per leapfrog x "
                                      "leapfrogs per launch / kernel time (HIP events)"}})
    elif hbm:
        out.update({"bound": "hbm", "achieved": hbm["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac_of_peak"]})
    else:
        # no PMC summary committed for this (dim, waves): only the stream-equivalent rate is known; capped, and labelled as such
        out.update({"bound": "hbm", "achieved": min(stream["GB/s"], HBM_PEAK_GBS), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": min(1.0, stream["over_hbm_peak"]), "note": "no PMC summary for this kernel in profiles/traffic.json: stream-equivalent rate, capped at the peak"})
    return out


def kernel_name(dim, W):
    nch = (dim + 127) // 128
    if W == 1 and nch <= 8:
        return f"k_advance<fused,W=1,NV={nch}>"
    if W in (2, 4) and (nch + W - 1) // W <= 8:
        return f"k_advance<fused,W={W},NV={(nch + W - 1) // W}>"
    if (W == 4 and (nch + 3) // 4 <= 20) or (W == 8 and (nch + 7) // 8 <= 10):
        return f"k_advance<fused,W={W},NV={(nch + W - 1) // W},lean>"
    return f"k_advance<fused,W={W},NV=0>"


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
    # The launch length the engine runs by default (host.hip: default_evals_per_launch — about 10 ms of kernel: 2048 leapfrogs per
    # chain at D = 1000, 512 at D = 10 000).  A launch boundary costs every chain a flush and a reload of its register state and the
    # device the tail of the slowest chain.
    E = args.evals_per_launch or hip.default_evals_per_launch(args.dim)
    # default number of timed launches: about 1.2 s of kernel time (178 M leapfrogs/s at D = 1000 scales like 1 / D)
    K = args.steps or max(20, int(1.2 * 1.7e11 / args.dim / (args.chains * E)))
    num_tune = 400
    # Draws to allocate: the timed region must end before any chain runs out of draws.  After warm-up a draw of these targets
    # takes ~200 (D = 1000) to ~500 (D = 10 000) leapfrogs; 1 / 64 of the leapfrogs is a 3x margin, checked below.
    # (+ 12 launches: the warm-up below is advanced 10 launches at a time and may overshoot the end of tuning by that much)
    n_draws = max(64, (args.warmup + K + 2 + 12) * E // 64) if args.phase == "sampling" else (args.warmup + K + 2) * E
    s = hip.PyNutsSettings.Diag(args.seed)
    s.update(num_tune=num_tune, num_draws=n_draws, num_chains=args.chains * world)
    m = hip.TridiagGaussianModel(model.diag, model.offdiag)
    store = args.phase == "sampling"
    smp = hip.PySampler(s, m, device=device, waves_per_chain=args.waves, chain_offset=rank * args.chains, n_local_chains=args.chains,
                        store_draws=store, evals_per_launch=E, manual=True)
    W = smp.waves_per_chain

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- set-up, untimed: the whole warm-up (tune = 400 draws per chain); its rate is reported as `tuning_phase`
    tuning_phase = None
    if args.phase == "sampling":
        t0 = time.perf_counter()
        n0 = total_leapfrogs(smp)
        launches = 0
        kms = 0.0
        while True:
            done, l, ms = smp.step(10)
            launches += l
            kms += ms
            if done:
                raise SystemExit("bench.py: chains finished during warm-up; increase the number of draws")
            if not any(p.tuning for p in smp.progress()):
                break
        torch.cuda.synchronize()
        n1 = total_leapfrogs(smp)
        tuning_phase = {"leapfrogs": int(n1 - n0), "launches": launches, "wall_s": time.perf_counter() - t0,
                        "leapfrogs_per_s_kernel_time": (n1 - n0) / (kms / 1e3), "note": "all chains through tune = 400 draws (untimed set-up of the "
                        "bench; includes initial points, step-size search, mass-matrix adaptation; wall time includes progress polling)"}
    smp.step(args.warmup)
    barrier()
    n0 = total_leapfrogs(smp)
    tuning0 = sum(p.tuning for p in smp.progress())
    barrier()
    t0 = time.perf_counter()
    done, launches, kernel_ms = smp.step(K)
    barrier()
    t1 = time.perf_counter()
    n1 = total_leapfrogs(smp)
    prog = smp.progress()
    tuning1 = sum(p.tuning for p in prog)
    if launches != K or done or max(p.finished_draws for p in prog) >= num_tune + n_draws:
        raise SystemExit("bench.py: a chain ran out of draws inside the timed region — the measurement is invalid; use fewer --steps")
    elapsed = t1 - t0
    leap = float(n1 - n0)
    per_rank = [{"rank": rank, "device": device, "leapfrogs": leap, "kernel_ms": kernel_ms, "elapsed_s": elapsed}]
    if dist is not None:
        mine = torch.tensor([rank, device, leap, kernel_ms, elapsed], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": int(t[0]), "device": int(t[1]), "leapfrogs": float(t[2]), "kernel_ms": float(t[3]), "elapsed_s": float(t[4])} for t in allr]
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([leap, kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        leap, kernel_ms_sum = float(c[0].item()), float(c[1].item())
        kernel_ms = kernel_ms_sum / world
    draws_stored = int(sum(p.finished_draws for p in prog)) if store else 0
    smp.close()

    avg_kernel_s = kernel_ms / 1000.0 / K
    leap_per_launch = leap / world / K
    out = {
        "metric": "leapfrog steps/sec (all chains)", "value": leap / elapsed, "unit": "leapfrog steps/s", "n_gpus": world,
        "steps": K, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.dim}-dim correlated Gaussian (AR(1) rho=0.9, analytic logp/grad fused), {args.chains} chains per GPU "
                               f"(BASELINE.json configs[{1 if args.dim == 1000 else 4}])", "dim": args.dim, "chains_per_gpu": args.chains, "waves_per_chain": W,
                   "evals_per_launch": E, "leapfrogs_per_step": leap / K, "phase": args.phase, "positions_stored": store,
                   "draws_finished_all_chains": draws_stored, "timed_region_s": elapsed,
                   "chains_tuning_at_start": int(tuning0), "chains_tuning_at_end": int(tuning1), "parallelism": f"chains{world}"},
        "roofline": roofline(args.dim, W, args.chains, leap_per_launch, avg_kernel_s),
        "ranks": {"world": world, "backend": (dist.get_backend() if dist is not None else None),
                  "collective_ranks": (dist.get_world_size() if dist is not None else 1), "per_rank": per_rank,
                  "note": "chains sharded by global chain id (rank r owns chains [r * chains_per_gpu, (r + 1) * chains_per_gpu)); no data-path "
                          "collective, only this report's gather and the timing all-reduce"},
    }
    if tuning_phase is not None:
        out["tuning_phase"] = tuning_phase
    if rank == 0 and world == 1 and not args.no_job and args.dim <= 2048:
        dims = sorted(set(np.linspace(0, args.dim - 1, 12).astype(int).tolist() + [int(np.argmax(model.diag)), int(np.argmin(model.diag))]))
        out["job"] = run_job(hip, model, args, device, 0, dims)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        tuned, port = cpu_baseline(model, args.seed, args.cpu_seconds)
        out["cpu_baseline"] = port
        out["gpu_over_cpu"] = out["value"] / port["value"]
        if tuned:
            out["cpu_baseline_tuned"] = tuned
            if "value" in tuned:
                out["gpu_over_cpu_tuned"] = out["value"] / tuned["value"]
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
