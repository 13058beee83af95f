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
  job           — the complete sampling job (tune 400 + draws 1000) wall time, total leapfrogs, min bulk ESS over ALL
                  dimensions (computed on the GPU that holds the trace) and ESS/s (the second half of BASELINE.json's metric).
  ranks         — world size, backend, and per rank: device, leapfrogs and kernel time of the timed region (N > 1: what a
                  scaling record is checked against).
  config5_shard — BASELINE.json configs[4]'s per-GPU shard on every rank (10 000 dimensions x 1024 chains, lean kernel): a
                  timed region of >= 10 launches in the sampling phase with its own `roofline` (bound "hbm"), then the
                  job runs to its end and its trace goes to rank 0 in ONE gather (nutpie_amd.distributed.gather_trace:
                  draws thinned on the device + per-chain moments + statistics) — seconds, bytes, ranks of the collective.
  other_configs — (N = 1) runs of the remaining BASELINE.json configs on this GPU: config 3 (radon, 512 chains) with the generated
                  density and with the torch density (also its cold compile time), config 4 (eight schools, 256 chains, host C
                  callback), config 2 (ii) (dense 1000-dim Gaussian) INSIDE the engine — the hand-written fp64 MFMA gradient in the
                  resident kernel: a timed region in the sampling phase with an `mfma` roofline against the matrix-core rate measured
                  in the same run — and behind the rocBLAS callback of round 5.  EVERY leg carries `cpu_baseline` (the oracle on the same
                  density, a bounded sample on this box's cores; `tuned` = its free-order SIMD build) and `gpu_over_cpu`, also where it
                  is below 1 (config 4: a 10-dimensional host callback is faster on 16 CPU threads than through the GPU).

Launching: under torch.distributed.run (RANK in the environment) this process is one rank.  WITHOUT it, `--gpus N` with
N > 1 re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`,
one rank per GPU, and fails loudly when fewer than N GPUs are visible.
"""
from __future__ import annotations

import argparse
import json
import math
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


def parse(argv=None):
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
    p.add_argument("--no-other-configs", action="store_true", help="skip the bounded runs of configs 3, 4, 2 (ii) (N = 1)")
    p.add_argument("--no-config5", action="store_true", help="skip the config-5 shard leg (10 000 dimensions, roofline + trace gather)")
    p.add_argument("--config5-dim", type=int, default=10000)
    p.add_argument("--config5-launches", type=int, default=10)
    p.add_argument("--engine-stub", default="", help="TEST PLUMBING ONLY: path of a module that stands in for nutpie_amd._lib (no GPU, gloo); "
                   "the line it prints is marked stub and measures nothing")
    return p.parse_args(argv)


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


def cpu_leg(run, chains_total, tune, draws, target_seconds, what, tuned_run=None):
    """`cpu_baseline` of one leg: the CPU oracle ("port") on this box's host cores on a BOUNDED sample of the leg's workload — the same
    density, settings and seed; one chain per thread on min(chains, cores) threads (the reference's `cores` model:
    python/nutpie/sample.py:856-857, 1061-1070); as many of the workload's chains as fit `target_seconds` (all of them when they do).
    `run(chains, tune, draws, threads)` -> oracle Trace.  `tuned_run`: the same with the free-order SIMD build (a second, faster CPU number)."""
    cores = effective_cores()
    cal_tune = max(4, min(tune, 20))
    cal = run(min(cores, chains_total), cal_tune, 4, cores)
    n_cal = float(cal.stats["n_steps"].sum())
    rate = n_cal / max(cal.seconds, 1e-6)
    per_chain = n_cal / min(cores, chains_total) / (cal_tune + 4) * (tune + draws) * 1.5   # (later draws take longer trees than the first ones)
    chains = int(min(chains_total, max(min(cores, chains_total), (rate * target_seconds / max(per_chain, 1.0)) // cores * cores)))
    tr = run(chains, tune, draws, cores)
    if tr.seconds < target_seconds / 3.0 and chains < chains_total:   # the estimate was pessimistic (early warm-up draws are the expensive ones): once more, larger
        chains = int(min(chains_total, max(chains, (chains * target_seconds / max(tr.seconds, 1e-3)) // cores * cores)))
        tr = run(chains, tune, draws, cores)
    n = int(tr.stats["n_steps"].sum())
    out = {"value": n / tr.seconds, "unit": "leapfrog steps/s", "cores": min(cores, chains), "kind": "port",
           "sample": f"CPU oracle (oracle/: C++ restatement of nuts-rs diag-NUTS) on {what}: {chains} of the workload's {chains_total} chains, tune {tune} + draws {draws}, "
                     f"one chain per thread on {min(cores, chains)} threads (host: {os.cpu_count()} logical CPUs, cgroup/affinity limit {cores}): {n} leapfrogs in {tr.seconds:.2f} s"}
    if tuned_run is not None:
        try:
            tt, chains_t = tuned_run(chains, tune, draws, cores), chains
            if tt.seconds < target_seconds / 3.0 and chains_t < chains_total:   # (much faster than the port: a sample of its own size)
                chains_t = int(min(chains_total, max(chains_t, (chains_t * target_seconds / max(tt.seconds, 1e-3)) // cores * cores)))
                tt = tuned_run(chains_t, tune, draws, cores)
            nt = int(tt.stats["n_steps"].sum())
            out["tuned"] = {"value": nt / tt.seconds, "unit": "leapfrog steps/s", "cores": min(cores, chains), "kind": "tuned",
                            "sample": f"the same sampler built for speed (AVX2 + FMA, free summation order: not a checker), {chains_t} chains: {nt} leapfrogs in {tt.seconds:.2f} s"}
        except Exception as e:   # optional evidence
            out["tuned"] = {"error": repr(e)}
    return out


def with_cpu(leg, cpu):
    """Attach a leg's CPU baseline and the ratio (also where it is below 1)."""
    leg["cpu_baseline"] = cpu
    if cpu and "value" in cpu and leg.get("leapfrogs_per_s"):
        leg["gpu_over_cpu"] = leg["leapfrogs_per_s"] / cpu["value"]
        if "tuned" in cpu and "value" in cpu["tuned"]:
            leg["gpu_over_cpu_tuned"] = leg["leapfrogs_per_s"] / cpu["tuned"]["value"]
    return leg


def run_job(hip, model, args, device, chain_offset, dims_for_ess, world=1):
    """The complete job: tune 400 + draws 1000 on this GPU's chains; returns wall seconds, leapfrogs, ESS — the bulk ESS of EVERY
    dimension (min over all dimensions), computed on the GPU that holds the trace (nutpie_amd.ess.ess_bulk_all: one batched
    sort + one batched FFT per block of dimensions); `dims_for_ess` are also evaluated by the per-dimension CPU routine as a cross-check."""
    from nutpie_amd.ess import ess_bulk, ess_bulk_all

    s = hip.PyNutsSettings.Diag(args.seed)
    s.update(num_tune=400, num_draws=1000, num_chains=args.chains * world)
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
    from nutpie_amd.distributed import device_tensor
    import torch

    d = device_tensor(smp.device_ptr("draws"), (args.chains, 1400, args.dim), "float64", device)
    t_ess = time.perf_counter()
    ess_all = ess_bulk_all(d[:, 400:, :], block=25)
    torch.cuda.synchronize()
    t_ess = time.perf_counter() - t_ess
    sub = d[:, 400:, torch.as_tensor(dims_for_ess, device=d.device)].cpu().numpy()
    ess_cpu = np.array([float(ess_bulk(sub[:, :, k])) for k in range(sub.shape[2])])
    del d
    smp.close()
    ess = ess_all
    return {
        "ess_seconds_on_device": t_ess, "ess_argmin_dim": int(np.nanargmin(ess_all)), "ess_median": float(np.nanmedian(ess_all)), "ess_max": float(np.nanmax(ess_all)),
        "ess_cross_check": {"dims": [int(k) for k in dims_for_ess], "max_rel_diff_device_vs_cpu_routine": float(np.max(np.abs(ess_all[dims_for_ess] - ess_cpu) / ess_cpu))},
        "seconds": secs, "alloc_seconds": t_alloc, "leapfrogs": int(n_steps.sum()), "leapfrogs_tune": int(n_steps[:, :400].sum()),
        "leapfrogs_sample": int(n_steps[:, 400:].sum()), "leapfrogs_per_s": float(n_steps.sum() / secs),
        "mean_depth_sample": float(depth[:, 400:].mean()), "divergences_sample": int(div[:, 400:].sum()),
        "final_step_size_mean": float(step[:, -1].mean()), "ess_dims": int(len(ess)), "ess_min": float(np.nanmin(ess)),
        "ess_min_per_s": float(np.nanmin(ess) / secs), "chains": args.chains, "draws": 1000, "tune": 400,
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


def roofline(dim, W, chains, leap_per_launch, avg_kernel_s, leap_per_draw=None):
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
        # the honest companion figure: of the instructions the kernel issues per leapfrog, how many are the fp64 operations
        # the mathematics needs — per element 19 for the leaf itself (leapfrog 9, gradient 5, logp 1, level-0 criterion 4) plus the
        # U-turn criteria of the merge levels >= 1 a leaf closes: 3 criteria x 6 operations per level, and a tree of final depth dd closes
        # (2^dd - 1 - dd) such levels and dd top-level merges (2.5 criteria on average) over its 2^dd - 1 leaves
        elems = ((dim + 127) // 128) * 2
        dd = math.log2((leap_per_draw or 31.0) + 1.0)
        leaves = 2.0 ** dd - 1.0
        levels_per_leaf = max(0.0, (leaves - dd) / leaves)
        top_per_leaf = dd / leaves
        per_elem = 19.0 + 18.0 * levels_per_leaf + 15.0 * top_per_leaf
        useful = per_elem * elems
        out["useful_work"] = {"useful_fp64_insts_per_leapfrog": useful, "fraction_of_issued": useful / ipl,
                              "fraction_of_issue_slots": useful * leap_per_launch / avg_kernel_s / 1e9 / peak,
                              "per_element": {"leaf": 19.0, "merge_levels_per_leaf": levels_per_leaf, "top_level_merges_per_leaf": top_per_leaf, "total": per_elem},
                              "leapfrogs_per_draw": leap_per_draw,
                              "note": "fp64 operations per element (leaf 19 + merge criteria of the levels a leaf closes, from the measured leapfrogs per draw) x elements "
                                      "per lane; everything else the kernel issues is tree bookkeeping, reductions, register traffic (AGPR moves, SGPR spill "
                                      "lanes) and address arithmetic"}
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


def free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves, one per GPU of this node."""
    import subprocess

    if not args.engine_stub:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node — refusing to run fewer ranks than asked for")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


class _stdout_to_stderr:
    """fd 1 -> fd 2 for the duration: RCCL prints a version banner through C stdio when its first communicator is created, and
    stdout is for the ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


class Env:
    """What differs between a real run (GPU, RCCL) and the stub used by the CPU plumbing test (no device, gloo)."""

    def __init__(self, args):
        import torch

        self.torch = torch
        self.stub = bool(args.engine_stub)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if "RANK" in os.environ and self.world != args.gpus:
            raise SystemExit(f"bench.py: launched with WORLD_SIZE={self.world} but --gpus {args.gpus}: they must agree")
        if not self.stub:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
            if torch.cuda.device_count() <= self.local_rank:
                raise SystemExit(f"bench.py: rank {self.rank} wants GPU {self.local_rank}, but only {torch.cuda.device_count()} are visible")
            torch.cuda.set_device(self.local_rank)
        self.device = self.local_rank if "RANK" in os.environ else 0
        if "RANK" in os.environ:  # launched by torch.distributed.run (also with a single rank: same code path)
            import torch.distributed as dist

            if self.stub:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
                with _stdout_to_stderr():   # (the communicator is created by the first collective)
                    dist.barrier()
                    torch.cuda.synchronize()
            self.dist = dist
        if self.stub:
            import importlib.util

            spec = importlib.util.spec_from_file_location("bench_engine_stub", args.engine_stub)
            self.hip = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(self.hip)
        else:
            from nutpie_amd import _lib as hip

            hip.lib()
            self.hip = hip

    def ensure_group(self):
        """A process group for the gather leg also when bench.py runs as a plain single process (one-rank RCCL)."""
        if self.dist is None:
            import torch.distributed as dist

            if self.stub:
                dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
            else:
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                                        device_id=self.torch.device("cuda", self.device))
                with _stdout_to_stderr():
                    dist.barrier()
                    self.torch.cuda.synchronize()
            self.dist = dist
        return self.dist

    def sync(self):
        if not self.stub:
            self.torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
            self.sync()

    def tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64, device="cpu" if self.stub else "cuda")

    def all_max(self, x):
        if self.dist is None:
            return float(x)
        t = self.tensor([x])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(self, values):
        if self.dist is None:
            return [float(v) for v in values]
        t = self.tensor(list(values))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def all_rows(self, row):
        """Every rank's row (a list of floats), in rank order."""
        if self.dist is None:
            return [list(map(float, row))]
        mine = self.tensor(list(row))
        allr = [self.torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allr, mine)
        return [t.tolist() for t in allr]


class InvalidRegion(RuntimeError):
    pass


def warm_up_to_sampling(smp, batch=10):
    """Advance every chain through its warm-up (manual mode).  Returns (leapfrogs, launches, kernel ms, wall s)."""
    t0 = time.perf_counter()
    n0 = total_leapfrogs(smp)
    launches, kms = 0, 0.0
    while True:
        done, l, ms = smp.step(batch)
        launches += l
        kms += ms
        if done:
            raise InvalidRegion("bench.py: chains finished during warm-up; increase the number of draws")
        if not any(p.tuning for p in smp.progress()):
            break
    return total_leapfrogs(smp) - n0, launches, kms, time.perf_counter() - t0


def timed_launches(env, smp, K, total_draws):
    """EXACTLY K launches between barrier + synchronize on both sides.  Returns (elapsed s of this rank, leapfrogs, kernel ms)."""
    env.barrier()
    n0 = total_leapfrogs(smp)
    env.barrier()
    t0 = time.perf_counter()
    done, launches, kernel_ms = smp.step(K)
    env.barrier()
    t1 = time.perf_counter()
    prog = smp.progress()
    # (the verdict is COLLECTIVE: a rank that left alone would leave the others waiting in the next collective)
    bad = launches != K or done or max(p.finished_draws for p in prog) >= total_draws
    if env.all_max(1.0 if bad else 0.0) > 0.0:
        raise InvalidRegion("bench.py: a chain ran out of draws inside the timed region" + ("" if bad else " (on another rank)") +
                            " — the measurement is invalid; use fewer --steps")
    return t1 - t0, float(total_leapfrogs(smp) - n0), kernel_ms, prog


def config5_shard(env, args):
    """BASELINE.json configs[4] on every rank: this GPU's shard (1024 chains x 10 000 dimensions, lean register kernel) — a
    timed region in the sampling phase with the `hbm` roofline, then the job to its end and ONE gather of the thinned trace
    + on-device moments + statistics to rank 0 (nutpie_amd.distributed.gather_trace; RCCL over xGMI for N > 1).
    Bounded: a SHORT warm-up (tune = 60) and 64 draws — the kernel's rate per leapfrog does not depend on how well adapted the chain is."""
    from nutpie_amd.distributed import gather_trace
    from nutpie_amd.gaussian import ar1_gaussian

    hip, dim, chains = env.hip, args.config5_dim, args.chains
    model = ar1_gaussian(dim)
    E = hip.default_evals_per_launch(dim)
    K, tune = max(1, args.config5_launches), 60
    n_draws = max(64, (K + 6) * E // 96)   # (a 10 000-dimensional draw takes 250 .. 1000 leapfrogs; checked by timed_launches)
    s = hip.PyNutsSettings.Diag(args.seed)
    s.update(num_tune=tune, num_draws=n_draws, num_chains=chains * env.world)
    smp = hip.PySampler(s, hip.TridiagGaussianModel(model.diag, model.offdiag), device=env.device, chain_offset=env.rank * chains,
                        n_local_chains=chains, store_draws=True, evals_per_launch=E, manual=True)
    W = smp.waves_per_chain
    t_all = time.perf_counter()
    err = None
    try:
        warm = warm_up_to_sampling(smp, batch=4)
    except InvalidRegion as e:
        err = str(e)
    if env.all_max(1.0 if err else 0.0) > 0.0:
        smp.close()
        raise InvalidRegion(err or "another rank's chains finished during warm-up")
    smp.step(2)
    try:
        elapsed, leap, kernel_ms, _ = timed_launches(env, smp, K, tune + n_draws)
    except InvalidRegion:
        smp.close()
        raise
    elapsed_max = env.all_max(elapsed)
    leap_sum, kms_sum = env.all_sum([leap, kernel_ms])
    t0 = time.perf_counter()
    while True:   # the rest of the job: the trace that is gathered is a finished one
        done, _, _ = smp.step(8)
        if done:
            break
    env.sync()
    finish_s = time.perf_counter() - t0
    dist = env.ensure_group()
    timing = {}
    thin = 8
    g = gather_trace(smp, chains, chains * env.world, thin=thin, moments_after=tune, timing=timing,
                     device=None if env.stub else env.device)
    check = None
    if g is not None:
        mean = g["draw_mean"]
        check = {"chains_gathered": int(mean.shape[0]), "draws_shape": list(g["draws"].shape), "moments_shape": list(mean.shape),
                 "posterior_mean_abs_max": float(mean.mean(0).abs().max())}
    smp.close()
    avg_kernel_s = kms_sum / env.world / 1000.0 / K
    out = {
        "workload": f"{dim}-dim correlated Gaussian (AR(1) rho=0.9, fused), {chains} chains per GPU x {env.world} GPU(s) (BASELINE.json configs[4] "
                    f"per-GPU shard); short warm-up (tune {tune}), timed region in the sampling phase with positions stored",
        "value": leap_sum / elapsed_max, "unit": "leapfrog steps/s", "steps": K, "ms_per_step": 1000.0 * elapsed_max / K,
        "waves_per_chain": W, "evals_per_launch": E, "warmup": {"leapfrogs": warm[0], "launches": warm[1], "kernel_ms": warm[2], "wall_s": warm[3]},
        "roofline": roofline(dim, W, chains, leap_sum / env.world / K, avg_kernel_s),
        "gather": {"backend": dist.get_backend(), "thin": thin, "finish_job_s": finish_s,
                   "note": "one torch.distributed.gather per array to rank 0 (RCCL: direct sends over xGMI); draws thinned and moments reduced on the device first",
                   **timing, "check": check},
        "leg_wall_s": time.perf_counter() - t_all,
    }
    if timing.get("gather_s") and timing.get("bytes_gathered") and env.world > 1:
        out["gather"]["GB_per_s_into_root"] = timing["bytes_gathered"] * (env.world - 1) / env.world / timing["gather_s"] / 1e9
    if env.rank == 0 and not env.stub and not args.no_cpu_baseline:
        # the CPU beside it: the oracle on the same 10 000-dimensional target (rank 0 only: the other ranks wait in the next collective)
        try:
            import oracle

            oracle.build()

            def run(chains_, tune_, draws_, threads):
                return oracle.sample_tridiag(oracle.default_settings(seed=args.seed, num_chains=chains_, num_tune=tune_, num_draws=draws_, n_threads=threads), model.diag, model.offdiag)

            def run_t(chains_, tune_, draws_, threads):
                return oracle.sample_tridiag_tuned(oracle.default_settings(seed=args.seed, num_chains=chains_, num_tune=tune_, num_draws=draws_, n_threads=threads), model.diag, model.offdiag)

            out["cpu_baseline"] = cpu_leg(run, chains, tune, 8, min(6.0, args.cpu_seconds), f"the {dim}-dim AR(1) Gaussian of this leg", tuned_run=run_t)
            out["gpu_over_cpu"] = out["value"] / env.world / out["cpu_baseline"]["value"]
            out["gpu_over_cpu_note"] = "one GPU's rate over the host's (the CPU sample is dominated by warm-up draws; the rate per leapfrog does not depend on the phase)"
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def job_rate(smp, t_create):
    """Wait for a sampler and describe the job it ran."""
    smp.wait()
    secs = smp.seconds
    n = smp._copy("n_steps", np.int64)
    div = smp._copy("diverging", np.bool_)
    tun = smp._copy("tuning", np.bool_)
    out = {"leapfrogs_per_s": float(n.sum() / secs), "job_s": secs, "wall_incl_setup_s": time.perf_counter() - t_create, "leapfrogs": int(n.sum()),
           "mean_leapfrogs_per_draw_sampling": float(n[~tun].mean()) if (~tun).any() else None, "divergences_sampling": int(div[~tun].sum()),
           "launches": smp.launches}
    smp.close()
    return out


def job_roofline(key, leapfrogs_per_s, waves):
    """`roofline` of a whole job whose kernel has a PMC summary in profiles/traffic.json (round 5: config 3's compiled densities): the
    resident kernels keep the chain state on chip — the resource that binds is instruction issue on the SIMDs the job occupies."""
    try:
        e = json.load(open(TRAFFIC_JSON)).get(key)
    except OSError:
        e = None
    if not e:
        return None
    ipl = e["insts_per_leapfrog"]["total"]
    peak = min(waves, N_SIMD * 8) * CLOCK_HZ / CYCLES_PER_ISSUE / 1e9
    ach = ipl * leapfrogs_per_s / 1e9
    gbs = e["bytes_per_leapfrog"] * leapfrogs_per_s / 1e9
    return {"bound": "issue", "achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
            "frac_of_whole_device": ach / (N_SIMD * CLOCK_HZ / CYCLES_PER_ISSUE / 1e9), "resident_waves": waves,
            "insts_per_leapfrog": e["insts_per_leapfrog"], "pmc_issuing_fraction_of_wave_cycles": e["issuing_fraction"],
            "pmc_waiting_fraction_of_wave_cycles": e["waiting_fraction"],
            "hbm_measured": {"GB/s": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "bytes_per_leapfrog": e["bytes_per_leapfrog"], "note": e.get("note")},
            "source": e["source"], "pmc_config": e["config"],
            "note": "whole job (warm-up included) over engine time; achieved = PMC instructions per leapfrog x leapfrogs/s; peak = one instruction per 4 cycles "
                    "for each of the job's waves (512 chains occupy 512 of the 1024 SIMDs)"}


def other_configs(env, args):
    """Bounded runs of the BASELINE.json configs the headline does not cover, on this GPU (rank 0, N = 1).  Each entry is a
    whole `sample`-shaped job (its wall time from sampler creation to the last draw, leapfrogs from the trace)."""
    hip = env.hip
    out = {}

    def leg(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:   # evidence legs: a failure is reported in the line, it does not lose the headline
            out[name] = {"error": repr(e)}
        out[name]["leg_wall_s"] = time.perf_counter() - t0

    def settings(chains, tune, draws, seed=20260926):
        s = hip.PyNutsSettings.Diag(seed)
        s.update(num_tune=tune, num_draws=draws, num_chains=chains)
        return s

    def c3_generated():
        from nutpie_amd.radon import radon_symbolic_model

        m = radon_symbolic_model().compile()
        t0 = time.perf_counter()
        r = job_rate(m._make_sampler(settings(512, 400, 1000), None, 1, None, None, None, None), t0)
        r["workload"] = "radon (D = 173, 85 counties, 919 observations), 512 chains, tune 400 + draws 1000; density generated by nutpie_amd.symbolic, compiled into its own resident kernel"
        r["roofline"] = job_roofline("config3_compiled_density", r["leapfrogs_per_s"], 512)
        return r

    def c3_low_rank():
        from nutpie_amd import low_rank
        from nutpie_amd.radon import radon_symbolic_model

        m = radon_symbolic_model().compile()
        s = hip.PyNutsSettings.LowRank(20260926)
        s.update(num_tune=400, num_draws=1000, num_chains=512)
        # the job three times: the first one pays what a process pays once (the low-rank build of the model's library is loaded, the
        # estimator kernel and its stream are set up, the allocator's first blocks), and who is handed in with whom depends on the clock
        # (low_rank.py::_run_fast) — the line's figures are those of the run with the median wall time, all three walls beside them
        runs = []
        for rep in range(3):
            t0 = time.perf_counter()
            smp = low_rank.make_sampler(m, s, None, 1, None, None, None, None)
            smp.wait()
            log = list(smp.switch_log)
            r = job_rate(smp, t0)
            r.update(hand_ins=len(log), estimating_s=float(sum(e[2] for e in log)), mean_columns=float(np.mean([e[1] for e in log])) if log else 0.0)
            runs.append(r)
        r = dict(sorted(runs, key=lambda x: x["wall_incl_setup_s"])[1])
        r["all_runs"] = [{k: x[k] for k in ("job_s", "wall_incl_setup_s", "hand_ins", "estimating_s")} for x in runs]
        r["workload"] = ("radon as above under adaptation='low_rank' (not a BASELINE config; SURVEY 8f N4): the metric on the register-resident leaf of the compiled "
                         "density, window estimates (nphip_low_rank_estimate: one kernel, a workgroup per chain) handed to each chain as it stops, chains that need no low-rank part released; job_s is engine time, wall_incl_setup_s the job")
        return r

    def c3_traced():
        from nutpie_amd.radon import radon_traced_model

        t_c = time.perf_counter()
        m = radon_traced_model()          # torch.fx trace -> expression graph -> symbolic gradient -> HIP source (-> hipcc unless cached)
        m.library_path()
        compile_s = time.perf_counter() - t_c
        t0 = time.perf_counter()
        r = job_rate(m._make_sampler(settings(512, 400, 1000), None, 1, None, None, None, None), t0)
        r["trace_and_compile_s"] = compile_s
        # what a user pays the first time: the same model traced and compiled with an EMPTY cache (hipcc on this box's host cores; the
        # figure above found the library the ahead-of-time build left in the tree)
        import shutil
        import tempfile

        cold_dir = tempfile.mkdtemp(prefix="nutpie_amd_cold_cache_")
        old = os.environ.get("NUTPIE_AMD_CACHE")
        os.environ["NUTPIE_AMD_CACHE"] = cold_dir
        try:
            t_c = time.perf_counter()
            radon_traced_model().library_path()
            r["cold_compile_s"] = time.perf_counter() - t_c
        except Exception as e:
            r["cold_compile_s"] = None
            r["cold_compile_error"] = repr(e)
        finally:
            if old is None:
                os.environ.pop("NUTPIE_AMD_CACHE", None)
            else:
                os.environ["NUTPIE_AMD_CACHE"] = old
            shutil.rmtree(cold_dir, ignore_errors=True)
        r["roofline"] = job_roofline("config3_traced_torch_density", r["leapfrogs_per_s"], 512)
        r["workload"] = ("radon, 512 chains, tune 400 + draws 1000; the model is a TORCH log-density (forward pass only, nutpie_amd.radon.radon_torch_density) that "
                         "nutpie_amd.from_torch_density traces (torch.fx), differentiates and compiles into its own resident kernel")
        return r

    def c3_two_jobs():
        # 512 chains are 512 wavefronts on 1024 SIMDs: a second, independent 512-chain job (its own sampler, driver thread and stream)
        # runs beside the first on the other half of the device
        from nutpie_amd.radon import radon_symbolic_model

        m = radon_symbolic_model().compile()
        t0 = time.perf_counter()
        # (streams of one priority share the device's few hardware queues; two that land on the same queue serialise their launches — this
        #  leg then measured 63 M leapfrogs/s instead of 120, in one bench run of three.  The second job runs on a high-priority stream: a
        #  queue of its own by construction)
        import torch

        hi = torch.cuda.Stream(env.device, priority=-1)
        smps = [m._make_sampler(settings(512, 400, 1000, seed=20260926 + k), None, 1, None, None, None, None, **({"stream": hi.cuda_stream} if k else {}))
                for k in range(2)]
        for smp in smps:
            smp.wait()
        wall = time.perf_counter() - t0
        n = sum(int(smp._copy("n_steps", np.int64).sum()) for smp in smps)
        secs = [smp.seconds for smp in smps]
        for smp in smps:
            smp.close()
        return {"leapfrogs_per_s": n / max(secs), "leapfrogs_per_s_over_wall": n / wall, "job_s": secs, "wall_incl_setup_s": wall, "leapfrogs": n,
                "workload": "TWO independent radon jobs of 512 chains each (generated density), started together: each occupies half of the SIMDs; "
                            "rate = leapfrogs of both / the longer job's engine time"}

    def c3_torch():
        from nutpie_amd.radon import radon_model

        m = radon_model(device=env.device, use_graph=True)
        t0 = time.perf_counter()
        r = job_rate(m._make_sampler(settings(512, 100, 50), None, 1, None, None, None, None, store_draws=False), t0)
        r["workload"] = "radon, 512 chains, torch log-density evaluated EAGERLY (hand-derived gradient, HIP-graph replay) behind the batched device callback; bounded sample: tune 100 + draws 50"
        return r

    def c4():
        import ctypes

        fix = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libeight_schools.so"))
        m = hip.HostCallbackModel(10, ctypes.cast(fix.eight_schools_logp, ctypes.c_void_p).value)
        m.set_init("normal")
        t0 = time.perf_counter()
        smp = hip.PySampler(settings(256, 400, 1000, seed=21), m, device=env.device)
        mode = smp.host_mode
        r = job_rate(smp, t0)
        r["host_mode"] = mode
        r["workload"] = "eight schools (D = 10, non-centred), 256 chains, tune 400 + draws 1000; raw C logp callback on the host (the reference's signature), PCIe inclusive"
        bpl = 56.0 * 10
        r["roofline"] = {"bound": "hbm", "achieved": bpl * r["leapfrogs_per_s"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bpl * r["leapfrogs_per_s"] / 1e9 / HBM_PEAK_GBS,
                         "traffic": None, "note": "SURVEY.md 8d algorithmic bytes (56 D per leapfrog, opaque gradient) x leapfrogs/s: a 10-dimensional model moves nothing — the leg is bound by the "
                                                  "round trip to the host's callback (one rendezvous over PCIe per evaluation of a group of chains), not by any device resource"}
        r["divergences_note"] = "eight schools' funnel diverges now and then at target_accept 0.8 also in the non-centred form: the count is the sampler's, the same order in the CPU oracle's run beside it"
        return r

    cpu_s = 0.0 if args.no_cpu_baseline else min(6.0, args.cpu_seconds)

    def oracle_mod():
        import oracle

        oracle.build()
        return oracle

    def c3_cpu():
        # the CPU beside config 3: the oracle sampling the radon density as a raw C callback (tests/fixtures/radon_host.c: the same 85 counties /
        # 919 observations, D = 173, the reference's callback signature) — what nuts-rs does with a numba cfunc
        import ctypes

        from nutpie_amd.radon import synthetic_radon_data

        oracle = oracle_mod()
        host = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libradon_host.so"))
        host.radon_host_create.restype = ctypes.c_void_p
        host.radon_host_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        data = synthetic_radon_data()
        n = int(data["county_idx"].max()) + 1
        cty = np.ascontiguousarray(data["county_idx"], dtype=np.int32)
        fl, y = np.ascontiguousarray(data["floor"]), np.ascontiguousarray(data["log_radon"])
        hh = host.radon_host_create(n, len(y), cty.ctypes.data, fl.ctypes.data, y.ctypes.data)
        fn = ctypes.cast(host.radon_host_logp, ctypes.c_void_p).value

        def run(chains_, tune_, draws_, threads, tuned=False):
            return oracle.sample_callback(oracle.default_settings(seed=20260926, num_chains=chains_, num_tune=tune_, num_draws=draws_, n_threads=threads), 2 * n + 3, fn, user=hh, tuned=tuned)

        return cpu_leg(run, 512, 400, 1000, cpu_s, "the radon density as a raw C logp callback (tests/fixtures/radon_host.c)", tuned_run=lambda *a: run(*a, tuned=True))

    def c4_cpu():
        import ctypes

        oracle = oracle_mod()
        fix = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libeight_schools.so"))
        fn = ctypes.cast(fix.eight_schools_logp, ctypes.c_void_p).value

        def run(chains_, tune_, draws_, threads, tuned=False):
            return oracle.sample_callback(oracle.default_settings(seed=21, num_chains=chains_, num_tune=tune_, num_draws=draws_, n_threads=threads, init_kind=1), 10, fn, tuned=tuned)

        # (the port emulates the engine's 64-lane summation geometry on every dot product — at 10 dimensions that emulation IS its cost; the
        #  `tuned` entry, plain loops, is the CPU number to compare a 10-dimensional model with)
        return cpu_leg(run, 256, 400, 1000, cpu_s, "the same eight-schools C callback (tests/fixtures/eight_schools.c)", tuned_run=lambda *a: run(*a, tuned=True))

    def c2_dense():
        # BASELINE.json configs[1] read as a DENSE correlated Gaussian (SURVEY.md 8d variant (ii)) INSIDE the engine: the gradients of all chains
        # of an evaluation round are one fp64 GEMM on the matrix cores, hand-written (csrc/dense_tile.h), called from the middle of the
        # register-resident leaf (kernels.hip: DENSEG / dg_round) — one resident launch runs 256 rounds.  Measured like the headline: a short
        # warm-up (untimed), then a timed region of launches in the sampling phase; the whole bounded job of round 5's line beside it.
        from nutpie_amd.gaussian import dense_precision

        D, chains, tune = 1000, 1024, 60
        P = dense_precision(D)
        model = hip.DenseGaussianModel(P)
        E = 256
        K = 24
        smp = hip.PySampler(settings(chains, tune, 96, seed=1), model, device=env.device, store_draws=False, evals_per_launch=E, manual=True)
        mode = smp.host_mode
        warm = warm_up_to_sampling(smp, batch=4)
        smp.step(2)
        elapsed, leap, kernel_ms, _ = timed_launches(env, smp, K, tune + 96)
        smp.close()
        rate = leap / elapsed
        peak = hip.mfma_f64_rate(env.device)
        flops = 2.0 * D * D
        try:
            pmc = json.load(open(TRAFFIC_JSON)).get("config2ii_dense_resident")
        except OSError:
            pmc = None
        r = {"leapfrogs_per_s": rate, "timed_region_s": elapsed, "steps": K, "ms_per_step": 1000.0 * elapsed / K, "kernel_ms_per_launch": kernel_ms / K,
             "rounds_per_launch": E, "us_per_round": 1e6 * (kernel_ms / 1000.0) / (K * E), "useful_leapfrogs_per_round_and_chain": leap / (K * E * chains),
             "host_mode": mode, "waves_per_chain": 1,
             "warmup": {"leapfrogs": warm[0], "launches": warm[1], "kernel_ms": warm[2], "wall_s": warm[3]},
             "roofline": {"bound": "mfma", "kernel": "k_advance<callback,W=1,NV=8,REMOTE,DENSEG> (the launch-wide gradient GEMM inside the register-resident leaf)",
                          "achieved": flops * rate / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops * rate / 1e12 / peak,
                          "traffic": (pmc["bytes_per_leapfrog"] * leap / K) if pmc else None,
                          "hbm_measured": ({"bytes_per_leapfrog": pmc["bytes_per_leapfrog"], "GB/s": pmc["bytes_per_leapfrog"] * rate / 1e9, "frac_of_peak": pmc["bytes_per_leapfrog"] * rate / 1e9 / HBM_PEAK_GBS,
                                            "source": pmc["source"], "note": "the precision matrix streams through the dies' L2s every round (8 MB against 4 MB of L2): L2 misses that the Infinity Cache serves are in these counters"} if pmc else None),
                          "pmc_issuing_fraction_of_wave_cycles": (pmc or {}).get("issuing_fraction"), "pmc_waiting_fraction_of_wave_cycles": (pmc or {}).get("waiting_fraction"),
                          "flops_per_leapfrog": flops, "peak_datasheet_TFLOPs": 78.6, "frac_of_datasheet": flops * rate / 1e12 / 78.6,
                          "note": "achieved = 2 D^2 flop per leapfrog x leapfrogs/s of the timed region (the whole step: leaf + rendezvous + GEMM); peak = the fp64 matrix-core "
                                  "rate MEASURED on this device in this run (v_mfma_f64_16x16x4_f64 back to back on every SIMD: 96 cycles per instruction at 2.4 GHz — "
                                  "nphip_test_mfma_f64_rate; MI355X_MICROARCH.md lists no fp64 MFMA figure, the datasheet's 78.6 TFLOP/s is beside it)"},
             "workload": f"dense {D}-dim Gaussian (condition 1e4), {chains} chains, gradient = the engine's own fp64 MFMA GEMM inside the resident kernel; "
                         f"warm-up tune {tune} (untimed), then {K} launches of {E} evaluation rounds in the sampling phase"}
        # the bounded whole job of round 5's line (tune 30 + draws 10, job-level rate: tail and warm-up phases included), both forms of the model
        t0 = time.perf_counter()
        r["bounded_job"] = job_rate(hip.PySampler(settings(chains, 30, 10, seed=1), hip.DenseGaussianModel(P), device=env.device, store_draws=False), t0)
        t0 = time.perf_counter()
        r["bounded_job_launch_per_evaluation"] = job_rate(hip.PySampler(settings(chains, 30, 10, seed=1), hip.DenseGaussianModel(P), device=env.device, store_draws=False, host_persist=1), t0)
        if cpu_s > 0:
            oracle = oracle_mod()

            def run(chains_, tune_, draws_, threads):
                return oracle.sample_dense(oracle.default_settings(seed=1, num_chains=chains_, num_tune=tune_, num_draws=draws_, n_threads=threads), P)

            def run_t(chains_, tune_, draws_, threads):
                return oracle.sample_dense(oracle.default_settings(seed=1, num_chains=chains_, num_tune=tune_, num_draws=draws_, n_threads=threads), P, tuned=True)

            with_cpu(r, cpu_leg(run, chains, 8, 2, cpu_s, "the same dense Gaussian (oracle.sample_dense: the contract's fma chain per gradient element)", tuned_run=run_t))
        return r

    def c2_dense_torch():
        from nutpie_amd.gaussian import dense_gaussian_torch

        m = dense_gaussian_torch(1000, device=env.device)
        t0 = time.perf_counter()
        r = job_rate(m._make_sampler(settings(1024, 30, 10, seed=1), None, 1, None, None, None, None, store_draws=False), t0)
        r["workload"] = "dense 1000-dim Gaussian, 1024 chains, gradient = fp64 GEMM (rocBLAS via torch) behind the batched device callback (round 5's path, kept as the comparison); bounded sample: tune 30 + draws 10"
        return r

    leg("config3_radon_generated_density", c3_generated)
    leg("config3_radon_generated_density_low_rank", c3_low_rank)
    leg("config3_two_concurrent_512_chain_jobs", c3_two_jobs)
    leg("config3_radon_torch_density", c3_traced)
    leg("config3_radon_torch_density_eager", c3_torch)
    leg("config4_eight_schools_host_callback", c4)
    leg("config2ii_dense_gaussian_in_engine_mfma", c2_dense)
    leg("config2ii_dense_gaussian_gemm_callback", c2_dense_torch)
    if cpu_s > 0:
        # the CPU path timed beside every leg (rank 0, same run): one baseline per workload, attached to each leg that runs it
        try:
            cpu3 = c3_cpu()
        except Exception as e:
            cpu3 = {"error": repr(e)}
        for k in list(out):
            if k.startswith("config3_") and "error" not in out[k]:
                with_cpu(out[k], cpu3)
        try:
            cpu4 = c4_cpu()
        except Exception as e:
            cpu4 = {"error": repr(e)}
        if "error" not in out["config4_eight_schools_host_callback"]:
            with_cpu(out["config4_eight_schools_host_callback"], cpu4)
        d_ = out.get("config2ii_dense_gaussian_in_engine_mfma", {})
        if "cpu_baseline" in d_ and "error" not in out["config2ii_dense_gaussian_gemm_callback"]:
            with_cpu(out["config2ii_dense_gaussian_gemm_callback"], d_["cpu_baseline"])
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, argv))
    env = Env(args)
    hip, world, rank, device, dist = env.hip, env.world, env.rank, env.device, env.dist

    from nutpie_amd.gaussian import ar1_gaussian

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

    # ---- set-up, untimed: the whole warm-up (tune = 400 draws per chain); its rate is reported as `tuning_phase`
    tuning_phase = None
    if args.phase == "sampling":
        err = None
        try:
            leapfrogs, launches, kms, wall = warm_up_to_sampling(smp)
        except InvalidRegion as e:
            err = str(e)
        if env.all_max(1.0 if err else 0.0) > 0.0:   # (collective: every rank leaves, or none)
            raise SystemExit(err or "bench.py: another rank's chains finished during warm-up")
        env.sync()
        tuning_phase = {"leapfrogs": int(leapfrogs), "launches": launches, "wall_s": wall,
                        "leapfrogs_per_s_kernel_time": leapfrogs / (kms / 1e3), "note": "all chains through tune = 400 draws (untimed set-up of the "
                        "bench; includes initial points, step-size search, mass-matrix adaptation; wall time includes progress polling)"}
    smp.step(args.warmup)
    prog0 = smp.progress()
    tuning0 = sum(p.tuning for p in prog0)
    draws0 = sum(p.finished_draws for p in prog0)
    try:
        elapsed, leap, kernel_ms, prog = timed_launches(env, smp, K, num_tune + n_draws)
    except InvalidRegion as e:
        raise SystemExit(str(e))
    tuning1 = sum(p.tuning for p in prog)
    rows = env.all_rows([rank, device, leap, kernel_ms, elapsed])
    per_rank = [{"rank": int(t[0]), "device": int(t[1]), "leapfrogs": float(t[2]), "kernel_ms": float(t[3]), "elapsed_s": float(t[4])} for t in rows]
    elapsed = env.all_max(elapsed)
    leap, kernel_ms_sum = env.all_sum([leap, kernel_ms])
    kernel_ms = kernel_ms_sum / world
    draws_stored = int(sum(p.finished_draws for p in prog)) if store else 0
    draws_in_region = env.all_sum([float(sum(p.finished_draws for p in prog) - draws0)])[0]
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
        "roofline": roofline(args.dim, W, args.chains, leap_per_launch, avg_kernel_s, leap_per_draw=(leap / draws_in_region) if draws_in_region > 0 else None),
        "ranks": {"world": world, "backend": (dist.get_backend() if dist is not None else None),
                  "collective_ranks": (dist.get_world_size() if dist is not None else 1), "per_rank": per_rank,
                  "launched_by": "torch.distributed.run" if "RANK" in os.environ else "single process",
                  "note": "chains sharded by global chain id (rank r owns chains [r * chains_per_gpu, (r + 1) * chains_per_gpu)); no data-path "
                          "collective, only this report's gather and the timing all-reduce"},
    }
    if env.stub:
        out["data"] = "stub"
        out["stub"] = "engine replaced by a test stand-in: this line exercises bench.py's multi-rank plumbing and measures nothing"
    if tuning_phase is not None:
        out["tuning_phase"] = tuning_phase
    if not args.no_config5 and args.dim != args.config5_dim:
        try:
            out["config5_shard"] = config5_shard(env, args)   # (collective: every rank takes part)
        except InvalidRegion as e:    # (raised on every rank together: see timed_launches)
            out["config5_shard"] = {"error": repr(e)}
        except Exception as e:
            if world > 1:
                raise   # a rank that drops out of a collective must not leave the others waiting
            out["config5_shard"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_job and args.dim <= 2048 and not env.stub:
        dims = sorted(set(np.linspace(0, args.dim - 1, 12).astype(int).tolist() + [int(np.argmax(model.diag)), int(np.argmin(model.diag))]))
        out["job"] = run_job(hip, model, args, device, 0, dims, world)
    if rank == 0 and world == 1 and not args.no_other_configs and not env.stub:
        out["other_configs"] = other_configs(env, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not env.stub:
        tuned, port = cpu_baseline(model, args.seed, args.cpu_seconds)
        out["cpu_baseline"] = port
        out["gpu_over_cpu"] = out["value"] / port["value"]
        if tuned:
            out["cpu_baseline_tuned"] = tuned
            if "value" in tuned:
                out["gpu_over_cpu_tuned"] = out["value"] / tuned["value"]
    if rank == 0:
        # (RCCL prints a version banner through C stdio when its first communicator is created; flush it BEFORE the line, so
        #  that the JSON line is the last thing on stdout)
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if env.dist is not None:
        env.dist.barrier()
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
