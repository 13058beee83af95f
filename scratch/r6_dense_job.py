"""Round 6: the dense Gaussian's resident kernel at BASELINE config 2 (ii) for the profiler — short warm-up, then K launches of 256 evaluation rounds in the
sampling phase (the region bench.py's config2ii leg times).  Prints `leapfrogs= launches=` of the timed launches (scratch/r5_pmc.sh reads it)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib
from nutpie_amd.gaussian import dense_precision

K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
P = dense_precision(1000)
s = _lib.PyNutsSettings.Diag(1)
s.update(num_tune=60, num_draws=96, num_chains=1024)
smp = _lib.PySampler(s, _lib.DenseGaussianModel(P), device=0, store_draws=False, evals_per_launch=256, manual=True)
while any(p.tuning for p in smp.progress()):
    smp.step(4)
smp.step(2)
n0 = sum(p.total_num_steps for p in smp.progress())
t0 = time.perf_counter()
done, launches, kms = smp.step(K)
dt = time.perf_counter() - t0
n = sum(p.total_num_steps for p in smp.progress()) - n0
print(f"mode={smp.host_mode} leapfrogs={n} launches={launches} seconds={dt:.4f} kernel_ms={kms:.2f} rate={n / dt / 1e6:.2f}M/s us_per_round={kms * 1e3 / (launches * 256):.1f}", flush=True)
if int(os.environ.get("NPHIP_DG_VARIANT", "0")) & 32:
    import ctypes as C
    out = (C.c_int64 * 16)()
    _lib.lib().nphip_sampler_profile(smp._h, out)
    pr = list(out)
    r = max(1, pr[12])
    h = pr[0:6]
    print("time between two rounds (the leaf), share of (wave, round) pairs: " + ", ".join(f"{lab} {v / max(1, sum(h)):.3f}" for lab, v in zip(("<5us", "<10us", "<15us", "<20us", "<30us", ">=30us"), h)), flush=True)
    print(f"per wave and round (whole job): wait-positions {pr[8] / r:.0f} cyc, GEMM {pr[9] / r:.0f} cyc, wait-gradients {pr[10] / r:.0f} cyc, round {pr[11] / r / 100:.1f} us", flush=True)
smp.close()
