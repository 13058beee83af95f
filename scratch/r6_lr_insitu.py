"""Round 6: the estimator kernel on REAL windows (radon, 512 chains: the trace of a finished low-rank job) — time and Jacobi sweeps."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip, low_rank
from nutpie_amd.radon import radon_symbolic_model
from nutpie_amd.distributed import device_tensor
m = radon_symbolic_model().compile()
s = hip.PyNutsSettings.LowRank(20260926)
s.update(num_tune=400, num_draws=1000, num_chains=512)
smp = low_rank.make_sampler(m, s, None, 1, None, None, None, None)
smp.wait()
inner = smp._inner
n, T, D = inner.num_chains, inner.total_draws, inner.dim
draws = device_tensor(inner.device_ptr("draws"), (n, T, D), "float64", 0)
grads = device_tensor(inner.device_ptr("gradient"), (n, T, D), "float64", 0)
for (lo, hi) in ((60, 119), (123, 199), (199, 240), (240, 340)):
    for nch in (1, 64, 243, 512):
        ch = None if nch == n else np.arange(nch)
        for native in (False, True):
            low_rank.NATIVE_ESTIMATOR = native
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = low_rank.estimate_window(draws, grads, ch, lo, hi, 1e-5, 100.0, basis_draws=32)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            extra = ""
            if native:
                d = hip.low_rank_estimate.last_scratch[:, 4096:4104].cpu().numpy()
                extra = f"   sweeps (Gram, Cg, M', S') {np.round(d[:, :4].mean(0), 1)}  kernel cycles {d[:, 6].mean():.0f}"
            print(f"window [{lo}, {hi}) chains {nch:4d} {'kernel' if native else 'torch '}: {dt * 1e3:6.2f} ms{extra}", flush=True)
smp.close()
