"""Round 6: nphip_low_rank_estimate on real windows, N dispatches (for rocprofv3: kernel statistics and PMC passes of k_lr_estimate).
A radon low-rank job of 512 chains runs first (its trace holds the windows); then the kernel alone: 243 chains, window [240, 340)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip, low_rank
from nutpie_amd.radon import radon_symbolic_model
from nutpie_amd.distributed import device_tensor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = radon_symbolic_model().compile()
s = hip.PyNutsSettings.LowRank(20260926)
s.update(num_tune=400, num_draws=1000, num_chains=512)
smp = low_rank.make_sampler(m, s, None, 1, None, None, None, None)
smp.wait()
inner = smp._inner
n, T, D = inner.num_chains, inner.total_draws, inner.dim
draws = device_tensor(inner.device_ptr("draws"), (n, T, D), "float64", 0)
grads = device_tensor(inner.device_ptr("gradient"), (n, T, D), "float64", 0)
ch = np.arange(243)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    low_rank.estimate_window(draws, grads, ch, 240, 340, 1e-5, 100.0, basis_draws=32)
torch.cuda.synchronize()
print(f"leapfrogs=0 launches={N} estimates of 243 chains: {(time.perf_counter() - t0) / N * 1e3:.2f} ms each")
smp.close()
