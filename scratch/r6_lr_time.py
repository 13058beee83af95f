"""Round 6: where the low-rank estimator's time goes (radon's shape: 512 chains, D = 173, windows of 32 .. 256 draws)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import low_rank
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for n, m in ((512, 40), (512, 80), (512, 256), (64, 80), (8, 80)):
    D = 173
    x = torch.randn(n, m, D, dtype=torch.float64, device=dev, generator=g)
    gx = -x * torch.exp(torch.randn(D, dtype=torch.float64, device=dev, generator=g))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        T = low_rank.estimate(x, gx, 1e-5, 100.0, basis_draws=low_rank.basis_draws_for(D))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"n={n} m={m}: estimate {dt * 1e3:.1f} ms", flush=True)
n, m, D = 512, 80, 173
x = torch.randn(n, m, D, dtype=torch.float64, device=dev, generator=g); gx = -x * 2.0
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    low_rank.estimate(x, gx, 1e-5, 100.0, basis_draws=low_rank.basis_draws_for(D)); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=10, max_name_column_width=60))
