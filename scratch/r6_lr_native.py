"""Round 6: the estimator as one kernel (nphip_low_rank_estimate) against the torch formulation — time per hand-in by number of chains
(radon's shape: D = 173, windows of 40 .. 256 draws, 32 basis draws), and the dense metrics of both."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import low_rank, _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
D = 173
for n, m in ((1, 80), (8, 80), (64, 80), (243, 80), (512, 80), (512, 256), (512, 40)):
    x = torch.randn(n, m, D, dtype=torch.float64, device=dev, generator=g)
    x[:, :, :3] *= 30.0
    gx = -x * torch.exp(torch.randn(D, dtype=torch.float64, device=dev, generator=g))
    for native in (False, True):
        low_rank.NATIVE_ESTIMATOR = native
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = low_rank.estimate_window(x, gx, None, 0, m, 1e-5, 100.0, basis_draws=low_rank.basis_draws_for(D))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"n={n:4d} m={m:4d} {'kernel' if native else 'torch '}: {dt * 1e3:7.2f} ms   columns used (mean) {float((out[2] != 1).sum(1).double().mean()):.2f}", flush=True)

# where the kernel's time goes: sweeps of the four eigenproblems and cycles (words 4096.. of a chain's scratch row)
low_rank.NATIVE_ESTIMATOR = True
for n, m in ((1, 80), (243, 80)):
    x = torch.randn(n, m, D, dtype=torch.float64, device=dev, generator=g)
    gx = -x * torch.exp(torch.randn(D, dtype=torch.float64, device=dev, generator=g))
    low_rank.estimate_window(x, gx, None, 0, m, 1e-5, 100.0, basis_draws=32)
    torch.cuda.synchronize()
    d = _lib.low_rank_estimate.last_scratch[:, 4096:4104].cpu().numpy()
    print(f"n={n}: sweeps (Gram, Cg, M', S') {d[:, :4].mean(0)}, cycles before the first eigenproblem {d[:, 4].mean():.0f}, in it {d[:, 5].mean():.0f}, kernel {d[:, 6].mean():.0f}")
