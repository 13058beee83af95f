import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nutpie_amd
from nutpie_amd import low_rank as lr
import symbolic_models as zoo
orig = lr.estimate
saved = []
def wrapped(x, g, gamma, cutoff, k_max=lr.K_MAX):
    T = orig(x, g, gamma, cutoff, k_max)
    bad = ~(torch.isfinite(T.stds).all(1) & torch.isfinite(T.V).flatten(1).all(1) & torch.isfinite(T.d).all(1))
    if bad.any() and not saved:
        idx = torch.nonzero(bad)[:, 0]
        np.savez(os.path.join(ROOT, "gpurun_out", "lr_nan_window.npz"), x=x[idx].cpu().numpy(), g=g[idx].cpu().numpy())
        saved.append(1)
        print("non-finite metric for", idx.tolist(), "window", tuple(x.shape), "finite inputs", bool(torch.isfinite(x[idx]).all()), bool(torch.isfinite(g[idx]).all()),
              "stds finite", bool(torch.isfinite(T.stds[idx]).all()), "V finite", bool(torch.isfinite(T.V[idx]).all()), "d finite", bool(torch.isfinite(T.d[idx]).all()))
    return T
lr.estimate = wrapped
cm = zoo.radon().compile()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
nutpie_amd.sample(cm, chains=96, tune=300, draws=100, seed=11, progress_bar=False, adaptation="low_rank")
