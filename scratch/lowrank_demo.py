"""adaptation='low_rank' against 'diag' on a correlated Gaussian (the target of tests/test_gpu_api.py): leapfrogs per draw, wall
time, window switches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nutpie_amd
from nutpie_amd import low_rank as lr
for D, nd, fac in ((60, 3, 400.0), (500, 6, 400.0)):
    rng = np.random.default_rng(11)
    scales = np.exp(rng.normal(size=D)); B = rng.normal(size=(D, nd))
    Sigma = np.diag(scales**2) + fac * (scales[:, None] * B) @ (scales[:, None] * B).T
    mu = rng.normal(size=D) * 3
    P = np.linalg.inv(Sigma)
    def make_logp():
        Pt, mt = torch.as_tensor(P, device="cuda"), torch.as_tensor(mu, device="cuda")
        def f(x):
            z = x - mt; g = -(z @ Pt)
            return 0.5 * (z * g).sum(-1), g
        return f
    m = nutpie_amd.from_torchfunc(D, make_logp)
    for adaptation in ("diag", "low_rank"):
        t0 = time.perf_counter()
        tr = nutpie_amd.sample(m, adaptation=adaptation, chains=256, tune=500, draws=500, seed=3, progress_bar=False)
        dt = time.perf_counter() - t0
        st = tr.sample_stats
        x = tr.posterior.x.values.reshape(-1, D)
        err = np.abs(np.sqrt(np.diag(np.cov(x.T))) / np.sqrt(np.diag(Sigma)) - 1).max()
        print(f"D={D} ({nd} strong directions) {adaptation:8s}: {st.n_steps.values.mean():7.1f} leapfrogs per draw (depth {st.depth.values.mean():.2f}), "
              f"warm-up {tr.warmup_sample_stats.n_steps.values.sum() / 1e6:.2f} M leapfrogs, job {dt:.2f} s, max rel. sd error {err:.3f}")
