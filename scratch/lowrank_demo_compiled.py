"""scratch/lowrank_demo.py's correlated Gaussians (Sigma = diag(s^2) + 400 (sB)(sB)') written as EXPRESSIONS, so that the job runs
on the register-resident kernels of a compiled density — under "diag" and under the low-rank metric (kernels.hip part 7,
-DNPHIP_JIT_LR=1).  Woodbury: z' Sigma^-1 z = sum z^2 / s^2 - sum_k <a_k, z>^2 with nd data vectors a_k.
    python scratch/lowrank_demo_compiled.py build     # here (no GPU): hipcc the four libraries into the in-tree cache
    python scratch/lowrank_demo_compiled.py           # on the GPU box"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import symbolic as S


def target(D, nd, fac, seed=11):
    rng = np.random.default_rng(seed)
    scales = np.exp(rng.normal(size=D)); B = rng.normal(size=(D, nd))
    U = scales[:, None] * B
    Sigma = np.diag(scales**2) + fac * U @ U.T
    mu = rng.normal(size=D) * 3
    Dinv = 1.0 / scales**2
    Cm = np.linalg.inv(np.eye(nd) / fac + U.T @ (Dinv[:, None] * U))        # nd x nd
    A = (Dinv[:, None] * U) @ np.linalg.cholesky(Cm)                        # D x nd: columns a_k
    P = np.diag(Dinv) - A @ A.T
    assert np.allclose(P @ Sigma, np.eye(D), atol=1e-8)
    m = S.Model()
    x = m.param("x", dim="d", size=D)
    z = x - m.data("mu", mu, dim="d")
    quad = (z * z * m.data("dinv", Dinv, dim="d")).sum()
    for k in range(nd):
        t = (m.data(f"a{k}", A[:, k], dim="d") * z).sum()
        quad = quad - t * t
    m.add_logp(-0.5 * quad)
    return m.compile(), Sigma


if __name__ == "__main__":
    build = len(sys.argv) > 1 and sys.argv[1] == "build"
    for D, nd, fac in ((60, 3, 400.0), (500, 6, 400.0)):
        cm, Sigma = target(D, nd, fac)
        if build:
            print(D, cm.library_path(False), cm.library_path(True))
            continue
        import nutpie_amd
        runs = [("diag", 500), ("low_rank", 500)] + ([("diag", 1000), ("diag", 2000), ("diag", 4000), ("low_rank", 1000)] if D == 500 else [])
        for adaptation, tune in runs:
            for rep in range(2):   # the second run: libraries loaded, rocSOLVER initialised
                t0 = time.perf_counter()
                tr = nutpie_amd.sample(cm, adaptation=adaptation, chains=256, tune=tune, draws=500, seed=3, progress_bar=False)
                dt = time.perf_counter() - t0
            st = tr.sample_stats
            x = tr.posterior.x.values.reshape(-1, D)
            err = np.abs(np.sqrt(np.diag(np.cov(x.T))) / np.sqrt(np.diag(Sigma)) - 1).max()
            print(f"D={D} ({nd} strong directions) {adaptation:8s} tune {tune:4d}: {st.n_steps.values.mean():7.1f} leapfrogs per draw (depth {st.depth.values.mean():.2f}), "
                  f"warm-up {tr.warmup_sample_stats.n_steps.values.sum() / 1e6:.2f} M leapfrogs, job {dt:.2f} s, max rel. sd error {err:.3f}")
