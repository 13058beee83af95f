"""Round 6: nphip_low_rank_estimate against estimate() (torch) over random shapes — gross errors (NaN, wrong column counts, large differences)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip, low_rank as lr
from oracle import low_rank_estimator as ref
rng = np.random.default_rng(0)
worst = 0.0
for case in range(40):
    D = int(rng.choice([3, 7, 31, 32, 33, 63, 64, 65, 100, 173, 256, 300, 512]))
    m = int(rng.choice([12, 20, 31, 32, 33, 64, 100, 256]))
    n_dir = int(rng.integers(0, 4))
    cutoff = float(rng.choice([1.5, 2.0, 10.0, 100.0]))
    k_max = int(rng.choice([1, 4, 16]))
    n = 3
    B = rng.normal(size=(n, D, max(n_dir, 1))) * (n_dir > 0)
    scales = np.exp(rng.normal(size=(n, D)))
    Sigma = np.stack([np.diag(scales[c] ** 2) + 25.0 * (scales[c][:, None] * B[c]) @ (scales[c][:, None] * B[c]).T for c in range(n)])
    x = np.stack([rng.multivariate_normal(np.zeros(D), Sigma[c], size=m) for c in range(n)]) + 1.0
    g = -np.stack([np.linalg.solve(Sigma[c], (x[c] - 1.0).T).T for c in range(n)])
    pick = lr.basis_pick(m, 32)
    if not hip.low_rank_estimate_supported(D, m, len(pick), k_max) or len(pick) > D:
        print(f"case {case}: D={D} m={m}: not the kernel's shape"); continue
    xt, gt = torch.as_tensor(np.ascontiguousarray(x), device="cuda"), torch.as_tensor(np.ascontiguousarray(g), device="cuda")
    s2, V, lam, ku = (t.cpu().numpy() for t in hip.low_rank_estimate(xt, gt, None, 0, m, pick, 1e-5, cutoff, k_max))
    T = lr.estimate(xt, gt, 1e-5, cutoff, k_max=k_max, basis_draws=32)
    s2t, Vt, lamt = (t.cpu().numpy() for t in lr.metric_of(T))
    err = 0.0
    for c in range(n):
        a, b_ = ref.dense_metric(s2[c], V[c].T, lam[c]), ref.dense_metric(s2t[c], Vt[c].T, lamt[c])
        err = max(err, np.abs(a - b_).max() / np.abs(b_).max())
    kt = (lamt != 1).sum(1)
    flag = "" if (np.isfinite(s2).all() and np.isfinite(V).all() and (ku == kt).all() and err < 1e-2) else "   <-- LOOK"
    worst = max(worst, err)
    print(f"case {case}: D={D} m={m} b={len(pick)} dirs={n_dir} cutoff={cutoff} k_max={k_max}: columns {ku.tolist()} (torch {kt.tolist()}), max rel diff of the dense metric {err:.2e}{flag}", flush=True)
print("worst", worst)
