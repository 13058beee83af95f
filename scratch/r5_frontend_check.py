"""Round 5: the front-end's late additions (ordered vectors, the new densities) as GENERATED DEVICE CODE against the numpy evaluation of the same graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.special import gammaln
from nutpie_amd import symbolic as S


def ordinal():
    rng = np.random.default_rng(9)
    n, K = 60, 5
    xcov = rng.normal(size=n)
    ycat = rng.integers(0, K, n)
    m = S.Model()
    cut = m.param("cut", dim="cutpoint", size=K - 1, ordered=True, initval=[-1.5, -0.5, 0.5, 1.5])
    beta = m.param("beta")
    xd = m.data("x", xcov, dim="obs")
    lo_idx = m.index("lo_idx", np.clip(ycat - 1, 0, K - 2), dim="obs", into="cutpoint")
    hi_idx = m.index("hi_idx", np.clip(ycat, 0, K - 2), dim="obs", into="cutpoint")
    is_first = m.data("is_first", (ycat == 0).astype(np.float64), dim="obs")
    is_last = m.data("is_last", (ycat == K - 1).astype(np.float64), dim="obs")
    eta = beta * xd
    p_hi = is_last + (1.0 - is_last) * S.sigmoid(cut[hi_idx] - eta)
    p_lo = (1.0 - is_first) * S.sigmoid(cut[lo_idx] - eta)
    m.add_logp(S.log(p_hi - p_lo).sum() + S.normal_lpdf(cut, 0.0, 3.0).sum() + S.normal_lpdf(beta, 0.0, 2.0))
    return m


def densities():
    rng = np.random.default_rng(0)
    cnt = rng.poisson(4.0, 50).astype(float)
    ntr = cnt + rng.integers(0, 5, 50)
    m = S.Model()
    phi = m.param("phi", lower=0.0); c = m.param("c"); a = m.param("a", lower=0.0); b = m.param("b", lower=0.0)
    y = m.data("cnt", cnt, dim="obs"); lf = m.data("lf", gammaln(cnt + 1), dim="obs"); u = m.data("u", rng.uniform(0.1, 0.9, 50), dim="obs")
    nt = m.data("ntr", ntr, dim="obs"); lb = m.data("lb", gammaln(ntr + 1) - gammaln(cnt + 1) - gammaln(ntr - cnt + 1), dim="obs")
    m.add_logp(S.negative_binomial_log_lpmf(y, c, phi, lf).sum() + S.beta_lpdf(u, a, b).sum() + S.student_t_lpdf(y, a + 1.0, c, b).sum() + S.weibull_lpdf(u, a, b).sum()
               + S.laplace_lpdf(u, c, b).sum() + S.logistic_lpdf(u, c, b).sum() + S.inverse_gamma_lpdf(u, a, b).sum() + S.gamma_lpdf(u, a, b).sum()
               + S.binomial_logit_lpmf(y, nt, c, lb).sum())
    return m


if __name__ == "__main__":
    for name, make in (("ordinal regression with ordered cut points", ordinal), ("the new densities", densities)):
        cm = make().compile()
        if len(sys.argv) > 1 and sys.argv[1] == "build":
            print(name, cm.library_path())
            continue
        x = 0.4 * np.random.default_rng(1).normal(size=(16, cm.n_dim))
        lp, g = cm.logp_and_grad(x)
        lp0, g0 = cm.logp_and_grad_numpy(x)
        print(f"{name}: max |logp - numpy| / |logp| = {np.max(np.abs(lp - lp0) / np.abs(lp0)):.2e}, max |grad - numpy| / max |grad| = {np.max(np.abs(g - g0)) / np.max(np.abs(g0)):.2e}")
        if name.startswith("ordinal"):
            import nutpie_amd
            tr = nutpie_amd.sample(cm, chains=64, tune=200, draws=100, seed=3, progress_bar=False)
            cuts = tr.posterior.cut.values
            print("   sampled: cut points increasing in every draw:", bool(np.all(np.diff(cuts, axis=-1) > 0)), " divergences:", int(tr.sample_stats.diverging.values.sum()))
