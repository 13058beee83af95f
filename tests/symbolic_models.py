"""Small models written with nutpie_amd.symbolic, shared by the CPU and the GPU tests of the front-end."""

import numpy as np

from nutpie_amd import symbolic as S


def radon(data=None):
    """Config 3's varying-intercept / varying-slope model written with the front-end (it lives in the package: bench.py runs it)."""
    from nutpie_amd.radon import radon_symbolic_model

    return radon_symbolic_model(data)


def logistic(seed=3, n_obs=500, n_group=7, n_item=11):
    """Two crossed random effects (two different groupings of the observations), Bernoulli likelihood, one covariate."""
    rng = np.random.default_rng(seed)
    group = rng.integers(0, n_group, n_obs)
    item = rng.integers(0, n_item, n_obs)
    xcov = rng.normal(size=n_obs)
    eta = 0.3 + 0.8 * xcov + 0.5 * rng.normal(size=n_group)[group] + 0.7 * rng.normal(size=n_item)[item]
    yy = (rng.uniform(size=n_obs) < 1.0 / (1.0 + np.exp(-eta))).astype(np.float64)
    m = S.Model()
    m.dim("group", n_group)
    m.dim("item", n_item)
    b0 = m.param("b0")
    b1 = m.param("b1")
    sg = m.param("sigma_group", lower=0.0)
    si = m.param("sigma_item", lower=0.0)
    g = m.param("group_raw", dim="group")
    it = m.param("item_effect", dim="item")
    y = m.data("y", yy, dim="obs")
    xc = m.data("x", xcov, dim="obs")
    gi = m.index("group_idx", group, dim="obs", into="group")
    ii = m.index("item_idx", item, dim="obs", into="item")
    m.add_logp(S.normal_lpdf(b0, 0.0, 2.0) + S.normal_lpdf(b1, 0.0, 2.0))
    m.add_logp(S.halfnormal_lpdf(sg, 1.0) + S.halfnormal_lpdf(si, 1.0))
    m.add_logp(S.normal_lpdf(g, 0.0, 1.0).sum())              # non-centred
    m.add_logp(S.normal_lpdf(it, 0.0, si).sum())              # centred
    m.add_logp(S.bernoulli_logit_lpmf(y, b0 + b1 * xc + (g * sg)[gi] + it[ii]).sum())
    m.deterministic("group_effect", g * sg)
    return m


def poisson_offsets(seed=5, n_obs=300, n_site=40):
    """Poisson counts with a site effect read straight from the parameter vector, a per-site data column gathered to the
    observations, a Student-t prior and a scalar data value that ``with_data`` can change."""
    from scipy.special import gammaln

    rng = np.random.default_rng(seed)
    site = rng.integers(0, n_site, n_obs)
    expo = rng.uniform(0.5, 2.0, size=n_site)
    counts = rng.poisson(expo[site] * np.exp(0.4 + 0.3 * rng.normal(size=n_site)[site])).astype(np.float64)
    m = S.Model()
    m.dim("site", n_site)
    a = m.param("a")
    tau = m.param("tau", lower=0.0)
    u = m.param("u", dim="site")
    y = m.data("y", counts, dim="obs")
    lf = m.data("log_fact", gammaln(counts + 1.0), dim="obs")
    le = m.data("log_exposure", np.log(expo), dim="site")
    prior_scale = m.data("prior_scale", 1.5)
    si = m.index("site_idx", site, dim="obs", into="site")
    m.add_logp(S.normal_lpdf(a, 0.0, prior_scale))
    m.add_logp(S.halfnormal_lpdf(tau, 1.0))
    m.add_logp(S.student_t_lpdf(u, 4.0, 0.0, tau).sum())
    m.add_logp(S.poisson_log_lpmf(y, a + u[si] + le[si], lf).sum())
    return m


def regression(seed=9, n_obs=400, n_coef=6, n_group=9):
    """A design matrix: ``X @ beta`` with shrunken coefficients ``beta = raw * tau`` (the elements of a COMPUTED vector feed the
    predictor), a group intercept, Normal likelihood."""
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n_obs, n_coef))
    group = rng.integers(0, n_group, n_obs)
    beta_true = np.array([1.0, -0.5, 0.0, 0.25, 0.0, 2.0])[:n_coef]
    yy = X @ beta_true + 0.3 * rng.normal(size=n_group)[group] + 0.5 * rng.normal(size=n_obs)
    m = S.Model()
    m.dim("group", n_group)
    tau = m.param("tau", lower=0.0)
    raw = m.param("beta_raw", dim="coef", size=n_coef)
    sg = m.param("sigma_group", lower=0.0)
    a = m.param("a", dim="group")
    sig = m.param("sigma", lower=0.0)
    Xm = m.matrix("X", X, dim="obs", cols="coef")
    y = m.data("y", yy, dim="obs")
    gi = m.index("group_idx", group, dim="obs", into="group")
    beta = raw * tau
    m.deterministic("beta", beta)
    m.add_logp(S.halfnormal_lpdf(tau, 1.0) + S.halfnormal_lpdf(sg, 1.0) + S.halfnormal_lpdf(sig, 1.0))
    m.add_logp(S.normal_lpdf(raw, 0.0, 1.0).sum() + S.normal_lpdf(a, 0.0, sg).sum())
    m.add_logp(S.normal_lpdf(y, Xm @ beta + a[gi], sig).sum())
    return m


def plain_regression(seed=4, n_obs=250, n_coef=3):
    """``X @ beta`` with the coefficients read straight from the parameter vector."""
    rng = np.random.default_rng(seed)
    X = np.column_stack([np.ones(n_obs), rng.normal(size=(n_obs, n_coef - 1))])
    yy = X @ np.array([0.5, 1.5, -1.0])[:n_coef] + 0.7 * rng.normal(size=n_obs)
    m = S.Model()
    beta = m.param("beta", dim="coef", size=n_coef)
    sig = m.param("sigma", lower=0.0)
    Xm = m.matrix("X", X, dim="obs", cols="coef")
    y = m.data("y", yy, dim="obs")
    m.add_logp(S.normal_lpdf(beta, 0.0, 5.0).sum() + S.halfnormal_lpdf(sig, 2.0))
    m.add_logp(S.normal_lpdf(y, Xm @ beta, sig).sum())
    return m


def eight_schools():
    """The classic, non-centred: data on a parameter's own dimension, a HalfCauchy scale, an interval-bounded and an upper-bounded
    nuisance parameter with their Jacobians, Cauchy / Gamma / Exponential / LogNormal terms."""
    y = np.array([28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0])
    sd = np.array([15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0])
    m = S.Model()
    mu = m.param("mu")
    tau = m.param("tau", lower=0.0)
    th = m.param("theta_raw", dim="school", size=8)
    w = m.param("w", lower=-1.0, upper=3.0)
    u = m.param("u", upper=2.0)
    r = m.param("rates", dim="school", lower=0.5, upper=4.0)
    yy = m.data("y", y, dim="school")
    ss = m.data("sd", sd, dim="school")
    m.add_logp(S.normal_lpdf(mu, 0.0, 5.0) + S.halfcauchy_lpdf(tau, 5.0) + S.normal_lpdf(th, 0.0, 1.0).sum())
    m.add_logp(S.normal_lpdf(yy, mu + tau * th, ss).sum())
    m.add_logp(S.cauchy_lpdf(w, 0.5, 1.5) + S.exponential_lpdf(2.0 - u, 0.7) + S.gamma_lpdf(r, 2.5, 1.3).sum() + S.lognormal_lpdf(r, 0.1 * w, 0.8).sum())
    m.deterministic("theta", mu + tau * th)
    return m


def nested(seed=21, n_obs=350, n_group=12, n_region=4):
    """Three levels: observations in groups in regions.  The region effect reaches the observations through TWO gathers
    (region -> group -> observation), so its gradient is a segment sum of a segment sum; one parameter is used both directly and
    through a gather."""
    rng = np.random.default_rng(seed)
    region_of_group = rng.integers(0, n_region, n_group)
    group = rng.integers(0, n_group, n_obs)
    yy = 0.5 + 0.8 * rng.normal(size=n_region)[region_of_group][group] + 0.4 * rng.normal(size=n_group)[group] + 0.6 * rng.normal(size=n_obs)
    m = S.Model()
    m.dim("region", n_region)
    m.dim("group", n_group)
    mu = m.param("mu")
    sr = m.param("sigma_region", lower=0.0)
    sg = m.param("sigma_group", lower=0.0)
    sig = m.param("sigma", lower=0.0)
    r = m.param("region_raw", dim="region")
    g = m.param("group_raw", dim="group")
    y = m.data("y", yy, dim="obs")
    rg = m.index("region_of_group", region_of_group, dim="group", into="region")
    gi = m.index("group_idx", group, dim="obs", into="group")
    group_effect = (r * sr)[rg] + g * sg            # on "group"
    m.deterministic("group_effect", group_effect)
    m.add_logp(S.normal_lpdf(mu, 0.0, 3.0) + S.halfnormal_lpdf(sr, 1.0) + S.halfnormal_lpdf(sg, 1.0) + S.halfnormal_lpdf(sig, 1.0))
    m.add_logp(S.normal_lpdf(r, 0.0, 1.0).sum() + S.normal_lpdf(g, 0.0, 1.0).sum())
    m.add_logp(S.normal_lpdf(y, mu + group_effect[gi], sig).sum())
    m.add_logp((-0.01 * (g * g)).sum())             # (the same parameter once more, directly)
    return m


def scalar_only():
    """No dimension at all: a banana in two scalars."""
    m = S.Model()
    a = m.param("a")
    b = m.param("b", lower=1.0)
    m.add_logp(S.normal_lpdf(a, 0.0, 1.0) + S.normal_lpdf(b, 1.0 + a * a, 0.5) + S.log1p(S.sqrt(b)) - S.softplus(a))
    return m


def store_extra():
    """The parameter kinds of the reference's test_pymc_model_store_extra (tests/test_pymc.py:303-349): a normal vector, a positive
    vector (log transform), a zero-sum vector and a Dirichlet (simplex transform) on a second dimension."""
    m = S.Model()
    m.dim("foo", 5)
    m.dim("bar", 4)
    a = m.param("a", dim="foo")
    b = m.param("b", dim="foo", lower=0.0)
    c = m.param("c", dim="foo", zero_sum=True)
    d = m.param("d", dim="bar", simplex=True)
    m.add_logp(S.normal_lpdf(a, 0.0, 1.0).sum() + S.halfnormal_lpdf(b, 1.0).sum() + (-0.5 * (c * c)).sum() + S.dirichlet_lpdf(d, 1.0))
    return m


def dirichlet_counts(seed=6):
    """A Dirichlet-multinomial: concentration 2.5, observed counts as data on the simplex's dimension (the posterior is Dirichlet
    with concentration 2.5 + counts: a known answer for the simplex transform and its Jacobian)."""
    counts = np.array([12.0, 3.0, 0.0, 7.0, 30.0, 1.0])
    m = S.Model()
    m.dim("cat", 6)
    p = m.param("p", dim="cat", simplex=True)
    k = m.data("counts", counts, dim="cat")
    m.add_logp(S.dirichlet_lpdf(p, 2.5) + (k * S.log(p)).sum())
    m.deterministic("odds_first", S.elem(p, 0) / S.elem(p, 4))
    return m


def dims_model():
    """The reference's test_dims_model (tests/test_pymc.py:618-640): a zero-sum normal over the core dimension a of a value with
    dims (a, b), and a deterministic reported with its axes the other way round."""
    m = S.Model()
    m.dim("a", 3)
    m.dim("b", 5)
    z = m.param("zero_sum", dims=("a", "b"), zero_sum=True)
    m.add_logp((-0.5 * (z * z)).sum())
    m.deterministic("one_sum", z + 1.0 / 3.0, dims=("b", "a"))
    m.deterministic("col_sum", m.reduce(z, over="a"))
    return m


def no_prior():
    """The reference's test_pymc_model_no_prior (tests/test_pymc.py:210-222): a flat prior, one observation."""
    m = S.Model()
    a = m.param("a")
    m.add_logp(S.flat_lpdf(a) + S.normal_lpdf(0.0, a, 1.0))
    return m


def uniform_det():
    """The reference's test_trafo / test_det (tests/test_pymc.py:352-380): a Uniform(0, 1) vector (interval transform) and twice it."""
    m = S.Model()
    a = m.param("a", dim="two", size=2, lower=0.0, upper=1.0)
    m.deterministic("b", 2.0 * a)
    m.add_logp(0.0 * a.sum())      # (uniform on the interval: only the transform's Jacobian remains)
    return m


def collinear_regression(seed=2, n=400):
    """A regression on two nearly collinear columns: a strongly correlated posterior (what adaptation="low_rank" is for)."""
    rng = np.random.default_rng(seed)
    x1 = rng.normal(size=n)
    X = np.stack([x1, x1 + 0.02 * rng.normal(size=n), rng.normal(size=n)], 1)
    y = X @ np.array([1.0, -0.5, 2.0]) + 0.5 * rng.normal(size=n)
    m = S.Model()
    Xm = m.matrix("X", X, dim="obs", cols="coef")
    beta = m.param("beta", dim="coef")
    sig = m.param("sigma", lower=0.0)
    yy = m.data("y", y, dim="obs")
    m.add_logp(S.normal_lpdf(beta, 0.0, 10.0).sum() + S.halfnormal_lpdf(sig, 2.0) + S.normal_lpdf(yy, Xm @ beta, sig).sum())
    return m


def ordinal_regression(seed=9, n=60, K=5):
    """ordered cut points (``param(ordered=True)``: PyMC's ``ordered`` transform) of an ordinal regression"""
    rng = np.random.default_rng(seed)
    xcov = rng.normal(size=n)
    ycat = rng.integers(0, K, n)
    m = S.Model()
    cut = m.param("cut", dim="cutpoint", size=K - 1, ordered=True, initval=list(np.linspace(-1.5, 1.5, K - 1)))
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


def more_densities(seed=0, n=50):
    """the densities added at the end of round 5, shape parameters as model parameters (``lgamma`` / ``digamma`` in the kernel)"""
    from scipy.special import gammaln

    rng = np.random.default_rng(seed)
    cnt = rng.poisson(4.0, n).astype(float)
    ntr = cnt + rng.integers(0, 5, n)
    m = S.Model()
    phi, c, a, b = m.param("phi", lower=0.0), m.param("c"), m.param("a", lower=0.0), m.param("b", lower=0.0)
    y, lf, u = m.data("cnt", cnt, dim="obs"), m.data("lf", gammaln(cnt + 1), dim="obs"), m.data("u", rng.uniform(0.1, 0.9, n), dim="obs")
    nt, lb = m.data("ntr", ntr, dim="obs"), m.data("lb", gammaln(ntr + 1) - gammaln(cnt + 1) - gammaln(ntr - cnt + 1), dim="obs")
    m.add_logp(S.negative_binomial_log_lpmf(y, c, phi, lf).sum() + S.beta_lpdf(u, a, b).sum() + S.student_t_lpdf(y, a + 1.0, c, b).sum()
               + S.weibull_lpdf(u, a, b).sum() + S.laplace_lpdf(u, c, b).sum() + S.logistic_lpdf(u, c, b).sum() + S.inverse_gamma_lpdf(u, a, b).sum()
               + S.gamma_lpdf(u, a, b).sum() + S.binomial_logit_lpmf(y, nt, c, lb).sum())
    return m


# ---- the models of the reference's frozen documentation whose outputs tests/golden/reference_doc_step_sizes.json holds
def funnel():
    """docs/sample-stats.qmd:18-22: log_sigma ~ N(0, 1); x[5] ~ N(0, exp(log_sigma))"""
    m = S.Model()
    ls = m.param("log_sigma")
    x = m.param("x", dim="k", size=5)
    m.add_logp(S.normal_lpdf(ls, 0.0, 1.0) + S.normal_lpdf(x, 0.0, S.exp(ls)).sum())
    return m


def correlated_102d():
    """docs/sample-stats.qmd:141-145: x ~ N(0, 1); y ~ N(x, 0.01); z[100] ~ N(y, 1)"""
    m = S.Model()
    x, y = m.param("x"), m.param("y")
    z = m.param("z", dim="k", size=100)
    m.add_logp(S.normal_lpdf(x, 0.0, 1.0) + S.normal_lpdf(y, x, 0.01) + S.normal_lpdf(z, y, 1.0).sum())
    return m


def funnel_101d():
    """docs/nf-adapt.qmd:60-64: log_sigma ~ N(0, 1); x[100] ~ N(0, exp(log_sigma / 2))"""
    m = S.Model()
    ls = m.param("log_sigma")
    x = m.param("x", dim="k", size=100)
    m.add_logp(S.normal_lpdf(ls, 0.0, 1.0) + S.normal_lpdf(x, 0.0, S.exp(0.5 * ls)).sum())
    return m


DOC_MODELS = {"funnel_diag": funnel, "correlated_102d": correlated_102d, "funnel_101d": funnel_101d}

ALL = {"ordinal_regression": ordinal_regression, "more_densities": more_densities, "collinear_regression": collinear_regression, "store_extra": store_extra, "dirichlet_counts": dirichlet_counts, "dims_model": dims_model, "no_prior": no_prior, "uniform_det": uniform_det, "radon": radon, "logistic": logistic, "poisson_offsets": poisson_offsets, "scalar_only": scalar_only, "regression": regression,
       "plain_regression": plain_regression, "eight_schools": eight_schools, "nested": nested}

# models whose library is also built for the low-rank metric ahead of time (tests/test_gpu_density.py; __graft_entry__.build)
LOW_RANK = ("eight_schools", "radon", "store_extra", "collinear_regression")
