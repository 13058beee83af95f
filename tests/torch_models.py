"""Torch log-densities for the tracer's tests (``nutpie_amd.torch_trace``): what a user of ``from_torch_density`` writes, and what
PyTensor's ``mode="PYTORCH"`` linker emits for the reference's test models — element-wise arithmetic on slices of the flat vector,
advanced indexing for group effects, reductions, small matrix products, ``torch.distributions`` log-probabilities.

Every entry of ``ALL``: ``name -> () -> (ndim, density_fn, batched, shared_data)``; all data are seeded."""

from __future__ import annotations

import math

import numpy as np
import torch


def _t(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


def radon():
    """BASELINE config 3's density exactly as ``nutpie_amd.radon.radon_model`` writes it (``logp_only``), on the CPU"""
    from nutpie_amd.radon import _extend_zero_sum, synthetic_radon_data

    data = synthetic_radon_data()
    n = int(np.max(data["county_idx"])) + 1
    D = 2 * n + 3
    o_int, o_raw, o_lsd, o_floor, o_craw, o_lcsd, o_lsig = 0, 1, n, n + 1, n + 2, 2 * n + 1, 2 * n + 2
    cidx = _t(data["county_idx"], torch.long)
    floor, y = _t(data["floor"]), _t(data["log_radon"])
    n_obs = y.shape[0]

    def logp(x):
        intercept = x[:, o_int]
        raw = x[:, o_raw:o_raw + n - 1]
        lsd = x[:, o_lsd]
        fe = x[:, o_floor]
        craw = x[:, o_craw:o_craw + n - 1]
        lcsd = x[:, o_lcsd]
        lsig = x[:, o_lsig]
        sd, csd, sig = torch.exp(lsd), torch.exp(lcsd), torch.exp(lsig)
        ce = _extend_zero_sum(raw) * sd[:, None]
        cfe = _extend_zero_sum(craw) * csd[:, None]
        mu = intercept[:, None] + ce[:, cidx] + fe[:, None] * floor + cfe[:, cidx] * floor
        r = (y - mu) / sig[:, None]
        lp = -0.5 * (intercept / 10.0) ** 2 - 0.5 * (fe / 2.0) ** 2
        lp = lp - 0.5 * (raw * raw).sum(-1) - 0.5 * (craw * craw).sum(-1)
        lp = lp - 0.5 * sd * sd + lsd - 0.5 * csd * csd + lcsd - 0.5 * (sig / 1.5) ** 2 + lsig
        lp = lp - 0.5 * (r * r).sum(-1) - n_obs * lsig
        return lp

    return D, logp, True, {}


def logistic_regression():
    """Bernoulli-logit regression with a data matrix of 6 columns (batched: ``beta @ X^T``), Student-t priors"""
    rng = np.random.default_rng(11)
    N, K = 300, 6
    X = rng.normal(size=(N, K))
    beta0 = rng.normal(size=K)
    yv = (rng.uniform(size=N) < 1.0 / (1.0 + np.exp(-(X @ beta0 + 0.3)))).astype(np.float64)
    Xt, yt = _t(X), _t(yv)

    def logp(x):
        alpha, beta = x[:, 0], x[:, 1:]
        eta = alpha[:, None] + beta @ Xt.T
        ll = (yt * eta - torch.nn.functional.softplus(eta)).sum(-1)
        prior = -2.5 * torch.log1p(beta ** 2 / 4.0).sum(-1) - 0.5 * (alpha / 5.0) ** 2
        return ll + prior

    return K + 1, logp, True, {}


def linear_regression_unbatched():
    """``x[D] -> scalar``: ``X @ beta`` (the design-matrix form), ``torch.distributions`` log-probabilities, shared data by name"""
    rng = np.random.default_rng(5)
    N, K = 200, 4
    X = rng.normal(size=(N, K))
    yv = X @ rng.normal(size=K) + 0.7 * rng.normal(size=N)

    def logp(x, X, y):
        beta, log_sigma = x[:K], x[K]
        sigma = torch.exp(log_sigma)
        mu = X @ beta
        ll = torch.distributions.Normal(mu, sigma).log_prob(y).sum()
        prior = torch.distributions.Normal(0.0, 2.0).log_prob(beta).sum() + torch.distributions.HalfNormal(1.0).log_prob(sigma) + log_sigma
        return ll + prior

    return K + 1, logp, False, {"X": X, "y": yv}


def eight_schools():
    """non-centred eight schools; the HalfCauchy prior through atan-free log1p, ``stack`` / ``cat`` of pieces"""
    y = _t([28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0])
    sigma = _t([15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0])

    def logp(x):
        mu, log_tau, eta = x[:, 0], x[:, 1], x[:, 2:]
        tau = torch.exp(log_tau)
        theta = mu[:, None] + tau[:, None] * eta
        z = (y - theta) / sigma
        lp = -0.5 * (z * z).sum(-1) - 0.5 * (eta * eta).sum(-1) - 0.5 * (mu / 5.0) ** 2
        lp = lp - torch.log1p((tau / 5.0) ** 2) + log_tau
        return lp

    return 10, logp, True, {}


def funnel_whole_vector():
    """Neal's funnel written on the WHOLE vector (``x * w`` before slicing): the tracer's one-parameter mode"""
    D = 12
    w = _t(np.linspace(0.5, 1.5, D))

    def logp(x):
        zs = x * w
        v, rest = zs[:, 0], zs[:, 1:]
        return -v * v / 18.0 - 0.5 * (rest * rest * torch.exp(-v)[:, None]).sum(-1) - 0.5 * (D - 1) * v

    return D, logp, True, {}


def gamma_poisson():
    """Poisson counts with a Gamma-distributed rate per group and a free shape: ``lgamma`` of a parameter (gradient: digamma),
    ``index_add`` for the per-group sums, ``clamp`` / ``where`` / ``abs``"""
    rng = np.random.default_rng(3)
    G, N = 7, 120
    grp = rng.integers(0, G, size=N)
    counts = rng.poisson(3.0, size=N).astype(np.float64)
    gt, ct = _t(grp, torch.long), _t(counts)
    lf = torch.lgamma(ct + 1.0)

    def logp(x):
        log_a, log_rate = x[0], x[1:1 + G]
        a = torch.exp(log_a)
        rate = torch.exp(log_rate)
        ll = (ct * log_rate[gt] - rate[gt] - lf).sum()
        # Gamma(a, 1) prior on the rates, Laplace prior on log a, a soft barrier written with clamp and where
        prior = ((a - 1.0) * log_rate - rate).sum() - G * torch.lgamma(a) + log_rate.sum() - torch.abs(log_a)
        tot = torch.zeros(G, dtype=x.dtype).index_add(0, gt, rate[gt])
        barrier = -0.01 * torch.clamp(tot - 50.0, min=0.0).sum() - torch.where(log_a > 2.0, (log_a - 2.0) ** 2, torch.zeros_like(log_a))
        return ll + prior + barrier

    return 1 + G, logp, False, {}


def matrix_factor():
    """a 2-D parameter (``reshape`` of a slice), a product with a data matrix on the left, a sum along one axis, ``tanh`` / ``erf``"""
    rng = np.random.default_rng(9)
    J, K, N = 3, 4, 25
    A = rng.normal(size=(N, J))
    Y = rng.normal(size=(N, K))
    At, Yt = _t(A), _t(Y)

    def logp(x):
        W = x[: J * K].reshape(J, K)
        b = x[J * K: J * K + K]
        pred = torch.tanh(At @ W + b)
        res = Yt - pred
        col = (res * res).sum(0)                     # per column
        return -0.5 * col.sum() - 0.5 * (W * W).sum() - 0.5 * (b * b).sum() + torch.log(0.5 * (1.0 + torch.erf(b / math.sqrt(2.0)))).sum() * 0.1

    return J * K + K, logp, False, {}


ALL = {
    "radon": radon,
    "logistic_regression": logistic_regression,
    "linear_regression_unbatched": linear_regression_unbatched,
    "eight_schools": eight_schools,
    "funnel_whole_vector": funnel_whole_vector,
    "gamma_poisson": gamma_poisson,
    "matrix_factor": matrix_factor,
}


def autograd(density_fn, x, batched: bool, shared):
    """(logp[N], grad[N, D]) of ``density_fn`` at the rows of ``x`` through ``torch.autograd``"""
    sh = {k: _t(v) if not isinstance(v, torch.Tensor) else v for k, v in shared.items()}
    xt = torch.tensor(np.asarray(x), dtype=torch.float64, requires_grad=True)
    lp = density_fn(xt, **sh) if batched else torch.stack([density_fn(row, **sh) for row in xt])
    (g,) = torch.autograd.grad(lp.sum(), xt)
    return lp.detach().numpy().reshape(-1), g.numpy()


def softmax_mixture():
    """a categorical likelihood through ``log_softmax`` of the whole logit vector, a two-component mixture through ``logsumexp``,
    ``amax`` as a soft barrier"""
    rng = np.random.default_rng(21)
    K = 6
    counts = _t(rng.integers(1, 30, size=K).astype(np.float64))
    obs = _t(rng.normal(size=40) + 1.5)

    def logp(x):
        logits, mu, log_w = x[:K], x[K:K + 2], x[K + 2:K + 4]
        ll = (counts * torch.log_softmax(logits, -1)).sum()
        comp = torch.log_softmax(log_w, 0)
        mix = torch.stack([torch.logsumexp(torch.stack([comp[0] - 0.5 * (o - mu[0]) ** 2, comp[1] - 0.5 * (o - mu[1]) ** 2]), 0) for o in obs[:8]]).sum()
        return ll + mix - 0.5 * (logits * logits).sum() - 0.05 * torch.amax(logits) - 0.5 * (mu * mu).sum() / 9.0 - 0.5 * (log_w * log_w).sum()

    return K + 4, logp, False, {}


ALL["softmax_mixture"] = softmax_mixture


def inplace_ops():
    """in-place operations and indexed assignments (functionalised by the tracer): ``y.mul_``, ``y += 1``, ``out[:3] = y``, ``out[3] = s``,
    ``tot[perm] += v`` with distinct indices, ``index_add_`` with repeated ones"""
    idx = torch.tensor([0, 2, 1, 2, 0, 1, 1])
    perm = torch.tensor([2, 0, 1])

    def logp(x):
        a, b = x[:7], x[7:10]
        tot = torch.ones(3, dtype=x.dtype)
        tot[perm] += b * b
        tot2 = torch.zeros(3, dtype=x.dtype)
        tot2.index_add_(0, idx, a)
        y = b.clone()
        y.mul_(2.0)
        y += 1.0
        out = torch.empty(4, dtype=x.dtype)
        out[:3] = y
        out[3] = a.sum()
        return -(tot * b).sum() - 0.5 * (tot2 * tot2).sum() - 0.5 * (out * out).sum() - 0.5 * (a * a).sum()

    return 10, logp, False, {}


ALL["inplace_ops"] = inplace_ops


def multinomial_logit():
    """a categorical regression: ``log_softmax`` along the class axis of [observations, classes] logits, a 2-D coefficient matrix with
    its last class pinned to zero (``cat`` with a constant), ``logsumexp`` / ``amax`` along an axis"""
    rng = np.random.default_rng(17)
    N, P, K = 60, 3, 4
    X = _t(rng.normal(size=(N, P)))
    ycls = torch.as_tensor(rng.integers(0, K, size=N))
    onehot = torch.nn.functional.one_hot(ycls, K).to(torch.float64)

    def logp(x):
        W = x[: P * (K - 1)].reshape(P, K - 1)
        b = x[P * (K - 1):]
        eta = torch.cat([X @ W + b, torch.zeros(N, 1, dtype=x.dtype)], 1)           # [N, K]
        ll = (onehot * torch.log_softmax(eta, -1)).sum()
        reg = -0.01 * torch.logsumexp(eta, 1).sum() - 0.01 * torch.amax(eta, 1).sum()
        return ll + reg - 0.5 * (W * W).sum() - 0.5 * (b * b).sum()

    return P * (K - 1) + (K - 1), logp, False, {}


ALL["multinomial_logit"] = multinomial_logit


def bounds_checked():
    """what a compiled PyMC logp looks like: every term guarded by its parameter check (``where(sigma > 0, logp, -inf)``, ``switch`` with a
    constant array of -inf), ``full_like`` / ``broadcast_to`` allocations, a ``set_subtensor`` (``index_put`` with distinct indices)"""
    rng = np.random.default_rng(4)
    yv = _t(rng.normal(size=30) * 1.3 + 0.4)

    def logp(x):
        mu, log_sigma, nu_raw = x[0], x[1], x[2]
        sigma = torch.exp(log_sigma)
        nu = 2.0 + torch.nn.functional.softplus(nu_raw)
        z = (yv - mu) / sigma
        t_lp = torch.lgamma(0.5 * (nu + 1.0)) - torch.lgamma(0.5 * nu) - 0.5 * torch.log(nu * math.pi) - log_sigma - 0.5 * (nu + 1.0) * torch.log1p(z * z / nu)
        t_lp = torch.where(sigma > 0, t_lp, torch.full_like(t_lp, -math.inf))
        t_lp = torch.where(torch.broadcast_to(nu > 0, t_lp.shape), t_lp, -math.inf)
        prior = torch.zeros(3, dtype=x.dtype)
        prior[torch.tensor([2, 0, 1])] = torch.stack([-0.5 * nu_raw ** 2, -0.5 * (mu / 5.0) ** 2, -0.5 * log_sigma ** 2 + log_sigma])
        return t_lp.sum() + prior.sum()

    return 3, logp, False, {}


ALL["bounds_checked"] = bounds_checked


def ordered_logistic():
    """an ordinal regression: cut points through PyMC's ``ordered`` transform (first raw value, then a ``cumsum`` of exponentials), category
    probabilities as differences of sigmoids, one ``index`` per observation"""
    rng = np.random.default_rng(8)
    N, K = 80, 5
    xcov = _t(rng.normal(size=N))
    ycat = torch.as_tensor(rng.integers(0, K, size=N))

    def logp(x):
        beta, raw = x[0], x[1:K]
        cuts = torch.cat([raw[:1], raw[:1] + torch.cumsum(torch.exp(raw[1:]), 0)])          # K - 1 increasing cut points
        eta = beta * xcov
        cdf = torch.sigmoid(cuts[None, :] - eta[:, None])                                       # [N, K - 1]
        p = torch.cat([cdf, torch.ones(N, 1, dtype=x.dtype)], 1) - torch.cat([torch.zeros(N, 1, dtype=x.dtype), cdf], 1)
        ll = torch.log(p[torch.arange(N), ycat]).sum()
        return ll - 0.5 * beta * beta - 0.5 * (raw * raw).sum() / 4.0 + raw[1:].sum()

    return K, logp, False, {}


ALL["ordered_logistic"] = ordered_logistic


# ---- models that only the CPU tests use (tests/test_torch_trace_cpu.py) — none at the moment
CPU_ONLY = {}


def negbin_and_pairwise():
    """``torch.distributions.NegativeBinomial`` / ``Categorical`` (an ``eq`` against a scalar, masks from ``ne``), ``logaddexp`` of two
    component densities, ``linalg.norm`` of a block, ``var`` / ``std`` of another (with and without Bessel's correction, along an axis), a
    ``MultivariateNormal`` with a constant ``scale_tril``"""
    rng = np.random.default_rng(33)
    N = 30
    X = _t(rng.normal(size=(N, 3)))
    cnt = _t(rng.poisson(3.0, N).astype(np.float64))
    y = _t(rng.normal(size=N))
    cls = torch.as_tensor(rng.integers(0, 3, N))
    dist = torch.distributions

    def logp(x):
        beta, b0, r, mu, z = x[:3], x[3], x[4], x[5:7], x[7:13]
        ll = dist.NegativeBinomial(r.exp(), logits=X @ beta + b0).log_prob(cnt).sum()
        ll = ll + torch.logaddexp(dist.Normal(mu[0], 1.0).log_prob(y), dist.Normal(mu[1], 1.0).log_prob(y) - 0.3).sum()
        ll = ll + dist.Categorical(logits=x[13:16]).log_prob(cls).sum()
        zz = z.reshape(2, 3)
        ll = ll - torch.linalg.norm(z) - 0.1 * torch.linalg.vector_norm(zz, ord=1) - zz.var(dim=1).sum() - z.std() - zz.var(dim=0, correction=0).sum()
        ll = ll + torch.where(cnt != 3.0, cnt * 0.01 * b0, -0.02 * b0 * b0).sum()
        # a multivariate normal with a constant Cholesky factor: solve_triangular with a constant matrix = a product with its inverse
        ll = ll + dist.MultivariateNormal(x[16:19], scale_tril=L).log_prob(X).sum()
        return ll - 0.5 * (x * x).sum() / 4.0

    L = torch.tril(_t(rng.normal(size=(3, 3)))) * 0.3 + torch.eye(3, dtype=torch.float64)
    return 19, logp, False, {}


ALL["negbin_and_pairwise"] = negbin_and_pairwise
