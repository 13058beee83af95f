"""Effective sample size (bulk ESS) — needed for the ESS/s half of the headline metric.

ArviZ is absent from the build image, so this restates the published estimator
(Vehtari, Gelman, Simpson, Carpenter, Bürkner 2021, "Rank-normalization, folding, and
localization"; the same algorithm ArviZ's ``ess(method="bulk")`` implements): rank-normalise
over the pooled draws, split every chain in half, FFT autocovariance, Geyer's initial
monotone positive sequence.  Works on numpy arrays or, for large traces, on torch tensors
(any device).
"""

from __future__ import annotations

import numpy as np


def _autocov_fft(x):
    """x: [..., n] -> biased autocovariance [..., n] (normalised by n)."""
    n = x.shape[-1]
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=-1, keepdims=True)
    f = np.fft.rfft(xc, n=m, axis=-1)
    ac = np.fft.irfft(f * np.conj(f), n=m, axis=-1)[..., :n]
    return ac / n


def _z_scale(x):
    """Rank-normalise over all elements (Blom offsets), as ArviZ's ``_z_scale``."""
    from scipy.special import ndtri
    from scipy.stats import rankdata

    r = rankdata(x.reshape(-1), method="average").reshape(x.shape)
    return ndtri((r - 0.375) / (x.size + 0.25))


def _split(x):
    n = x.shape[1] // 2
    return np.concatenate([x[:, :n], x[:, -n:]], axis=0) if x.shape[1] % 2 == 0 else np.concatenate([x[:, :n], x[:, n + 1:]], axis=0)


def ess_from_chains(x):
    """Geyer ESS for x[chain, draw] (no rank normalisation / splitting)."""
    x = np.asarray(x, dtype=np.float64)
    n_chain, n_draw = x.shape
    if n_draw < 4:
        return float("nan")
    acov = _autocov_fft(x)
    chain_mean = x.mean(axis=1)
    mean_var = acov[:, 0].mean() * n_draw / (n_draw - 1.0)
    var_plus = mean_var * (n_draw - 1.0) / n_draw
    if n_chain > 1:
        var_plus += chain_mean.var(ddof=1)
    if not np.isfinite(var_plus) or var_plus <= 0:
        return float("nan")
    rho = np.zeros(n_draw)
    rho[0] = 1.0
    rho[1] = 1.0 - (mean_var - acov[:, 1].mean()) / var_plus
    t = 1
    rho_even, rho_odd = 1.0, rho[1]
    while t < n_draw - 3 and (rho_even + rho_odd) > 0.0:
        rho_even = 1.0 - (mean_var - acov[:, t + 1].mean()) / var_plus
        rho_odd = 1.0 - (mean_var - acov[:, t + 2].mean()) / var_plus
        if rho_even + rho_odd >= 0:
            rho[t + 1] = rho_even
            rho[t + 2] = rho_odd
        t += 2
    max_t = t - 2
    if rho_even > 0:
        rho[max_t + 1] = rho_even
    # initial monotone sequence
    t = 1
    while t <= max_t - 2:
        if rho[t + 1] + rho[t + 2] > rho[t - 1] + rho[t]:
            rho[t + 1] = (rho[t - 1] + rho[t]) / 2.0
            rho[t + 2] = rho[t + 1]
        t += 2
    n_total = n_chain * n_draw
    tau = -1.0 + 2.0 * rho[: max_t + 1].sum() + rho[max_t + 1 : max_t + 2].sum()
    tau = max(tau, 1.0 / np.log10(n_total))
    return n_total / tau


def ess_bulk(x):
    """Bulk ESS of one scalar quantity, x[chain, draw]."""
    x = np.asarray(x, dtype=np.float64)
    return ess_from_chains(_z_scale(_split(x)))


def ess_bulk_min(draws, dims=None):
    """min over the selected dimensions of the bulk ESS of draws[chain, draw, dim]."""
    draws = np.asarray(draws)
    idx = range(draws.shape[2]) if dims is None else dims
    vals = np.array([ess_bulk(draws[:, :, d]) for d in idx])
    return float(np.nanmin(vals)), vals
