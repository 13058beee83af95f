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


# ---- all dimensions at once (torch; any device) -----------------------------------------------------------------------------
# BASELINE.json's ESS/s is the MIN OVER ALL DIMENSIONS of the bulk ESS: at config 2 that is 1000 dimensions x
# 1024 chains x 1000 draws = 1e9 values — one batched sort (rank normalisation) and one batched FFT per block of dimensions on the
# GPU that holds the trace; only the chain-averaged autocovariances [dim, draws / 2] come back to the host, where Geyer's
# truncation runs vectorised over the dimensions.  Same estimator as ess_bulk above (checked against it in tests/test_host_logic.py).


def _geyer_vectorised(acov_mean, chain_mean_var, n_chain, n_draw):
    """ess_from_chains' truncation for many quantities at once.  acov_mean[q, t]: autocovariance at lag t averaged over the
    chains; chain_mean_var[q]: variance (ddof = 1) of the chain means.  Returns ESS[q]."""
    Q, n = acov_mean.shape
    assert n == n_draw
    if n_draw < 4:
        return np.full(Q, np.nan)
    mean_var = acov_mean[:, 0] * n_draw / (n_draw - 1.0)
    var_plus = mean_var * (n_draw - 1.0) / n_draw
    if n_chain > 1:
        var_plus = var_plus + chain_mean_var
    bad = ~np.isfinite(var_plus) | (var_plus <= 0)
    vp = np.where(bad, 1.0, var_plus)
    R = 1.0 - (mean_var[:, None] - acov_mean) / vp[:, None]   # rho_hat at every lag
    rho = np.zeros((Q, n_draw))
    rho[:, 0] = 1.0
    rho[:, 1] = R[:, 1]
    t = np.ones(Q, dtype=np.int64)
    re, ro = np.ones(Q), R[:, 1].copy()
    active = np.ones(Q, dtype=bool)
    for tt in range(1, n_draw - 3, 2):
        active &= (re + ro) > 0.0
        if not active.any():
            break
        e, o = R[:, tt + 1], R[:, tt + 2]
        re = np.where(active, e, re)
        ro = np.where(active, o, ro)
        ok = active & ((e + o) >= 0)
        rho[ok, tt + 1] = e[ok]
        rho[ok, tt + 2] = o[ok]
        t = np.where(active, tt + 2, t)
    max_t = t - 2
    rows = np.arange(Q)
    pos = re > 0
    rho[rows[pos], max_t[pos] + 1] = re[pos]
    for tt in range(1, int(max_t.max()) - 1, 2):
        m = (tt <= max_t - 2) & (rho[:, tt + 1] + rho[:, tt + 2] > rho[:, tt - 1] + rho[:, tt])
        if m.any():
            v = (rho[:, tt - 1] + rho[:, tt]) / 2.0
            rho[m, tt + 1] = v[m]
            rho[m, tt + 2] = v[m]
    lag = np.arange(n_draw)[None, :]
    n_total = n_chain * n_draw
    tau = -1.0 + 2.0 * np.where(lag <= max_t[:, None], rho, 0.0).sum(axis=1) + np.where(lag == (max_t + 1)[:, None], rho, 0.0).sum(axis=1)
    tau = np.maximum(tau, 1.0 / np.log10(n_total))
    out = n_total / tau
    out[bad] = np.nan
    return out


def ess_bulk_all(draws, block=32):
    """Bulk ESS of EVERY dimension of ``draws[chain, draw, dim]`` (a torch tensor on any device, or a numpy array): numpy [dim].
    Rank normalisation (average ranks, Blom offsets), split chains, FFT autocovariance on the tensor's device, in blocks of
    ``block`` dimensions; Geyer's truncation on the host (vectorised)."""
    import torch

    d = draws if isinstance(draws, torch.Tensor) else torch.as_tensor(np.asarray(draws, dtype=np.float64))
    C, N, D = d.shape
    half = N // 2
    n_chain, n_draw = 2 * C, half
    m = 1 << int(np.ceil(np.log2(2 * max(n_draw, 1))))
    out = np.empty(D)
    for lo in range(0, D, block):
        hi = min(D, lo + block)
        x = d[:, :, lo:hi].to(torch.float64)
        # _split: first half and last half of every chain (the middle draw of an odd-length chain is dropped)
        x = torch.cat([x[:, :half], x[:, N - half:]], dim=0).permute(2, 0, 1).contiguous()   # [B, 2C, half]
        B = x.shape[0]
        flat = x.reshape(B, -1)
        tot = flat.shape[1]
        v, idx = torch.sort(flat, dim=1)
        lo_r = torch.searchsorted(v, v, right=False)
        hi_r = torch.searchsorted(v, v, right=True)
        r_sorted = (lo_r + hi_r + 1).to(torch.float64) * 0.5          # average rank of ties, 1-based
        del lo_r, hi_r, v
        r = torch.empty_like(r_sorted)
        r.scatter_(1, idx, r_sorted)
        del r_sorted, idx
        z = torch.special.ndtri((r - 0.375) / (tot + 0.25)).reshape(B, n_chain, n_draw)
        del r
        zc = z - z.mean(dim=2, keepdim=True)
        f = torch.fft.rfft(zc, n=m, dim=2)
        ac = torch.fft.irfft(f * torch.conj(f), n=m, dim=2)[:, :, :n_draw] / n_draw
        acov_mean = ac.mean(dim=1).cpu().numpy()
        cm = z.mean(dim=2)
        cmv = (cm.var(dim=1, unbiased=True) if n_chain > 1 else torch.zeros(B, dtype=torch.float64, device=z.device)).cpu().numpy()
        del z, zc, f, ac
        out[lo:hi] = _geyer_vectorised(acov_mean, cmv, n_chain, n_draw)
    return out
