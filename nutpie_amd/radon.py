"""Radon hierarchical model (BASELINE.json config 3) as a hand-written batched torch log-density.

The model is the one in the reference's README (``README.md:60-88``): intercept, two ``ZeroSumNormal``
county effects scaled by ``HalfNormal`` standard deviations, a floor effect and a ``HalfNormal(1.5)``
observation noise — written directly on the unconstrained scale PyMC would sample on (log transforms
with their Jacobians; PyMC's isometric zero-sum extension, so the zero-sum block is a standard normal
in ``n - 1`` free coordinates).  The real ``radon.csv`` is a network download in the reference
(``README.md:54``); here a synthetic data set of the same shape (85 counties, 919 observations) is drawn
once from the prior with a fixed seed.

Unconstrained vector (D = 2 n + 3 = 173):
    [intercept, county_raw (n-1), log county_sd, floor_effect, county_floor_raw (n-1), log county_floor_sd, log sigma]
"""

from __future__ import annotations

import numpy as np

from nutpie_amd.compiled_pyfunc import from_torchfunc

N_COUNTIES = 85
N_OBS = 919


def synthetic_radon_data(n_counties=N_COUNTIES, n_obs=N_OBS, seed=20260926):
    rng = np.random.default_rng(seed)
    county_idx = rng.integers(0, n_counties, size=n_obs)
    county_idx[:n_counties] = np.arange(n_counties)  # every county observed at least once
    floor = (rng.uniform(size=n_obs) < 0.18).astype(np.float64)
    county_effect = 0.3 * rng.normal(size=n_counties)
    county_effect -= county_effect.mean()
    cf_effect = 0.25 * rng.normal(size=n_counties)
    cf_effect -= cf_effect.mean()
    mu = 1.3 + county_effect[county_idx] - 0.6 * floor + cf_effect[county_idx] * floor
    y = mu + 0.75 * rng.normal(size=n_obs)
    return {"county_idx": county_idx, "floor": floor, "log_radon": y}


def _extend_zero_sum(x):
    """PyMC's ZeroSumTransform.backward (isometric R^{n-1} -> zero-sum R^n), batched over the leading axes."""
    import torch

    n = x.shape[-1] + 1
    s = x.sum(-1, keepdim=True)
    norm = s / (np.sqrt(n) + n)
    fill = norm - s / np.sqrt(n)
    return torch.cat([x, fill], -1) - norm


def radon_model(data=None, device=0, use_graph=False, expand_on_device=True):
    """Returns a :class:`~nutpie_amd.compiled_pyfunc.TorchFuncModel` for the radon model."""
    import torch

    data = data or synthetic_radon_data()
    n = int(np.max(data["county_idx"])) + 1
    D = 2 * n + 3
    o_int, o_raw, o_lsd, o_floor, o_craw, o_lcsd, o_lsig = 0, 1, n, n + 1, n + 2, 2 * n + 1, 2 * n + 2

    def make_logp():
        dev = torch.device(device) if isinstance(device, str) else torch.device("cuda", device)
        cidx = torch.as_tensor(data["county_idx"], device=dev, dtype=torch.long)
        floor = torch.as_tensor(data["floor"], device=dev, dtype=torch.float64)
        y = torch.as_tensor(data["log_radon"], device=dev, dtype=torch.float64)
        n_obs = y.shape[0]

        def logp_only(x):
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

        # One-hot design matrix of the county effects: row j of `design` spreads county effect j over its
        # observations, row n + j does the same for the county-specific floor effect (times the floor indicator), so
        # the gather of both effects is ONE fp64 GEMM forwards and the scatter-add of their gradients ONE backwards.
        onehot = torch.zeros((n, n_obs), dtype=torch.float64, device=dev)
        onehot[cidx, torch.arange(n_obs, device=dev)] = 1.0
        design = torch.cat([onehot, onehot * floor], 0)          # [2n, n_obs]
        design_t = design.t().contiguous()                       # [n_obs, 2n]
        c1, c2 = 1.0 / (np.sqrt(n) + n), 1.0 / np.sqrt(n)

        def extend_t(u):
            """Transpose of the zero-sum extension: R^n -> R^{n-1}, batched."""
            return u[:, :-1] - (c1 * u[:, :-1].sum(-1, keepdim=True) + c2 * u[:, -1:])

        def logp(x):
            """Log-density and its hand-derived gradient: ~35 kernels instead of the ~190 of autograd (the evaluation
            is launch-bound: 512 x 919 elements per kernel)."""
            x = x.detach()
            intercept, fe = x[:, o_int], x[:, o_floor]
            raw, craw = x[:, o_raw:o_raw + n - 1], x[:, o_craw:o_craw + n - 1]
            lsd, lcsd, lsig = x[:, o_lsd], x[:, o_lcsd], x[:, o_lsig]
            sd, csd, sig = torch.exp(lsd), torch.exp(lcsd), torch.exp(lsig)
            ext, cext = _extend_zero_sum(raw), _extend_zero_sum(craw)
            eff = torch.cat([ext * sd[:, None], cext * csd[:, None]], 1)                  # [B, 2n]
            mu = torch.addmm(intercept[:, None] + fe[:, None] * floor, eff, design)        # [B, n_obs]
            inv_sig = 1.0 / sig
            r = (y - mu) * inv_sig[:, None]
            rr = (r * r).sum(-1)
            lp = (-0.005 * intercept * intercept - 0.125 * fe * fe - 0.5 * (raw * raw).sum(-1) - 0.5 * (craw * craw).sum(-1)
                  - 0.5 * sd * sd + lsd - 0.5 * csd * csd + lcsd - (0.5 / 2.25) * sig * sig + lsig - 0.5 * rr - n_obs * lsig)
            w = r * inv_sig[:, None]                                                       # d lp / d mu
            gw = w @ design_t                                                              # [B, 2n]
            g_ext, g_cext = gw[:, :n], gw[:, n:]
            g = torch.empty_like(x)
            g[:, o_int] = -0.01 * intercept + w.sum(-1)
            g[:, o_floor] = -0.25 * fe + (w * floor).sum(-1)
            g[:, o_raw:o_raw + n - 1] = extend_t(g_ext * sd[:, None]) - raw
            g[:, o_craw:o_craw + n - 1] = extend_t(g_cext * csd[:, None]) - craw
            g[:, o_lsd] = 1.0 - sd * sd + sd * (ext * g_ext).sum(-1)
            g[:, o_lcsd] = 1.0 - csd * csd + csd * (cext * g_cext).sum(-1)
            g[:, o_lsig] = 1.0 - sig * sig / 2.25 + rr - n_obs
            return lp, g

        def logp_autograd(x):
            """The same density differentiated by autograd (kept as the reference of the hand-derived gradient)."""
            xg = x.detach().requires_grad_(True)
            lp = logp_only(xg)
            (g,) = torch.autograd.grad(lp.sum(), xg)
            return lp.detach(), g

        logp.autograd_reference = logp_autograd
        return logp

    def expand(x):
        """unconstrained [N, D] (numpy) -> dict of the model's named variables."""
        x = np.asarray(x)

        def ext(v):
            m = v.shape[-1] + 1
            s = v.sum(-1, keepdims=True)
            norm = s / (np.sqrt(m) + m)
            return np.concatenate([v, norm - s / np.sqrt(m)], -1) - norm

        raw, craw = ext(x[:, o_raw:o_raw + n - 1]), ext(x[:, o_craw:o_craw + n - 1])
        sd, csd = np.exp(x[:, o_lsd]), np.exp(x[:, o_lcsd])
        return {
            "intercept": x[:, o_int], "county_raw": raw, "county_sd": sd, "county_effect": raw * sd[:, None],
            "floor_effect": x[:, o_floor], "county_floor_raw": craw, "county_floor_sd": csd,
            "county_floor_effect": craw * csd[:, None], "sigma": np.exp(x[:, o_lsig]),
        }

    def expand_device(x):
        """Same map as :func:`expand`, on the GPU (x: Tensor[N, D])."""
        raw, craw = _extend_zero_sum(x[:, o_raw:o_raw + n - 1]), _extend_zero_sum(x[:, o_craw:o_craw + n - 1])
        sd, csd = torch.exp(x[:, o_lsd]), torch.exp(x[:, o_lcsd])
        return {
            "intercept": x[:, o_int], "county_raw": raw, "county_sd": sd, "county_effect": raw * sd[:, None],
            "floor_effect": x[:, o_floor], "county_floor_raw": craw, "county_floor_sd": csd,
            "county_floor_effect": craw * csd[:, None], "sigma": torch.exp(x[:, o_lsig]),
        }

    names = ["intercept", "county_raw", "county_sd", "county_effect", "floor_effect", "county_floor_raw", "county_floor_sd",
             "county_floor_effect", "sigma"]
    shapes = [(), (n,), (), (n,), (), (n,), (), (n,), ()]
    dims = {k: ("county",) for k in ("county_raw", "county_effect", "county_floor_raw", "county_floor_effect")}
    model = from_torchfunc(D, make_logp, expand, shapes, names, coords={"county": np.arange(n)}, dims=dims, use_graph=use_graph,
                           expand_device_fn=expand_device if expand_on_device else None)
    return model
