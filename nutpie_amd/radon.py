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


def radon_torch_density(data=None, device="cpu"):
    """The model's log-density as a user of ``from_torch_density`` writes it (forward pass only, batched over chains): what
    BASELINE config 3 names as its backend.  Returns ``(D, logp)``; the data tensors live on ``device``."""
    import torch

    data = data or synthetic_radon_data()
    n = int(np.max(data["county_idx"])) + 1
    D = 2 * n + 3
    o_int, o_raw, o_lsd, o_floor, o_craw, o_lcsd, o_lsig = 0, 1, n, n + 1, n + 2, 2 * n + 1, 2 * n + 2
    dev = torch.device(device) if isinstance(device, str) else torch.device("cuda", device)
    cidx = torch.as_tensor(data["county_idx"], device=dev, dtype=torch.long)
    floor = torch.as_tensor(data["floor"], device=dev, dtype=torch.float64)
    y = torch.as_tensor(data["log_radon"], device=dev, dtype=torch.float64)
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

    return D, logp


def radon_expand(data=None):
    """(expand function of a numpy block ``[N, D]``, names, shapes, dims, coords) of the radon model's reported variables"""
    data = data or synthetic_radon_data()
    n = int(np.max(data["county_idx"])) + 1
    o_raw, o_lsd, o_floor, o_craw, o_lcsd, o_lsig = 1, n, n + 1, n + 2, 2 * n + 1, 2 * n + 2

    def expand(x, **_data):
        x = np.asarray(x)

        def ext(v):
            m = v.shape[-1] + 1
            s = v.sum(-1, keepdims=True)
            norm = s / (np.sqrt(m) + m)
            return np.concatenate([v, norm - s / np.sqrt(m)], -1) - norm

        raw, craw = ext(x[:, o_raw:o_raw + n - 1]), ext(x[:, o_craw:o_craw + n - 1])
        sd, csd = np.exp(x[:, o_lsd]), np.exp(x[:, o_lcsd])
        return {"intercept": x[:, 0], "county_raw": raw, "county_sd": sd, "county_effect": raw * sd[:, None], "floor_effect": x[:, o_floor],
                "county_floor_raw": craw, "county_floor_sd": csd, "county_floor_effect": craw * csd[:, None], "sigma": np.exp(x[:, o_lsig])}

    names = ["intercept", "county_raw", "county_sd", "county_effect", "floor_effect", "county_floor_raw", "county_floor_sd", "county_floor_effect", "sigma"]
    shapes = [(), (n,), (), (n,), (), (n,), (), (n,), ()]
    dims = {k: ("county",) for k in ("county_raw", "county_effect", "county_floor_raw", "county_floor_effect")}
    return expand, names, shapes, dims, {"county": np.arange(n)}


def radon_traced_model(data=None, compile=True, **kw):
    """Config 3 with the backend BASELINE.json names — a torch log-density — compiled: :func:`nutpie_amd.from_torch_density` traces
    ``radon_torch_density`` (``torch.fx``), differentiates the graph and compiles the generated HIP density into the model's own
    resident kernel.  ``compile=False``: the same function evaluated eagerly behind the batched device callback (the data on the GPU)."""
    from nutpie_amd.compiled_pyfunc import from_torch_density

    D, logp = radon_torch_density(data, device="cpu" if compile else kw.pop("device", 0))
    expand, names, shapes, dims, coords = radon_expand(data)
    return from_torch_density(D, logp, compile=compile, expand_fn=expand, expanded_names=names, expanded_shapes=shapes, dims=dims, coords=coords, **kw)


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


# --------------------------------------------------------------------------------------------------------------------------
# The same density as HIP source (nutpie_amd/density.py): compiled at run time into the model's own resident kernel.
# One wavefront per chain; LDS scratch per chain: county effects [128] | county floor effects [128] | d lp / d mu per observation, twice;
# the observations themselves are staged once per workgroup in LDS shared by its four chains (a lone wave waits out every L2 access).
# The formulas are those of the native callback kernel tests/fixtures/radon_device.hip (the round-2 form of this model); the wave
# sums use the engine's DPP reduction.
# --------------------------------------------------------------------------------------------------------------------------
RADON_MAX_COUNTIES = 128

RADON_DENSITY_SOURCE = r"""
// the workgroup's shared LDS block: y[n_obs] | floor[n_obs] | county[n_obs], pos[n_obs], row_start[n + 1] as 32-bit integers
// (pos[o] = where observation o sits when the observations are grouped by county: the per-county sums then read contiguous
//  ranges instead of chasing an index per term)
__device__ void nphip_density_stage(const NphipData& d, double* shared, int thread, int n_threads) {
    const int n_obs = d.n_y, n = d.n_counties;
    int* ints = (int*)(shared + 2 * n_obs);
    for (int o = thread; o < n_obs; o += n_threads) {
        shared[o] = d.y[o];
        shared[n_obs + o] = d.floor[o];
        ints[o] = d.county[o];
        ints[n_obs + o] = d.pos[o];
    }
    for (int j = thread; j <= n; j += n_threads) ints[2 * n_obs + j] = d.row_start[j];
}

// A lone wave pays the full latency of every access it has to wait for (~100 cycles in LDS, several hundred in L2), so the loops
// below first issue the independent reads of a few iterations and then compute.
__device__ double nphip_density(const NphipData& d, int dim, const double* x, double* g, double* lds, const double* shared, int lane) {
    const int n = d.n_counties, n_obs = d.n_y;
    const auto y_ = NPHIP_LDS_CPTR(double, shared);
    const auto floor_ = y_ + n_obs;
    const auto county_ = NPHIP_LDS_CPTR(int, shared + 2 * n_obs);
    const auto pos_ = county_ + n_obs;
    const auto row_start_ = county_ + 2 * n_obs;
    const auto eff = NPHIP_LDS_PTR(double, lds);   // [ce(n) | cfe(n)] then, grouped by county, w[n_obs] and (w * floor)[n_obs]
    const auto cfe = eff + 128;
    const auto ws = eff + 256;
    const auto wfs = ws + n_obs;
    const int o_raw = 1, o_lsd = n, o_floor = n + 1, o_craw = n + 2, o_lcsd = 2 * n + 1, o_lsig = 2 * n + 2;
    const double intercept = x[0], fe = x[o_floor], lsd = x[o_lsd], lcsd = x[o_lcsd], lsig = x[o_lsig];
    // the raw county vectors of this lane (n <= 128: at most two per lane)
    double xa[2] = {0.0, 0.0}, xb[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        if (j < n - 1) { xa[t] = x[o_raw + j]; xb[t] = x[o_craw + j]; }
    }
    const double sd = exp(lsd), csd = exp(lcsd), sig = exp(lsig), inv_sig = 1.0 / sig;
    const double c1 = 1.0 / (sqrt((double)n) + n), c2 = 1.0 / sqrt((double)n);
    // zero-sum extension of the two raw vectors (PyMC ZeroSumTransform.backward)
    double s_raw = 0.0, s_craw = 0.0, ss = 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (lane + 64 * t < n - 1) { s_raw += xa[t]; s_craw += xb[t]; ss += xa[t] * xa[t] + xb[t] * xb[t]; }
    }
    nphip_wave_sum3(s_raw, s_craw, ss);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        if (j < n) {
            const double e = (j < n - 1) ? xa[t] - s_raw * c1 : -s_raw * c2;
            const double ce = (j < n - 1) ? xb[t] - s_craw * c1 : -s_craw * c2;
            eff[j] = e * sd;
            cfe[j] = ce * csd;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // observations: residuals, d lp / d mu — four iterations' reads in flight at a time
    double rr = 0.0, sw = 0.0, swf = 0.0;
    constexpr int U = 4;
    for (int o0 = lane; o0 < n_obs; o0 += 64 * U) {
        int cty[U], ps[U];
        double fl[U], yy[U], e1[U], e2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int o = o0 + 64 * u, oo = o < n_obs ? o : 0;
            cty[u] = county_[oo]; ps[u] = pos_[oo]; fl[u] = floor_[oo]; yy[u] = y_[oo];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { e1[u] = eff[cty[u]]; e2[u] = cfe[cty[u]]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (o0 + 64 * u < n_obs) {
                const double mu = intercept + e1[u] + fl[u] * (fe + e2[u]);
                const double r = (yy[u] - mu) * inv_sig;
                const double wo = r * inv_sig;
                ws[ps[u]] = wo;
                wfs[ps[u]] = wo * fl[u];
                rr += r * r; sw += wo; swf += wo * fl[u];
            }
        }
    }
    nphip_wave_sum3(rr, sw, swf);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // per-county sums of w (county effect) and w * floor (county floor effect): one lane per county, contiguous ranges, fixed order
    double dot_e = 0.0, dot_c = 0.0, su_e = 0.0, su_c = 0.0;
    double ge_[2] = {0.0, 0.0}, gc_[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        if (j < n) {
            double a = 0.0, b = 0.0;
            const int k0 = row_start_[j], k1 = row_start_[j + 1];
#pragma unroll 4
            for (int k = k0; k < k1; ++k) { a += ws[k]; b += wfs[k]; }
            ge_[t] = a; gc_[t] = b;
            const double ext = eff[j] / sd, cext = cfe[j] / csd;
            dot_e += ext * a; dot_c += cext * b;
            if (j < n - 1) { su_e += a * sd; su_c += b * csd; }
        }
    }
    nphip_wave_sum4(dot_e, dot_c, su_e, su_c);
    // last county's (scaled) gradient, needed by the transpose of the extension
    const int last_lane = (n - 1) & 63, last_t = (n - 1) >> 6;
    const double gl_e = __shfl(last_t == 0 ? ge_[0] : ge_[1], last_lane, 64) * sd;
    const double gl_c = __shfl(last_t == 0 ? gc_[0] : gc_[1], last_lane, 64) * csd;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        if (j < n - 1) {
            g[o_raw + j] = (ge_[t] * sd - (c1 * su_e + c2 * gl_e)) - xa[t];
            g[o_craw + j] = (gc_[t] * csd - (c1 * su_c + c2 * gl_c)) - xb[t];
        }
    }
    if (lane == 0) {
        g[0] = -0.01 * intercept + sw;
        g[o_floor] = -0.25 * fe + swf;
        g[o_lsd] = 1.0 - sd * sd + sd * dot_e;
        g[o_lcsd] = 1.0 - csd * csd + csd * dot_c;
        g[o_lsig] = 1.0 - sig * sig / 2.25 + rr - n_obs;
    }
    return -0.005 * intercept * intercept - 0.125 * fe * fe - 0.5 * ss - 0.5 * sd * sd + lsd - 0.5 * csd * csd + lcsd
           - (0.5 / 2.25) * sig * sig + lsig - 0.5 * rr - n_obs * lsig;
}
"""


def radon_density_data(data=None):
    """The ``data`` dict of the HIP-source radon model: the observations, and their grouping by county (the per-county gradient
    sums are taken by one lane per county over a contiguous range, in a fixed order)."""
    data = data or synthetic_radon_data()
    county = np.asarray(data["county_idx"], dtype=np.int32)
    n = int(county.max()) + 1
    if n > RADON_MAX_COUNTIES:
        raise ValueError(f"the radon density source holds up to {RADON_MAX_COUNTIES} counties in its LDS scratch")
    counts = np.bincount(county, minlength=n)
    row_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    pos = np.empty(len(county), dtype=np.int32)
    pos[np.argsort(county, kind="stable")] = np.arange(len(county), dtype=np.int32)   # where observation o sits, grouped by county
    return {"county": county, "floor": np.asarray(data["floor"], dtype=np.float64), "y": np.asarray(data["log_radon"], dtype=np.float64),
            "row_start": row_start, "pos": pos, "n_counties": n}


def radon_density_model(data=None, resident=True):
    """Config 3's model as a runtime-compiled device density (:func:`nutpie_amd.from_density_source`): the same variables as
    :func:`radon_model`; ``with_data`` swaps the observations without recompiling."""
    from nutpie_amd.density import from_density_source

    dd = radon_density_data(data)
    n = dd["n_counties"]
    D = 2 * n + 3
    o_raw, o_lsd, o_floor, o_craw, o_lcsd, o_lsig = 1, n, n + 1, n + 2, 2 * n + 1, 2 * n + 2

    def expand(x, **_data):
        x = np.asarray(x)

        def ext(v):
            m = v.shape[-1] + 1
            s = v.sum(-1, keepdims=True)
            norm = s / (np.sqrt(m) + m)
            return np.concatenate([v, norm - s / np.sqrt(m)], -1) - norm

        raw, craw = ext(x[:, o_raw:o_raw + n - 1]), ext(x[:, o_craw:o_craw + n - 1])
        sd, csd = np.exp(x[:, o_lsd]), np.exp(x[:, o_lcsd])
        return {"intercept": x[:, 0], "county_raw": raw, "county_sd": sd, "county_effect": raw * sd[:, None], "floor_effect": x[:, o_floor],
                "county_floor_raw": craw, "county_floor_sd": csd, "county_floor_effect": craw * csd[:, None], "sigma": np.exp(x[:, o_lsig])}

    names = ["intercept", "county_raw", "county_sd", "county_effect", "floor_effect", "county_floor_raw", "county_floor_sd", "county_floor_effect", "sigma"]
    shapes = [(), (n,), (), (n,), (), (n,), (), (n,), ()]
    dims = {k: ("county",) for k in ("county_raw", "county_effect", "county_floor_raw", "county_floor_effect")}
    def shared_doubles(d):   # y | floor | (county, pos, row_start) as 32-bit integers
        n_obs = len(d["y"])
        return 2 * n_obs + (2 * n_obs + int(d["n_counties"]) + 1 + 1) // 2

    return from_density_source(D, RADON_DENSITY_SOURCE, dd, lds_doubles_per_chain=lambda d: 2 * RADON_MAX_COUNTIES + 2 * len(d["y"]),
                               lds_doubles_shared=shared_doubles, expand_fn=expand,
                               expanded_names=names, expanded_shapes=shapes, coords={"county": np.arange(n)}, dims=dims, resident=resident)


def radon_symbolic_model(data=None):
    """Config 3's model written with the front-end (:mod:`nutpie_amd.symbolic`): expressions -> symbolic gradient -> generated HIP
    density -> the model's own resident kernel (``.compile()``).  Same variables and priors as :func:`radon_model`."""
    from nutpie_amd import symbolic as S

    d = data or synthetic_radon_data()
    n = int(np.max(d["county_idx"])) + 1
    m = S.Model()
    m.dim("county", n)
    intercept = m.param("intercept")
    raw = m.param("county_raw", dim="county", zero_sum=True)
    sd = m.param("county_sd", lower=0.0)
    fe = m.param("floor_effect")
    craw = m.param("county_floor_raw", dim="county", zero_sum=True)
    csd = m.param("county_floor_sd", lower=0.0)
    sig = m.param("sigma", lower=0.0)
    y = m.data("y", d["log_radon"], dim="obs")
    fl = m.data("floor", d["floor"], dim="obs")
    ci = m.index("county", d["county_idx"], dim="obs", into="county")
    eff, cfe = raw * sd, craw * csd
    m.deterministic("county_effect", eff)
    m.deterministic("county_floor_effect", cfe)
    # the priors of radon_model, constants dropped as there (the hand-written density is the comparison)
    m.add_logp(-0.005 * (intercept * intercept) - 0.125 * (fe * fe))
    m.add_logp((-0.5 * (raw * raw)).sum() + (-0.5 * (craw * craw)).sum())
    m.add_logp(-0.5 * (sd * sd) - 0.5 * (csd * csd) - (0.5 / 2.25) * (sig * sig))
    mu = intercept + eff[ci] + fl * (fe + cfe[ci])
    z = (y - mu) / sig
    m.add_logp((-0.5 * (z * z) - S.log(sig)).sum())
    return m
