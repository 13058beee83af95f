"""Analytic Gaussian targets (BASELINE.json configs 1, 2 and 5).

``TridiagGaussian`` is evaluated inside the leapfrog kernel (fused analytic gradient);
``DenseGaussian`` (``dense_gaussian``) is the dense-precision variant: the gradients of all chains of a
step are one fp64 GEMM on the matrix cores, hand-written inside the engine (``nphip_model_dense_gaussian``,
csrc/dense_tile.h); ``dense_gaussian_torch`` is the same target behind the batched torch callback
(rocBLAS), kept as the comparison.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from nutpie_amd import _lib
from nutpie_amd.sample import CompiledModel

BENCH_RNG_SEED = 20260926  # inputs from numpy.random.default_rng(20260926)


@dataclass(frozen=True)
class TridiagGaussian(CompiledModel):
    """logp(x) = -1/2 (x - mu)' L (x - mu) with L = tridiag(offdiag, diag, offdiag)."""

    diag: np.ndarray = None
    offdiag: np.ndarray | None = None
    mu: np.ndarray | None = None
    name: str = "x"
    init: str = "uniform"

    @property
    def n_dim(self):
        return int(len(self.diag))

    @property
    def shapes(self):
        return {self.name: (self.n_dim,)}

    @property
    def coords(self):
        return {}

    def _make_model(self, init_mean=None, settings=None):
        m = _lib.TridiagGaussianModel(self.diag, self.offdiag, self.mu)
        m.set_init(self.init)
        return m

    def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
        return _lib.PySampler.from_pyfunc(settings, cores, self._make_model(), progress_type, extra_callback, extra_callback_rate, store, **engine_kw)

    def _expand_draws(self, draws):
        return {self.name: draws}

    # analytic moments, used by the statistical tests
    def covariance(self):
        n = self.n_dim
        L = np.diag(np.asarray(self.diag, dtype=np.float64))
        if self.offdiag is not None and n > 1:
            L += np.diag(self.offdiag, 1) + np.diag(self.offdiag, -1)
        return np.linalg.inv(L)


@dataclass(frozen=True)
class DenseGaussian(CompiledModel):
    """logp(x) = -1/2 (x - mu)' P (x - mu) with a dense symmetric precision matrix P (in-engine fp64 MFMA gradient)."""

    precision: np.ndarray = None
    mu: np.ndarray | None = None
    name: str = "x"
    init: str = "uniform"

    @property
    def n_dim(self):
        return int(self.precision.shape[0])

    @property
    def shapes(self):
        return {self.name: (self.n_dim,)}

    @property
    def coords(self):
        return {}

    def _make_model(self, init_mean=None, settings=None):
        m = _lib.DenseGaussianModel(self.precision, self.mu)
        m.set_init(self.init)
        return m

    def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
        return _lib.PySampler.from_pyfunc(settings, cores, self._make_model(), progress_type, extra_callback, extra_callback_rate, store, **engine_kw)

    def _expand_draws(self, draws):
        return {self.name: draws}

    def covariance(self):
        return np.linalg.inv(np.asarray(self.precision, dtype=np.float64))


def std_normal(dim: int) -> TridiagGaussian:
    """Config 1: D-dimensional standard normal."""
    return TridiagGaussian(dims={}, diag=np.ones(dim))


def diag_gaussian(sd, mu=None) -> TridiagGaussian:
    sd = np.asarray(sd, dtype=np.float64)
    return TridiagGaussian(dims={}, diag=1.0 / sd**2, mu=None if mu is None else np.asarray(mu, dtype=np.float64))


def ar1_gaussian(dim: int, rho: float = 0.9, scales=None, seed: int = BENCH_RNG_SEED) -> TridiagGaussian:
    """Config 2/5 (variant i of SURVEY.md §8d): x_i = s_i y_i, y a stationary AR(1) process with
    correlation rho — tridiagonal precision, per-dimension scales s_i = exp(N(0,1))."""
    if scales is None:
        scales = np.exp(np.random.default_rng(seed).normal(size=dim))
    s = np.asarray(scales, dtype=np.float64)
    c = 1.0 / (1.0 - rho * rho)
    d = np.full(dim, (1.0 + rho * rho) * c)
    d[0] = d[-1] = c
    if dim == 1:
        d[0] = 1.0
    off = np.full(max(dim - 1, 0), -rho * c)
    diag = d / s**2
    offdiag = off / (s[:-1] * s[1:])
    return TridiagGaussian(dims={}, diag=diag, offdiag=offdiag if dim > 1 else None)


def dense_precision(dim: int, seed: int = BENCH_RNG_SEED, cond_lo=1e-2, cond_hi=1e2) -> np.ndarray:
    """Precision matrix of config 2 variant (ii): covariance S C S with C = Q diag(lam) Q', Q random orthogonal,
    lam log-uniform in [cond_lo, cond_hi], S = diag(exp(N(0,1))); inputs from numpy.random.default_rng(seed)."""
    rng = np.random.default_rng(seed)
    s = np.exp(rng.normal(size=dim))
    Q, _ = np.linalg.qr(rng.normal(size=(dim, dim)))
    lam = np.exp(rng.uniform(np.log(cond_lo), np.log(cond_hi), size=dim))
    prec = (Q / lam) @ Q.T
    prec = prec / np.outer(s, s)
    return 0.5 * (prec + prec.T)


def dense_gaussian(dim: int, seed: int = BENCH_RNG_SEED, cond_lo=1e-2, cond_hi=1e2, device=0) -> DenseGaussian:
    """Config 2 variant (ii) inside the engine: the gradient GEMM is the engine's own fp64 MFMA kernel."""
    return DenseGaussian(dims={}, precision=dense_precision(dim, seed, cond_lo, cond_hi))


def dense_gaussian_torch(dim: int, seed: int = BENCH_RNG_SEED, cond_lo=1e-2, cond_hi=1e2, device=0):
    """The same target behind the batched torch callback (gradient = fp64 GEMM via rocBLAS, outside the leapfrog kernel)."""
    import torch

    from nutpie_amd.compiled_pyfunc import from_torchfunc

    prec = dense_precision(dim, seed, cond_lo, cond_hi)

    def make_logp():
        dev = torch.device("cuda", device)
        negP = torch.as_tensor(-prec, dtype=torch.float64, device=dev)
        half = torch.full((dim,), 0.5, dtype=torch.float64, device=dev)
        tmp = {}

        def logp(x, out_logp, out_grad):
            # three kernels, all writing preallocated buffers: grad = -x P (fp64 GEMM) into the engine's staging buffer, x * grad, and its
            # row sums times 1/2 (a GEMV) into the staging log-density
            key = (x.data_ptr(), x.shape[0])       # (engine option host_groups: groups of chains run concurrently on their own streams)
            t = tmp.get(key)
            if t is None:
                t = tmp[key] = torch.empty_like(x)
            torch.mm(x, negP, out=out_grad)
            torch.mul(x, out_grad, out=t)
            torch.mv(t, half, out=out_logp)

        logp.writes_staging = True
        return logp

    model = from_torchfunc(dim, make_logp)
    object.__setattr__(model, "_precision", prec)
    return model
