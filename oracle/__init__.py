"""ctypes front-end of the CPU oracle (oracle/nuts_oracle.cpp).

TEST INFRASTRUCTURE ONLY — imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg; never by the ``nutpie_amd`` package.

PARITY UNPINNED: see the header of ``oracle/nuts_oracle.h``.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnuts_oracle.so")
_TUNED_PATH = os.path.join(_HERE, "libnuts_oracle_tuned.so")

LOGP_FN = C.CFUNCTYPE(C.c_int64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


class Settings(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("num_tune", C.c_uint64),
        ("num_draws", C.c_uint64),
        ("num_chains", C.c_uint64),
        ("maxdepth", C.c_uint64),
        ("mindepth", C.c_uint64),
        ("check_turning", C.c_int32),
        ("use_grad_based_mass_matrix", C.c_int32),
        ("max_energy_error", C.c_double),
        ("early_window", C.c_double),
        ("step_size_window", C.c_double),
        ("mass_matrix_switch_freq", C.c_uint64),
        ("early_mass_matrix_switch_freq", C.c_uint64),
        ("mass_matrix_update_freq", C.c_uint64),
        ("initial_step", C.c_double),
        ("target_accept", C.c_double),
        ("step_size_jitter", C.c_double),
        ("max_step_size", C.c_double),
        ("da_k", C.c_double),
        ("da_t0", C.c_double),
        ("da_gamma", C.c_double),
        ("fixed_step_size", C.c_int32),
        ("adapt_mass_matrix", C.c_int32),
        ("init_kind", C.c_int32),
        ("num_try_init", C.c_int32),
        ("waves_per_chain", C.c_int32),
        ("n_threads", C.c_int32),
        ("chain_offset", C.c_uint64),
        ("store_gradient", C.c_int32),
        ("store_mass_matrix", C.c_int32),
        ("adam", C.c_int32),
        ("crate_arithmetic", C.c_int32),
        ("adam_learning_rate", C.c_double),
        ("store_divergences", C.c_int32),
        ("low_rank_metric", C.c_int32),
        ("n_metric_updates", C.c_int32),
        ("metric_k", C.c_int32),
        ("metric_draws", C.c_uint64 * 16),
        ("metric_sig2", C.c_void_p),
        ("metric_V", C.c_void_p),
        ("metric_lam", C.c_void_p),
    ]

    def set_metric_schedule(self, draws, sig2, V=None, lam=None):
        """Metrics handed to the sampler (nuts_oracle.h: low_rank_metric): update u applies before draw ``draws[u]``;
        ``sig2[u, chain, dim]``, ``V[u, chain, k, dim]``, ``lam[u, chain, k]``."""
        sig2 = np.ascontiguousarray(sig2, dtype=np.float64)
        U, n, _ = sig2.shape
        assert U == len(draws) <= 16 and n == int(self.num_chains)
        k = 0 if V is None else int(np.asarray(V).shape[2])
        self._keep = [sig2]
        self.low_rank_metric, self.n_metric_updates, self.metric_k = 1, U, k
        for i, d in enumerate(draws):
            self.metric_draws[i] = int(d)
        self.metric_sig2 = sig2.ctypes.data
        if k:
            # (the contract stores the columns in single precision — include/nphip_spec.h, engine_types.h: lr_V — and computes with the
            #  values fp32 holds: the oracle is handed the same rounded columns)
            V = np.ascontiguousarray(np.asarray(V, dtype=np.float64).astype(np.float32), dtype=np.float64)
            lam = np.ascontiguousarray(lam, dtype=np.float64)
            assert V.shape == (U, n, k, sig2.shape[2]) and lam.shape == (U, n, k)
            self._keep += [V, lam]
            self.metric_V, self.metric_lam = V.ctypes.data, lam.ctypes.data
        return self


class _Trace(C.Structure):
    _fields_ = [
        ("draws", C.c_void_p),
        ("depth", C.c_void_p),
        ("n_steps", C.c_void_p),
        ("index_in_trajectory", C.c_void_p),
        ("diverging", C.c_void_p),
        ("maxdepth_reached", C.c_void_p),
        ("tuning", C.c_void_p),
        ("energy", C.c_void_p),
        ("energy_error", C.c_void_p),
        ("logp", C.c_void_p),
        ("step_size", C.c_void_p),
        ("step_size_bar", C.c_void_p),
        ("mean_tree_accept", C.c_void_p),
        ("mean_tree_accept_sym", C.c_void_p),
        ("gradient", C.c_void_p),
        ("mass_matrix_inv", C.c_void_p),
        ("divergence_start", C.c_void_p),
        ("divergence_end", C.c_void_p),
        ("divergence_momentum", C.c_void_p),
        ("divergence_start_gradient", C.c_void_p),
    ]


_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (g++ only)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("nuts_oracle.cpp", "nuts_oracle.h", "Makefile")
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "libnuts_oracle.so", "libnuts_oracle_tuned.so"], check=True, capture_output=True)
    return _LIB_PATH


_tuned = None


def sample_tridiag_tuned(settings: "Settings", diag, offdiag=None):
    """The "tuned" CPU build (free summation order, fused AVX2 passes): ONLY a speed baseline for bench.py — its floats are not
    the contract's, it is never used as a checker.  Returns (total leapfrogs, seconds)."""
    global _tuned
    if _tuned is None:
        if not os.path.exists(_TUNED_PATH):
            build(force=True)
        _tuned = C.CDLL(_TUNED_PATH)
        _tuned.oracle_last_error.restype = C.c_char_p
    diag = np.ascontiguousarray(np.asarray(diag, dtype=np.float64))
    dim = diag.shape[0]
    offdiag = _vec(offdiag, max(dim - 1, 0)) if offdiag is not None and dim > 1 else None
    draws, st, tr = _alloc(settings, dim)
    secs = C.c_double(0)
    rc = _tuned.oracle_sample_tridiag(C.byref(settings), C.c_uint64(dim), None, _p(diag), _p(offdiag), None, C.byref(tr), C.byref(secs))
    if rc != 0:
        raise RuntimeError(_tuned.oracle_last_error().decode())
    return Trace(draws, st, secs.value)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_last_error.restype = C.c_char_p
        _lib.oracle_logaddexp.restype = C.c_double
        _lib.oracle_logaddexp.argtypes = [C.c_double, C.c_double]
        _lib.oracle_w_leaf.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        _lib.oracle_w_add.argtypes = [C.c_double, C.c_int64, C.c_double, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        _lib.oracle_dot.restype = C.c_double
        _lib.oracle_leapfrog_tridiag.restype = C.c_double
    return _lib


def set_variant(min_refresh: int = 3, search_mode: int = 2, late_sym: int = 1, last_bar: int = 1, floor_windows: int = 0) -> None:
    """Experiment knobs of the sensitivity study (nuts_oracle.h: oracle_set_variant); the defaults are the restatement the parity
    tests run.  Process-global: reset with ``set_variant()``."""
    lib().oracle_set_variant(int(min_refresh), int(search_mode), int(late_sym), int(last_bar), int(floor_windows))


def default_settings(**kw) -> Settings:
    s = Settings()
    lib().oracle_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class Trace:
    draws: np.ndarray
    stats: dict = field(default_factory=dict)
    seconds: float = 0.0

    def __getattr__(self, name):
        try:
            return self.stats[name]
        except KeyError as e:
            raise AttributeError(name) from e


def _alloc(s: Settings, dim: int):
    n, T = int(s.num_chains), int(s.num_tune + s.num_draws)
    st = {
        "depth": np.zeros((n, T), np.int64),
        "n_steps": np.zeros((n, T), np.int64),
        "index_in_trajectory": np.zeros((n, T), np.int64),
        "diverging": np.zeros((n, T), np.uint8),
        "maxdepth_reached": np.zeros((n, T), np.uint8),
        "tuning": np.zeros((n, T), np.uint8),
        "energy": np.zeros((n, T)),
        "energy_error": np.zeros((n, T)),
        "logp": np.zeros((n, T)),
        "step_size": np.zeros((n, T)),
        "step_size_bar": np.zeros((n, T)),
        "mean_tree_accept": np.zeros((n, T)),
        "mean_tree_accept_sym": np.zeros((n, T)),
    }
    if s.store_gradient:
        st["gradient"] = np.zeros((n, T, dim))
    if s.store_mass_matrix:
        st["mass_matrix_inv"] = np.zeros((n, T, dim))
    if s.store_divergences:
        for k in ("divergence_start", "divergence_end", "divergence_momentum", "divergence_start_gradient"):
            st[k] = np.full((n, T, dim), np.nan)
    draws = np.zeros((n, T, dim))
    tr = _Trace()
    tr.draws = _p(draws)
    for k, v in st.items():
        setattr(tr, k, _p(v))
    return draws, st, tr


def _vec(x, n):
    if x is None:
        return None
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    assert x.shape == (n,), (x.shape, n)
    return x


def sample_tridiag(settings: Settings, diag, offdiag=None, mu=None, init_points=None) -> Trace:
    diag = np.ascontiguousarray(np.asarray(diag, dtype=np.float64))
    dim = diag.shape[0]
    offdiag = _vec(offdiag, max(dim - 1, 0)) if offdiag is not None and dim > 1 else None
    mu = _vec(mu, dim)
    ip = None
    if init_points is not None:
        ip = np.ascontiguousarray(np.asarray(init_points, dtype=np.float64))
        assert ip.shape == (int(settings.num_chains), dim)
    draws, st, tr = _alloc(settings, dim)
    secs = C.c_double(0)
    rc = lib().oracle_sample_tridiag(C.byref(settings), C.c_uint64(dim), _p(mu), _p(diag), _p(offdiag), _p(ip), C.byref(tr), C.byref(secs))
    if rc != 0:
        raise RuntimeError(lib().oracle_last_error().decode())
    return Trace(draws, st, secs.value)


def sample_dense(settings: Settings, precision, mu=None, init_points=None, tuned=False) -> Trace:
    """Dense-precision Gaussian (the oracle counterpart of ``nphip_model_dense_gaussian``).  ``tuned``: the free-order SIMD build
    (a speed baseline only, as ``sample_tridiag_tuned``)."""
    global _tuned
    P = np.ascontiguousarray(precision, dtype=np.float64)
    dim = P.shape[0]
    assert P.shape == (dim, dim)
    mu = _vec(mu, dim)
    ip = None
    if init_points is not None:
        ip = np.ascontiguousarray(np.asarray(init_points, dtype=np.float64))
        assert ip.shape == (int(settings.num_chains), dim)
    draws, st, tr = _alloc(settings, dim)
    secs = C.c_double(0)
    if tuned:
        if _tuned is None:
            if not os.path.exists(_TUNED_PATH):
                build(force=True)
            _tuned = C.CDLL(_TUNED_PATH)
            _tuned.oracle_last_error.restype = C.c_char_p
        L = _tuned
    else:
        L = lib()
    rc = L.oracle_sample_dense(C.byref(settings), C.c_uint64(dim), _p(mu), _p(P), _p(ip), C.byref(tr), C.byref(secs))
    if rc != 0:
        raise RuntimeError(L.oracle_last_error().decode())
    return Trace(draws, st, secs.value)


def dense_grad(x, precision, mu=None, waves=1):
    """One evaluation of the dense model on the rows of ``x``: ``(grad[n, dim], logp[n])``."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = np.ascontiguousarray(precision, dtype=np.float64)
    n, dim = x.shape
    mu = _vec(mu, dim)
    g, lp = np.empty_like(x), np.empty(n)
    lib().oracle_dense_grad(C.c_uint64(n), C.c_uint64(dim), _p(x), _p(mu), _p(P), C.c_int(waves), _p(g), _p(lp))
    return g, lp


def _tuned_lib():
    global _tuned
    if _tuned is None:
        if not os.path.exists(_TUNED_PATH):
            build(force=True)
        _tuned = C.CDLL(_TUNED_PATH)
        _tuned.oracle_last_error.restype = C.c_char_p
    return _tuned


def sample_callback(settings: Settings, dim: int, fn, user=None, init_points=None, tuned=False) -> Trace:
    """fn: either a ctypes function pointer (LOGP_FN / raw address) or a Python callable
    ``f(x: ndarray) -> (logp, grad)`` wrapped here.  ``tuned``: the free-order SIMD build (a speed baseline only: its floats
    are not the contract's)."""
    keep = None
    if callable(fn) and not isinstance(fn, (int, C._CFuncPtr)):
        pyfn = fn

        def _cb(d, x, g, lp, _u):
            xs = np.ctypeslib.as_array(x, shape=(d,))
            try:
                val, grad = pyfn(xs.copy())
            except Exception:
                return 1
            np.ctypeslib.as_array(g, shape=(d,))[:] = grad
            lp[0] = val
            return 0

        keep = LOGP_FN(_cb)
        fnptr = C.cast(keep, C.c_void_p)
    elif isinstance(fn, int):
        fnptr = C.c_void_p(fn)
    else:
        fnptr = C.cast(fn, C.c_void_p)
    ip = None
    if init_points is not None:
        ip = np.ascontiguousarray(np.asarray(init_points, dtype=np.float64))
    draws, st, tr = _alloc(settings, dim)
    secs = C.c_double(0)
    L = _tuned_lib() if tuned else lib()
    rc = L.oracle_sample_callback(C.byref(settings), C.c_uint64(dim), fnptr, C.c_void_p(user), _p(ip), C.byref(tr), C.byref(secs))
    del keep
    if rc != 0:
        raise RuntimeError(L.oracle_last_error().decode())
    return Trace(draws, st, secs.value)


# ---- unit-level helpers -----------------------------------------------------


def philox(seed, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox(C.c_uint64(seed), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), out)
    return tuple(out)


def detmath(fn: str, x):
    code = {"exp": 0, "log": 1, "log1p": 2, "sin2pi": 3, "cos2pi": 4}[fn]
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    y = np.empty_like(x)
    lib().oracle_detmath(code, C.c_uint64(x.size), _p(x), _p(y))
    return y


def logaddexp(a, b):
    return lib().oracle_logaddexp(a, b)


def w_leaf(neg_energy_error):
    """Extended-range weight (m, e) of a leaf: m * 2**e ~= exp(neg_energy_error)."""
    m, e = C.c_double(), C.c_int64()
    lib().oracle_w_leaf(float(neg_energy_error), C.byref(m), C.byref(e))
    return m.value, e.value


def w_add(a, b):
    m, e = C.c_double(), C.c_int64()
    lib().oracle_w_add(float(a[0]), int(a[1]), float(b[0]), int(b[1]), C.byref(m), C.byref(e))
    return m.value, e.value


def normals(seed, chain, draw, purpose, n):
    out = np.empty(n)
    lib().oracle_normals(C.c_uint64(seed), C.c_uint32(chain), C.c_uint32(draw), C.c_uint32(purpose), C.c_uint64(n), _p(out))
    return out


def dot(x, y, waves=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().oracle_dot(_p(x), _p(y), C.c_uint64(x.size), C.c_int(waves))


def leapfrog_tridiag(q, p, g, sig2, eps, diag, offdiag=None, mu=None, waves=1):
    dim = len(q)
    q = np.array(q, dtype=np.float64)
    p = np.array(p, dtype=np.float64)
    g = np.array(g, dtype=np.float64)
    sig2 = _vec(sig2, dim)
    diag = _vec(diag, dim)
    offdiag = _vec(offdiag, dim - 1) if offdiag is not None and dim > 1 else None
    mu = _vec(mu, dim)
    K, U = C.c_double(), C.c_double()
    e = lib().oracle_leapfrog_tridiag(C.c_uint64(dim), _p(mu), _p(diag), _p(offdiag), _p(sig2), C.c_double(eps), C.c_int(waves), _p(q), _p(p), _p(g), C.byref(K), C.byref(U))
    return q, p, g, K.value, U.value, e


def dual_average(accept, initial_step=0.1, target=0.8, k=0.75, t0=10.0, gamma=0.05):
    accept = np.ascontiguousarray(accept, dtype=np.float64)
    step = np.empty_like(accept)
    bar = np.empty_like(accept)
    lib().oracle_dual_average(C.c_double(initial_step), C.c_double(target), C.c_double(k), C.c_double(t0), C.c_double(gamma), C.c_uint64(accept.size), _p(accept), _p(step), _p(bar))
    return step, bar


def welford(samples):
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    n, dim = samples.shape
    mean = np.empty(dim)
    m2 = np.empty(dim)
    lib().oracle_welford(C.c_uint64(n), C.c_uint64(dim), _p(samples), _p(mean), _p(m2))
    return mean, m2


def is_turning(sig2, idx1, p1, psum1, idx2, p2, psum2, waves=1):
    sig2 = np.ascontiguousarray(sig2, dtype=np.float64)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (p1, psum1, p2, psum2)]
    return bool(lib().oracle_is_turning(C.c_uint64(sig2.size), _p(sig2), C.c_int(waves), C.c_int64(idx1), _p(a[0]), _p(a[1]), C.c_int64(idx2), _p(a[2]), _p(a[3])))


def lr_velocity(sig2, V, lam, p, waves=1):
    """v = M^-1 p, M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 with V given as k rows."""
    sig2 = np.ascontiguousarray(sig2, dtype=np.float64)
    V = np.ascontiguousarray(V, dtype=np.float64).reshape(-1, sig2.size)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    v = np.empty_like(p)
    lib().oracle_lr_velocity(C.c_uint64(sig2.size), C.c_int(V.shape[0]), _p(sig2), _p(V), _p(lam), C.c_int(waves), _p(p), _p(v))
    return v
