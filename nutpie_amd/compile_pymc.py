"""``compile_pymc_model`` — API shell of the reference's ``python/nutpie/compile_pymc.py``.

The reference turns a PyMC model into one joined-vector logp+gradient function
(``_make_functions``, compile_pymc.py:668-871) and hands nuts-rs a raw C function pointer made
with ``numba.cfunc`` (``_make_c_logp_func``, compile_pymc.py:970-1006; Rust side
``src/pymc.rs:21-62``).  The HIP engine accepts exactly that pointer
(:class:`nutpie_amd._lib.HostCallbackModel`, signature
``int64 logp(uint64 dim, const double* x, double* grad, double* logp, void* user_data)``), so a
PyMC model compiled by the reference's own machinery can be sampled unchanged.

PyMC / PyTensor / numba are not installable in the build image, so the graph compilation itself
is OUT OF SCOPE here: this function requires an importable ``pymc`` and raises ``ImportError``
otherwise.  When PyMC is present it compiles value-and-gradient through PyTensor's default
backend and evaluates it behind the host-callback path.
"""

from __future__ import annotations

from dataclasses import dataclass
from importlib.util import find_spec
from typing import Any

import numpy as np

from nutpie_amd.compiled_pyfunc import from_pyfunc


def from_raw_callback(n_dim: int, logp_address: int, user_data: int = 0, *, name: str = "x", n_threads: int = 0,
                      keep_alive: Any = None, init="uniform"):
    """Sample a model given the address of a reference-style raw C logp callback
    (e.g. ``numba.cfunc(c_sig)(logp_numba).address``, compile_pymc.py:334, 975-1004)."""
    from nutpie_amd import _lib
    from nutpie_amd.sample import CompiledModel

    @dataclass(frozen=True)
    class RawCallbackModel(CompiledModel):
        @property
        def n_dim(self):
            return n_dim

        @property
        def shapes(self):
            return {name: (n_dim,)}

        @property
        def coords(self):
            return {}

        def _make_model(self, init_mean=None, settings=None):
            m = _lib.HostCallbackModel(n_dim, int(logp_address), user_data, n_threads, keep_alive)
            if isinstance(init, str):
                m.set_init(init)
            else:
                m.set_init("explicit", np.asarray(init, dtype=np.float64))
            return m

        def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
            return _lib.PySampler.from_pymc(settings, cores, self._make_model(), progress_type, extra_callback, extra_callback_rate, store, **engine_kw)

        def _expand_draws(self, draws):
            return {name: draws}

    return RawCallbackModel(dims={})


def compile_pymc_model(model, *, backend="numba", gradient_backend="pytensor", initial_points=None, jitter_rvs=None,
                       default_initialization_strategy="support_point", var_names=None, freeze_model=None, **kwargs):
    """Same keyword signature as the reference (compile_pymc.py:523-537)."""
    if find_spec("pymc") is None:
        raise ImportError(
            "pymc is not installed in this environment.  PyMC graph compilation is outside the scope of the HIP "
            "engine; use nutpie_amd.from_torchfunc (batched torch logp), nutpie_amd.from_pyfunc, or "
            "nutpie_amd.compile_pymc.from_raw_callback with the numba cfunc address the reference produces."
        )
    import pymc as pm  # pragma: no cover - not installable in the build image
    from pymc.initial_point import make_initial_point_fn  # pragma: no cover

    if backend.lower() not in ("numba", "jax"):  # pragma: no cover
        raise ValueError(f"Backend must be one of numba and jax. Got {backend}")
    fn = model.logp_dlogp_function(ravel_inputs=True)  # pragma: no cover
    n_dim = int(fn._extra_vars_shared and fn.size or fn.size)  # pragma: no cover
    value_vars = list(model.value_vars)  # pragma: no cover
    names = [v.name for v in value_vars]  # pragma: no cover
    ip = model.initial_point()  # pragma: no cover
    shapes = [tuple(np.shape(ip[nm])) for nm in names]  # pragma: no cover
    sizes = [int(np.prod(s, dtype=np.int64)) for s in shapes]  # pragma: no cover

    def make_logp():  # pragma: no cover
        def logp(x, **_):
            val, grad = fn(x)
            return float(val), np.asarray(grad, dtype=np.float64)

        return logp

    def make_expand(*_):  # pragma: no cover
        def expand(x, **_):
            out, o = {}, 0
            for nm, shp, sz in zip(names, shapes, sizes):
                out[nm] = np.asarray(x[o:o + sz]).reshape(shp)
                o += sz
            return out

        return expand

    init_fn = make_initial_point_fn(model=model, overrides=initial_points, jitter_rvs=set(model.free_RVs) if jitter_rvs is None else jitter_rvs,
                                    default_strategy=default_initialization_strategy, return_transformed=True)  # pragma: no cover

    def make_initial_point(seed):  # pragma: no cover
        pt = init_fn(seed)
        return np.concatenate([np.ravel(pt[nm]) for nm in names]).astype(np.float64)

    return from_pyfunc(n_dim, make_logp, make_expand, [np.float64] * len(names), shapes, names,
                       make_initial_point_fn=make_initial_point)  # pragma: no cover
