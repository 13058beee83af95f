"""``compile_pymc_model`` — API shell of the reference's ``python/nutpie/compile_pymc.py``.

The reference turns a PyMC model into one joined-vector logp+gradient function
(``_make_functions``, compile_pymc.py:668-871) and hands nuts-rs a raw C function pointer made
with ``numba.cfunc`` (``_make_c_logp_func``, compile_pymc.py:970-1006; Rust side
``src/pymc.rs:21-62``).  The HIP engine accepts exactly that pointer
(:class:`nutpie_amd._lib.HostCallbackModel`, signature
``int64 logp(uint64 dim, const double* x, double* grad, double* logp, void* user_data)``), so a
PyMC model compiled by the reference's own machinery can be sampled unchanged.

PyMC / PyTensor / numba are not installable in the build image, so the graph compilation itself
is OUT OF SCOPE here: ``compile_pymc_model`` raises ``ImportError`` without ``pymc`` and
``NotImplementedError`` with it (no untested guess at PyMC's internals).  What IS supported is the
boundary below it: ``from_raw_callback`` takes the two raw C callbacks a reference-compiled model
carries (logp and expand).
"""

from __future__ import annotations

from dataclasses import dataclass
from importlib.util import find_spec
from typing import Any

import numpy as np



def from_raw_callback(n_dim: int, logp_address: int, user_data: int = 0, *, name: str = "x", n_threads: int = 0,
                      keep_alive: Any = None, init="uniform", expand_address: int | None = None, expand_user_data: int = 0,
                      expanded_shapes: dict[str, tuple[int, ...]] | None = None, dims=None, coords=None):
    """Sample a model given the addresses of the reference-style raw C callbacks a compiled PyMC model carries:
    the logp function (``numba.cfunc(c_sig)(logp_numba).address``, compile_pymc.py:334, 975-1004 — ``LogpFunc`` in
    src/pymc.rs:38-62) and, optionally, the expand function (compile_pymc.py:1018-1041 — ``ExpandFunc``,
    src/pymc.rs:64-95) with the shapes of the variables it writes, in order (``expanded_shapes``)."""
    if (expand_address is None) != (expanded_shapes is None):
        raise ValueError("expand_address and expanded_shapes go together")
    out_shapes = {name: (n_dim,)} if expanded_shapes is None else {k: tuple(int(n) for n in v) for k, v in expanded_shapes.items()}
    n_expanded = int(sum(np.prod(v, dtype=np.int64) for v in out_shapes.values()))
    from nutpie_amd import _lib
    from nutpie_amd.sample import CompiledModel

    @dataclass(frozen=True)
    class RawCallbackModel(CompiledModel):
        @property
        def n_dim(self):
            return n_dim

        @property
        def shapes(self):
            return out_shapes

        @property
        def coords(self):
            return dict(coords or {})

        def _make_model(self, init_mean=None, settings=None):
            m = _lib.HostCallbackModel(n_dim, int(logp_address), user_data, n_threads, keep_alive)
            if expand_address is not None:
                m.set_expand(n_expanded, int(expand_address), expand_user_data, keep_alive)
            if isinstance(init, str):
                m.set_init(init)
            else:
                m.set_init("explicit", np.asarray(init, dtype=np.float64))
            return m

        def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
            return _lib.PySampler.from_pymc(settings, cores, self._make_model(), progress_type, extra_callback, extra_callback_rate, store, **engine_kw)

        def _expand_draws(self, draws):
            if expand_address is not None:
                raise RuntimeError("this model expands behind the C-ABI (nphip_sampler_copy_expanded); it needs the sampler's stored draws")
            return {name: draws}

    return RawCallbackModel(dims=dict(dims or {}))


def compile_pymc_model(model, *, backend="numba", gradient_backend="pytensor", initial_points=None, jitter_rvs=None,
                       default_initialization_strategy="support_point", var_names=None, freeze_model=None, **kwargs):
    """Same keyword signature as the reference (compile_pymc.py:523-537).  A :class:`nutpie_amd.symbolic.Model` — this package's own
    model front-end — is compiled to its generated device density, with the reference's initial points (support point + U(-1, 1)
    jitter; ``initial_points`` a dict of per-variable values, or explicit start positions ``[chains, n_dim]``).  A callable is taken
    as a batched torch log-density ``logp(x[chains, n_dim]) -> [chains]`` (``n_dim=`` required) and goes through the tracer
    (:mod:`nutpie_amd.torch_trace`).  A PyMC model needs PyMC and PyTensor, which the target image does not have.

    ``var_names`` as the reference (compile_pymc.py:821-822): ``None`` stores every computed variable, ``[]`` none, a list only those named;
    the free variables are always stored.  ``freeze_model`` (compile_pymc.py:587-592: data and dimension lengths become constants of the
    compiled function): ``None`` / ``True`` = the front-end's default (data lengths are constants of the generated source; ``with_data``
    compiles again when one changes), ``False`` = one library for data of any length.  Keyword arguments the engine does not know are an
    error, not dropped."""
    from nutpie_amd import symbolic

    if backend not in ("numba", "jax", "hip"):
        raise ValueError(f"backend={backend!r}: the reference knows 'numba' and 'jax' (both name the engine here), this package also 'hip'")
    if isinstance(model, symbolic.Model):
        known = ("resident", "waves_per_chain", "coords", "dims")
        unknown = sorted(set(kwargs) - set(known))
        if unknown:
            raise TypeError(f"compile_pymc_model() got unexpected keyword arguments {unknown} (the front-end takes {list(known)})")
        extra = {k: kwargs[k] for k in known if k in kwargs}
        extra["var_names"] = var_names
        if freeze_model is not None:
            extra["specialize"] = bool(freeze_model)
        # the reference's initial points (compile_pymc.py:593-602): support point + U(-1, 1) jitter on `jitter_rvs` (default: every
        # free variable); `initial_points` = {variable: constrained value} overrides a support point.  An array [chains, n_dim]
        # is taken as explicit start positions; default_initialization_strategy="prior" has no counterpart (the front-end does not
        # know a variable's prior as a distribution) and is refused.
        if isinstance(initial_points, dict) or initial_points is None:
            if default_initialization_strategy not in ("support_point", "moment"):
                raise ValueError(f"default_initialization_strategy={default_initialization_strategy!r} is not supported (use 'support_point')")
            jitter = None if jitter_rvs is None else {getattr(v, "name", v) for v in jitter_rvs}
            init = model.jittered_init(initial_points, jitter)
        else:
            init = np.asarray(initial_points, dtype=np.float64)
        return model.compile(init=init, **extra)
    if callable(model) and not hasattr(model, "free_RVs"):
        # a torch log-density (what PyTensor's mode="PYTORCH" linker emits for a model's logp): traced and compiled
        from nutpie_amd.compiled_pyfunc import from_torch_density

        if "n_dim" not in kwargs:
            raise TypeError("compile_pymc_model(torch_logp, n_dim=...) needs the length of the unconstrained vector")
        from nutpie_amd.density import JitteredInit

        n_dim = int(kwargs.pop("n_dim"))
        if freeze_model is False:
            raise ValueError("freeze_model=False has no meaning for a traced torch log-density: the trace holds the data's shapes as constants")
        if var_names is not None:
            # a traced density reports what its expand function returns (default: the free vector `x`, which is always stored)
            names = kwargs.get("expanded_names")
            wanted = {getattr(v, "name", v) for v in var_names}
            if names is None:
                if wanted - {"x"}:
                    raise KeyError(f"var_names: the model reports no variable named {sorted(wanted - {'x'})}")
            else:
                unknown = wanted - set(names)
                if unknown:
                    raise KeyError(f"var_names: the model reports no variable named {sorted(unknown)}")
                keep = [i for i, n_ in enumerate(names) if n_ in wanted]
                user_expand = kwargs["expand_fn"]
                kept_names = [names[i] for i in keep]
                kwargs["expanded_names"] = kept_names
                kwargs["expanded_shapes"] = [kwargs["expanded_shapes"][i] for i in keep]
                kwargs["expand_fn"] = lambda x, **kw_: {k: v for k, v in user_expand(x, **kw_).items() if k in kept_names}
        if initial_points is None or (isinstance(initial_points, np.ndarray) and initial_points.ndim == 1):
            center = np.zeros(n_dim) if initial_points is None else np.asarray(initial_points, dtype=np.float64)
            init = JitteredInit(center=center, jitter=np.ones(n_dim) if jitter_rvs is None else np.asarray(jitter_rvs, dtype=np.float64))
        else:
            init = np.asarray(initial_points, dtype=np.float64)
        return from_torch_density(n_dim, model, compile=kwargs.pop("compile", True), init=init, **kwargs)
    if find_spec("pymc") is None:
        raise ImportError(
            "pymc is not installed in this environment.  Write the model with nutpie_amd.symbolic (expressions -> generated "
            "HIP density -> resident kernel), or use nutpie_amd.from_density_source (HIP source), nutpie_amd.from_torchfunc "
            "(batched torch logp), nutpie_amd.from_pyfunc, or nutpie_amd.compile_pymc.from_raw_callback with the numba cfunc "
            "address the reference produces."
        )
    raise NotImplementedError(
        "No PyTensor graph translator was built or tested here (PyTensor is not installable on the target image).  The "
        "back half of that pipeline exists: nutpie_amd.symbolic turns an expression graph into a HIP density with its "
        "gradient and compiles it into the resident kernel.  Otherwise compile the model with the reference and pass its "
        "numba cfunc addresses to nutpie_amd.compile_pymc.from_raw_callback(n_dim, logp_address, expand_address=..., "
        "expanded_shapes=...), or write the density for nutpie_amd.from_torchfunc / from_density_source."
    )
