"""``compile_stan_model`` — API shell of the reference's ``python/nutpie/compile_stan.py``.

The reference compiles Stan code with BridgeStan (``compile_stan.py:250-386``) and evaluates
``log_density_gradient(position, propto=True, jacobian=True)`` per chain-step from Rust
(``src/stan.rs:454-463``).  Here the same BridgeStan entry point is called, row by row on a host
thread pool, behind the engine's host-callback path (pinned ``hipMemcpyAsync`` both ways).

BridgeStan and the Stan toolchain are not installable in the build image, so model compilation
and its sha256 cache (``compile_stan.py:151-224``) are OUT OF SCOPE: this module needs an
importable ``bridgestan`` and otherwise raises ``ImportError``.  The native adapter underneath
(``nphip_model_bridgestan``, and ``nphip_model_set_bridgestan_expand`` for the expand step) is tested with stand-in
libraries that export BridgeStan's C API (tests/fixtures/eight_schools.c, tests/fixtures/bs_standin.c).
"""

from __future__ import annotations

import dataclasses
from functools import cached_property
import json
from dataclasses import dataclass
from importlib.util import find_spec
from typing import Any, Optional

import numpy as np

from nutpie_amd import _lib
from nutpie_amd.sample import CompiledModel


@dataclass(frozen=True)
class CompiledStanModel(CompiledModel):
    """Mirror of the reference's ``CompiledStanModel`` (compile_stan.py:17-148)."""

    code: str = ""
    data: Optional[dict[str, Any]] = None
    library: Any = None           # path of the compiled BridgeStan model library
    model: Any = None             # bridgestan.StanModel bound to `data`
    _coords: Optional[dict[str, Any]] = None

    def with_data(self, *, seed=None, **updates):
        import bridgestan

        data = dict(self.data or {})
        data.update(updates)
        model = bridgestan.StanModel(self.library, data=json.dumps({k: np.asarray(v).tolist() for k, v in data.items()}), seed=seed or 0)
        return dataclasses.replace(self, data=data, model=model)

    def with_coords(self, **coords):
        c = dict(self._coords or {})
        c.update(coords)
        return dataclasses.replace(self, _coords=c)

    def with_dims(self, **dims):
        d = dict(self.dims or {})
        d.update(dims)
        return dataclasses.replace(self, dims=d)

    @cached_property
    def _default_bound(self):
        return self.with_data().model

    def _bound(self):
        # (a model compiled without data is bound once, not on every n_dim / shapes access)
        return self.model if self.model is not None else self._default_bound

    @property
    def n_dim(self):
        return int(self._bound().param_unc_num())

    def _variables(self):
        from nutpie_amd.stan_names import parse_stan_variables

        return parse_stan_variables(",".join(self._bound().param_names(include_tp=True, include_gq=True)))

    @property
    def shapes(self):
        return {v.name: v.shape for v in self._variables()}

    @property
    def coords(self):
        return dict(self._coords or {})

    def _make_model(self, init_mean=None, settings=None):
        from nutpie_amd.stan_names import c_order_permutation

        m = self._bound()
        model = _lib.BridgeStanModel(self.n_dim, m.stanlib, m.model_rng if hasattr(m, "model_rng") else m.model, keep_alive=m)
        model.set_init("normal")  # src/stan.rs:798-808
        # the expand step runs behind the C-ABI: bs_param_constrain per draw with one bs_rng per chain, chains concurrently on
        # the host pool, column-major blocks re-ordered natively (src/stan.rs:473-520, 671-711, 787-796)
        variables = self._variables()
        if variables:
            model.set_bridgestan_expand(variables[-1].end, c_order_permutation(variables))
        return model

    def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
        return _lib.PySampler.from_stan(settings, cores, self._make_model(), progress_type, extra_callback, extra_callback_rate, store, **engine_kw)

    def _unflatten(self, flat):
        # [chain, draw, n_constrained], blocks already in C order -> variables (names and shapes parsed as src/stan.rs:93-251)
        return {v.name: flat[..., v.start:v.end].reshape(*flat.shape[:-1], *v.shape) for v in self._variables()}

    def _expand_draws(self, draws, seed: int = 0):
        raise RuntimeError("a Stan model expands behind the C-ABI (nphip_model_set_bridgestan_expand); it needs the sampler's stored draws")


def compile_stan_model(*, code: Optional[str] = None, filename: Optional[str] = None, extra_compile_args=None,
                       extra_stanc_args=None, dims=None, coords=None, model_name=None, cleanup: bool = True,
                       cache: bool = False, prune_cache: bool = True) -> CompiledStanModel:
    """Same keyword signature as the reference (compile_stan.py:250-262)."""
    if find_spec("bridgestan") is None:
        raise ImportError(
            "BridgeStan is not installed, please install it with something like 'pip install bridgestan'. "
            "(Stan compilation is outside the scope of the HIP engine; the BridgeStan *evaluation* path is "
            "nutpie_amd._lib.BridgeStanModel.)"
        )
    import pathlib
    import tempfile

    import bridgestan

    if code is not None and filename is not None:
        raise ValueError("Specify exactly one of `code` and `filename`")
    if code is None:
        if filename is None:
            raise ValueError("Either code or filename have to be specified")
        code = pathlib.Path(filename).read_text()
    basedir = pathlib.Path(tempfile.mkdtemp())
    name = model_name or "model"
    path = basedir / f"{name}.stan"
    path.write_text(code)
    so = bridgestan.compile_model(path, make_args=["STAN_THREADS=true", *(extra_compile_args or [])], stanc_args=extra_stanc_args or [])
    return CompiledStanModel(dims=dims or {}, code=code, data=None, library=str(so), model=None, _coords=coords or {})


def prune_stan_cache(*a, **k):
    """The compile cache is out of scope (see module docstring)."""
    return 0
