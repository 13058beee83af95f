"""nutpie_amd — an MI355X-native NUTS engine behind nutpie's Python API.

Public surface mirrors the reference's ``python/nutpie/__init__.py:10-18``
(``compile_pymc_model``, ``compile_stan_model``, ``sample``, ``ChainProgress``) and adds the
batched front-ends the GPU engine is built for (``from_torchfunc``, analytic Gaussians).
See DESIGN.md for the hot path and INTEGRATION.md for the C-ABI boundary.
"""

from nutpie_amd import _lib
from nutpie_amd._lib import __version__
from nutpie_amd.compile_pymc import compile_pymc_model
from nutpie_amd.compile_stan import compile_stan_model, prune_stan_cache
from nutpie_amd.compiled_pyfunc import from_pyfunc, from_torch_density, from_torchfunc
from nutpie_amd.density import from_density_source
from nutpie_amd import symbolic
from nutpie_amd.gaussian import ar1_gaussian, dense_gaussian, diag_gaussian, std_normal
from nutpie_amd.sample import CompiledModel, sample

ChainProgress = _lib.PyChainProgress


def zarr_store(*args, **kwargs):
    """Present for import compatibility (reference ``__init__.py:2``: ``zarr_store = _lib.store``); the HIP engine keeps
    the trace in HBM and hands it over as dense arrays, so Zarr storage is outside its scope."""
    raise NotImplementedError("zarr_store is outside the scope of the HIP engine (the trace lives in HBM)")


def install_as_nutpie(force: bool = False):
    """Make ``import nutpie`` resolve to this package (``nutpie``, ``nutpie.sample``, ``nutpie.compiled_pyfunc``,
    ``nutpie.compile_pymc``, ``nutpie.compile_stan``, ``nutpie._lib``): code written against the reference runs on the HIP
    engine unchanged.  Opt-in, per process; refuses to shadow an installed reference unless ``force``."""
    import importlib
    import importlib.util
    import sys

    if not force and "nutpie" not in sys.modules and importlib.util.find_spec("nutpie") is not None:
        raise RuntimeError("a real `nutpie` is importable in this environment; pass force=True to shadow it for this process")
    me = sys.modules[__name__]
    sys.modules["nutpie"] = me
    for sub in ("sample", "compiled_pyfunc", "compile_pymc", "compile_stan", "_lib"):
        sys.modules["nutpie." + sub] = importlib.import_module(__name__ + "." + sub)
    return me


__all__ = [
    "__version__",
    "ChainProgress",
    "CompiledModel",
    "compile_pymc_model",
    "compile_stan_model",
    "prune_stan_cache",
    "sample",
    "zarr_store",
    "from_pyfunc",
    "from_torchfunc",
    "from_torch_density",
    "from_density_source",
    "symbolic",
    "std_normal",
    "diag_gaussian",
    "ar1_gaussian",
    "dense_gaussian",
    "install_as_nutpie",
]
