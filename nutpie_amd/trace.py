"""Trace container: the same groups and layout the reference hands to ArviZ.

The reference converts per-chain Arrow record batches to ``[chain, draw, *shape]`` numpy
arrays, splits warm-up from posterior with the boolean ``tuning`` column and NaN-pads
chains of unequal length (``python/nutpie/sample.py:62-214``).  The HIP engine already
produces dense ``[chain, draw, ...]`` arrays; this module reproduces the grouping.  When
``arviz`` is importable the result is converted with ``arviz.from_dict`` exactly as the
reference does; otherwise a light-weight stand-in with attribute access is returned
(``trace.posterior.x.values``, ``trace.sample_stats.diverging`` ...).
"""

from __future__ import annotations

from importlib.util import find_spec

import numpy as np


class Var(np.ndarray):
    """ndarray with the two xarray attributes user code touches most (``.values``, ``.dims``)."""

    def __new__(cls, arr, dims=()):
        obj = np.asarray(arr).view(cls)
        obj.dims = tuple(dims)
        return obj

    def __array_finalize__(self, obj):
        self.dims = getattr(obj, "dims", ())

    @property
    def values(self):
        return np.asarray(self)


class Dataset(dict):
    """name -> Var, with attribute access and ``attrs`` (stand-in for ``xarray.Dataset``)."""

    def __init__(self, data=None, dims=None, attrs=None):
        super().__init__()
        dims = dims or {}
        for k, v in (data or {}).items():
            self[k] = Var(v, ("chain", "draw", *dims.get(k, [f"{k}_dim_{i}" for i in range(np.ndim(v) - 2)])))
        self.attrs = dict(attrs or {})

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    @property
    def data_vars(self):
        return self


class DataTree(dict):
    """group name -> Dataset, with attribute access (stand-in for ``xarray.DataTree``)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def groups(self):
        return list(self.keys())


def _pad_split(arr, n_tune_per_chain, finished, max_tune, max_post):
    """[chain, T, ...] -> (warmup [chain, max_tune, ...], posterior [chain, max_post, ...]), NaN/zero padded
    as ``_add_arrow_data`` does (sample.py:167-214)."""
    n = arr.shape[0]
    item = arr.shape[2:]
    if n and np.all(n_tune_per_chain == n_tune_per_chain[0]) and np.all(finished == finished[0]):
        # every chain got equally far (the normal case): no padding, so no copy of a possibly multi-GB array
        nt, nf = int(n_tune_per_chain[0]), int(finished[0])
        return arr[:, :nt], arr[:, nt:nf]
    if arr.dtype.kind == "f":
        tune = np.full((n, max_tune, *item), np.nan, dtype=arr.dtype)
        post = np.full((n, max_post, *item), np.nan, dtype=arr.dtype)
    else:
        tune = np.zeros((n, max_tune, *item), dtype=arr.dtype)
        post = np.zeros((n, max_post, *item), dtype=arr.dtype)
    for c in range(n):
        nt, nf = int(n_tune_per_chain[c]), int(finished[c])
        tune[c, :nt] = arr[c, :nt]
        post[c, : nf - nt] = arr[c, nt:nf]
    return tune, post


def _to_arviz(groups, dims, coords, attrs):
    """ArviZ < 1.0 takes the groups as keyword arguments (InferenceData), >= 1.0 as one dict (DataTree); the reference
    supports both and branches on the installed version (python/nutpie/sample.py:122-147)."""
    from importlib.metadata import version

    import arviz

    major_minor = tuple(int(x) for x in version("arviz").split(".")[:2])
    if major_minor >= (1, 0):
        out = arviz.from_dict(groups, dims=dims, coords=coords)
    else:
        out = arviz.from_dict(**groups, dims=dims, coords=coords)
    try:
        out["sample_stats"].attrs.update(attrs)
    except Exception:
        pass
    return out


def build_trace(expanded, stats, finished, *, dims=None, coords=None, save_warmup=True, skip_vars=(),
                reparameterized_names=None, keep_unconstrained_draw=False, attrs=None, use_arviz=None):
    """Assemble the output of ``nutpie.sample``.

    expanded: dict name -> [chain, T, *shape]   (constrained / expanded variables)
    stats:    dict name -> [chain, T(, dim)]    (must contain the bool column ``tuning``)
    finished: [chain] number of completed draws per chain
    """
    dims = dict(dims or {})
    finished = np.asarray(finished, dtype=np.int64)
    tuning = np.asarray(stats["tuning"], dtype=bool)
    n_tune = np.array([int(tuning[c, : finished[c]].sum()) for c in range(len(finished))], dtype=np.int64)
    max_tune = int(n_tune.max()) if len(n_tune) else 0
    max_post = int((finished - n_tune).max()) if len(n_tune) else 0

    data_tune, data_post, stats_tune, stats_post = {}, {}, {}, {}
    for k, v in expanded.items():
        data_tune[k], data_post[k] = _pad_split(np.asarray(v), n_tune, finished, max_tune, max_post)
    for k, v in stats.items():
        if k in skip_vars:
            continue
        stats_tune[k], stats_post[k] = _pad_split(np.asarray(v), n_tune, finished, max_tune, max_post)

    reparameterized_names = list(reparameterized_names or [])
    uc_post = {k: data_post.pop(k) for k in reparameterized_names if k in data_post}
    uc_tune = {k: data_tune.pop(k) for k in reparameterized_names if k in data_tune}

    groups = {"posterior": data_post, "sample_stats": stats_post}
    if save_warmup:
        groups["warmup_posterior"] = data_tune
        groups["warmup_sample_stats"] = stats_tune
    if keep_unconstrained_draw and uc_post:
        groups["unconstrained_posterior"] = uc_post
        if save_warmup and uc_tune:
            groups["warmup_unconstrained_posterior"] = uc_tune

    if use_arviz is None:
        use_arviz = find_spec("arviz") is not None
    if use_arviz:
        try:
            return _to_arviz(groups, dims, coords, attrs or {})
        except Exception as e:  # the GPU job is done: never lose its trace to a conversion problem
            import warnings

            warnings.warn(f"conversion to an ArviZ object failed ({e!r}); returning the built-in DataTree", RuntimeWarning, stacklevel=2)

    stat_dims = {
        k: [f"unconstrained_parameter"] for k in ("gradient", "unconstrained_draw", "mass_matrix_inv", "divergence_start",
                                                    "divergence_end", "divergence_momentum", "divergence_start_gradient")
    }
    tree = DataTree()
    for gname, g in groups.items():
        is_stats = gname.endswith("sample_stats")
        tree[gname] = Dataset(g, dims=stat_dims if is_stats else dims, attrs=(attrs or {}) if is_stats else {})
    tree.coords = dict(coords or {})
    return tree
