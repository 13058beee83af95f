"""Generic models from Python callables.

``from_pyfunc`` / ``PyFuncModel`` keep the reference's signature and semantics
(``python/nutpie/compiled_pyfunc.py:14-155``; binding ``src/pyfunc.rs``): one
``logp_fn(x: float64[D]) -> (logp, grad)`` per chain, Uniform(-2, 2) default initial
points (``src/pyfunc.rs:540-544``), dict-returning expand function
(``src/pyfunc.rs:236-268``).  On the HIP engine such a model is evaluated through the
host-callback path (rows of the batch looped on the host) — a drop-in, not a fast path.

``from_torchfunc`` / ``TorchFuncModel`` is the batched device form the engine is built
for: ``logp_fn(x: Tensor[chains, D]) -> (logp[chains], grad[chains, D])`` on the GPU;
tensors never leave HBM.
"""

from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from functools import partial
from typing import Any, Callable

import numpy as np

from nutpie_amd import _lib
from nutpie_amd.sample import CompiledModel

SeedType = int


@dataclass(frozen=True)
class ExpandedLayout:
    """Names, shapes and dtypes of the expanded variables and their place in the flat fp64 vector the C-ABI expand callback
    fills (``RawExpandFunc``, src/pymc.rs:31-37; the reference keeps the same information in ``PyVariable`` objects)."""

    names: tuple[str, ...]
    shapes: tuple[tuple[int, ...], ...]
    dtypes: tuple[np.dtype, ...]

    @property
    def sizes(self):
        return tuple(int(np.prod(shape, dtype=np.int64)) for shape in self.shapes)

    @property
    def total(self) -> int:
        return int(sum(self.sizes))

    def as_shapes(self) -> dict[str, tuple[int, ...]]:
        return dict(zip(self.names, self.shapes))

    def flatten_row(self, values: dict) -> np.ndarray:
        """One draw's dict of variables -> flat fp64 row; dtype and size are checked (src/pyfunc.rs:290-380)."""
        parts = []
        for name, size, dtype in zip(self.names, self.sizes, self.dtypes):
            v = np.asarray(values[name])
            if v.dtype != dtype:
                raise TypeError(f"Expanded variable {name} has dtype {v.dtype}, expected {dtype}")
            if v.size != size:
                raise ValueError(f"Expanded variable {name} has incorrect shape")
            parts.append(v.reshape(size).astype(np.float64, copy=False))
        return np.concatenate(parts) if parts else np.empty(0)

    def split(self, flat: np.ndarray) -> dict[str, np.ndarray]:
        """[..., total] fp64 -> dict name -> [..., *shape] in the variable's dtype (one slice + cast per variable)."""
        out, start = {}, 0
        lead = flat.shape[:-1]
        for name, shape, size, dtype in zip(self.names, self.shapes, self.sizes, self.dtypes):
            block = flat[..., start:start + size].reshape(*lead, *shape)
            out[name] = block if dtype == np.float64 else _cast_block(block, dtype)
            start += size
        return out


def _cast_block(block: np.ndarray, dtype: np.dtype) -> np.ndarray:
    # rows of unfinished draws are NaN in the flat transport; integer / bool variables get 0 / False there
    filled = np.where(np.isnan(block), 0.0, block)
    return filled.astype(dtype)


@dataclass(frozen=True)
class PyFuncModel(CompiledModel):
    """A model given by Python callables, one position at a time (the reference's ``PyFuncModel``, compiled_pyfunc.py:14-106)."""

    logp_factory: Callable                      # () -> f(x, **data) -> (logp, grad)
    expand_factory: Callable                    # (seed1, seed2, chain) -> f(x, **data) -> dict
    initial_point_fn: Callable[[SeedType], np.ndarray] | None
    layout: ExpandedLayout
    unconstrained_dim: int
    data: dict[str, Any]
    coordinates: dict[str, Any]
    raw_logp_fn: Callable | None = None

    @property
    def shapes(self) -> dict[str, tuple[int, ...]]:
        return self.layout.as_shapes()

    @property
    def coords(self):
        return self.coordinates

    @property
    def n_dim(self):
        return self.unconstrained_dim

    def with_data(self, **updates):
        """A copy of the model with some shared data replaced (``PyFuncModel.with_data`` of the reference)."""
        unknown = next((name for name in updates if name not in self.data), None)
        if unknown is not None:
            raise ValueError(f"Unknown data variable: {unknown}")
        return dataclasses.replace(self, data={**self.data, **updates})

    def _init_points(self, settings) -> np.ndarray | None:
        """The reference calls ``init_point_func(seed)`` once per chain with a seed drawn from the
        chain's RNG (src/pyfunc.rs:546-568); here seeds derive from (seed, chain)."""
        if self.initial_point_fn is None:
            return None
        n = int(settings.num_chains)
        base = int(settings.seed)
        pts = np.empty((n, self.unconstrained_dim))
        for c in range(n):
            p = np.asarray(self.initial_point_fn((base * 0x9E3779B97F4A7C15 + c) % (1 << 64)), dtype=np.float64)
            if p.shape != (self.unconstrained_dim,):
                raise ValueError("Initial point has incorrect length")
            pts[c] = p
        return pts

    def _make_model(self, init_mean, settings=None):
        logp_fn = partial(self.logp_factory(), **self.data)

        def row_fn(x):
            val, grad = logp_fn(x)
            grad = np.asarray(grad)
            if grad.dtype != np.float64 or grad.shape != x.shape:
                raise TypeError("Return type of logp function should be (float, float64 array)")  # src/pyfunc.rs ReturnTypeError
            return float(val), grad

        model = _lib.HostCallbackModel(self.unconstrained_dim, row_fn)
        if self.layout.total:
            # the expand step goes through the C-ABI (nphip_model_set_expand): the engine walks the stored trace and calls
            # this row function per draw; splitting into variables is one numpy slice per variable afterwards
            expand_fn = partial(self.expand_factory(0, 0, 0), **self.data)
            model.set_expand(self.layout.total, lambda x: self.layout.flatten_row(expand_fn(x)))
        pts = self._init_points(settings) if settings is not None else None
        if pts is not None:
            model.set_init("explicit", pts)
        return model

    def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
        model = self._make_model(init_mean, settings)
        return _lib.PySampler.from_pyfunc(settings, cores, model, progress_type, extra_callback, extra_callback_rate, store, **engine_kw)

    def _unflatten(self, flat):
        return self.layout.split(flat)

    def _expand_draws(self, draws):
        """Host-side expand of an array of draws that did not come out of a sampler (tests, re-expansion)."""
        expand_fn = partial(self.expand_factory(0, 0, 0), **self.data)
        rows = draws.reshape(-1, draws.shape[-1])
        flat = np.stack([self.layout.flatten_row(expand_fn(r)) for r in rows]) if len(rows) else np.empty((0, self.layout.total))
        return self.layout.split(flat.reshape(*draws.shape[:-1], self.layout.total))


def from_pyfunc(
    ndim: int,
    make_logp_fn: Callable,
    make_expand_fn: Callable,
    expanded_dtypes: list[np.dtype],
    expanded_shapes: list[tuple[int, ...]],
    expanded_names: list[str],
    *,
    coords: dict[str, Any] | None = None,
    dims: dict[str, tuple[str, ...]] | None = None,
    shared_data: dict[str, Any] | None = None,
    make_initial_point_fn: Callable[[SeedType], np.ndarray] | None = None,
    make_transform_adapter=None,
    raw_logp_fn=None,
    reparameterized_names=None,
):
    """Same signature as the reference's ``from_pyfunc`` (compiled_pyfunc.py:108-155)."""
    if make_transform_adapter is not None:
        raise NotImplementedError("normalizing-flow adaptation is outside the scope of the HIP engine")
    layout = ExpandedLayout(
        names=tuple(expanded_names),
        shapes=tuple(tuple(int(n) for n in shape) for shape in expanded_shapes),
        dtypes=tuple(np.dtype(d) for d in expanded_dtypes),
    )
    return PyFuncModel(
        dims=dict(dims or {}),
        reparameterized_names=reparameterized_names,
        logp_factory=make_logp_fn,
        expand_factory=make_expand_fn,
        initial_point_fn=make_initial_point_fn,
        layout=layout,
        unconstrained_dim=int(ndim),
        data=dict(shared_data or {}),
        coordinates=dict(coords or {}),
        raw_logp_fn=raw_logp_fn,
    )


# ----------------------------------------------------------------------------- batched torch models
@dataclass(frozen=True)
class TorchFuncModel(CompiledModel):
    """Batched device model: all chains evaluated by one torch call per leapfrog."""

    _make_logp_func: Callable           # () -> f(x[chains, D]) -> (logp[chains], grad[chains, D])
    _expand_func: Callable | None       # (x[N, D] numpy) -> dict name -> [N, *shape]; None = identity variable "x"
    _n_dim: int
    _names: list[str]
    _shapes: list[tuple[int, ...]]
    _coords: dict[str, Any]
    _shared_data: dict[str, Any]
    _init: Any = "uniform"              # "uniform" | "normal" | ndarray [chains, D]
    _use_graph: bool = False            # capture logp+grad in a HIP graph (torch.cuda.CUDAGraph) and replay it per step
    _expand_device_func: Callable | None = None  # (x: Tensor[N, D] on the GPU) -> dict name -> Tensor[N, *shape]

    @property
    def shapes(self):
        return {n: tuple(int(v) for v in shp) for n, shp in zip(self._names, self._shapes)}

    @property
    def coords(self):
        return self._coords

    @property
    def n_dim(self):
        return self._n_dim

    def with_data(self, **updates):
        unknown = next((name for name in updates if name not in self._shared_data), None)
        if unknown is not None:
            raise ValueError(f"Unknown data variable: {unknown}")
        return dataclasses.replace(self, _shared_data={**self._shared_data, **updates})

    def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("TorchFuncModel needs a GPU: the nutpie-hip engine has no CPU fallback")
        device = int(engine_kw.get("device", 0) or 0)
        n = int(engine_kw.get("n_local_chains") or settings.num_chains)
        D = self._n_dim
        dev = torch.device("cuda", device)
        q = torch.zeros((n, D), dtype=torch.float64, device=dev)
        g = torch.zeros((n, D), dtype=torch.float64, device=dev)
        lp = torch.zeros((n,), dtype=torch.float64, device=dev)
        logp_fn = partial(self._make_logp_func(), **self._shared_data)
        # A function marked ``writes_staging`` has the form f(x, out_logp, out_grad) and fills the engine's buffers with `out=` operations
        # only: no copy kernels, and — nothing being allocated — the engine can capture `graph_steps` x (its kernel, this callback)
        # into one HIP graph (host.hip: iteration_graph)
        in_place = bool(getattr(logp_fn.func, "writes_staging", False))
        streams = {}
        graph = None
        if self._use_graph:
            # launch-bound models (dozens of small torch kernels per evaluation): capture once, replay per leapfrog
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    if in_place:
                        logp_fn(q, lp, g)
                    else:
                        val, grad = logp_fn(q)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                if in_place:
                    logp_fn(q, lp, g)
                else:
                    val, grad = logp_fn(q)
                    lp.copy_(val.reshape(n).to(torch.float64))
                    g.copy_(grad.reshape(n, D).to(torch.float64))
            q.zero_()

        q_base = q.data_ptr()

        def cb(n_chains, dim, _q, _g, _lp, stream_ptr):
            # run the torch work on the engine's stream: no cross-stream synchronisation needed
            st = streams.get(stream_ptr)
            if st is None:
                st = torch.cuda.ExternalStream(stream_ptr, device=dev) if stream_ptr else torch.cuda.default_stream(dev)
                streams[stream_ptr] = st
            # (engine option host_groups: the callback is asked for ONE group of chains — the pointers are rows of the staging buffers)
            lo = (int(_q) - q_base) // (8 * D) if _q else 0
            whole = lo == 0 and int(n_chains) == n
            qv, gv, lpv = (q, g, lp) if whole else (q[lo:lo + n_chains], g[lo:lo + n_chains], lp[lo:lo + n_chains])
            with torch.cuda.stream(st):
                if graph is not None and whole:
                    graph.replay()
                elif in_place:
                    logp_fn(qv, lpv, gv)         # writes the engine's staging buffers itself: no copies, nothing allocated
                else:
                    val, grad = logp_fn(qv)
                    lpv.copy_(val.reshape(-1).to(torch.float64))
                    gv.copy_(grad.reshape(-1, D).to(torch.float64))
            return 0

        model = _lib.DeviceCallbackModel(D, cb)
        if isinstance(self._init, str):
            model.set_init(self._init)
        else:
            model.set_init("explicit", np.asarray(self._init, dtype=np.float64))
        sampler = _lib.PySampler.from_pyfunc(
            settings, cores, model, progress_type, extra_callback, extra_callback_rate, store,
            staging=(q.data_ptr(), g.data_ptr(), lp.data_ptr()), **engine_kw,
        )
        sampler._keep_tensors = (q, g, lp, graph)
        if self._expand_device_func is not None:
            sampler._device_expand = partial(self._expand_on_device, device=device)
        return sampler

    def _expand_on_device(self, sampler, device=0, block=1 << 22):
        """Expand step (reference src/pyfunc.rs:236-391: one Python call per draw per chain) as a few batched
        torch calls over the engine's ``draws[chain, draw, D]`` buffer in HBM; only the expanded variables are
        copied to the host."""
        import torch

        from nutpie_amd.distributed import device_tensor

        n, T, D = sampler.num_chains, sampler.total_draws, self._n_dim
        ptr = sampler.device_ptr("draws")
        if not ptr:
            return None
        flat = device_tensor(ptr, (n * T, D), "float64", device)
        out = {name: np.empty((n * T, *shape)) for name, shape in zip(self._names, self._shapes)}
        rows = max(1, block // max(D, 1))
        with torch.no_grad():
            for lo in range(0, n * T, rows):
                vals = self._expand_device_func(flat[lo:lo + rows], **self._shared_data)
                for name, shape in zip(self._names, self._shapes):
                    out[name][lo:lo + rows] = vals[name].reshape(-1, *shape).to(torch.float64).cpu().numpy()
        return {name: out[name].reshape(n, T, *shape) for name, shape in zip(self._names, self._shapes)}

    def _make_model(self, *a, **k):
        raise NotImplementedError("TorchFuncModel builds its model inside _make_sampler (staging tensors are per sampler)")

    def _expand_draws(self, draws):
        n, T, D = draws.shape
        if self._expand_func is None:
            return {self._names[0]: draws.reshape(n, T, *self._shapes[0])}
        flat = self._expand_func(draws.reshape(n * T, D), **self._shared_data)
        out = {}
        for name, shape in zip(self._names, self._shapes):
            out[name] = np.asarray(flat[name]).reshape(n, T, *shape)
        return out


def from_torchfunc(
    ndim: int,
    make_logp_fn: Callable,
    expand_fn: Callable | None = None,
    expanded_shapes: list[tuple[int, ...]] | None = None,
    expanded_names: list[str] | None = None,
    *,
    coords: dict[str, Any] | None = None,
    dims: dict[str, tuple[str, ...]] | None = None,
    shared_data: dict[str, Any] | None = None,
    init="uniform",
    reparameterized_names=None,
    use_graph: bool = False,
    expand_device_fn: Callable | None = None,
):
    """Batched analogue of :func:`from_pyfunc`: ``make_logp_fn() -> f`` with
    ``f(x: Tensor[chains, ndim]) -> (logp: Tensor[chains], grad: Tensor[chains, ndim])`` on the GPU.

    ``expand_fn`` maps a numpy ``[N, ndim]`` block on the host; ``expand_device_fn`` (optional, preferred) maps a
    ``Tensor[N, ndim]`` on the GPU to a dict of tensors and is applied to the engine's draws buffer in HBM."""
    if expanded_names is None:
        expanded_names, expanded_shapes = ["x"], [(ndim,)]
        if expand_fn is not None or expand_device_fn is not None:
            raise ValueError("expand_fn needs expanded_names and expanded_shapes")
    return TorchFuncModel(
        dims=dict(dims or {}),
        _make_logp_func=make_logp_fn,
        _expand_func=expand_fn,
        _n_dim=ndim,
        _names=list(expanded_names),
        _shapes=[tuple(s) for s in expanded_shapes],
        _coords=dict(coords or {}),
        _shared_data=dict(shared_data or {}),
        _init=init,
        _use_graph=use_graph,
        _expand_device_func=expand_device_fn,
        reparameterized_names=reparameterized_names,
    )


def autograd_logp(density_fn: Callable) -> Callable:
    """``density_fn(x: Tensor[chains, ndim]) -> Tensor[chains]``  ->  ``f(x) -> (logp, grad)`` through ``torch.autograd``
    (rows are independent, so the gradient of ``logp.sum()`` is the batch of per-chain gradients)."""

    def logp(x):
        import torch

        xg = x.detach().requires_grad_(True)
        with torch.enable_grad():
            lp = density_fn(xg)
            (g,) = torch.autograd.grad(lp.sum(), xg)
        return lp.detach(), g

    return logp


def from_torch_density(ndim: int, density_fn: Callable, *, compile: Any = "auto", batched: bool = True, waves_per_chain: int | None = None, **kwargs):
    """A model from a torch log-density alone: ``density_fn(x: Tensor[chains, ndim], **shared_data) -> Tensor[chains]``.

    ``compile`` (default ``"auto"``): the function is TRACED once (:mod:`nutpie_amd.torch_trace`: ``torch.fx`` -> the front-end's
    expression graph -> symbolic gradient -> generated HIP density) and runs inside the model's own resident NUTS kernel — no torch
    call, no kernel launch and no memory round trip per gradient evaluation (the reference compiles a model's logp graph likewise,
    ``python/nutpie/compile_pymc.py:668-871``).  ``"auto"`` falls back to the eager path below when the function uses an operation
    the tracer cannot map (a ``UserWarning`` names it); ``True`` raises :class:`nutpie_amd.torch_trace.UnsupportedTorchOp` instead;
    ``False`` never traces.  ``batched=False``: ``density_fn(x: Tensor[ndim]) -> scalar`` (compiled path only).

    The eager path: a batched device callback, the gradient from ``torch.autograd`` (``use_graph=True`` replays the whole evaluation
    from a HIP graph).  Keyword arguments as :func:`from_torchfunc`; ``shared_data`` entries are passed to ``density_fn`` as keyword
    arguments (``with_data`` replaces them; a compiled model is traced again, its library is re-used when only values changed)."""
    if compile not in (True, False, "auto"):
        raise ValueError("compile must be True, False or 'auto'")
    if compile:
        from nutpie_amd.torch_trace import UnsupportedTorchOp, traced_model

        try:
            return traced_model(ndim, density_fn, batched=batched, waves_per_chain=waves_per_chain,
                                **{k: v for k, v in kwargs.items() if k not in ("use_graph", "expand_device_fn")})
        except (UnsupportedTorchOp, NotImplementedError, ValueError, RuntimeError) as e:
            # "auto": whatever keeps the function from becoming a resident kernel — an operation the tracer cannot map, a derivative the IR
            # does not have, dimensions the front-end cannot join, a library that does not compile or fit the LDS — leaves the eager path,
            # which runs any differentiable torch function.  compile=True reports it instead.
            if compile is True:
                raise
            import warnings

            warnings.warn(f"the torch log-density is evaluated eagerly (one launch per operation): {type(e).__name__}: {e}", UserWarning, stacklevel=2)
    if not batched:
        raise ValueError("batched=False needs the compiled path")

    def make_logp():
        def logp(x, **shared):
            return autograd_logp(lambda t: density_fn(t, **shared))(x)

        return logp

    return from_torchfunc(ndim, make_logp, **kwargs)
