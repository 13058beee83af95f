"""``nutpie.sample`` for the HIP engine.

Same call shape, keyword names and control flow as the reference's
``python/nutpie/sample.py:823-1102`` (``sample``), ``:17-59`` (``CompiledModel``) and
``:481-725`` (``_BackgroundSampler``); the sampler underneath is ``libnutpie_hip.so``
instead of ``nuts_rs::Sampler``.  Additions (keyword-only, all optional): ``device``,
``waves_per_chain``, ``store_draws``.
"""

from __future__ import annotations

import json
import os
import sys
import threading
import warnings
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

from nutpie_amd import _lib
from nutpie_amd.trace import build_trace


@dataclass(frozen=True)
class CompiledModel:
    """Abstract compiled model — reference sample.py:17-38."""

    dims: Optional[dict[str, tuple[str, ...]]]
    reparameterized_names: list[str] | None = field(default=None, kw_only=True)

    @property
    def n_dim(self) -> int:
        raise NotImplementedError()

    @property
    def shapes(self) -> Optional[dict[str, tuple[int, ...]]]:
        raise NotImplementedError()

    @property
    def coords(self):
        raise NotImplementedError()

    def _make_sampler(self, *args, **kwargs):
        raise NotImplementedError()

    def _make_model(self, *args, **kwargs):
        raise NotImplementedError()

    def _expand_draws(self, draws: np.ndarray) -> dict[str, np.ndarray]:
        """[chain, draw, n_dim] unconstrained positions -> dict of expanded variables."""
        raise NotImplementedError()

    def _unflatten(self, flat: np.ndarray) -> dict[str, np.ndarray]:
        """[chain, draw, expanded_dim] flat output of a C-ABI expand callback -> dict of variables (``shapes`` order, fp64)."""
        out, start = {}, 0
        for name, shape in self.shapes.items():
            size = int(np.prod(shape, dtype=np.int64))
            out[name] = flat[..., start:start + size].reshape(*flat.shape[:-1], *shape)
            start += size
        return out

    def benchmark_logp(self, point, num_evals, cores):
        raise NotImplementedError("benchmark_logp is not exposed (it is commented out in the reference binding too: src/pymc.rs:474-492)")


def _progress_line(progress) -> str:
    done = sum(p.finished_draws for p in progress)
    total = sum(p.total_draws for p in progress)
    div = sum(p.divergences for p in progress)
    steps = sum(p.total_num_steps for p in progress)
    return f"\rnutpie-hip: {done}/{total} draws, {len(progress)} chains, {div} divergences, {steps} leapfrogs"


class _BackgroundSampler:
    """Handle on a running sampler — reference sample.py:481-725."""

    def __init__(self, compiled_model, settings, init_mean, cores, *, progress_bar=True, progress_callback=None,
                 save_warmup=True, return_raw_trace=False, progress_template=None, progress_style=None,
                 progress_rate=100, store=None, store_unconstrained=False, engine_kwargs=None):
        self._settings = settings
        self._compiled_model = compiled_model
        self._save_warmup = save_warmup
        self._return_raw_trace = return_raw_trace
        self._store_unconstrained = store_unconstrained
        self._html = None
        if store is not None:
            raise NotImplementedError("zarr_store is outside the scope of the HIP engine (trace lives in HBM)")
        make = compiled_model._make_sampler
        if getattr(settings, "_adaptation", "diag") == "low_rank":
            # the low-rank metric lives in the engine; its window estimator drives the sampler from the host (nutpie_amd/low_rank.py)
            from functools import partial

            from nutpie_amd import low_rank

            make = partial(low_rank.make_sampler, compiled_model)
        self._sampler = make(settings, init_mean, cores, None, None, progress_rate, None, **(engine_kwargs or {}))
        # (the progress callback is driven by the poll thread below, not by the sampler handle)
        # raw unconstrained draws only cross PCIe when somebody asked for them (device-expanding models)
        self._sampler._keep_host_draws = bool(return_raw_trace or store_unconstrained or settings.store_unconstrained)
        self._stop = threading.Event()
        self._thread = None
        show_bar = bool(progress_bar) and sys.stderr.isatty()
        if progress_callback is not None or show_bar:
            rate = max(1, int(progress_rate)) / 1000.0

            def poll():
                # contract of `progress_callback`: reference sample.py:942-963
                while True:
                    finished = self._stop.wait(rate)
                    try:
                        if self._sampler.is_empty():
                            return
                        prog = self._sampler.progress()
                        if progress_callback is not None:
                            try:
                                progress_callback(prog)
                            except Exception as e:  # printed, not raised (sample.py:958-960)
                                print(f"Error in progress callback: {e!r}", file=sys.stderr)
                        if show_bar:
                            sys.stderr.write(_progress_line(prog))
                            sys.stderr.flush()
                    except Exception:
                        return
                    if finished:
                        if show_bar:
                            sys.stderr.write("\n")
                        return

            self._thread = threading.Thread(target=poll, daemon=True)
            self._thread.start()

    def _finish_progress(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def wait(self, *, timeout=None):
        """Wait until sampling is finished and return the trace (resumes a paused sampler)."""
        self._sampler.wait(timeout)
        self._finish_progress()
        results = self._sampler.take_results()
        return self._extract(results)

    def _extract(self, results):
        if self._return_raw_trace:
            return results
        settings_dict = self._settings.as_dict()
        st = settings_dict["settings"]
        stats = dict(results.stats)
        # optional statistics appear only when requested (reference sample.py:626-664)
        skips = []
        if not st.get("store_gradient", False):
            skips.append("gradient")
        if not st["adapt_options"]["mass_matrix_options"].get("store_mass_matrix", False):
            skips.append("mass_matrix_inv")
        if not st.get("store_divergences", False):
            skips += ["divergence_start", "divergence_end", "divergence_momentum", "divergence_start_gradient"]
        if st.get("store_unconstrained", False) and results.draws is not None:
            stats["unconstrained_draw"] = results.draws
        expanded = getattr(results, "expanded", None)
        if expanded is not None and "__flat__" in expanded:
            # the expand step ran behind the C-ABI (nphip_sampler_copy_expanded): one flat fp64 block per draw
            expanded = self._compiled_model._unflatten(expanded["__flat__"])
        if expanded is None:
            if results.draws is None:
                raise RuntimeError("the sampler was run with store_draws=False; use return_raw_trace=True")
            expanded = self._compiled_model._expand_draws(results.draws)
        attrs = {
            "inference_library": "nutpie",
            "inference_library_version": _lib.__version__,
            "inference_library_settings": json.dumps(settings_dict),
        }
        return build_trace(
            expanded, stats, results.finished,
            dims=self._compiled_model.dims, coords=self._compiled_model.coords, save_warmup=self._save_warmup,
            skip_vars=skips, reparameterized_names=self._compiled_model.reparameterized_names,
            keep_unconstrained_draw=self._store_unconstrained, attrs=attrs,
        )

    def inspect(self):
        """Get a copy of the current state of the trace."""
        return self._extract(self._sampler.inspect())

    def pause(self):
        self._sampler.pause()

    def resume(self):
        self._sampler.resume()

    @property
    def is_finished(self):
        return self._sampler.is_finished()

    def abort(self):
        """Abort sampling and return the trace produced so far."""
        self._sampler.abort()
        self._finish_progress()
        results = self._sampler.take_results()
        return self._extract(results)

    def cancel(self):
        """Abort sampling and discard progress."""
        if not self._sampler.is_empty():   # (an error after the results were taken — in the expand step, say — must stay visible)
            self._sampler.abort()
        self._finish_progress()
        self._sampler.close()

    def __del__(self):
        if not hasattr(self, "_sampler"):
            return
        try:
            if not self._sampler.is_empty(ignore_error=True):
                self.cancel()
        except Exception:
            pass

    def _repr_html_(self):
        return self._html


# the two flags older releases of the reference accepted in place of `adaptation=` (reference sample.py:979-1013)
_LEGACY_ADAPTATION_FLAGS = {"low_rank_modified_mass_matrix": "low_rank", "transform_adapt": "flow"}
_ADAPTATIONS = ("diag", "draw_diag", "low_rank", "flow")


def _legacy_adaptation(adaptation: str, kwargs: dict):
    """Pops the deprecated keywords out of ``kwargs``; returns (adaptation, use_grad_based or None).  A set legacy flag
    selects its adaptation (FutureWarning) unless ``adaptation`` was given too (ValueError)."""
    for flag, implied in _LEGACY_ADAPTATION_FLAGS.items():
        if not kwargs.pop(flag, False):
            continue
        warnings.warn(f"`{flag}` is deprecated. Use `adaptation='{implied}'` instead.", FutureWarning, stacklevel=3)
        if adaptation != "diag":
            raise ValueError(f"`{flag}` is deprecated and cannot be combined with the `adaptation` argument.")
        adaptation = implied
    grad_based = None
    if "use_grad_based_mass_matrix" in kwargs:
        grad_based = kwargs.pop("use_grad_based_mass_matrix")
        warnings.warn("`use_grad_based_mass_matrix` is deprecated. Use `adaptation='draw_diag'` instead of "
                      "`use_grad_based_mass_matrix=False`.", FutureWarning, stacklevel=3)
    return adaptation, grad_based


def _settings_for(sampler: str, adaptation: str, seed):
    families = {"nuts": _lib.PyNutsSettings, "mclmc": _lib.PyMclmcSettings}
    if sampler not in families:
        raise ValueError(f"Unknown sampler '{sampler}'. Expected one of: 'nuts', 'mclmc'.")
    if adaptation not in _ADAPTATIONS:
        raise ValueError(f"Unknown adaptation strategy '{adaptation}'. Expected one of: 'diag', 'draw_diag', 'low_rank', 'flow'.")
    constructor = {"low_rank": "LowRank", "flow": "Flow"}.get(adaptation, "Diag")
    return getattr(families[sampler], constructor)(seed)


def _host_cores() -> int:
    counter = getattr(os, "process_cpu_count", os.cpu_count)  # process_cpu_count: Python >= 3.13
    return counter() or 1


def sample(
    compiled_model: CompiledModel,
    *,
    draws: int | None = None,
    tune: int | None = None,
    chains: int | None = None,
    cores: int | None = None,
    seed: int | None = None,
    save_warmup: bool = True,
    progress_bar: bool = True,
    sampler: str = "nuts",
    adaptation: str = "diag",
    init_mean: np.ndarray | None = None,
    return_raw_trace: bool = False,
    blocking: bool = True,
    progress_callback: Any | None = None,
    progress_template: str | None = None,
    progress_style: str | None = None,
    progress_rate: int = 100,
    zarr_store=None,
    store_unconstrained: bool = False,
    device: int | None = None,
    waves_per_chain: int = 0,
    store_draws: bool = True,
    host_groups: int = 0,
    **kwargs,
):
    """Sample the posterior of a compiled model on an MI355X.

    Keyword arguments, defaults and error behaviour follow ``nutpie.sample``
    (reference sample.py:823-1102).  ``adaptation`` accepts ``"diag"`` (default, draw+gradient
    variance), ``"draw_diag"`` and ``"low_rank"`` (every model flavour: the metric is applied inside the engine) with
    ``mass_matrix_eigval_cutoff`` / ``mass_matrix_gamma``; ``"flow"`` and ``sampler="mclmc"`` raise
    ``NotImplementedError`` (out of scope for the HIP engine).  ``cores`` is accepted and ignored:
    all chains run concurrently on the GPU.

    Engine-specific: ``waves_per_chain`` (0 = a function of the dimension only, so that a chain's result never
    depends on how many chains run with it; with fewer than ~256 chains and a fused model of D >= 512,
    ``waves_per_chain=4`` is about 20 % faster), ``store_draws``, ``device``, ``host_groups`` (callback models: the chains in that
    many groups, each group's engine kernel and callback on a stream of its own — one group's callback overlaps another's kernel).
    """
    # behaviour (accepted keywords, warnings, error texts) documented at reference sample.py:979-1070; written independently
    adaptation, grad_based = _legacy_adaptation(adaptation, kwargs)
    settings = _settings_for(sampler, adaptation, seed)
    if adaptation == "draw_diag" or grad_based is False:
        settings.use_grad_based_mass_matrix = False
    overrides = {"num_tune": tune, "num_draws": draws, "num_chains": chains}
    settings.update({**kwargs, **{k: v for k, v in overrides.items() if v is not None}})
    if store_unconstrained:
        settings.store_unconstrained = True
    if cores is None:
        cores = _host_cores() if chains is None else min(chains, _host_cores())
    if init_mean is None:
        init_mean = np.zeros(compiled_model.n_dim)

    engine_kwargs = {"waves_per_chain": waves_per_chain, "store_draws": store_draws}
    if host_groups:
        engine_kwargs["host_groups"] = int(host_groups)
    if device is not None:
        engine_kwargs["device"] = device

    background_sampler = _BackgroundSampler(
        compiled_model, settings, init_mean, cores,
        progress_bar=progress_bar, progress_callback=progress_callback, save_warmup=save_warmup,
        return_raw_trace=return_raw_trace, progress_template=progress_template, progress_style=progress_style,
        progress_rate=progress_rate, store=zarr_store, store_unconstrained=store_unconstrained,
        engine_kwargs=engine_kwargs,
    )
    if not blocking:
        return background_sampler
    try:
        result = background_sampler.wait()
    except KeyboardInterrupt:
        result = background_sampler.abort()
    except BaseException:
        background_sampler.cancel()
        raise
    return result
