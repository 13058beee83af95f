"""ctypes binding of ``libnutpie_hip.so`` shaped like the reference's ``nutpie._lib``.

The reference's ``_lib`` is a PyO3 module (``src/wrapper.rs``); this module exposes
objects with the same names and methods for the diag-NUTS path, implemented on top of
the C-ABI declared in ``include/nutpie_hip.h``:

========================  ==========================================================
reference (file:line)     here
========================  ==========================================================
``PyNutsSettings``        :class:`PyNutsSettings`  (wrapper.rs:106-826)
``PyChainProgress``       :class:`PyChainProgress` (wrapper.rs:47-104)
``PySampler``             :class:`PySampler`       (wrapper.rs:953-1457)
``PyMcModel``/``LogpFunc``  :class:`HostCallbackModel` (src/pymc.rs:21-62, 188-215)
``PyModel``               :class:`DeviceCallbackModel` (src/pyfunc.rs:201-230, batched)
``PyTrace``               :class:`PyTrace`         (wrapper.rs:1459-1495)
========================  ==========================================================

There is no CPU fallback: creating a sampler without the HIP library or without a GPU
raises.
"""

from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("NUTPIE_HIP_LIB") or os.path.join(_HERE, "libnutpie_hip.so")  # override: A/B measurements
_CSRC = os.path.join(_HERE, "csrc")

__version__ = "0.1.0"

NPHIP_OK = 0
NPHIP_ERR_UNKNOWN_ATTR = -2
NPHIP_ERR_NOT_AVAILABLE = -3
NPHIP_ERR_BAD_VALUE = -4
WAIT_DONE, WAIT_TIMEOUT, WAIT_ERROR = 0, 1, 2

# src/pymc.rs:23-29 / 31-37: both raw callbacks return `std::os::raw::c_int`
RAW_LOGP_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
RAW_EXPAND_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
DEVICE_EXPAND_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
DEVICE_LOGP_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


class _Launch(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("waves_per_chain", C.c_int32),
        ("chain_offset", C.c_uint64),
        ("n_local_chains", C.c_uint64),
        ("stream", C.c_void_p),
        ("store_draws", C.c_int32),
        ("evals_per_launch", C.c_int32),
        ("start_paused", C.c_int32),
        ("manual", C.c_int32),
        ("staging_q", C.c_void_p),
        ("staging_grad", C.c_void_p),
        ("staging_logp", C.c_void_p),
        ("no_register_kernel", C.c_int32),
        ("no_stream_cache", C.c_int32),
        ("graph_steps", C.c_int32),
        ("host_groups", C.c_int32),
        ("host_persist", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class _Progress(C.Structure):
    _fields_ = [
        ("finished_draws", C.c_uint64),
        ("total_draws", C.c_uint64),
        ("divergences", C.c_uint64),
        ("tuning", C.c_int32),
        ("started", C.c_int32),
        ("latest_num_steps", C.c_uint64),
        ("total_num_steps", C.c_uint64),
        ("step_size", C.c_double),
        ("runtime_ms", C.c_uint64),
    ]


_lib = None
_lib_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("kernels.hip", "host.hip", "linalg.hip", "lowrank_est.hip", "engine_types.h", "dense_tile.h", "Makefile")]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("nutpie_hip.h", "nphip_spec.h")]
    stale = not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-j", str(max(1, min(8, os.cpu_count() or 1))), "-C", _CSRC] + (["-B"] if force else []), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building libnutpie_hip.so failed:\n" + r.stdout + r.stderr)
    return _LIB_PATH


def lib():
    """Load the HIP engine.  Fails loudly if the extension is missing."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise ImportError(
                    f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(the nutpie-hip engine has no CPU fallback)"
                )
            # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.
            # If torch is installed, let it load its runtime FIRST; our DT_NEEDED `libamdhip64.so.7` then resolves
            # (by SONAME) to the copy already in the process instead of pulling /opt/rocm's as a second runtime —
            # two HSA runtimes initialised in the wrong order leave the later one without devices.
            from importlib.util import find_spec

            if find_spec("torch") is not None:
                try:
                    import torch  # noqa: F401
                except Exception:  # pragma: no cover - a broken torch must not break the engine
                    pass
            L = C.CDLL(_LIB_PATH)
            L.nphip_last_error.restype = C.c_char_p
            L.nphip_version.restype = C.c_char_p
            L.nphip_settings_new_diag.restype = C.c_void_p
            L.nphip_settings_new_diag.argtypes = [C.c_uint64]
            L.nphip_settings_clone.restype = C.c_void_p
            L.nphip_settings_clone.argtypes = [C.c_void_p]
            L.nphip_settings_free.argtypes = [C.c_void_p]
            L.nphip_settings_set_f64.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
            L.nphip_settings_set_u64.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
            L.nphip_settings_set_bool.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
            L.nphip_settings_set_str.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
            L.nphip_settings_get_f64.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
            L.nphip_settings_get_u64.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
            L.nphip_settings_to_json.restype = C.c_int64
            L.nphip_settings_to_json.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
            L.nphip_model_tridiag_gaussian.restype = C.c_void_p
            L.nphip_model_tridiag_gaussian.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
            L.nphip_model_dense_gaussian.restype = C.c_void_p
            L.nphip_model_dense_gaussian.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p]
            L.nphip_test_dense_grad.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.nphip_model_host_callback.restype = C.c_void_p
            L.nphip_model_host_callback.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
            L.nphip_model_device_callback.restype = C.c_void_p
            L.nphip_model_device_callback.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p]
            L.nphip_model_bridgestan.restype = C.c_void_p
            L.nphip_model_bridgestan.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            L.nphip_model_jit_density.restype = C.c_void_p
            L.nphip_model_jit_density.argtypes = [C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
            L.nphip_model_jit_low_rank.argtypes = [C.c_void_p, C.c_int]
            L.nphip_model_set_init.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
            L.nphip_model_free.argtypes = [C.c_void_p]
            L.nphip_model_set_expand.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
            L.nphip_model_set_device_expand.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
            L.nphip_model_set_bridgestan_expand.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            L.nphip_model_expanded_dim.argtypes = [C.c_void_p]
            L.nphip_model_expanded_dim.restype = C.c_uint64
            L.nphip_sampler_copy_expanded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
            L.nphip_settings_set_pause_draws.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
            L.nphip_sampler_waiting.argtypes = [C.c_void_p, C.c_void_p]
            L.nphip_sampler_waiting.restype = C.c_int64
            L.nphip_sampler_resume_at.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
            L.nphip_sampler_set_metric.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            L.nphip_sampler_stage_metric.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            L.nphip_sampler_chain_draws.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            L.nphip_sampler_chain_draws.restype = C.c_int64
            L.nphip_sampler_set_evals_per_launch.argtypes = [C.c_void_p, C.c_int32]
            L.nphip_sampler_release.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
            L.nphip_batched_eigh.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
            L.nphip_low_rank_estimate_supported.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
            L.nphip_low_rank_estimate.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                  C.c_double, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
            L.nphip_test_eigh_stage.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            L.nphip_launch_defaults.argtypes = [C.POINTER(_Launch)]
            L.nphip_default_evals_per_launch.argtypes = [C.c_uint64]
            L.nphip_abi_struct_size.restype = C.c_uint64
            L.nphip_abi_struct_size.argtypes = [C.c_int]
            if L.nphip_abi_struct_size(0) != C.sizeof(_Launch) or L.nphip_abi_struct_size(1) != C.sizeof(_Progress):
                raise RuntimeError("libnutpie_hip.so and nutpie_amd/_lib.py disagree on the C-ABI struct layouts (stale build?)")
            L.nphip_sampler_create.restype = C.c_void_p
            L.nphip_sampler_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Launch)]
            L.nphip_sampler_free.argtypes = [C.c_void_p]
            L.nphip_sampler_wait.argtypes = [C.c_void_p, C.c_int64]
            L.nphip_sampler_step.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
            for f in ("pause", "resume", "abort", "is_finished", "waves_per_chain", "host_mode"):
                getattr(L, "nphip_sampler_" + f).argtypes = [C.c_void_p]
            for f in ("num_chains", "dim", "total_draws", "launches"):
                getattr(L, "nphip_sampler_" + f).argtypes = [C.c_void_p]
                getattr(L, "nphip_sampler_" + f).restype = C.c_uint64
            L.nphip_sampler_seconds.argtypes = [C.c_void_p]
            L.nphip_sampler_seconds.restype = C.c_double
            L.nphip_sampler_progress.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_Progress)]
            L.nphip_sampler_finished_draws.argtypes = [C.c_void_p, C.c_void_p]
            L.nphip_sampler_copy_stat.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
            L.nphip_sampler_device_ptr.restype = C.c_void_p
            L.nphip_sampler_device_ptr.argtypes = [C.c_void_p, C.c_char_p]
            L.nphip_test_detmath.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
            L.nphip_test_dot.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
            L.nphip_test_rowpool.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
            _lib = L
    return _lib


def _err() -> str:
    return lib().nphip_last_error().decode()


def _check_setting(rc: int):
    # same two exception classes as wrapper.rs:138-145 (ValueError) and 610-614 (AttributeError)
    if rc == NPHIP_OK:
        return
    if rc == NPHIP_ERR_UNKNOWN_ATTR:
        raise AttributeError(_err())
    raise ValueError(_err())


# --------------------------------------------------------------------------- settings
_BOOL_KEYS = {
    "check_turning", "store_mass_matrix", "use_grad_based_mass_matrix", "store_unconstrained", "store_gradient",
    "store_transformed", "store_divergences", "train_on_orbit", "microcanonical_trajectory",
    "exact_normal_trajectory", "adapt_mass_matrix", "low_rank_metric",
}
_U64_KEYS = {
    "num_tune", "num_draws", "num_chains", "maxdepth", "mindepth", "window_switch_freq", "mass_matrix_switch_freq",
    "early_window_switch_freq", "mass_matrix_update_freq", "extra_doublings", "seed", "num_try_init",
}
_OPT_F64_KEYS = {"target_integration_time", "mass_matrix_eigval_cutoff", "mass_matrix_gamma", "step_size_adam_learning_rate", "step_size_jitter"}


class PyNutsSettings:
    """Settings object with the reference's flat attribute names (wrapper.rs:210-451, 563-620)."""

    __slots__ = ("_h", "_adaptation", "_low_rank")

    def __init__(self, handle, adaptation="diag", low_rank=None):
        object.__setattr__(self, "_h", handle)
        object.__setattr__(self, "_adaptation", adaptation)
        # options of adaptation="low_rank" (src/wrapper.rs:307-334); kept on the Python side, where that adaptation lives
        object.__setattr__(self, "_low_rank", dict(low_rank or {}))

    # wrapper.rs:717-737
    @staticmethod
    def Diag(seed=None):
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")  # random_seed(), wrapper.rs:453-458
        return PyNutsSettings(C.c_void_p(lib().nphip_settings_new_diag(C.c_uint64(int(seed)))))

    @staticmethod
    def LowRank(seed=None):
        """``PyNutsSettings::LowRank`` (wrapper.rs:725-729).  The engine integrates under the low-rank metric (setting
        ``low_rank_metric``); the window estimator that supplies it lives in nutpie_amd/low_rank.py; ``mass_matrix_eigval_cutoff`` (> 1) and ``mass_matrix_gamma`` (> 0) as in
        python/nutpie/sample.py:921-933 — defaults 100 and 1e-5: what the reference's docstring states (``sample.py:926``; the crate's own
        default cannot be read here).  Measured against 2.0 (round 4's default) on the reference's window schedule, profiles/
        r5_low_rank_schedule.txt: radon 0 divergences and 14.7 leapfrogs per draw against 5 and 15.0, the D = 60 demo 8.5 leapfrogs per
        draw against 26.9, D = 500 15.1 against 24.6 at 0.12 instead of 0.09 relative error of the posterior sd — fewer, surer columns."""
        s = PyNutsSettings.Diag(seed)
        object.__setattr__(s, "_adaptation", "low_rank")
        object.__setattr__(s, "_low_rank", {"mass_matrix_eigval_cutoff": 100.0, "mass_matrix_gamma": 1e-5})
        s.mass_matrix_update_freq = 10   # the low-rank default (recalled); a value the user sets afterwards — 1 included — is the schedule's
        return s

    @staticmethod
    def Flow(seed=None):
        raise NotImplementedError("adaptation='flow' is outside the scope of the HIP engine (diag / draw_diag only)")

    def __del__(self):
        try:
            if self._h:
                lib().nphip_settings_free(self._h)
        except Exception:
            pass

    def clone(self):
        return PyNutsSettings(C.c_void_p(lib().nphip_settings_clone(self._h)), self._adaptation, self._low_rank)

    def _apply_update(self, name: str, value):
        L = lib()
        key = name.encode()
        if self._adaptation == "low_rank" and name in ("mass_matrix_eigval_cutoff", "mass_matrix_gamma"):
            if value is None:
                return
            v = float(value)
            if name == "mass_matrix_eigval_cutoff" and not v > 1.0:
                raise ValueError("mass_matrix_eigval_cutoff must be greater than one")
            if name == "mass_matrix_gamma" and not v > 0.0:
                raise ValueError("mass_matrix_gamma must be positive")
            self._low_rank[name] = v
            return
        if name == "step_size_adapt_method":
            if not isinstance(value, str):
                raise ValueError("step_size_adapt_method must be a string")
            return _check_setting(L.nphip_settings_set_str(self._h, key, value.encode()))
        if name in _OPT_F64_KEYS:
            if value is None:
                if name == "step_size_jitter":
                    value = 0.0
                else:
                    return
            return _check_setting(L.nphip_settings_set_f64(self._h, key, float(value)))
        if name in _BOOL_KEYS:
            if not isinstance(value, (bool, np.bool_)):
                raise TypeError(f"'{type(value).__name__}' object cannot be converted to 'bool'")
            return _check_setting(L.nphip_settings_set_bool(self._h, key, int(bool(value))))
        if name in _U64_KEYS:
            if isinstance(value, (bool, np.bool_)) or not isinstance(value, (int, np.integer)) or int(value) < 0:
                raise TypeError(f"can't convert {value!r} to a non-negative integer for {name}")
            return _check_setting(L.nphip_settings_set_u64(self._h, key, int(value)))
        return _check_setting(L.nphip_settings_set_f64(self._h, key, float(value)))

    def update(self, kwargs=None, **kw):
        """``settings.update(dict)`` as in wrapper.rs:739-745."""
        items = dict(kwargs or {})
        items.update(kw)
        for k, v in items.items():
            self._apply_update(k, v)

    def __setattr__(self, name, value):
        self._apply_update(name, value)

    def __getattr__(self, name):
        if name in ("mass_matrix_eigval_cutoff", "mass_matrix_gamma") and object.__getattribute__(self, "_adaptation") == "low_rank":
            return object.__getattribute__(self, "_low_rank")[name]
        L = lib()
        out = C.c_uint64()
        if L.nphip_settings_get_u64(self._h, name.encode(), C.byref(out)) == NPHIP_OK:
            return bool(out.value) if name in _BOOL_KEYS else out.value
        outf = C.c_double()
        if L.nphip_settings_get_f64(self._h, name.encode(), C.byref(outf)) == NPHIP_OK:
            return outf.value
        raise AttributeError(_err())

    def set_pause_draws(self, draws):
        """Host-driven adaptation hook: chains stop (``PySampler.waiting``) after exactly these numbers of finished draws."""
        d = np.ascontiguousarray(sorted(int(x) for x in draws), dtype=np.uint64)
        _check_setting(lib().nphip_settings_set_pause_draws(self._h, C.c_uint64(len(d)), d.ctypes.data_as(C.c_void_p)))

    def as_dict(self):
        """``{"sampler", "adaptation", "settings": nested}`` as wrapper.rs:755-769."""
        L = lib()
        need = L.nphip_settings_to_json(self._h, None, 0)
        buf = C.create_string_buffer(int(need))
        L.nphip_settings_to_json(self._h, buf, need)
        nested = json.loads(buf.value.decode())
        if self._adaptation == "low_rank":
            nested["adapt_options"]["mass_matrix_options"] = {
                "store_mass_matrix": nested["adapt_options"]["mass_matrix_options"]["store_mass_matrix"],
                "gamma": self._low_rank["mass_matrix_gamma"], "eigval_cutoff": self._low_rank["mass_matrix_eigval_cutoff"]}
        return {"sampler": "nuts", "adaptation": self._adaptation, "settings": nested}


class PyMclmcSettings:
    @staticmethod
    def _no(*_a, **_k):
        raise NotImplementedError("sampler='mclmc' is outside the scope of the HIP engine")

    Diag = LowRank = Flow = _no


# --------------------------------------------------------------------------- models
class _Model:
    def __init__(self, handle, dim, keep=None):
        if not handle:
            raise RuntimeError(_err())
        self._h = C.c_void_p(handle)
        self.dim = int(dim)
        self._keep = keep

    def set_init(self, kind: str, points=None):
        code = {"uniform": 0, "normal": 1, "explicit": 2}[kind]
        if code == 2:
            pts = np.ascontiguousarray(points, dtype=np.float64)
            if pts.ndim != 2 or pts.shape[1] != self.dim:
                raise ValueError("Initial point has incorrect length")  # src/pyfunc.rs:561-563
            rc = lib().nphip_model_set_init(self._h, 2, pts.ctypes.data_as(C.c_void_p), C.c_uint64(pts.shape[0]))
        else:
            rc = lib().nphip_model_set_init(self._h, code, None, C.c_uint64(0))
        if rc != NPHIP_OK:
            raise ValueError(_err())

    def set_expand(self, expanded_dim: int, fn, user_data=0, keep_alive=None):
        """``ExpandFunc(dim, expanded_dim, ptr, user_data_ptr, keep_alive)`` of the reference (src/pymc.rs:74-95).

        ``fn``: address of a raw C expand callback ``int f(dim, expanded_dim, x, out, user_data)`` (src/pymc.rs:31-37, e.g. a
        numba cfunc address), a ctypes function pointer, or a Python callable ``x[dim] -> flat[expanded_dim]`` (wrapped)."""
        self._keep = list(self._keep or []) if isinstance(self._keep, (list, tuple)) else [self._keep]
        self._keep.append(keep_alive)
        if isinstance(fn, int):
            addr = C.c_void_p(fn)
        elif isinstance(fn, C._CFuncPtr):
            self._keep.append(fn)
            addr = C.cast(fn, C.c_void_p)
        else:
            pyfn, E, model = fn, int(expanded_dim), self

            def _cb(d, e, x, out, _u):
                if e != E:
                    return -1
                try:
                    np.ctypeslib.as_array(out, shape=(E,))[:] = np.asarray(pyfn(np.ctypeslib.as_array(x, shape=(d,)).copy()), dtype=np.float64).reshape(E)
                except Exception as exc:  # noqa: BLE001 - must not unwind through C (numba side: compile_pymc.py:1037-1039)
                    model.expand_exception = exc
                    return -2
                return 0

            cb = RAW_EXPAND_FN(_cb)
            self._keep.append(cb)
            addr = C.cast(cb, C.c_void_p)
        self.expand_exception = None
        if lib().nphip_model_set_expand(self._h, C.c_uint64(int(expanded_dim)), addr, C.c_void_p(user_data)) != NPHIP_OK:
            raise ValueError(_err())

    def set_device_expand(self, expanded_dim: int, fn_addr: int, user_data: int = 0, keep_alive=None):
        """Batched device expand (``nphip_device_expand_fn``): ``x[n][dim] -> out[n][expanded_dim]`` on the engine's stream."""
        self._keep = list(self._keep or []) if isinstance(self._keep, (list, tuple)) else [self._keep]
        self._keep.append(keep_alive)
        if lib().nphip_model_set_device_expand(self._h, C.c_uint64(int(expanded_dim)), C.c_void_p(fn_addr), C.c_void_p(user_data)) != NPHIP_OK:
            raise ValueError(_err())

    @property
    def expanded_dim(self):
        return int(lib().nphip_model_expanded_dim(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().nphip_model_free(self._h)
        except Exception:
            pass


class TridiagGaussianModel(_Model):
    """Fused analytic model: ``logp(x) = -1/2 (x-mu)' L (x-mu)``, L tridiagonal."""

    def __init__(self, diag, offdiag=None, mu=None):
        diag = np.ascontiguousarray(diag, dtype=np.float64)
        dim = diag.shape[0]
        off = None if offdiag is None or dim < 2 else np.ascontiguousarray(offdiag, dtype=np.float64)
        m = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        if off is not None and off.shape != (dim - 1,):
            raise ValueError("offdiag must have length dim-1")
        if m is not None and m.shape != (dim,):
            raise ValueError("mu must have length dim")
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
        super().__init__(lib().nphip_model_tridiag_gaussian(C.c_uint64(dim), p(m), p(diag), p(off)), dim)


class DenseGaussianModel(_Model):
    """Dense-precision Gaussian ``logp(x) = -1/2 (x-mu)' P (x-mu)`` evaluated by the engine's own fp64 MFMA GEMM
    (``nphip_model_dense_gaussian``): BASELINE.json configs[1] read as a dense correlated Gaussian."""

    def __init__(self, precision, mu=None):
        P = np.ascontiguousarray(precision, dtype=np.float64)
        if P.ndim != 2 or P.shape[0] != P.shape[1]:
            raise ValueError("precision must be a square matrix")
        dim = P.shape[0]
        m = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        if m is not None and m.shape != (dim,):
            raise ValueError("mu must have length dim")
        h = lib().nphip_model_dense_gaussian(C.c_uint64(dim), None if m is None else m.ctypes.data_as(C.c_void_p), P.ctypes.data_as(C.c_void_p))
        if not h:
            raise ValueError(_err())
        super().__init__(h, dim)


def mfma_f64_rate(device=0):
    """Measured fp64 matrix-core rate of the device in TFLOP/s (``nphip_test_mfma_f64_rate``)."""
    v = C.c_double(0.0)
    lib().nphip_test_mfma_f64_rate.argtypes = [C.c_int, C.POINTER(C.c_double)]
    if lib().nphip_test_mfma_f64_rate(int(device), C.byref(v)) != NPHIP_OK:
        raise RuntimeError(_err())
    return float(v.value)


def test_dense_grad(x, precision, mu=None, waves=1, device=0):
    """Test hook: the engine's dense-Gaussian evaluation (gradient GEMM + log-density) of the rows of ``x`` on the device."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = np.ascontiguousarray(precision, dtype=np.float64)
    n, dim = x.shape
    m = np.zeros(dim) if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
    g, lp = np.empty_like(x), np.empty(n)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    if lib().nphip_test_dense_grad(int(device), int(waves), C.c_uint64(n), C.c_uint64(dim), p(x), p(m), p(P), p(g), p(lp)) != NPHIP_OK:
        raise RuntimeError(_err())
    return g, lp


class HostCallbackModel(_Model):
    """Raw C logp callback with the reference's exact signature (src/pymc.rs:23-29).

    ``fn`` may be an integer address (e.g. ``numba.cfunc(...).address``), a ctypes function
    pointer, or a Python callable ``f(x) -> (logp, grad)`` (wrapped; slow, test use only)."""

    def __init__(self, dim, fn, user_data=0, n_threads=0, keep_alive=None):
        keep = [keep_alive]
        if isinstance(fn, int):
            addr = C.c_void_p(fn)
        elif isinstance(fn, C._CFuncPtr):
            keep.append(fn)
            addr = C.cast(fn, C.c_void_p)
        else:
            pyfn = fn
            self.exception = None

            def _cb(d, x, g, lp, _u):
                xs = np.ctypeslib.as_array(x, shape=(d,))
                try:
                    val, grad = pyfn(xs.copy())
                except Exception as e:  # recoverable iff flagged, as src/pyfunc.rs:100-116
                    if getattr(e, "is_recoverable", False):
                        return 1
                    self.exception = e  # re-raised (chained) by PySampler.wait, like DeviceCallbackModel
                    return -1
                np.ctypeslib.as_array(g, shape=(d,))[:] = grad
                lp[0] = val
                return 0

            cb = RAW_LOGP_FN(_cb)
            keep.append(cb)
            addr = C.cast(cb, C.c_void_p)
            n_threads = 1
            self._python_callable = True
        super().__init__(lib().nphip_model_host_callback(C.c_uint64(dim), addr, C.c_void_p(user_data), int(n_threads)), dim, keep)


class BridgeStanModel(_Model):
    """BridgeStan model handle evaluated on the host pool (src/stan.rs:454-463).

    ``stanlib`` is the loaded model library (``ctypes.CDLL``), ``bs_model`` its ``bs_model*``."""

    def __init__(self, dim, stanlib, bs_model, n_threads=0, keep_alive=None):
        ldg = C.cast(stanlib.bs_log_density_gradient, C.c_void_p)
        free = C.cast(stanlib.bs_free_error_msg, C.c_void_p) if hasattr(stanlib, "bs_free_error_msg") else None
        ptr = bs_model if isinstance(bs_model, C.c_void_p) else C.cast(bs_model, C.c_void_p)
        super().__init__(lib().nphip_model_bridgestan(C.c_uint64(dim), ptr, ldg, free, int(n_threads)), dim, [stanlib, bs_model, keep_alive])
        self._bs = (stanlib, ptr, free)

    def set_bridgestan_expand(self, expanded_dim: int, perm=None):
        """The expand step behind the C-ABI (reference src/stan.rs:473-520): ``bs_param_constrain(include_tp, include_gq)`` per
        stored draw with one ``bs_rng`` per chain, chains concurrently on the host pool; ``perm`` (``stan_names.c_order_permutation``)
        re-orders BridgeStan's column-major blocks to C order natively."""
        stanlib, ptr, free = self._bs
        fn = lambda name: C.cast(getattr(stanlib, name), C.c_void_p)  # noqa: E731
        p = None
        if perm is not None:
            p = np.ascontiguousarray(perm, dtype=np.uint64)
            if p.shape != (int(expanded_dim),):
                raise ValueError("perm must have expanded_dim entries")
        rc = lib().nphip_model_set_bridgestan_expand(self._h, C.c_uint64(int(expanded_dim)), ptr, fn("bs_param_constrain"), fn("bs_rng_construct"),
                                                    fn("bs_rng_destruct"), free, None if p is None else p.ctypes.data_as(C.c_void_p))
        if rc != NPHIP_OK:
            raise ValueError(_err())


class DeviceCallbackModel(_Model):
    """Batched device callback: ``fn(n_chains, dim, q_ptr, grad_ptr, logp_ptr, stream_ptr) -> int``."""

    def __init__(self, dim, fn):
        def _cb(n, d, q, g, lp, stream, _u):
            try:
                rc = fn(int(n), int(d), q, g, lp, stream)
                return 0 if rc is None else int(rc)
            except BaseException as e:  # noqa: BLE001 - must not unwind through C
                self.exception = e
                return -1

        self.exception = None
        cb = DEVICE_LOGP_FN(_cb)
        super().__init__(lib().nphip_model_device_callback(C.c_uint64(dim), C.cast(cb, C.c_void_p), None), dim, [cb, fn])


class NativeDeviceCallbackModel(_Model):
    """Batched device callback given as a raw C function pointer (``nphip_device_logp_fn``) plus its ``user_data``:
    a model compiled to native code (e.g. a HIP kernel launcher) — no Python on the per-leapfrog path."""

    def __init__(self, dim, fn_addr: int, user_data: int = 0, keep_alive=None):
        super().__init__(lib().nphip_model_device_callback(C.c_uint64(dim), C.c_void_p(fn_addr), C.c_void_p(user_data)), dim, [keep_alive])
        self.exception = None


class JitDensityModel(_Model):
    """A device density compiled at run time into its own resident kernel (``nphip_model_jit_density``; nutpie_amd/density.py):
    ``launch_addr`` / ``nv`` come from the model's library, ``data_ptr`` is its data block in device memory."""

    def __init__(self, dim, launch_addr: int, nv: int, data_ptr: int, lds_bytes_per_chain: int = 0, lds_bytes_shared: int = 0, keep_alive=None, waves_per_chain: int = 1,
                 low_rank: bool = False):
        super().__init__(lib().nphip_model_jit_density(C.c_uint64(dim), C.c_void_p(launch_addr), int(nv), C.c_void_p(data_ptr), C.c_uint64(int(lds_bytes_per_chain)),
                                                      C.c_uint64(int(lds_bytes_shared)), int(waves_per_chain)), dim, [keep_alive])
        self.exception = None
        if low_rank and lib().nphip_model_jit_low_rank(self._h, 1) != NPHIP_OK:   # (the library's resident kernel was built with -DNPHIP_JIT_LR=1)
            raise ValueError(_err())


# --------------------------------------------------------------------------- progress / trace
class PyChainProgress:
    """Field set of ``ChainProgress`` (wrapper.rs:47-104)."""

    __slots__ = ("finished_draws", "total_draws", "divergences", "tuning", "started", "latest_num_steps",
                 "total_num_steps", "step_size", "runtime_ms", "divergent_draws")

    def __init__(self, p: _Progress, divergent_draws):
        self.finished_draws = int(p.finished_draws)
        self.total_draws = int(p.total_draws)
        self.divergences = int(p.divergences)
        self.tuning = bool(p.tuning)
        self.started = bool(p.started)
        self.latest_num_steps = int(p.latest_num_steps)
        self.total_num_steps = int(p.total_num_steps)
        self.step_size = float(p.step_size)
        self.runtime_ms = int(p.runtime_ms)
        self.divergent_draws = list(divergent_draws)

    @property
    def num_steps(self):  # backward-compatible alias, wrapper.rs:78-82
        return self.latest_num_steps


_STAT_DTYPES = {
    "depth": np.int64, "n_steps": np.int64, "index_in_trajectory": np.int64,
    "diverging": np.bool_, "maxdepth_reached": np.bool_, "tuning": np.bool_,
    "energy": np.float64, "energy_error": np.float64, "logp": np.float64, "step_size": np.float64,
    "step_size_bar": np.float64, "mean_tree_accept": np.float64, "mean_tree_accept_sym": np.float64,
}
_VECTOR_STATS = ("gradient", "mass_matrix_inv", "divergence_start", "divergence_end", "divergence_momentum", "divergence_start_gradient")


class PyTrace:
    """Dense trace: ``draws[chain, draw, dim]`` + ``stats[name][chain, draw(, dim)]`` + ``finished[chain]``."""

    def __init__(self, draws, stats, finished, chain_offset=0, expanded=None):
        self.draws = draws
        self.stats = stats
        self.finished = finished
        self.chain_offset = chain_offset
        self.expanded = expanded  # name -> [chain, draw, *shape] when the model expanded the draws on the device

    def is_arrow(self):
        return False

    def is_zarr(self):
        return False

    def is_dense(self):
        return True


class PySampler:
    """Sampler handle (wrapper.rs:953-1457)."""

    def __init__(self, settings: PyNutsSettings, model: _Model, *, device=0, waves_per_chain=0, chain_offset=0,
                 n_local_chains=0, stream=None, store_draws=True, evals_per_launch=0, start_paused=False, manual=False,
                 staging=None, no_register_kernel=False, no_stream_cache=False, graph_steps=0, host_groups=0, host_persist=None):
        L = lib()
        la = _Launch()
        L.nphip_launch_defaults(C.byref(la))
        la.device = int(device)
        la.waves_per_chain = int(waves_per_chain)
        la.chain_offset = int(chain_offset)
        la.n_local_chains = int(n_local_chains)
        la.stream = C.c_void_p(stream) if stream else None
        la.store_draws = int(bool(store_draws))
        la.evals_per_launch = int(evals_per_launch)
        la.start_paused = int(bool(start_paused))
        la.manual = int(bool(manual))
        la.no_register_kernel = int(bool(no_register_kernel))
        la.no_stream_cache = int(bool(no_stream_cache))
        la.graph_steps = int(graph_steps)
        la.host_groups = int(host_groups)
        if host_persist is None:
            # resident launches (the kernel waits on the device for the host's evaluation) are for compiled callbacks; a Python
            # callable may itself submit work to this GPU, which could queue behind the waiting kernel — one launch per evaluation
            host_persist = 1 if getattr(model, "_python_callable", False) else 0
        la.host_persist = int(host_persist)
        if staging is not None:
            la.staging_q, la.staging_grad, la.staging_logp = (C.c_void_p(int(p)) for p in staging)
        self._model = model
        self._settings = settings
        self._chain_offset = int(chain_offset)
        self._store_draws = bool(store_draws)
        h = L.nphip_sampler_create(settings._h, model._h, C.byref(la))
        if not h:
            raise RuntimeError(_err())
        self._h = C.c_void_p(h)
        self._results = None

    # constructors named as the reference's (wrapper.rs:1189-1250)
    @classmethod
    def from_pyfunc(cls, settings, cores, model, progress_type=None, extra_callback=None, extra_callback_rate=None, store=None, **kw):
        # (the reference's constructors, wrapper.rs:1189-1250; progress reporting is done one layer up — nutpie_amd.sample's
        #  background sampler polls progress() and calls the callback — and the engine has no storage back-ends: neither is
        #  dropped silently)
        if store is not None:
            raise NotImplementedError("storage back-ends (zarr / arrow) are outside the scope of the HIP engine: the trace lives in HBM")
        if progress_type is not None or extra_callback is not None:
            raise NotImplementedError("progress templates / callbacks are handled by nutpie_amd.sample(), not by the sampler handle")
        return cls(settings, model, **kw)

    from_pymc = from_pyfunc
    from_stan = from_pyfunc

    def _require(self):
        if self._h is None:
            raise RuntimeError("Sampler is empty (results were taken)")

    @property
    def num_chains(self):
        return int(lib().nphip_sampler_num_chains(self._h))

    @property
    def dim(self):
        return int(lib().nphip_sampler_dim(self._h))

    @property
    def total_draws(self):
        return int(lib().nphip_sampler_total_draws(self._h))

    @property
    def waves_per_chain(self):
        return int(lib().nphip_sampler_waves_per_chain(self._h))

    @property
    def seconds(self):
        return float(lib().nphip_sampler_seconds(self._h))

    @property
    def launches(self):
        return int(lib().nphip_sampler_launches(self._h))

    @property
    def host_mode(self):
        """How a host-callback model is being driven: 'none' (not one), 'launch-per-evaluation', 'groups' (the same, groups of
        chains pipelined), 'resident' (the kernel waits on the device for the evaluations), 'fell-back' (resident launches that
        went back to a launch per evaluation because the device could not hold all chains at once)."""
        self._require()
        return ("none", "launch-per-evaluation", "groups", "resident", "fell-back")[int(lib().nphip_sampler_host_mode(self._h))]

    def wait(self, timeout_seconds=None):
        """Blocks (GIL released by ctypes) — raises ``TimeoutError`` and leaves the sampler running
        on timeout (wrapper.rs:1112-1117, 1305-1330)."""
        self._require()
        ms = -1 if timeout_seconds is None else max(0, int(timeout_seconds * 1000))
        # poll in 100 ms slices so KeyboardInterrupt is seen (wrapper.rs:1139-1141)
        remaining = ms
        while True:
            step = 100 if ms < 0 else min(100, remaining)
            rc = lib().nphip_sampler_wait(self._h, step)
            if rc == WAIT_DONE:
                return
            if rc == WAIT_ERROR:
                exc = getattr(self._model, "exception", None)
                if exc is not None:
                    raise RuntimeError(f"logp callback raised: {exc!r}") from exc
                raise RuntimeError(_err())
            if ms >= 0:
                remaining -= step
                if remaining <= 0:
                    raise TimeoutError("Timeout while waiting for sampler to finish")

    def step(self, n_launches=1):
        """Manual mode: run ``n_launches`` engine iterations on this thread.
        Returns ``(done, launches_performed, kernel_ms)`` — kernel_ms from HIP events on the engine stream."""
        self._require()
        ms = C.c_double(0.0)
        cnt = C.c_uint64(0)
        rc = lib().nphip_sampler_step(self._h, C.c_uint64(int(n_launches)), C.byref(ms), C.byref(cnt))
        if rc == WAIT_ERROR:
            exc = getattr(self._model, "exception", None)
            if exc is not None:
                raise RuntimeError(f"logp callback raised: {exc!r}") from exc
            raise RuntimeError(_err())
        return rc == WAIT_DONE, int(cnt.value), float(ms.value)

    def pause(self):
        self._require()
        lib().nphip_sampler_pause(self._h)

    def resume(self):
        self._require()
        lib().nphip_sampler_resume(self._h)

    def abort(self):
        self._require()
        lib().nphip_sampler_abort(self._h)

    def is_finished(self):
        self._require()
        return bool(lib().nphip_sampler_is_finished(self._h))

    def is_empty(self, ignore_error=False):
        return self._h is None

    def progress(self):
        """list[PyChainProgress], one per local chain."""
        self._require()
        n = self.num_chains
        arr = (_Progress * n)()
        if lib().nphip_sampler_progress(self._h, C.c_uint64(2**64 - 1), arr) != NPHIP_OK:
            raise RuntimeError(_err())
        div = self._copy("diverging", np.bool_)
        fin = [int(p.finished_draws) for p in arr]
        return [PyChainProgress(arr[i], np.nonzero(div[i, : fin[i]])[0].tolist()) for i in range(n)]

    def _copy(self, name, dtype, vec=False):
        n, T, d = self.num_chains, self.total_draws, self.dim
        shape = (n, T, d) if vec else (n, T)
        out = np.empty(shape, dtype=dtype)
        rc = lib().nphip_sampler_copy_stat(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), C.c_uint64(out.nbytes))
        if rc != NPHIP_OK:
            raise RuntimeError(_err())
        return out

    def device_ptr(self, name):
        return lib().nphip_sampler_device_ptr(self._h, name.encode())

    def waiting_codes(self):
        """Per local chain: 0 running, 1 stopped at a pause draw (``PyNutsSettings.set_pause_draws``), 2 finished / failed."""
        self._require()
        mask = np.zeros(self.num_chains, dtype=np.uint8)
        if lib().nphip_sampler_waiting(self._h, mask.ctypes.data_as(C.c_void_p)) < 0:
            raise RuntimeError(_err())
        return mask

    def waiting(self):
        return self.waiting_codes() == 1

    def resume_at(self, chains, positions):
        """Resume waiting chains at new positions: ``positions`` is a numpy array ``[n, dim]`` or a CUDA tensor."""
        self._require()
        ch = np.ascontiguousarray(chains, dtype=np.uint64)
        if hasattr(positions, "data_ptr"):
            pos = positions.contiguous()
            assert tuple(pos.shape) == (len(ch), self.dim) and str(pos.dtype) == "torch.float64"
            rc = lib().nphip_sampler_resume_at(self._h, C.c_uint64(len(ch)), ch.ctypes.data_as(C.c_void_p), C.c_void_p(pos.data_ptr()), 1)
        else:
            pos = np.ascontiguousarray(positions, dtype=np.float64)
            assert pos.shape == (len(ch), self.dim)
            rc = lib().nphip_sampler_resume_at(self._h, C.c_uint64(len(ch)), ch.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), 0)
        if rc != NPHIP_OK:
            raise RuntimeError(_err())

    def release(self, chains):
        """Chains stopped at a pause draw go on as if they had not stopped (``nphip_sampler_release``)."""
        self._require()
        ch = np.ascontiguousarray(chains, dtype=np.uint64)
        if lib().nphip_sampler_release(self._h, C.c_uint64(len(ch)), ch.ctypes.data_as(C.c_void_p)) != NPHIP_OK:
            raise RuntimeError(_err())

    def set_evals_per_launch(self, evals: int):
        """Manual mode: evaluations per chain of the launches that follow (0: the default)."""
        self._require()
        if lib().nphip_sampler_set_evals_per_launch(self._h, C.c_int32(int(evals))) != NPHIP_OK:
            raise RuntimeError(_err())

    def chain_draws(self):
        """(draws finished per local chain, state per chain: 0 running / 1 stopped at a pause draw / 2 finished or failed)."""
        self._require()
        draws = np.zeros(self.num_chains, dtype=np.int64)
        state = np.zeros(self.num_chains, dtype=np.uint8)
        if lib().nphip_sampler_chain_draws(self._h, draws.ctypes.data_as(C.c_void_p), state.ctypes.data_as(C.c_void_p)) < 0:
            raise RuntimeError(_err())
        return draws, state

    def stage_metric(self, chains, sig2, V=None, lam=None):
        """The hand-in of :meth:`set_metric` for chains that RUN (``nphip_sampler_stage_metric``): each chain takes the metric itself
        at the end of the draw it is working on.  Returns the number of chains that parked it (chains past their warm-up do not)."""
        return self.set_metric(chains, sig2, V, lam, _staged=True)

    def set_metric(self, chains, sig2, V=None, lam=None, _staged=False):
        """A new metric ``M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2`` for chains stopped at a pause draw (settings
        ``low_rank_metric``): ``sig2[n, dim]``, ``V[n, k, dim]`` (row j = column j of V), ``lam[n, k]`` — numpy arrays or CUDA
        tensors.  The chains keep their positions, re-run the step-size search and go on."""
        self._require()
        ch = np.ascontiguousarray(chains, dtype=np.uint64)
        n, d = len(ch), self.dim
        on_device = hasattr(sig2, "data_ptr")
        if on_device:
            arrs = [sig2.contiguous(), None if V is None else V.contiguous(), None if lam is None else lam.contiguous()]
            k = 0 if V is None else int(arrs[1].shape[1])
            assert tuple(arrs[0].shape) == (n, d) and all(str(a.dtype) == "torch.float64" for a in arrs if a is not None)
            ptrs = [C.c_void_p(a.data_ptr()) if a is not None else None for a in arrs]
        else:
            arrs = [np.ascontiguousarray(sig2, dtype=np.float64), None if V is None else np.ascontiguousarray(V, dtype=np.float64),
                    None if lam is None else np.ascontiguousarray(lam, dtype=np.float64)]
            k = 0 if V is None else int(arrs[1].shape[1])
            assert arrs[0].shape == (n, d)
            ptrs = [a.ctypes.data_as(C.c_void_p) if a is not None else None for a in arrs]
        if k:
            assert tuple(arrs[1].shape) == (n, k, d) and tuple(arrs[2].shape) == (n, k)
        if _staged:
            taken = C.c_uint64(0)
            rc = lib().nphip_sampler_stage_metric(self._h, C.c_uint64(n), ch.ctypes.data_as(C.c_void_p), C.c_uint64(k), ptrs[0], ptrs[1], ptrs[2], int(on_device), C.byref(taken))
            if rc != NPHIP_OK:
                raise RuntimeError(_err())
            return int(taken.value)
        rc = lib().nphip_sampler_set_metric(self._h, C.c_uint64(n), ch.ctypes.data_as(C.c_void_p), C.c_uint64(k), ptrs[0], ptrs[1], ptrs[2], int(on_device))
        if rc != NPHIP_OK:
            raise RuntimeError(_err())

    def expanded(self):
        """The model's expand function over the stored trace: ``[chain, draw, expanded_dim]`` (NaN rows for unfinished draws);
        host rows on the thread pool or one batched device call per block (src/pymc.rs:217-286, per draw in the reference)."""
        self._require()
        E = self._model.expanded_dim
        if E == 0:
            raise RuntimeError("the model has no expand function")
        out = np.empty((self.num_chains, self.total_draws, E), dtype=np.float64)
        if lib().nphip_sampler_copy_expanded(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint64(out.nbytes)) != NPHIP_OK:
            exc = getattr(self._model, "expand_exception", None)
            if exc is not None:
                raise RuntimeError(f"expand function raised: {exc!r}") from exc
            raise RuntimeError(_err())
        return out

    def _snapshot(self):
        n = self.num_chains
        fin = np.zeros(n, dtype=np.uint64)
        if lib().nphip_sampler_finished_draws(self._h, fin.ctypes.data_as(C.c_void_p)) != NPHIP_OK:
            raise RuntimeError(_err())
        stats = {k: self._copy(k, dt) for k, dt in _STAT_DTYPES.items()}
        for k in _VECTOR_STATS:
            if self.device_ptr(k):
                stats[k] = self._copy(k, np.float64, vec=True)
        expanded = None
        expand = getattr(self, "_device_expand", None)
        if expand is not None and self._store_draws:
            # expand step batched on the device, straight from the engine's draws buffer
            expanded = expand(self)
        elif self._store_draws and self._model.expanded_dim:
            expanded = {"__flat__": self.expanded()}   # behind the C-ABI; split into variables by the model's _expand_draws
        need_host_draws = self._store_draws and (expanded is None or getattr(self, "_keep_host_draws", True))
        draws = self._copy("draws", np.float64, vec=True) if need_host_draws else None
        return PyTrace(draws, stats, fin.astype(np.int64), self._chain_offset, expanded)

    def inspect(self):
        """Copy of the current state of the trace (wrapper.rs:1401-1429)."""
        self._require()
        return self._snapshot()

    def take_results(self):
        """Transfers the trace and empties the sampler (wrapper.rs:1431-1456)."""
        self._require()
        res = self._snapshot()
        self.close()
        return res

    def close(self):
        if self._h is not None:
            lib().nphip_sampler_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def batched_eigh(A, _stage=0):
    """``nphip_batched_eigh`` on a CUDA float64 tensor ``A[b, n, n]`` (symmetric, the lower triangle is read; n <= 128):
    ``(w[b, n] ascending, V[b, n, n])`` with ``A[i] @ V[i] == V[i] * w[i]`` — the shapes of ``torch.linalg.eigh``.  One workgroup per
    matrix on torch's current stream of A's device (nutpie_amd/csrc/linalg.hip)."""
    import torch

    if not (A.is_cuda and A.dtype == torch.float64 and A.dim() == 3 and A.shape[1] == A.shape[2]):
        raise ValueError("batched_eigh: a CUDA float64 tensor [batch, n, n]")
    b, n = int(A.shape[0]), int(A.shape[1])
    V = A.contiguous().clone()
    w = torch.empty(b, n, dtype=torch.float64, device=A.device)
    with torch.cuda.device(A.device):
        st = C.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)
        if _stage:   # (test hook: the stages on their own, include/nutpie_hip.h: nphip_test_eigh_stage)
            rc = lib().nphip_test_eigh_stage(C.c_uint64(b), C.c_uint64(n), C.c_void_p(V.data_ptr()), C.c_void_p(w.data_ptr()), st, int(_stage))
        else:
            rc = lib().nphip_batched_eigh(C.c_uint64(b), C.c_uint64(n), C.c_void_p(V.data_ptr()), C.c_void_p(w.data_ptr()), st)
    if rc != NPHIP_OK:
        raise RuntimeError(_err())
    return w, V


def low_rank_estimate_supported(dim: int, m: int, n_pick: int, k_max: int) -> bool:
    return bool(lib().nphip_low_rank_estimate_supported(C.c_uint64(int(dim)), C.c_uint64(int(m)), C.c_uint64(int(n_pick)), C.c_uint64(int(k_max))))


def low_rank_estimate(draws, grads, chains, lo, hi, pick, gamma, cutoff, k_max=16, max_workgroups=0):
    """``nphip_low_rank_estimate`` (nutpie_amd/csrc/lowrank_est.hip): the low-rank metric of ``chains`` (a CUDA int64 tensor of rows of
    ``draws``, or None for all rows) from the window ``draws[:, lo:hi]``, ``grads[:, lo:hi]`` — CUDA float64 tensors ``[n_all, T, D]``
    with contiguous rows, e.g. views of the engine's trace — and the basis draws ``pick`` (indices into the window).  Returns
    ``(sigma2[n, D], V[n, k_max, D], lam[n, k_max], k_used[n] int32)`` on torch's current stream, nothing synchronised."""
    import torch

    if not (draws.is_cuda and grads.is_cuda and draws.dtype == torch.float64 and grads.dtype == torch.float64 and draws.dim() == 3 and draws.shape == grads.shape):
        raise ValueError("low_rank_estimate: CUDA float64 tensors [chains, draws, dim] of one shape")
    if draws.stride(2) != 1 or grads.stride(2) != 1 or draws.stride() != grads.stride():
        raise ValueError("low_rank_estimate: rows must be contiguous and both arrays laid out alike")
    D, m = int(draws.shape[2]), int(hi) - int(lo)
    pick = np.ascontiguousarray(pick, dtype=np.int32)
    n = int(draws.shape[0]) if chains is None else int(chains.numel())
    if chains is not None and not (chains.is_cuda and chains.dtype == torch.int64 and chains.is_contiguous()):
        raise ValueError("low_rank_estimate: chains must be a contiguous CUDA int64 tensor")
    dev = draws.device
    sig2 = torch.empty(n, D, dtype=torch.float64, device=dev)
    V = torch.empty(n, int(k_max), D, dtype=torch.float64, device=dev)
    lam = torch.empty(n, int(k_max), dtype=torch.float64, device=dev)
    k_used = torch.empty(n, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(n, 1), 4112, dtype=torch.float64, device=dev)
    off = int(lo) * int(draws.stride(1)) * 8
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib().nphip_low_rank_estimate(C.c_uint64(n), C.c_uint64(D), C.c_uint64(m), C.c_uint64(len(pick)), pick.ctypes.data_as(C.c_void_p),
                                           C.c_void_p(draws.data_ptr() + off), C.c_void_p(grads.data_ptr() + off), C.c_int64(int(draws.stride(0))),
                                           C.c_int64(int(draws.stride(1))), None if chains is None else C.c_void_p(chains.data_ptr()), C.c_double(float(gamma)),
                                           C.c_double(float(cutoff)), C.c_uint64(int(k_max)), C.c_void_p(sig2.data_ptr()), C.c_void_p(V.data_ptr()),
                                           C.c_void_p(lam.data_ptr()), C.c_void_p(k_used.data_ptr()), C.c_void_p(scratch.data_ptr()), C.c_uint64(int(max_workgroups)), st)
    if rc != NPHIP_OK:
        raise RuntimeError("nphip_low_rank_estimate refused the shape (dim <= 512, at most 32 basis draws, k_max <= 16, cutoff > 1, gamma > 0)")
    low_rank_estimate.last_scratch = scratch   # (diagnostics: words 4096.. of every chain's row — scratch/r6_lr_native.py)
    return sig2, V, lam, k_used


def default_evals_per_launch(dim: int) -> int:
    """Leapfrogs per chain per kernel launch a fused model of this dimension runs by default."""
    return int(lib().nphip_default_evals_per_launch(C.c_uint64(int(dim))))


# --------------------------------------------------------------------------- test hooks
def test_rowpool(threads, rows, batches, use=0):
    """Host-only: run the evaluation pool of the host-callback path; returns (per-row sums, usable cores)."""
    out = np.zeros(int(rows), dtype=np.uint64)
    cores = C.c_int(0)
    if lib().nphip_test_rowpool(int(threads), C.c_uint64(int(rows)), int(batches), int(use), out.ctypes.data_as(C.c_void_p), C.byref(cores)) != NPHIP_OK:
        raise RuntimeError(_err())
    return out, int(cores.value)


def test_detmath(fn: str, x, device=0):
    code = {"exp": 0, "log": 1, "log1p": 2, "sin2pi": 3, "cos2pi": 4, "sqrt": 5, "recip": 6}[fn]
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    if lib().nphip_test_detmath(device, code, C.c_uint64(x.size), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)) != NPHIP_OK:
        raise RuntimeError(_err())
    return y


def test_normals(seed, chain, draw, purpose, n, device=0):
    x = np.array([seed, chain, draw, purpose], dtype=np.float64)
    y = np.empty(n)
    if lib().nphip_test_detmath(device, 7, C.c_uint64(n), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)) != NPHIP_OK:
        raise RuntimeError(_err())
    return y


def test_dot(x, y, waves=1, device=0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = C.c_double()
    if lib().nphip_test_dot(device, waves, C.c_uint64(x.size), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.byref(out)) != NPHIP_OK:
        raise RuntimeError(_err())
    return out.value
