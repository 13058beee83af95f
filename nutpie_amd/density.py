"""Device densities compiled at run time: a model as HIP source -> its own resident NUTS kernel.

The reference turns a PyMC model into ONE compiled log-density function (``compile_pymc.py:668-871``: logp + gradient joined
into a single vector function, numba- or JAX-compiled) plus the shared data it reads (``:239-269``), and swaps that data
with ``with_data`` without recompiling (``:140-166``).  The GPU form of that is this module: the model is a HIP device function

    __device__ double nphip_density(const NphipData& data, int dim, const double* x, double* grad, double* lds, const double* shared, int lane);

evaluated by one wavefront per chain — ``x[dim]`` the unconstrained position, ``grad[dim]`` receives the gradient, the return
value is the log-density (the same in every lane), ``lds`` is per-chain scratch, ``data`` the model's arrays and scalars.
Optionally (``lds_doubles_shared > 0``) the source also defines

    __device__ void nphip_density_stage(const NphipData& data, double* shared, int thread, int n_threads);

which all threads of a workgroup run once per launch to fill ``shared`` — LDS common to the workgroup's chains, typically the
model's data: a lone wave waits out every L2 access of its density, LDS is ten times closer.  It is
compiled with the ROCm compiler driver into the model's own instantiation of the engine's kernel (``kernels.hip`` part 7):
the register-resident leaf of the fused models with the evaluation as a CALL in its middle — no kernel launch and no memory
round trip of the chain state per gradient evaluation (``nphip_model_jit_density``).  The same library also exports the density
as a plain batched device callback (``nphip_jit_logp``), used where the resident kernel does not apply (``dim > 1024``,
``store_divergences``, ``adaptation="low_rank"``) — identical arithmetic, so identical draws.

``NphipData`` is generated from the ``data`` dict: a float64 array ``y`` becomes ``const double* y; int n_y;``, an integer
array ``const int* idx; int n_idx;``, a Python float / int a ``double`` / ``int`` field.  ``with_data(**updates)`` replaces
arrays and scalars (same names and kinds) and re-uses the compiled library.

Helpers available to the density source: everything in ``include/nphip_spec.h`` (``nphip_exp``, ``nphip_log`` ... — the engine's
reproducible elementary functions; ``exp`` / ``log`` of the HIP device library work too) and ``nphip_wave_sum(double)``, the sum
over the 64 lanes in the engine's fixed order (``nphip_wave_sum3`` / ``nphip_wave_sum4``: several sums in one pass), and ``NPHIP_LDS_PTR(type, p)`` /
``NPHIP_LDS_CPTR(type, p)`` to address ``lds`` / ``shared`` as LDS (``ds_read`` instead of flat loads).

hiprtc is not used: the contract header includes the C standard headers, which hiprtc's built-in include set lacks; the
compiler driver is the same one that builds the engine, with the same flags (``-ffp-contract=off``).
"""

from __future__ import annotations

import ctypes as C
import dataclasses
import hashlib
import os
import re
import struct
import subprocess
import tempfile
from dataclasses import dataclass
from typing import Any, Callable

import numpy as np

from nutpie_amd import _lib
from nutpie_amd.sample import CompiledModel

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INCLUDE = os.path.join(_HERE, "..", "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-function"]


def cache_dir() -> str:
    """Where compiled densities are kept: ``$NUTPIE_AMD_CACHE``, else ``nutpie_amd/_density_cache`` beside the engine's library
    (so that libraries built ahead of time travel with the tree), else ``~/.cache/nutpie_amd``."""
    d = os.environ.get("NUTPIE_AMD_CACHE")
    if not d:
        d = os.path.join(_HERE, "_density_cache")
        try:
            os.makedirs(d, exist_ok=True)
            if not os.access(d, os.W_OK):
                raise OSError
        except OSError:
            d = os.path.join(os.path.expanduser("~"), ".cache", "nutpie_amd")
    os.makedirs(d, exist_ok=True)
    return d


# --------------------------------------------------------------------------- the data block
def _kind(v):
    if isinstance(v, (bool, np.bool_)):
        raise TypeError("boolean data is not supported: pass 0 / 1 integers")
    if isinstance(v, (int, np.integer)):
        return "int"
    if isinstance(v, (float, np.floating)):
        return "double"
    a = np.asarray(v)
    if a.dtype.kind == "f":
        return "double*"
    if a.dtype.kind in "iu":
        return "int*"
    raise TypeError(f"unsupported data type {a.dtype}")


def data_layout(data: dict[str, Any]):
    """Field order of ``NphipData``: pointers, then doubles, then ints (no padding anywhere).  Returns [(name, kind)]."""
    kinds = {k: _kind(v) for k, v in data.items()}
    ptrs = [(k, kinds[k]) for k in data if kinds[k].endswith("*")]
    dbls = [(k, "double") for k in data if kinds[k] == "double"]
    ints = [("n_" + k, "int") for k, _ in ptrs] + [(k, "int") for k in data if kinds[k] == "int"]
    names = [n for n, _ in ptrs + dbls + ints]
    dup = next((n for n in names if names.count(n) > 1), None)
    if dup is not None:
        raise ValueError(f"data field {dup!r} is defined twice (an array `y` brings its length as `n_y`)")
    return ptrs + dbls + ints


def struct_source(layout) -> str:
    lines = ["struct NphipData {"]
    for name, kind in layout:
        lines.append(f"    const {kind[:-1]}* {name};" if kind.endswith("*") else f"    {kind} {name};")
    if not layout:
        lines.append("    int unused_;")
    lines.append("};")
    return "\n".join(lines)


_EXPAND_DEFINITION = re.compile(r"__device__\s+double\s+nphip_expand\s*\(")


def generated_source(user_source: str, layout) -> str:
    has_expand = bool(_EXPAND_DEFINITION.search(user_source))     # a DEFINITION, not a mention in a comment or a call
    return "\n".join([
        "// generated by nutpie_amd.density: one model's density compiled into the engine's resident kernel",
        "#include <hip/hip_runtime.h>",
        '#include "nphip_spec.h"',
        struct_source(layout),
        "__device__ double nphip_density(const NphipData& data, int dim, const double* x, double* grad, double* lds, const double* shared, int lane);",
        "__device__ void nphip_density_stage(const NphipData& data, double* shared, int thread, int n_threads);",
        "// the model also defines its expand step as a device function (nphip_expand: generated by nutpie_amd.symbolic)",
        "#define NPHIP_JIT_EXPAND 1" if has_expand else "",
        "__device__ double nphip_expand(const NphipData& data, int dim, const double* x, double* out, double* lds, const double* shared, int lane);" if has_expand else "",
        "// chains per workgroup of the running launch (one wave per chain): written by the kernels before their first barrier — LaunchSlice::cpb",
        "__shared__ int nphip_chains_per_block_;",
        '#include "kernels.hip"',
        "// the engine's wave reduction (sum over the 64 lanes in the contract's order; the same value in every lane)",
        "// `lds` and `shared` are LDS: through these casts the compiler emits ds_read / ds_write instead of flat accesses",
        "#define NPHIP_LDS_PTR(type, p) ((__attribute__((address_space(3))) type*)(p))",
        "#define NPHIP_LDS_CPTR(type, p) ((const __attribute__((address_space(3))) type*)(p))",
        "static __device__ __forceinline__ double nphip_wave_sum(double v) { return nphip::wave_sum(v); }",
        "// the same over ALL threads of the chain (waves_per_chain > 1: wave totals added in wave order through LDS; every thread gets the",
        "// same value), the number of threads that evaluate one chain's density, and the barrier between its phases",
        "#define NPHIP_CHAIN_THREADS (64 * NPHIP_JIT_W)",
        "// which of the launch's resident chains this is (one wave per chain: up to four per workgroup) — the index of its block of",
        "// data.scratch__ when the model asked for scratch in device memory (scratch_doubles_per_chain)",
        "#define NPHIP_CHAIN_SLOT (NPHIP_JIT_W == 1 ? (int)(blockIdx.x * nphip_chains_per_block_ + (threadIdx.x >> 6)) : (int)blockIdx.x)",
        "template <int N> static __device__ __forceinline__ void nphip_chain_sumN(double (&v)[N]) {",
        "    if (NPHIP_JIT_W == 1) {   // (the same bits whichever way the wave sums are taken: kernels.hip, wave_sumN_halving)",
        "        if constexpr (N == 2 || N == 4 || N == 8) nphip::wave_sumN_halving(v);",
        "        else if constexpr (N == 3) { double w[4] = {v[0], v[1], v[2], 0.0}; nphip::wave_sumN_halving(w); v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; }",
        "        else nphip::wave_sumN(v);",
        "        return;",
        "    }",
        "    __shared__ double red_[8 * NPHIP_JIT_W];",
        "    nphip::reduceN<NPHIP_JIT_W, N>(v, (NPHIP_LDS double*)red_);",
        "}",
        "static __device__ __forceinline__ double nphip_chain_sum(double a) { double v[1] = {a}; nphip_chain_sumN(v); return v[0]; }",
        "static __device__ __forceinline__ void nphip_chain_sum2(double& a, double& b) { double v[2] = {a, b}; nphip_chain_sumN(v); a = v[0]; b = v[1]; }",
        "static __device__ __forceinline__ void nphip_chain_sum3(double& a, double& b, double& c) { double v[3] = {a, b, c}; nphip_chain_sumN(v); a = v[0]; b = v[1]; c = v[2]; }",
        "static __device__ __forceinline__ void nphip_chain_sum4(double& a, double& b, double& c, double& d) { double v[4] = {a, b, c, d}; nphip_chain_sumN(v); a = v[0]; b = v[1]; c = v[2]; d = v[3]; }",
        "// the largest value over all threads of the chain (every thread gets it)",
        "static __device__ __forceinline__ double nphip_chain_max(double v) {",
        "    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));",
        "    if (NPHIP_JIT_W > 1) {",
        "        __shared__ double mx_[NPHIP_JIT_W];",
        "        __syncthreads();",
        "        if ((threadIdx.x & 63) == 0) mx_[threadIdx.x >> 6] = v;",
        "        __syncthreads();",
        "        v = mx_[0];",
        "        for (int w = 1; w < NPHIP_JIT_W; ++w) v = fmax(v, mx_[w]);",
        "    }",
        "    return v;",
        "}",
        "static __device__ __forceinline__ void nphip_chain_barrier() {",
        '    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");',
        "    if (NPHIP_JIT_W == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();",
        "}",
        "// several sums at once (the same bits as one nphip_wave_sum each, issued stage by stage: a lone wave otherwise waits out every step)",
        "static __device__ __forceinline__ void nphip_wave_sum3(double& a, double& b, double& c) { double v[3] = {a, b, c}; nphip::wave_sumN(v); a = v[0]; b = v[1]; c = v[2]; }",
        "static __device__ __forceinline__ void nphip_wave_sum4(double& a, double& b, double& c, double& d) { double v[4] = {a, b, c, d}; nphip::wave_sumN(v); a = v[0]; b = v[1]; c = v[2]; d = v[3]; }",
        "#line 1 \"density source\"",
        user_source,
        "" if "nphip_density_stage" in user_source else "__device__ void nphip_density_stage(const NphipData&, double*, int, int) {}",
        "",
    ])


def compile_density(user_source: str, layout, ndim: int, *, waves: int = 1, verbose: bool = False, low_rank: bool = False) -> str:
    """Build (or find in the cache) the model's library; returns its path.  ``waves`` wavefronts evaluate one chain's density.
    ``low_rank``: the library's resident kernel integrates under the low-rank metric (``adaptation="low_rank"``)."""
    if waves not in (1, 2, 4):
        raise ValueError("waves_per_chain must be 1, 2 or 4")
    nv = ((int(ndim) + 127) // 128 + waves - 1) // waves   # chunks of 128 dimensions per wave
    src = generated_source(user_source, layout)
    deps = [os.path.join(_CSRC, f) for f in ("kernels.hip", "engine_types.h", "dense_tile.h")] + [os.path.join(_INCLUDE, "nphip_spec.h")]
    h = hashlib.sha256()
    h.update(src.encode())
    for d in deps:
        h.update(open(d, "rb").read())
    # (no inter-procedural register allocation: the leaf with the model's density inlined is coupled to the register use of the out-of-line draw end
    # otherwise — config 3 same-box +0.5 % / +2.4 % / +5 % at 1 / 2 / 4 waves per chain, its low-rank kernels +3.4 % (the FUSED models' low-rank leaf
    # loses by it, csrc/Makefile): DESIGN.md §4, profiles/r6_call_placement_and_draw_end.txt)
    flags = (_FLAGS + ["-mllvm", "-enable-ipra=0", "-DNPHIP_JIT_DENSITY=1", "-DNPHIP_PART=7", f"-DNPHIP_JIT_NV={max(1, nv)}", f"-DNPHIP_JIT_W={waves}"]
             + (["-DNPHIP_JIT_LR=1"] if low_rank else []) + os.environ.get("NUTPIE_AMD_JIT_FLAGS", "").split())
    h.update(" ".join(flags).encode())
    out = os.path.join(cache_dir(), f"density_{h.hexdigest()[:24]}.so")
    if os.path.exists(out):
        return out
    # (the temporary directory lives INSIDE the cache directory: os.replace below must not cross a filesystem — /tmp often is a tmpfs)
    with tempfile.TemporaryDirectory(dir=cache_dir(), prefix=".build_") as tmp:
        path = os.path.join(tmp, "density.hip")
        with open(path, "w") as f:
            f.write(src)
        tmp_out = os.path.join(tmp, "density.so")
        r = subprocess.run([HIPCC, *flags, "-I", _CSRC, "-I", _INCLUDE, "-o", tmp_out, path], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compiling the density failed:\n" + r.stderr[-4000:])
        if verbose and r.stderr:
            print(r.stderr)
        os.replace(tmp_out, out)   # atomic: concurrent ranks may compile the same model
    return out


class _Batch(C.Structure):
    _fields_ = [("data", C.c_void_p), ("lds_doubles", C.c_int32), ("shared_doubles", C.c_int32)]


class DensityLibrary:
    """The loaded library of one compiled density."""

    def __init__(self, path: str):
        _lib.lib()   # the engine (and torch's HIP runtime) first: one runtime per process
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.nphip_jit_nv.restype = C.c_int
        self.nv = int(self.lib.nphip_jit_nv())
        self.lib.nphip_jit_w.restype = C.c_int
        self.waves = int(self.lib.nphip_jit_w())
        self.launch_addr = C.cast(self.lib.nphip_jit_launch, C.c_void_p).value
        self.logp_addr = C.cast(self.lib.nphip_jit_logp, C.c_void_p).value
        self.lib.nphip_jit_lr.restype = C.c_int
        self.low_rank = bool(self.lib.nphip_jit_lr())
        self.lib.nphip_jit_has_expand.restype = C.c_int
        self.expand_addr = C.cast(self.lib.nphip_jit_expand, C.c_void_p).value if int(self.lib.nphip_jit_has_expand()) else None


class DeviceData:
    """``NphipData`` in device memory: the arrays as torch tensors on the GPU, the block itself as a byte tensor."""

    def __init__(self, data: dict[str, Any], layout, device: int, device_arrays: dict[str, Any] | None = None):
        import torch

        dev = torch.device("cuda", device)
        self.tensors = {}
        device_arrays = device_arrays or {}
        blob = b""
        for name, kind in layout:
            if kind.endswith("*") and name in device_arrays:      # allocated on the device by the caller (scratch)
                t = device_arrays[name]
                self.tensors[name] = t
                blob += struct.pack("<Q", t.data_ptr())
            elif kind.endswith("*"):
                a = np.ascontiguousarray(data[name], dtype=np.float64 if kind == "double*" else np.int32)
                t = torch.as_tensor(a).to(dev) if a.size else torch.zeros(1, dtype=torch.float64 if kind == "double*" else torch.int32, device=dev)
                self.tensors[name] = t
                blob += struct.pack("<Q", t.data_ptr())
            elif kind == "double":
                blob += struct.pack("<d", float(data[name]))
            elif name in data:
                blob += struct.pack("<i", int(data[name]))
            elif name[2:] in device_arrays:
                blob += struct.pack("<i", int(min(device_arrays[name[2:]].numel(), 2**31 - 1)))
            else:   # n_<array>
                blob += struct.pack("<i", int(np.asarray(data[name[2:]]).size))
        blob += b"\0" * ((-len(blob)) % 8 + 8)
        self.block = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        torch.cuda.synchronize(dev)
        self.ptr = self.block.data_ptr()


@dataclass(frozen=True)
class JitteredInit:
    """PyMC's initial points (the reference's default for PyMC models: ``make_initial_point_fn(default_strategy="support_point",
    jitter_rvs=set(model.free_RVs))``, ``python/nutpie/compile_pymc.py:593-602``): every chain starts at the model's support point
    on the unconstrained scale plus U(-1, 1) on every coordinate that belongs to a jittered variable.  ``center[D]``; ``jitter[D]``
    = 1 where the coordinate is jittered.  The reference draws the jitter from a seed the chain's generator hands to
    ``initial_point_fn`` (``src/pymc.rs:505-534``); here the stream is keyed by (seed, global chain id), so the points do not depend
    on how the chains are sharded over GPUs."""

    center: Any
    jitter: Any

    def points(self, seed: int, n_chains: int) -> np.ndarray:
        c = np.asarray(self.center, dtype=np.float64).reshape(-1)
        j = np.broadcast_to(np.asarray(self.jitter, dtype=np.float64), c.shape)
        pts = np.empty((n_chains, c.size))
        for chain in range(n_chains):
            pts[chain] = c + j * np.random.default_rng([int(seed) & 0xFFFFFFFFFFFFFFFF, chain, 0x1417]).uniform(-1.0, 1.0, c.size)
        return pts


# --------------------------------------------------------------------------- the model front-end
@dataclass(frozen=True)
class DensitySourceModel(CompiledModel):
    """A model given as HIP source (module docstring).  Same contract as the other front-ends: ``n_dim``, ``shapes``, ``coords``,
    ``with_data``, ``_make_sampler``."""

    _source: str
    _n_dim: int
    _data: dict[str, Any]
    _lds_bytes: Any      # bytes of LDS scratch per chain: an int, or a function of the data dict
    _shared_bytes: Any   # bytes of LDS shared by a workgroup's chains: likewise
    _names: list[str]
    _shapes: list[tuple[int, ...]]
    _coords: dict[str, Any]
    _expand_func: Callable | None = None     # (x[N, D] numpy, **data) -> dict name -> [N, *shape]
    _init: Any = "uniform"
    _resident: bool = True                   # False: always the batched callback (launch per evaluation)
    _waves: int = 1                          # wavefronts that evaluate one chain's density together
    _scratch: Any = 0                        # doubles of device-memory scratch per chain (int or function of the data): data.scratch__
    _expand_lds_bytes: Any = 0               # the source defines nphip_expand (the expand step as a device function): its LDS scratch per row

    @property
    def n_dim(self):
        return self._n_dim

    @property
    def shapes(self):
        return {n: tuple(int(v) for v in s) for n, s in zip(self._names, self._shapes)}

    @property
    def coords(self):
        return self._coords

    @property
    def data(self):
        return dict(self._data)

    def with_data(self, **updates):
        """New data, same compiled library (reference compile_pymc.py:140-166: shared variables are swapped, nothing is
        recompiled).  Names and kinds (float array / integer array / float / int) must match."""
        for k, v in updates.items():
            if k not in self._data:
                raise ValueError(f"Unknown data variable: {k}")
            if _kind(v) != _kind(self._data[k]):
                raise ValueError(f"Data variable {k} must stay a {_kind(self._data[k])}")
        return dataclasses.replace(self, _data={**self._data, **updates})

    def _lds(self):
        r = lambda v: int(v(self._data)) if callable(v) else int(v)  # noqa: E731
        return r(self._lds_bytes), r(self._shared_bytes)

    def _device_data(self, n_chains: int, device: int) -> "DeviceData":
        """the data block, with ``scratch__`` — one block per resident chain of a launch — allocated on the device when asked for"""
        per = int(self._scratch(self._data)) if callable(self._scratch) else int(self._scratch)
        if per <= 0:
            return DeviceData(self._data, data_layout(self._data), device)
        import torch

        cpb = 4 if self._waves == 1 else 1
        slots = (max(1, int(n_chains)) + cpb - 1) // cpb * cpb
        scratch = torch.empty(slots * per, dtype=torch.float64, device=torch.device("cuda", device))
        return DeviceData(self._data, data_layout(self._data), device, device_arrays={"scratch__": scratch})

    def library_path(self, low_rank: bool = False) -> str:
        """Build (or find) this model's library without loading it — what an ahead-of-time build calls (no GPU needed).
        ``low_rank``: the variant whose resident kernel integrates under the low-rank metric."""
        return compile_density(self._source, data_layout(self._data), self._n_dim, waves=self._waves, low_rank=low_rank)

    def library(self, low_rank: bool = False) -> DensityLibrary:
        return DensityLibrary(self.library_path(low_rank))

    def logp_and_grad(self, x, device: int = 0, return_data: bool = False):
        """The compiled density on a block of positions ``x[N, n_dim]`` (one launch of the batched form, ``nphip_jit_logp``):
        ``(logp[N], grad[N, n_dim])`` as numpy arrays.  For checking a model; sampling never goes through the host."""
        import torch

        lib = self.library()
        dev = torch.device("cuda", device)
        xt = torch.as_tensor(np.atleast_2d(np.asarray(x, dtype=np.float64))).to(dev).contiguous()
        N, D = xt.shape
        dd = self._device_data(N, device)
        if D != self._n_dim:
            raise ValueError(f"positions have {D} columns, the model {self._n_dim} dimensions")
        g = torch.empty_like(xt)
        lp = torch.empty(N, dtype=torch.float64, device=dev)
        lds_bytes, shared_bytes = self._lds()
        batch = _Batch(dd.ptr, lds_bytes // 8, shared_bytes // 8)
        call = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(lib.logp_addr)
        with torch.cuda.device(dev):
            rc = call(N, D, xt.data_ptr(), g.data_ptr(), lp.data_ptr(), torch.cuda.current_stream().cuda_stream, C.addressof(batch))
            torch.cuda.synchronize()
        if rc != 0:
            raise RuntimeError(f"launching the density failed ({rc}): too much LDS?")
        if return_data:   # (the data block as it is after the call: sources that write into their data — cycle counters — are read back)
            return lp.cpu().numpy(), g.cpu().numpy(), {k: t.cpu().numpy() for k, t in dd.tensors.items()}
        return lp.cpu().numpy(), g.cpu().numpy()

    def _make_model(self, init_mean=None, settings=None, device: int = 0, resident: bool | None = None):
        low_rank = settings is not None and getattr(settings, "_adaptation", "diag") == "low_rank"
        # adaptation="low_rank": the resident kernel under the metric where the model has a resident kernel at all (up to 1024
        # dimensions), else the batched callback on the memory-resident kernels
        lr_resident = low_rank and self._n_dim <= 1024
        lib = self.library(low_rank=lr_resident)
        n_chains = int(getattr(settings, "num_chains", 0) or 0) if settings is not None else 0
        if (callable(self._scratch) or self._scratch) and n_chains <= 0:
            raise ValueError("a density with device-memory scratch needs the settings (num_chains) to size it")
        dd = self._device_data(n_chains, device)
        use_resident = self._resident if resident is None else resident
        if settings is not None and (bool(getattr(settings, "store_divergences", False)) or (low_rank and not lr_resident)):
            use_resident = False   # the divergence record needs the pre-step state in memory
        if lr_resident and not use_resident:
            lib = self.library()   # (the batched form comes from the plain library)
        if self._n_dim > 1024:
            use_resident = False
        lds_bytes, shared_bytes = self._lds()
        if use_resident:
            model = _lib.JitDensityModel(self._n_dim, lib.launch_addr, lib.nv, dd.ptr, lds_bytes, shared_bytes, keep_alive=(lib, dd), waves_per_chain=lib.waves,
                                         low_rank=lib.low_rank)
        else:
            batch = _Batch(dd.ptr, lds_bytes // 8, shared_bytes // 8)
            model = _lib.NativeDeviceCallbackModel(self._n_dim, lib.logp_addr, C.addressof(batch), keep_alive=(lib, dd, batch))
        if isinstance(self._init, str):
            model.set_init(self._init)
        elif isinstance(self._init, JitteredInit):
            if settings is None:
                raise ValueError("jittered initial points need the settings (seed, num_chains)")
            model.set_init("explicit", self._init.points(int(settings.seed), int(settings.num_chains)))
        else:
            model.set_init("explicit", np.asarray(self._init, dtype=np.float64))
        if lib.expand_addr is not None:
            # the expand step behind the C-ABI, batched on the device (nphip_model_set_device_expand): one launch per block of stored
            # draws; the flat rows are split into variables by CompiledModel._unflatten
            e_lds = int(self._expand_lds_bytes(self._data)) if callable(self._expand_lds_bytes) else int(self._expand_lds_bytes)
            ebatch = _Batch(dd.ptr, e_lds // 8, shared_bytes // 8)
            total = sum(int(np.prod(shp, dtype=np.int64)) if len(shp) else 1 for shp in self._shapes)
            model.set_device_expand(total, lib.expand_addr, C.addressof(ebatch), keep_alive=(lib, dd, ebatch))
        return model

    def _make_sampler(self, settings, init_mean, cores, progress_type, extra_callback, extra_callback_rate, store, **engine_kw):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("a device density needs a GPU: the nutpie-hip engine has no CPU fallback")
        device = int(engine_kw.get("device", 0) or 0)
        model = self._make_model(settings=settings, device=device)
        if int(engine_kw.get("host_groups", 0) or 0) >= 2 and (callable(self._scratch) or self._scratch):
            # groups of chains run their (kernel, callback) pairs CONCURRENTLY on streams of their own, and the batched form of a density counts
            # its scratch slots (NPHIP_CHAIN_SLOT) from 0 in every launch: two groups would share the blocks of data.scratch__ — a data race.
            # A model that spills scratch to device memory runs ungrouped.
            import warnings

            warnings.warn("host_groups is ignored for a density with device-memory scratch (concurrent groups would share its scratch blocks)", UserWarning, stacklevel=3)
            engine_kw = {**engine_kw, "host_groups": 0}
        if not engine_kw.get("waves_per_chain") and (self._waves > 1 or self._n_dim <= 1024):
            # the batched form runs the engine with as many waves per chain as the resident kernel has: a chain's sums are taken
            # in the same order either way, so both forms of one library draw identically
            engine_kw = {**engine_kw, "waves_per_chain": self._waves}
        return _lib.PySampler.from_pyfunc(settings, cores, model, progress_type, extra_callback, extra_callback_rate, store, **engine_kw)

    def _expand_draws(self, draws):
        n, T, D = draws.shape
        if self._expand_func is None:
            return {self._names[0]: draws.reshape(n, T, *self._shapes[0])}
        flat = self._expand_func(draws.reshape(n * T, D), **self._data)
        return {name: np.asarray(flat[name]).reshape(n, T, *shape) for name, shape in zip(self._names, self._shapes)}


def _times8(v):
    return (lambda data: 8 * int(v(data))) if callable(v) else 8 * int(v)


def from_density_source(ndim: int, source: str, data: dict[str, Any] | None = None, *, lds_doubles_per_chain: int = 0, lds_doubles_shared: int = 0,
                        expand_fn: Callable | None = None, expanded_names: list[str] | None = None, expanded_shapes=None,
                        coords=None, dims=None, init="uniform", resident: bool = True, reparameterized_names=None, waves_per_chain: int = 1,
                        scratch_doubles_per_chain=0, expand_lds_doubles=0) -> DensitySourceModel:
    """A model from the HIP source of its log-density (module docstring): ``source`` defines ``nphip_density``; ``data`` are the
    arrays / scalars it reads through ``NphipData``; ``lds_doubles_per_chain`` the LDS scratch it uses per chain, ``lds_doubles_shared``
    the LDS its ``nphip_density_stage`` fills once per workgroup (each an int or a function of the data dict: ``with_data`` may
    change the sizes).  ``expand_fn`` (optional)
    maps a numpy block ``[N, ndim]`` of draws to the dict of expanded variables, as :func:`nutpie_amd.from_torchfunc` does.  A source
    that also defines ``__device__ double nphip_expand(const NphipData&, int dim, const double* x, double* out, double* lds, const
    double* shared, int lane)`` — one row of the flat expanded vector per draw, ``expand_lds_doubles`` of LDS scratch per row —
    gets its expand step run on the device behind the C-ABI instead.
    ``waves_per_chain`` (1, 2 or 4): that many wavefronts evaluate one chain's density together — the source then strides its loops
    by ``NPHIP_CHAIN_THREADS`` (``lane`` runs over ``0 .. NPHIP_CHAIN_THREADS - 1``), sums with ``nphip_chain_sum*`` and separates
    its phases with ``nphip_chain_barrier()``; with fewer chains than the device has SIMDs (1024) this is what fills it."""
    if waves_per_chain not in (1, 2, 4):
        raise ValueError("waves_per_chain must be 1, 2 or 4")
    if expanded_names is None:
        if expand_fn is not None:
            raise ValueError("expand_fn needs expanded_names and expanded_shapes")
        expanded_names, expanded_shapes = ["x"], [(ndim,)]
    if "nphip_density" not in source:
        raise ValueError("the source must define `__device__ double nphip_density(const NphipData& data, int dim, const double* x, double* grad, double* lds, "
                         "const double* shared, int lane)`")
    if lds_doubles_shared and "nphip_density_stage" not in source:
        raise ValueError("lds_doubles_shared needs `__device__ void nphip_density_stage(const NphipData& data, double* shared, int thread, int n_threads)` in the source")
    data = dict(data or {})
    if callable(scratch_doubles_per_chain) or scratch_doubles_per_chain:
        # ``scratch_doubles_per_chain`` (int or function of the data): scratch in DEVICE memory for what does not fit the LDS — the
        # source finds its block at ``(double*)data.scratch__ + (size_t)NPHIP_CHAIN_SLOT * <doubles per chain>``
        if "scratch__" in data:
            raise ValueError("the data field `scratch__` is reserved for the device-memory scratch")
        data["scratch__"] = np.zeros(1)
    return DensitySourceModel(dims=dict(dims or {}), _source=source, _n_dim=int(ndim), _data=data, _lds_bytes=_times8(lds_doubles_per_chain), _shared_bytes=_times8(lds_doubles_shared),
                              _names=list(expanded_names), _shapes=[tuple(s) for s in expanded_shapes], _coords=dict(coords or {}),
                              _expand_func=expand_fn, _init=init, _resident=bool(resident), _waves=int(waves_per_chain), _scratch=scratch_doubles_per_chain,
                              _expand_lds_bytes=_times8(expand_lds_doubles), reparameterized_names=reparameterized_names)
