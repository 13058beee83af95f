import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as _oracle

    _oracle.build()
    return _oracle


@pytest.fixture(scope="session")
def radon_device_lib():
    """hipcc-built shared library with the native device log-density of the radon model (tests/fixtures/radon_device.hip):
    a model written against the batched DEVICE callback of the C-ABI."""
    src = os.path.join(FIXTURES, "radon_device.hip")
    out = os.path.join(FIXTURES, "libradon_device.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", out, src], check=True)
    from nutpie_amd import _lib

    _lib.lib()  # first: it loads torch's HIP runtime before any other library can pull in a second one
    lib = ctypes.CDLL(out)
    lib.radon_device_create.restype = ctypes.c_void_p
    lib.radon_device_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.radon_device_free.argtypes = [ctypes.c_void_p]
    return lib


@pytest.fixture(scope="session")
def scaled_normal_device_lib():
    """hipcc-built device twin of eight_schools.c::scaled_normal_logp (tests/fixtures/scaled_normal_device.hip): the same IEEE
    operations in the same order behind the batched DEVICE callback, so that a device-callback job can be compared with the
    oracle bit for bit."""
    src = os.path.join(FIXTURES, "scaled_normal_device.hip")
    out = os.path.join(FIXTURES, "libscaled_normal_device.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, src], check=True)
    from nutpie_amd import _lib

    _lib.lib()  # first: it loads torch's HIP runtime before any other library can pull in a second one
    return ctypes.CDLL(out)


@pytest.fixture(scope="session")
def fixture_lib():
    """gcc-built shared library with the eight-schools test model (tests/fixtures/eight_schools.c)."""
    src = os.path.join(FIXTURES, "eight_schools.c")
    out = os.path.join(FIXTURES, "libeight_schools.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src, "-lm"], check=True)
    lib = ctypes.CDLL(out)
    lib.bs_model_construct.restype = ctypes.c_void_p
    lib.bs_model_construct.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_void_p]
    lib.bs_model_destruct.argtypes = [ctypes.c_void_p]
    return lib


@pytest.fixture(scope="session")
def bs_standin():
    """gcc-built stand-in for a BridgeStan model library (tests/fixtures/bs_standin.c): BridgeStan's C API for the Stan program
    of the reference's golden-vector test and for a matrix-valued model, plus the same density as raw C callbacks."""
    src = os.path.join(FIXTURES, "bs_standin.c")
    out = os.path.join(FIXTURES, "libbs_standin.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src, "-lm"], check=True)
    lib = ctypes.CDLL(out)
    lib.bs_model_construct.restype = ctypes.c_void_p
    lib.bs_model_construct.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_void_p]
    lib.bs_model_destruct.argtypes = [ctypes.c_void_p]
    lib.bs_param_unc_num.argtypes = [ctypes.c_void_p]
    lib.bs_param_num.argtypes = [ctypes.c_void_p, ctypes.c_bool, ctypes.c_bool]
    lib.bs_param_names.argtypes = [ctypes.c_void_p, ctypes.c_bool, ctypes.c_bool]
    lib.bs_param_names.restype = ctypes.c_char_p
    lib.bs_rng_destruct.argtypes = [ctypes.c_void_p]
    lib.bs_param_constrain.argtypes = [ctypes.c_void_p, ctypes.c_bool, ctypes.c_bool, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.bs_log_density_gradient.argtypes = [ctypes.c_void_p, ctypes.c_bool, ctypes.c_bool, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


class FakeBridgeStanModel:
    """What ``bridgestan.StanModel`` offers to nutpie_amd.compile_stan (BridgeStan itself is not installable here): the loaded
    library, the model handle, ``param_unc_num`` and ``param_names``."""

    def __init__(self, lib, data: bytes):
        self.stanlib = lib
        self.model = ctypes.c_void_p(lib.bs_model_construct(data, 0, None))
        assert self.model.value

    def param_unc_num(self):
        return self.stanlib.bs_param_unc_num(self.model)

    def param_names(self, include_tp=False, include_gq=False):
        return self.stanlib.bs_param_names(self.model, include_tp, include_gq).decode().split(",")

    def __del__(self):
        self.stanlib.bs_model_destruct(self.model)


@pytest.fixture(scope="session")
def hip():
    """The HIP engine binding; GPU tests fail (not skip) if the extension is missing."""
    from nutpie_amd import _lib

    _lib.lib()
    return _lib


def fn_addr(cfunc):
    return ctypes.cast(cfunc, ctypes.c_void_p).value


INT_STATS = ("depth", "n_steps", "index_in_trajectory", "diverging", "maxdepth_reached", "tuning")
FLOAT_STATS = ("energy", "energy_error", "logp", "step_size", "step_size_bar", "mean_tree_accept", "mean_tree_accept_sym")


def assert_trace_equal(got, want, *, draws=True, float_rtol=0.0):
    """got: nutpie_amd PyTrace, want: oracle Trace.  Integer statistics must be bit-identical;
    floating-point statistics/draws bit-identical by default (float_rtol=0)."""
    for k in INT_STATS:
        a = np.asarray(got.stats[k]).astype(np.int64)
        b = np.asarray(want.stats[k]).astype(np.int64)
        bad = np.argwhere(a != b)
        assert bad.size == 0, f"{k}: {len(bad)} mismatches, first at {bad[0]}: {a[tuple(bad[0])]} != {b[tuple(bad[0])]}"
    for k in FLOAT_STATS:
        a, b = np.asarray(got.stats[k]), np.asarray(want.stats[k])
        if float_rtol == 0.0:
            assert np.array_equal(a, b), f"{k}: max abs diff {np.nanmax(np.abs(a - b))}"
        else:
            np.testing.assert_allclose(a, b, rtol=float_rtol, atol=0, err_msg=k)
    if draws:
        if float_rtol == 0.0:
            assert np.array_equal(got.draws, want.draws), f"draws: max abs diff {np.abs(got.draws - want.draws).max()}"
        else:
            np.testing.assert_allclose(got.draws, want.draws, rtol=float_rtol, atol=0)
