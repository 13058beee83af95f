"""nutpie_amd.sample end to end on the GPU — the behavioural pins of the reference's own tests
(tests/test_pymc.py, tests/test_stan.py) that do not depend on nuts-rs' RNG stream."""
import os
import time

import ctypes

import numpy as np
import pytest

import nutpie_amd

pytestmark = pytest.mark.gpu


def test_sample_shapes_and_groups():
    # tests/test_pymc.py:189-191: draws=17, tune=100, chains=1 -> posterior shape (1, 17, ...)
    tr = nutpie_amd.sample(nutpie_amd.std_normal(5), chains=1, draws=17, tune=100, seed=1, progress_bar=False)
    assert tr.posterior.x.shape == (1, 17, 5)
    assert tr.warmup_posterior.x.shape == (1, 100, 5)
    for k in ("depth", "maxdepth_reached", "logp", "energy", "diverging", "step_size", "step_size_bar", "n_steps"):  # docs/sample-stats.qmd:47-56
        assert k in tr.sample_stats, k
    assert tr.sample_stats.attrs["inference_library"] == "nutpie"
    tr = nutpie_amd.sample(nutpie_amd.std_normal(5), chains=2, draws=10, tune=20, seed=1, progress_bar=False, save_warmup=False)
    assert "warmup_posterior" not in tr and tr.posterior.x.shape == (2, 10, 5)
    # default sizes: 6 chains x 1000 draws (tests/test_stan.py:241)
    tr = nutpie_amd.sample(nutpie_amd.std_normal(2), seed=1, progress_bar=False)
    assert tr.posterior.x.shape == (6, 1000, 2)


def test_seed_semantics():
    # tests/test_stan.py:67-101
    m = nutpie_amd.std_normal(3)
    a = nutpie_amd.sample(m, seed=42, chains=4, draws=50, tune=50, progress_bar=False)
    b = nutpie_amd.sample(m, seed=42, chains=4, draws=50, tune=50, progress_bar=False)
    c = nutpie_amd.sample(m, seed=43, chains=4, draws=50, tune=50, progress_bar=False)
    assert np.array_equal(a.posterior.x.values, b.posterior.x.values)      # max-ULP identical (test_stan.py:298-301)
    assert not np.allclose(a.posterior.x.values, c.posterior.x.values)
    for i in range(4):
        for j in range(i + 1, 4):
            assert not np.allclose(a.posterior.x.values[i], a.posterior.x.values[j])


def test_store_flags_add_exactly_the_listed_stats():
    # tests/test_pymc.py:303-349
    m = nutpie_amd.diag_gaussian([1.0, 100.0, 0.01])
    base = nutpie_amd.sample(m, chains=2, draws=20, tune=50, seed=1, progress_bar=False)
    for k in ("gradient", "unconstrained_draw", "mass_matrix_inv", "divergence_start"):
        assert k not in base.sample_stats
    tr = nutpie_amd.sample(m, chains=2, draws=20, tune=50, seed=1, progress_bar=False, store_gradient=True, store_unconstrained=True,
                           store_mass_matrix=True, store_divergences=True, max_energy_error=0.3)
    assert tr.sample_stats.gradient.shape == (2, 20, 3) and tr.sample_stats.unconstrained_draw.shape == (2, 20, 3)
    assert tr.sample_stats.mass_matrix_inv.shape == (2, 20, 3)
    np.testing.assert_allclose(tr.sample_stats.gradient.values, -tr.posterior.x.values / np.array([1.0, 100.0, 0.01]) ** 2, rtol=1e-12)
    assert np.array_equal(tr.sample_stats.unconstrained_draw.values, tr.posterior.x.values)
    for k in ("divergence_start", "divergence_end", "divergence_momentum", "divergence_start_gradient"):
        assert k in tr.warmup_sample_stats
    div = tr.warmup_sample_stats.diverging.values
    assert div.sum() > 0
    ds = tr.warmup_sample_stats.divergence_start.values
    assert np.all(np.isfinite(ds[div])) and np.all(np.isnan(ds[~div]))


def test_adaptation_draw_diag_and_settings_attr():
    m = nutpie_amd.diag_gaussian([0.1, 1.0, 10.0])
    tr = nutpie_amd.sample(m, chains=4, draws=300, tune=300, seed=3, progress_bar=False, adaptation="draw_diag", store_mass_matrix=True)
    import json

    st = json.loads(tr.sample_stats.attrs["inference_library_settings"])
    assert st["settings"]["adapt_options"]["mass_matrix_options"]["use_grad_based_estimate"] is False
    mm = tr.sample_stats.mass_matrix_inv.values[:, -1]
    assert np.all(np.abs(np.log(mm / np.array([0.01, 1.0, 100.0]))) < 1.0)
    sd = tr.posterior.x.values.std((0, 1))
    np.testing.assert_allclose(sd, [0.1, 1.0, 10.0], rtol=0.15)


def test_progress_callback_final_state():
    # tests/test_pymc.py:37-66
    seen = []
    tr = nutpie_amd.sample(nutpie_amd.std_normal(50), chains=3, draws=200, tune=200, seed=1, progress_bar=False,
                           progress_callback=seen.append, progress_rate=10)
    assert tr.posterior.x.shape == (3, 200, 50)
    assert len(seen) >= 1
    last = seen[-1]
    assert len(last) == 3
    for p in last:
        assert isinstance(p, nutpie_amd.ChainProgress)
        assert p.finished_draws == p.total_draws == 400
        assert isinstance(p.divergences, int) and isinstance(p.tuning, bool) and isinstance(p.started, bool)
        assert isinstance(p.step_size, float) and p.step_size > 0
        assert p.total_num_steps >= p.latest_num_steps >= 1 and p.num_steps == p.latest_num_steps
        assert isinstance(p.runtime_ms, int) and isinstance(p.divergent_draws, list)


def test_non_blocking_timeout_pause_abort():
    # tests/test_pymc.py:224-286: blocking=False, wait(timeout) raises TimeoutError, cancel returns fast,
    # pause/resume, abort returns the partial trace
    from nutpie_amd.gaussian import ar1_gaussian

    m = ar1_gaussian(500, rho=0.99)   # deep trees: slow enough to observe (trace: 256 x 21000 x 500 x 8 B = 21 GB)
    smp = nutpie_amd.sample(m, chains=256, draws=20000, tune=1000, seed=1, progress_bar=False, blocking=False, maxdepth=10)
    with pytest.raises(TimeoutError):
        smp.wait(timeout=0.3)
    assert not smp.is_finished
    smp.pause()
    time.sleep(0.2)
    part = smp.inspect()
    n1 = int(np.isfinite(part.warmup_posterior.x.values[:, :, 0]).sum() + np.isfinite(part.posterior.x.values[:, :, 0]).sum()) if "posterior" in part else 0
    time.sleep(0.3)
    part2 = smp.inspect()
    n2 = int(np.isfinite(part2.warmup_posterior.x.values[:, :, 0]).sum())
    assert n2 == n1 or n2 - n1 <= 256 * 2       # paused: (almost) no progress
    smp.resume()
    time.sleep(0.3)
    t0 = time.time()
    tr = smp.abort()
    assert time.time() - t0 < 20
    wx = tr.warmup_posterior.x.values
    assert wx.shape[0] == 256 and np.isfinite(wx[:, 0, 0]).all()      # every chain produced at least one draw
    assert np.isnan(wx).any() or wx.shape[1] < 1000                     # but not the whole run
    # cancel: discards and returns quickly
    smp2 = nutpie_amd.sample(m, chains=64, draws=20000, tune=1000, seed=1, progress_bar=False, blocking=False)
    time.sleep(0.2)
    t0 = time.time()
    smp2.cancel()
    assert time.time() - t0 < 10


def test_from_pyfunc_drop_in():
    # the reference's generic model API (compiled_pyfunc.py:108-155) on the GPU engine (host-callback path)
    def make_logp():
        return lambda x: (-0.5 * float(x @ x) - 0.5 * float((x[0] - 1) ** 2), -x - np.array([x[0] - 1, 0.0]))

    def make_expand(seed1, seed2, chain):
        return lambda x: {"y": x, "ysq": np.array(x[0] ** 2)}

    m = nutpie_amd.from_pyfunc(2, make_logp, make_expand, [np.float64, np.float64], [(2,), ()], ["y", "ysq"],
                               make_initial_point_fn=lambda seed: np.zeros(2))
    tr = nutpie_amd.sample(m, chains=2, draws=150, tune=150, seed=1, progress_bar=False)
    assert tr.posterior.y.shape == (2, 150, 2) and tr.posterior.ysq.shape == (2, 150)
    assert abs(tr.posterior.y.values[..., 0].mean() - 0.5) < 0.3   # N(0.5, 1/2) in the first coordinate
    assert np.array_equal(tr.posterior.ysq.values, tr.posterior.y.values[..., 0] ** 2)


def test_raw_callback_front_end(fixture_lib):
    # the pointer a reference-compiled PyMC model carries (numba cfunc address) plugs in unchanged
    from nutpie_amd.compile_pymc import from_raw_callback
    from tests.conftest import fn_addr

    m = from_raw_callback(10, fn_addr(fixture_lib.eight_schools_logp), name="theta", n_threads=4, init="normal", keep_alive=fixture_lib)
    tr = nutpie_amd.sample(m, chains=256, draws=200, tune=300, seed=4, progress_bar=False)  # BASELINE.json config 4 sizes
    assert tr.posterior.theta.shape == (256, 200, 10)
    mu = tr.posterior.theta.values[..., 0]
    assert 3.0 < mu.mean() < 6.0 and 2.0 < mu.std() < 5.0          # eight schools: mu ~ 4.4 +- 3.3
    assert tr.sample_stats.diverging.values.mean() < 0.05


def test_raw_expand_callback_through_the_c_abi(hip, oracle, fixture_lib):
    """The reference's ExpandFunc (src/pymc.rs:31-37, 64-95, 217-286) behind the C-ABI: a C expand callback evaluated by the
    engine over the stored trace, rows of unfinished draws NaN; error codes surface as 'Expand function returned error code'."""
    from tests.conftest import fn_addr

    m = hip.HostCallbackModel(10, fn_addr(fixture_lib.eight_schools_logp), n_threads=2, keep_alive=fixture_lib)
    m.set_init("normal")
    m.set_expand(18, fn_addr(fixture_lib.eight_schools_expand))
    assert m.expanded_dim == 18
    s = hip.PyNutsSettings.Diag(11)
    s.update(num_tune=50, num_draws=30, num_chains=5)
    smp = hip.PySampler(s, m)
    smp.wait()
    ex = smp.expanded()
    raw = smp._copy("draws", np.float64, vec=True)
    assert ex.shape == (5, 80, 18)
    assert np.array_equal(ex[..., 0], raw[..., 0]) and np.array_equal(ex[..., 2:10], raw[..., 2:])
    np.testing.assert_allclose(ex[..., 1], np.exp(raw[..., 1]), rtol=1e-14)                       # (libm's exp vs numpy's)
    assert np.array_equal(ex[..., 10:], ex[..., :1] + ex[..., 1:2] * raw[..., 2:])
    # "expanded" is also a name of the trace hand-off
    again = np.empty_like(ex)
    assert hip.lib().nphip_sampler_copy_stat(smp._h, b"expanded", again.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(again.nbytes)) == 0
    assert np.array_equal(again, ex)
    smp.close()
    # a partial trace: unfinished rows are NaN
    s.update(num_tune=3000, num_draws=3000)
    smp = hip.PySampler(s, m, evals_per_launch=1)
    import time

    time.sleep(0.3)
    smp.abort()
    ex = smp.expanded()
    fin = smp.inspect().finished
    assert fin.max() < 6000
    for c in range(5):
        assert np.isfinite(ex[c, : fin[c]]).all() and np.isnan(ex[c, fin[c]:]).all()
    smp.close()
    # wrong sizes / failing callbacks
    bad = hip.HostCallbackModel(10, fn_addr(fixture_lib.eight_schools_logp), n_threads=1)
    bad.set_init("normal")
    bad.set_expand(17, fn_addr(fixture_lib.eight_schools_expand))   # the callback checks `expanded` (compile_pymc.py:1026-1029)
    s.update(num_tune=5, num_draws=5)
    smp = hip.PySampler(s, bad)
    smp.wait()
    with pytest.raises(RuntimeError, match="Expand function returned error code -1"):
        smp.expanded()
    smp.close()


def test_batched_device_expand_through_the_c_abi(hip):
    """nphip_model_set_device_expand: the expand step as ONE batched call per block of stored draws, on the engine's stream
     — here a callback that wraps the device pointers as torch tensors."""
    import torch

    from nutpie_amd.distributed import device_tensor

    calls = []

    def dev_expand(n_rows, dim, expanded, x_ptr, out_ptr, stream_ptr, _user):
        x = device_tensor(x_ptr, (n_rows, dim), "float64", 0)
        out = device_tensor(out_ptr, (n_rows, expanded), "float64", 0)
        st = torch.cuda.ExternalStream(stream_ptr) if stream_ptr else torch.cuda.current_stream()
        with torch.cuda.stream(st):
            out[:, :dim] = x
            out[:, dim] = (x * x).sum(1)
        calls.append(int(n_rows))
        return 0

    cb = hip.DEVICE_EXPAND_FN(dev_expand)
    m = hip.TridiagGaussianModel(np.linspace(0.5, 2.0, 7))
    m.set_device_expand(8, ctypes.cast(cb, ctypes.c_void_p).value, keep_alive=cb)
    s = hip.PyNutsSettings.Diag(2)
    s.update(num_tune=30, num_draws=20, num_chains=9)
    smp = hip.PySampler(s, m)
    smp.wait()
    ex = smp.expanded()
    raw = smp._copy("draws", np.float64, vec=True)
    assert ex.shape == (9, 50, 8) and sum(calls) == 9 * 50
    assert np.array_equal(ex[..., :7], raw)
    np.testing.assert_allclose(ex[..., 7], (raw**2).sum(-1), rtol=1e-14)
    smp.close()


def test_logp_return_is_read_as_c_int(hip, oracle, fixture_lib):
    # numba declares the callback int64 (compile_pymc.py:975-981), the reference reads c_int (src/pymc.rs:23-29): a return
    # register with a dirty upper half and a clean low word is "0 = ok" — identical trace to the clean function
    from tests.conftest import assert_trace_equal, fn_addr
    from tests.test_gpu_parity import oracle_settings, run_engine

    got, W = run_engine(hip, hip.HostCallbackModel(10, fn_addr(fixture_lib.eight_schools_logp_dirty_high), n_threads=2), chains=6, tune=80, draws=30,
                        seed=5, init="normal")
    want = oracle.sample_callback(oracle_settings(oracle, chains=6, tune=80, draws=30, seed=5, W=W, init_kind=1), 10, fn_addr(fixture_lib.eight_schools_logp))
    assert_trace_equal(got, want)


def test_raw_callback_front_end_with_expand(fixture_lib):
    # what a reference-compiled PyMC model carries — logp AND expand cfunc addresses — gives the expanded variables
    from nutpie_amd.compile_pymc import from_raw_callback
    from tests.conftest import fn_addr

    m = from_raw_callback(10, fn_addr(fixture_lib.eight_schools_logp), n_threads=4, init="normal", keep_alive=fixture_lib,
                          expand_address=fn_addr(fixture_lib.eight_schools_expand),
                          expanded_shapes={"mu": (), "tau": (), "theta_tilde": (8,), "theta": (8,)}, dims={"theta": ("school",)},
                          coords={"school": list("ABCDEFGH")})
    tr = nutpie_amd.sample(m, chains=32, draws=100, tune=200, seed=4, progress_bar=False, store_unconstrained=True)
    assert tr.posterior.theta.shape == (32, 100, 8) and tr.posterior.tau.shape == (32, 100)
    assert (tr.posterior.tau.values > 0).all()
    assert np.allclose(tr.posterior.theta.values, tr.posterior.mu.values[..., None] + tr.posterior.tau.values[..., None] * tr.posterior.theta_tilde.values)
    assert 2.0 < tr.posterior.mu.values.mean() < 7.0


def test_python_callable_errors_keep_their_traceback(hip):
    # the user's exception is chained, not swallowed into "fatal error"
    import nutpie_amd

    def make_logp():
        def f(x):
            return 0.0, np.zeros(3, dtype=np.float32)   # wrong dtype: ReturnTypeError of src/pyfunc.rs

        return f

    m = nutpie_amd.from_pyfunc(3, make_logp, lambda *a: (lambda x: {"y": x}), [np.float64], [(3,)], ["y"])
    with pytest.raises(RuntimeError, match="Return type of logp function") as e:
        nutpie_amd.sample(m, chains=2, tune=5, draws=5, progress_bar=False)
    assert isinstance(e.value.__cause__, TypeError)


def test_radon_torch_model_config3():
    # BASELINE.json config 3: Radon hierarchical model, 512 chains, batched torch logp (synthetic radon-shaped data)
    import torch

    from nutpie_amd.radon import radon_model, synthetic_radon_data

    data = synthetic_radon_data()
    m = radon_model(data)
    assert m.n_dim == 173
    # gradient of the hand-written density agrees with finite differences (fp64)
    f = m._make_logp_func()
    x = torch.randn(3, 173, dtype=torch.float64, device="cuda") * 0.3
    lp, g = f(x)
    eps = 1e-6
    for k in (0, 5, 85, 86, 100, 171, 172):
        xp = x.clone(); xp[:, k] += eps
        xm = x.clone(); xm[:, k] -= eps
        fd = (f(xp)[0] - f(xm)[0]) / (2 * eps)
        assert torch.allclose(fd, g[:, k], rtol=1e-5, atol=1e-6), k
    tr = nutpie_amd.sample(m, chains=512, tune=300, draws=200, seed=7, progress_bar=False)
    assert tr.posterior.county_effect.shape == (512, 200, 85)
    assert np.abs(tr.posterior.county_effect.values.sum(-1)).max() < 1e-9       # zero-sum constraint
    # the synthetic data were generated with intercept 1.3, floor effect -0.6, sigma 0.75
    assert abs(tr.posterior.intercept.values.mean() - 1.3) < 0.15
    assert abs(tr.posterior.floor_effect.values.mean() + 0.6) < 0.2
    assert abs(tr.posterior.sigma.values.mean() - 0.75) < 0.08
    assert tr.sample_stats.diverging.values.mean() < 0.02
    # HIP-graph replay of the same density gives the same chains (same kernels, same order)
    tr2 = nutpie_amd.sample(radon_model(data, use_graph=True), chains=64, tune=50, draws=20, seed=7, progress_bar=False)
    tr3 = nutpie_amd.sample(radon_model(data), chains=64, tune=50, draws=20, seed=7, progress_bar=False)
    assert np.array_equal(tr2.posterior.sigma.values, tr3.posterior.sigma.values)
    # expand step on the device (default) == expand step on the host, and only then are raw draws copied back
    tr4 = nutpie_amd.sample(radon_model(data, expand_on_device=False), chains=64, tune=50, draws=20, seed=7, progress_bar=False,
                            store_unconstrained=True)
    for name in ("intercept", "county_effect", "county_floor_effect", "sigma", "county_sd"):
        np.testing.assert_allclose(tr3.posterior[name].values, tr4.posterior[name].values, rtol=1e-13, atol=1e-15)
    assert "unconstrained_draw" in tr4.sample_stats and "unconstrained_draw" not in tr3.sample_stats
    raw = nutpie_amd.sample(radon_model(data), chains=8, tune=20, draws=5, seed=7, progress_bar=False, return_raw_trace=True)
    assert raw.draws is not None and raw.expanded["sigma"].shape == (8, 25)


def test_from_torch_density_autograd_model():
    import torch

    sd = torch.tensor([0.5, 1.0, 3.0], dtype=torch.float64, device="cuda")
    m = nutpie_amd.from_torch_density(3, lambda x: -0.5 * ((x / sd) ** 2).sum(-1))
    tr = nutpie_amd.sample(m, chains=64, tune=300, draws=300, seed=2, progress_bar=False)
    x = tr.posterior.x.values
    assert x.shape == (64, 300, 3)
    np.testing.assert_allclose(x.std((0, 1)), [0.5, 1.0, 3.0], rtol=0.08)
    assert np.abs(x.mean((0, 1)) / np.array([0.5, 1.0, 3.0])).max() < 0.1


def test_native_device_callback_radon_kernel(radon_device_lib):
    """A model written in HIP against the batched DEVICE callback of the C-ABI (tests/fixtures/radon_device.hip: the
    radon density of config 3 as one kernel per evaluation): agrees with the torch density, and samples the same
    posterior — with no Python and no launch-bound torch graph on the per-leapfrog path."""
    import ctypes

    import torch

    from nutpie_amd import _lib
    from nutpie_amd.radon import radon_model, synthetic_radon_data

    data = synthetic_radon_data()
    n = int(data["county_idx"].max()) + 1
    cty = np.ascontiguousarray(data["county_idx"], dtype=np.int32)
    fl, y = np.ascontiguousarray(data["floor"]), np.ascontiguousarray(data["log_radon"])
    h = radon_device_lib.radon_device_create(n, len(y), cty.ctypes.data, fl.ctypes.data, y.ctypes.data)
    assert h
    try:
        D = 2 * n + 3
        fn = ctypes.cast(radon_device_lib.radon_device_logp, ctypes.c_void_p).value
        # 1. the kernel against the torch density (hand-derived gradient) on random points
        f = radon_model(data)._make_logp_func()
        x = torch.randn(37, D, dtype=torch.float64, device="cuda") * 0.4
        g = torch.empty_like(x)
        lp = torch.empty(37, dtype=torch.float64, device="cuda")
        call = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_void_p)(fn)
        assert call(37, D, x.data_ptr(), g.data_ptr(), lp.data_ptr(), 0, h) == 0
        torch.cuda.synchronize()
        lp_ref, g_ref = f(x)
        assert torch.allclose(lp, lp_ref, rtol=1e-12, atol=1e-9) and torch.allclose(g, g_ref, rtol=1e-10, atol=1e-9)
        # 2. sampling through the engine: deterministic, and the posterior of the synthetic data
        def run(graph_steps=0):
            s = _lib.PyNutsSettings.Diag(7)
            s.update(num_tune=300, num_draws=200, num_chains=256)
            smp = _lib.PySampler(s, _lib.NativeDeviceCallbackModel(D, fn, h, keep_alive=radon_device_lib), graph_steps=graph_steps)
            smp.wait()
            return smp.take_results()
        a, b = run(), run()
        assert np.array_equal(a.draws, b.draws)
        # (engine kernel + callback) x 16 captured in a HIP graph and replayed: same chains
        c = run(graph_steps=16)
        assert np.array_equal(a.draws, c.draws) and np.array_equal(a.stats["n_steps"], c.stats["n_steps"])
        post = a.draws[:, 300:]
        assert abs(post[..., 0].mean() - 1.3) < 0.15 and abs(post[..., n + 1].mean() + 0.6) < 0.2
        assert abs(np.exp(post[..., 2 * n + 2]).mean() - 0.75) < 0.08
        assert a.stats["diverging"][:, 300:].mean() < 0.02
    finally:
        radon_device_lib.radon_device_free(h)


def test_radon_device_callback_against_the_oracle(radon_device_lib, oracle):
    """BASELINE.json config 3's density through the device-callback engine against the CPU oracle sampling the SAME density
    (tests/fixtures/radon_host.c: the formulas of radon_device.hip with sequential sums).  The two sides evaluate the density
    in different summation orders (a wavefront butterfly against a loop) and with different exp implementations (ocml / libm),
    so their floats agree to rounding only and a chaotic integrator amplifies that:
      * tolerance on the FIRST draws of every chain: rtol 1e-9 on positions and energies, integer statistics equal;
      * afterwards the comparison is distributional: posterior means within 4 Monte-Carlo standard errors, step sizes
        and tree depths of the two samplers within 5 %."""
    import ctypes

    from nutpie_amd import _lib
    from nutpie_amd.radon import synthetic_radon_data
    from tests.conftest import FIXTURES
    from tests.test_gpu_parity import oracle_settings

    import subprocess

    src, out = os.path.join(FIXTURES, "radon_host.c"), os.path.join(FIXTURES, "libradon_host.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src, "-lm"], check=True)
    host = ctypes.CDLL(out)
    host.radon_host_create.restype = ctypes.c_void_p
    host.radon_host_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    host.radon_host_free.argtypes = [ctypes.c_void_p]
    data = synthetic_radon_data()
    n = int(data["county_idx"].max()) + 1
    cty = np.ascontiguousarray(data["county_idx"], dtype=np.int32)
    fl, y = np.ascontiguousarray(data["floor"]), np.ascontiguousarray(data["log_radon"])
    D = 2 * n + 3
    hd = radon_device_lib.radon_device_create(n, len(y), cty.ctypes.data, fl.ctypes.data, y.ctypes.data)
    hh = host.radon_host_create(n, len(y), cty.ctypes.data, fl.ctypes.data, y.ctypes.data)
    try:
        chains, tune, draws = 96, 250, 150
        s = _lib.PyNutsSettings.Diag(17)
        s.update(num_tune=tune, num_draws=draws, num_chains=chains)
        smp = _lib.PySampler(s, _lib.NativeDeviceCallbackModel(D, ctypes.cast(radon_device_lib.radon_device_logp, ctypes.c_void_p).value, hd,
                                                              keep_alive=radon_device_lib))
        smp.wait()
        W = smp.waves_per_chain
        got = smp.take_results()
        want = oracle.sample_callback(oracle_settings(oracle, chains=chains, tune=tune, draws=draws, seed=17, W=W),
                                      D, ctypes.cast(host.radon_host_logp, ctypes.c_void_p).value, user=hh)
        # first draws: same trees, floats to rounding
        for k in ("depth", "n_steps", "index_in_trajectory", "diverging"):
            assert np.array_equal(np.asarray(got.stats[k])[:, :2].astype(np.int64), want.stats[k][:, :2].astype(np.int64)), k
        np.testing.assert_allclose(got.draws[:, :2], want.draws[:, :2], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(got.stats["energy"][:, :2], want.stats["energy"][:, :2], rtol=1e-9)
        # the rest: the same posterior and the same adaptation, statistically
        a, b = got.draws[:, tune:], want.draws[:, tune:]
        for idx in (0, n + 1, 2 * n + 2, n, 2 * n + 1, 5, n + 7):
            ma, mb = a[..., idx].mean(1), b[..., idx].mean(1)          # per-chain means: independent between chains
            se = np.sqrt(ma.var(ddof=1) / chains + mb.var(ddof=1) / chains)
            assert abs(ma.mean() - mb.mean()) < 4 * se, (idx, ma.mean(), mb.mean(), se)
        sa, sb = got.stats["step_size"][:, -1].mean(), want.stats["step_size"][:, -1].mean()
        assert abs(sa / sb - 1) < 0.05
        da, db = np.asarray(got.stats["depth"])[:, tune:].mean(), want.stats["depth"][:, tune:].mean()
        assert abs(da / db - 1) < 0.05
        assert got.stats["diverging"][:, tune:].mean() < 0.02 and want.stats["diverging"][:, tune:].mean() < 0.02
    finally:
        radon_device_lib.radon_device_free(hd)
        host.radon_host_free(hh)


def _two_rank_worker(rank, world, port, out_path, backend="nccl", one_gpu=False):
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nutpie_amd import _lib
    from nutpie_amd.distributed import sample_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = 0 if one_gpu else rank
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        diag = np.linspace(0.5, 3.0, 300)

        def make(offset, n_local):
            s = _lib.PyNutsSettings.Diag(8)
            s.update(num_tune=60, num_draws=20, num_chains=10)
            return _lib.PySampler(s, _lib.TridiagGaussianModel(diag), device=dev, chain_offset=offset, n_local_chains=n_local)

        smp, got = sample_sharded(make, 10, thin=2, stats=("n_steps", "depth", "step_size"), device=dev, moments_after=60)
        if rank == 0:
            np.savez(out_path, **{k: (v.cpu().numpy() if hasattr(v, "cpu") else np.asarray(v)) for k, v in got.items()})
        dist.barrier()
        smp.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_rccl_sharding_matches_the_single_gpu_job(tmp_path):
    """a REAL two-rank RCCL run (one process per GPU, chain shards, trace gather over xGMI) equals the
    one-GPU job chain for chain.  Needs two GPUs: skipped on the one-GPU test box, runs where the driver has a node."""
    import socket

    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    from nutpie_amd import _lib

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "two_rank.npz")
    mp.spawn(_two_rank_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    s = _lib.PyNutsSettings.Diag(8)
    s.update(num_tune=60, num_draws=20, num_chains=10)
    smp = _lib.PySampler(s, _lib.TridiagGaussianModel(np.linspace(0.5, 3.0, 300)))
    smp.wait()
    full = smp.take_results()
    assert np.array_equal(got["draws"], full.draws[:, ::2])
    for k in ("n_steps", "depth", "step_size"):
        assert np.array_equal(got[k], np.asarray(full.stats[k])), k
    np.testing.assert_allclose(got["draw_mean"], full.draws[:, 60:].mean(1), rtol=1e-12, atol=1e-14)


def test_three_ranks_on_one_gpu_shard_the_job_chain_for_chain(tmp_path):
    """What the one-GPU test box CAN run of the multi-GPU path with the real engine: three processes (ragged shards of 4 + 3 + 3 chains), each
    with its own sampler on the same device, chains keyed by their GLOBAL id, the trace gathered to rank 0 (gloo: RCCL refuses two ranks on
    one GPU) — equal to the one-process job chain for chain.  The RCCL form of the same run is the test above."""
    import socket

    import torch.multiprocessing as mp

    from nutpie_amd import _lib

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "three_rank.npz")
    mp.spawn(_two_rank_worker, args=(3, port, out, "gloo", True), nprocs=3, join=True)
    got = np.load(out)
    s = _lib.PyNutsSettings.Diag(8)
    s.update(num_tune=60, num_draws=20, num_chains=10)
    smp = _lib.PySampler(s, _lib.TridiagGaussianModel(np.linspace(0.5, 3.0, 300)))
    smp.wait()
    full = smp.take_results()
    assert got["draws"].shape == (10, 40, 300)
    assert np.array_equal(got["draws"], full.draws[:, ::2])
    for k in ("n_steps", "depth", "step_size"):
        assert np.array_equal(got[k], np.asarray(full.stats[k])), k
    np.testing.assert_allclose(got["draw_mean"], full.draws[:, 60:].mean(1), rtol=1e-12, atol=1e-14)


def test_sharded_sampling_and_rccl_gather_single_rank(tmp_path):
    """The multi-GPU path (chain shard + RCCL gather of device-resident trace arrays) with a one-rank NCCL group:
    the same code the 8-GPU job runs, on the one GPU a test box has."""
    import os

    import torch
    import torch.distributed as dist

    from nutpie_amd import _lib
    from nutpie_amd.distributed import sample_sharded

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        diag = np.linspace(0.5, 2.0, 40)

        def make(offset, n_local):
            s = _lib.PyNutsSettings.Diag(5)
            s.update(num_tune=40, num_draws=20, num_chains=6)
            return _lib.PySampler(s, _lib.TridiagGaussianModel(diag), chain_offset=offset, n_local_chains=n_local)

        smp, got = sample_sharded(make, 6, thin=2, dims=[0, 3, 39], moments_after=40)
        ref = smp.take_results()
        assert got["draws"].shape == (6, 30, 3)
        # per-chain moments of the post-warm-up draws, reduced on the device
        np.testing.assert_allclose(got["draw_mean"].cpu().numpy(), ref.draws[:, 40:].mean(1), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(got["draw_var"].cpu().numpy(), ref.draws[:, 40:].var(1, ddof=1), rtol=1e-10)
        # summary only: no draws cross the wire
        smp2, got2 = sample_sharded(make, 6, gather_draws=False, moments_after=40)
        assert "draws" not in got2 and np.array_equal(got2["draw_mean"].cpu().numpy(), got["draw_mean"].cpu().numpy())
        smp2.close()
        assert np.array_equal(got["draws"].cpu().numpy(), ref.draws[:, ::2][:, :, [0, 3, 39]])
        assert np.array_equal(got["n_steps"].cpu().numpy(), ref.stats["n_steps"])
        assert np.array_equal(got["diverging"].cpu().numpy().astype(bool), ref.stats["diverging"])
    finally:
        dist.destroy_process_group()


def test_pause_and_resume_hook(hip):
    """The host-driven adaptation hook of the C-ABI (nphip_settings_set_pause_draws / nphip_sampler_waiting / _resume_at): every
    chain stops after exactly the listed number of draws; resumed at a new position it re-runs the initial-point sequence there
    and goes on with its next draw."""
    diag = np.linspace(0.5, 2.0, 12)
    s = hip.PyNutsSettings.Diag(5)
    s.update(num_tune=40, num_draws=20, num_chains=6)
    s.set_pause_draws([15, 30])
    smp = hip.PySampler(s, hip.TridiagGaussianModel(diag), manual=True, evals_per_launch=50)
    for _ in range(200):
        done, _, _ = smp.step(1)
        if smp.waiting().all():
            break
    assert not done and smp.waiting_codes().tolist() == [1] * 6
    assert [p.finished_draws for p in smp.progress()] == [15] * 6
    smp.step(3)                                                   # waiting chains do not move
    assert [p.finished_draws for p in smp.progress()] == [15] * 6
    pos = np.tile(np.linspace(-1, 1, 12), (6, 1)) * np.arange(1, 7)[:, None]
    smp.resume_at(np.arange(6), pos)
    for _ in range(400):
        done, _, _ = smp.step(1)
        if smp.waiting().all():
            break
    assert [p.finished_draws for p in smp.progress()] == [30] * 6
    smp.resume_at(np.arange(3), pos[:3])                          # a subset; the others keep waiting
    for _ in range(400):
        done, _, _ = smp.step(1)
        c = smp.waiting_codes()
        if (c[:3] == 2).all():
            break
    assert c.tolist() == [2, 2, 2, 1, 1, 1]
    smp.resume_at(np.arange(3, 6), pos[3:])
    while not smp.step(10)[0]:
        pass
    tr = smp.take_results()
    assert tr.finished.tolist() == [60] * 6
    # draw 15 (the first after the first resume) starts from the supplied position: its tree is rooted there, so with a tiny
    # trajectory it cannot be far; more telling: the draws are finite and the run is reproducible
    assert np.isfinite(tr.draws).all() and np.isfinite(tr.stats["energy"]).all()
    lp_at_pos = -0.5 * (pos**2 * diag).sum(1)
    e15 = tr.stats["energy"][:, 15] - tr.stats["energy_error"][:, 15]          # initial energy of draw 15 = K0 - logp(pos)
    assert (e15 >= -lp_at_pos - 1e-9).all()                                     # kinetic energy is non-negative
    with pytest.raises(RuntimeError, match="manual"):
        s2 = hip.PyNutsSettings.Diag(5)
        s2.update(num_tune=10, num_draws=5, num_chains=2)
        auto = hip.PySampler(s2, hip.TridiagGaussianModel(diag))
        auto.wait()
        auto.resume_at([0], pos[:1])


def test_resume_at_a_position_that_does_not_evaluate_fails_the_chain(hip):
    """a chain resumed by the host is never restarted from a random point behind the host's back: the position either evaluates
    or the chain ends with an error that names the cause"""
    s = hip.PyNutsSettings.Diag(5)
    s.update(num_tune=40, num_draws=20, num_chains=4)
    s.set_pause_draws([10])
    smp = hip.PySampler(s, hip.TridiagGaussianModel(np.ones(5)), manual=True, evals_per_launch=50)
    for _ in range(200):
        smp.step(1)
        if smp.waiting().all():
            break
    pos = np.zeros((4, 5))
    pos[2, 3] = np.nan
    smp.resume_at(np.arange(4), pos)
    with pytest.raises(RuntimeError, match="resume_at does not evaluate"):
        for _ in range(400):
            if smp.step(10)[0]:
                break
        smp.take_results()


def test_low_rank_adaptation_on_a_correlated_gaussian():
    """adaptation="low_rank" (reference docs/sampling-options.qmd:124-144): a 60-dimensional Gaussian with three strong
    correlated directions on top of heterogeneous scales.  A diagonal metric cannot undo the rotation and needs long
    trajectories; the low-rank metric finds the directions during warm-up: same posterior, several times fewer leapfrogs."""
    import torch

    D = 60
    rng = np.random.default_rng(11)
    scales = np.exp(rng.normal(size=D))
    B = rng.normal(size=(D, 3))
    Sigma = np.diag(scales**2) + 400.0 * (scales[:, None] * B) @ (scales[:, None] * B).T
    mu = rng.normal(size=D) * 3
    P = np.linalg.inv(Sigma)

    def make_logp():
        Pt, mt = torch.as_tensor(P, device="cuda"), torch.as_tensor(mu, device="cuda")

        def f(x):
            z = x - mt
            g = -(z @ Pt)
            return 0.5 * (z * g).sum(-1), g

        return f

    m = nutpie_amd.from_torchfunc(D, make_logp)
    kw = dict(chains=128, tune=500, draws=300, seed=3, progress_bar=False)
    lr = nutpie_amd.sample(m, adaptation="low_rank", **kw)
    dg = nutpie_amd.sample(m, adaptation="diag", **kw)
    assert lr.posterior.x.shape == (128, 300, D) and lr.warmup_posterior.x.shape == (128, 500, D)
    steps_lr, steps_dg = lr.sample_stats.n_steps.values.mean(), dg.sample_stats.n_steps.values.mean()
    assert steps_lr * 2.5 < steps_dg, (steps_lr, steps_dg)      # measured: 8 against 60+ leapfrogs per draw
    assert lr.sample_stats.diverging.values.mean() < 0.01
    x = lr.posterior.x.values.reshape(-1, D)
    sd = np.sqrt(np.diag(Sigma))
    assert np.abs((x.mean(0) - mu) / sd).max() < 0.06                  # 38 400 draws
    emp = np.cov(x.T)
    assert np.abs(np.sqrt(np.diag(emp)) / sd - 1).max() < 0.06
    corr = emp / np.sqrt(np.outer(np.diag(emp), np.diag(emp)))
    true_corr = Sigma / np.outer(sd, sd)
    assert np.abs(corr - true_corr).max() < 0.06
    # the warm-up part of the trace is in model coordinates too (rewritten at every window switch): late warm-up draws look
    # like the posterior
    late = lr.warmup_posterior.x.values[:, 400:].reshape(-1, D)
    assert np.abs((late.mean(0) - mu) / sd).max() < 0.2
    assert "gradient" not in lr.sample_stats                       # the estimator's gradients stay internal
    assert lr.sample_stats.attrs["inference_library_settings"].count('"adaptation": "low_rank"') == 1



def hip_settings(kw):
    from nutpie_amd import _lib

    s = _lib.PyNutsSettings.Diag(kw["seed"])
    s.update(num_tune=kw["tune"], num_draws=kw["draws"], num_chains=kw["chains"])
    return s


def test_device_callback_in_groups_draws_the_same():
    """Engine option ``host_groups`` for a batched device callback (round 5): the chains in groups, each group's engine kernel and
    callback on a stream of its own so that one group's callback overlaps the other's kernel.  A callback whose rows do not depend on
    what else is in the batch gives the same trace, bit for bit, as one launch and one callback for all chains."""
    import torch

    sd = torch.exp(torch.linspace(-1.0, 1.0, 37, dtype=torch.float64, device="cuda"))

    def make_logp():
        def f(x):
            z = x / sd
            return -0.5 * (z * z).sum(-1), -z / sd

        return f

    m = nutpie_amd.from_torchfunc(37, make_logp)
    kw = dict(chains=96, tune=120, draws=60, seed=4, progress_bar=False)
    a = nutpie_amd.sample(m, **kw)
    for groups in (2, 3):
        b = nutpie_amd.sample(m, host_groups=groups, **kw)
        assert np.array_equal(a.posterior.x.values, b.posterior.x.values)
        assert np.array_equal(a.sample_stats.n_steps.values, b.sample_stats.n_steps.values)
    # ... and with the groups as parallel branches of one captured HIP graph (graph_steps): the same trace again
    s = hip_settings(kw)
    smp = m._make_sampler(s, None, 1, None, None, None, None, host_groups=2, graph_steps=8)
    smp.wait()
    got = smp.take_results()
    assert np.array_equal(np.asarray(got.draws)[:, 120:], a.posterior.x.values)
    np.testing.assert_allclose(a.posterior.x.values.std((0, 1)), sd.cpu().numpy(), rtol=0.12)
