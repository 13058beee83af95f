"""Parity of the HIP engine with the CPU oracle, through the C-ABI, on a real MI355X.

Bar (north star): tree depth, n_steps, divergences, index_in_trajectory and the tuning flag bit-identical;
draws, energies and step sizes within a stated fp64 tolerance.  Because host and device share one numerics
contract (include/nphip_spec.h: counter-based RNG, fma-only elementary functions, fixed summation order) the
tolerance used for the fused analytic models is ZERO: every float is compared bit-for-bit.  Models whose
gradient comes from another library (torch GEMM) use rtol = 1e-9 on floats and equality on integers is then
not required (documented in the test).
"""
import ctypes
import os

import numpy as np
import pytest

from nutpie_amd.gaussian import ar1_gaussian
from tests.conftest import GOLDEN, assert_trace_equal, fn_addr

pytestmark = pytest.mark.gpu


def run_engine(hip, model, *, chains, tune, draws, seed, waves=0, init=None, launch=None, info=None, **settings):
    s = hip.PyNutsSettings.Diag(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, **settings)
    if init is not None:
        model.set_init(*init) if isinstance(init, tuple) else model.set_init(init)
    smp = hip.PySampler(s, model, waves_per_chain=waves, **(launch or {}))
    smp.wait()
    W = smp.waves_per_chain
    if info is not None:
        info["host_mode"] = smp.host_mode
    return smp.take_results(), W


ORACLE_KEYS = {"use_grad_based_mass_matrix": "use_grad_based_mass_matrix", "max_energy_error": "max_energy_error", "maxdepth": "maxdepth",
               "mindepth": "mindepth", "check_turning": "check_turning", "target_accept": "target_accept", "initial_step": "initial_step",
               "step_size_jitter": "step_size_jitter", "window_switch_freq": "mass_matrix_switch_freq",
               "early_window_switch_freq": "early_mass_matrix_switch_freq", "max_step_size": "max_step_size",
               "adapt_mass_matrix": "adapt_mass_matrix", "num_try_init": "num_try_init"}


def oracle_settings(oracle, *, chains, tune, draws, seed, W, init_kind=0, chain_offset=0, **settings):
    kw = {}
    for k, v in settings.items():
        if k == "step_size_adapt_method" and v == "adam":
            kw["adam"] = 1
        elif k == "step_size_adam_learning_rate":
            kw["adam_learning_rate"] = float(v)
        elif k == "step_size_adapt_method":
            kw["fixed_step_size"] = 1
            kw["initial_step"] = float(v)
        elif k in ("store_gradient", "store_mass_matrix", "store_divergences"):
            kw[k] = int(v)
        elif k == "store_unconstrained":
            continue
        else:
            kw[ORACLE_KEYS[k]] = int(v) if isinstance(v, bool) else v
    return oracle.default_settings(seed=seed, num_chains=chains, num_tune=tune, num_draws=draws, n_threads=8, waves_per_chain=W,
                                   init_kind=init_kind, chain_offset=chain_offset, **kw)


# ------------------------------------------------------------------------------------ numerics contract
def test_device_detmath_is_bit_identical_to_the_oracle(hip, oracle):
    rng = np.random.default_rng(0)
    cases = [("exp", rng.uniform(-745, 710, 300000)), ("exp", rng.uniform(-3, 3, 100000)),
             ("log", np.exp(rng.uniform(-700, 700, 300000))), ("log", 1 + rng.uniform(-1e-3, 1e-3, 100000)),
             ("log", np.array([5e-324, 1e-310, 2.2250738585072014e-308, 1.0, np.inf, 0.0])),
             ("log1p", rng.uniform(0, 1, 200000)), ("sin2pi", rng.uniform(0, 1, 200000)), ("cos2pi", rng.uniform(0, 1, 200000))]
    for fn, x in cases:
        a, b = hip.test_detmath(fn, x), oracle.detmath(fn, x)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), fn


def test_device_sqrt_and_division_are_correctly_rounded(hip):
    rng = np.random.default_rng(1)
    x = np.exp(rng.uniform(-600, 600, 500000))
    assert np.array_equal(hip.test_detmath("sqrt", x), np.sqrt(x))
    assert np.array_equal(hip.test_detmath("recip", x), 1.0 / x)


def test_device_normals_and_dot(hip, oracle):
    for n in (1, 2, 7, 1000, 4097):
        assert np.array_equal(hip.test_normals(123, 5, 7, 1, n), oracle.normals(123, 5, 7, 1, n))
    rng = np.random.default_rng(2)
    for W in (1, 2, 4, 8, 16):
        for n in (1, 63, 128, 129, 1000, 10000, 40001):
            x, y = rng.normal(size=n), rng.normal(size=n)
            assert hip.test_dot(x, y, W) == oracle.dot(x, y, W), (W, n)


# ------------------------------------------------------------------------------------ sampler parity
def test_config1_std_normal_bit_identical(hip, oracle):
    # BASELINE.json config 1 through the GPU engine (the reference runs it on the CPU)
    got, W = run_engine(hip, hip.TridiagGaussianModel(np.ones(10)), chains=4, tune=400, draws=1000, seed=123)
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=4, tune=400, draws=1000, seed=123, W=W), np.ones(10))
    assert_trace_equal(got, want)
    assert got.finished.tolist() == [1400] * 4


@pytest.mark.parametrize("name", ["stdnormal_d10", "ar1_d257", "diag_d1000_w2", "divergent_d3", "maxdepth3_d64"])
def test_engine_matches_committed_golden_vectors(hip, name):
    # golden vectors produced by the repo's CPU oracle (tests/golden/make_golden.py); no oracle run needed here
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    kw, mspec = mg.CASES[name]
    kw = dict(kw)
    margs = mg.model_args(mspec)
    W = kw.pop("waves_per_chain", 1)
    extra = {k: kw.pop(k) for k in list(kw) if k in ("maxdepth", "max_energy_error")}
    got, _ = run_engine(hip, hip.TridiagGaussianModel(margs["diag"], margs.get("offdiag"), margs.get("mu")), chains=kw["num_chains"],
                        tune=kw["num_tune"], draws=kw["num_draws"], seed=kw["seed"], waves=W, **extra)
    gold = np.load(os.path.join(GOLDEN, f"oracle_{name}.npz"))
    for k in ("depth", "n_steps", "index_in_trajectory", "diverging", "maxdepth_reached", "tuning"):
        assert np.array_equal(np.asarray(got.stats[k]).astype(gold[k].dtype), gold[k]), k
    for k in ("energy", "logp", "step_size", "step_size_bar", "mean_tree_accept"):
        assert np.array_equal(got.stats[k], gold[k]), k
    assert np.array_equal(got.draws[:, ::10, :8], gold["draws_thin"])


@pytest.mark.parametrize("dim,waves", [(1, 1), (2, 1), (127, 1), (128, 1), (129, 2), (1000, 1), (1000, 4), (2500, 2), (5003, 8), (9000, 16),
                                       # one wave per chain with 2 .. 7 chunks per lane (with 8: the family whose draw end changed at the end of
                                       # round 6 — the call behind the loop of leaves, read-ahead passes, no inter-procedural allocation)
                                       (200, 1), (380, 1), (500, 1), (640, 1), (700, 1), (896, 1),
                                       # register-resident kernels with several waves per chain
                                       (1100, 2), (1500, 2), (2048, 2), (2500, 4), (3300, 4), (4096, 4),
                                       (256, 2), (700, 2), (1000, 2), (512, 4), (900, 4), (1536, 4),
                                       # lean register-resident kernels (8 waves per chain, 1..10 chunks per wave); (10000, 8) is
                                       # the default geometry of BASELINE.json config 5
                                       (900, 8), (2000, 8), (4200, 8), (6000, 8), (7100, 8), (8192, 8), (10000, 8), (10240, 8),
                                       # lean kernels with 4 waves per chain (state in VGPRs + AGPRs), 9..20 chunks per wave
                                       (4200, 4), (5000, 4), (7100, 4), (9000, 4), (10000, 4), (10240, 4),
                                       # ... 21..24 chunks per wave (round 6: the state spills; the default geometry up to D = 12288), and the
                                       # memory-resident kernels beyond
                                       (10300, 4), (11264, 4), (12000, 4), (12288, 4), (12289, 8)])
def test_correlated_gaussian_all_geometries(hip, oracle, dim, waves):
    rng = np.random.default_rng(dim)
    sd = np.exp(0.7 * rng.normal(size=dim))
    rho = 0.8
    c = 1 / (1 - rho * rho)
    d = np.full(dim, (1 + rho * rho) * c); d[0] = d[-1] = c
    if dim == 1:
        d[:] = 1.0
    diag = d / sd**2
    off = -rho * c / (sd[:-1] * sd[1:]) if dim > 1 else None
    mu = rng.normal(size=dim)
    chains, tune, draws = (8, 120, 40) if dim <= 1000 else (4, 40, 10)
    got, W = run_engine(hip, hip.TridiagGaussianModel(diag, off, mu), chains=chains, tune=tune, draws=draws, seed=dim + 1, waves=waves,
                        store_gradient=True, store_mass_matrix=True)
    assert W == waves
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=chains, tune=tune, draws=draws, seed=dim + 1, W=W, store_gradient=True, store_mass_matrix=True), diag, off, mu)
    assert_trace_equal(got, want)
    assert np.array_equal(got.stats["gradient"], want.stats["gradient"])
    assert np.array_equal(got.stats["mass_matrix_inv"], want.stats["mass_matrix_inv"])


@pytest.mark.parametrize("dim", [200, 380])
def test_small_kernels_with_and_without_the_register_cap_draw_the_same(hip, oracle, dim):
    # 2 / 3 chunks per lane: a job of up to 1024 chains (at most one wave per SIMD) runs the kernel built without the two-waves-per-SIMD
    # register cap (k_advance<..., WIDE>), a larger one the capped kernel: chains are keyed by their global id, so the first chains
    # of a 1100-chain job are the chains of a small one — and those are the oracle's
    model = ar1_gaussian(dim)
    m = hip.TridiagGaussianModel(model.diag, model.offdiag)
    big, W = run_engine(hip, m, chains=1100, tune=40, draws=10, seed=dim)
    small, _ = run_engine(hip, m, chains=24, tune=40, draws=10, seed=dim)
    assert W == 1
    assert np.array_equal(big.draws[:24], small.draws) and np.array_equal(big.draws[-1].shape, small.draws[0].shape)
    for k in ("n_steps", "energy", "step_size", "diverging"):
        assert np.array_equal(np.asarray(big.stats[k])[:24], np.asarray(small.stats[k])), k
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=24, tune=40, draws=10, seed=dim, W=1), model.diag, model.offdiag)
    assert_trace_equal(small, want)


@pytest.mark.parametrize("dim", [5, 300, 1000])
def test_streaming_kernel_single_wave_bit_identical(hip, oracle, dim):
    # the memory-resident (streaming) fused kernel with one wave per chain — the family used for D > 1024,
    # store_divergences and callbacks — forced here where the register-resident kernel would normally run
    rng = np.random.default_rng(dim + 7)
    diag = np.exp(rng.normal(size=dim))
    off = 0.2 * rng.normal(size=dim - 1) * np.sqrt(diag[:-1] * diag[1:])
    got, W = run_engine(hip, hip.TridiagGaussianModel(diag, off), chains=6, tune=100, draws=40, seed=31, launch=dict(no_register_kernel=True))
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=6, tune=100, draws=40, seed=31, W=W), diag, off)
    assert W == 1
    assert_trace_equal(got, want)


@pytest.mark.parametrize("dim,waves", [(130, 1), (700, 1), (1024, 1), (2000, 2), (5000, 8), (9000, 16)])
def test_stream_cache_is_transparent(hip, oracle, dim, waves):
    # memory-resident fused kernels keep the cursor's (sigma^2, grad, p, rho) in VGPRs between leaves (W = 1; every
    # store still happens) or sigma^2 in LDS (W >= 8): same trace as the kernel that reloads everything, and as the oracle
    model = ar1_gaussian(dim)
    kw = dict(chains=4, tune=30, draws=8, seed=dim, waves=waves)
    m = hip.TridiagGaussianModel(model.diag, model.offdiag)
    cached, W = run_engine(hip, m, launch=dict(no_register_kernel=True), **kw)
    plain, _ = run_engine(hip, m, launch=dict(no_register_kernel=True, no_stream_cache=True), **kw)
    assert W == waves
    assert_trace_equal(cached, plain)
    if dim <= 2000:
        want = oracle.sample_tridiag(oracle_settings(oracle, chains=4, tune=30, draws=8, seed=dim, W=W), model.diag, model.offdiag)
        assert_trace_equal(cached, want)


@pytest.mark.parametrize("settings", [
    dict(use_grad_based_mass_matrix=False),                 # adaptation="draw_diag"
    dict(max_energy_error=0.2),                             # many divergences
    dict(maxdepth=2),                                       # maxdepth_reached path
    dict(maxdepth=12, target_accept=0.99),                  # deep trees
    dict(mindepth=3),
    dict(check_turning=False, maxdepth=4),
    dict(step_size_jitter=0.3),
    dict(step_size_adapt_method="0.3"),                     # fixed step size
    dict(step_size_adapt_method="adam"),                    # Adam on log(step size)
    dict(step_size_adapt_method="adam", step_size_adam_learning_rate=0.1, target_accept=0.9),
    dict(window_switch_freq=20, early_window_switch_freq=4),
    dict(adapt_mass_matrix=False),
    dict(max_step_size=0.2),
    dict(initial_step=5.0),                                 # search goes downwards
    dict(initial_step=1e-4),                                # search goes upwards
])
def test_settings_variants_bit_identical(hip, oracle, settings):
    rng = np.random.default_rng(5)
    diag = np.exp(rng.normal(size=24) * 1.5)
    got, W = run_engine(hip, hip.TridiagGaussianModel(diag), chains=6, tune=150, draws=60, seed=99, **settings)
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=6, tune=150, draws=60, seed=99, W=W, **settings), diag)
    assert_trace_equal(got, want)
    if "max_energy_error" in settings:
        assert got.stats["diverging"].sum() > 20
    if settings.get("maxdepth") == 2:
        assert got.stats["maxdepth_reached"].sum() > 0 and got.stats["depth"].max() == 2


@pytest.mark.parametrize("dim,settings,launch", [
    (5000, dict(max_energy_error=0.5), {}),                              # divergences end draws from inside leaf_lean
    (5000, dict(maxdepth=3), dict(evals_per_launch=5)),                  # flush / reload of the registers at launch ends
    (6100, dict(maxdepth=4), dict(evals_per_launch=3)),                  # (a launch boundary between a leaf = 3 mod 4 and the next)
    (9000, dict(step_size_jitter=0.2, mindepth=2), dict(evals_per_launch=13)),
    (4500, dict(use_grad_based_mass_matrix=False, store_gradient=True), {}),
    (4500, dict(check_turning=False, maxdepth=4), {}),
    (7000, dict(maxdepth=12, target_accept=0.95), {}),                   # deep trees: merges up to level >= 5
])
@pytest.mark.parametrize("waves", [0, 8])
def test_lean_register_kernels_variants(hip, oracle, dim, settings, launch, waves):
    # lean register-resident kernels (4096 < D <= 10240: 4 waves per chain by default, 8 on request) under the awkward settings
    model = ar1_gaussian(dim)
    kw = dict(chains=3, tune=50, draws=12, seed=dim + 3)
    got, W = run_engine(hip, hip.TridiagGaussianModel(model.diag, model.offdiag), launch=launch, waves=waves, **kw, **settings)
    assert W == (waves or 4)
    want = oracle.sample_tridiag(oracle_settings(oracle, W=W, **kw, **settings), model.diag, model.offdiag)
    assert_trace_equal(got, want)
    if "store_gradient" in settings:
        assert np.array_equal(got.stats["gradient"], want.stats["gradient"])


DIV_KEYS = ("divergence_start", "divergence_end", "divergence_momentum", "divergence_start_gradient")


@pytest.mark.parametrize("dim,waves,launch", [
    (3, 0, {}),                                          # register kernel, one chunk
    (24, 0, dict(evals_per_launch=7)),                   # ... with launch boundaries inside doublings
    (1000, 0, {}),                                       # the headline kernel (8 chunks per lane)
    (1000, 0, dict(no_register_kernel=True)),            # memory-resident, cursor cached in VGPRs (keeps no gradient in a tree)
    (300, 0, dict(no_register_kernel=True, no_stream_cache=True)),
    (1300, 2, {}),                                       # register kernels with 2 / 4 waves per chain
    (2600, 4, dict(evals_per_launch=9)),
    (5000, 0, {}),                                       # lean register kernels, 4 / 8 waves per chain
    (5000, 8, dict(evals_per_launch=11)),
    (11000, 0, {}),                                      # lean, 22 chunks per wave (spilling build)
    (13000, 0, {}),                                      # memory-resident, 8 waves per chain
])
def test_divergence_records_bit_identical(hip, oracle, dim, waves, launch):
    # store_divergences (python/nutpie/sample.py:631-650, tests/test_pymc.py:303-349): the state a failed leapfrog started from
    # (position, momentum, gradient) and the position it ended at, every float against the oracle's — in every kernel family.
    # The register kernels rebuild the pre-step state in the rare path (kernels.hip: replay_divergence).
    model = ar1_gaussian(dim) if dim > 24 else None
    args = (model.diag, model.offdiag) if model else (np.exp(np.random.default_rng(dim).normal(size=dim) * 1.5),)
    kw = dict(chains=4 if dim <= 2600 else 3, tune=60 if dim <= 2600 else 40, draws=15 if dim <= 2600 else 8, seed=dim + 17)
    settings = dict(store_divergences=True, max_energy_error=0.3 if dim <= 24 else 0.6)
    got, W = run_engine(hip, hip.TridiagGaussianModel(*args), launch=launch, waves=waves, **kw, **settings)
    want = oracle.sample_tridiag(oracle_settings(oracle, W=W, **kw, **settings), *args)
    assert_trace_equal(got, want)
    div = np.asarray(got.stats["diverging"]).astype(bool)
    assert div.sum() >= 5, "the case is meant to diverge"
    if dim > 3:
        assert np.asarray(got.stats["depth"])[div].max() >= 2, "divergences deep inside a doubling are the point (leaves replayed)"
    for k in DIV_KEYS:
        a, b = got.stats[k], want.stats[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), k
        assert np.all(np.isnan(a[~div])), k
    assert np.all(np.isfinite(got.stats["divergence_start"][div]))


def test_divergence_records_host_callback_and_logp_errors(hip, oracle, fixture_lib):
    # host C callback (launch per evaluation keeps the pre-step state in memory); a recoverable logp error diverges WITHOUT an
    # end position (src/pymc.rs:166-180: code > 0): divergence_end stays NaN there, the other three are set
    fn = fn_addr(fixture_lib.eight_schools_logp)
    kw = dict(chains=8, tune=80, draws=40, seed=31)
    settings = dict(store_divergences=True, max_energy_error=0.8)
    got, W = run_engine(hip, hip.HostCallbackModel(10, fn), **kw, **settings)
    want = oracle.sample_callback(oracle_settings(oracle, W=W, **kw, **settings), 10, fn)
    assert_trace_equal(got, want)
    assert got.stats["diverging"].sum() >= 5
    for k in DIV_KEYS:
        assert np.array_equal(got.stats[k], want.stats[k], equal_nan=True), k

    def wall(x):   # a density with a wall: beyond it the callback reports a recoverable error
        if x[0] > 1.0:
            raise RecoverableError()
        return -0.5 * float(x @ x), -x

    class RecoverableError(Exception):
        is_recoverable = True

    kw = dict(chains=3, tune=40, draws=25, seed=8)
    got, W = run_engine(hip, hip.HostCallbackModel(3, wall), store_divergences=True, **kw)
    want = oracle.sample_callback(oracle_settings(oracle, W=W, store_divergences=True, **kw), 3, wall)
    assert_trace_equal(got, want)
    div = np.asarray(got.stats["diverging"]).astype(bool)
    assert div.sum() >= 3
    no_end = np.isnan(got.stats["divergence_end"][div]).all(-1)
    assert no_end.sum() >= 3 and np.all(np.isfinite(got.stats["divergence_start"][div]))   # (an energy error far out has an end)
    for k in DIV_KEYS:
        assert np.array_equal(got.stats[k], want.stats[k], equal_nan=True), k


def test_lean_kernel_equals_streaming_kernel(hip):
    # same chains through the lean register kernel and through the memory-resident kernel of the same geometry (W = 8)
    model = ar1_gaussian(10000)
    kw = dict(chains=6, tune=40, draws=10, seed=77)
    m = hip.TridiagGaussianModel(model.diag, model.offdiag)
    a, W = run_engine(hip, m, **kw)
    b, _ = run_engine(hip, m, launch=dict(no_register_kernel=True), **kw)
    assert W == 4
    assert np.array_equal(a.draws, b.draws)
    for k in ("depth", "n_steps", "index_in_trajectory", "energy", "step_size", "mean_tree_accept"):
        assert np.array_equal(a.stats[k], b.stats[k]), k


@pytest.mark.parametrize("dim,settings,launch", [
    (1300, dict(max_energy_error=0.5), {}),                              # divergences end draws from inside leaf_reg
    (1300, dict(maxdepth=3), dict(evals_per_launch=5)),                  # flush / reload of registers and ring at launch ends
    (2600, dict(step_size_jitter=0.2, mindepth=2), dict(evals_per_launch=13)),
    (2600, dict(use_grad_based_mass_matrix=False, store_gradient=True), {}),
])
def test_multiwave_register_kernels_variants(hip, oracle, dim, settings, launch):
    # register-resident kernels with 2 (D = 1300) and 4 (D = 2600) waves per chain under the awkward settings
    model = ar1_gaussian(dim)
    kw = dict(chains=3, tune=50, draws=12, seed=dim + 3)
    got, W = run_engine(hip, hip.TridiagGaussianModel(model.diag, model.offdiag), launch=launch, waves=2 if dim <= 2048 else 4, **kw, **settings)
    assert W == (2 if dim <= 2048 else 4)
    want = oracle.sample_tridiag(oracle_settings(oracle, W=W, **kw, **settings), model.diag, model.offdiag)
    assert_trace_equal(got, want)
    if "store_gradient" in settings:
        assert np.array_equal(got.stats["gradient"], want.stats["gradient"])


def test_four_waves_per_chain_for_small_batches(hip, oracle):
    # fewer chains than SIMDs: waves_per_chain = 4 (register-resident, +20 % at 64..256 chains) is the caller's choice,
    # never the engine's: by default a chain's floats do not depend on how many chains run with it
    model = ar1_gaussian(1000)
    kw = dict(chains=6, tune=60, draws=20, seed=12)
    got, W = run_engine(hip, hip.TridiagGaussianModel(model.diag, model.offdiag), waves=4, **kw)
    assert W == 4
    want = oracle.sample_tridiag(oracle_settings(oracle, W=W, **kw), model.diag, model.offdiag)
    assert_trace_equal(got, want)
    _, Wd = run_engine(hip, hip.TridiagGaussianModel(model.diag, model.offdiag), chains=6, tune=3, draws=2, seed=12)
    assert Wd == 1


def test_init_strategies(hip, oracle):
    diag = np.array([1.0, 4.0, 0.25])
    # N(0,1) initial points: src/stan.rs:798-808
    got, W = run_engine(hip, hip.TridiagGaussianModel(diag), chains=4, tune=30, draws=10, seed=3, init="normal")
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=4, tune=30, draws=10, seed=3, W=W, init_kind=1), diag)
    assert_trace_equal(got, want)
    # explicit initial points: src/pymc.rs:505-534
    pts = np.random.default_rng(0).normal(size=(4, 3))
    got, W = run_engine(hip, hip.TridiagGaussianModel(diag), chains=4, tune=30, draws=10, seed=3, init=("explicit", pts))
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=4, tune=30, draws=10, seed=3, W=W, init_kind=2), diag, init_points=pts)
    assert_trace_equal(got, want)


def test_chain_sharding_is_invariant(hip, oracle):
    # two "GPUs" = two samplers with chain offsets; concatenation equals the single job
    diag = np.linspace(0.5, 3.0, 40)
    full, W = run_engine(hip, hip.TridiagGaussianModel(diag), chains=10, tune=60, draws=20, seed=8)
    lo, _ = run_engine(hip, hip.TridiagGaussianModel(diag), chains=10, tune=60, draws=20, seed=8, launch=dict(chain_offset=0, n_local_chains=6))
    hi, _ = run_engine(hip, hip.TridiagGaussianModel(diag), chains=10, tune=60, draws=20, seed=8, launch=dict(chain_offset=6, n_local_chains=4))
    assert np.array_equal(full.draws, np.concatenate([lo.draws, hi.draws]))
    for k in ("depth", "n_steps", "step_size"):
        assert np.array_equal(full.stats[k], np.concatenate([lo.stats[k], hi.stats[k]]))


def test_result_does_not_depend_on_launch_slicing(hip):
    diag = np.exp(np.random.default_rng(1).normal(size=300))
    a, _ = run_engine(hip, hip.TridiagGaussianModel(diag), chains=8, tune=50, draws=20, seed=4, launch=dict(evals_per_launch=1))
    b, _ = run_engine(hip, hip.TridiagGaussianModel(diag), chains=8, tune=50, draws=20, seed=4, launch=dict(evals_per_launch=7))
    c, _ = run_engine(hip, hip.TridiagGaussianModel(diag), chains=8, tune=50, draws=20, seed=4, launch=dict(evals_per_launch=100000))
    for x in (b, c):
        assert np.array_equal(a.draws, x.draws) and np.array_equal(a.stats["n_steps"], x.stats["n_steps"])


# ------------------------------------------------------------------------------------ callback flavours
def test_host_callback_eight_schools_bit_identical(hip, oracle, fixture_lib):
    # BASELINE.json config 4 at test size: raw C callback (src/pymc.rs:23-29 signature), pinned D2H/H2D per step
    addr = fn_addr(fixture_lib.eight_schools_logp)
    got, W = run_engine(hip, hip.HostCallbackModel(10, addr, n_threads=4), chains=16, tune=200, draws=100, seed=21, init="normal")
    want = oracle.sample_callback(oracle_settings(oracle, chains=16, tune=200, draws=100, seed=21, W=W, init_kind=1), 10, addr)
    assert_trace_equal(got, want)
    # sanity of the posterior: mu around 4.4, tau positive
    mu = got.draws[:, 200:, 0]
    assert 2.0 < mu.mean() < 7.0


@pytest.mark.parametrize("persist", [0, 3, 1, -37], ids=["resident", "resident-3-evals-per-launch", "launch-per-eval", "fall-back-after-37"])
@pytest.mark.parametrize("groups", [0, 3], ids=["default-groups", "3-groups"])
def test_host_callback_launch_modes_are_bit_identical(hip, oracle, fixture_lib, persist, groups):
    """The three ways a host-callback job can run — resident launches (the chain state stays in registers across the
    evaluations, kernels.hip: REMOTE), one launch per evaluation, and a resident job that falls back mid-way the way a failed
    roll call makes it — give the oracle's trace bit for bit, with launch boundaries every 3 evaluations as well as every 256."""
    addr = fn_addr(fixture_lib.eight_schools_logp)
    info = {}
    got, W = run_engine(hip, hip.HostCallbackModel(10, addr), chains=40, tune=120, draws=60, seed=5, init="normal",
                        launch=dict(host_persist=persist, host_groups=groups), store_gradient=True, info=info)
    assert info["host_mode"] == {0: "resident", 3: "resident", 1: "groups", -37: "fell-back"}[persist]   # (the path meant is the path taken)
    want = oracle.sample_callback(oracle_settings(oracle, chains=40, tune=120, draws=60, seed=5, W=W, init_kind=1, store_gradient=True), 10, addr)
    assert_trace_equal(got, want)
    assert np.array_equal(got.stats["gradient"], want.stats["gradient"])


@pytest.mark.parametrize("dim", [130, 700, 1024, 1100, 2048, 2500, 4096])
def test_resident_host_callback_wider_rows(hip, oracle, dim):
    """Resident launches with 2, 6 and 8 chunks of registers per chain, and — above 1024 dimensions — with two and four waves
    per chain (Python callable, two groups of chains)."""
    sd = np.exp(np.random.default_rng(dim).normal(size=dim) * 0.7)

    def logp(x):
        z = x / sd
        return -0.5 * float(z @ z), -z / sd

    info = {}
    got, W = run_engine(hip, hip.HostCallbackModel(dim, logp), chains=6, tune=50, draws=25, seed=dim, launch=dict(host_groups=2, host_persist=256), info=info)
    assert info["host_mode"] == "resident"
    want = oracle.sample_callback(oracle_settings(oracle, chains=6, tune=50, draws=25, seed=dim, W=W), dim, logp)
    assert_trace_equal(got, want)


def test_two_resident_jobs_at_once(hip, oracle, fixture_lib):
    """Two host-callback jobs running at the same time in one process (their resident launches share the device and possibly a
    hardware queue): both finish, each with the oracle's trace."""
    addr = fn_addr(fixture_lib.eight_schools_logp)
    smps = []
    for seed in (31, 32, 33):
        s = hip.PyNutsSettings.Diag(seed)
        s.update(num_tune=150, num_draws=100, num_chains=64)
        m = hip.HostCallbackModel(10, addr)
        m.set_init("normal")
        smps.append((seed, hip.PySampler(s, m)))
    for seed, smp in smps:
        smp.wait()
        assert smp.host_mode in ("resident", "fell-back")
        want = oracle.sample_callback(oracle_settings(oracle, chains=64, tune=150, draws=100, seed=seed, W=1, init_kind=1), 10, addr)
        assert_trace_equal(smp.take_results(), want)


def test_resident_launch_that_fills_the_device(hip, oracle, fixture_lib):
    """1000 dimensions x 1024 chains through the host-callback path: 8 register chunks per chain, i.e. one workgroup per CU and
    256 workgroups — the resident launch needs the whole device.  Whether its roll call succeeds or (a CU is not available) fails
    and the job falls back to one launch per evaluation, the trace is the oracle's."""
    addr = fn_addr(fixture_lib.scaled_normal_logp)
    info = {}
    got, W = run_engine(hip, hip.HostCallbackModel(1000, addr), chains=1024, tune=24, draws=8, seed=77, info=info, launch=dict(host_persist=256))
    assert info["host_mode"] in ("resident", "fell-back")
    print("full-device resident launch:", info["host_mode"])
    want = oracle.sample_callback(oracle_settings(oracle, chains=1024, tune=24, draws=8, seed=77, W=W), 1000, addr)
    assert_trace_equal(got, want)


def test_resident_job_survives_progress_pause_and_partial_reads(hip, oracle, fixture_lib):
    """Reading a running resident job (progress, partial trace), pausing and resuming it each bring the launch to a boundary
    and start a new one: the finished job is still the oracle's, and an aborted one is a prefix of it."""
    import time

    addr = fn_addr(fixture_lib.eight_schools_logp)
    kw = dict(chains=48, tune=300, draws=300, seed=9)
    want = oracle.sample_callback(oracle_settings(oracle, W=1, init_kind=1, **kw), 10, addr)

    def start():
        s = hip.PyNutsSettings.Diag(kw["seed"])
        s.update(num_tune=kw["tune"], num_draws=kw["draws"], num_chains=kw["chains"])
        m = hip.HostCallbackModel(10, addr)
        m.set_init("normal")
        return hip.PySampler(s, m)

    smp = start()
    seen = []
    for _ in range(6):
        seen.append(sum(p.finished_draws for p in smp.progress()))
        part = smp.inspect()
        k = int(part.finished.min())
        assert np.array_equal(part.draws[:, :k], want.draws[:, :k])
        smp.pause()
        time.sleep(0.005)
        smp.resume()
    smp.wait()
    assert seen == sorted(seen)
    assert_trace_equal(smp.take_results(), want)

    smp = start()
    time.sleep(0.02)
    smp.abort()
    try:
        smp.wait()
    except RuntimeError:
        pass
    part = smp.take_results()
    fin = np.asarray(part.finished)
    assert (fin <= kw["tune"] + kw["draws"]).all()
    for c in range(kw["chains"]):
        assert np.array_equal(part.draws[c, : fin[c]], want.draws[c, : fin[c]])


@pytest.mark.parametrize("persist,mode", [(256, "resident"), (-20, "fell-back"), (1, "groups")])
def test_multi_wave_host_callback_launch_modes(hip, oracle, persist, mode):
    """1500 dimensions = two waves per chain: resident launches, a resident job that falls back after 20 evaluations (the first
    launch per evaluation performs the deferred first half of the leapfrog), and launches per evaluation from the start."""
    dim = 1500
    sd = np.exp(np.random.default_rng(7).normal(size=dim) * 0.5)

    def logp(x):
        z = x / sd
        return -0.5 * float(z @ z), -z / sd

    info = {}
    got, W = run_engine(hip, hip.HostCallbackModel(dim, logp), chains=5, tune=40, draws=20, seed=11, launch=dict(host_groups=2, host_persist=persist), info=info)
    assert W == 2 and info["host_mode"] == mode
    want = oracle.sample_callback(oracle_settings(oracle, chains=5, tune=40, draws=20, seed=11, W=W), dim, logp)
    assert_trace_equal(got, want)


def test_bridgestan_adapter_matches_raw_callback(hip, oracle, fixture_lib):
    # the BridgeStan C API stand-in evaluates the same density: identical trace through nphip_model_bridgestan
    model_ptr = fixture_lib.bs_model_construct(None, 0, None)
    try:
        m = hip.BridgeStanModel(10, fixture_lib, ctypes.c_void_p(model_ptr), n_threads=2)
        got, W = run_engine(hip, m, chains=8, tune=120, draws=40, seed=22, init="normal")
    finally:
        fixture_lib.bs_model_destruct(ctypes.c_void_p(model_ptr))
    want = oracle.sample_callback(oracle_settings(oracle, chains=8, tune=120, draws=40, seed=22, W=W, init_kind=1), 10, fn_addr(fixture_lib.eight_schools_logp))
    assert_trace_equal(got, want)


def test_recoverable_and_fatal_callback_codes(hip, oracle, fixture_lib):
    addr = fn_addr(fixture_lib.failing_logp)
    got, W = run_engine(hip, hip.HostCallbackModel(3, addr, n_threads=2), chains=4, tune=60, draws=60, seed=8, init="normal")
    want = oracle.sample_callback(oracle_settings(oracle, chains=4, tune=60, draws=60, seed=8, W=W, init_kind=1), 3, addr)
    assert_trace_equal(got, want)
    assert got.stats["diverging"].sum() > 0 and got.draws[:, :, 0].max() <= 2.5 + 1e-12
    # a negative code is fatal: wait() raises, as src/pymc.rs:166-180 + wrapper.rs:1131-1136
    s = hip.PyNutsSettings.Diag(1)
    s.update(num_tune=10, num_draws=10, num_chains=2)
    smp = hip.PySampler(s, hip.HostCallbackModel(3, fn_addr(fixture_lib.fatal_logp)))
    with pytest.raises(RuntimeError, match="fatal"):
        smp.wait()


@pytest.mark.parametrize("dim,chains,waves,graph_steps", [(10, 40, 0, 0), (10, 40, 0, 16), (200, 24, 0, 0), (700, 12, 0, 16), (1000, 9, 0, 0),
                                                         (1000, 9, 1, 16), (1000, 9, 4, 0), (1000, 9, 8, 16), (2500, 5, 0, 16), (5003, 3, 0, 0)])
def test_device_callback_bit_identical(hip, oracle, fixture_lib, scaled_normal_device_lib, dim, chains, waves, graph_steps):
    """The launch-per-evaluation kernels behind a batched DEVICE callback (src/pyfunc.rs:206-230 flavour; what a torch
    density runs on), plain and replayed from a HIP graph, against the oracle — zero tolerance: the device density
    (tests/fixtures/scaled_normal_device.hip, one thread per chain) performs the host C function's operations in its order."""
    fn = fn_addr(scaled_normal_device_lib.scaled_normal_device_seq)
    got, W = run_engine(hip, hip.NativeDeviceCallbackModel(dim, fn, 0, keep_alive=scaled_normal_device_lib), chains=chains, tune=70, draws=30,
                        seed=dim + chains, waves=waves, launch=dict(graph_steps=graph_steps), store_gradient=True)
    want = oracle.sample_callback(oracle_settings(oracle, chains=chains, tune=70, draws=30, seed=dim + chains, W=W, store_gradient=True), dim,
                                  fn_addr(fixture_lib.scaled_normal_logp))
    assert_trace_equal(got, want)
    assert np.array_equal(got.stats["gradient"], want.stats["gradient"])


@pytest.mark.parametrize("dim", [24, 333, 900])
@pytest.mark.parametrize("settings", [
    dict(maxdepth=12, target_accept=0.99),                  # deep trees: many merge levels per leaf, the top-level merge of deep doublings
    dict(max_energy_error=0.2),                             # many divergences (with the speculative first half taken)
    dict(max_energy_error=0.2, store_divergences=True),     # ... and with the divergence record (no speculation)
    dict(maxdepth=2),
    dict(mindepth=3),
    dict(check_turning=False, maxdepth=4),
], ids=["deep", "divergences", "divergence-records", "maxdepth2", "mindepth3", "no-turning"])
def test_device_callback_settings_variants_bit_identical(hip, oracle, fixture_lib, scaled_normal_device_lib, dim, settings):
    """The fused leaf of the launch-per-evaluation kernels (kernels.hip: leaf_cb) under the settings that change what a leaf does:
    one (D = 24), two (D = 333, odd: element-wise dense rows) and four (D = 900) waves per chain, replayed from a HIP graph."""
    fn = fn_addr(scaled_normal_device_lib.scaled_normal_device_seq)
    kw = dict(chains=5, tune=60, draws=30, seed=7 + dim)
    got, W = run_engine(hip, hip.NativeDeviceCallbackModel(dim, fn, 0, keep_alive=scaled_normal_device_lib), launch=dict(graph_steps=8), **kw, **settings)
    assert W == {24: 1, 333: 2, 900: 4}[dim]
    want = oracle.sample_callback(oracle_settings(oracle, W=W, **kw, **settings), dim, fn_addr(fixture_lib.scaled_normal_logp))
    assert_trace_equal(got, want)
    if settings.get("store_divergences"):
        for k in DIV_KEYS:
            assert np.array_equal(got.stats[k], want.stats[k], equal_nan=True), k
    if "max_energy_error" in settings:
        assert got.stats["diverging"].sum() > 10
    if settings.get("maxdepth") == 12:
        assert got.stats["depth"].max() >= 5


def test_python_callable_through_host_callback(hip, oracle):
    def logp(x):
        return -0.5 * float(x @ x), -x

    got, W = run_engine(hip, hip.HostCallbackModel(4, logp), chains=3, tune=40, draws=20, seed=2)
    want = oracle.sample_callback(oracle_settings(oracle, chains=3, tune=40, draws=20, seed=2, W=W), 4, logp)
    assert_trace_equal(got, want)


def test_torch_device_callback_matches_fused_model(hip, oracle):
    """Batched torch logp (BASELINE.json config 3 flavour).  The diagonal Gaussian evaluated by torch uses the same
    IEEE operations per element (mul, neg) but torch reduces logp in its own order: integer statistics can
    legitimately differ once a U-turn dot product rounds differently, so the comparison is statistical plus a
    tight tolerance on the first draws, where no reordering has accumulated."""
    import torch

    import nutpie_amd

    sd = np.array([0.5, 1.0, 2.0, 4.0, 0.1, 10.0])

    def make_logp():
        prec = torch.as_tensor(1 / sd**2, device="cuda")

        def f(x):
            g = -(x * prec)
            return 0.5 * (x * g).sum(-1), g

        return f

    m = nutpie_amd.from_torchfunc(6, make_logp)
    tr = nutpie_amd.sample(m, chains=64, tune=300, draws=300, seed=5, progress_bar=False, return_raw_trace=True)
    want = oracle.sample_tridiag(oracle_settings(oracle, chains=64, tune=300, draws=300, seed=5, W=1), 1 / sd**2)
    np.testing.assert_allclose(tr.draws[:, :3], want.draws[:, :3], rtol=1e-9)
    d = tr.draws[:, 300:]
    assert np.abs(d.mean((0, 1)) / sd).max() < 0.08
    assert np.abs(d.std((0, 1)) / sd - 1).max() < 0.06
    assert tr.stats["diverging"][:, 300:].sum() == 0


def test_torch_callback_exception_and_nonfinite_logp(hip):
    import torch

    import nutpie_amd

    def make_bad():
        def f(x):
            raise ValueError("boom")

        return f

    with pytest.raises(RuntimeError, match="boom"):
        nutpie_amd.sample(nutpie_amd.from_torchfunc(3, make_bad), chains=2, tune=5, draws=5, progress_bar=False)

    def make_wall():
        def f(x):  # -inf beyond a wall at x0 = 1.5: non-finite logp is recoverable (src/pyfunc.rs:218-220)
            lp = -0.5 * (x * x).sum(-1)
            lp = torch.where(x[:, 0] > 1.5, torch.full_like(lp, -float("inf")), lp)
            return lp, -x

        return f

    tr = nutpie_amd.sample(nutpie_amd.from_torchfunc(3, make_wall, init="normal"), chains=8, tune=100, draws=100, seed=1, progress_bar=False)
    assert tr.sample_stats.diverging.sum() + tr.warmup_sample_stats.diverging.sum() > 0
    assert np.nanmax(tr.posterior.x.values[..., 0]) <= 1.5


# ------------------------------------------------------------------------------------ full-size properties (config 2)
def test_config2_full_size_properties(hip):
    """BASELINE.json config 2 at full width (1000-dim AR(1) Gaussian, 1024 chains), short run: properties that do not
    need the oracle — determinism, per-chain independence from batch composition, analytic moments."""
    from nutpie_amd.gaussian import ar1_gaussian

    m = ar1_gaussian(1000)
    a, W = run_engine(hip, hip.TridiagGaussianModel(m.diag, m.offdiag), chains=1024, tune=150, draws=50, seed=20260926)
    assert W == 1 and a.finished.min() == 200
    # same seed, only the first 32 chains: identical rows (a chain never depends on its neighbours)
    b, _ = run_engine(hip, hip.TridiagGaussianModel(m.diag, m.offdiag), chains=32, tune=150, draws=50, seed=20260926)
    assert np.array_equal(a.draws[:32], b.draws) and np.array_equal(a.stats["n_steps"][:32], b.stats["n_steps"])
    d = a.draws[:, 150:]
    var = np.diag(m.covariance())
    z = d.mean((0, 1)) / np.sqrt(var)
    assert np.abs(z).max() < 0.15                       # 51200 draws per dim, autocorrelated
    ratio = d.var((0, 1)) / var
    assert 0.85 < ratio.min() and ratio.max() < 1.15
    assert a.stats["diverging"][:, 150:].mean() < 0.01
    # energy conservation of accepted points: |energy_error| is O(1), never near max_energy_error
    assert np.abs(a.stats["energy_error"][:, 150:]).max() < 50


@pytest.mark.parametrize("dim,chains", [(10000, 64), (3000, 128)])
def test_large_dimension_shapes_properties(hip, dim, chains):
    """BASELINE.json config 5's shape (D = 10 000: lean register kernel, 4 waves per chain, sigma^2 in LDS) and the
    4-wave register kernels with the LDS ring (D = 3000), short runs: determinism, independence from the batch, analytic moments."""
    m = ar1_gaussian(dim)
    kw = dict(tune=120, draws=40, seed=dim)
    a, W = run_engine(hip, hip.TridiagGaussianModel(m.diag, m.offdiag), chains=chains, **kw)
    assert W == 4 and a.finished.min() == 160
    b, _ = run_engine(hip, hip.TridiagGaussianModel(m.diag, m.offdiag), chains=4, **kw)
    assert np.array_equal(a.draws[:4], b.draws) and np.array_equal(a.stats["n_steps"][:4], b.stats["n_steps"])
    d = a.draws[:, 120:]
    var = np.diag(m.covariance()) if dim <= 3000 else None
    if var is not None:
        ratio = d.var((0, 1)) / var
        assert 0.6 < ratio.min() and ratio.max() < 1.6       # 5120 autocorrelated draws per dimension
    assert a.stats["diverging"][:, 120:].mean() < 0.02
    acc = a.stats["mean_tree_accept"][:, 120:].mean()
    assert 0.7 < acc < 0.95                                   # dual averaging reached the 0.8 target
    assert np.abs(a.stats["energy_error"][:, 120:]).max() < 50


def test_config5_full_size_properties(hip):
    """BASELINE.json config 5 on one GPU at full width: 1024 chains x D = 10 000 (the lean register kernel, one chain per CU),
    short run: every chain finishes, chains do not depend on the batch (the first four equal a 4-chain job bit for bit),
    dual averaging reaches its target, energy errors stay O(1)."""
    m = ar1_gaussian(10000)
    kw = dict(tune=40, draws=10, seed=10000)
    a, W = run_engine(hip, hip.TridiagGaussianModel(m.diag, m.offdiag), chains=1024, **kw)
    assert W == 4 and a.finished.min() == 50
    b, _ = run_engine(hip, hip.TridiagGaussianModel(m.diag, m.offdiag), chains=4, **kw)
    assert np.array_equal(a.draws[:4], b.draws) and np.array_equal(a.stats["n_steps"][:4], b.stats["n_steps"])
    assert a.stats["diverging"][:, 30:].mean() < 0.05
    acc = a.stats["mean_tree_accept"][:, 30:].mean()
    assert 0.6 < acc < 0.97
    assert np.abs(a.stats["energy_error"][:, 30:]).max() < 50

