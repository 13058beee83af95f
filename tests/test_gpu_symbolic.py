"""The expression front-end on the GPU (nutpie_amd/symbolic.py): generated densities against the numpy evaluation of the same
graph, the generated radon model against the hand-written HIP density of nutpie_amd/radon.py, sampling and data swapping."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import symbolic_models as zoo  # noqa: E402

import nutpie_amd  # noqa: E402
from nutpie_amd.radon import radon_density_model, synthetic_radon_data  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(zoo.ALL))
def test_generated_density_equals_the_numpy_evaluation_of_its_graph(hip, name):
    m = zoo.ALL[name]().compile()
    rng = np.random.default_rng(7)
    x = 0.4 * rng.normal(size=(37, m.n_dim))
    lp, g = m.logp_and_grad(x)
    lp_ref, g_ref = m.logp_and_grad_numpy(x)
    # same operations in the same order (-ffp-contract=off); exp / log / log1p differ from numpy's in the last place,
    # and the sums run in the wave's order
    np.testing.assert_allclose(lp, lp_ref, rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-9)


def test_generated_radon_equals_the_hand_written_density(hip):
    d = synthetic_radon_data()
    gen = zoo.radon(d).compile()
    hand = radon_density_model(d)
    assert gen.n_dim == hand.n_dim
    x = 0.4 * np.random.default_rng(1).normal(size=(64, gen.n_dim))
    lp_a, g_a = gen.logp_and_grad(x)
    lp_b, g_b = hand.logp_and_grad(x)
    np.testing.assert_allclose(lp_a, lp_b, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(g_a, g_b, rtol=1e-9, atol=1e-9)
    # and the same posterior through the resident kernel
    a = nutpie_amd.sample(gen, chains=128, tune=300, draws=200, seed=3, progress_bar=False)
    b = nutpie_amd.sample(hand, chains=128, tune=300, draws=200, seed=3, progress_bar=False)
    for k in ("intercept", "floor_effect", "sigma", "county_sd", "county_floor_sd"):
        va, vb = a.posterior[k].values, b.posterior[k].values
        assert abs(va.mean() - vb.mean()) < 4 * vb.std() / np.sqrt(2000), k
    ea, eb = a.posterior.county_effect.values, b.posterior.county_effect.values
    assert np.abs(ea.mean((0, 1)) - eb.mean((0, 1))).max() < 0.03
    assert np.abs(ea.sum(-1)).max() < 1e-9 and a.sample_stats.diverging.values.mean() < 0.02


def test_resident_and_batched_forms_draw_the_same(hip):
    m = zoo.logistic()
    a = nutpie_amd.sample(m.compile(), chains=32, tune=120, draws=50, seed=5, progress_bar=False)
    b = nutpie_amd.sample(m.compile(resident=False), chains=32, tune=120, draws=50, seed=5, progress_bar=False)
    assert np.array_equal(a.posterior.b0.values, b.posterior.b0.values)
    assert np.array_equal(a.sample_stats.n_steps.values, b.sample_stats.n_steps.values)
    # the posterior recovers the generating coefficients (0.3, 0.8) within its own spread
    assert abs(a.posterior.b0.values.mean() - 0.3) < 0.6 and abs(a.posterior.b1.values.mean() - 0.8) < 0.4
    assert a.posterior.group_effect.shape == (32, 50, 7)


def test_with_data_on_a_generated_model(hip):
    from scipy.special import gammaln

    compiled = zoo.poisson_offsets().compile(specialize=False)      # one library for data of any length
    rng = np.random.default_rng(11)
    site = rng.integers(0, 40, 777)
    y2 = rng.poisson(3.0, 777).astype(np.float64)
    swapped = compiled.with_data(y=y2, log_fact=gammaln(y2 + 1.0), site_idx=site, prior_scale=0.7)
    assert swapped.library().path == compiled.library().path
    # the default: a source specialised to the lengths of its data — another length is compiled when it arrives
    special = zoo.poisson_offsets().compile()
    respecialised = special.with_data(y=y2, log_fact=gammaln(y2 + 1.0), site_idx=site, prior_scale=0.7)
    assert respecialised.library().path != special.library().path
    x = 0.3 * rng.normal(size=(9, compiled.n_dim))
    for mm in (compiled, swapped, special, respecialised):
        lp, g = mm.logp_and_grad(x)
        lp_ref, g_ref = mm.logp_and_grad_numpy(x)
        np.testing.assert_allclose(lp, lp_ref, rtol=1e-11, atol=1e-10)
        np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-9)
    tr = nutpie_amd.sample(swapped, chains=16, tune=150, draws=80, seed=2, progress_bar=False)
    assert tr.posterior.u.shape == (16, 80, 40) and np.all(tr.posterior.tau.values > 0)


@pytest.mark.parametrize("waves", [2, 4])
def test_several_waves_per_chain(hip, waves):
    """``waves_per_chain``: two or four wavefronts evaluate one chain's density together (fewer chains than SIMDs).  The sums run
    in another order than with one wave, so the comparison with one wave is to rounding; resident and batched forms of the SAME
    library draw identically."""
    for name in ("radon", "logistic", "scalar_only"):
        front = zoo.ALL[name]()
        m = front.compile(waves_per_chain=waves)
        x = 0.4 * np.random.default_rng(3).normal(size=(19, m.n_dim))
        lp, g = m.logp_and_grad(x)
        lp_ref, g_ref = m.logp_and_grad_numpy(x)
        np.testing.assert_allclose(lp, lp_ref, rtol=1e-11, atol=1e-10)
        np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-9)
    front = zoo.radon()
    a = nutpie_amd.sample(front.compile(waves_per_chain=waves), chains=48, tune=150, draws=60, seed=9, progress_bar=False)
    b = nutpie_amd.sample(front.compile(waves_per_chain=waves, resident=False), chains=48, tune=150, draws=60, seed=9, progress_bar=False)
    assert np.array_equal(a.posterior.sigma.values, b.posterior.sigma.values)
    assert np.array_equal(a.sample_stats.n_steps.values, b.sample_stats.n_steps.values)
    assert abs(a.posterior.sigma.values.mean() - 0.75) < 0.08 and a.sample_stats.diverging.values.mean() < 0.02
    one = nutpie_amd.sample(front.compile(), chains=48, tune=150, draws=60, seed=9, progress_bar=False)
    # same model, same seed: the first draws agree to rounding (the trajectories drift apart later, as any two summation orders do)
    np.testing.assert_allclose(a.warmup_posterior.sigma.values[:, :3], one.warmup_posterior.sigma.values[:, :3], rtol=1e-6)


def test_design_matrix_regression_recovers_its_coefficients(hip):
    """``X @ beta`` (nutpie_amd.symbolic.Matrix): the predictor is a sum over the matrix's columns, the gradient with respect to the
    coefficients one wave-wide sum per column; here the coefficients are a computed vector (raw * tau)."""
    m = zoo.regression().compile()
    tr = nutpie_amd.sample(m, chains=64, tune=300, draws=200, seed=4, progress_bar=False)
    beta = tr.posterior.beta.values.reshape(-1, 6)
    want = np.array([1.0, -0.5, 0.0, 0.25, 0.0, 2.0])
    assert np.abs(beta.mean(0) - want).max() < 0.12, beta.mean(0)
    assert abs(tr.posterior.sigma.values.mean() - 0.5) < 0.08 and tr.sample_stats.diverging.values.mean() < 0.03
    # new rows (another number of them: the specialised source is compiled for it)
    rng = np.random.default_rng(8)
    X2 = rng.normal(size=(150, 6))
    y2 = X2 @ np.array([0.0, 0.0, 3.0, 0.0, 0.0, 0.0]) + 0.5 * rng.normal(size=150)
    m2 = m.with_data(X=X2, y=y2, group_idx=rng.integers(0, 9, 150))
    tr2 = nutpie_amd.sample(m2, chains=64, tune=300, draws=200, seed=4, progress_bar=False)
    assert abs(tr2.posterior.beta.values[..., 2].mean() - 3.0) < 0.2


def test_eight_schools_through_the_front_end(hip):
    """the classic (non-centred), with bounded nuisance parameters riding along: the known posterior of (mu, tau) and the bounds"""
    m = zoo.eight_schools().compile()
    tr = nutpie_amd.sample(m, chains=256, tune=400, draws=400, seed=12, progress_bar=False, target_accept=0.9)
    mu, tau = tr.posterior.mu.values, tr.posterior.tau.values
    assert 3.3 < mu.mean() < 5.5 and 2.6 < tau.mean() < 4.6, (mu.mean(), tau.mean())       # Gelman et al.: E mu ~ 4.4, E tau ~ 3.6
    assert tr.posterior.theta.shape == (256, 400, 8) and abs(tr.posterior.theta.values[..., 0].mean() - 6.2) < 1.0
    w, u, r = tr.posterior.w.values, tr.posterior.u.values, tr.posterior.rates.values
    assert w.min() > -1.0 and w.max() < 3.0 and u.max() < 2.0 and r.min() > 0.5 and r.max() < 4.0
    assert tr.sample_stats.diverging.values.mean() < 0.02


def test_a_model_too_large_for_four_chains_per_workgroup(hip):
    """3000 observations: the per-chain scratch only fits with one chain per workgroup — compile() picks two waves per chain, the
    data stay in global memory (too much to stage); same checks as the small model"""
    from nutpie_amd.radon import synthetic_radon_data

    m = zoo.radon(synthetic_radon_data(n_obs=3000)).compile()
    assert m._waves == 2
    x = 0.3 * np.random.default_rng(5).normal(size=(11, m.n_dim))
    lp, g = m.logp_and_grad(x)
    lp_ref, g_ref = m.logp_and_grad_numpy(x)
    np.testing.assert_allclose(lp, lp_ref, rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-8)
    tr = nutpie_amd.sample(m, chains=64, tune=200, draws=100, seed=6, progress_bar=False)
    assert abs(tr.posterior.sigma.values.mean() - 0.75) < 0.05 and tr.sample_stats.diverging.values.mean() < 0.02


def test_intermediate_arrays_in_device_memory(hip):
    """40 000 observations: 640 KB of adjoints per chain — the generated density keeps them in device memory (one block per resident
    chain, data.scratch__), everything else as before; resident and batched forms of the library draw the same"""
    from nutpie_amd.radon import synthetic_radon_data

    front = zoo.radon(synthetic_radon_data(n_obs=40000))
    m = front.compile()
    assert m._waves == 1 and m._scratch(m._data) == 80000
    x = 0.2 * np.random.default_rng(5).normal(size=(9, m.n_dim))
    lp, g = m.logp_and_grad(x)
    lp_ref, g_ref = m.logp_and_grad_numpy(x)
    np.testing.assert_allclose(lp, lp_ref, rtol=1e-11, atol=1e-8)
    np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-7)
    a = nutpie_amd.sample(m, chains=24, tune=100, draws=40, seed=6, progress_bar=False)
    b = nutpie_amd.sample(front.compile(resident=False), chains=24, tune=100, draws=40, seed=6, progress_bar=False)
    assert np.array_equal(a.posterior.sigma.values, b.posterior.sigma.values)
    assert abs(a.posterior.sigma.values.mean() - 0.75) < 0.03


@pytest.mark.parametrize("name", ["store_extra", "dirichlet_counts", "dims_model", "uniform_det", "eight_schools", "radon", "regression"])
def test_generated_expand_equals_the_numpy_evaluation(hip, name):
    """The expand step as generated device code (nphip_expand behind nphip_model_set_device_expand: SURVEY §8f N2 for models of
    the front-end) against the numpy evaluation of the same graph: a sampled trace's expanded variables, variable by variable —
    shapes as the reference reports them (transposed two-dimensional values included)."""
    m = zoo.ALL[name]().compile()
    tr = nutpie_amd.sample(m, chains=6, tune=60, draws=40, seed=9, progress_bar=False, return_raw_trace=True)
    assert tr.expanded is not None and "__flat__" in tr.expanded          # the engine's expand ran, on the device
    got = m._unflatten(tr.expanded["__flat__"])
    want = m._expand_draws(tr.draws)
    assert list(got) == list(want) == list(m.shapes)
    for k in want:
        assert got[k].shape == want[k].shape == (6, 100, *m.shapes[k])
        np.testing.assert_allclose(got[k], want[k], rtol=1e-12, atol=1e-13, err_msg=k)


def test_reference_front_end_models_sample(hip):
    """The reference's own front-end test models (tests/test_pymc.py:303-349, 618-640, 210-222) through `nutpie_amd.sample`: what
    the reference asserts about `trace.posterior`, and the posteriors' known answers."""
    tr = nutpie_amd.sample(zoo.store_extra().compile(), chains=64, tune=300, draws=300, seed=3, progress_bar=False, store_unconstrained=True,
                           store_mass_matrix=True, store_gradient=True)
    assert tr.posterior.c.dims == ("chain", "draw", "foo") and tr.posterior.d.dims == ("chain", "draw", "bar")
    # tests/test_pymc.py:331-346: the unconstrained values under PyMC's names; a transform that takes an element away takes the dim away
    up = tr.unconstrained_posterior
    assert up.b_log__.dims == ("chain", "draw", "foo") and up.c_zerosum__.dims != ("chain", "draw", "foo") and up.d_simplex__.dims != ("chain", "draw", "bar")
    assert "b_log__" not in tr.posterior and up.c_zerosum__.shape == (64, 300, 4) and up.d_simplex__.shape == (64, 300, 3)
    np.testing.assert_allclose(np.exp(up.b_log__.values), tr.posterior.b.values, rtol=1e-13)
    np.testing.assert_array_equal(tr.sample_stats.unconstrained_draw.values[..., 5:10], up.b_log__.values)
    assert tr.sample_stats.gradient.shape == tr.sample_stats.mass_matrix_inv.shape == (64, 300, 17)
    c, d, b = tr.posterior.c.values, tr.posterior.d.values, tr.posterior.b.values
    np.testing.assert_allclose(c.sum(-1), 0, atol=1e-12)
    np.testing.assert_allclose(d.sum(-1), 1, atol=1e-12)
    assert abs(d.mean() - 0.25) < 0.01 and abs(b.mean() - np.sqrt(2 / np.pi)) < 0.03
    # Dirichlet(1, 1, 1, 1): every component is Beta(1, 3): variance 3 / 80
    assert abs(d.var() - 3.0 / 80.0) < 0.004
    post = nutpie_amd.sample(zoo.dims_model().compile(), chains=8, tune=200, draws=100, seed=1, progress_bar=False).posterior
    assert post["zero_sum"].dims == ("chain", "draw", "a", "b") and post["one_sum"].dims == ("chain", "draw", "b", "a")
    np.testing.assert_allclose(post["zero_sum"].values.sum(2), 0, atol=1e-5)       # over a
    np.testing.assert_allclose(post["one_sum"].values.sum(3), 1, atol=1e-5)
    tr = nutpie_amd.sample(zoo.no_prior().compile(), chains=64, tune=300, draws=300, seed=5, progress_bar=False)
    a = tr.posterior.a.values
    assert abs(a.mean()) < 0.05 and abs(a.std() - 1.0) < 0.05      # a | b = 0 ~ N(0, 1)
    # Dirichlet-multinomial: the posterior is Dirichlet(2.5 + counts)
    tr = nutpie_amd.sample(zoo.dirichlet_counts().compile(), chains=128, tune=400, draws=400, seed=8, progress_bar=False)
    alpha = 2.5 + np.array([12.0, 3.0, 0.0, 7.0, 30.0, 1.0])
    np.testing.assert_allclose(tr.posterior.p.values.mean((0, 1)), alpha / alpha.sum(), atol=0.004)
    assert tr.sample_stats.diverging.values.mean() < 0.01
