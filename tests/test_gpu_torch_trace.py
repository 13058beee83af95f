"""Torch log-densities compiled into the engine (``nutpie_amd.torch_trace`` behind ``nutpie_amd.from_torch_density``): the generated
device code — density AND symbolic gradient — against ``torch.autograd`` on the user's function, the traced radon model of BASELINE
config 3 against the model written with the front-end by hand, sampling through ``nutpie_amd.sample``."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import torch_models as TM  # noqa: E402

import nutpie_amd  # noqa: E402
from nutpie_amd.radon import radon_symbolic_model, radon_traced_model, synthetic_radon_data  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(TM.ALL))
def test_compiled_density_and_gradient_equal_autograd(hip, name):
    D, fn, batched, shared = TM.ALL[name]()
    m = nutpie_amd.from_torch_density(D, fn, compile=True, batched=batched, shared_data=shared)
    x = 0.4 * np.random.default_rng(4).normal(size=(41, D))
    lp, g = m.logp_and_grad(x)                       # the generated HIP code, one wave per row
    lp0, g0 = TM.autograd(fn, x, batched, shared)     # torch on the host
    # (device exp / log / lgamma differ from the host's in the last place; the sums run in the wave's order)
    np.testing.assert_allclose(lp, lp0, rtol=1e-12, atol=1e-12 * np.abs(lp0).max())
    np.testing.assert_allclose(g, g0, rtol=1e-10, atol=1e-12 * np.abs(g0).max())


def test_traced_radon_is_the_hand_written_model(hip):
    d = synthetic_radon_data()
    traced = radon_traced_model(d)
    hand = radon_symbolic_model(d).compile()
    assert traced.n_dim == hand.n_dim == 173
    x = 0.4 * np.random.default_rng(1).normal(size=(64, 173))
    lp_a, g_a = traced.logp_and_grad(x)
    lp_b, g_b = hand.logp_and_grad(x)
    # the hand-written model drops the additive constants of its priors as the torch function does: same density to rounding
    np.testing.assert_allclose(lp_a, lp_b, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(g_a, g_b, rtol=1e-9, atol=1e-9)
    a = nutpie_amd.sample(traced, chains=128, tune=300, draws=200, seed=3, progress_bar=False)
    b = nutpie_amd.sample(hand, chains=128, tune=300, draws=200, seed=3, progress_bar=False)
    # same seed, same initial points, densities equal to rounding: the first draws coincide to rounding ...
    np.testing.assert_allclose(a.warmup_posterior.sigma.values[:, :3], b.warmup_posterior.sigma.values[:, :3], rtol=1e-6)
    # ... and the posteriors agree
    for k in ("intercept", "floor_effect", "sigma", "county_sd", "county_floor_sd"):
        va, vb = a.posterior[k].values, b.posterior[k].values
        assert abs(va.mean() - vb.mean()) < 4 * vb.std() / np.sqrt(2000), k
    assert np.abs(a.posterior.county_effect.values.sum(-1)).max() < 1e-9
    assert a.sample_stats.diverging.values.mean() < 0.02
    assert a.posterior.county_effect.shape == (128, 200, 85)


def test_compiled_and_eager_paths_sample_the_same_posterior(hip):
    import torch

    D, fn, batched, shared = TM.eight_schools()
    y = torch.tensor([28.0, 8.0, -3.0, 7.0, -1.0, 1.0, 18.0, 12.0], dtype=torch.float64, device="cuda")
    sigma = torch.tensor([15.0, 10.0, 16.0, 11.0, 9.0, 11.0, 10.0, 18.0], dtype=torch.float64, device="cuda")

    def fn_cuda(x):       # the same density with its data on the GPU (the eager path evaluates it there)
        mu, log_tau, eta = x[:, 0], x[:, 1], x[:, 2:]
        tau = torch.exp(log_tau)
        z = (y - (mu[:, None] + tau[:, None] * eta)) / sigma
        return -0.5 * (z * z).sum(-1) - 0.5 * (eta * eta).sum(-1) - 0.5 * (mu / 5.0) ** 2 - torch.log1p((tau / 5.0) ** 2) + log_tau

    compiled = nutpie_amd.from_torch_density(D, fn, compile=True)
    eager = nutpie_amd.from_torch_density(D, fn_cuda, compile=False)
    a = nutpie_amd.sample(compiled, chains=64, tune=300, draws=300, seed=8, progress_bar=False)
    b = nutpie_amd.sample(eager, chains=64, tune=300, draws=300, seed=8, progress_bar=False)
    xa, xb = a.posterior.x.values.reshape(-1, D), b.posterior.x.values.reshape(-1, D)
    assert np.abs(xa.mean(0) - xb.mean(0)).max() < 0.25 and abs(xa[:, 0].mean() - 4.4) < 1.0
    assert np.abs(xa.std(0) / xb.std(0) - 1).max() < 0.15


def test_with_data_on_a_traced_model_samples_the_new_posterior(hip):
    D, fn, batched, shared = TM.linear_regression_unbatched()
    m = nutpie_amd.from_torch_density(D, fn, compile=True, batched=False, shared_data=shared)
    rng = np.random.default_rng(0)
    beta = np.array([2.0, -1.0, 0.5, 0.0])
    y2 = shared["X"] @ beta + 0.3 * rng.normal(size=shared["X"].shape[0])
    m2 = m.with_data(y=y2)
    assert m2.library().path == m.library().path         # only values changed: nothing is compiled again
    tr = nutpie_amd.sample(m2, chains=32, tune=300, draws=200, seed=1, progress_bar=False)
    x = tr.posterior.x.values.reshape(-1, D)
    assert np.abs(x[:, :4].mean(0) - beta).max() < 0.1 and abs(np.exp(x[:, 4]).mean() - 0.3) < 0.06
