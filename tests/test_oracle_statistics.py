"""The oracle is a correct NUTS: analytic moments, the reference's golden files as distributional
fixtures ("parity unpinned"), determinism/sharding behaviour, golden regression."""
import os

import numpy as np
import pytest
from scipy import stats

from tests.conftest import GOLDEN


def _mcse_ok(draws, mean, var, z=5.0):
    """mean/variance of pooled draws [chain, draw, dim] against analytic values, z * MCSE bands (ESS-aware)."""
    from nutpie_amd.ess import ess_bulk

    n_eff = np.array([ess_bulk(draws[:, :, d]) for d in range(draws.shape[2])])
    sd = np.sqrt(var)
    m_err = np.abs(draws.mean((0, 1)) - mean) / (sd / np.sqrt(n_eff))
    v_err = np.abs(draws.var((0, 1)) / var - 1) / np.sqrt(2.0 / (n_eff / 2))  # variance ESS is roughly half
    return m_err.max() < z and v_err.max() < z, (m_err.max(), v_err.max())


def test_config1_std_normal(oracle):
    # BASELINE.json config 1: 10-dim standard normal, 4 chains, tune 400 / draws 1000, seed 123
    s = oracle.default_settings(seed=123, num_chains=4, num_tune=400, num_draws=1000, n_threads=4)
    tr = oracle.sample_tridiag(s, np.ones(10))
    d = tr.draws[:, 400:]
    ok, info = _mcse_ok(d, np.zeros(10), np.ones(10))
    assert ok, info
    assert tr.stats["diverging"][:, 400:].sum() == 0
    assert 0.6 < tr.stats["mean_tree_accept"][:, 400:].mean() < 0.95     # target_accept = 0.8
    assert np.all(tr.stats["step_size"][:, -1] > 0.4) and np.all(tr.stats["step_size"][:, -1] < 1.6)
    # per-dimension normality of the pooled draws (thinned to reduce autocorrelation)
    assert stats.kstest(d[:, ::5, 0].ravel(), "norm").pvalue > 1e-3


def test_correlated_gaussian_moments(oracle):
    from nutpie_amd.gaussian import ar1_gaussian

    m = ar1_gaussian(40, rho=0.9, seed=1)
    s = oracle.default_settings(seed=2, num_chains=8, num_tune=400, num_draws=1000, n_threads=8)
    tr = oracle.sample_tridiag(s, m.diag, m.offdiag)
    d = tr.draws[:, 400:]
    cov = m.covariance()
    ok, info = _mcse_ok(d, np.zeros(40), np.diag(cov))
    assert ok, info
    # neighbouring correlations recovered
    flat = d.reshape(-1, 40)
    emp = np.corrcoef(flat.T)
    want = cov / np.sqrt(np.outer(np.diag(cov), np.diag(cov)))
    assert np.abs(emp - want).max() < 0.08
    assert tr.stats["diverging"][:, 400:].sum() == 0


def _halfnormal_log_scale():
    # HalfNormal(1) on the log scale (PyMC's default transform of `pm.HalfNormal("a")`,
    # reference tests/test_pymc.py:533-552): x = log a, logp(x) = x - exp(2x)/2
    def logp(x):
        e = np.exp(2 * x[0])
        return float(x[0] - 0.5 * e), np.array([1.0 - e])

    return logp


def test_halfnormal_fixture_is_distributionally_compatible(oracle):
    ref = np.loadtxt(os.path.join(GOLDEN, "reference_halfnormal_numba.txt"))
    assert ref.shape == (200,)
    s = oracle.default_settings(seed=123, num_chains=2, num_tune=100, num_draws=100)
    tr = oracle.sample_callback(s, 1, _halfnormal_log_scale())
    ours = np.exp(tr.draws[:, 100:, 0]).ravel()
    # the engine's RNG stream is not nuts-rs' stream: compare distributions, not values.
    # 200 autocorrelated draws: use a loose two-sample KS and the analytic law.
    s_big = oracle.default_settings(seed=5, num_chains=8, num_tune=300, num_draws=2000, n_threads=8)
    big = np.exp(oracle.sample_callback(s_big, 1, _halfnormal_log_scale()).draws[:, 300:, 0])
    assert abs(big.mean() - np.sqrt(2 / np.pi)) < 0.03           # E|Z| = 0.798
    assert stats.kstest(big[:, ::10].ravel(), "halfnorm").pvalue > 1e-3
    # the reference's 200 golden values and our 200 values (same settings: 2 chains x 100 draws after only
    # 100 tuning draws, strongly autocorrelated: chain means 0.48 / 0.64 in the fixture) are both
    # HalfNormal(1)-like: loose KS distance and summary bands, not a p-value on 200 dependent draws
    for sample in (ref, ours):
        assert np.all(sample > 0) and sample.max() < 4.5
        assert 0.3 < sample.mean() < 1.3
        assert stats.ks_2samp(sample, big[:, ::10].ravel()).statistic < 0.35
    # like the reference fixture (values 3-4 are identical), rejected transitions repeat values
    assert np.any(np.diff(ref) == 0)


def test_stan_fixture_shape():
    ref = np.loadtxt(os.path.join(GOLDEN, "reference_halfnormal_stan.txt"))
    assert ref.shape == (2, 10) and np.all(ref > 0)


def test_seed_semantics(oracle):
    # reference tests/test_stan.py:67-101: same seed -> same draws; other seed -> different; chains pairwise distinct
    a = oracle.sample_tridiag(oracle.default_settings(seed=42, num_chains=3, num_tune=50, num_draws=50), np.ones(3))
    b = oracle.sample_tridiag(oracle.default_settings(seed=42, num_chains=3, num_tune=50, num_draws=50, n_threads=3), np.ones(3))
    c = oracle.sample_tridiag(oracle.default_settings(seed=43, num_chains=3, num_tune=50, num_draws=50), np.ones(3))
    assert np.array_equal(a.draws, b.draws)      # also: thread count does not matter
    assert not np.allclose(a.draws, c.draws)
    for i in range(3):
        for j in range(i + 1, 3):
            assert not np.allclose(a.draws[i], a.draws[j])
        for j in range(3):
            assert not np.allclose(a.draws[i], c.draws[j])


def test_chain_sharding_invariance(oracle):
    # chains are keyed by their GLOBAL id: two shards == one job (multi-GPU contract)
    full = oracle.sample_tridiag(oracle.default_settings(seed=9, num_chains=5, num_tune=40, num_draws=30), np.arange(1.0, 8.0))
    lo = oracle.sample_tridiag(oracle.default_settings(seed=9, num_chains=3, num_tune=40, num_draws=30, chain_offset=0), np.arange(1.0, 8.0))
    hi = oracle.sample_tridiag(oracle.default_settings(seed=9, num_chains=2, num_tune=40, num_draws=30, chain_offset=3), np.arange(1.0, 8.0))
    assert np.array_equal(full.draws, np.concatenate([lo.draws, hi.draws]))
    assert np.array_equal(full.stats["n_steps"], np.concatenate([lo.stats["n_steps"], hi.stats["n_steps"]]))


def test_reduction_geometry_changes_bits_not_statistics(oracle):
    a = oracle.sample_tridiag(oracle.default_settings(seed=1, num_chains=2, num_tune=60, num_draws=40, waves_per_chain=1), np.ones(300))
    b = oracle.sample_tridiag(oracle.default_settings(seed=1, num_chains=2, num_tune=60, num_draws=40, waves_per_chain=4), np.ones(300))
    assert np.array_equal(a.draws[:, 0], b.draws[:, 0]) or np.allclose(a.draws[:, 0], b.draws[:, 0], rtol=1e-9)
    assert abs(a.stats["n_steps"].mean() - b.stats["n_steps"].mean()) < 0.5 * a.stats["n_steps"].mean()


@pytest.mark.parametrize("name", ["stdnormal_d10", "ar1_d257", "diag_d1000_w2", "divergent_d3", "maxdepth3_d64"])
def test_oracle_matches_committed_golden_vectors(oracle, name):
    # fixtures generated by tests/golden/make_golden.py from THIS REPO'S oracle (not from nuts-rs)
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    tr = mg.run_case(name)
    gold = np.load(os.path.join(GOLDEN, f"oracle_{name}.npz"))
    for k in ("depth", "n_steps", "index_in_trajectory", "diverging", "maxdepth_reached", "tuning"):
        assert np.array_equal(tr.stats[k], gold[k]), k
    for k in ("energy", "logp", "step_size", "step_size_bar", "mean_tree_accept"):
        assert np.array_equal(tr.stats[k], gold[k]), k
    assert np.array_equal(tr.draws[:, ::10, :8], gold["draws_thin"])
