"""The oracle against numbers nuts-rs itself produced and the reference's tree holds (CPU; the engine's counterpart — the same
comparison on the GPU, engine == oracle bit for bit — is tests/test_gpu_reference_fixtures.py).

A. ``tests/golden/reference_doc_step_sizes.json`` — the FINAL step sizes of 36 chains in the reference's frozen documentation
   (docs/_freeze, extracted by tests/golden/make_reference_doc_step_sizes.py): three analytic models under default settings.  They
   pin what the step size converges to: the acceptance statistic (A.6), dual averaging and its constants (A.7), the symmetric
   statistic in the late windows and ``step_size_bar`` on the last tuning draw (A.8) — profiles/r5_reference_sensitivity.txt shows
   which recalled details these numbers can tell apart (plain acceptance late: z = +4.7; target 0.75 / 0.85: z = -10 / +13;
   dual-averaging gamma 0.1: z = -7; the last draw keeping its step: spread x 8) and which they cannot (window lengths, switch
   frequencies, the minimum count of a refresh, when the step-size search runs).
B. ``tests/golden/reference_halfnormal_stan.txt`` (2 x 10 draws, N(0, 1) initial points) ranked inside an ensemble of runs of its shape.
"""
import ctypes
import json
import os

import numpy as np
import pytest
from scipy import stats

from tests.conftest import FIXTURES, GOLDEN

DOC = json.load(open(os.path.join(GOLDEN, "reference_doc_step_sizes.json")))


def gaussian(name):
    """(diag, offdiag, mean) of the model's posterior precision — tridiagonal, so the oracle's analytic model covers it"""
    if name == "normal_1d":                                   # mu ~ N(0, 1), y = [1, 2, 3] ~ N(mu, 1): posterior N(1.5, 1/4)
        return np.array([4.0]), None, np.array([1.5])
    x = np.array([1.0, 2.0, 3.0]) if name == "regression_x123" else np.array([4.0, 5.0, 6.0])
    X = np.stack([np.ones(3), x], 1)                          # intercept, slope ~ N(0, 1); y = [1, 2, 3] ~ N(X b, 0.1)
    P = np.eye(2) + X.T @ X / 0.01
    return np.diag(P).copy(), np.array([P[0, 1]]), np.linalg.solve(P, X.T @ np.array([1.0, 2.0, 3.0]) / 0.01)


def doc_values(name, key):
    return np.array([row[key] for run in DOC[name]["runs"] for row in run], dtype=np.float64)


@pytest.mark.parametrize("name", ["normal_1d", "regression_x123", "regression_x456"])
def test_final_step_sizes_of_the_reference_docs(oracle, name):
    diag, off, mu = gaussian(name)
    n = 1000
    s = oracle.default_settings(seed=11, num_chains=n, num_tune=400, num_draws=20, n_threads=8, init_kind=2)
    pts = np.random.default_rng(5).uniform(-1, 1, size=(n, len(diag)))          # PyMC: support point 0 + U(-1, 1)
    tr = oracle.sample_tridiag(s, diag, off, mu=mu, init_points=pts)
    ours = tr.stats["step_size"][:, 400]                                        # the step size sampling runs with
    assert np.array_equal(ours, tr.stats["step_size"][:, 401]) and not tr.stats["diverging"][:, 400:].any()
    ref = doc_values(name, "step_size")
    z = (ref.mean() - ours.mean()) / (ours.std() / np.sqrt(len(ref)))
    print(f"{name}: reference {ref.mean():.3f} +- {ref.std(ddof=1):.3f} (n = {len(ref)}), oracle {ours.mean():.3f} +- {ours.std():.3f}, z = {z:+.2f}")
    assert abs(z) < 3.0
    assert stats.ks_2samp(ref, ours).pvalue > 0.01
    lo, hi = np.percentile(ours, [0.1, 99.9])
    assert np.all((ref > lo - 0.005) & (ref < hi + 0.005))                      # (the docs print two decimals)
    # gradient evaluations of the chain's last draw.  The reference's counts are not all 2^depth - 1 (9, 11, 13, 19, 27 appear): a
    # doubling whose new half meets a U-turn in one of its sub-trees stops building there (`extend` returns at the first
    # turning sub-tree) — the restatement produces the same counts, with the same mean
    g_ref = doc_values(name, "gradients_last_draw")
    g = tr.stats["n_steps"][:, 400:].ravel()
    support = set(np.unique(g).tolist())
    assert set(g_ref.astype(int).tolist()) <= support, (sorted(set(g_ref.astype(int).tolist()) - support), sorted(support))
    zg = (g_ref.mean() - g.mean()) / (g.std() / np.sqrt(len(g_ref)))
    print(f"   gradients in a sampling draw: reference mean {g_ref.mean():.2f}, oracle {g.mean():.2f} (z = {zg:+.2f}); oracle support {sorted(support)}")
    assert abs(zg) < 3.0
    assert stats.binomtest(int(np.sum(g_ref == 1)), len(g_ref), float(np.mean(g == 1))).pvalue > 0.01


def run_shape_stats(a):
    """statistics of one run a[2 chains, n draws]"""
    l = np.log(a)
    lc = l - l.mean(1, keepdims=True)
    return {"pooled_mean": a.mean(), "median": np.median(a), "lag1_log": float((lc[:, 1:] * lc[:, :-1]).sum() / max((lc * lc).sum(), 1e-300)),
            "repeat_fraction": np.mean(a[:, 1:] == a[:, :-1]), "min_log": l.min(), "max": a.max()}


def test_stan_halfnormal_fixture_lies_in_the_bulk_of_the_ensemble(oracle):
    """tests/test_stan.py:282-302 returns the first 10 draws of 2 chains (seed 123, tune 100, N(0, 1) initial points): 2000 runs of
    that shape; every statistic of the reference's 20 values inside the 0.5 - 99.5 % band."""
    fix = ctypes.CDLL(os.path.join(FIXTURES, "libbs_standin.so"))
    R = 2000
    s = oracle.default_settings(seed=123, num_chains=2 * R, num_tune=100, num_draws=10, n_threads=8, init_kind=1)
    tr = oracle.sample_callback(s, 1, ctypes.cast(fix.halfnormal_logp, ctypes.c_void_p).value)
    a = np.exp(tr.draws[:, 100:, 0]).reshape(R, 2, 10)
    ens = [run_shape_stats(a[r]) for r in range(R)]
    ref = run_shape_stats(np.loadtxt(os.path.join(GOLDEN, "reference_halfnormal_stan.txt")).reshape(2, 10))
    ranks = {k: float(np.mean(np.array([e[k] for e in ens]) < v)) for k, v in ref.items()}
    print("ranks of the Stan fixture:", {k: round(v, 3) for k, v in ranks.items()})
    for k, r in ranks.items():
        assert 0.005 <= r <= 0.995, (k, r)


def test_funnel_of_the_reference_docs(oracle):
    """docs/sample-stats.qmd:18-35: Neal's funnel (log_sigma ~ N(0, 1), x[5] ~ N(0, exp(log_sigma))), tune 1000 + draws 1000, default
    ("diag") adaptation — 6 chains of nuts-rs: final step sizes 0.34 .. 0.53, divergences 21 / 24 / 0 / 13 / 0 / 5, 7 gradients in the last
    draw of five chains and 15 in one.  A non-Gaussian pin: the step size the warm-up converges to on a target whose curvature varies by
    orders of magnitude, and how often the sampling phase diverges with it."""
    import subprocess

    src, out = os.path.join(FIXTURES, "funnel.c"), os.path.join(FIXTURES, "libfunnel.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src, "-lm"], check=True)
    fix = ctypes.CDLL(out)
    n = 600
    s = oracle.default_settings(seed=42, num_chains=n, num_tune=1000, num_draws=1000, n_threads=8, init_kind=2)
    pts = np.random.default_rng(3).uniform(-1, 1, size=(n, 6))                        # PyMC: support point 0 + U(-1, 1)
    tr = oracle.sample_callback(s, 6, ctypes.cast(fix.funnel_logp, ctypes.c_void_p).value, init_points=pts)
    step = tr.stats["step_size"][:, -1]
    div = tr.stats["diverging"][:, 1000:].sum(1)
    last = tr.stats["n_steps"][:, -1]
    ref = DOC["funnel_diag"]["runs"][0]
    r_step, r_div, r_last = (np.array([row[k] for row in ref], dtype=np.float64) for k in ("step_size", "divergences", "gradients_last_draw"))
    z = (r_step.mean() - step.mean()) / (step.std() / np.sqrt(len(r_step)))
    print(f"funnel: reference step {r_step.mean():.3f} +- {r_step.std(ddof=1):.3f}, oracle {step.mean():.3f} +- {step.std():.3f} (z = {z:+.2f}); divergences per chain: "
          f"reference {sorted(r_div.astype(int).tolist())}, oracle pct 5/25/50/75/95 {np.percentile(div, [5, 25, 50, 75, 95])}, chains with none {np.mean(div == 0):.2f}; "
          f"last-draw gradients: reference {sorted(r_last.astype(int).tolist())}, oracle P(7) = {np.mean(last == 7):.2f}, P(15) = {np.mean(last == 15):.2f}")
    assert abs(z) < 3.0
    lo, hi = np.percentile(step, [0.5, 99.5])
    assert np.all((r_step > lo) & (r_step < hi))
    # the reference's divergence counts lie inside this sampler's per-chain distribution (the progress table counts the sampling phase)
    assert np.all(r_div <= np.percentile(div, 99.5)) and stats.mannwhitneyu(r_div, div).pvalue > 0.01
    assert set(r_last.astype(int).tolist()) <= set(np.unique(last).tolist())


def test_correlated_102_dimensional_model_of_the_reference_docs(oracle):
    """docs/sample-stats.qmd:141-157: x ~ N(0, 1), y ~ N(x, 0.01), z[100] ~ N(y, 1) under the default adaptation, tune 1000 — 6 chains of
    nuts-rs: final step sizes 0.13 .. 0.22, 31 gradients in the last draw of five chains and 15 in one, no divergence.  102 dimensions with
    one very stiff direction a diagonal metric cannot remove: pins the step size AND the tree depth the adapted sampler ends up with."""
    import subprocess

    src, out = os.path.join(FIXTURES, "funnel.c"), os.path.join(FIXTURES, "libfunnel.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src, "-lm"], check=True)
    fix = ctypes.CDLL(out)
    n = 256
    s = oracle.default_settings(seed=42, num_chains=n, num_tune=1000, num_draws=200, n_threads=8, init_kind=2)
    pts = np.random.default_rng(3).uniform(-1, 1, size=(n, 102))
    tr = oracle.sample_callback(s, 102, ctypes.cast(fix.correlated_102d_logp, ctypes.c_void_p).value, init_points=pts)
    step = tr.stats["step_size"][:, -1]
    g = tr.stats["n_steps"][:, 1000:].ravel()
    ref = DOC["correlated_102d"]["runs"][0]
    r_step, r_last, r_div = (np.array([row[k] for row in ref], dtype=np.float64) for k in ("step_size", "gradients_last_draw", "divergences"))
    z = (r_step.mean() - step.mean()) / (step.std() / np.sqrt(len(r_step)))
    zg = (r_last.mean() - g.mean()) / (g.std() / np.sqrt(len(r_last)))
    print(f"correlated 102-d: reference step {r_step.mean():.3f} +- {r_step.std(ddof=1):.3f}, oracle {step.mean():.3f} +- {step.std():.3f} (z = {z:+.2f}); gradients per draw: "
          f"reference {sorted(r_last.astype(int).tolist())} (mean {r_last.mean():.1f}), oracle mean {g.mean():.1f} (z = {zg:+.2f}), P(31) = {np.mean(g == 31):.2f}, P(15) = {np.mean(g == 15):.2f}; "
          f"oracle divergences in the sampling phase: {int(tr.stats['diverging'][:, 1000:].sum())}")
    assert abs(z) < 3.0 and abs(zg) < 3.0
    assert set(r_last.astype(int).tolist()) <= set(np.unique(g).tolist())
    assert r_div.sum() == 0 and tr.stats["diverging"][:, 1000:].mean() < 1e-3


def test_effective_sample_size_of_the_reference_docs(oracle):
    """docs/pymc-usage.qmd:105-108: `az.ess(trace)` of the regression run (6 chains x 1000 draws after tune 400) — bulk ESS 1517.47 (intercept) and
    1517.44 (slope).  How well the adapted sampler MIXES (trajectory lengths, the multinomial draw along the trajectory), measured with this
    repository's restatement of the same estimator (nutpie_amd/ess.py): 100 runs of that shape, the reference's value ranked among them."""
    from nutpie_amd import ess

    diag, off, mu = gaussian("regression_x123")
    R = 100
    s = oracle.default_settings(seed=5, num_chains=6 * R, num_tune=400, num_draws=1000, n_threads=8, init_kind=2)
    tr = oracle.sample_tridiag(s, diag, off, mu=mu, init_points=np.random.default_rng(5).uniform(-1, 1, size=(6 * R, 2)))
    x = tr.draws[:, 400:, :].reshape(R, 6, 1000, 2)
    ours = np.array([[ess.ess_bulk(x[r, :, :, j]) for j in range(2)] for r in range(R)])
    ref = DOC["regression_x123"]["bulk_ess_of_the_first_run"]
    ranks = [float(np.mean(ours[:, j] < ref[k])) for j, k in enumerate(("intercept", "slope"))]
    print(f"bulk ESS of 6000 draws: reference {ref['intercept']:.0f} / {ref['slope']:.0f}, this sampler {ours.mean(0).round(0)} +- {ours.std(0).round(0)}, ranks {ranks}")
    assert all(0.005 <= r <= 0.995 for r in ranks)
    assert abs(ours[:, 0].mean() / ref["intercept"] - 1.0) < 0.25


def test_total_gradient_evaluations_of_the_101_dimensional_funnel(oracle):
    """docs/nf-adapt.qmd:60-78, 115-122 (frozen output): log_sigma ~ N(0, 1), x[100] ~ N(0, exp(log_sigma / 2)); ``nutpie.sample(compiled, seed=1)``
    under the default adaptation — 6 chains x (400 + 1000) draws of nuts-rs took **124 219 gradient evaluations in total, warm-up included**, reached a
    minimum bulk ESS of 31.46, final step sizes 0.28 .. 0.45, no divergence.  Every other reference-held number sees only the END of the warm-up (the
    final step size, the last draw); this one integrates its COST: the early windows, the step-size searches, the trees hitting maxdepth while the
    metric is still poor.  30 runs of the reference's shape (180 chains): the reference's total, mean step and minimum ESS must lie inside them."""
    import subprocess

    from nutpie_amd.ess import ess_bulk_all

    src, out = os.path.join(FIXTURES, "funnel.c"), os.path.join(FIXTURES, "libfunnel.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src, "-lm"], check=True)
    fix = ctypes.CDLL(out)
    ref = DOC["funnel_101d"]
    R = 30
    n = 6 * R
    s = oracle.default_settings(seed=1, num_chains=n, num_tune=400, num_draws=1000, n_threads=8, init_kind=2)
    pts = np.random.default_rng(7).uniform(-1, 1, size=(n, 101))                      # PyMC: support point 0 + U(-1, 1)
    tr = oracle.sample_callback(s, 101, ctypes.cast(fix.funnel_101d_logp, ctypes.c_void_p).value, init_points=pts)
    per_chain = tr.stats["n_steps"].sum(1).astype(np.float64)                          # warm-up + sampling, as the page sums them
    r_total = ref["totals"]["gradient_evaluations"]
    boot = per_chain[np.random.default_rng(0).integers(0, n, size=(4000, 6))].sum(1)   # runs of six chains drawn from the 180
    rank_total = float(np.mean(boot < r_total))
    z_total = (r_total / 6 - per_chain.mean()) / (per_chain.std() / np.sqrt(6))
    step = tr.stats["step_size"][:, -1]
    r_step = np.array([row["step_size"] for row in ref["runs"][0]])
    z_step = (r_step.mean() - step.mean()) / (step.std() / np.sqrt(6))
    ess = np.array([np.nanmin(ess_bulk_all(tr.draws[6 * r:6 * r + 6, 400:, :], block=101)) for r in range(R)])
    rank_ess = float(np.mean(ess < ref["totals"]["min_ess"]))
    print(f"101-d funnel: gradient evaluations per run, oracle {6 * per_chain.mean():.0f} +- {boot.std():.0f}, reference {r_total} (rank {rank_total:.3f}, z = {z_total:+.2f}); "
          f"warm-up share {tr.stats['n_steps'][:, :400].sum() / per_chain.sum():.2f}; step {step.mean():.3f} +- {step.std():.3f}, reference {r_step.mean():.3f} (z = {z_step:+.2f}); "
          f"min ESS per run pct 5/50/95 {np.percentile(ess, [5, 50, 95]).round(1)}, reference {ref['totals']['min_ess']:.1f} (rank {rank_ess:.2f})")
    assert 0.005 <= rank_total <= 0.995 and abs(z_total) < 3.0
    assert abs(z_step) < 3.0
    lo, hi = np.percentile(step, [0.5, 99.5])
    assert np.all((r_step > lo - 0.005) & (r_step < hi + 0.005))
    assert 0.02 <= rank_ess <= 0.98
    r_last = np.array([row["gradients_last_draw"] for row in ref["runs"][0]])
    assert set(r_last.tolist()) <= set(np.unique(tr.stats["n_steps"][:, 400:]).tolist())
    assert tr.stats["diverging"][:, 400:].sum(1).mean() < 1.0                          # (the reference's six chains: none)
