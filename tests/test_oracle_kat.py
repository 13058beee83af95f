"""Known-answer tests of the oracle's building blocks (the reference holds none)."""
import math
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_leapfrog_harmonic_oscillator_energy_and_reversibility(oracle):
    # H = q^2/2 + p^2/2 ; analytic solution is a rotation
    q0, p0 = np.array([1.0]), np.array([0.3])
    sig2, diag = np.ones(1), np.ones(1)
    drifts = []
    for eps in (0.1, 0.05, 0.025):
        q, p, g = q0.copy(), p0.copy(), -q0.copy()
        n = int(round(1.0 / eps))
        e0 = 0.5 * (q0[0] ** 2 + p0[0] ** 2)
        worst = 0.0
        for _ in range(n):
            q, p, g, K, U, e = oracle.leapfrog_tridiag(q, p, g, sig2, eps, diag)
            worst = max(worst, abs(e - e0))
        drifts.append(worst)
        # position close to the exact rotation by t = n*eps
        t = n * eps
        assert q[0] == pytest.approx(q0[0] * math.cos(t) + p0[0] * math.sin(t), abs=eps**2)
    # second-order integrator: energy error shrinks ~4x when eps halves
    assert 3.0 < drifts[0] / drifts[1] < 5.0 and 3.0 < drifts[1] / drifts[2] < 5.0
    # reversibility: integrate forward, flip momentum, integrate again -> back at the start
    q, p, g = q0.copy(), p0.copy(), -q0.copy()
    for _ in range(25):
        q, p, g, *_ = oracle.leapfrog_tridiag(q, p, g, sig2, 0.1, diag)
    p = -p
    for _ in range(25):
        q, p, g, *_ = oracle.leapfrog_tridiag(q, p, g, sig2, 0.1, diag)
    assert q[0] == pytest.approx(q0[0], abs=1e-13) and -p[0] == pytest.approx(p0[0], abs=1e-13)


def test_leapfrog_formula_matches_plain_numpy(oracle):
    rng = np.random.default_rng(0)
    D = 37
    a = np.exp(rng.normal(size=D)); b = 0.1 * rng.normal(size=D - 1); mu = rng.normal(size=D)
    L = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
    sig2 = np.exp(rng.normal(size=D))
    q = rng.normal(size=D); p = rng.normal(size=D); g = -L @ (q - mu)
    eps = -0.07
    q1, p1, g1, K, U, e = oracle.leapfrog_tridiag(q, p, g, sig2, eps, a, b, mu)
    ph = p + eps / 2 * g
    qn = q + eps * sig2 * ph
    gn = -L @ (qn - mu)
    pn = ph + eps / 2 * gn
    np.testing.assert_allclose(q1, qn, rtol=1e-14)
    np.testing.assert_allclose(g1, gn, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(p1, pn, rtol=1e-12, atol=1e-13)
    assert K == pytest.approx(0.5 * np.sum(pn * sig2 * pn), rel=1e-13)
    assert U == pytest.approx(0.5 * (qn - mu) @ L @ (qn - mu), rel=1e-12)


def test_uturn_triggers_when_the_trajectory_turns_back(oracle):
    # harmonic oscillator, unit mass, q0=0, p0=1: q = sin t, p = cos t, rho(t) ~ sum p ~ sin t / eps.
    # rho . v_end = sin t cos t < 0 first holds just after t = pi/2, where the particle turns around.
    eps = 0.1
    q, p, g = np.array([0.0]), np.array([1.0]), np.array([0.0])
    sig2 = np.ones(1)
    p0 = p.copy()
    psum = p.copy()
    first = None
    for k in range(1, 80):
        q, p, g, *_ = oracle.leapfrog_tridiag(q, p, g, sig2, eps, np.ones(1))
        psum = psum + p
        if oracle.is_turning(sig2, 0, p0, p0, k, p, psum):
            first = k
            break
    assert first is not None
    assert abs(first * eps - math.pi / 2) < 0.2


def test_is_turning_index_cases(oracle):
    rng = np.random.default_rng(1)
    D = 5
    sig2 = np.exp(rng.normal(size=D))
    ps = rng.normal(size=(7, D))            # momenta at trajectory indices -3..3
    idx = np.arange(-3, 4)
    # one-sided running sums: rho_k for k>=0 includes p_0; for k<0 it does not
    rho = {}
    for k in idx:
        rho[k] = ps[3:3 + k + 1].sum(0) if k >= 0 else ps[3 + k:3].sum(0)
    for a in idx:
        for b in idx:
            if a >= b:
                continue
            span = ps[3 + a:3 + b + 1].sum(0)
            want = (span @ (sig2 * ps[3 + b]) < 0) or (span @ (sig2 * ps[3 + a]) < 0)
            got = oracle.is_turning(sig2, int(a), ps[3 + a], rho[a], int(b), ps[3 + b], rho[b])
            assert got == want, (a, b)
            # argument order must not matter
            assert oracle.is_turning(sig2, int(b), ps[3 + b], rho[b], int(a), ps[3 + a], rho[a]) == want


def test_dual_averaging_matches_recurrence(oracle):
    rng = np.random.default_rng(2)
    acc = rng.uniform(0, 1, 300)
    step, bar = oracle.dual_average(acc, initial_step=0.25, target=0.8)
    k, t0, gamma, mu = 0.75, 10.0, 0.05, math.log(10 * 0.25)
    hbar, log_bar = 0.0, math.log(0.25)
    for i, a in enumerate(acc):
        n = i + 1
        w = 1.0 / (n + t0)
        hbar = (1 - w) * hbar + w * (0.8 - a)
        log_step = mu - hbar * math.sqrt(n) / gamma
        m = n ** (-k)
        log_bar = m * log_step + (1 - m) * log_bar
        assert step[i] == pytest.approx(math.exp(log_step), rel=1e-12)
        assert bar[i] == pytest.approx(math.exp(log_bar), rel=1e-12)
    # feeding the target acceptance forever keeps the step at 10x the initial value (mu)
    step, _ = oracle.dual_average(np.full(50, 0.8), initial_step=0.1)
    assert step[-1] == pytest.approx(1.0, rel=1e-12)


def test_adam_step_size_adaptation_reaches_the_target(oracle):
    # step_size_adapt_method = "adam": Adam on log(step size), gradient = accept - target; no averaged iterate,
    # so step_size_bar == step_size; the sampling-phase acceptance statistic sits near the target
    diag = np.linspace(0.5, 2.0, 30)
    for target in (0.7, 0.9):
        s = oracle.default_settings(seed=4, num_chains=8, num_tune=500, num_draws=300, n_threads=8, adam=1, adam_learning_rate=0.05,
                                    target_accept=target)
        tr = oracle.sample_tridiag(s, diag)
        post = ~tr.stats["tuning"].astype(bool)
        assert abs(tr.stats["mean_tree_accept"][post].mean() - target) < 0.06
        assert np.array_equal(tr.stats["step_size"][post], tr.stats["step_size_bar"][post])
        # frozen after tuning
        assert np.all(tr.stats["step_size"][:, 500:] == tr.stats["step_size"][:, 500:501])
    lo = oracle.sample_tridiag(oracle.default_settings(seed=4, num_chains=4, num_tune=300, num_draws=50, adam=1, target_accept=0.6), diag)
    hi = oracle.sample_tridiag(oracle.default_settings(seed=4, num_chains=4, num_tune=300, num_draws=50, adam=1, target_accept=0.95), diag)
    assert lo.stats["step_size"][:, -1].mean() > 1.5 * hi.stats["step_size"][:, -1].mean()


def test_extended_range_tree_weights_equal_log_domain_weights(oracle):
    # the (m, e) weights (nphip_spec.h) carry the same numbers as nuts-rs's log_size / logaddexp, for any finite
    # energy error, without overflow
    import math

    rng = np.random.default_rng(3)
    for scale in (0.5, 30.0, 3000.0):
        xs = rng.normal(size=64) * scale             # -energy_error of 64 leaves
        w = oracle.w_leaf(xs[0])
        ls = xs[0]
        for x in xs[1:]:
            w = oracle.w_add(w, oracle.w_leaf(x))
            ls = np.logaddexp(ls, x)
            assert 0.0 < w[0] < 200.0                # mantissa stays O(number of leaves)
            assert abs(math.log(w[0]) + w[1] * math.log(2.0) - ls) <= 1e-12 * max(1.0, abs(ls))
    m, e = oracle.w_leaf(-1e300)                     # absurd energy errors are clamped, not propagated
    assert math.isfinite(m) and m > 0 and e < -10**9
    big, small = oracle.w_leaf(5000.0), oracle.w_leaf(-5000.0)
    assert oracle.w_add(big, small) == big and oracle.w_add(small, big) == big   # below 2^-1000 of the total: dropped
    one = oracle.w_leaf(0.0)
    assert one == (1.0, 0) and oracle.w_add(one, one) == (2.0, 0)


def test_welford_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    x = rng.normal(size=(50, 7)) * np.arange(1, 8)
    mean, m2 = oracle.welford(x)
    np.testing.assert_allclose(mean, x.mean(0), rtol=1e-13)
    np.testing.assert_allclose(m2 / (len(x) - 1), x.var(0, ddof=1), rtol=1e-12)
    mean1, m21 = oracle.welford(x[:1])
    assert np.array_equal(mean1, x[0]) and np.all(m21 == 0)


def test_mass_matrix_formula(oracle):
    # After tuning on a diagonal Gaussian the grad-based estimate sqrt(Var q / Var g) equals the posterior
    # variance: g = -q/s^2 => Var g = Var q / s^4 => sqrt(ratio) = s^2 exactly, whatever the draws were.
    # (in-tree corroboration: python/nutpie/normalizing_flow.py:1906-1915)
    sd = np.array([0.01, 0.3, 1.0, 7.0, 120.0])
    s = oracle.default_settings(seed=4, num_chains=2, num_tune=120, num_draws=5, store_mass_matrix=1)
    tr = oracle.sample_tridiag(s, 1 / sd**2)
    mm = tr.stats["mass_matrix_inv"][:, 100, :]
    np.testing.assert_allclose(mm, np.broadcast_to(sd**2, mm.shape), rtol=1e-9)
    # draw_diag variant estimates the plain draw variance -> only statistically close
    s = oracle.default_settings(seed=4, num_chains=2, num_tune=400, num_draws=5, store_mass_matrix=1, use_grad_based_mass_matrix=0)
    tr = oracle.sample_tridiag(s, 1 / sd**2)
    mm = tr.stats["mass_matrix_inv"][:, 330, :]
    assert np.all(np.abs(np.log(mm / sd**2)) < 1.0)
    # the matrix is frozen during the final step-size window (docs/sample-stats.qmd:85-88)
    assert np.array_equal(tr.stats["mass_matrix_inv"][:, 345, :], tr.stats["mass_matrix_inv"][:, 399, :])


def test_initial_mass_matrix_from_gradient(oracle):
    # first matrix: sig2 = 1/|g| at the initial point (normalizing_flow.py:1906-1908: diag = 1/sqrt|g|)
    sd = np.array([0.5, 2.0, 10.0])
    init = np.array([[1.0, -3.0, 20.0]])
    s = oracle.default_settings(seed=1, num_chains=1, num_tune=3, num_draws=1, store_mass_matrix=1, init_kind=2)
    tr = oracle.sample_tridiag(s, 1 / sd**2, init_points=init)
    g0 = -init[0] / sd**2
    np.testing.assert_allclose(tr.stats["mass_matrix_inv"][0, 0], 1 / np.abs(g0), rtol=1e-15)


def test_fixed_step_size_and_schedule(oracle):
    s = oracle.default_settings(seed=2, num_chains=2, num_tune=50, num_draws=20, fixed_step_size=1, initial_step=0.37)
    tr = oracle.sample_tridiag(s, np.ones(4))
    assert np.all(tr.stats["step_size"] == 0.37)
    # tuning flag: exactly num_tune draws are warm-up
    assert np.all(tr.stats["tuning"].sum(1) == 50)
    # last tuning draw switches to the averaged step size and sampling keeps it
    s = oracle.default_settings(seed=2, num_chains=2, num_tune=50, num_draws=20)
    tr = oracle.sample_tridiag(s, np.ones(4))
    assert np.array_equal(tr.stats["step_size"][:, 49], tr.stats["step_size_bar"][:, 49])
    assert np.all(tr.stats["step_size"][:, 50:] == tr.stats["step_size"][:, 49:50])


def test_tree_bookkeeping_invariants(oracle):
    s = oracle.default_settings(seed=3, num_chains=4, num_tune=100, num_draws=100, maxdepth=3)
    tr = oracle.sample_tridiag(s, np.exp(np.random.default_rng(0).normal(size=20)))
    depth, n_steps, div, md = tr.stats["depth"], tr.stats["n_steps"], tr.stats["diverging"], tr.stats["maxdepth_reached"]
    assert depth.max() <= 3
    ok = div == 0
    # a finished doubling sequence of depth d used between 2^(d-1) (turn inside the last sub-tree) ... 2^d - 1 steps,
    # or up to 2^(d+1) - 1 when the last sub-tree was abandoned
    assert np.all(n_steps[ok] >= 2.0 ** depth[ok] - 1)
    assert np.all(n_steps[ok] <= 2.0 ** (depth[ok] + 1) - 1)
    assert np.all(depth[md == 1] == 3) and md.sum() > 0  # a U-turn at the last doubling is not 'maxdepth reached'
    idx = tr.stats["index_in_trajectory"]
    assert np.all(np.abs(idx) <= n_steps)
    # energy bookkeeping: energy_error == energy - H0 and |error| small for accepted points
    assert np.all(np.isfinite(tr.stats["energy"]))
    assert np.all(tr.stats["mean_tree_accept"] >= 0) and np.all(tr.stats["mean_tree_accept"] <= 1)


def test_recoverable_error_is_divergence_and_fatal_raises(oracle, fixture_lib):
    from tests.conftest import fn_addr

    s = oracle.default_settings(seed=8, num_chains=2, num_tune=60, num_draws=60, init_kind=1)
    tr = oracle.sample_callback(s, 3, fn_addr(fixture_lib.failing_logp))
    assert tr.stats["diverging"].sum() > 0          # x0 > 2.5 is refused -> divergences
    assert tr.draws[:, :, 0].max() <= 2.5 + 1e-12   # and never accepted
    with pytest.raises(RuntimeError, match="fatal"):
        oracle.sample_callback(s, 3, fn_addr(fixture_lib.fatal_logp))


def test_crate_arithmetic_forms_give_the_same_decisions(oracle):
    """the oracle carries tree weights as m * 2^e and the acceptance statistic as sum / count (the engine's
    forms).  With the crate's own forms (SURVEY App. A.3 / A.6 verbatim: log_size + logaddexp + exp, running mean) ...
    * bit 0, log-domain weights: EVERY decision and EVERY float of the five golden cases is unchanged — the weight form is
      immaterial (same uniforms, thresholds never within rounding of them);
    * bit 1, running mean: the statistic differs in the last bits, dual averaging passes that on to the step size
      (~1e-16 relative) and the integrator amplifies it; decisions stay identical on four cases and fork at one draw of the
      fifth (ar1_d257) — a rounding fork: the first draws agree to 1e-9."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    keys = ("depth", "n_steps", "index_in_trajectory", "diverging", "maxdepth_reached")
    forks = {}
    for name, (kw, mspec) in mg.CASES.items():
        margs = mg.model_args(mspec)
        base = oracle.sample_tridiag(oracle.default_settings(**kw), **margs)
        logw = oracle.sample_tridiag(oracle.default_settings(crate_arithmetic=1, **kw), **margs)
        for k in keys + ("energy", "step_size", "mean_tree_accept"):
            assert np.array_equal(base.stats[k], logw.stats[k]), (name, k)
        assert np.array_equal(base.draws, logw.draws), name
        run = oracle.sample_tridiag(oracle.default_settings(crate_arithmetic=3, **kw), **margs)
        same = all(np.array_equal(base.stats[k], run.stats[k]) for k in keys)
        if not same:
            bad = np.argwhere(base.stats["n_steps"] != run.stats["n_steps"])
            chain, draw = (int(v) for v in bad[0])
            forks[name] = (chain, draw)
            # a rounding fork, not a different algorithm: the first draws agree to 1e-9 (the perturbation then grows
            # exponentially along the trajectories of the early warm-up — 1e-2 in position by the time a U-turn test flips)
            assert draw > 10
            np.testing.assert_allclose(run.draws[chain, :4], base.draws[chain, :4], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(run.stats["mean_tree_accept"][chain, :4], base.stats["mean_tree_accept"][chain, :4], rtol=1e-9)
        else:
            np.testing.assert_allclose(run.stats["mean_tree_accept"], base.stats["mean_tree_accept"], rtol=1e-4, atol=1e-9)
    assert set(forks) <= {"ar1_d257"}, forks



def test_divergence_records_are_the_failed_leapfrog(oracle):
    # store_divergences (python/nutpie/sample.py:631-650): divergence_start / _momentum / _start_gradient are the state the
    # failed leapfrog started from, divergence_end the position it reached.  Known answer: one leapfrog from the
    # recorded start with the step size and mass matrix the draw ran with lands on the recorded end, bit for bit; the recorded
    # gradient is the model's gradient at the recorded start; rows of draws that did not diverge are NaN.
    diag = np.array([1.0, 100.0, 0.01])
    s = oracle.default_settings(seed=5, num_chains=4, num_tune=150, num_draws=100, max_energy_error=0.3, store_divergences=1, store_mass_matrix=1)
    tr = oracle.sample_tridiag(s, diag)
    div = tr.stats["diverging"].astype(bool)
    assert div.sum() > 20
    start, end, mom, grad = (tr.stats[k] for k in ("divergence_start", "divergence_end", "divergence_momentum", "divergence_start_gradient"))
    for a in (start, end, mom, grad):
        assert np.all(np.isnan(a[~div])) and np.all(np.isfinite(a[div]))
    assert np.array_equal(grad[div], -(diag * start[div]))
    checked = 0
    for c, d in zip(*np.nonzero(div)):
        if d == 0:
            continue   # (the first draw runs with the step size of the initial search, which is not in the trace)
        # the statistics of draw d - 1 are written after its adaptation: they are what draw d ran with
        step, sig2 = tr.stats["step_size"][c, d - 1], tr.stats["mass_matrix_inv"][c, d - 1]
        ends = []
        for eps in (step, -step):   # (extended precision stands in for the fused multiply-adds: compared to 1 ulp)
            L = np.longdouble
            ph = (L(0.5 * eps) * grad[c, d].astype(L) + mom[c, d].astype(L)).astype(np.float64)
            ends.append((L(eps) * (sig2 * ph).astype(L) + start[c, d].astype(L)).astype(np.float64))
        assert any(np.allclose(e, end[c, d], rtol=3e-16, atol=0) for e in ends), (c, d)
        checked += 1
    assert checked > 10
    # the records do not change anything else: same trace without them
    s0 = oracle.default_settings(seed=5, num_chains=4, num_tune=150, num_draws=100, max_energy_error=0.3)
    tr0 = oracle.sample_tridiag(s0, diag)
    assert np.array_equal(tr0.draws, tr.draws) and np.array_equal(tr0.stats["n_steps"], tr.stats["n_steps"])


def _low_rank_parts(cov, k):
    """(sigma^2, V rows, lambda) with D^1/2 (I + V (Lambda - I) V') D^1/2 == cov, keeping the k eigenvalues of the correlation
    matrix furthest from 1 (all of them when k == dim)."""
    sd = np.sqrt(np.diag(cov))
    lam, U = np.linalg.eigh(cov / np.outer(sd, sd))
    keep = np.argsort(-np.abs(np.log(lam)))[:k]
    return sd**2, U[:, keep].T.copy(), lam[keep]


def test_low_rank_velocity_is_the_metric_applied(oracle):
    # v = M^-1 p with M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 (src/wrapper.rs:307-334) against dense numpy algebra
    rng = np.random.default_rng(0)
    D = 37
    A = rng.normal(size=(D, D))
    cov = A @ A.T / D + np.diag(np.exp(rng.normal(size=D)))
    sig2, V, lam = _low_rank_parts(cov, D)              # all directions: M^-1 == cov
    p = rng.normal(size=D)
    for W in (1, 2):
        np.testing.assert_allclose(oracle.lr_velocity(sig2, V, lam, p, W), cov @ p, rtol=1e-11)
    sig2, V, lam = _low_rank_parts(cov, 5)
    Minv = np.sqrt(sig2)[:, None] * (np.eye(D) + V.T @ np.diag(lam - 1) @ V) * np.sqrt(sig2)[None, :]
    np.testing.assert_allclose(oracle.lr_velocity(sig2, V, lam, p), Minv @ p, rtol=1e-11)
    np.testing.assert_array_equal(oracle.lr_velocity(sig2, V[:0], lam[:0], p), np.sqrt(sig2) * (np.sqrt(sig2) * p))   # k = 0: the diagonal


def test_sampler_under_a_handed_in_low_rank_metric(oracle):
    # A correlated Gaussian whose exact covariance is handed in as the metric before draw 20: NUTS then sees a standard normal —
    # short trees, the right moments, momenta drawn from N(0, M) (kinetic energy ~ D / 2) — while the diagonal run needs long ones.
    rng = np.random.default_rng(3)
    D, chains, tune, draws = 12, 6, 150, 400
    B = rng.normal(size=(D, 2))
    cov = np.diag(np.exp(rng.normal(size=D))) + 60.0 * B @ B.T
    P = np.linalg.inv(cov)

    def logp(x):
        g = -(P @ x)
        return 0.5 * float(x @ g), g

    sig2, V, lam = _low_rank_parts(cov, D)
    kw = dict(seed=5, num_chains=chains, num_tune=tune, num_draws=draws, n_threads=6)
    s = oracle.default_settings(**kw)
    s.set_metric_schedule([20], np.tile(sig2, (1, chains, 1)), np.tile(V, (1, chains, 1, 1)), np.tile(lam, (1, chains, 1)))
    lr = oracle.sample_callback(s, D, logp)
    dg = oracle.sample_callback(oracle.default_settings(**kw), D, logp)
    assert np.array_equal(lr.draws[:, :20], dg.draws[:, :20])            # before the update: the same chain
    assert lr.stats["n_steps"][:, tune:].mean() < 8 < 3 * 8 < dg.stats["n_steps"][:, tune:].mean()
    x = lr.draws[:, tune:].reshape(-1, D)
    sd = np.sqrt(np.diag(cov))
    assert np.abs(x.mean(0) / sd).max() < 0.12 and np.abs(np.cov(x.T) / cov - 1)[np.abs(cov) > 0.3 * np.outer(sd, sd)].max() < 0.35
    # energy - potential at the start of a draw is the kinetic energy of the fresh momentum: p' M^-1 p / 2 ~ chi^2_D / 2
    K0 = (lr.stats["energy"] - lr.stats["energy_error"] + lr.stats["logp"] * 0)[:, tune:]   # H0 of each draw
    U0 = -np.concatenate([lr.stats["logp"][:, tune - 1:tune], lr.stats["logp"][:, tune:-1]], 1)
    assert abs((K0 - U0).mean() - D / 2) < 0.4
    assert lr.stats["diverging"][:, 25:].sum() == 0


def test_dense_model_is_the_gaussian_and_its_order_is_the_contract(oracle):
    """oracle.dense_grad: -P (x - mu) with ONE fma chain per output in the order k = k0 + 4 t + s (k0 step 16, s = 0..3, t = 0..3) — the
    order in which the engine's fp64 matrix-core tile consumes a row (nutpie_amd/csrc/dense_tile.h) — and logp = 1/2 (x - mu) . grad in
    the engine's summation geometry."""
    import math

    from nutpie_amd.gaussian import dense_precision

    rng = np.random.default_rng(0)
    for dim in (1, 3, 16, 37, 100):
        P = dense_precision(dim, seed=dim)
        mu = rng.normal(size=dim)
        x = rng.normal(size=(4, dim))
        g, lp = oracle.dense_grad(x, P, mu, waves=1)
        np.testing.assert_allclose(g, -(x - mu) @ P, rtol=0, atol=1e-12 * np.abs(P).max() * dim)
        np.testing.assert_allclose(lp, -0.5 * np.einsum("ij,jk,ik->i", x - mu, P, x - mu), rtol=1e-12)
        # the order, restated in Python on row 0
        z = x[0] - mu
        for j in (0, dim - 1):
            acc = 0.0
            for k0 in range(0, dim, 16):
                for s in range(4):
                    for t in range(4):
                        k = k0 + 4 * t + s
                        if k < dim:
                            acc = math.fma(z[k], P[j, k], acc) if hasattr(math, "fma") else float(np.float64(np.longdouble(z[k]) * np.longdouble(P[j, k]) + np.longdouble(acc)))
            if hasattr(math, "fma"):
                assert g[0, j] == -acc
            else:
                assert abs(g[0, j] + acc) <= 2e-16 * max(1.0, abs(acc)) * dim
        assert lp[0] == 0.5 * oracle.dot(z, g[0], waves=1)


def test_dense_sampler_recovers_the_covariance(oracle):
    from nutpie_amd.gaussian import dense_precision

    dim = 6
    P = dense_precision(dim, seed=9, cond_lo=0.5, cond_hi=2)
    tr = oracle.sample_dense(oracle.default_settings(seed=4, num_chains=16, num_tune=300, num_draws=500, n_threads=8), P)
    x = tr.draws[:, 300:].reshape(-1, dim)
    cov = np.linalg.inv(P)
    np.testing.assert_allclose(np.cov(x.T), cov, atol=0.12 * np.abs(cov).max())
    assert not tr.stats["diverging"][:, 300:].any()
