"""adaptation="low_rank" (SURVEY.md §8f N4), the host-side mathematics on CPU tensors: the linear re-parametrisation and the
batched estimator of nutpie_amd/low_rank.py.  The engine side (pause / resume hook, end-to-end sampling) is in the GPU tests."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _random_transform(n, D, k, seed=0):
    from nutpie_amd import low_rank as lr

    g = torch.Generator().manual_seed(seed)
    V = torch.linalg.qr(torch.randn(n, D, k, generator=g, dtype=torch.float64))[0]
    V = torch.cat([V, torch.zeros(n, D, lr.K_MAX - k, dtype=torch.float64)], 2)
    lam = torch.exp(torch.randn(n, k, generator=g, dtype=torch.float64) * 1.5)
    d = torch.cat([lam.sqrt() - 1, torch.zeros(n, lr.K_MAX - k, dtype=torch.float64)], 1)
    return lr.Transform(torch.randn(n, D, generator=g, dtype=torch.float64), torch.exp(torch.randn(n, D, generator=g, dtype=torch.float64)), V, d)


def test_transform_is_a_consistent_linear_map():
    n, D = 3, 40
    T = _random_transform(n, D, 5)
    g = torch.Generator().manual_seed(1)
    y = torch.randn(n, D, generator=g, dtype=torch.float64)
    x = T.forward(y)
    assert torch.allclose(T.inverse(x), y, rtol=1e-11, atol=1e-11)
    ys = torch.randn(n, 7, D, generator=g, dtype=torch.float64)          # [chain, draw, dim] form
    assert torch.allclose(T.forward(ys)[:, 3], T.forward(ys[:, 3]), rtol=1e-13, atol=1e-13)
    assert torch.allclose(T.inverse(T.forward(ys)), ys, rtol=1e-10, atol=1e-10)
    # grad_to_y is the adjoint of the linear part: <L dy, gx> == <dy, L' gx>; grad_to_x inverts it
    dy, gx = torch.randn(n, D, generator=g, dtype=torch.float64), torch.randn(n, D, generator=g, dtype=torch.float64)
    Ldy = T.forward(dy) - T.forward(torch.zeros_like(dy))
    assert torch.allclose((Ldy * gx).sum(1), (dy * T.grad_to_y(gx)).sum(1), rtol=1e-11)
    assert torch.allclose(T.grad_to_x(T.grad_to_y(gx)), gx, rtol=1e-10, atol=1e-10)
    # chain rule on a real density: logp_y(y) = logp_x(forward(y))
    A = torch.randn(n, D, D, generator=g, dtype=torch.float64)
    P = A @ A.transpose(1, 2) + torch.eye(D, dtype=torch.float64)
    yy = y.clone().requires_grad_(True)
    xx = T.forward(yy)
    lp = -0.5 * torch.einsum("nd,nde,ne->n", xx, P, xx)
    (auto,) = torch.autograd.grad(lp.sum(), yy)
    gx_exact = -torch.einsum("nde,ne->nd", P, xx.detach())
    assert torch.allclose(T.grad_to_y(gx_exact), auto, rtol=1e-9, atol=1e-9)


def test_estimator_whitens_a_correlated_gaussian():
    """Draws of N(0, Sigma) with their gradients -Sigma^-1 x: in the estimated coordinates the covariance of the window is
    close to the identity along the directions the window spans — the strong correlation is gone."""
    from nutpie_amd import low_rank as lr

    n, D, m = 4, 30, 48
    g = torch.Generator().manual_seed(3)
    # a few dominant correlated directions on top of heterogeneous scales
    B = torch.randn(n, D, 3, generator=g, dtype=torch.float64)
    scales = torch.exp(torch.randn(n, D, generator=g, dtype=torch.float64))
    Sigma = torch.diag_embed(scales ** 2) + 25.0 * (scales[:, :, None] * B) @ (scales[:, :, None] * B).transpose(1, 2)
    Lc = torch.linalg.cholesky(Sigma)
    x = torch.einsum("nde,nme->nmd", Lc, torch.randn(n, m, D, generator=g, dtype=torch.float64)) + 2.0
    gx = -torch.linalg.solve(Sigma, (x - 2.0).transpose(1, 2)).transpose(1, 2)
    T = lr.estimate(x, gx, gamma=1e-5, cutoff=2.0)
    used = (T.d != 0).sum(1)
    assert (used >= 3).all() and (used <= lr.K_MAX).all()
    # metric implied by the transform vs the true covariance: condition number of Sigma in the new coordinates
    eye = torch.eye(D, dtype=torch.float64).expand(n, D, D)
    Lmat = torch.stack([T.forward(eye[:, :, j]) - T.forward(torch.zeros(n, D, dtype=torch.float64)) for j in range(D)], 2)   # columns L e_j
    C = torch.linalg.solve(Lmat, torch.linalg.solve(Lmat, Sigma).transpose(1, 2))       # L^-1 Sigma L^-T
    ev = torch.linalg.eigvalsh(0.5 * (C + C.transpose(1, 2)))
    ev0 = torch.linalg.eigvalsh(Sigma / (scales[:, :, None] * scales[:, None, :]))        # what a perfect DIAGONAL metric leaves
    assert ((ev[:, -1] / ev[:, 0]) < 0.2 * (ev0[:, -1] / ev0[:, 0])).all()
    assert ((ev[:, -1] / ev[:, 0]) < 30).all()


def test_estimator_degenerate_windows_fall_back_to_the_diagonal():
    from nutpie_amd import low_rank as lr

    n, D, m = 2, 10, 8
    x = torch.zeros(n, m, D, dtype=torch.float64)           # a chain that did not move: no direction, unit scales
    T = lr.estimate(x, x.clone(), gamma=1e-5, cutoff=2.0)
    assert torch.isfinite(T.stds).all() and (T.d == 0).all() and torch.isfinite(T.V).all()
    y = torch.randn(n, D, dtype=torch.float64)
    assert torch.allclose(T.inverse(T.forward(y)), y)
    # isotropic draws: nothing outside [1/cutoff, cutoff]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 40, D, generator=g, dtype=torch.float64)
    T = lr.estimate(x, -x, gamma=1e-5, cutoff=4.0)
    assert (T.d == 0).all()


def test_window_schedule_follows_the_reference_keys():
    """src/wrapper.rs:198-240: `window_switch_freq`, `early_window_switch_freq` (and `early_window`, `step_size_window`) are set on Diag
    AND LowRank settings — the low-rank estimator runs on the diagonal adaptation's foreground / background windows.  The hand-ins
    are the main-phase switches and thinned refreshes; a window is the foreground: the draws since the second-last switch."""
    from nutpie_amd import low_rank as lr

    # tune 400, defaults: early phase until draw 120 (switches every 10 draws: 9, 19, ... 119) — ONE hand-in there, at its last switch,
    # from its second half; one main-phase switch at 199 (the next, 279, could not collect 80 more draws before the final window at
    # 341), its window the foreground = the draws since the second-last switch; refreshes every >= 40 draws, the last at 340
    assert lr.window_schedule(400) == [(119, 59), (199, 123), (240, 119), (280, 119), (320, 119), (340, 119)]
    assert lr.pause_draws(400) == [119, 199, 240, 280, 320, 340]
    assert lr.window_schedule(400, early_hand_ins=False) == [(199, 119), (240, 119), (280, 119), (320, 119), (340, 119)]
    # the keys move it: more frequent main switches -> shorter, later windows
    s = lr.window_schedule(400, switch_freq=50, early_switch_freq=20, early_hand_ins=False)
    assert [p for p, _ in s][:3] == [169, 200, 219] and s[2] == (219, 169) and s[-1][0] == 340
    # early_window / step_size_window: where the main phase starts and the metric freezes
    s = lr.window_schedule(1000, early_window=0.1, step_size_window=0.3)
    assert s[0] == (99, 49) and s[1][1] >= 99 and s[-1][0] <= 700 and all(b > a for (a, _), (b, _) in zip(s, s[1:]))
    for T in (30, 100, 200, 400, 1000, 4000):
        s = lr.window_schedule(T)
        assert 1 <= len(s) <= lr.MAX_HAND_INS and all(lr.MIN_WINDOW <= p - a and 0 <= a < p < T for p, a in s)
    assert lr.window_schedule(10) == []


def test_schedule_of_takes_the_update_frequency_as_set():
    """``mass_matrix_update_freq`` (src/wrapper.rs:307-334): LowRank settings carry its low-rank default (10), and a value the user sets —
    1 included — is the schedule's (an explicit 1 used to be read as "unset")."""
    from nutpie_amd import _lib as hip
    from nutpie_amd import low_rank as lr

    s = hip.PyNutsSettings.LowRank(1)
    s.update(num_tune=400)
    assert s.mass_matrix_update_freq == 10
    assert lr.schedule_of(s) == lr.window_schedule(400) == lr.window_schedule(400, update_freq=10)
    s.mass_matrix_update_freq = 1
    assert lr.schedule_of(s) == lr.window_schedule(400, update_freq=1)
    s.mass_matrix_update_freq = 25
    assert lr.schedule_of(s) == lr.window_schedule(400, update_freq=25)
    assert hip.PyNutsSettings.Diag(1).mass_matrix_update_freq == 1


def test_estimator_uses_every_draw_for_the_diagonal_and_a_thinned_basis_for_the_columns():
    from nutpie_amd import low_rank as lr

    g = torch.Generator().manual_seed(2)
    n, m, D = 2, 120, 90
    scale = torch.exp(torch.randn(D, generator=g, dtype=torch.float64))
    z = torch.randn(n, m, D, generator=g, dtype=torch.float64)
    z[:, :, 0] = z[:, :, 1] * 0.98 + 0.2 * z[:, :, 0]                 # one strongly correlated pair
    x = z * scale
    P = torch.eye(D, dtype=torch.float64)
    gx = -(x / scale**2)
    full = lr.estimate(x, gx, 1e-5, 2.0)
    thin = lr.estimate(x, gx, 1e-5, 2.0, basis_draws=32)
    # the diagonal part before re-centring is the same ratio of standard deviations over all 120 draws: the thinned estimate differs
    # from the full one by its re-centring factor only
    ratio = thin.stds / full.stds
    assert torch.allclose(ratio, ratio[:, :1].expand_as(ratio), rtol=1e-10)
    assert (thin.d != 0).sum() <= (full.d != 0).sum() + 4 and torch.isfinite(thin.V).all()
    y = torch.randn(n, D, generator=g, dtype=torch.float64)
    assert torch.allclose(thin.inverse(thin.forward(y)), y, atol=1e-9)


@pytest.mark.parametrize("n,D,m,seed,gamma,cutoff,tol", [(4, 30, 48, 3, 1e-5, 2.0, 1e-9), (2, 60, 64, 7, 1e-3, 4.0, 1e-9), (3, 120, 40, 5, 1e-5, 2.0, 1e-4)])
def test_estimator_against_its_cpu_oracle(n, D, m, seed, gamma, cutoff, tol):
    """the window estimator had property tests only.  oracle/low_rank_estimator.py restates it independently
    (numpy, one chain at a time, SVD for the subspace, scipy's matrix square root for the geometric mean); the two are compared
    through the dense metric M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 — invariant to the order, sign and basis of the columns —
    and through the number of columns kept.  (Third case: a window that spans fewer directions than the model has, sixteen
    columns kept: the cap cuts through nearly equal eigenvalues, hence the looser tolerance.)"""
    from nutpie_amd import low_rank as lr
    from oracle.low_rank_estimator import dense_metric, estimate_chain

    g = torch.Generator().manual_seed(seed)
    B = torch.randn(n, D, 3, generator=g, dtype=torch.float64)
    scales = torch.exp(torch.randn(n, D, generator=g, dtype=torch.float64))
    Sigma = torch.diag_embed(scales ** 2) + 25.0 * (scales[:, :, None] * B) @ (scales[:, :, None] * B).transpose(1, 2)
    x = torch.einsum("nde,nme->nmd", torch.linalg.cholesky(Sigma), torch.randn(n, m, D, generator=g, dtype=torch.float64)) + 2.0
    gx = -torch.linalg.solve(Sigma, (x - 2.0).transpose(1, 2)).transpose(1, 2)
    sig2, V, lam = lr.metric_of(lr.estimate(x, gx, gamma=gamma, cutoff=cutoff))
    for c in range(n):
        s2, Vc, lc = estimate_chain(x[c].numpy(), gx[c].numpy(), gamma, cutoff)
        assert len(lc) == int((lam[c] != 1).sum()) >= 3
        want = dense_metric(s2, Vc, lc)
        got = dense_metric(sig2[c].numpy(), V[c].numpy().T, lam[c].numpy())
        assert np.abs(got - want).max() <= tol * np.abs(want).max()
        # (and the metric does what it is for: in its coordinates the window's covariance is within the cutoff of the identity
        #  along the directions the window spans — the property the older test checks on the engine's version alone)
        np.testing.assert_allclose(np.sort(lc), np.sort(lam[c].numpy()[lam[c].numpy() != 1]), rtol=max(tol, 1e-9) * 100)


class _FakeEngine:
    """What LowRankSampler needs of a manual-mode PySampler, on the CPU: chains that advance at their own speed, stop at the
    pause draws, and go on when `set_metric` names them.  Draws and gradients are fixed random arrays, so the metric a chain
    must receive at a boundary can be recomputed from its own window."""

    def __init__(self, n, total, dim, pauses, speed, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.num_chains, self.total_draws, self.dim = n, total, dim
        self.launches_per_look = 1                       # (a launch of this engine is many draws, like a resident kernel's)
        self.draws = torch.randn(n, total, dim, generator=g, dtype=torch.float64)
        self.grads = -self.draws * torch.exp(torch.randn(n, 1, dim, generator=g, dtype=torch.float64)) + 0.1 * torch.randn(n, total, dim, generator=g, dtype=torch.float64)
        self.pauses, self.speed = list(pauses), np.asarray(speed)
        self.at = np.zeros(n, dtype=np.int64)            # finished draws
        self.waiting = np.zeros(n, dtype=bool)
        self.passed = np.zeros(n, dtype=np.int64)        # boundaries this chain has been resumed at
        self.log = []                                    # (step number, chain, boundary draw, sigma2 row, k)
        self.steps = 0

    def step(self, n_launches=1):
        import time

        for _ in range(int(n_launches)):
            time.sleep(0.002)                            # (a launch takes a while: the estimates run meanwhile, on their thread)
            self.steps += 1
            for c in range(self.num_chains):
                if self.waiting[c] or self.at[c] >= self.total_draws:
                    continue
                stop = self.pauses[self.passed[c]] if self.passed[c] < len(self.pauses) else self.total_draws
                self.at[c] = min(self.at[c] + self.speed[c], stop, self.total_draws)
                if self.at[c] == stop and stop < self.total_draws:
                    self.waiting[c] = True
        return bool((self.at >= self.total_draws).all()), int(n_launches), 10.0 * n_launches

    def waiting_codes(self):
        return np.where(self.at >= self.total_draws, 2, np.where(self.waiting, 1, 0)).astype(np.uint8)

    def set_metric(self, chains, sig2, V, lam):
        chains = np.asarray(chains)
        assert self.waiting[chains].all(), "a metric for a chain that has not stopped"
        assert sig2.shape == (len(chains), self.dim) and (V is None or V.shape[0] == len(chains))
        for r, c in enumerate(chains):
            self.log.append((self.steps, int(c), int(self.at[c]), sig2[r].clone(), 0 if V is None else V.shape[1]))
            self.passed[c] += 1
            self.waiting[c] = False


def test_driver_hands_every_chain_its_own_windows_without_lock_step():
    """LowRankSampler._run: every chain receives one metric per boundary, in order, estimated from ITS window of draws and
    gradients — and a chain that crawls does not hold the others (they are handed in, and finish, long before it arrives)."""
    from nutpie_amd import low_rank as lr

    n, total, dim = 6, 240, 8
    schedule = [(60, 20), (120, 60), (180, 60)]
    pauses = [p for p, _ in schedule]
    eng = _FakeEngine(n, total, dim, pauses, speed=[1, 40, 40, 30, 40, 20])

    class Driver(lr.LowRankSampler):
        def _views(self):
            return self._inner.draws, self._inner.grads

        def _n_steps(self):
            return torch.ones(n, total, dtype=torch.int64)

    smp = Driver(eng, 0, 1e-5, 2.0, schedule)
    smp.wait(timeout_seconds=120)
    assert smp.is_finished() and (eng.at == total).all()
    by_chain = {c: [e for e in eng.log if e[1] == c] for c in range(n)}
    for c, entries in by_chain.items():
        assert [e[2] for e in entries] == pauses, f"chain {c}: boundaries {[e[2] for e in entries]}"
        for i, (_, _, hi, sig2, _) in enumerate(entries):
            lo = schedule[i][1]
            T = lr.estimate(eng.draws[c:c + 1, lo:hi], eng.grads[c:c + 1, lo:hi], 1e-5, 2.0, basis_draws=lr.basis_draws_for(dim))
            assert torch.allclose(sig2, (T.stds * T.stds)[0], rtol=1e-12, atol=0), f"chain {c}, boundary {hi}: not its own window"
    # the crawling chain reaches its first boundary after every other chain has passed its last one
    first_slow = by_chain[0][0][0]
    assert all(by_chain[c][-1][0] < first_slow for c in range(1, n))
    # the hand-ins are batched: fewer estimates than (chain, boundary) pairs
    assert len(smp.switch_log) < n * len(pauses)
    assert sum(e[3] for e in smp.switch_log) == n * len(pauses)


def test_estimator_ignores_a_window_with_non_finite_entries():
    """One chain's window holds an inf: that chain gets the identity (sigma = 1, no columns), the others are not touched."""
    from nutpie_amd import low_rank as lr

    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 20, 7, generator=g, dtype=torch.float64)
    gx = -x + 0.1 * torch.randn(3, 20, 7, generator=g, dtype=torch.float64)
    want = lr.estimate(x, gx, 1e-5, 2.0)
    bad = gx.clone()
    bad[1, 4, 2] = float("inf")
    got = lr.estimate(x, bad, 1e-5, 2.0)
    for a, b in ((got.stds, want.stds), (got.V, want.V), (got.d, want.d)):
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    assert torch.equal(got.stds[1], torch.ones(7, dtype=torch.float64)) and not got.d[1].any() and not got.V[1].any()
    sig2, V, lam = lr.metric_of(got)
    assert torch.isfinite(sig2).all() and torch.isfinite(V).all() and torch.isfinite(lam).all()


def test_basis_pick_is_what_estimate_thins_to():
    from nutpie_amd import low_rank as lr

    assert np.array_equal(lr.basis_pick(20, 32), np.arange(20)) and np.array_equal(lr.basis_pick(40, None), np.arange(40))
    p = lr.basis_pick(100, 32)
    assert len(p) == 32 and p[0] == 0 and p[-1] == 99 and (np.diff(p) >= 3).all()      # evenly thinned, ends included
    # (the rule of the estimator kernel's C-ABI: rint(t (m - 1) / (b - 1)) — numpy's linspace + round-half-even)
    assert np.array_equal(p, np.rint(np.arange(32) * (99 / 31)).astype(np.int64))


def test_estimate_window_on_the_cpu_is_estimate_on_the_window():
    """estimate_window (the driver's entry: the kernel on the GPU) falls back to estimate() + metric_of() where there is no kernel."""
    from nutpie_amd import low_rank as lr

    g = torch.Generator().manual_seed(2)
    x = torch.randn(5, 90, 12, generator=g, dtype=torch.float64)
    x[:, :, 0] *= 20.0
    gx = -x * torch.exp(torch.randn(12, generator=g, dtype=torch.float64))
    s2, V, lam = lr.estimate_window(x, gx, [3, 1], 10, 70, 1e-5, 2.0, basis_draws=32)
    want = lr.metric_of(lr.estimate(x[[3, 1], 10:70], gx[[3, 1], 10:70], 1e-5, 2.0, basis_draws=32))
    for a, b in zip((s2, V, lam), want):
        assert torch.equal(a, b)


def test_driver_releases_chains_that_need_no_low_rank_part():
    """A chain on its own diagonal metric whose window shows nothing outside the cutoff is RELEASED (it goes on adapting that metric
    itself); a chain that needs columns is handed the metric, and from then on is handed one at every boundary."""
    from nutpie_amd import low_rank as lr

    n, total, dim = 4, 200, 6
    schedule = [(60, 20), (120, 60)]
    pauses = [p for p, _ in schedule]

    class Engine(_FakeEngine):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.released = []

        def release(self, chains):
            chains = np.asarray(chains)
            assert self.waiting[chains].all()
            for c in chains:
                self.released.append((int(c), int(self.at[c])))
                self.passed[c] += 1
                self.waiting[c] = False

    eng = Engine(n, total, dim, pauses, speed=[30, 30, 20, 30])
    # chain 0: two dimensions correlated at 0.995 — after the diagonal scaling one direction stays far outside the cutoff in every window
    Sigma = torch.eye(dim, dtype=torch.float64)
    Sigma[0, 1] = Sigma[1, 0] = 0.995
    z = torch.randn(total, dim, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    eng.draws[0] = z @ torch.linalg.cholesky(Sigma).T
    eng.grads[0] = -torch.linalg.solve(Sigma, eng.draws[0].T).T

    class Driver(lr.LowRankSampler):
        def _views(self):
            return self._inner.draws, self._inner.grads

        def _n_steps(self):
            return torch.ones(n, total, dtype=torch.int64)

    smp = Driver(eng, 0, 1e-5, 2.0, schedule)
    smp.wait(timeout_seconds=120)
    handed = {c for _, c, _, _, _ in eng.log}
    rel = {c for c, _ in eng.released}
    assert smp.is_finished() and handed | rel == set(range(n))
    for c in range(n):
        k_at = [lr.metric_of(lr.estimate(eng.draws[c:c + 1, lo:hi], eng.grads[c:c + 1, lo:hi], 1e-5, 2.0, basis_draws=lr.basis_draws_for(dim)))[2] for hi, lo in schedule]
        needs = [bool((k != 1.0).any()) for k in k_at]
        got_metric = sorted(at for _, cc, at, _, _ in eng.log if cc == c)
        got_release = sorted(at for cc, at in eng.released if cc == c)
        # handed a metric from the first boundary that needs one on; released before that
        first = needs.index(True) if True in needs else len(needs)
        assert got_release == pauses[:first] and got_metric == pauses[first:], (c, needs, got_release, got_metric)
    assert 0 in handed       # (the test is not vacuous: chain 0 needs columns)
