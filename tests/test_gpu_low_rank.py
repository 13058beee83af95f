"""The low-rank metric M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 inside the engine (reference: adaptation="low_rank",
src/wrapper.rs:307-334, python/nutpie/sample.py:921-933): leapfrog, momentum draw, kinetic energy and U-turn criteria under
the metric, bit for bit against the oracle, with the metrics handed in at the pause draws (nphip_sampler_set_metric)."""
import numpy as np
import pytest

from nutpie_amd.gaussian import ar1_gaussian
from tests.conftest import assert_trace_equal, fn_addr
from tests.test_gpu_parity import oracle_settings

pytestmark = pytest.mark.gpu


def random_metrics(rng, updates, chains, dim, k):
    sig2 = np.exp(rng.normal(size=(updates, chains, dim)))
    V = np.zeros((updates, chains, k, dim))
    for u in range(updates):
        for c in range(chains):
            q, _ = np.linalg.qr(rng.normal(size=(dim, k)))
            V[u, c] = q.T
    lam = np.exp(rng.uniform(np.log(0.15), np.log(8.0), size=(updates, chains, k)))
    return sig2, V, lam


def run_engine_with_metrics(hip, model, pauses, sig2, V, lam, *, chains, tune, draws, seed, waves=0, launch=None, **settings):
    s = hip.PyNutsSettings.Diag(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, low_rank_metric=True, **settings)
    s.set_pause_draws(pauses)
    smp = hip.PySampler(s, model, waves_per_chain=waves, manual=True, **(launch or {}))
    nxt = 0
    for _ in range(100000):
        done, _, _ = smp.step(4)
        if done:
            break
        if nxt < len(pauses) and smp.waiting().all():
            k = V.shape[2]
            smp.set_metric(np.arange(chains), sig2[nxt], V[nxt] if k else None, lam[nxt] if k else None)
            nxt += 1
    assert done and nxt == len(pauses)
    W = smp.waves_per_chain
    return smp.take_results(), W


@pytest.mark.parametrize("dim,waves,k,launch", [
    (24, 0, 3, {}),                                 # one wave per chain: the REGISTER-RESIDENT leaf under the metric (round 4)
    (24, 0, 0, {}),                                 # k = 0: a host-supplied DIAGONAL metric (v = std (std p))
    (24, 0, 3, dict(no_register_kernel=True)),      # ... and the memory-resident kernels on the same job
    (300, 0, 11, dict(evals_per_launch=9)),         # more than 8 columns: two reductions; launch boundaries every 9 leapfrogs
    (300, 0, 11, dict(no_register_kernel=True)),
    (300, 2, 16, {}),
    (1000, 0, 4, dict(evals_per_launch=37)),        # eight chunks per lane in registers (+ the velocity: the full register file)
    (1300, 0, 4, {}),                               # two waves per chain, six chunks each: the register-resident leaf (launch_fam_rw_lr)
    (1300, 0, 4, dict(no_register_kernel=True)),    # ... and the memory-resident kernels on the same job
    (2500, 0, 9, dict(evals_per_launch=21)),        # four waves per chain, five chunks each, register-resident; two reductions
    (5003, 0, 2, dict(evals_per_launch=13)),        # four waves per chain (memory-resident kernels under the low-rank metric)
])
def test_fused_model_under_handed_in_metrics_bit_identical(hip, oracle, dim, waves, k, launch):
    rng = np.random.default_rng(dim + k)
    model = ar1_gaussian(dim) if dim > 24 else None
    args = (model.diag, model.offdiag) if model else (np.exp(rng.normal(size=dim) * 1.5),)
    chains, tune, draws, pauses = 3, 40, 10, [12, 27]
    sig2, V, lam = random_metrics(rng, 2, chains, dim, k)
    sig2 *= 0.05 if dim > 24 else 1.0
    kw = dict(chains=chains, tune=tune, draws=draws, seed=dim + 5)
    got, W = run_engine_with_metrics(hip, hip.TridiagGaussianModel(*args), pauses, sig2, V, lam, waves=waves, launch=launch, store_mass_matrix=True, **kw)
    s = oracle_settings(oracle, W=W, store_mass_matrix=True, **kw)
    s.set_metric_schedule(pauses, sig2, V if k else None, lam if k else None)
    want = oracle.sample_tridiag(s, *args)
    assert_trace_equal(got, want)
    assert np.array_equal(got.stats["mass_matrix_inv"], want.stats["mass_matrix_inv"])
    assert np.array_equal(got.stats["mass_matrix_inv"][:, 30], sig2[1])              # the diagonal part is the handed-in sigma^2, kept


def test_job_without_updates_equals_the_diagonal_job(hip, oracle):
    # the setting alone changes nothing: until the first metric arrives a chain runs on the diagonal metric it adapts itself
    diag = np.exp(np.random.default_rng(1).normal(size=40))
    kw = dict(chains=4, tune=60, draws=20, seed=8)
    s = hip.PyNutsSettings.Diag(8)
    s.update(num_tune=60, num_draws=20, num_chains=4, low_rank_metric=True)
    smp = hip.PySampler(s, hip.TridiagGaussianModel(diag), manual=True)
    while not smp.step(16)[0]:
        pass
    got = smp.take_results()
    want = oracle.sample_tridiag(oracle_settings(oracle, W=1, **kw), diag)
    assert_trace_equal(got, want)


def test_host_callback_models_under_handed_in_metrics(hip, oracle, fixture_lib):
    fn = fn_addr(fixture_lib.eight_schools_logp)
    rng = np.random.default_rng(4)
    chains, tune, draws, pauses = 8, 60, 20, [10, 35]
    sig2, V, lam = random_metrics(rng, 2, chains, 10, 3)
    kw = dict(chains=chains, tune=tune, draws=draws, seed=21)
    got, W = run_engine_with_metrics(hip, hip.HostCallbackModel(10, fn), pauses, sig2, V, lam, **kw)
    s = oracle_settings(oracle, W=W, **kw)
    s.set_metric_schedule(pauses, sig2, V, lam)
    want = oracle.sample_callback(s, 10, fn)
    assert_trace_equal(got, want)


def test_set_metric_errors(hip):
    s = hip.PyNutsSettings.Diag(1)
    s.update(num_tune=20, num_draws=5, num_chains=2)
    smp = hip.PySampler(s, hip.TridiagGaussianModel(np.ones(4)), manual=True)
    with pytest.raises(RuntimeError, match="low_rank_metric"):
        smp.set_metric([0], np.ones((1, 4)))
    smp.close()
    s.update(low_rank_metric=True)
    smp = hip.PySampler(s, hip.TridiagGaussianModel(np.ones(4)), manual=True)
    with pytest.raises(RuntimeError, match="at most 16"):
        smp.set_metric([0], np.ones((1, 4)), np.zeros((1, 17, 4)), np.ones((1, 17)))
    with pytest.raises(RuntimeError, match="2 of 2 chains were not stopped"):
        smp.set_metric([0, 1], np.ones((2, 4)))      # nobody is waiting: nothing changes, and the caller is told
    while not smp.step(8)[0]:
        pass
    smp.close()


def _correlated_gaussian(D, n_dir, factor, seed):
    rng = np.random.default_rng(seed)
    scales = np.exp(rng.normal(size=D))
    B = rng.normal(size=(D, n_dir))
    Sigma = np.diag(scales**2) + factor * (scales[:, None] * B) @ (scales[:, None] * B).T
    return Sigma, np.linalg.inv(Sigma), rng.normal(size=D) * 3


def test_low_rank_adaptation_end_to_end_on_a_host_callback_model(hip):
    """adaptation="low_rank" through sample() on a raw-callback model (round 2 could only do batched torch densities): a
    30-dimensional Gaussian with two strong correlated directions — the same posterior as "diag" with several times fewer
    leapfrogs per draw (reference docs/sampling-options.qmd:124-144)."""
    import nutpie_amd

    D = 30
    Sigma, P, mu = _correlated_gaussian(D, 2, 200.0, 7)

    def logp(x):
        g = -(P @ (x - mu))
        return 0.5 * float((x - mu) @ g), g

    m = nutpie_amd.from_pyfunc(D, lambda: logp, lambda *a: (lambda x: {"x": x}), [np.float64], [(D,)], ["x"])
    kw = dict(chains=16, tune=400, draws=200, seed=3, progress_bar=False)
    lr = nutpie_amd.sample(m, adaptation="low_rank", **kw)
    dg = nutpie_amd.sample(m, adaptation="diag", **kw)
    steps_lr, steps_dg = lr.sample_stats.n_steps.values.mean(), dg.sample_stats.n_steps.values.mean()
    assert steps_lr * 2.5 < steps_dg, (steps_lr, steps_dg)
    x = lr.posterior.x.values.reshape(-1, D)
    sd = np.sqrt(np.diag(Sigma))
    assert np.abs((x.mean(0) - mu) / sd).max() < 0.15 and np.abs(np.sqrt(np.diag(np.cov(x.T))) / sd - 1).max() < 0.15
    assert lr.sample_stats.diverging.values.mean() < 0.01 and "gradient" not in lr.sample_stats
    assert lr.warmup_posterior.x.shape == (16, 400, D)


def test_low_rank_adaptation_on_a_fused_model_and_a_compiled_density(hip):
    import nutpie_amd

    # fused AR(1) Gaussian with rho = 0.995: one dominant direction — the low-rank metric shortens the trees
    m = nutpie_amd.ar1_gaussian(48, rho=0.995, scales=np.ones(48))
    kw = dict(chains=32, tune=400, draws=200, seed=5, progress_bar=False)
    lr = nutpie_amd.sample(m, adaptation="low_rank", mass_matrix_eigval_cutoff=1.5, **kw)
    dg = nutpie_amd.sample(m, adaptation="diag", **kw)
    assert lr.sample_stats.n_steps.values.mean() * 1.5 < dg.sample_stats.n_steps.values.mean()
    x = lr.posterior.x.values.reshape(-1, 48)
    assert np.abs(x.std(0) - 1).max() < 0.2 and abs(np.corrcoef(x[:, 0], x[:, 1])[0, 1] - 0.995) < 0.01
    # a runtime-compiled density takes the low-rank metric through its batched callback
    from nutpie_amd.radon import radon_density_model

    tr = nutpie_amd.sample(radon_density_model(), adaptation="low_rank", chains=16, tune=300, draws=100, seed=2, progress_bar=False)
    assert abs(tr.posterior.intercept.values.mean() - 1.3) < 0.2 and tr.sample_stats.diverging.values.mean() < 0.02


@pytest.mark.parametrize("order", [1, 2, 3, 8, 42, 64, 65, 96, 128])
def test_batched_eigh_against_lapack(hip, order):
    """nphip_batched_eigh (nutpie_amd/csrc/linalg.hip; the dense-linear-algebra kernel of the low-rank estimator): residual,
    orthogonality and eigenvalues against numpy's LAPACK on random, rank-deficient, diagonal, repeated-eigenvalue, badly scaled
    and zero matrices — and the lower triangle is the matrix."""
    import torch

    rng = np.random.default_rng(order)
    mats = []
    for _ in range(5):
        B = rng.normal(size=(order, order))
        mats.append(B + B.T)
    Z = rng.normal(size=(order, max(1, order // 3)))
    mats.append(Z @ Z.T)                                             # positive semi-definite, rank order / 3 (a Gram matrix)
    mats.append(np.diag(rng.normal(size=order)))                    # already diagonal
    q, _ = np.linalg.qr(rng.normal(size=(order, order)))
    mats.append((q * np.repeat([1.0, 2.0], [order - order // 2, order // 2])) @ q.T)    # two eigenvalues, high multiplicity
    mats.append(mats[0] * 1e-180)
    mats.append(mats[1] * 1e150)
    mats.append(np.zeros((order, order)))
    mats.append(np.eye(order) + 1e-9 * mats[2])                      # a tight cluster
    r = max(1, order // 2)                                           # gamma I + a covariance of half the rank with a wide spectrum: a cluster
    mats.append(1e-5 * np.eye(order) + (q[:, :r] * np.exp(rng.uniform(np.log(1e-3), np.log(20.0), size=r))) @ q[:, :r].T)   # next to eigenvalues 10^6 times larger
    A = np.stack([(m + m.T) / 2 for m in mats])
    dirty = A.copy()
    iu = np.triu_indices(order, 1)
    dirty[:, iu[0], iu[1]] = 7.0                                     # the upper triangle is never read
    w, V = hip.batched_eigh(torch.as_tensor(dirty, device="cuda"))
    w, V = w.cpu().numpy(), V.cpu().numpy()
    for b in range(len(A)):
        scale = max(np.abs(A[b]).max(), 1e-300)
        assert np.all(np.diff(w[b]) >= 0)
        np.testing.assert_allclose(w[b], np.linalg.eigvalsh(A[b]), rtol=0, atol=2e-13 * scale * max(order, 4))
        assert np.abs(V[b].T @ V[b] - np.eye(order)).max() < 1e-12 * max(order, 4)
        assert np.abs(A[b] @ V[b] - V[b] * w[b][None, :]).max() < 2e-13 * scale * max(order, 4)


def test_estimator_on_the_gpu_against_its_cpu_oracle(hip):
    """estimate() on the device — with the engine's batched eigendecomposition — against the numpy restatement
    (oracle/low_rank_estimator.py): the same dense metric for every chain, and independent of what else is in the batch."""
    import torch

    from nutpie_amd import low_rank as lr
    from oracle import low_rank_estimator as ref

    rng = np.random.default_rng(5)
    n, m, D = 6, 30, 30      # (2m = 60: the engine's own eigendecomposition; larger windows go to rocSOLVER, low_rank.NATIVE_EIGH_MAX)
    B = rng.normal(size=(n, D, 3))
    scales = np.exp(rng.normal(size=(n, D)))
    Sigma = np.stack([np.diag(scales[c] ** 2) + 25.0 * (scales[c][:, None] * B[c]) @ (scales[c][:, None] * B[c]).T for c in range(n)])
    x = np.stack([rng.multivariate_normal(np.zeros(D), Sigma[c], size=m) for c in range(n)]) + 2.0
    g = -np.stack([np.linalg.solve(Sigma[c], (x[c] - 2.0).T).T for c in range(n)])
    xt, gt = torch.as_tensor(x, device="cuda"), torch.as_tensor(g, device="cuda")
    T = lr.estimate(xt, gt, 1e-5, 2.0)
    sig2, V, lam = (t.cpu().numpy() for t in lr.metric_of(T))
    for c in range(n):
        want = ref.dense_metric(*ref.estimate_chain(x[c], g[c], 1e-5, 2.0, lr.K_MAX))
        got = ref.dense_metric(sig2[c], V[c].T, lam[c])
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want).max()
    T1 = lr.estimate(xt[2:3], gt[2:3], 1e-5, 2.0)
    assert torch.equal(T1.stds, T.stds[2:3]) and torch.equal(T1.V, T.V[2:3]) and torch.equal(T1.d, T.d[2:3])   # no dependence on the batch


def _window_problem(rng, n, m, D, n_dir, factor=25.0):
    B = rng.normal(size=(n, D, n_dir))
    scales = np.exp(rng.normal(size=(n, D)))
    Sigma = np.stack([np.diag(scales[c] ** 2) + factor * (scales[c][:, None] * B[c]) @ (scales[c][:, None] * B[c]).T for c in range(n)])
    x = np.stack([rng.multivariate_normal(np.zeros(D), Sigma[c], size=m) for c in range(n)]) + 2.0
    g = -np.stack([np.linalg.solve(Sigma[c], (x[c] - 2.0).T).T for c in range(n)])
    return x, g


@pytest.mark.parametrize("n,m,D,basis,n_dir,cutoff", [(6, 30, 30, None, 3, 2.0), (5, 80, 173, 32, 4, 2.0), (3, 256, 40, 32, 2, 3.0), (4, 20, 64, 32, 5, 2.0),
                                                      (3, 12, 5, 32, 1, 2.0), (2, 64, 300, 32, 6, 100.0)])
def test_native_estimator_kernel_against_the_torch_formulation_and_the_cpu_oracle(hip, n, m, D, basis, n_dir, cutoff):
    """nphip_low_rank_estimate (one kernel, Jacobi eigensolvers, reads the window out of a trace-shaped array through a chain list)
    against low_rank.estimate (torch) — and, where every draw is a basis draw, against the numpy restatement — through the dense
    metric, which does not depend on order, sign or basis of the columns."""
    import torch

    from nutpie_amd import low_rank as lr
    from oracle import low_rank_estimator as ref

    rng = np.random.default_rng(100 + D)
    x, g = _window_problem(rng, n, m, D, n_dir)
    # a trace-shaped array: more chains and draws than the window, the window somewhere inside
    n_all, T, lo = n + 3, m + 17, 9
    chains = rng.permutation(n_all)[:n]
    tr_x, tr_g = rng.normal(size=(n_all, T, D)), rng.normal(size=(n_all, T, D))
    tr_x[chains, lo:lo + m], tr_g[chains, lo:lo + m] = x, g
    dx, dg = torch.as_tensor(tr_x, device="cuda"), torch.as_tensor(tr_g, device="cuda")
    pick = lr.basis_pick(m, basis)
    if len(pick) > D:
        assert not hip.low_rank_estimate_supported(D, m, len(pick), lr.K_MAX)   # (the full-space branch of estimate(): not the kernel's)
        return
    assert hip.low_rank_estimate_supported(D, m, len(pick), lr.K_MAX)
    sig2, V, lam, k_used = hip.low_rank_estimate(dx, dg, torch.as_tensor(chains, device="cuda"), lo, lo + m, pick, 1e-5, cutoff, lr.K_MAX)
    sig2, V, lam, k_used = sig2.cpu().numpy(), V.cpu().numpy(), lam.cpu().numpy(), k_used.cpu().numpy()
    T_t = lr.estimate(torch.as_tensor(x, device="cuda"), torch.as_tensor(g, device="cuda"), 1e-5, cutoff, basis_draws=basis)
    s2_t, V_t, lam_t = (t.cpu().numpy() for t in lr.metric_of(T_t))
    for c in range(n):
        got = ref.dense_metric(sig2[c], V[c].T, lam[c])
        want = ref.dense_metric(s2_t[c], V_t[c].T, lam_t[c])
        # Where the basis has full rank (2b <= D) the formulation is ill-conditioned in ANY arithmetic: the geometric mean goes through
        # Cg^1/2 Cx Cg^1/2, whose condition number is the product of two that are 1 / gamma each — two exact-arithmetic-equal
        # formulations evaluated with LAPACK differ by 5e-4 on these problems (the kernel works in the eigenbasis of Cg, estimate() does
        # not; tests/test_low_rank_cpu.py allows estimate() 1e-4 against the numpy restatement for the same reason).  With a
        # rank-deficient basis the dropped directions take the small eigenvalues with them and everything agrees to rounding.
        tol = 1e-6 if 2 * len(pick) > D else 3e-3
        assert np.abs(got - want).max() <= tol * np.abs(want).max(), (c, np.abs(got - want).max() / np.abs(want).max())
        assert k_used[c] == int((lam_t[c] != 1.0).sum()) and (lam[c][k_used[c]:] == 1.0).all() and (V[c][k_used[c]:] == 0.0).all()
        vu = V[c][:k_used[c]]
        assert np.abs(vu @ vu.T - np.eye(k_used[c])).max() < 1e-9          # orthonormal columns
        if basis is None or m <= basis:
            want2 = ref.dense_metric(*ref.estimate_chain(x[c], g[c], 1e-5, cutoff, lr.K_MAX))
            assert np.abs(got - want2).max() <= max(10 * tol, 1e-6) * np.abs(want2).max()
    assert k_used.max() >= 1 or cutoff > 50   # (the problems have strong directions: the test is not about empty metrics)
    # a window with a non-finite entry: exactly the identity for that chain, the others untouched
    tr_x[chains[1], lo + 3, D // 2] = np.inf
    s2b, Vb, lamb, kb = hip.low_rank_estimate(torch.as_tensor(tr_x, device="cuda"), dg, torch.as_tensor(chains, device="cuda"), lo, lo + m, pick, 1e-5, cutoff, lr.K_MAX)
    assert (s2b[1] == 1.0).all() and (Vb[1] == 0.0).all() and (lamb[1] == 1.0).all() and int(kb[1]) == 0
    assert np.array_equal(s2b[0].cpu().numpy(), sig2[0]) and np.array_equal(Vb[0].cpu().numpy(), V[0])
    # estimate_window takes the kernel on the GPU and gives what estimate() + metric_of() give
    s2w, Vw, lamw = lr.estimate_window(dx, dg, chains, lo, lo + m, 1e-5, cutoff, basis_draws=basis)
    assert np.array_equal(s2w.cpu().numpy(), sig2) and np.array_equal(Vw.cpu().numpy(), V)


@pytest.mark.parametrize("dim,k,launch", [(48, 3, {}), (48, 0, dict(no_register_kernel=True)), (1300, 4, {}), (300, 11, {})])
def test_staged_metric_is_the_metric_handed_in_at_a_pause(hip, dim, k, launch):
    """nphip_sampler_stage_metric: a chain that RUNS takes the parked metric at the end of the draw it is working on — the same job, bit
    for bit, as one whose chains stop at that draw (pause draws) and are handed the metric there (nphip_sampler_set_metric).  One and two
    waves per chain on the register-resident leaf, the memory-resident kernels, a metric without columns."""
    chains, tune, draws, seed = 12, 60, 12, 11
    rng = np.random.default_rng(3)
    pauses = [20, 41]
    sig2, V, lam = random_metrics(rng, len(pauses), chains, dim, k)
    ar = ar1_gaussian(dim, rho=0.9, scales=np.exp(rng.normal(size=dim)))
    model = hip.TridiagGaussianModel(ar.diag, ar.offdiag)
    sig2 *= 0.05
    want, W = run_engine_with_metrics(hip, model, pauses, sig2, V, lam, chains=chains, tune=tune, draws=draws, seed=seed, launch=launch)
    s = hip.PyNutsSettings.Diag(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, low_rank_metric=True)
    smp = hip.PySampler(s, model, manual=True, evals_per_launch=1, **launch)     # (one evaluation per launch: a chain's draw counter moves by at most one per look)
    nxt = np.zeros(chains, dtype=int)
    for _ in range(200000):
        done, _, _ = smp.step(1)
        if done:
            break
        d, state = smp.chain_draws()
        for u, p in enumerate(pauses):
            grp = np.nonzero((nxt == u) & (d == p - 1) & (state == 0))[0]
            if len(grp):
                assert smp.stage_metric(grp, sig2[u][grp], V[u][grp] if k else None, lam[u][grp] if k else None) == len(grp)
                nxt[grp] = u + 1
    assert done and (nxt == len(pauses)).all()
    assert smp.waves_per_chain == W
    assert_trace_equal(smp.take_results(), want)
    # a chain past its warm-up ignores a hand-in, and says so
    smp = hip.PySampler(s, model, manual=True)
    while True:
        done, _, _ = smp.step(1)
        d, state = smp.chain_draws()
        if done or (d >= tune).all():
            break
    assert smp.stage_metric(np.arange(chains), sig2[0], V[0] if k else None, lam[0] if k else None) == 0


@pytest.mark.parametrize("kind", ["fused", "host_callback"])
def test_released_chains_go_on_as_if_they_had_not_stopped(hip, fixture_lib, kind):
    """nphip_sampler_release: a job whose chains stop at pause draws and are released there equals, bit for bit, the job without pause
    draws — the metric, the step size and the chain's own mass-matrix adaptation are what they were."""
    chains, tune, draws, seed = 10, 70, 15, 4
    if kind == "fused":
        ar = ar1_gaussian(200, rho=0.8, scales=np.exp(np.random.default_rng(1).normal(size=200)))
        make = lambda: hip.TridiagGaussianModel(ar.diag, ar.offdiag)   # noqa: E731
    else:
        make = lambda: hip.HostCallbackModel(10, fn_addr(fixture_lib.eight_schools_logp))   # noqa: E731
    s = hip.PyNutsSettings.Diag(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, low_rank_metric=True)
    plain = hip.PySampler(s, make(), manual=True)
    while not plain.step(8)[0]:
        pass
    want = plain.take_results()
    s2 = s.clone()
    s2.set_pause_draws([13, 40, 41])
    smp = hip.PySampler(s2, make(), manual=True)
    released = 0
    for _ in range(100000):
        done, _, _ = smp.step(3)
        if done:
            break
        w = np.nonzero(smp.waiting())[0]
        if len(w):
            smp.release(w)
            released += len(w)
    assert done and released == 3 * chains
    assert_trace_equal(smp.take_results(), want)
    with pytest.raises(RuntimeError, match="not stopped"):
        hip.PySampler(s2, make(), manual=True).release(np.arange(chains))   # (nothing has stopped yet)
