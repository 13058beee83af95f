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
    assert lr.pause_draws(400) == [32, 80, 160, 260] and lr.pause_draws(20) == [13] and lr.pause_draws(10) == []


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
