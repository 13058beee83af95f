"""The algorithm of nphip_batched_eigh (nutpie_amd/csrc/linalg.hip) on the CPU: oracle/batched_eigh_port.py runs the same three stages in
the same in-place layout; here against LAPACK on the matrix families of the GPU test (tests/test_gpu_low_rank.py::
test_batched_eigh_against_lapack) — including the one the first version of the deflation test never finished on: gamma I plus a
low-rank covariance, a cluster sharing an unreduced block with eigenvalues a million times larger."""
import numpy as np
import pytest

from oracle import batched_eigh_port as port


def _families(order, rng):
    mats = []
    for _ in range(2):
        B = rng.normal(size=(order, order))
        mats.append(B + B.T)
    Z = rng.normal(size=(order, max(1, order // 3)))
    mats.append(Z @ Z.T)
    mats.append(np.diag(rng.normal(size=order)))
    q, _ = np.linalg.qr(rng.normal(size=(order, order)))
    mats.append((q * np.repeat([1.0, 2.0], [order - order // 2, order // 2])) @ q.T)
    mats.append(mats[0] * 1e-180)
    mats.append(np.zeros((order, order)))
    r = max(1, order // 2)
    mats.append(1e-5 * np.eye(order) + (q[:, :r] * np.exp(rng.uniform(np.log(1e-3), np.log(20.0), size=r))) @ q[:, :r].T)
    # block structure: coordinates that couple to nothing (zero columns of the projected window)
    C = mats[-1].copy()
    dead = rng.choice(order, size=max(1, order // 4), replace=False)
    C[dead, :] = 0.0
    C[:, dead] = 0.0
    C[dead, dead] = 1e-5
    mats.append(C)
    return [(m + m.T) / 2 for m in mats]


@pytest.mark.parametrize("order", [1, 2, 3, 8, 33, 64, 128])
def test_port_against_lapack(order):
    rng = np.random.default_rng(order)
    for A in _families(order, rng):
        dirty = A.copy()
        dirty[np.triu_indices(order, 1)] = 7.0                      # the upper triangle is never read
        w, V = port.eigh(dirty)
        scale = max(np.abs(A).max(), 1e-300)
        assert np.all(np.diff(w) >= 0)
        np.testing.assert_allclose(w, np.linalg.eigvalsh(A), rtol=0, atol=2e-13 * scale * max(order, 4))
        assert np.abs(V.T @ V - np.eye(order)).max() < 1e-12 * max(order, 4)
        assert np.abs(A @ V - V * w[None, :]).max() < 2e-13 * scale * max(order, 4)
