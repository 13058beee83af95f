"""CPU restatement (numpy, one chain at a time) of the window estimator of ``adaptation="low_rank"`` — TEST INFRASTRUCTURE, like
everything under oracle/: only tests/ import it; it checks ``nutpie_amd/low_rank.py::estimate`` (batched torch on the GPU).

What it restates: the published description of nutpie's low-rank mass-matrix adaptation (Seyboldt et al., "Preconditioning
Hamiltonian Monte Carlo by minimizing Fisher divergence"; option semantics: reference ``python/nutpie/sample.py:921-933``,
``docs/sampling-options.qmd:124-144``, ``src/wrapper.rs:307-334``; the diagonal scaling is the formula of
``python/nutpie/normalizing_flow.py:1906-1915``).  nuts-rs' own code for it is not in the tree — PARITY UNPINNED, like the sampler.
It is written differently from the engine's version on purpose (SVD instead of a Gram eigen-decomposition for the subspace,
scipy's matrix square root for the geometric mean, plain loops), and the two are compared through the DENSE metric
``M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2``, which does not depend on the order, sign or basis choices of the columns.
"""
import numpy as np
from scipy import linalg as sla


def estimate_chain(x, g, gamma, cutoff, k_max=16):
    """x, g: [m, D] draws and gradients of one chain's window.  Returns (sigma2 [D], V [D, k], lam [k])."""
    m, D = x.shape
    sx, sg = x.std(0, ddof=1), g.std(0, ddof=1)
    s = np.sqrt(sx / sg)
    s = np.where(np.isfinite(s) & (s > 0), s, 1.0).clip(1e-10, 1e10)
    X = (x - x.mean(0)) / s
    G = (g - g.mean(0)) * s
    Z = np.concatenate([X, G], 0)                                # [2m, D]
    # orthonormal basis of the span of the window (rows of Z): right singular vectors with non-negligible singular values
    _, sv, Vt = np.linalg.svd(Z, full_matrices=False)
    keep = sv**2 > 1e-10 * max(sv[0] ** 2, 1e-300)
    Q = Vt[keep].T                                               # [D, r]
    Px, Pg = X @ Q, G @ Q
    r = Q.shape[1]
    Cx = Px.T @ Px / m + gamma * np.eye(r)
    Cg = Pg.T @ Pg / m + gamma * np.eye(r)
    # geometric mean S = Cx # Cg^-1: the symmetric positive solution of S Cg S = Cx
    half = np.real(sla.sqrtm(Cg))
    ihalf = np.linalg.inv(half)
    S = ihalf @ np.real(sla.sqrtm(half @ Cx @ half)) @ ihalf
    es, W = np.linalg.eigh(0.5 * (S + S.T))
    # (the exact spectrum lies in [sqrt(gamma / |Cg|), sqrt(|Cx| / gamma)]; what rounding puts outside is clamped — as in the engine's version)
    es = np.clip(es, np.sqrt(gamma / max(np.linalg.eigvalsh(Cg)[-1], gamma)), np.sqrt(max(np.trace(Cx), gamma) / gamma))
    # re-centre the spectrum on its bulk (the median eigenvalue goes into the diagonal scaling)
    # (the LOWER median for an even count, as torch.nanmedian takes it in the engine's version)
    centre = np.exp(np.sort(np.log(es))[(len(es) - 1) // 2])
    es = es / centre
    s = s * np.sqrt(centre)
    score = np.abs(np.log(es))
    cand = [i for i in np.argsort(-score) if score[i] > np.log(cutoff)][:k_max]
    return s * s, Q @ W[:, cand], es[cand]


def dense_metric(sigma2, V, lam):
    """M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 as a [D, D] matrix."""
    sd = np.sqrt(sigma2)
    return sd[:, None] * (np.eye(len(sigma2)) + (V * (lam - 1.0)) @ V.T) * sd[None, :]
