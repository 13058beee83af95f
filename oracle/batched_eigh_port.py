"""CPU model of nphip_batched_eigh (nutpie_amd/csrc/linalg.hip): the same three stages in the same in-place layout, one matrix at a time
in numpy — Householder tridiagonalisation with the reflectors parked below the sub-diagonal (LAPACK dsytd2), Q formed in place
(dorgtr / dorg2r), implicit QL with the deflation test against eps |T| (EISPACK imtql2 / tqli).

Test infrastructure: only tests/ may import it.  It is NOT a restatement of the reference (nuts-rs uses faer's eigensolver); it
exists so that the logic of the device routine — index arithmetic of the in-place Q, the sweep's recurrence, deflation inside
clusters — is exercised where there is no GPU.  The device routine differs in the order of its sums and in 1 / r (v_rsq_f64 + Newton)."""
import numpy as np

EPS = 2.220446049250313e-16


def eigh(A, max_sweeps=60):
    """-> (w ascending, V with A V = V diag(w)); the LOWER triangle of A is the matrix."""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    V = np.tril(A) + np.tril(A, -1).T
    amax = np.abs(V).max()
    if not amax > 0.0:
        return np.zeros(n), np.eye(n)
    V = V / amax
    d, e, tau = np.zeros(n), np.zeros(n), np.zeros(n)
    # 1. tridiagonalisation
    for k in range(n - 2):
        x = V[k + 1:, k].copy()
        xn2 = float((x[1:] ** 2).sum())
        x0 = x[0]
        d[k] = V[k, k]
        if xn2 == 0.0:
            tau[k], e[k] = 0.0, x0
            continue
        beta = -np.copysign(np.sqrt(x0 * x0 + xn2), x0)
        tk = (beta - x0) / beta
        v = x / (x0 - beta)
        v[0] = 1.0
        V[k + 2:, k] = v[1:]                       # parked
        tau[k], e[k] = tk, beta
        A22 = V[k + 1:, k + 1:]
        p = tk * (A22 @ v)
        w = p - 0.5 * tk * (p @ v) * v
        A22 -= np.outer(v, w) + np.outer(w, v)
    if n >= 2:
        d[n - 2], e[n - 2] = V[n - 2, n - 2], V[n - 1, n - 2]
    d[n - 1] = V[n - 1, n - 1]
    e[n - 1] = 0.0
    # 2. Q in place: reflectors one column to the right, row / column 0 the unit vector, then backwards
    for r in range(n):
        for c in range(r - 1, 0, -1):
            V[r, c] = V[r, c - 1]
        V[r, r:] = 0.0
        V[r, r] = 1.0
        if r >= 1:
            V[r, 0] = 0.0
    for j in range(n - 3, -1, -1):
        rj, tj = j + 1, tau[j]
        v = np.concatenate([[1.0], V[rj + 1:, rj]])
        for c in range(rj + 1, n):
            dot = tj * (v @ V[rj:, c])
            V[rj:, c] -= dot * v
        V[rj + 1:, rj] *= -tj
        V[rj, rj] = 1.0 - tj
    # 3. implicit QL
    anorm = float(np.max(np.abs(d) + np.abs(e)))
    for l in range(n):
        it = 0
        while True:
            m = l
            while m < n - 1:
                if abs(e[m]) <= EPS * max(abs(d[m]) + abs(d[m + 1]), anorm):
                    break
                m += 1
            if m == l:
                break
            if it >= max_sweeps:
                raise RuntimeError("the QL iteration did not converge")
            it += 1
            g = (d[l + 1] - d[l]) / (2.0 * e[l])
            r = np.sqrt(g * g + 1.0)
            g = d[m] - d[l] + e[l] / (g + np.copysign(r, g))
            s = c = 1.0
            p = 0.0
            broke = False
            i = m - 1
            while i >= l:
                f, b = s * e[i], c * e[i]
                r = np.sqrt(f * f + g * g)
                e[i + 1] = r
                if r == 0.0:
                    d[i + 1] -= p
                    e[m] = 0.0
                    broke = True
                    break
                s, c = f / r, g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + 2.0 * c * b
                p = s * r
                d[i + 1] = g + p
                g = c * r - b
                hi = V[:, i + 1].copy()
                V[:, i + 1] = s * V[:, i] + c * hi
                V[:, i] = c * V[:, i] - s * hi
                i -= 1
            if not broke:
                d[l] -= p
                e[l] = g
                e[m] = 0.0
    order = np.argsort(d, kind="stable")
    return d[order] * amax, V[:, order]
