"""numpy port of linalg.hip's algorithm (for debugging): tridiagonalisation + the QL loop exactly as the kernel runs it."""
import sys, numpy as np
A = np.load(sys.argv[1]) if len(sys.argv) > 1 else None

def tridiag(A):
    n = A.shape[0]; V = np.tril(A) + np.tril(A, -1).T; amax = np.abs(V).max(); V = V / amax
    d = np.zeros(n); e = np.zeros(n)
    for k in range(n - 2):
        x = V[k + 1:, k].copy(); xn2 = (x[1:] ** 2).sum(); x0 = x[0]
        if xn2 == 0.0:
            e[k] = x0; d[k] = V[k, k]; continue
        beta = -np.copysign(np.sqrt(x0 * x0 + xn2), x0); tk = (beta - x0) / beta
        v = x / (x0 - beta); v[0] = 1.0
        e[k] = beta; d[k] = V[k, k]
        A22 = V[k + 1:, k + 1:]
        p = tk * A22 @ v; K = 0.5 * tk * (p @ v); w = p - K * v
        A22 -= np.outer(v, w) + np.outer(w, v)
    d[n - 2] = V[n - 2, n - 2]; e[n - 2] = V[n - 1, n - 2]; d[n - 1] = V[n - 1, n - 1]; e[n - 1] = 0.0
    return d, e, amax

def ql(d, e, max_sweeps=60, verbose=False):
    n = len(d); d = d.copy(); e = e.copy(); eps = 2.220446049250313e-16; tst1 = 0.0; total = 0
    for l in range(n):
        it = 0
        while True:
            tst1 = max(tst1, abs(d[l]) + abs(e[l]))
            m = l
            while m < n - 1:
                dd = abs(d[m]) + abs(d[m + 1])
                if abs(e[m]) <= eps * max(dd, tst1): break
                m += 1
            if m == l: break
            if it >= max_sweeps:
                print("no convergence at l =", l, "m =", m, "e[l..l+3] =", e[l:l + 4], "d[l..l+3] =", d[l:l + 4], "tst1", tst1); return None
            g = (d[l + 1] - d[l]) / (2.0 * e[l]); r = np.sqrt(g * g + 1.0)
            g = d[m] - d[l] + e[l] / (g + np.copysign(r, g))
            s = c = 1.0; p = 0.0; broke = False
            i = m - 1
            while i >= l:
                f = s * e[i]; b = c * e[i]
                r = np.sqrt(f * f + g * g); e[i + 1] = r
                if r == 0.0:
                    d[i + 1] -= p; e[m] = 0.0; broke = True; break
                s = f / r; c = g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + 2.0 * c * b
                p = s * r; d[i + 1] = g + p; g = c * r - b
                i -= 1
            if not broke:
                d[l] -= p; e[l] = g; e[m] = 0.0
            it += 1; total += 1
        if verbose and l % 16 == 0: print("l", l, "sweeps so far", total)
    return np.sort(d), total

if A is not None:
    d, e, amax = tridiag(A)
    print("tridiagonal: |e| min/max", np.abs(e[:-1]).min(), np.abs(e[:-1]).max())
    out = ql(d, e, verbose=True)
    if out is not None:
        w, total = out
        print("sweeps", total, "max eigenvalue error", np.abs(w * amax - np.linalg.eigvalsh(A)).max())
