import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scratch"))
import numpy as np, torch
from nutpie_amd import _lib as hip
import eigh_port as port
A = np.load(os.path.join(ROOT, "scratch", "eigh_fail.npy"))
At = torch.as_tensor(A[None], device="cuda")
d, T = hip.batched_eigh(At, _stage=1)
d = d[0].cpu().numpy(); e = T[0, 0].cpu().numpy()
dp, ep, amax = port.tridiag(A)
print("tridiagonal: diag diff", np.abs(d - dp * amax).max(), "sub-diagonal diff", np.abs(np.abs(e) - np.abs(ep * amax)).max(), "finite", np.isfinite(d).all(), np.isfinite(e).all())
Tm = np.diag(d) + np.diag(e[:-1], 1) + np.diag(e[:-1], -1)
print("eigenvalues of the device's T against A:", np.abs(np.linalg.eigvalsh(Tm) - np.linalg.eigvalsh(A)).max())
print("port's QL on the device's (d, e):", (port.ql(d / amax, e / amax) or [None, None])[1])
d2, Q = hip.batched_eigh(At, _stage=2)
Q = Q[0].cpu().numpy()
L = np.tril(A) + np.tril(A, -1).T
print("Q orthogonal:", np.abs(Q.T @ Q - np.eye(len(A))).max(), " Q'AQ - T:", np.abs(Q.T @ L @ Q - Tm).max())
try:
    w, V = hip.batched_eigh(At)
    print("full: ok", np.abs(w[0].cpu().numpy() - np.linalg.eigvalsh(A)).max())
except RuntimeError as ex:
    print("full:", ex)
