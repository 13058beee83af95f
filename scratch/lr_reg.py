"""The low-rank metric on the register-resident leaf against the memory-resident kernels: a fused AR(1) Gaussian with metrics handed
in at two pause draws (random orthonormal columns: what the kernel costs, not what the metric buys).
usage: python scratch/lr_reg.py dim chains k"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from nutpie_amd import _lib as hip
from nutpie_amd.gaussian import ar1_gaussian
dim, chains, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1)
model = ar1_gaussian(dim)
q, _ = np.linalg.qr(rng.normal(size=(dim, max(k, 1))))
V = np.broadcast_to(q.T[None, :k], (chains, k, dim)).copy()
lam = np.broadcast_to(np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=k))[None], (chains, k)).copy()
sig2 = np.broadcast_to(1.0 / model.diag[None], (chains, dim)).copy()
for noreg in (False, True):
    s = hip.PyNutsSettings.Diag(3)
    s.update(num_tune=200, num_draws=200, num_chains=chains, low_rank_metric=True)
    s.set_pause_draws([20])
    smp = hip.PySampler(s, hip.TridiagGaussianModel(model.diag, model.offdiag), manual=True, no_register_kernel=noreg, store_draws=False)
    t0 = time.perf_counter(); set_ = False
    while True:
        done, _, _ = smp.step(4)
        if done: break
        if not set_ and smp.waiting().all():
            smp.set_metric(np.arange(chains), sig2, V if k else None, lam if k else None); set_ = True
    dt = time.perf_counter() - t0
    n = smp._copy("n_steps", np.int64)
    print(f"D={dim} chains={chains} k={k} waves={smp.waves_per_chain} {'memory-resident kernels' if noreg else 'register-resident leaf'}: {n.sum() / dt / 1e6:.2f} M leapfrogs/s, job {dt:.3f} s, mean depth {smp._copy('depth', np.int64).mean():.2f}", flush=True)
    smp.close()
