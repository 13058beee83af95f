"""D = 500 demo under variants of the estimator's constants (env: BASIS, WMAX, KMAX)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from nutpie_amd import low_rank
from lowrank_demo_compiled import target
from r5_lowrank import job, show
cm, Sigma = target(500, 6, 400.0)
for basis, wmax, kmax, cutoff, tune in ((64, 256, 16, 2.0, 500), (64, 256, 16, 100.0, 500), (64, 256, 16, 10.0, 500)):
    low_rank.BASIS_DRAWS, low_rank.WINDOW_MAX, low_rank.K_MAX = basis, wmax, kmax
    for rep in range(2 if cutoff == 2.0 else 1):
        r, res = job(cm, "low_rank", 256, tune, 500, seed=3, mass_matrix_eigval_cutoff=cutoff)
    x = np.asarray(res.draws)[:, tune:].reshape(-1, 500)
    err = np.abs(np.sqrt(np.diag(np.cov(x.T))) / np.sqrt(np.diag(Sigma)) - 1).max()
    show(f"basis {basis} wmax {wmax} kmax {kmax} cutoff {cutoff} tune {tune}: sd err {err:.3f}", r)
