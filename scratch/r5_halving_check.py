"""bit-identity of a developer library (NUTPIE_HIP_LIB) against the oracle at D = 1000, one wave per chain (the headline kernel's geometry)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from nutpie_amd import _lib
from nutpie_amd.gaussian import ar1_gaussian
m = ar1_gaussian(1000)
s = _lib.PyNutsSettings.Diag(7); s.update(num_tune=40, num_draws=15, num_chains=8)
smp = _lib.PySampler(s, _lib.TridiagGaussianModel(m.diag, m.offdiag), device=0); smp.wait()
got = smp.take_results()
want = oracle.sample_tridiag(oracle.default_settings(seed=7, num_chains=8, num_tune=40, num_draws=15, waves_per_chain=1), m.diag, m.offdiag)
ok = np.array_equal(got.draws, want.draws) and np.array_equal(np.asarray(got.stats["n_steps"]).astype(np.int64), want.stats["n_steps"].astype(np.int64)) and np.array_equal(got.stats["energy"], want.stats["energy"])
print(os.environ.get("NUTPIE_HIP_LIB", "in-tree"), "bit-identical to the oracle:", ok, "leapfrogs", int(want.stats["n_steps"].sum()))
