"""One engine-vs-oracle comparison: python scratch/parity_one.py DIM WAVES [CHAINS TUNE DRAWS]  (honours NUTPIE_HIP_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from nutpie_amd import _lib as hip
from nutpie_amd.gaussian import ar1_gaussian
from tests.conftest import assert_trace_equal
from tests.test_gpu_parity import run_engine, oracle_settings
dim, waves = int(sys.argv[1]), int(sys.argv[2])
chains, tune, draws = (int(a) for a in (sys.argv[3:6] if len(sys.argv) > 5 else (4, 40, 10)))
oracle.build()
m = ar1_gaussian(dim)
got, W = run_engine(hip, hip.TridiagGaussianModel(m.diag, m.offdiag), chains=chains, tune=tune, draws=draws, seed=dim + 1, waves=waves)
want = oracle.sample_tridiag(oracle_settings(oracle, chains=chains, tune=tune, draws=draws, seed=dim + 1, W=W), m.diag, m.offdiag)
assert_trace_equal(got, want)
print(f"parity ok: dim={dim} W={W} leapfrogs={int(got.stats['n_steps'].sum())}")
