"""Config 3 behind the batched device callback only (launch per evaluation), for the profiler."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip
from nutpie_amd.radon import radon_density_model
m = radon_density_model(resident=False)
s = hip.PyNutsSettings.Diag(20260926)
s.update(num_tune=400, num_draws=1000, num_chains=512)
smp = m._make_sampler(s, None, 1, None, None, None, None)
smp.wait()
n = smp._copy("n_steps", np.int64)
print(f"callback path: {n.sum() / smp.seconds / 1e6:.2f} M leapfrogs/s, job {smp.seconds:.3f} s, launches {smp.launches}")
