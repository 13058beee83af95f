"""Round 5 (final build): waves per chain for config 3's generated density."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip
from nutpie_amd.radon import radon_symbolic_model

for chains in (512, 256):
    for w in (1, 2, 4):
        m = radon_symbolic_model().compile(waves_per_chain=w)
        for rep in range(2):
            s = hip.PyNutsSettings.Diag(20260926)
            s.update(num_tune=400, num_draws=1000, num_chains=chains)
            smp = m._make_sampler(s, None, 1, None, None, None, None)
            smp.wait()
            n = smp._copy("n_steps", np.int64)
            print(f"[chains={chains} waves={w}] rep {rep}: {n.sum() / smp.seconds / 1e6:.2f} M leapfrogs/s, job {smp.seconds:.3f} s")
            smp.close()
