"""Config 3 with the generated density at several unroll depths of the generated loops.  usage: python scratch/c3unroll.py [chains]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import symbolic_models as zoo
from nutpie_amd import _lib as hip, symbolic
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for U in (4, 2, 8, 6):
    symbolic._UNROLL = U
    for rep in range(2):
        m = zoo.radon().compile()
        s = hip.PyNutsSettings.Diag(20260926)
        s.update(num_tune=400, num_draws=1000, num_chains=chains)
        smp = m._make_sampler(s, None, 1, None, None, None, None)
        smp.wait()
        n = smp._copy("n_steps", np.int64)
        print(f"radon chains={chains} generated, unroll {U}: {n.sum() / smp.seconds / 1e6:.2f} M leapfrogs/s, job {smp.seconds:.3f} s", flush=True)
        smp.close()
