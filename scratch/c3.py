"""BASELINE.json config 3 (radon, D = 173, 512 chains) as a runtime-compiled device density: the resident kernel against the
launch-per-evaluation device callback of the same library.  usage: python scratch/c3.py [chains]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nutpie_amd
from nutpie_amd import _lib as hip
from nutpie_amd.radon import radon_density_model

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for resident, E in ((True, 0), (False, 0), (True, 128), (True, 512), (True, 2048)):
    m = radon_density_model(resident=resident)
    s = hip.PyNutsSettings.Diag(20260926)
    s.update(num_tune=400, num_draws=1000, num_chains=chains)
    t0 = time.perf_counter()
    smp = m._make_sampler(s, None, 1, None, None, None, None, evals_per_launch=E)
    smp.wait()
    secs, wall = smp.seconds, time.perf_counter() - t0
    n = smp._copy("n_steps", np.int64)
    div = smp._copy("diverging", np.bool_)
    print(f"radon D=173 chains={chains} resident={resident} evals_per_launch={E}: {n.sum() / secs / 1e6:.2f} M leapfrogs/s, job {secs:.3f} s (wall incl. set-up {wall:.3f} s), "
          f"{int(n.sum())} leapfrogs, mean leapfrogs per draw (sampling) {n[:, 400:].mean():.1f}, divergences {int(div[:, 400:].sum())}, launches {smp.launches}")
    smp.close()
