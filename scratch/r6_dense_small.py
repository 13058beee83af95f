"""Round 6: the dense-precision Gaussian's two forms by job size — resident (the GEMM inside the register-resident leaf, clusters of 16
workgroups) against a launch per evaluation (host_persist = 1): where is the crossover?  D = 1000 and D = 256."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip
from nutpie_amd.gaussian import dense_precision
for D in (1000, 256):
    P = dense_precision(D)
    for chains in (4, 16, 64, 128, 256, 512, 1024):
        out = []
        for hp in (0, 1):
            s = hip.PyNutsSettings.Diag(1)
            s.update(num_tune=30, num_draws=10, num_chains=chains)
            best = None
            for rep in range(2):
                smp = hip.PySampler(s, hip.DenseGaussianModel(P), store_draws=False, host_persist=hp)
                t0 = time.perf_counter(); smp.wait(); wall = time.perf_counter() - t0
                n = smp._copy("n_steps", np.int64).sum()
                r = (n / smp.seconds, wall, smp.host_mode)
                best = r if best is None or r[0] > best[0] else best
                smp.close()
            out.append(best)
        print(f"D={D:5d} chains={chains:5d}: resident {out[0][0] / 1e6:7.3f} M leapfrogs/s (wall {out[0][1]:.2f} s, mode {out[0][2]})   launch per evaluation {out[1][0] / 1e6:7.3f} (wall {out[1][1]:.2f} s, mode {out[1][2]})", flush=True)
