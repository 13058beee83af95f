"""Host-callback path with wide rows: independent normals of any dimension (tests/fixtures: scaled_normal_logp).
python scratch/cwide.py DIM CHAINS PERSIST [THREADS]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip
dim, chains, persist = (int(a) for a in sys.argv[1:4])
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
fix = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fixtures", "libeight_schools.so"))
addr = ctypes.cast(fix.scaled_normal_logp, ctypes.c_void_p).value
s = hip.PyNutsSettings.Diag(3)
s.update(num_tune=200, num_draws=200, num_chains=chains)
m = hip.HostCallbackModel(dim, addr, n_threads=threads)
smp = hip.PySampler(s, m, host_persist=persist)
smp.wait()
n = int(smp._copy("n_steps", np.int64).sum())
print(f"dim={dim} chains={chains} persist={persist} threads={threads}: {smp.host_mode}, job {smp.seconds:.3f} s, {n / smp.seconds / 1e6:.2f} M leapfrogs/s, "
      f"{smp.seconds / max(1, smp.launches) * 1e6:.1f} us per evaluation of a group")
smp.close()
