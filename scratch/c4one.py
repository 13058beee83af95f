"""One config-4 job (eight schools, host C callback): python scratch/c4one.py CHAINS GROUPS PERSIST THREADS"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip
chains, groups, persist, threads = (int(a) for a in sys.argv[1:5])
fix = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fixtures", "libeight_schools.so"))
addr = ctypes.cast(fix.eight_schools_logp, ctypes.c_void_p).value
s = hip.PyNutsSettings.Diag(21)
s.update(num_tune=400, num_draws=1000, num_chains=chains)
m = hip.HostCallbackModel(10, addr, n_threads=threads)
m.set_init("normal")
smp = hip.PySampler(s, m, host_groups=groups, host_persist=persist)
smp.wait()
n = int(smp._copy("n_steps", np.int64).sum())
print(f"chains={chains} groups={groups} persist={persist} threads={threads}: job {smp.seconds:.3f} s, {n / smp.seconds / 1e6:.2f} M leapfrogs/s, {smp.launches} steps, "
      f"{smp.seconds / smp.launches * 1e6:.1f} us per step")
smp.close()
