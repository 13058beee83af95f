"""BASELINE.json config 4 (eight schools, 256 chains, host C callback): whole job with 1 (no pipelining), 2 and 4 chain groups."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip
fix = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fixtures", "libeight_schools.so"))
addr = ctypes.cast(fix.eight_schools_logp, ctypes.c_void_p).value
ref = None
for chains in (256, 1024):
    for groups in (1, 2, 4, 2, 1):
        s = hip.PyNutsSettings.Diag(21)
        s.update(num_tune=400, num_draws=1000, num_chains=chains)
        m = hip.HostCallbackModel(10, addr, n_threads=1 if chains == 256 else 0)
        m.set_init("normal")
        t0 = time.perf_counter()
        smp = hip.PySampler(s, m, host_groups=groups)
        smp.wait()
        secs = smp.seconds
        n = int(smp._copy("n_steps", np.int64).sum())
        d = smp._copy("draws", np.float64, vec=True)
        if ref is None or ref[0] != chains:
            ref = (chains, d)
        print(f"chains={chains} host_groups={groups}: job {secs:.3f} s, {n / secs / 1e6:.2f} M leapfrogs/s, {secs / smp.launches * 1e6:.1f} us per launch, "
              f"identical to the first run: {np.array_equal(d, ref[1])}")
        smp.close()
