"""BASELINE.json config 4 (eight schools, host C callback): whole job for the ways a host-callback job can run —
host_persist = 1: one kernel launch per evaluation (round 1; host_groups chain groups pipelined since round 2),
host_persist = 0: resident launches (the kernel keeps the chain state in registers and waits on the device for the host's word)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip
fix = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fixtures", "libeight_schools.so"))
addr = ctypes.cast(fix.eight_schools_logp, ctypes.c_void_p).value
ref = None
variants = [(1, 1), (4, 1), (2, 0), (4, 0), (8, 0), (0, 0), (0, 0), (4, 1), (1, 1)]
for chains in (256, 1024, 64):
    for groups, persist in variants:
        s = hip.PyNutsSettings.Diag(21)
        s.update(num_tune=400, num_draws=1000, num_chains=chains)
        m = hip.HostCallbackModel(10, addr)
        m.set_init("normal")
        smp = hip.PySampler(s, m, host_groups=groups, host_persist=persist)
        smp.wait()
        secs = smp.seconds
        n = int(smp._copy("n_steps", np.int64).sum())
        d = smp._copy("draws", np.float64, vec=True)
        if ref is None or ref[0] != chains:
            ref = (chains, d)
        print(f"chains={chains} host_groups={groups or 'default'} {'launch per evaluation' if persist == 1 else 'resident launches'}: job {secs:.3f} s, "
              f"{n / secs / 1e6:.2f} M leapfrogs/s, identical to the first run: {np.array_equal(d, ref[1])}", flush=True)
        smp.close()
