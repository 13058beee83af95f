"""Round 6: the low-rank job ten times in one process — which hand-ins are slow when a job is slow?"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip, low_rank
from nutpie_amd.radon import radon_symbolic_model
m = radon_symbolic_model().compile()
for rep in range(10):
    s = hip.PyNutsSettings.LowRank(20260926)
    s.update(num_tune=400, num_draws=1000, num_chains=512)
    t0 = time.perf_counter()
    smp = low_rank.make_sampler(m, s, None, 1, None, None, None, None)
    smp.wait()
    wall = time.perf_counter() - t0
    log = list(smp.switch_log)
    slow = [(e[0], e[3], round(e[2] * 1e3, 1), round(e[4] * 1e3, 1)) for e in log if e[2] > 0.008]
    print(f"rep {rep}: wall {wall:.3f} s, engine {smp.seconds:.3f} s, launches {smp.launches}, hand-ins {len(log)}, estimating {sum(e[2] for e in log):.3f} s; slow hand-ins (boundary, chains, ms, at ms): {slow}", flush=True)
    smp.close()
