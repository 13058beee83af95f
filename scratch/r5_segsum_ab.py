"""Round 5: the generated density's segment sums — batched reads of the first `cap` elements against the plain loop.
Run once per mode (NUTPIE_AMD_SEG_MODE=loop | select): per-section cycles of one evaluation (Model.profile) and the config-3 job."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from nutpie_amd import _lib as hip
from nutpie_amd.radon import radon_symbolic_model, radon_traced_model

mode = os.environ.get("NUTPIE_AMD_SEG_MODE", "select")
prof = radon_symbolic_model().profile()
tot = sum(c for _, c in prof)
print(f"[{mode}] one evaluation: {tot:.0f} cycles")
for name, c in prof:
    if c > 150:
        print(f"[{mode}]    {c:8.0f}  {name}")
for label, make in (("generated", lambda: radon_symbolic_model().compile()), ("traced", radon_traced_model)):
    m = make()
    for rep in range(2):
        s = hip.PyNutsSettings.Diag(20260926)
        s.update(num_tune=400, num_draws=1000, num_chains=512)
        t0 = time.perf_counter()
        r = bench.job_rate(m._make_sampler(s, None, 1, None, None, None, None), t0)
        print(f"[{mode}] {label} rep {rep}: {r['leapfrogs_per_s'] / 1e6:.2f} M leapfrogs/s, job {r['job_s']:.3f} s, divergences {r['divergences_sampling']}, "
              f"leapfrogs {r['leapfrogs']}")
