"""Wall time of the public API on a config-2-shaped job (trace copy and conversion included)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nutpie_amd as nutpie

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = nutpie.ar1_gaussian(1000)
for rep in range(2):
    t = time.perf_counter()
    tr = nutpie.sample(m, chains=chains, tune=400, draws=1000, seed=1, progress_bar=False)
    el = time.perf_counter() - t
    n = int(tr.sample_stats.n_steps.values.sum()) + int(tr.warmup_sample_stats.n_steps.values.sum())
    print(f"sample(chains={chains}): {el:.2f} s wall, {n/el/1e6:.1f} M leapfrogs/s end to end, posterior {tr.posterior.x.values.nbytes/1e9:.2f} GB")
    del tr
