import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nutpie_amd
from nutpie_amd.radon import radon_model
m = radon_model(use_graph=True)
t = time.time()
tr = nutpie_amd.sample(m, chains=512, tune=100, draws=100, seed=1, progress_bar=False, return_raw_trace=True)
el = time.time() - t
n = tr.stats["n_steps"].sum()
print(f"radon 512 chains graph: {el:.2f}s, {n} leapfrogs, {n/el/1e6:.2f} M leapfrogs/s, steps(max over chains per draw summed) {tr.stats['n_steps'].max(0).sum()}")
