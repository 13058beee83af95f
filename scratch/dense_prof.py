"""Config 2 (ii) under the profiler: short job, kernel-trace statistics.  usage: dense_prof.py [waves_per_chain]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nutpie_amd
w = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m = nutpie_amd.dense_gaussian(1000)
t = time.time(); tr = nutpie_amd.sample(m, chains=1024, tune=60, draws=20, seed=1, progress_bar=False, return_raw_trace=True, store_draws=False, waves_per_chain=w); el = time.time() - t
n = int(tr.stats["n_steps"].sum()); ticks = int(tr.stats["n_steps"].sum(1).max())
print(f"dense gaussian D=1000, 1024 chains, waves_per_chain={w}: {el:.2f} s, {n/el/1e6:.2f} M leapfrogs/s, {el/ticks*1e6:.0f} us per leapfrog of all chains, ticks {ticks}")
