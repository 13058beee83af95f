"""Config 2 (ii): waves per chain of the launch-per-evaluation kernel behind a device callback (dense Gaussian, D = 1000 and 200)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nutpie_amd
for D in tuple(int(a) for a in sys.argv[1:]) or (1000, 200):
    m = nutpie_amd.dense_gaussian(D)
    for w in (1, 2, 4, 8):
        t = time.time(); tr = nutpie_amd.sample(m, chains=1024, tune=60, draws=20, seed=1, progress_bar=False, return_raw_trace=True, store_draws=False, waves_per_chain=w); el = time.time() - t
        n = int(tr.stats["n_steps"].sum()); ticks = int(tr.stats["n_steps"].sum(1).max())
        print(f"dense gaussian D={D}, 1024 chains, waves_per_chain={w}: {el:.2f} s, {n/el/1e6:.2f} M leapfrogs/s, {el/ticks*1e6:.0f} us per leapfrog of all chains", flush=True)
