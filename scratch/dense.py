"""Config 2 variant (ii): dense 1000-dim Gaussian, gradient = fp64 GEMM (rocBLAS through torch) behind the device callback."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nutpie_amd, dataclasses
m = nutpie_amd.dense_gaussian(1000)
for graph in (False, True):
    mm = dataclasses.replace(m, _use_graph=graph)
    t = time.time(); tr = nutpie_amd.sample(mm, chains=1024, tune=150, draws=50, seed=1, progress_bar=False, return_raw_trace=True, store_draws=False); el = time.time() - t
    n = int(tr.stats["n_steps"].sum()); ticks = int(tr.stats["n_steps"].sum(1).max())
    print(f"dense gaussian D=1000, 1024 chains, graph={graph}: {el:.2f} s, {n/el/1e6:.2f} M leapfrogs/s, {el/ticks*1e6:.0f} us per leapfrog of all chains, mean depth {tr.stats['depth'].mean():.2f}")
