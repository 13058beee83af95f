"""Config 2 (ii) — dense 1000-dim Gaussian, gradient = fp64 GEMM behind the batched device callback — round 5: the callback writes the
engine's staging buffers itself (two kernels, no copies), and `graph_steps` captures (engine kernel, callback) x 16 in one HIP graph."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nutpie_amd
from nutpie_amd import _lib
m = nutpie_amd.dense_gaussian(1000)
for label, kw in (("plain", {}), ("graph_steps=16", {"graph_steps": 16}), ("host_groups=2", {"host_groups": 2}), ("host_groups=2 + graph_steps=16", {"host_groups": 2, "graph_steps": 16}), ("host_groups=4 + graph_steps=16", {"host_groups": 4, "graph_steps": 16})):
    for rep in range(2):
        s = _lib.PyNutsSettings.Diag(1); s.update(num_tune=30, num_draws=10, num_chains=1024)
        t = time.perf_counter()
        smp = m._make_sampler(s, None, 1, None, None, None, None, store_draws=False, **kw); smp.wait()
        el = time.perf_counter() - t
        eng, launches = smp.seconds, smp.launches
        res = smp.take_results()
    n = int(np.asarray(res.stats["n_steps"]).sum())
    print(f"dense D=1000 x 1024 chains, {label}: engine {eng:.2f} s (wall {el:.2f}), {n / eng / 1e6:.2f} M leapfrogs/s, {eng / launches * 1e6:.1f} us per step, {launches} steps", flush=True)
