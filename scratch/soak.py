"""Longer job: config-2 shape, tune 1000 + draws 5000, positions not stored; checks completion and basic sanity."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip
from nutpie_amd.gaussian import ar1_gaussian
m = ar1_gaussian(1000)
s = hip.PyNutsSettings.Diag(99); s.update(num_tune=1000, num_draws=5000, num_chains=1024)
t = time.time(); smp = hip.PySampler(s, hip.TridiagGaussianModel(m.diag, m.offdiag), store_draws=False); smp.wait(); el = time.time() - t
tr = smp.take_results()
n = int(tr.stats["n_steps"].sum())
print(f"soak: {el:.1f} s, {n/1e9:.2f} G leapfrogs, {n/el/1e6:.1f} M/s, finished {tr.finished.min()}..{tr.finished.max()}, div(post) {int(tr.stats['diverging'][:,1000:].sum())}, accept {tr.stats['mean_tree_accept'][:,1000:].mean():.3f}, depth {tr.stats['depth'][:,1000:].mean():.2f}")
