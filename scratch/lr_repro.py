import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nutpie_amd
from nutpie_amd import low_rank as lr
import symbolic_models as zoo
cm = zoo.radon().compile()
kw = dict(chains=96, tune=300, draws=100, seed=11, progress_bar=False, adaptation="low_rank")
for native in (64, 0):
    lr.NATIVE_EIGH_MAX = native
    a = nutpie_amd.sample(cm, **kw); b = nutpie_amd.sample(cm, **kw)
    ea, eb = a.warmup_sample_stats["energy"].values, b.warmup_sample_stats["energy"].values
    bad = np.argwhere(~((ea == eb) | (np.isnan(ea) & np.isnan(eb))))
    print(f"native eigh up to order {native}: NaN energies {int(np.isnan(ea).sum())} / {int(np.isnan(eb).sum())}, entries that differ {len(bad)}, first {bad[:3].tolist()}, "
          f"n_steps equal {np.array_equal(a.warmup_sample_stats['n_steps'].values, b.warmup_sample_stats['n_steps'].values)}, "
          f"step_size equal {np.array_equal(a.warmup_sample_stats['step_size'].values, b.warmup_sample_stats['step_size'].values)}")
    if len(bad):
        c, t = bad[0]
        print("   ", ea[c, max(0, t - 2):t + 3], eb[c, max(0, t - 2):t + 3])
lr.NATIVE_EIGH_MAX = 64
a = nutpie_amd.sample(cm, **kw)
ea = a.warmup_sample_stats["energy"].values
pos = np.argwhere(np.isnan(ea))
print("NaN energies at (chain, draw):", pos[:40].tolist())
for name in ("n_steps", "depth", "step_size", "diverging", "logp", "energy_error"):
    v = a.warmup_sample_stats[name].values
    print(name, [v[c, t] for c, t in pos[:6]])
print("pauses", lr.pause_draws(300))
