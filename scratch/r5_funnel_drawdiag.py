"""Which metric makes the funnel run like the reference's low_rank run (step 0.158, 31 gradients, no divergence)?  adaptation="draw_diag"
(sigma^2 = Var x: 7.4 for x, 1 for log_sigma) against "diag" (sqrt(Var x / Var g): 1 and 0.3)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nutpie_amd
from nutpie_amd import symbolic as S
m = S.Model()
ls = m.param("log_sigma"); x = m.param("x", dim="k", size=5)
m.add_logp(S.normal_lpdf(ls, 0.0, 1.0) + S.normal_lpdf(x, 0.0, S.exp(ls)).sum())
cm = nutpie_amd.compile_pymc_model(m)
for kw in ({"adaptation": "diag"}, {"adaptation": "draw_diag"}, {"adaptation": "diag", "target_accept": 0.95}, {"adaptation": "diag", "target_accept": 0.99}):
    tr = nutpie_amd.sample(cm, chains=1024, tune=1000, draws=1000, seed=42, progress_bar=False, store_mass_matrix=True, **kw)
    st = tr.sample_stats
    step, ns, div = st.step_size.values[:, -1], st.n_steps.values, st.diverging.values.sum(1)
    mm = st.mass_matrix_inv.values[:, -1]
    c = collections.Counter(ns[:, -1].tolist())
    print(f"{kw}: step {step.mean():.3f} +- {step.std():.3f}, mean grads/draw {ns.mean():.1f}, last-draw {dict(sorted(c.items()))}, div/chain mean {div.mean():.1f} none {np.mean(div == 0):.2f}, "
          f"mass_matrix_inv median {np.median(mm, 0).round(2)}", flush=True)
