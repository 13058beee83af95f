"""The reference's frozen docs hold two runs of Neal's funnel (docs/sample-stats.qmd: tune 1000, 6 chains): adaptation="diag" and
adaptation="low_rank" — final step sizes, last-draw gradient counts, divergences.  The same model through the engine, 1024 chains."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nutpie_amd
from nutpie_amd import symbolic as S

m = S.Model()
ls = m.param("log_sigma")
x = m.param("x", dim="k", size=5)
m.add_logp(S.normal_lpdf(ls, 0.0, 1.0) + S.normal_lpdf(x, 0.0, S.exp(ls)).sum())
cm = nutpie_amd.compile_pymc_model(m)
doc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_doc_step_sizes.json")))
for key, kw in (("funnel_diag", {}), ("funnel_low_rank", {"adaptation": "low_rank"}), ("funnel_low_rank", {"adaptation": "low_rank", "mass_matrix_eigval_cutoff": 2.0}),
                ("funnel_low_rank", {"adaptation": "low_rank", "mass_matrix_eigval_cutoff": 10.0})):
    ref = doc[key]["runs"][0]
    tr = nutpie_amd.sample(cm, chains=1024, tune=1000, draws=1000, seed=42, progress_bar=False, **kw)
    st = tr.sample_stats
    step = st.step_size.values[:, -1]
    ns = st.n_steps.values
    div = st.diverging.values.sum(1)
    c = collections.Counter(ns[:, -1].tolist())
    print(f"{key} {kw}: reference step {np.mean([r['step_size'] for r in ref]):.3f} +- {np.std([r['step_size'] for r in ref], ddof=1):.3f}, grads {[r['gradients_last_draw'] for r in ref]}, divergences {[r['divergences'] for r in ref]}")
    print(f"    engine  step {step.mean():.3f} +- {step.std():.3f} (pct 5/50/95: {np.percentile(step, [5, 50, 95]).round(3)}), mean grads/draw {ns.mean():.1f}, last-draw grads {dict(sorted(c.items()))}, "
          f"divergences per chain mean {div.mean():.1f} (pct 5/50/95 {np.percentile(div, [5, 50, 95])}), chains with 0: {np.mean(div == 0):.2f}", flush=True)
