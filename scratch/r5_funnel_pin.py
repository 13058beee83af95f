"""The reference's frozen docs (docs/sample-stats.qmd, tune 1000, 6 chains each) hold two more runs of nuts-rs: Neal's funnel and a
102-dimensional correlated Gaussian, both under the default adaptation — final step sizes, last-draw gradient counts, divergences.
The same models through the engine (written with nutpie_amd.symbolic), 1024 chains."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nutpie_amd
from nutpie_amd import symbolic as S

doc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_doc_step_sizes.json")))


def funnel():
    m = S.Model()
    ls = m.param("log_sigma")
    x = m.param("x", dim="k", size=5)
    m.add_logp(S.normal_lpdf(ls, 0.0, 1.0) + S.normal_lpdf(x, 0.0, S.exp(ls)).sum())
    return m


def correlated_102d():
    m = S.Model()
    x, y = m.param("x"), m.param("y")
    z = m.param("z", dim="k", size=100)
    m.add_logp(S.normal_lpdf(x, 0.0, 1.0) + S.normal_lpdf(y, x, 0.01) + S.normal_lpdf(z, y, 1.0).sum())
    return m


if __name__ == "__main__":
    for key, make in (("funnel_diag", funnel), ("correlated_102d", correlated_102d)):
        ref = doc[key]["runs"][0]
        tr = nutpie_amd.sample(nutpie_amd.compile_pymc_model(make()), chains=1024, tune=1000, draws=1000, seed=42, progress_bar=False)
        st = tr.sample_stats
        step, ns, div = st.step_size.values[:, -1], st.n_steps.values, st.diverging.values.sum(1)
        c = collections.Counter(ns[:, -1].tolist())
        rs = np.array([r["step_size"] for r in ref])
        print(f"{key}: reference step {rs.mean():.3f} +- {rs.std(ddof=1):.3f}, grads {[r['gradients_last_draw'] for r in ref]}, divergences {[r['divergences'] for r in ref]}")
        print(f"    engine  step {step.mean():.3f} +- {step.std():.3f} (z = {(rs.mean() - step.mean()) / (step.std() / np.sqrt(len(rs))):+.2f}; pct 5/50/95: {np.percentile(step, [5, 50, 95]).round(3)}), "
              f"mean grads/draw {ns.mean():.1f}, last-draw grads {dict(sorted(c.items()))}, divergences per chain mean {div.mean():.2f} (pct 5/50/95 {np.percentile(div, [5, 50, 95])}), "
              f"chains with 0: {np.mean(div == 0):.2f}", flush=True)
