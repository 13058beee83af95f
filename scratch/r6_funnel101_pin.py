"""Round 6: the one reference-held number that integrates the WHOLE warm-up — docs/nf-adapt.qmd: 101-dim funnel, nutpie.sample(compiled, seed=1)
under the default adaptation: 124 219 gradient evaluations incl. warm-up over 6 chains x (400 + 1000) draws, min bulk ESS 31.46 — against ensembles
of the oracle, and the sensitivity sweep of round 5 (scratch/r5_reference_sensitivity.py) re-run on this statistic."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from nutpie_amd.ess import ess_bulk_all

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_doc_step_sizes.json")))["funnel_101d"]
fix = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libfunnel.so"))
fn = ctypes.cast(fix.funnel_101d_logp, ctypes.c_void_p).value
r_total = ref["totals"]["gradient_evaluations"]
r_step = np.array([r["step_size"] for r in ref["runs"][0]])
R = int(sys.argv[1]) if len(sys.argv) > 1 else 60

def ensemble(label, ess=False, **kw):
    n = 6 * R
    s = oracle.default_settings(seed=1, num_chains=n, num_tune=400, num_draws=1000, n_threads=os.cpu_count(), init_kind=2, **kw)
    pts = np.random.default_rng(7).uniform(-1, 1, size=(n, 101))       # PyMC: support point 0 + U(-1, 1)
    t0 = time.time()
    tr = oracle.sample_callback(s, 101, fn, init_points=pts)
    per_chain = tr.stats["n_steps"].sum(1).astype(np.float64)
    warm = tr.stats["n_steps"][:, :400].sum(1).astype(np.float64)
    runs = per_chain.reshape(R, 6).sum(1)
    step = tr.stats["step_size"][:, -1]
    z_total = (r_total / 6 - per_chain.mean()) / (per_chain.std() / np.sqrt(6))
    z_step = (r_step.mean() - step.mean()) / (step.std() / np.sqrt(6))
    line = (f"{label:42s} total/run {runs.mean():9.0f} +- {runs.std():7.0f} (reference {r_total}: rank {np.mean(runs < r_total):.3f}, z = {z_total:+.2f}); warm-up share {warm.sum() / per_chain.sum():.2f}; "
            f"step {step.mean():.3f} +- {step.std():.3f} (reference {r_step.mean():.3f}: z = {z_step:+.2f}); divergences/chain sampling {tr.stats['diverging'][:, 400:].sum(1).mean():.2f}")
    if ess:
        e = np.array([np.nanmin(ess_bulk_all(tr.draws[6 * r:6 * r + 6, 400:, :], block=101)) for r in range(R)])
        line += f"; min ESS per run pct 5/50/95 {np.percentile(e, [5, 50, 95]).round(1)} (reference {ref['totals']['min_ess']:.1f}: rank {np.mean(e < ref['totals']['min_ess']):.3f})"
    print(line + f"  [{time.time() - t0:.0f} s]", flush=True)

ensemble("restatement (defaults)", ess=True)
if len(sys.argv) > 2:
    for label, kw in (("early_window 0.5", dict(early_window=0.5)), ("step_size_window 0.10", dict(step_size_window=0.10)),
                      ("early switch freq 20", dict(early_mass_matrix_switch_freq=20)), ("switch freq 50", dict(mass_matrix_switch_freq=50)),
                      ("target_accept 0.75", dict(target_accept=0.75)), ("target_accept 0.85", dict(target_accept=0.85)),
                      ("da_gamma 0.1", dict(da_gamma=0.1)), ("da_t0 5", dict(da_t0=5.0)), ("draw_diag (no gradient estimate)", dict(use_grad_based_mass_matrix=0)),
                      ("initial_step 1.0", dict(initial_step=1.0)), ("maxdepth 8", dict(maxdepth=8))):
        ensemble(label, **kw)
    for label, v in (("refresh from 1 draw", dict(min_refresh=1)), ("refresh from 10 draws", dict(min_refresh=10)), ("search: only at chain start", dict(search_mode=0)),
                     ("search: at every refresh", dict(search_mode=1)), ("plain acceptance late", dict(late_sym=0)), ("last draw keeps its step", dict(last_bar=0)),
                     ("truncated window bounds", dict(floor_windows=1))):
        oracle.set_variant(**v)
        ensemble("variant: " + label)
        oracle.set_variant()
