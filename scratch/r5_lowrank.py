"""Round 5: adaptation="low_rank" on the reference's window schedule (nutpie_amd/low_rank.py::window_schedule) against "diag":
radon (config 3's model, 512 chains) under mass_matrix_eigval_cutoff 2 and 100, and the correlated Gaussians of
scratch/lowrank_demo_compiled.py.   python scratch/r5_lowrank.py [radon|demo|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nutpie_amd
from nutpie_amd import _lib, low_rank
from nutpie_amd.radon import radon_symbolic_model

what = (sys.argv[1] if len(sys.argv) > 1 else "all") if __name__ == "__main__" else "none"


def job(m, adaptation, chains, tune, draws, seed=20260926, **kw):
    s = (_lib.PyNutsSettings.LowRank if adaptation == "low_rank" else _lib.PyNutsSettings.Diag)(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, **kw)
    t0 = time.perf_counter()
    smp = low_rank.make_sampler(m, s, None, 1, None, None, None, None) if adaptation == "low_rank" else m._make_sampler(s, None, 1, None, None, None, None)
    smp.wait()
    wall = time.perf_counter() - t0
    log = list(getattr(smp, "switch_log", []))
    fb = getattr(smp, "fallbacks", 0)
    eng = smp.seconds
    res = smp.take_results()
    ns, div = np.asarray(res.stats["n_steps"]), np.asarray(res.stats["diverging"])
    return dict(engine_s=eng, wall_s=wall, lf_per_draw=float(ns[:, tune:].mean()), lf_warm=float(ns[:, :tune].mean()), div=int(div[:, tune:].sum()), div_warm=int(div[:, :tune].sum()),
                hand_ins=len(log), est_s=float(sum(e[2] for e in log)), cols=float(np.mean([e[1] for e in log])) if log else 0.0, fallbacks=fb,
                max_chain_lf=int(ns.sum(1).max()), mean_chain_lf=float(ns.sum(1).mean())), res


def show(label, r):
    print(f"{label:44s} engine {r['engine_s']:.3f} s  wall {r['wall_s']:.2f} s  lf/draw {r['lf_per_draw']:.1f} (warm-up {r['lf_warm']:.1f})  div {r['div']} (warm-up {r['div_warm']})  "
          f"hand-ins {r['hand_ins']} est {r['est_s']:.2f} s cols {r['cols']:.1f} fallbacks {r['fallbacks']}  slowest chain {r['max_chain_lf'] / r['mean_chain_lf']:.2f} x mean", flush=True)


if what in ("radon", "all"):
    m = radon_symbolic_model().compile()
    for rep in range(2):
        r, _ = job(m, "diag", 512, 400, 1000)
    show("radon diag", r)
    for cutoff in (2.0, 100.0):
        for rep in range(2):
            r, _ = job(m, "low_rank", 512, 400, 1000, mass_matrix_eigval_cutoff=cutoff)
        show(f"radon low_rank cutoff {cutoff:g}", r)
    r, _ = job(m, "low_rank", 512, 400, 1000, window_switch_freq=50)
    show("radon low_rank cutoff 2, window_switch_freq 50", r)

if what in ("demo", "all"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from lowrank_demo_compiled import target

    for D, nd, fac in ((60, 3, 400.0), (500, 6, 400.0)):
        cm, Sigma = target(D, nd, fac)
        for adaptation, tune, kw in (("diag", 500, {}), ("low_rank", 500, {}), ("low_rank", 500, {"mass_matrix_eigval_cutoff": 100.0})):
            for rep in range(2):
                r, res = job(cm, adaptation, 256, tune, 500, seed=3, **kw)
            x = np.asarray(res.draws)[:, tune:].reshape(-1, D)
            err = np.abs(np.sqrt(np.diag(np.cov(x.T))) / np.sqrt(np.diag(Sigma)) - 1).max()
            show(f"D={D} {adaptation} {kw} tune {tune}: sd err {err:.3f}", r)
