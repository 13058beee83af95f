"""One profiled job per kernel that had no PMC summary: prints `leapfrogs=<n> launches=<m>` so that the counters of a
pass (profiles/pmc_summary.py: mean per k_advance dispatch) can be put per leapfrog.
    python scratch/r5_pmc_jobs.py c3        config 3's compiled density, 512 chains (the resident kernel of the generated radon density)
    python scratch/r5_pmc_jobs.py c3traced  the same model as a traced torch density
    python scratch/r5_pmc_jobs.py lr D k    the low-rank metric on the register-resident leaf (fused AR(1) Gaussian, metric handed in)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip

what = sys.argv[1]
if what in ("c3", "c3traced"):
    from nutpie_amd.radon import radon_symbolic_model, radon_traced_model

    m = radon_symbolic_model().compile() if what == "c3" else radon_traced_model()
    s = hip.PyNutsSettings.Diag(20260926)
    s.update(num_tune=400, num_draws=1000, num_chains=512)
    smp = m._make_sampler(s, None, 1, None, None, None, None)
    smp.wait()
    n = smp._copy("n_steps", np.int64)
    print(f"{what}: leapfrogs={int(n.sum())} launches={smp.launches} engine_s={smp.seconds:.4f} rate={n.sum() / smp.seconds / 1e6:.2f}M/s")
else:
    from nutpie_amd.gaussian import ar1_gaussian

    dim, k, chains = int(sys.argv[2]), int(sys.argv[3]), (512 if int(sys.argv[2]) < 500 else 1024)
    rng = np.random.default_rng(1)
    model = ar1_gaussian(dim)
    q, _ = np.linalg.qr(rng.normal(size=(dim, max(k, 1))))
    V = np.broadcast_to(q.T[None, :k], (chains, k, dim)).copy()
    lam = np.broadcast_to(np.exp(rng.uniform(np.log(0.5), np.log(2.0), size=k))[None], (chains, k)).copy()
    sig2 = np.broadcast_to(1.0 / model.diag[None], (chains, dim)).copy()
    s = hip.PyNutsSettings.Diag(3)
    s.update(num_tune=200, num_draws=200, num_chains=chains, low_rank_metric=True)
    s.set_pause_draws([20])
    smp = hip.PySampler(s, hip.TridiagGaussianModel(model.diag, model.offdiag), manual=True, store_draws=False)
    t0 = time.perf_counter(); set_ = False; pre = None
    while True:
        done, _, _ = smp.step(4)
        if done:
            break
        if not set_ and smp.waiting().all():
            pre = (int(smp._copy("n_steps", np.int64).sum()), smp.launches)
            smp.set_metric(np.arange(chains), sig2, V, lam); set_ = True
    n = smp._copy("n_steps", np.int64)
    # (the dispatches after the hand-in run under the metric: pmc_summary's "-last_n" = launches - pre[1])
    print(f"lr D={dim} k={k}: leapfrogs={int(n.sum()) - pre[0]} launches={smp.launches - pre[1]} total_launches={smp.launches} engine_s={smp.seconds:.4f}")
