"""Round 6: the in-engine dense Gaussian — (1) is the matrix core's fp64 accumulation the sequential fma chain the oracle assumes?
(2) engine vs oracle on a small job, (3) time per step at BASELINE config 2 (ii)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from nutpie_amd import _lib
from nutpie_amd.gaussian import dense_precision

rng = np.random.default_rng(1)
print("== fp64 MFMA rate (v_mfma_f64_16x16x4_f64 back to back, all SIMDs):", _lib.mfma_f64_rate(0), "TFLOP/s")
print("== gradient GEMM vs the oracle's fma chain")
for n, D in ((5, 3), (64, 64), (70, 100), (33, 257), (128, 1000), (1024, 1000)):
    P = dense_precision(D, seed=3)
    mu = rng.normal(size=D)
    x = rng.normal(size=(n, D)) * 3
    for W in (1, 4):
        g, lp = _lib.test_dense_grad(x, P, mu, waves=W)
        go, lpo = oracle.dense_grad(x, P, mu, waves=W)
        ref = -(x - mu) @ P
        print(f"n={n:5d} D={D:5d} W={W}: grad bit-equal {np.array_equal(g, go)} (mismatches {int((g != go).sum())} of {g.size}, max rel {np.max(np.abs(g - go) / (np.abs(go) + 1e-300)):.2e}); "
              f"logp bit-equal {np.array_equal(lp, lpo)}; vs numpy max rel {np.max(np.abs(g - ref)) / np.max(np.abs(ref)):.2e}")

print("== sampler parity, small job")
D, chains, tune, draws = 100, 8, 60, 20
P = dense_precision(D, seed=5, cond_lo=0.1, cond_hi=10)
for gs in (-1, 0):
    s = _lib.PyNutsSettings.Diag(7)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains)
    smp = _lib.PySampler(s, _lib.DenseGaussianModel(P), device=0, graph_steps=gs)
    smp.wait()
    W = smp.waves_per_chain
    got = smp.take_results()
    want = oracle.sample_dense(oracle.default_settings(seed=7, num_chains=chains, num_tune=tune, num_draws=draws, waves_per_chain=W), P)
    print(f"graph_steps={gs} W={W}:", {k: bool(np.array_equal(np.asarray(got.stats[k]).astype(np.int64), want.stats[k].astype(np.int64))) for k in ("depth", "n_steps", "diverging", "index_in_trajectory")},
          "draws equal", bool(np.array_equal(got.draws, want.draws)), "energy equal", bool(np.array_equal(got.stats["energy"], want.stats["energy"])),
          "max draw diff", float(np.max(np.abs(got.draws - want.draws))))

print("== config 2 (ii): 1000 dims x 1024 chains, tune 30 + draws 10")
P = dense_precision(1000)
for gs in (0,):
    s = _lib.PyNutsSettings.Diag(1)
    s.update(num_tune=30, num_draws=10, num_chains=1024)
    t0 = time.perf_counter()
    smp = _lib.PySampler(s, _lib.DenseGaussianModel(P), device=0, store_draws=False, graph_steps=gs)
    smp.wait()
    n = smp._copy("n_steps", np.int64)
    print(f"graph_steps={gs}: {n.sum() / smp.seconds / 1e6:.2f} M leapfrogs/s, {smp.seconds:.3f} s engine, launches {smp.launches}, us per step {smp.seconds / smp.launches * 1e6:.1f}, wall {time.perf_counter() - t0:.2f}")
    smp.close()
