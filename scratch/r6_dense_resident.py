"""Round 6: the dense Gaussian's resident kernel — parity with the oracle and time per round."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from nutpie_amd import _lib
from nutpie_amd.gaussian import dense_precision

def run(P, mu, chains, tune, draws, seed, **kw):
    s = _lib.PyNutsSettings.Diag(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains)
    t0 = time.perf_counter()
    smp = _lib.PySampler(s, _lib.DenseGaussianModel(P, mu), device=0, **kw)
    smp.wait()
    mode, W, secs, launches = smp.host_mode, smp.waves_per_chain, smp.seconds, smp.launches
    return smp.take_results(), W, mode, secs, launches, time.perf_counter() - t0

print("== parity", flush=True)
for D, chains in ((100, 8), (37, 5), (300, 70), (100, 1), (1000, 6), (129, 1024)):
    P = dense_precision(D, seed=5, cond_lo=0.1, cond_hi=10)
    mu = np.linspace(-1, 1, D)
    tune, draws = (60, 20) if D < 1000 and chains < 1000 else (30, 8)
    for kw in ({}, {"host_persist": 1}, {"evals_per_launch": 7}):
        got, W, mode, secs, launches, wall = run(P, mu, chains, tune, draws, 11, **kw)
        want = oracle.sample_dense(oracle.default_settings(seed=11, num_chains=chains, num_tune=tune, num_draws=draws, waves_per_chain=W, n_threads=16), P, mu)
        eq = {k: bool(np.array_equal(np.asarray(got.stats[k]).astype(np.int64), want.stats[k].astype(np.int64))) for k in ("depth", "n_steps", "diverging", "index_in_trajectory")}
        print(f"D={D} chains={chains} {kw} mode={mode} W={W}: ints {all(eq.values())} draws {bool(np.array_equal(got.draws, want.draws))} energy {bool(np.array_equal(got.stats['energy'], want.stats['energy']))} "
              f"logp {bool(np.array_equal(got.stats['logp'], want.stats['logp']))} maxdiff {float(np.max(np.abs(got.draws - want.draws))):.3g} engine {secs:.3f}s launches {launches}", flush=True)

print("== config 2 (ii): 1000 dims x 1024 chains, tune 30 + draws 10", flush=True)
P = dense_precision(1000)
for kw in ({}, {"evals_per_launch": 64}, {"evals_per_launch": 1024}, {"host_persist": 1}):
    s = _lib.PyNutsSettings.Diag(1)
    s.update(num_tune=30, num_draws=10, num_chains=1024)
    smp = _lib.PySampler(s, _lib.DenseGaussianModel(P), device=0, store_draws=False, **kw)
    smp.wait()
    n = smp._copy("n_steps", np.int64)
    print(f"{kw} mode={smp.host_mode}: {n.sum() / smp.seconds / 1e6:.2f} M leapfrogs/s, {smp.seconds:.3f} s engine, {n.sum()} leapfrogs, launches {smp.launches}, us per evaluation round (leapfrogs / chains) {smp.seconds / (n.sum() / 1024) * 1e6:.1f}", flush=True)
    smp.close()
