import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip
from nutpie_amd.gaussian import ar1_gaussian
def run(dim, chains, noreg, tune=30, draws=5, W=0, E=0, n_local=0):
    m = ar1_gaussian(dim)
    s = hip.PyNutsSettings.Diag(20260926)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains)
    smp = hip.PySampler(s, hip.TridiagGaussianModel(m.diag, m.offdiag), waves_per_chain=W, no_register_kernel=noreg, evals_per_launch=E, n_local_chains=n_local)
    smp.wait()
    return smp.take_results()
for dim in (1000, 200):
    for noreg in (False, True):
        ref = run(dim, 8, noreg)
        for chains in (32, 128, 256, 512, 1024):
            t = run(dim, chains, noreg)
            same = np.array_equal(t.draws[:8], ref.draws)
            if not same:
                bad = np.argwhere(t.draws[:8] != ref.draws)
                c0, d0 = bad[0][0], bad[:, 1].min()
                print(f"dim={dim} noreg={noreg} chains={chains}: DIFFERENT; first differing draw index {d0}; chains affected {sorted(set(bad[:,0].tolist()))}; step0 {t.stats['step_size'][c0,0]} vs {ref.stats['step_size'][c0,0]}; nsteps0 {t.stats['n_steps'][c0,:3]} vs {ref.stats['n_steps'][c0,:3]}")
            else:
                print(f"dim={dim} noreg={noreg} chains={chains}: same")
