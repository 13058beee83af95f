import sys, time, numpy as np
sys.path.insert(0, '.')
import oracle
from nutpie_amd import _lib

rng = np.random.default_rng(0)
# 1. detmath parity
for fn, x in [("exp", rng.uniform(-700, 700, 200000)), ("log", np.exp(rng.uniform(-300, 300, 200000))),
              ("log1p", rng.uniform(0, 1, 200000)), ("sin2pi", rng.uniform(0, 1, 200000)), ("cos2pi", rng.uniform(0, 1, 200000))]:
    a = _lib.test_detmath(fn, x); b = oracle.detmath(fn, x)
    print(fn, "bit-equal:", np.array_equal(a.view(np.uint64), b.view(np.uint64)), "mismatch", (a.view(np.uint64) != b.view(np.uint64)).sum())
x = np.exp(rng.uniform(-300, 300, 400000))
print("sqrt exact:", np.array_equal(_lib.test_detmath("sqrt", x), np.sqrt(x)), "recip exact:", np.array_equal(_lib.test_detmath("recip", x), 1.0 / x))
a = _lib.test_normals(123, 5, 7, 1, 1001); b = oracle.normals(123, 5, 7, 1, 1001)
print("normals equal:", np.array_equal(a, b))
for W in (1, 2, 4, 8, 16):
    for n in (1, 127, 1000, 10000):
        x = rng.normal(size=n); y = rng.normal(size=n)
        a = _lib.test_dot(x, y, W); b = oracle.dot(x, y, W)
        if a != b: print("DOT MISMATCH", W, n, a, b)
print("dot done")

def compare(dim, chains, tune, draws, W=0, seed=42, diag=None, off=None, mu=None, **kw):
    diag = np.ones(dim) if diag is None else diag
    s = _lib.PyNutsSettings.Diag(seed)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains, store_gradient=True, store_mass_matrix=True, **kw)
    m = _lib.TridiagGaussianModel(diag, off, mu)
    t = time.time()
    smp = _lib.PySampler(s, m, waves_per_chain=W)
    smp.wait()
    el = time.time() - t
    Wused = smp.waves_per_chain
    secs, launches = smp.seconds, smp.launches
    tr = smp.take_results()
    os_ = oracle.default_settings(seed=seed, num_tune=tune, num_draws=draws, num_chains=chains, n_threads=8,
                                  waves_per_chain=Wused, store_gradient=1, store_mass_matrix=1, **{k: v for k, v in kw.items()})
    otr = oracle.sample_tridiag(os_, diag, off, mu)
    ok = True
    for k in ("depth", "n_steps", "index_in_trajectory", "diverging", "maxdepth_reached", "tuning"):
        eq = np.array_equal(tr.stats[k].astype(np.int64), otr.stats[k].astype(np.int64))
        ok &= eq
        if not eq:
            bad = np.argwhere(tr.stats[k].astype(np.int64) != otr.stats[k].astype(np.int64))
            print("  MISMATCH", k, len(bad), "first", bad[0], tr.stats[k][tuple(bad[0])], otr.stats[k][tuple(bad[0])])
    for k in ("energy", "logp", "step_size", "step_size_bar", "mean_tree_accept", "mean_tree_accept_sym", "energy_error"):
        eq = np.array_equal(tr.stats[k], otr.stats[k]); ok &= eq
        if not eq: print("  MISMATCH", k, np.abs(tr.stats[k] - otr.stats[k]).max())
    eq = np.array_equal(tr.draws, otr.draws); ok &= eq
    if not eq: print("  MISMATCH draws", np.abs(tr.draws - otr.draws).max())
    for k in ("gradient", "mass_matrix_inv"):
        eq = np.array_equal(tr.stats[k], otr.stats[k]); ok &= eq
        if not eq: print("  MISMATCH", k, np.abs(tr.stats[k] - otr.stats[k]).max())
    nl = tr.stats["n_steps"].sum()
    print(f"dim={dim} chains={chains} W={Wused} bit-identical={ok} gpu {secs:.3f}s ({el:.3f}s wall, {launches} launches) leapfrogs={nl} -> {nl/secs:.3e}/s ; oracle {otr.seconds:.3f}s (8 thr) -> {nl/otr.seconds:.3e}/s")
    return ok

compare(10, 4, 100, 100)
compare(10, 4, 400, 1000, seed=123)
compare(301, 8, 200, 100, diag=np.exp(rng.normal(size=301)), off=0.3 * rng.normal(size=300) * 0.2, mu=rng.normal(size=301))
compare(1000, 16, 100, 50, W=1)
compare(1000, 16, 100, 50, W=2)
compare(1000, 16, 100, 50, W=4)
compare(5000, 8, 60, 20)
compare(3, 4, 200, 200, max_energy_error=0.5, store_divergences=False)
compare(64, 8, 150, 50, maxdepth=3)
