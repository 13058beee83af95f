"""CHANGELOG.md:124 of the reference: "Update reference draws due to change in window lengths" — the HalfNormal files were regenerated for window
lengths that may differ from the recalled ones.  A grid over the four window constants at the fixture's run shape (tune 100): rank of the
reference's pooled mean / median / lag-1 inside ensembles of 1000 runs of the oracle.  CPU only."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r5_reference_sensitivity.py")).read().split("VARIANTS = [")[0]
sys.argv = ["x", "10"]
exec(src.replace("R = 2000", "R = 1000"))
ref = run_shape_stats(ref_numba)
rows = []
for ew, sw, es, ls in itertools.product((0.1, 0.3, 0.6), (0.05, 0.15, 0.4), (5, 10, 30), (20, 80, 300)):
    ens = halfnormal_ensemble({"early_window": ew, "step_size_window": sw, "early_mass_matrix_switch_freq": es, "mass_matrix_switch_freq": ls}, "numba")
    rk = [float(np.mean(ens[:, j] < ref[j])) for j in range(5)]
    rows.append((ew, sw, es, ls, rk))
    print(ew, sw, es, ls, [round(r, 4) for r in rk], flush=True)
best = sorted(rows, key=lambda r: -min(r[4][0], r[4][1]))[:8]
print("closest to the bulk (rank of pooled mean, median):")
for r in best:
    print(r[:4], [round(x, 4) for x in r[4]])
