"""VERDICT r4 "next" item 2: interrogate the reference-held evidence instead of bounding it.  CPU only (the oracle).

Evidence A — the reference's frozen documentation (tests/golden/reference_doc_step_sizes.json, extracted by
tests/golden/make_reference_doc_step_sizes.py): FINAL step sizes of 36 chains of nuts-rs itself on three analytic models.
Evidence B — tests/golden/reference_halfnormal_{numba,stan}.txt: 2 x 100 (PyMC flavour) and 2 x 10 (Stan flavour) draws of HalfNormal.

For every variant of the recalled warm-up constants (oracle/nuts_oracle.h) the oracle runs an ensemble and the
reference's numbers are placed inside it.  Writes profiles/r5_reference_sensitivity.txt.
    python scratch/r5_reference_sensitivity.py [chains-per-ensemble]
"""
import ctypes
import json
import os
import sys

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
R = 2000                      # runs of the HalfNormal run shape
GOLD = os.path.join(ROOT, "tests", "golden")
doc = json.load(open(os.path.join(GOLD, "reference_doc_step_sizes.json")))
ref_numba = np.loadtxt(os.path.join(GOLD, "reference_halfnormal_numba.txt")).reshape(2, 100)
ref_stan = np.loadtxt(os.path.join(GOLD, "reference_halfnormal_stan.txt")).reshape(2, 10)
fix = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libbs_standin.so"))
halfnormal = ctypes.cast(fix.halfnormal_logp, ctypes.c_void_p).value


def gaussian(name):
    if name == "normal_1d":
        return np.array([4.0]), None, np.array([1.5])
    P = np.array([[301.0, 600.0], [600.0, 1401.0]]) if name == "regression_x123" else np.array([[301.0, 1500.0], [1500.0, 7701.0]])
    b = np.array([600.0, 1400.0]) if name == "regression_x123" else np.array([600.0, 3200.0])
    return np.diag(P).copy(), np.array([P[0, 1]]), np.linalg.solve(P, b)


def doc_values(name, key):
    return np.array([row[key] for run in doc[name]["runs"] for row in run], dtype=np.float64)


def step_size_ensemble(name, settings_kw, init):
    diag, off, mu = gaussian(name)
    s = oracle.default_settings(seed=11, num_chains=N, num_tune=400, num_draws=2, n_threads=8, **settings_kw)
    pts = None
    if init == "pymc":       # support point (0 for a Normal prior) + U(-1, 1) jitter: compile_pymc.py:593-602
        s.init_kind = 2
        pts = np.random.default_rng(5).uniform(-1, 1, size=(N, len(diag)))
    tr = oracle.sample_tridiag(s, diag, off, mu=mu, init_points=pts)
    return tr.stats["step_size"][:, 400], tr.stats["n_steps"][:, 401]


def run_shape_stats(a):
    """a: [2, n] values of `a` of one run"""
    l = np.log(a)
    lc = l - l.mean(1, keepdims=True)
    rep = np.mean(a[:, 1:] == a[:, :-1])
    return np.array([a.mean(), np.median(a), float((lc[:, 1:] * lc[:, :-1]).sum() / max((lc * lc).sum(), 1e-300)), rep, np.log(a).min()])


STAT_NAMES = ["pooled mean", "median", "lag-1 autocorr of log a", "repeat fraction", "min log a"]


fix2 = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libfunnel.so"))


def callback_step_sizes(fn, dim, settings_kw, n):
    s = oracle.default_settings(seed=42, num_chains=n, num_tune=1000, num_draws=40, n_threads=8, init_kind=2, **settings_kw)
    pts = np.random.default_rng(3).uniform(-1, 1, size=(n, dim))
    tr = oracle.sample_callback(s, dim, ctypes.cast(fn, ctypes.c_void_p).value, init_points=pts)
    return tr.stats["step_size"][:, -1], tr.stats["n_steps"][:, 1000:].ravel()


def halfnormal_ensemble(settings_kw, flavour, init_override=None):
    s = oracle.default_settings(seed=123, num_chains=2 * R, num_tune=100, num_draws=100, n_threads=8, **settings_kw)
    pts = None
    init = init_override or ("pymc" if flavour == "numba" else "stan")
    if init == "pymc":
        s.init_kind = 2
        pts = np.random.default_rng(7).uniform(-1, 1, size=(2 * R, 1))     # support point of HalfNormal(1): a = 1, log a = 0
    elif init == "stan":
        s.init_kind = 1                                                     # N(0, 1): src/stan.rs:798-808
    tr = oracle.sample_callback(s, 1, halfnormal, init_points=pts)
    a = np.exp(tr.draws[:, 100:, 0]).reshape(R, 2, 100)
    if flavour == "stan":
        a = a[:, :, :10]
    return np.array([run_shape_stats(a[r]) for r in range(R)])


VARIANTS = [
    ("base (the restatement)", {}, {}),
    ("early_window 0.5", {"early_window": 0.5}, {}),
    ("step_size_window 0.10", {"step_size_window": 0.10}, {}),
    ("early switch freq 20", {"early_mass_matrix_switch_freq": 20}, {}),
    ("late switch freq 50", {"mass_matrix_switch_freq": 50}, {}),
    ("refresh from 1 draw", {}, {"min_refresh": 1}),
    ("refresh from 10 draws", {}, {"min_refresh": 10}),
    ("no step-size search", {}, {"search_mode": 0}),
    ("search at the start only", {}, {"search_mode": 1}),
    ("plain acceptance late", {}, {"late_sym": 0}),
    ("last draw keeps step", {}, {"last_bar": 0}),
    ("window bounds truncated", {}, {"floor_windows": 1}),
    ("target_accept 0.75", {"target_accept": 0.75}, {}),
    ("target_accept 0.85", {"target_accept": 0.85}, {}),
    ("da gamma 0.1", {"da_gamma": 0.1}, {}),
    ("da t0 5", {"da_t0": 5.0}, {}),
    ("da k 0.65", {"da_k": 0.65}, {}),
]

lines = []


def out(s=""):
    print(s, flush=True)
    lines.append(s)


out(f"Reference-held evidence against oracle ensembles ({N} chains per step-size ensemble, {R} runs per HalfNormal ensemble)")
out("A: final step sizes of the reference's frozen docs (z = (reference mean - ensemble mean) / (ensemble sd / sqrt(n)); KS p of the n values against the ensemble;")
out("   f1 = fraction of last draws with ONE gradient evaluation, reference vs ensemble)")
out("B: HalfNormal files: rank of the reference's statistic inside the ensemble of runs of the same shape (0.5 = median)")
out()
refs = {k: doc_values(k, "step_size") for k in ("normal_1d", "regression_x123", "regression_x456")}
ref_f1 = {k: np.mean(doc_values(k, "gradients_last_draw") == 1) for k in refs}
hdr = f"{'variant':28s} | " + " | ".join(f"{k:>32s}" for k in refs) + " | " + f"{'funnel (tune 1000)':>24s} | {'correlated 102-d (tune 1000)':>36s}" + " | numba: " + " ".join(f"{n[:11]:>11s}" for n in STAT_NAMES) + " | stan: mean  repeat"
out(hdr)
ref_fun, ref_102 = doc_values("funnel_diag", "step_size"), doc_values("correlated_102d", "step_size")
ref_102g = doc_values("correlated_102d", "gradients_last_draw")
out(f"{'(reference)':28s} | " + " | ".join(f"n={len(v):2d} mean {v.mean():.3f} sd {v.std(ddof=1):.3f} f1 {ref_f1[k]:.2f}" for k, v in refs.items())
    + f" | n=6 mean {ref_fun.mean():.3f} sd {ref_fun.std(ddof=1):.3f} | n=6 mean {ref_102.mean():.3f} sd {ref_102.std(ddof=1):.3f} grads {ref_102g.mean():.1f}"
    + " | " + " ".join(f"{v:11.3f}" for v in run_shape_stats(ref_numba)) + " | " + " ".join(f"{v:6.3f}" for v in run_shape_stats(ref_stan)[[0, 3]]))
for label, skw, vkw in VARIANTS:
    oracle.set_variant(**vkw)
    cells = []
    for k, v in refs.items():
        ss, nlast = step_size_ensemble(k, skw, "pymc")
        z = (v.mean() - ss.mean()) / (ss.std() / np.sqrt(len(v)))
        p = stats.ks_2samp(v, ss).pvalue
        cells.append(f"{ss.mean():.3f}±{ss.std():.3f} z{z:+5.1f} p{p:.2f} f1 {np.mean(nlast == 1):.2f}")
    sf, _ = callback_step_sizes(fix2.funnel_logp, 6, skw, 300)
    cells.append(f"{sf.mean():.3f}±{sf.std():.3f} z{(ref_fun.mean() - sf.mean()) / (sf.std() / np.sqrt(6)):+5.1f}".rjust(24))
    s1, g1 = callback_step_sizes(fix2.correlated_102d_logp, 102, skw, 128)
    cells.append(f"{s1.mean():.3f}±{s1.std():.3f} z{(ref_102.mean() - s1.mean()) / (s1.std() / np.sqrt(6)):+5.1f} grads {g1.mean():.1f}".rjust(36))
    ens = halfnormal_ensemble(skw, "numba")
    rk = [np.mean(ens[:, j] < run_shape_stats(ref_numba)[j]) for j in range(5)]
    ens_s = halfnormal_ensemble(skw, "stan")
    rs = [np.mean(ens_s[:, j] < run_shape_stats(ref_stan)[j]) for j in (0, 3)]
    out(f"{label:28s} | " + " | ".join(f"{c:>32s}" for c in cells) + " | " + " ".join(f"{v:11.4f}" for v in rk) + " | " + " ".join(f"{v:6.3f}" for v in rs))
oracle.set_variant()
# the initial points of the flavour matter little: the base variant with U(-2, 2) everywhere
ens = halfnormal_ensemble({}, "numba", init_override="uniform")
out(f"{'base, init U(-2, 2)':28s} | " + " " * 104 + " | " + " ".join(f"{np.mean(ens[:, j] < run_shape_stats(ref_numba)[j]):11.4f}" for j in range(5)))
out()
out("(funnel / correlated 102-d: docs/sample-stats.qmd, tune 1000, 300 / 128 chains per ensemble; z as in A; grads = mean gradients per sampling draw, reference 28.3)")
open(os.path.join(ROOT, "profiles", "r5_reference_sensitivity.txt"), "w").write("\n".join(lines) + "\n")
