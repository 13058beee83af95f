"""Regenerates tests/golden/*.

Two kinds of fixtures:

1. ``reference_halfnormal_numba.txt`` / ``reference_halfnormal_stan.txt`` — DATA files held by the
   reference's own tests (``/root/reference/tests/reference/test_deterministic_sampling_numba.txt``,
   ``..._stan.txt``; compared there at atol=rtol=1e-4, tests/test_pymc.py:533-552,
   tests/test_stan.py:282-302).  They pin nuts-rs' exact RNG stream, which cannot be reproduced
   here ("parity unpinned"), so they are used as *distributional* fixtures only.
2. ``oracle_*.npz`` — golden vectors produced by THIS REPO'S CPU oracle (oracle/), clearly labelled
   as such: they pin the oracle against regressions and are what the HIP engine is compared with
   on the GPU box, where neither /root/reference nor a long oracle run is wanted.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402

REF = "/root/reference/tests/reference"


def copy_reference_data():
    for src, dst in [("test_deterministic_sampling_numba.txt", "reference_halfnormal_numba.txt"),
                     ("test_deterministic_sampling_stan.txt", "reference_halfnormal_stan.txt")]:
        p = os.path.join(REF, src)
        if os.path.exists(p):
            shutil.copyfile(p, os.path.join(HERE, dst))


CASES = {
    # name: (settings kwargs, model)
    "stdnormal_d10": (dict(seed=123, num_chains=4, num_tune=400, num_draws=1000), dict(diag=np.ones(10))),
    "ar1_d257": (dict(seed=7, num_chains=6, num_tune=200, num_draws=100), "ar1_257"),
    "diag_d1000_w2": (dict(seed=11, num_chains=4, num_tune=100, num_draws=50, waves_per_chain=2), "diag_1000"),
    "divergent_d3": (dict(seed=5, num_chains=4, num_tune=150, num_draws=100, max_energy_error=0.3), dict(diag=np.array([1.0, 100.0, 0.01]))),
    "maxdepth3_d64": (dict(seed=9, num_chains=4, num_tune=100, num_draws=50, maxdepth=3), "ar1_64"),
}


def model_args(spec):
    if isinstance(spec, dict):
        return spec
    rng = np.random.default_rng(2026)
    if spec.startswith("ar1_"):
        d = int(spec.split("_")[1])
        s = np.exp(0.5 * rng.normal(size=d))
        rho = 0.9
        c = 1.0 / (1 - rho * rho)
        dd = np.full(d, (1 + rho * rho) * c)
        dd[0] = dd[-1] = c
        return dict(diag=dd / s**2, offdiag=-rho * c / (s[:-1] * s[1:]), mu=rng.normal(size=d))
    if spec.startswith("diag_"):
        d = int(spec.split("_")[1])
        return dict(diag=np.exp(rng.normal(size=d)))
    raise KeyError(spec)


def run_case(name):
    kw, spec = CASES[name]
    s = oracle.default_settings(n_threads=8, **kw)
    tr = oracle.sample_tridiag(s, **model_args(spec))
    return tr


def main():
    copy_reference_data()
    for name in CASES:
        tr = run_case(name)
        keep = {k: tr.stats[k] for k in ("depth", "n_steps", "index_in_trajectory", "diverging", "maxdepth_reached", "tuning",
                                         "energy", "logp", "step_size", "step_size_bar", "mean_tree_accept")}
        # draws are thinned to keep the fixture small: every 10th draw, first 8 dims
        np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), draws_thin=tr.draws[:, ::10, :8], **keep)
        print(name, "leapfrogs", int(tr.stats["n_steps"].sum()), "divergences", int(tr.stats["diverging"].sum()))


if __name__ == "__main__":
    main()
