"""The reference's own golden-vector tests, through the HIP engine.

The reference holds three data files that pin its sampler end to end (``tests/reference/test_deterministic_sampling_*.txt``,
committed here as ``tests/golden/reference_halfnormal_{numba,stan}.txt``): ``HalfNormal("a")`` through PyMC
(tests/test_pymc.py:533-552) and ``real<lower=0> a; a ~ normal(0, 1)`` through Stan (tests/test_stan.py:282-302), both with
``chains=2, seed=123, draws=100, tune=100``.  They depend on nuts-rs' ChaCha8 stream, which cannot be reproduced ("parity
unpinned", oracle/nuts_oracle.h), so they cannot be matched value for value; what CAN be checked on the GPU is that the engine,
run in the reference's run shape on the reference's density (x = log a: logp = x - exp(2x)/2), produces the same LAW as those
files — and, bit for bit, what the CPU oracle produces.  Every flavour of the boundary is used: the raw C logp + expand
callbacks a compiled PyMC model carries, a batched device density, and BridgeStan's C API.
"""
import ctypes
import os

import numpy as np
import pytest
from scipy import stats

import nutpie_amd
from tests.conftest import GOLDEN, FakeBridgeStanModel, assert_trace_equal, fn_addr

pytestmark = pytest.mark.gpu

RUN = dict(chains=2, seed=123, draws=100, tune=100)    # the reference's run shape
MEAN = float(np.sqrt(2 / np.pi))                       # E|Z| = 0.798


def reference_values(flavour):
    return np.loadtxt(os.path.join(GOLDEN, f"reference_halfnormal_{flavour}.txt"))


def check_halfnormal_law(a_long, a_short, ref):
    """a_long: [chains, draws] of a long run; a_short: the 200 values of the reference's run shape; ref: the reference's values."""
    assert np.all(a_long > 0) and np.all(a_short > 0)
    assert abs(a_long.mean() - MEAN) < 0.02
    assert abs((a_long**2).mean() - 1.0) < 0.05                               # E Z^2 = 1
    thin = a_long[:, ::10].ravel()
    assert stats.kstest(thin, "halfnorm").pvalue > 1e-3
    # the reference's values and ours in the same run shape: 200 strongly autocorrelated draws after 100 tuning draws (chain
    # means 0.48 / 0.64 in the reference's file) — compared with the long run by KS distance and summary bands, not by p-value
    # (the calibrated form of this comparison — the reference's values ranked inside an ensemble of 1000 runs of the same shape — is
    #  test_reference_values_against_a_calibrated_ensemble; here only coarse bands, for every flavour of the boundary)
    for sample in (ref.ravel(), a_short.ravel()):
        assert sample.max() < 4.5 and 0.3 < sample.mean() < 1.3
        assert stats.ks_2samp(sample, thin).statistic < 0.35


def test_halfnormal_raw_callbacks_pymc_flavour(hip, oracle, bs_standin):
    from nutpie_amd.compile_pymc import from_raw_callback

    logp, expand = fn_addr(bs_standin.halfnormal_logp), fn_addr(bs_standin.halfnormal_expand)
    m = from_raw_callback(1, logp, expand_address=expand, expanded_shapes={"a": ()}, keep_alive=bs_standin)
    tr = nutpie_amd.sample(m, progress_bar=False, store_unconstrained=True, **RUN)
    a = tr.posterior.a.values
    assert a.shape == (2, 100)
    # bit for bit what the CPU oracle produces on the same C function
    want = oracle.sample_callback(oracle.default_settings(seed=123, num_chains=2, num_tune=100, num_draws=100), 1, logp)
    assert np.array_equal(tr.sample_stats.unconstrained_draw.values[..., 0], want.draws[:, 100:, 0])
    assert np.array_equal(tr.sample_stats.n_steps.values, want.stats["n_steps"][:, 100:])
    np.testing.assert_allclose(a, np.exp(want.draws[:, 100:, 0]), rtol=4e-16, atol=0)   # the C expand callback is libm's exp (numpy's differs by an ulp)
    long = nutpie_amd.sample(m, chains=64, seed=5, draws=1000, tune=300, progress_bar=False)
    check_halfnormal_law(long.posterior.a.values, a, reference_values("numba"))
    assert long.sample_stats.diverging.values.mean() < 0.01


def run_shape_summary(x):
    """Statistics of ONE run in the reference's shape (x: [2 chains, 100 draws] of `a`): the two chain means (sorted), the pooled
    mean, the KS distance of the 200 values to HalfNormal(1), the lag-1 autocorrelation of log a."""
    m = np.sort(x.mean(1))
    l = np.log(x)
    l = l - l.mean(1, keepdims=True)
    return {"chain_mean_lo": m[0], "chain_mean_hi": m[1], "pooled_mean": x.mean(), "ks_to_halfnormal": stats.kstest(x.ravel(), "halfnorm").statistic,
            "lag1_autocorrelation": float((l[:, 1:] * l[:, :-1]).sum() / (l * l).sum())}


def test_reference_values_against_a_calibrated_ensemble(hip, bs_standin):
    """the law test, quantified.  The engine runs the reference's run shape (2 chains, tune 100, draws 100:
    tests/test_pymc.py:533-552) 4000 times — one job of 8000 chains; a chain's stream is keyed by (seed, chain id), so chains
    (2k, 2k + 1) are run k and run 0 is the very call the reference's test makes — and the reference's 200 values are ranked inside
    that ensemble, statistic by statistic.

    Measured (the same 4000 runs of the CPU oracle, which is this sampler bit for bit): the reference's realisation lies in the
    TAILS of this sampler's law — pooled mean 0.560 at rank 0.0010 (4 of 4000 runs below it; ensemble median 0.797), chain means
    0.477 / 0.643 at ranks 0.005 / 0.008, KS distance to HalfNormal(1) 0.211 and lag-1 autocorrelation of log a 0.853 at rank
    0.992 — one excursion (six consecutive draws below 0.13 in chain 0) seen by five correlated statistics.  One realisation
    cannot tell a 1-in-1000 draw of the same law from a sampler whose short warm-up adapts differently (the constants of nuts-rs
    0.18.3's adaptation are recalled, not pinned: oracle/nuts_oracle.h), so this stays a law test — but a quantified one: every
    statistic of the reference's file lies INSIDE the range the ensemble spans (rank between 1 / R and 1 - 1 / R), the ensemble
    itself is calibrated against HalfNormal(1), and the ranks are printed.  (The previous test accepted any KS distance below
    0.35 — looser than the ensemble's own 99.9th percentile, 0.23.)"""
    from nutpie_amd.compile_pymc import from_raw_callback

    R = 4000
    m = from_raw_callback(1, fn_addr(bs_standin.halfnormal_logp), expand_address=fn_addr(bs_standin.halfnormal_expand), expanded_shapes={"a": ()},
                          keep_alive=bs_standin)
    tr = nutpie_amd.sample(m, chains=2 * R, seed=123, draws=100, tune=100, progress_bar=False)
    a = tr.posterior.a.values.reshape(R, 2, 100)
    ens = [run_shape_summary(a[r]) for r in range(R)]
    ref = run_shape_summary(reference_values("numba").reshape(2, 100))
    ranks = {}
    for k, v in ref.items():
        col = np.array([e[k] for e in ens])
        ranks[k] = float(np.mean(col < v))
        assert 1.0 / R <= ranks[k] <= 1.0 - 1.0 / R, f"{k}: the reference's {v:.4f} lies outside the ensemble ({np.percentile(col, [0.1, 50, 99.9])})"
    # round 5: the real assertions.  What is NOT dominated by the time the file spends in the left tail lies
    # inside the 0.5 - 99.5 % band: the lag-1 autocorrelation, the fraction of exact repeats, the depth of the deepest excursion.
    # The pooled mean / chain means do not (rank ~ 0.001 under every variant of the recalled warm-up constants: oracle/nuts_oracle.h
    # item 10, profiles/r5_reference_sensitivity.txt) — asserted here only as "inside the ensemble's range", and stated as open.
    x = np.log(a)
    extra = {"repeat_fraction": (np.mean(a[:, :, 1:] == a[:, :, :-1], axis=(1, 2)), np.mean(np.diff(reference_values("numba").reshape(2, 100), axis=1) == 0)),
             "min_log_a": (x.min(axis=(1, 2)), np.log(reference_values("numba")).min())}
    for k, (col, v) in extra.items():
        ranks[k] = float(np.mean(col < v))
    for k in ("lag1_autocorrelation", "repeat_fraction", "min_log_a"):
        assert 0.005 <= ranks[k] <= 0.995, (k, ranks[k])
    print("ranks of the reference's values in the ensemble:", {k: round(v, 4) for k, v in ranks.items()})
    # the ensemble itself: the law of this run shape against HalfNormal(1)
    pooled = np.array([e["pooled_mean"] for e in ens])
    assert abs(np.median(pooled) - MEAN) < 0.02 and abs(pooled.mean() - MEAN) < 0.01
    assert np.median([e["ks_to_halfnormal"] for e in ens]) < 0.12
    assert abs((a**2).mean() - 1.0) < 0.02                                    # E Z^2 = 1 over 800 000 draws
    assert stats.kstest(a[:, :, ::20].ravel(), "halfnorm").pvalue > 1e-3


def test_halfnormal_device_density(hip):
    import torch

    def make_logp():
        def f(x):
            e = torch.exp(2.0 * x[:, 0])
            return x[:, 0] - 0.5 * e, (1.0 - e)[:, None]

        return f

    m = nutpie_amd.from_torchfunc(1, make_logp, expand_device_fn=lambda x: {"a": torch.exp(x[:, 0])}, expanded_names=["a"], expanded_shapes=[()])
    short = nutpie_amd.sample(m, progress_bar=False, **RUN)
    long = nutpie_amd.sample(m, chains=64, seed=5, draws=1000, tune=300, progress_bar=False)
    check_halfnormal_law(long.posterior.a.values, short.posterior.a.values, reference_values("numba"))
    assert long.sample_stats.diverging.values.mean() < 0.01


def test_halfnormal_bridgestan_flavour(hip, oracle, bs_standin):
    from nutpie_amd.compile_stan import CompiledStanModel

    m = CompiledStanModel(dims={}, code="parameters { real<lower=0> a; } model { a ~ normal(0, 1); } generated quantities { real b = normal_rng(0, 1) + a; }",
                          model=FakeBridgeStanModel(bs_standin, b"halfnormal"))
    assert m.n_dim == 1 and m.shapes == {"a": (), "b": ()}
    tr = nutpie_amd.sample(m, progress_bar=False, store_unconstrained=True, **RUN)
    tr2 = nutpie_amd.sample(m, progress_bar=False, **RUN)
    # tests/test_stan.py:298-301: two runs with the same seed agree to the last bit, generated quantities included
    np.testing.assert_array_max_ulp(tr.posterior.a.values, tr2.posterior.a.values, maxulp=0)
    np.testing.assert_array_max_ulp(tr.posterior.b.values, tr2.posterior.b.values, maxulp=0)
    first10 = tr.posterior.a.values[:, :10]                                     # what the reference's test returns
    ref = reference_values("stan")
    assert first10.shape == ref.shape == (2, 10) and np.all(first10 > 0)
    # bit for bit the oracle on the same density with Stan's initial points (N(0, 1): src/stan.rs:798-808)
    ldg = bs_standin.bs_log_density_gradient

    def density(x):
        val, grad, err = ctypes.c_double(), (ctypes.c_double * 1)(), ctypes.c_char_p()
        theta = (ctypes.c_double * 1)(*x)
        rc = ldg(m.model.model, True, True, theta, ctypes.byref(val), grad, ctypes.byref(err))
        assert rc == 0
        return val.value, np.array([grad[0]])

    want = oracle.sample_callback(oracle.default_settings(seed=123, num_chains=2, num_tune=100, num_draws=100, init_kind=1), 1, density)
    assert np.array_equal(tr.sample_stats.unconstrained_draw.values[..., 0], want.draws[:, 100:, 0])
    np.testing.assert_allclose(tr.posterior.a.values, np.exp(want.draws[:, 100:, 0]), rtol=4e-16, atol=0)
    long = nutpie_amd.sample(m, chains=64, seed=5, draws=1000, tune=300, progress_bar=False)
    check_halfnormal_law(long.posterior.a.values, tr.posterior.a.values, ref)
    # generated quantity: b - a ~ N(0, 1), one generator per chain (src/stan.rs:787-796): chains differ, law is right
    noise = (long.posterior.b.values - long.posterior.a.values)
    assert stats.kstest(noise[:, ::7].ravel(), "norm").pvalue > 1e-3
    assert not np.allclose(noise[0, :50], noise[1, :50])


def test_bridgestan_expand_reorders_column_major_blocks(hip, bs_standin):
    # src/stan.rs:507-516, 671-711: Stan writes matrices column-major, the trace holds them in C order — natively, per draw
    from nutpie_amd.compile_stan import CompiledStanModel

    m = CompiledStanModel(dims={"m": ("r", "c")}, model=FakeBridgeStanModel(bs_standin, b"matrix"))
    assert m.n_dim == 6 and m.shapes == {"m": (2, 3), "mt": (3, 2), "s": ()}
    tr = nutpie_amd.sample(m, chains=4, seed=9, draws=50, tune=60, progress_bar=False, store_unconstrained=True)
    x = tr.sample_stats.unconstrained_draw.values                               # the column-major serialisation of m
    mm = tr.posterior.m.values
    assert mm.shape == (4, 50, 2, 3)
    assert np.array_equal(mm, x.reshape(4, 50, 3, 2).transpose(0, 1, 3, 2))
    assert np.array_equal(tr.posterior.mt.values, mm.transpose(0, 1, 3, 2))
    assert np.all(np.abs(tr.posterior.s.values - mm.sum((-1, -2))) < 6.0) and not np.allclose(tr.posterior.s.values, mm.sum((-1, -2)))
    # a Stan error in the expand step fails the hand-off the way the reference words it (src/stan.rs:493-494)
    model = hip.BridgeStanModel(6, bs_standin, m.model.model)
    model.set_bridgestan_expand(13)
    model.set_init("explicit", np.full((1, 6), 2e6))
    s = hip.PyNutsSettings.Diag(1)
    s.update(num_tune=3, num_draws=2, num_chains=1, step_size_adapt_method="1e-9", adapt_mass_matrix=False)
    smp = hip.PySampler(s, model)
    smp.wait()
    with pytest.raises(RuntimeError, match="Failed to constrain the parameters of the draw: constrain failed"):
        smp.expanded()
    smp.close()


@pytest.mark.parametrize("name", ["normal_1d", "regression_x123", "regression_x456"])
def test_engine_reproduces_the_final_step_sizes_of_the_reference_docs(hip, name):
    """tests/test_oracle_reference_pins.py on the GPU: 4000 chains of the fused Gaussian kernel on the three analytic models of the
    reference's frozen documentation (tests/golden/reference_doc_step_sizes.json: final step sizes and last-draw gradient counts of
    36 chains of nuts-rs itself, default settings: tune 400)."""
    from tests.test_oracle_reference_pins import doc_values, gaussian

    diag, off, mu = gaussian(name)
    n = 4000
    s = hip.PyNutsSettings.Diag(11)
    s.update(num_tune=400, num_draws=20, num_chains=n)
    m = hip.TridiagGaussianModel(diag, off, mu=mu)
    m.set_init("explicit", np.random.default_rng(5).uniform(-1, 1, size=(n, len(diag))))      # PyMC: support point 0 + U(-1, 1)
    smp = hip.PySampler(s, m)
    smp.wait()
    got = smp.take_results()
    ours = np.asarray(got.stats["step_size"])[:, 400]
    ref = doc_values(name, "step_size")
    z = (ref.mean() - ours.mean()) / (ours.std() / np.sqrt(len(ref)))
    print(f"{name}: reference {ref.mean():.3f} +- {ref.std(ddof=1):.3f} (n = {len(ref)}), engine {ours.mean():.3f} +- {ours.std():.3f}, z = {z:+.2f}")
    assert abs(z) < 3.0 and stats.ks_2samp(ref, ours).pvalue > 0.01
    g_ref, g = doc_values(name, "gradients_last_draw"), np.asarray(got.stats["n_steps"])[:, 400:].ravel()
    assert set(g_ref.astype(int).tolist()) <= set(np.unique(g).tolist())
    assert abs(g_ref.mean() - g.mean()) / (g.std() / np.sqrt(len(g_ref))) < 3.0


@pytest.mark.parametrize("key", ["funnel_diag", "correlated_102d"])
def test_engine_on_the_funnel_and_the_correlated_model_of_the_reference_docs(hip, key):
    """docs/sample-stats.qmd (tune 1000, default adaptation; tests/golden/reference_doc_step_sizes.json): Neal's funnel and the 102-dimensional
    Gaussian with one stiff direction, written with the front-end and sampled on their resident kernels — final step sizes, gradients per
    draw and divergences of 1024 chains against the six chains of nuts-rs (the oracle's side: tests/test_oracle_reference_pins.py)."""
    import json

    from tests.symbolic_models import DOC_MODELS

    ref = json.load(open(os.path.join(GOLDEN, "reference_doc_step_sizes.json")))[key]["runs"][0]
    model = nutpie_amd.compile_pymc_model(DOC_MODELS[key]())
    tr = nutpie_amd.sample(model, chains=1024, tune=1000, draws=400, seed=42, progress_bar=False)
    st = tr.sample_stats
    step, g, div = st.step_size.values[:, -1], st.n_steps.values.ravel(), st.diverging.values.sum(1) * (1000 / 400)
    r_step, r_last, r_div = (np.array([row[k] for row in ref], dtype=np.float64) for k in ("step_size", "gradients_last_draw", "divergences"))
    z = (r_step.mean() - step.mean()) / (step.std() / np.sqrt(len(r_step)))
    print(f"{key}: reference step {r_step.mean():.3f}, engine {step.mean():.3f} +- {step.std():.3f} (z = {z:+.2f}); gradients per draw reference {r_last.mean():.1f}, engine {g.mean():.1f}")
    assert abs(z) < 3.0
    assert set(r_last.astype(int).tolist()) <= set(np.unique(g).tolist())
    if key == "correlated_102d":
        assert abs(r_last.mean() - g.mean()) / (g.std() / np.sqrt(len(r_last))) < 3.0 and div.sum() == 0
    else:
        assert np.all(r_div <= np.percentile(div, 99.5)) and stats.mannwhitneyu(r_div, div).pvalue > 0.01


def test_engine_total_gradient_evaluations_of_the_101_dimensional_funnel(hip):
    """docs/nf-adapt.qmd:60-78, 115-122: the 101-dimensional funnel under the default adaptation, seed 1 — nuts-rs took 124 219 gradient evaluations
    for 6 chains x (400 + 1000) draws, warm-up INCLUDED (tests/golden/reference_doc_step_sizes.json: "funnel_101d"; the oracle's side with the details:
    tests/test_oracle_reference_pins.py).  The model written with the front-end, 1020 chains = 170 runs of the reference's shape on the resident
    kernel of its generated density: the reference's total, step sizes and minimum ESS inside the engine's ensemble."""
    import json

    from nutpie_amd.ess import ess_bulk_all
    from tests.symbolic_models import DOC_MODELS

    ref = json.load(open(os.path.join(GOLDEN, "reference_doc_step_sizes.json")))["funnel_101d"]
    model = nutpie_amd.compile_pymc_model(DOC_MODELS["funnel_101d"]())
    R = 170
    tr = nutpie_amd.sample(model, chains=6 * R, tune=400, draws=1000, seed=1, progress_bar=False)
    per_chain = (tr.sample_stats.n_steps.values.sum(1) + tr.warmup_sample_stats.n_steps.values.sum(1)).astype(np.float64)
    runs = per_chain.reshape(R, 6).sum(1)
    r_total = ref["totals"]["gradient_evaluations"]
    rank_total = float(np.mean(runs < r_total))
    z_total = (r_total / 6 - per_chain.mean()) / (per_chain.std() / np.sqrt(6))
    step = tr.sample_stats.step_size.values[:, -1]
    r_step = np.array([row["step_size"] for row in ref["runs"][0]])
    z_step = (r_step.mean() - step.mean()) / (step.std() / np.sqrt(6))
    x = np.concatenate([tr.posterior["log_sigma"].values[..., None], tr.posterior["x"].values], axis=2)
    ess = np.array([np.nanmin(ess_bulk_all(x[6 * r:6 * r + 6], block=101)) for r in range(R)])
    rank_ess = float(np.mean(ess < ref["totals"]["min_ess"]))
    print(f"101-d funnel on the engine: gradient evaluations per run {runs.mean():.0f} +- {runs.std():.0f}, reference {r_total} (rank {rank_total:.3f}, z = {z_total:+.2f}); "
          f"step {step.mean():.3f} +- {step.std():.3f}, reference {r_step.mean():.3f} (z = {z_step:+.2f}); min ESS per run pct 5/50/95 {np.percentile(ess, [5, 50, 95]).round(1)}, "
          f"reference {ref['totals']['min_ess']:.1f} (rank {rank_ess:.2f})")
    assert 0.005 <= rank_total <= 0.995 and abs(z_total) < 3.0
    assert abs(z_step) < 3.0
    assert 0.02 <= rank_ess <= 0.98
    assert tr.sample_stats.diverging.values.sum(1).mean() < 1.0
