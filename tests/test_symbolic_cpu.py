"""The expression front-end on the host: the symbolic gradient against finite differences, the generated source against the
compiler (hipcc cross-compiles without a GPU), data swapping."""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import symbolic_models as zoo  # noqa: E402

from nutpie_amd import symbolic as S  # noqa: E402


def _numeric_grad(f, x, h=1e-6):
    g = np.zeros_like(x)
    for i in range(x.size):
        e = np.zeros_like(x)
        e[i] = h
        g[i] = (f(x + e) - f(x - e)) / (2 * h)
    return g


@pytest.mark.parametrize("name", list(zoo.ALL))
def test_symbolic_gradient_matches_finite_differences(name):
    m = zoo.ALL[name]()
    logp = m.logp_expr()
    grads = S.gradient(logp, m._params)
    rng = np.random.default_rng(1)
    x = 0.3 * rng.normal(size=(3, m.n_dim))
    vals = S.evaluate([logp] + grads, x, m._data)
    assert np.all(np.isfinite(vals[0]))
    for r in range(x.shape[0]):
        f = lambda v: float(S.evaluate([logp], v[None, :], m._data)[0][0])  # noqa: E731
        num = _numeric_grad(f, x[r])
        ana = np.zeros(m.n_dim)
        for p, v in zip(m._params, vals[1:]):
            if p.dim is None:
                ana[p.payload] = v[r]
            else:
                off, nv = p.payload
                ana[off:off + nv] = (v[r] if v.ndim == 2 else np.full(nv, v[r]))[:nv]
        np.testing.assert_allclose(ana, num, rtol=2e-5, atol=2e-6)


def test_radon_graph_is_the_hand_written_density():
    """same value as the closed form of nutpie_amd/radon.py's density (restated here with numpy)"""
    from nutpie_amd.radon import synthetic_radon_data

    d = synthetic_radon_data()
    m = zoo.radon(d)
    n = int(d["county_idx"].max()) + 1
    rng = np.random.default_rng(2)
    x = 0.2 * rng.normal(size=m.n_dim)

    def ext(v):
        k = v.size + 1
        s = v.sum()
        return np.concatenate([v - s / (np.sqrt(k) + k), [-s / np.sqrt(k)]])

    icpt, raw, lsd, fe, craw, lcsd, lsig = x[0], x[1:n], x[n], x[n + 1], x[n + 2:2 * n + 1], x[2 * n + 1], x[2 * n + 2]
    sd, csd, sig = np.exp(lsd), np.exp(lcsd), np.exp(lsig)
    mu = icpt + (ext(raw) * sd)[d["county_idx"]] + d["floor"] * (fe + (ext(craw) * csd)[d["county_idx"]])
    r = (d["log_radon"] - mu) / sig
    want = (-0.005 * icpt ** 2 - 0.125 * fe ** 2 - 0.5 * (raw ** 2).sum() - 0.5 * (craw ** 2).sum() - 0.5 * sd ** 2 + lsd - 0.5 * csd ** 2 + lcsd
            - (0.5 / 2.25) * sig ** 2 + lsig - 0.5 * (r ** 2).sum() - r.size * lsig)
    got = S.evaluate([m.logp_expr()], x[None, :], m._data)[0][0]
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_generated_source_layout_and_errors():
    m = zoo.logistic()
    src, gen = m.generate()
    assert "nphip_density_stage" in src and "nphip_chain_sum" in src and "nphip_chain_barrier" in src
    # two groupings of the observations -> two arrays of adjoints stored grouped by target, and the non-centred effect as a gather source
    kinds = sorted(how for how, _ in gen.stored.values())
    assert kinds.count("grouped") == 2 and kinds.count("plain") >= 1
    with pytest.raises(ValueError, match="different dimensions"):
        _ = m._params[4] + m._params[5]
    with pytest.raises(ValueError, match="scalar"):
        m.add_logp(m._params[4])
    bad = S.Model()
    bad.dim("g", 2)
    with pytest.raises(ValueError, match="must lie in"):
        bad.index("i", [0, 5], dim="obs", into="g")


def test_compiles_for_gfx950_and_swaps_data_without_recompiling(tmp_path, monkeypatch):
    monkeypatch.setenv("NUTPIE_AMD_CACHE", str(tmp_path))
    from nutpie_amd.density import compile_density, data_layout

    m = zoo.poisson_offsets()
    compiled = m.compile(specialize=False)     # one library for data of any length
    path = compile_density(compiled._source, data_layout(compiled._data), compiled.n_dim)
    assert os.path.exists(path)
    rng = np.random.default_rng(0)
    # new observations of a different length: same layout, same source -> same library
    n2 = 123
    site = rng.integers(0, 40, n2)
    y2 = rng.poisson(2.0, n2).astype(np.float64)
    from scipy.special import gammaln

    swapped = compiled.with_data(y=y2, log_fact=gammaln(y2 + 1.0), site_idx=site, prior_scale=3.0)
    assert compile_density(swapped._source, data_layout(swapped._data), swapped.n_dim) == path
    # the default: the lengths of the data are constants of the source — new values of the same length keep the library, another
    # length is another source (compiled when it is first used)
    special = zoo.poisson_offsets().compile()
    n1 = len(special._data["y"])
    assert f"const int n_obs = {n1};" in special._source and "data.n_y" not in special._source
    same = special.with_data(y=special._data["y"] + 1.0, log_fact=gammaln(special._data["y"] + 2.0))
    assert same._source == special._source
    other = special.with_data(y=y2, log_fact=gammaln(y2 + 1.0), site_idx=site, prior_scale=3.0)
    assert f"const int n_obs = {n2};" in other._source and len(other._data["site_idx__pos"]) == n2
    assert np.allclose(other.logp_and_grad_numpy(0.1 * np.ones((1, other.n_dim)))[0], swapped.logp_and_grad_numpy(0.1 * np.ones((1, other.n_dim)))[0])
    assert len(swapped._data["site_idx__pos"]) == n2 and swapped._data["site_idx__rows"][-1] == n2
    lds, shared = swapped._lds()
    lds0, shared0 = compiled._lds()
    assert shared < shared0 and lds < lds0
    with pytest.raises(ValueError, match="share one length"):
        compiled.with_data(y=y2)
    with pytest.raises(ValueError, match="Unknown data variable"):
        compiled.with_data(nope=1.0)
    # the host restatement follows the data
    x = 0.1 * rng.normal(size=(2, compiled.n_dim))
    a, _ = compiled.logp_and_grad_numpy(x)
    b, _ = swapped.logp_and_grad_numpy(x)
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b)) and not np.allclose(a, b)
    # expanded variables: the constrained values of the parameters
    out = compiled._expand_draws(x.reshape(1, 2, -1))
    np.testing.assert_allclose(out["tau"][0], np.exp(x[:, 1]))
    assert out["u"].shape == (1, 2, 40)


def test_design_matrix_lowering():
    """``X @ v``: one data column and one extracted element per term; the gradient with respect to a computed ``v`` comes back as
    one vector written by the scalar code (stack), the one with respect to a parameter vector likewise"""
    m = zoo.regression()
    src, gen = m.generate()
    kinds = [how for how, _ in gen.stored.values()]
    assert kinds.count("scalars") == 1 and "D_X[j_0 * 6 + 5]" in src
    compiled = m.compile()
    X2 = np.random.default_rng(0).normal(size=(77, 6))
    swapped = compiled.with_data(X=X2, y=np.zeros(77), group_idx=np.zeros(77, dtype=int))
    assert swapped._lds()[1] < compiled._lds()[1] and len(swapped._data["X"]) == 77 * 6
    with pytest.raises(ValueError, match="keep its 6 columns"):
        compiled.with_data(X=X2[:, :3], y=np.zeros(77), group_idx=np.zeros(77, dtype=int))
    with pytest.raises(ValueError, match="share one length"):
        compiled.with_data(X=X2)
    with pytest.raises(ValueError, match="multiplies a vector on dimension"):
        _ = S.Matrix("X", m._dims["obs"], m._dims["coef"]) @ m._params[3]


def test_compile_pymc_model_accepts_the_front_end(tmp_path, monkeypatch):
    import nutpie_amd

    m = zoo.scalar_only()
    compiled = nutpie_amd.compile_pymc_model(m)
    assert compiled.n_dim == 2 and [k for k in compiled.shapes if k not in compiled.reparameterized_names] == ["a", "b"]
    with pytest.raises((ImportError, NotImplementedError)):
        nutpie_amd.compile_pymc_model(object())


def test_waves_per_chain_follow_the_lds_the_model_needs():
    """one wave per chain = four chains per workgroup = a quarter of the LDS each: a model with more observations gets one chain per
    workgroup (two waves) by itself, and one that cannot fit even so keeps its large arrays in device memory"""
    from nutpie_amd.radon import synthetic_radon_data

    assert zoo.radon().compile()._waves == 1
    big = zoo.radon(synthetic_radon_data(n_obs=3000)).compile()
    assert big._waves == 2 and big._lds()[1] == 0            # (and its data are read through L2: too much to stage)
    assert zoo.radon(synthetic_radon_data(n_obs=3000)).compile(waves_per_chain=4)._waves == 4
    # nothing fits: four chains per workgroup again, the two observation-sized arrays of adjoints in device memory (data.scratch__)
    huge = zoo.radon(synthetic_radon_data(n_obs=40000)).compile(specialize=False)
    assert huge._waves == 1 and huge._scratch(huge._data) == 80000 and "NPHIP_CHAIN_SLOT" in huge._source
    assert huge._lds()[0] < 4096 and huge.with_data(y=np.zeros(10), floor=np.zeros(10), county=np.zeros(10, dtype=int))._scratch(
        {"y": np.zeros(10)}) == 20
    # the default (a source specialised to the lengths of its data): data of another length are planned — and compiled — afresh
    huge = zoo.radon(synthetic_radon_data(n_obs=40000)).compile()
    assert huge._waves == 1 and huge._scratch(huge._data) == 80000 and "const int n_obs = 40000;" in huge._source
    small = huge.with_data(y=np.zeros(10), floor=np.zeros(10), county=np.zeros(10, dtype=int))
    assert small._waves == 1 and not small._scratch and "NPHIP_CHAIN_SLOT" not in small._source and "const int n_obs = 10;" in small._source


@pytest.mark.parametrize("name,waves", [("nested", 2), ("regression", 4)])
def test_multi_wave_sources_cross_compile(name, waves):
    """the generated source with several waves per chain (chain-wide sums and barriers) builds for gfx950"""
    from nutpie_amd.density import compile_density, data_layout

    m = zoo.ALL[name]().compile(waves_per_chain=waves)
    assert m._waves == waves and "nphip_chain_barrier" in m._source
    assert os.path.exists(compile_density(m._source, data_layout(m._data), m.n_dim, waves=waves))


def test_simplex_transform_and_its_jacobian():
    """`param(simplex=True)` (PyMC's SimplexTransform, the default of pm.Dirichlet: tests/test_pymc.py:312): positive, sums to one,
    and the log-Jacobian the model adds equals log |det d value[:n-1] / d raw| (finite differences)."""
    m = S.Model()
    m.dim("bar", 4)
    d = m.param("d", dim="bar", simplex=True)
    m.add_logp(S.flat_lpdf(d))          # nothing but the transform's Jacobian
    cm = m.compile()
    assert cm.n_dim == 3 and cm.shapes == {"d": (4,), "d_simplex__": (3,)}
    rng = np.random.default_rng(1)
    for x in rng.normal(size=(4, 3)) * np.array([[0.3], [1.0], [2.0], [4.0]]):
        value = lambda v: cm._expand_func(v[None], **cm._data)["d"][0]  # noqa: E731
        p = value(x)
        assert np.all(p > 0) and abs(p.sum() - 1.0) < 1e-14
        J = np.stack([(value(x + h)[:3] - value(x - h)[:3]) / 2e-6 for h in 1e-6 * np.eye(3)], axis=1)
        lp, _ = cm.logp_and_grad_numpy(x[None])
        np.testing.assert_allclose(lp[0], np.log(abs(np.linalg.det(J))), rtol=1e-6)


def test_reference_test_models_shapes_and_constraints():
    """The model shapes of the reference's own front-end tests, written with the front-end: what `trace.posterior` must look like
    (tests/test_pymc.py:303-349: dims of c and d; :618-640: (a, b) / transposed (b, a), sums to 0 and to 1; :352-380; :210-222)."""
    rng = np.random.default_rng(3)
    se = zoo.store_extra().compile()
    assert se.n_dim == 5 + 5 + 4 + 3
    assert se.shapes == {"a": (5,), "b": (5,), "c": (5,), "d": (4,), "b_log__": (5,), "c_zerosum__": (4,), "d_simplex__": (3,)}
    assert se.dims["c"] == ("foo",) and se.dims["d"] == ("bar",)
    # the unconstrained values (the trace's unconstrained_posterior group): b_log__ keeps the dimension, c_zerosum__ and d_simplex__
    # lose an element and with it the dimension (tests/test_pymc.py:332-346)
    assert se.reparameterized_names == ["b_log__", "c_zerosum__", "d_simplex__"]
    assert se.dims["b_log__"] == ("foo",) and se.dims["c_zerosum__"] != ("foo",) and se.dims["d_simplex__"] != ("bar",)
    xs = rng.normal(size=(6, se.n_dim))
    ex = se._expand_func(xs, **se._data)
    assert np.array_equal(ex["b_log__"], xs[:, 5:10]) and np.array_equal(ex["c_zerosum__"], xs[:, 10:14]) and np.array_equal(ex["d_simplex__"], xs[:, 14:17])
    assert np.all(ex["b"] > 0) and np.abs(ex["c"].sum(-1)).max() < 1e-14 and np.abs(ex["d"].sum(-1) - 1).max() < 1e-14 and np.all(ex["d"] > 0)
    dm = zoo.dims_model().compile()
    assert dm.n_dim == 2 * 5 and dm.shapes == {"zero_sum": (3, 5), "one_sum": (5, 3), "col_sum": (5,), "zero_sum_zerosum__": (2, 5)}
    assert {k: dm.dims[k] for k in ("zero_sum", "one_sum", "col_sum")} == {"zero_sum": ("a", "b"), "one_sum": ("b", "a"), "col_sum": ("b",)}
    assert set(dm.coords) == {"a", "b"}
    x = rng.normal(size=(4, dm.n_dim))
    ex = dm._expand_func(x, **dm._data)
    np.testing.assert_allclose(ex["zero_sum"].sum(1), 0, atol=1e-14)          # along a, for every b
    np.testing.assert_allclose(ex["one_sum"].sum(2), 1, atol=1e-14)
    np.testing.assert_array_equal(ex["one_sum"], ex["zero_sum"].transpose(0, 2, 1) + 1.0 / 3.0)
    lp, _ = dm.logp_and_grad_numpy(x)
    np.testing.assert_allclose(lp, -0.5 * (x * x).sum(1), rtol=1e-13)         # the zero-sum extension is an isometry
    ud = zoo.uniform_det().compile()
    ex = ud._expand_func(3.0 * rng.normal(size=(50, 2)), **ud._data)
    assert ex["a"].shape == (50, 2) and ex["a"].min() > 0 and ex["a"].max() < 1 and np.array_equal(ex["b"], 2.0 * ex["a"])
    npm = zoo.no_prior().compile()
    lp, g = npm.logp_and_grad_numpy(np.array([[0.7]]))
    np.testing.assert_allclose(g[0, 0], -0.7)                                   # a flat prior adds nothing


def test_expand_step_is_generated_too():
    """`deterministic` values and the constrained parameters are a generated device function (`nphip_expand`) next to the density,
    exported behind the C-ABI's device-expand signature; values on a data dimension stay with the host evaluation."""
    for name in ("store_extra", "dims_model", "eight_schools", "radon"):
        cm = zoo.ALL[name]().compile()
        assert "__device__ double nphip_expand(const NphipData& data" in cm._source and cm._source.count("nphip_density_stage(") <= 1
    assert "((i_0 % 5) * 3 + i_0 / 5)" in zoo.dims_model().compile()._source       # one_sum is stored transposed: (a, b) -> (b, a)
    from nutpie_amd.density import data_layout, generated_source

    cm = zoo.store_extra().compile()
    assert "#define NPHIP_JIT_EXPAND 1" in generated_source(cm._source, data_layout(cm._data))
    m = S.Model()
    mu = m.param("mu")
    y = m.data("y", np.arange(5.0), dim="obs")
    m.add_logp(S.normal_lpdf(y, mu, 1.0).sum())
    m.deterministic("resid", y - mu)                    # lives on a data dimension: its length changes under with_data
    assert "nphip_expand" not in m.compile()._source


def test_with_data_rederives_shapes_and_checks_the_plan():
    """`with_data` may change the length of a data dimension (only the rank is fixed, as in the reference:
    compile_pymc.py:140-166).  Shapes of reported values on that dimension follow the new data, and data that no longer fit the LDS
    plan of compile() are refused at `with_data` — not after the run, in a reshape."""
    m = S.Model()
    mu = m.param("mu")
    y = m.data("y", np.arange(5.0), dim="obs")
    m.add_logp(S.normal_lpdf(y, mu, 1.0).sum())
    m.deterministic("resid", y - mu)
    cm = m.compile()
    assert cm.shapes == {"mu": (), "resid": (5,)}
    cm2 = cm.with_data(y=np.arange(9.0))
    assert cm2.shapes == {"mu": (), "resid": (9,)} and cm.shapes["resid"] == (5,)
    ex = cm2._expand_draws(np.zeros((2, 3, 1)))
    assert ex["resid"].shape == (2, 3, 9) and np.array_equal(ex["resid"][0, 0], np.arange(9.0))
    # one library for data of any length (specialize=False): data that outgrow its LDS plan are refused here ...
    with pytest.raises(ValueError, match="do not fit the LDS plan"):
        m.compile(specialize=False).with_data(y=np.zeros(200_000))
    # ... the default compiles for the new length, with a plan of its own (here: the observation-sized array in device memory)
    big = cm.with_data(y=np.zeros(200_000))
    assert big.shapes["resid"] == (200_000,) and "const int n_obs = 200000;" in big._source


def test_expression_table_does_not_outlive_its_models():
    import gc

    m = zoo.eight_schools()
    _ = m.logp_expr()
    assert len(S.Expr._table) > 50
    before = len(S.Expr._table)
    del m, _
    gc.collect()
    assert len(S.Expr._table) < before


def test_initial_points_follow_pymc_support_point_plus_jitter():
    """reference compile_pymc.py:593-602: ``make_initial_point_fn(default_strategy="support_point", jitter_rvs=set(model.free_RVs))`` —
    the support point on the unconstrained scale plus U(-1, 1); ``initial_points`` overrides per variable; the transforms' forward
    directions are PyMC's (log, interval, ZeroSumTransform, SimplexTransform)."""
    import nutpie_amd
    from nutpie_amd import symbolic as S
    from nutpie_amd.density import JitteredInit

    m = S.Model()
    a = m.param("a", lower=0.0, initval=2.0)
    b = m.param("b", dim="k", size=4, simplex=True, initval=[0.1, 0.2, 0.3, 0.4])
    c = m.param("c", dim="j", size=3, zero_sum=True, initval=[1.0, -3.0, 2.0])
    d = m.param("d", lower=-1.0, upper=3.0)
    e = m.param("e", dims=("j", "k"), zero_sum=True)
    m.add_logp(-a + S.log(b).sum() - (c * c).sum() - d * d - (e * e).sum())
    x0 = m.initial_point()
    assert x0.shape == (m.n_dim,) == (1 + 3 + 2 + 1 + 8,)
    cm = m.compile(init="support_point")
    assert isinstance(cm._init, JitteredInit)
    back = cm._expand_draws(x0[None, None, :])                      # constrain(unconstrain(initval)) == initval
    np.testing.assert_allclose(back["a"], [[2.0]], rtol=1e-14)
    np.testing.assert_allclose(back["b"][0, 0], [0.1, 0.2, 0.3, 0.4], rtol=1e-13)
    np.testing.assert_allclose(back["c"][0, 0], [1.0, -3.0, 2.0], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(back["d"], [[1.0]], atol=1e-14)      # no initval: unconstrained 0 = the middle of the interval
    assert np.all(back["e"] == 0)
    pts = cm._init.points(seed=7, n_chains=64)
    assert pts.shape == (64, m.n_dim) and np.all(np.abs(pts - x0) <= 1.0) and np.abs(pts - x0).max() > 0.9
    np.testing.assert_array_equal(pts[:8], cm._init.points(seed=7, n_chains=8))         # a chain's point does not depend on the number of chains
    assert not np.array_equal(pts, cm._init.points(seed=8, n_chains=64))
    # compile_pymc_model: the reference's keywords
    cm2 = nutpie_amd.compile_pymc_model(m, initial_points={"d": 2.5}, jitter_rvs={"a", "d"})
    _, off_d, _ = m._unconstrained["d"]
    assert abs(cm2._init.center[off_d] - np.log(3.5 / 0.5)) < 1e-14
    assert cm2._init.jitter.sum() == 2.0 and cm2._init.jitter[0] == 1.0 and cm2._init.jitter[off_d] == 1.0
    with pytest.raises(ValueError, match="inside its support"):
        m.initial_point({"a": -1.0})
    with pytest.raises(ValueError, match="not supported"):
        nutpie_amd.compile_pymc_model(m, default_initialization_strategy="prior")


def test_densities_of_the_front_end_against_scipy():
    """every ``*_lpdf`` / ``*_lpmf`` of the front-end on a model of its own: the value against ``scipy.stats`` at random points (the
    parameters of the density are model parameters where the function takes expressions: their gradient is the symbolic one, checked
    by central differences)"""
    from scipy import stats
    from scipy.special import gammaln

    rng = np.random.default_rng(5)
    yv = rng.normal(size=7)
    pos = np.abs(rng.normal(size=7)) + 0.2
    unit = rng.uniform(0.05, 0.95, size=7)
    cnt = rng.poisson(4.0, 7).astype(np.float64)
    ntr = cnt + rng.integers(0, 5, 7)

    def build(fn):
        m = S.Model()
        a = m.param("a", lower=0.0)
        b = m.param("b", lower=0.0)
        c = m.param("c")
        d = {k: m.data(k, v, dim="obs") for k, v in dict(y=yv, pos=pos, unit=unit, cnt=cnt, ntr=ntr, lf=gammaln(cnt + 1.0),
                                                           lb=gammaln(ntr + 1.0) - gammaln(cnt + 1.0) - gammaln(ntr - cnt + 1.0)).items()}
        m.add_logp(fn(a, b, c, d).sum())
        return m.compile()

    cases = {
        "student_t (nu a parameter)": (lambda a, b, c, d: S.student_t_lpdf(d["y"], a + 1.0, c, b), lambda a, b, c: stats.t.logpdf(yv, a + 1.0, c, b)),
        "student_t (nu a number)": (lambda a, b, c, d: S.student_t_lpdf(d["y"], 3.5, c, b), lambda a, b, c: stats.t.logpdf(yv, 3.5, c, b)),
        "gamma (shape a parameter)": (lambda a, b, c, d: S.gamma_lpdf(d["pos"], a, b), lambda a, b, c: stats.gamma.logpdf(pos, a, scale=1.0 / b)),
        "inverse_gamma": (lambda a, b, c, d: S.inverse_gamma_lpdf(d["pos"], a, b), lambda a, b, c: stats.invgamma.logpdf(pos, a, scale=b)),
        "inverse_gamma (number)": (lambda a, b, c, d: S.inverse_gamma_lpdf(d["pos"], 2.5, b), lambda a, b, c: stats.invgamma.logpdf(pos, 2.5, scale=b)),
        "beta": (lambda a, b, c, d: S.beta_lpdf(d["unit"], a, b), lambda a, b, c: stats.beta.logpdf(unit, a, b)),
        "beta (numbers)": (lambda a, b, c, d: S.beta_lpdf(d["unit"], 2.0, 3.5) + 0.0 * c, lambda a, b, c: stats.beta.logpdf(unit, 2.0, 3.5)),
        "laplace": (lambda a, b, c, d: S.laplace_lpdf(d["y"], c, b), lambda a, b, c: stats.laplace.logpdf(yv, c, b)),
        "logistic": (lambda a, b, c, d: S.logistic_lpdf(d["y"], c, b), lambda a, b, c: stats.logistic.logpdf(yv, c, b)),
        "weibull": (lambda a, b, c, d: S.weibull_lpdf(d["pos"], a, b), lambda a, b, c: stats.weibull_min.logpdf(pos, a, scale=b)),
        "binomial_logit": (lambda a, b, c, d: S.binomial_logit_lpmf(d["cnt"], d["ntr"], c, d["lb"]), lambda a, b, c: stats.binom.logpmf(cnt, ntr, 1.0 / (1.0 + np.exp(-c)))),
        "negative_binomial_log": (lambda a, b, c, d: S.negative_binomial_log_lpmf(d["cnt"], c, a, d["lf"]),
                                  lambda a, b, c: stats.nbinom.logpmf(cnt, a, a / (a + np.exp(c)))),
    }
    for name, (fn, ref) in cases.items():
        cm = build(fn)
        x = rng.normal(size=(5, 3)) * 0.5
        lp, g = cm.logp_and_grad_numpy(x)
        for i, row in enumerate(x):
            a, b, c = np.exp(row[0]), np.exp(row[1]), row[2]
            want = ref(a, b, c).sum() + row[0] + row[1]          # (+ the log-Jacobians of the two positive parameters)
            assert abs(lp[i] - want) < 1e-10 * max(1.0, abs(want)), (name, lp[i], want)
        h = 1e-6
        for j in range(3):
            e = np.zeros(3)
            e[j] = h
            fd = (cm.logp_and_grad_numpy(x + e)[0] - cm.logp_and_grad_numpy(x - e)[0]) / (2 * h)
            np.testing.assert_allclose(g[:, j], fd, rtol=2e-6, atol=2e-6, err_msg=name)
    assert S.uniform_lpdf(0.3, -1.0, 3.0).is_const(-np.log(4.0))
    with pytest.raises(ValueError, match="phi is a parameter"):
        S.negative_binomial_log_lpmf(1.0, 0.0, 2.0, 0.0)


def test_ordered_transform_and_an_ordinal_regression():
    """`param(ordered=True)` (PyMC's ``ordered`` transform, the cut points of ``pm.OrderedLogistic``): increasing values, the log-Jacobian
    ``sum(raw[1:])``, the forward direction for initial values — and the generated source of an ordinal regression builds for gfx950"""
    from nutpie_amd.density import compile_density, data_layout

    rng = np.random.default_rng(9)
    n, K = 60, 5
    xcov = rng.normal(size=n)
    ycat = rng.integers(0, K, n)
    m = S.Model()
    cut = m.param("cut", dim="cutpoint", size=K - 1, ordered=True, initval=[-1.5, -0.5, 0.5, 1.5])
    beta = m.param("beta")
    xd = m.data("x", xcov, dim="obs")
    # P(y <= k) = sigmoid(cut_k - eta); the two cut points around each observation's category (padded with -inf / +inf as +-30)
    lo_idx = m.index("lo_idx", np.clip(ycat - 1, 0, K - 2), dim="obs", into="cutpoint")
    hi_idx = m.index("hi_idx", np.clip(ycat, 0, K - 2), dim="obs", into="cutpoint")
    is_first = m.data("is_first", (ycat == 0).astype(np.float64), dim="obs")
    is_last = m.data("is_last", (ycat == K - 1).astype(np.float64), dim="obs")
    eta = beta * xd
    p_hi = is_last + (1.0 - is_last) * S.sigmoid(cut[hi_idx] - eta)
    p_lo = (1.0 - is_first) * S.sigmoid(cut[lo_idx] - eta)
    m.add_logp(S.log(p_hi - p_lo).sum() + S.normal_lpdf(cut, 0.0, 3.0).sum() + S.normal_lpdf(beta, 0.0, 2.0))
    cm = m.compile()
    assert cm.n_dim == K and cm.shapes["cut"] == (K - 1,) and cm.shapes["cut_ordered__"] == (K - 1,)
    np.testing.assert_allclose(m.initial_point()[:K - 1], [-1.5, 0.0, 0.0, 0.0], atol=1e-12)
    x = 0.4 * rng.normal(size=(4, K))
    lp, g = cm.logp_and_grad_numpy(x)
    vals = cm._expand_func(x, **cm._data)["cut"]
    assert np.all(np.diff(vals, axis=1) > 0)

    def ref(row):
        raw, b = row[:K - 1], row[K - 1]
        c = np.concatenate([raw[:1], raw[0] + np.cumsum(np.exp(raw[1:]))])
        cdf = 1.0 / (1.0 + np.exp(-(np.concatenate([[-np.inf], c, [np.inf]])[None, :] - (b * xcov)[:, None])))
        like = np.log(cdf[np.arange(n), ycat + 1] - cdf[np.arange(n), ycat]).sum()
        prior = (-0.5 * (c / 3.0) ** 2 - np.log(3.0) - 0.5 * np.log(2 * np.pi)).sum() - 0.5 * (b / 2.0) ** 2 - np.log(2.0) - 0.5 * np.log(2 * np.pi)
        return like + prior + raw[1:].sum()

    with np.errstate(over="ignore"):
        np.testing.assert_allclose(lp, [ref(r) for r in x], rtol=1e-12)
    h = 1e-6
    for j in range(K):
        e = np.zeros(K)
        e[j] = h
        fd = (cm.logp_and_grad_numpy(x + e)[0] - cm.logp_and_grad_numpy(x - e)[0]) / (2 * h)
        np.testing.assert_allclose(g[:, j], fd, rtol=1e-6, atol=1e-6)
    path = compile_density(cm._source, data_layout(cm._data), cm.n_dim)
    assert os.path.exists(path)
    os.remove(path)
    with pytest.raises(ValueError, match="ordered excludes"):
        S.Model().param("c", dim="k", size=3, ordered=True, lower=0.0)


def test_var_names_filter_the_computed_variables_as_the_reference_does():
    """tests/test_pymc.py:425-468 of the reference: var_names=None stores the deterministics b and c, [] neither, ["b"] only b — the free variables
    always (a transformed parameter's free variable is its unconstrained value)."""
    import nutpie_amd

    def make():
        m = S.Model()
        a = m.param("a", dim="k", size=3)
        sd = m.param("sd", lower=0.0)
        m.add_logp(S.normal_lpdf(a, -0.1, 1.0).sum() + S.normal_lpdf(sd, 0.0, 1.0))
        b = -0.1 * a
        m.deterministic("b", b)
        m.deterministic("c", -0.1 * b)
        return m

    names = lambda **kw: list(nutpie_amd.compile_pymc_model(make(), **kw).shapes)  # noqa: E731
    assert names(var_names=None) == ["a", "sd", "b", "c", "sd_log__"]
    assert names(var_names=[]) == ["a", "sd_log__"]
    assert names(var_names=["b"]) == ["a", "b", "sd_log__"]
    assert names(var_names=["sd", "c"]) == ["a", "sd", "c", "sd_log__"]
    with pytest.raises(KeyError, match="zz"):
        names(var_names=["zz"])
    with pytest.raises(TypeError, match="bogus"):
        nutpie_amd.compile_pymc_model(make(), bogus=1)
    # freeze_model (compile_pymc.py:587-592) = whether the data's lengths are constants of the generated source
    assert "sd_log__" in names(freeze_model=False) and "sd_log__" in names(freeze_model=True)
    # the filtered expand step evaluates
    cm = nutpie_amd.compile_pymc_model(make(), var_names=["b"])
    out = cm._expand_func(np.zeros((2, 4)))
    assert set(out) == {"a", "b", "sd_log__"} and out["b"].shape == (2, 3)
