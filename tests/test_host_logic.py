"""Host-side logic that runs without a GPU: trace assembly, ESS, sample() argument handling, Gaussian targets."""
import os
import warnings

import numpy as np
import pytest


def test_trace_warmup_split_and_nan_padding():
    # reference layout rules: python/nutpie/sample.py:62-214 (split by `tuning`, NaN-pad unequal chains)
    from nutpie_amd.trace import build_trace

    n, T, D = 3, 10, 2
    draws = np.arange(n * T * D, dtype=float).reshape(n, T, D)
    tuning = np.zeros((n, T), bool)
    tuning[:, :4] = True
    stats = {"tuning": tuning, "depth": np.ones((n, T), np.int64), "energy": np.full((n, T), 2.5),
             "diverging": np.zeros((n, T), bool), "gradient": draws * 2}
    finished = np.array([10, 7, 3])  # chains 1, 2 were aborted early (chain 2 inside warm-up)
    tr = build_trace({"x": draws}, stats, finished, save_warmup=True, skip_vars=["gradient"], use_arviz=False)
    assert set(tr.groups()) == {"posterior", "sample_stats", "warmup_posterior", "warmup_sample_stats"}
    assert tr.posterior.x.shape == (3, 6, 2) and tr.warmup_posterior.x.shape == (3, 4, 2)
    assert np.array_equal(tr.posterior.x.values[0], draws[0, 4:])
    assert np.array_equal(tr.posterior.x.values[1, :3], draws[1, 4:7]) and np.all(np.isnan(tr.posterior.x.values[1, 3:]))
    assert np.all(np.isnan(tr.posterior.x.values[2]))
    assert np.array_equal(tr.warmup_posterior.x.values[2, :3], draws[2, :3]) and np.all(np.isnan(tr.warmup_posterior.x.values[2, 3:]))
    assert "gradient" not in tr.sample_stats and tr.sample_stats.depth.dtype == np.int64
    assert tr.sample_stats.depth.values[1, 3:].sum() == 0          # integer columns are zero padded
    assert tr.posterior.x.dims == ("chain", "draw", "x_dim_0")
    tr2 = build_trace({"x": draws}, stats, finished, save_warmup=False, use_arviz=False)
    assert set(tr2.groups()) == {"posterior", "sample_stats"} and "gradient" in tr2.sample_stats


def test_trace_unconstrained_groups():
    from nutpie_amd.trace import build_trace

    n, T = 2, 6
    tuning = np.zeros((n, T), bool); tuning[:, :2] = True
    ex = {"a": np.ones((n, T)), "a_log__": np.zeros((n, T))}
    tr = build_trace(ex, {"tuning": tuning}, [T, T], reparameterized_names=["a_log__"], keep_unconstrained_draw=True, use_arviz=False)
    assert "a_log__" not in tr.posterior and "a_log__" in tr.unconstrained_posterior and "a_log__" in tr.warmup_unconstrained_posterior


def test_ess_iid_and_ar1():
    from nutpie_amd.ess import ess_bulk, ess_bulk_min

    rng = np.random.default_rng(0)
    x = rng.normal(size=(4, 1000))
    assert 3200 < ess_bulk(x) < 4800                      # iid: ESS ~ N
    phi = 0.9                                             # AR(1): tau = (1+phi)/(1-phi) = 19
    y = np.zeros((4, 4000))
    e = rng.normal(size=y.shape)
    for t in range(1, y.shape[1]):
        y[:, t] = phi * y[:, t - 1] + e[:, t]
    ess = ess_bulk(y)
    assert 16000 / 19 * 0.6 < ess < 16000 / 19 * 1.6
    # a chain stuck elsewhere collapses the ESS (between-chain variance enters var_plus)
    z = x.copy(); z[0] += 5
    assert ess_bulk(z) < 50
    m, vals = ess_bulk_min(np.stack([x, y[:, :1000]], -1))
    assert m == vals.min() and vals[0] > vals[1]


def test_sample_argument_handling():
    import nutpie_amd

    m = nutpie_amd.std_normal(3)
    with pytest.raises(ValueError, match="Unknown adaptation strategy 'foo'"):     # sample.py:1026-1030
        nutpie_amd.sample(m, adaptation="foo")
    with pytest.raises(ValueError, match="Unknown sampler 'hmc'"):                 # sample.py:1044-1047
        nutpie_amd.sample(m, sampler="hmc")
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="needs a GPU"):                          # every model flavour takes the low-rank metric — on a GPU
            nutpie_amd.sample(m, adaptation="low_rank")
    with pytest.raises(NotImplementedError):
        nutpie_amd.sample(m, adaptation="flow")
    with pytest.raises(NotImplementedError):
        nutpie_amd.sample(m, sampler="mclmc")
    with pytest.raises(AttributeError, match="Unknown settings attribute: bogus"):
        nutpie_amd.sample(m, bogus=1)
    with pytest.raises(ValueError, match="not available for diag adaptation"):
        nutpie_amd.sample(m, mass_matrix_gamma=1e-5)
    with warnings.catch_warnings(record=True) as w:                                # deprecated aliases, sample.py:979-1013
        warnings.simplefilter("always")
        if not torch.cuda.is_available():
            with pytest.raises(RuntimeError, match="needs a GPU"):
                nutpie_amd.sample(m, low_rank_modified_mass_matrix=True)
        else:
            nutpie_amd.sample(m, low_rank_modified_mass_matrix=True, chains=2, tune=50, draws=10, progress_bar=False)
        assert any(issubclass(x.category, FutureWarning) for x in w)
    with pytest.raises(ValueError, match="cannot be combined"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            nutpie_amd.sample(m, transform_adapt=True, adaptation="draw_diag")
    with pytest.raises(NotImplementedError):
        nutpie_amd.sample(m, zarr_store=object(), chains=1)


def test_front_end_shells_raise_import_errors():
    import nutpie_amd

    with pytest.raises(ImportError, match="pymc"):
        nutpie_amd.compile_pymc_model(object())
    with pytest.raises(ImportError, match="BridgeStan"):
        nutpie_amd.compile_stan_model(code="parameters { real a; } model { a ~ normal(0,1); }")


def test_from_pyfunc_contract():
    import nutpie_amd

    def make_logp():
        return lambda x, scale: (-0.5 * float(x @ x) / scale, -x / scale)

    def make_expand(seed1, seed2, chain):
        return lambda x, scale: {"y": x, "ysum": np.array(x.sum())}

    m = nutpie_amd.from_pyfunc(3, make_logp, make_expand, [np.float64, np.float64], [(3,), ()], ["y", "ysum"], shared_data={"scale": 2.0})
    assert m.n_dim == 3 and m.shapes == {"y": (3,), "ysum": ()}
    with pytest.raises(ValueError, match="Unknown data variable"):                   # compiled_pyfunc.py:39-46
        m.with_data(nope=1)
    m2 = m.with_data(scale=4.0)
    assert m2.data["scale"] == 4.0 and m.data["scale"] == 2.0
    ex = m._expand_draws(np.arange(12.0).reshape(2, 2, 3))
    assert ex["y"].shape == (2, 2, 3) and ex["ysum"].shape == (2, 2) and ex["ysum"][1, 1] == 9 + 10 + 11
    bad = nutpie_amd.from_pyfunc(3, make_logp, lambda *a: (lambda x: {"y": x.astype(np.float32)}), [np.float64], [(3,)], ["y"])
    with pytest.raises(TypeError, match="dtype"):
        bad._expand_draws(np.zeros((1, 1, 3)))


def test_gaussian_targets():
    from nutpie_amd.gaussian import ar1_gaussian, diag_gaussian, std_normal

    assert np.array_equal(std_normal(5).diag, np.ones(5))
    np.testing.assert_allclose(diag_gaussian([2.0, 0.5]).covariance(), np.diag([4.0, 0.25]))
    m = ar1_gaussian(6, rho=0.9, scales=np.array([1, 2, 3, 1, 2, 3.0]))
    cov = m.covariance()
    s = np.array([1, 2, 3, 1, 2, 3.0])
    want = 0.9 ** np.abs(np.subtract.outer(np.arange(6), np.arange(6))) * np.outer(s, s)
    np.testing.assert_allclose(cov, want, rtol=1e-10)
    # default scales come from numpy.random.default_rng(20260926)
    a, b = ar1_gaussian(1000), ar1_gaussian(1000)
    assert np.array_equal(a.diag, b.diag) and a.n_dim == 1000


def test_shard_chains():
    from nutpie_amd.distributed import shard_chains

    for n, w in [(1024, 8), (10, 4), (3, 8), (8192, 8), (7, 1)]:
        parts = [shard_chains(n, w, r) for r in range(w)]
        assert sum(p[1] for p in parts) == n
        assert parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(p[1] for p in parts) - min(p[1] for p in parts) <= 1


def test_radon_hand_derived_gradient_matches_autograd():
    # config 3's log-density (nutpie_amd/radon.py) carries a hand-derived gradient (GEMM gather / scatter-add, ~35
    # kernels); autograd of the same density is the reference.  Runs on the CPU: the formulas, not the device.
    import torch

    from nutpie_amd.radon import radon_model

    m = radon_model(device="cpu")
    f = m._make_logp_func()
    x = torch.randn(5, m.n_dim, dtype=torch.float64) * 0.4
    lp, g = f(x)
    lp_ref, g_ref = f.autograd_reference(x)
    assert torch.allclose(lp, lp_ref, rtol=1e-12, atol=1e-10)
    assert torch.allclose(g, g_ref, rtol=1e-10, atol=1e-9)


def test_autograd_logp_wrapper():
    # from_torch_density: gradient of a batched log-density through torch.autograd, one row per chain
    import torch

    from nutpie_amd.compiled_pyfunc import autograd_logp, from_torch_density

    scale = torch.tensor([1.0, 2.0, 0.5], dtype=torch.float64)
    f = autograd_logp(lambda x: -0.5 * ((x / scale) ** 2).sum(-1))
    x = torch.randn(7, 3, dtype=torch.float64)
    lp, g = f(x)
    assert torch.allclose(lp, -0.5 * ((x / scale) ** 2).sum(-1)) and torch.allclose(g, -x / scale**2)
    assert not lp.requires_grad and not g.requires_grad
    m = from_torch_density(3, lambda x, s: -0.5 * ((x / s) ** 2).sum(-1), shared_data={"s": scale}, compile=False)
    lp2, g2 = m._make_logp_func()(x, **m._shared_data)
    assert torch.equal(lp2, lp) and torch.equal(g2, g) and m.n_dim == 3
    # the default ("auto") traces the function and compiles it into the engine instead (tests/test_torch_trace_cpu.py)
    mc = from_torch_density(3, lambda x, s: -0.5 * ((x / s) ** 2).sum(-1), shared_data={"s": scale})
    assert type(mc).__name__ == "TracedTorchModel" and mc.n_dim == 3
    lp3, g3 = mc.logp_and_grad_numpy(x.numpy())
    np.testing.assert_allclose(lp3, lp.numpy(), rtol=1e-13)
    np.testing.assert_allclose(g3, g.numpy(), rtol=1e-13)


def test_arviz_conversion_follows_the_installed_version(monkeypatch):
    """arviz < 1.0 takes the groups as keywords, >= 1.0 as one dict (reference sample.py:122-147); a failing
    conversion must not lose the trace."""
    import importlib.metadata
    import sys
    import types

    from nutpie_amd import trace as T

    calls = {}
    stub = types.ModuleType("arviz")

    def from_dict(*args, **kw):
        calls["args"], calls["kw"] = args, kw

        class Out(dict):
            pass

        o = Out()
        o["sample_stats"] = types.SimpleNamespace(attrs={})
        return o

    stub.from_dict = from_dict
    monkeypatch.setitem(sys.modules, "arviz", stub)
    real_version = importlib.metadata.version
    stats = {"tuning": np.array([[True, False, False]]), "depth": np.array([[1, 2, 3]])}
    ex = {"x": np.arange(3.0).reshape(1, 3, 1)}
    for ver, as_kwargs in (("0.23.4", True), ("1.0.0", False), ("1.2", False)):
        monkeypatch.setattr(importlib.metadata, "version", lambda name, v=ver: v if name == "arviz" else real_version(name))
        out = T.build_trace(ex, stats, np.array([3]), use_arviz=True, attrs={"inference_library": "nutpie"})
        assert out["sample_stats"].attrs["inference_library"] == "nutpie"
        if as_kwargs:
            assert calls["args"] == () and {"posterior", "sample_stats", "dims", "coords"} <= set(calls["kw"])
        else:
            assert len(calls["args"]) == 1 and set(calls["args"][0]) >= {"posterior", "sample_stats"} and "posterior" not in calls["kw"]
    # a conversion that raises falls back to the built-in container (with a warning) instead of losing the trace
    stub.from_dict = lambda *a, **k: (_ for _ in ()).throw(TypeError("boom"))
    with pytest.warns(RuntimeWarning, match="conversion to an ArviZ object failed"):
        out = T.build_trace(ex, stats, np.array([3]), use_arviz=True)
    assert out.posterior.x.shape == (1, 2, 1)


def test_stan_model_shell_binds_once_and_expands_behind_the_c_abi(monkeypatch, tmp_path):
    """The Stan front-end shell around a stub ``bridgestan``: the model is bound once (not per property access), names are parsed
    into variables, and the expand step is NOT a Python loop any more — it runs behind the C-ABI
    (nphip_model_set_bridgestan_expand: bs_param_constrain per draw with one bs_rng per chain, src/stan.rs:473-520, 787-796;
    GPU tests: tests/test_gpu_reference_fixtures.py), so the shell only splits the flat C-order rows into variables."""
    import sys
    import types

    bs = types.ModuleType("bridgestan")
    made = {"models": 0}

    class StanModel:
        def __init__(self, lib, data=None, seed=0):
            made["models"] += 1

        def param_unc_num(self):
            return 2

        def param_names(self, include_tp=False, include_gq=False):
            return ["a", "b"] + (["t"] if include_tp else []) + (["g.1.1", "g.2.1", "g.1.2", "g.2.2", "g.1.3", "g.2.3"] if include_gq else [])

    bs.StanModel = StanModel
    monkeypatch.setitem(sys.modules, "bridgestan", bs)
    from nutpie_amd.compile_stan import CompiledStanModel
    from nutpie_amd.stan_names import c_order_permutation, expand_constrained

    m = CompiledStanModel(dims={}, code="", data=None, library="x.so", model=None, _coords={})
    assert m.n_dim == 2 and m.shapes == {"a": (), "b": (), "t": (), "g": (2, 3)} and m.n_dim == 2
    assert made["models"] == 1                     # bound once, not per property access
    # what the native adapter does to a row (out[j] = theta[perm[j]]) followed by the shell's split == the reference's
    # per-variable fortran_to_c_order (src/stan.rs:507-516, 671-711; restated vectorised in stan_names.expand_constrained)
    rng = np.random.default_rng(0)
    theta = rng.normal(size=(3, 4, 9))             # BridgeStan's flat rows: column-major blocks
    perm = c_order_permutation(m._variables())
    assert sorted(perm.tolist()) == list(range(9)) and perm[:3].tolist() == [0, 1, 2]
    got = m._unflatten(theta[..., perm])
    want = expand_constrained(theta, m._variables())
    assert got.keys() == want.keys() and all(np.array_equal(got[k], want[k]) for k in got)
    assert got["g"].shape == (3, 4, 2, 3) and got["g"][0, 0, 1, 2] == theta[0, 0, 3 + 1 + 2 * 2]
    with pytest.raises(RuntimeError, match="behind the C-ABI"):
        m._expand_draws(np.zeros((1, 1, 2)))


def test_bridgestan_stand_in_speaks_the_c_api(bs_standin):
    """The stand-in library the BridgeStan adapters are tested against (tests/fixtures/bs_standin.c) — no GPU involved:
    names parse to the declared shapes and its column-major output re-orders to what the Stan program means."""
    import ctypes as C

    from nutpie_amd.stan_names import c_order_permutation, parse_stan_variables

    h = C.c_void_p(bs_standin.bs_model_construct(b"matrix", 0, None))
    assert bs_standin.bs_param_unc_num(h) == 6 and bs_standin.bs_param_num(h, True, True) == 13
    variables = parse_stan_variables(bs_standin.bs_param_names(h, True, True).decode())
    assert [(v.name, v.shape) for v in variables] == [("m", (2, 3)), ("mt", (3, 2)), ("s", ())]
    bs_standin.bs_rng_construct.restype = C.c_void_p
    rng = C.c_void_p(bs_standin.bs_rng_construct(3, None))
    unc = (C.c_double * 6)(*range(1, 7))
    out = (C.c_double * 13)()
    assert bs_standin.bs_param_constrain(h, True, True, unc, out, rng, None) == 0
    row = np.array(out[:])[c_order_permutation(variables)]
    mm = row[:6].reshape(2, 3)
    assert mm.tolist() == [[1, 3, 5], [2, 4, 6]] and np.array_equal(row[6:12].reshape(3, 2), mm.T)
    assert bs_standin.bs_param_constrain(h, True, True, unc, out, None, None) == 1      # generated quantities need an rng
    bs_standin.bs_rng_destruct(rng)
    bs_standin.bs_model_destruct(h)


def test_install_as_nutpie_alias():
    """`import nutpie` can resolve to the HIP engine (opt-in): the reference's import lines work unchanged."""
    import os
    import subprocess
    import sys

    code = (
        "import nutpie_amd; nutpie_amd.install_as_nutpie();"
        "import nutpie; from nutpie import _lib; from nutpie.sample import sample, CompiledModel;"
        "from nutpie.compiled_pyfunc import from_pyfunc; from nutpie.compile_pymc import compile_pymc_model;"
        "assert nutpie.sample is sample and nutpie.ChainProgress is _lib.PyChainProgress and nutpie.__version__;"
        "print('alias ok')"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root)
    assert r.returncode == 0 and "alias ok" in r.stdout, r.stderr


@pytest.mark.parametrize("threads,use", [(1, 0), (2, 0), (5, 0), (5, 2), (8, 1), (3, 7)])
def test_evaluation_pool_runs_every_row_exactly_once(threads, use):
    """The host-callback evaluation pool (spin-waiting workers that fall asleep when idle, a per-batch thread count): every row
    of every batch is evaluated exactly once, whatever the split."""
    from nutpie_amd import _lib

    for rows, batches in ((1, 5), (7, 40), (257, 200), (4096, 30)):
        got, cores = _lib.test_rowpool(threads, rows, batches, use)
        want = (np.arange(1, rows + 1, dtype=np.uint64) * np.uint64(batches * (batches + 1) // 2))
        assert np.array_equal(got, want), (rows, batches)
    assert 1 <= cores <= os.cpu_count()


def test_sampler_constructors_do_not_drop_arguments_silently():
    """The reference's PySampler.from_pymc / from_stan / from_pyfunc take a progress type, a callback and a storage back-end
    (wrapper.rs:1189-1250).  Here progress is driven one layer up and there is no storage back-end: passing either raises."""
    from nutpie_amd import _lib

    with pytest.raises(NotImplementedError, match="storage"):
        _lib.PySampler.from_pymc(None, 1, None, None, None, None, store=object())
    with pytest.raises(NotImplementedError, match="progress"):
        _lib.PySampler.from_stan(None, 1, None, "template", None, None, None)
