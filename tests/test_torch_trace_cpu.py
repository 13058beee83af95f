"""The torch tracer (``nutpie_amd.torch_trace``) on the CPU: the traced graph and its SYMBOLIC gradient, evaluated with numpy from the
same IR the device code is printed from, against ``torch.autograd`` on the user's function; what is refused and how; the shapes
of what comes out.  (The generated device code itself is checked on the GPU: ``tests/test_gpu_torch_trace.py``.)"""

import os
import sys
import warnings

import numpy as np
import pytest
import torch

import nutpie_amd
from nutpie_amd.torch_trace import UnsupportedTorchOp, trace

sys.path.insert(0, os.path.dirname(__file__))
import torch_models as TM  # noqa: E402


@pytest.mark.parametrize("name", list(TM.ALL) + list(TM.CPU_ONLY))
def test_traced_density_and_gradient_equal_autograd(name):
    D, fn, batched, shared = {**TM.ALL, **TM.CPU_ONLY}[name]()
    tr = trace(fn, D, batched=batched, shared_data=shared)
    cm = tr.compile()
    assert cm.n_dim == D and cm.shapes == {"x": (D,)}
    x = 0.4 * np.random.default_rng(1).normal(size=(8, D))
    lp, g = cm.logp_and_grad_numpy(x)
    lp0, g0 = TM.autograd(fn, x, batched, shared)
    np.testing.assert_allclose(lp, lp0, rtol=1e-12)
    np.testing.assert_allclose(g, g0, rtol=1e-12, atol=1e-12 * np.abs(g0).max())


def test_pieces_of_x_become_parameters_and_gaps_have_zero_gradient():
    def logp(x):
        a, b = x[:, 2:5], x[:, 7]
        return -0.5 * (a * a).sum(-1) - 0.5 * b * b

    tr = trace(logp, 10)
    assert not tr.whole_vector
    m = tr.model
    assert [p.payload if p.dim is None else p.payload for p in m._params] == [(0, 2), (2, 3), (5, 2), 7, (8, 2)]
    cm = tr.compile()
    x = np.random.default_rng(0).normal(size=(3, 10))
    lp, g = cm.logp_and_grad_numpy(x)
    np.testing.assert_allclose(lp, -0.5 * (x[:, 2:5] ** 2).sum(-1) - 0.5 * x[:, 7] ** 2, rtol=1e-14)
    want = np.zeros_like(x)
    want[:, 2:5], want[:, 7] = -x[:, 2:5], -x[:, 7]
    np.testing.assert_array_equal(g, want)


def test_overlapping_pieces_fall_back_to_one_parameter():
    def logp(x):
        return -0.5 * (x[:, 0:4] * x[:, 2:6]).sum(-1) - 0.5 * (x * x).sum(-1)

    tr = trace(logp, 6)
    assert tr.whole_vector
    x = np.random.default_rng(0).normal(size=(4, 6))
    lp, g = tr.compile().logp_and_grad_numpy(x)
    lp0, g0 = TM.autograd(logp, x, True, {})
    np.testing.assert_allclose(lp, lp0, rtol=1e-13)
    np.testing.assert_allclose(g, g0, rtol=1e-13, atol=1e-14)


def test_an_operation_without_counterpart_is_named():
    def logp(x):
        return -torch.cumprod(x * x, -1)[:, -1]

    with pytest.raises(UnsupportedTorchOp, match="cumprod"):
        trace(logp, 5)
    with pytest.raises(UnsupportedTorchOp, match="cumprod"):
        nutpie_amd.from_torch_density(5, logp, compile=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = nutpie_amd.from_torch_density(5, logp)          # "auto": the eager device callback, and a warning that says why
    assert type(m).__name__ == "TorchFuncModel"
    assert any("cumprod" in str(x.message) for x in w)


def test_control_flow_on_the_position_is_refused():
    def logp(x):
        if x[0] > 0:
            return -x[0]
        return x[0]

    with pytest.raises(UnsupportedTorchOp, match="control flow"):
        trace(logp, 2, batched=False)


def test_result_must_be_one_number_per_chain():
    with pytest.raises(UnsupportedTorchOp, match="one value per chain"):
        trace(lambda x: -x * x, 3)
    with pytest.raises(UnsupportedTorchOp, match="does not depend"):
        trace(lambda x: torch.zeros(1, dtype=torch.float64), 3)


def test_with_data_traces_again_and_keeps_the_library():
    D, fn, batched, shared = TM.linear_regression_unbatched()
    m = nutpie_amd.from_torch_density(D, fn, compile=True, batched=False, shared_data=shared)
    assert set(m.data) >= {"X", "y"}                 # shared data keep their names
    x = 0.3 * np.random.default_rng(2).normal(size=(3, D))
    y2 = shared["y"] + 1.0
    m2 = m.with_data(y=y2)
    lp2, g2 = m2.logp_and_grad_numpy(x)
    lp0, g0 = TM.autograd(fn, x, False, {**shared, "y": y2})
    np.testing.assert_allclose(lp2, lp0, rtol=1e-12)
    np.testing.assert_allclose(g2, g0, rtol=1e-12, atol=1e-12)
    assert m2._source == m._source                    # same generated code: the compiled library is found in the cache
    with pytest.raises(ValueError, match="Unknown data variable"):
        m.with_data(z=1.0)


def test_user_expand_function_and_names():
    D, fn, batched, shared = TM.eight_schools()

    def expand(x):
        return {"mu": x[:, 0], "tau": np.exp(x[:, 1]), "theta": x[:, :1] + np.exp(x[:, 1:2]) * x[:, 2:]}

    m = nutpie_amd.from_torch_density(D, fn, compile=True, expand_fn=expand, expanded_names=["mu", "tau", "theta"], expanded_shapes=[(), (), (8,)],
                                      dims={"theta": ("school",)}, coords={"school": np.arange(8)})
    assert m.shapes == {"mu": (), "tau": (), "theta": (8,)}
    draws = np.random.default_rng(0).normal(size=(2, 5, D))
    out = m._expand_draws(draws)
    np.testing.assert_allclose(out["tau"], np.exp(draws[..., 1]))
    assert out["theta"].shape == (2, 5, 8)


def test_generated_source_of_a_traced_model_compiles_for_gfx950():
    D, fn, batched, shared = TM.gamma_poisson()           # (lgamma / digamma helper, selects, segment sums)
    m = nutpie_amd.from_torch_density(D, fn, compile=True, batched=False)
    assert os.path.exists(m.library_path())


def test_an_indexed_assignment_with_repeated_indices_is_refused():
    idx = torch.tensor([0, 1, 1])

    def logp(x):
        tot = torch.zeros(2, dtype=x.dtype)
        tot[idx] += x[:3]                 # (eager torch keeps ONE of the two writes to tot[1]: not a sum)
        return -(tot * tot).sum()

    with pytest.raises(UnsupportedTorchOp, match="repeated indices"):
        trace(logp, 3, batched=False)


def test_integer_powers_of_negative_bases_have_their_values():
    """x ** 20 (and x ** -3, x ** 17) at negative x: repeated squaring, not exp(c log x) — which is NaN there, silently turning the sign of a
    coordinate into a region of rejections."""
    def logp(x):
        return -(x[:, 0] ** 20) - 0.1 * x[:, 1] ** 17 + 0.01 * (2.0 + x[:, 2] * x[:, 2]) ** -3 - 0.5 * (x * x).sum(-1)

    cm = trace(logp, 3).compile()
    x = np.array([[-0.9, -1.1, -0.3], [0.7, 0.5, 2.0], [-1.05, 1.2, -2.0]])
    lp, g = cm.logp_and_grad_numpy(x)
    lp0, g0 = TM.autograd(logp, x, True, {})
    assert np.isfinite(lp).all() and np.isfinite(g).all()
    np.testing.assert_allclose(lp, lp0, rtol=1e-12)
    np.testing.assert_allclose(g, g0, rtol=1e-11)


def test_a_traced_exponent_needs_a_base_that_is_positive_by_construction():
    def ok(x):
        return (torch.exp(x[:, 0]) ** x[:, 1]) * 0.1 - 0.5 * (x * x).sum(-1)

    x = np.random.default_rng(2).normal(size=(5, 2))
    lp, g = trace(ok, 2).compile().logp_and_grad_numpy(x)
    lp0, g0 = TM.autograd(ok, x, True, {})
    np.testing.assert_allclose(lp, lp0, rtol=1e-12)
    np.testing.assert_allclose(g, g0, rtol=1e-11, atol=1e-13)

    def bad(x):
        return (x[:, 0] ** x[:, 1]) - 0.5 * (x * x).sum(-1)

    with pytest.raises(UnsupportedTorchOp, match="positive by construction"):
        trace(bad, 2)


def test_auto_falls_back_when_the_gradient_has_no_counterpart():
    """compile="auto": a function that traces but whose derivative the IR does not have (digamma -> trigamma) takes the eager path with a
    warning, as the docstring says; compile=True names the reason."""
    def logp(x):
        return torch.digamma(torch.exp(x)).sum(-1) - 0.5 * (x * x).sum(-1)

    with pytest.raises((UnsupportedTorchOp, NotImplementedError)):
        nutpie_amd.from_torch_density(5, logp, compile=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = nutpie_amd.from_torch_density(5, logp)          # "auto"
    assert any("eagerly" in str(x.message) for x in w)
    assert m.n_dim == 5 and not hasattr(m, "library_path")   # the batched-callback model of from_torchfunc


def test_var_names_of_a_traced_model_filter_the_expand_functions_variables():
    def logp(x):
        return -0.5 * (x * x).sum(-1)

    def expand(x):
        return {"x": x, "y": np.exp(x[:, :1])}

    kw = dict(n_dim=3, expand_fn=expand, expanded_names=["x", "y"], expanded_shapes=[(3,), (1,)])
    assert list(nutpie_amd.compile_pymc_model(logp, **kw).shapes) == ["x", "y"]
    m = nutpie_amd.compile_pymc_model(logp, var_names=["y"], **kw)
    assert list(m.shapes) == ["y"]
    with pytest.raises(KeyError):
        nutpie_amd.compile_pymc_model(logp, var_names=["nope"], **kw)
    with pytest.raises(ValueError, match="freeze_model"):
        nutpie_amd.compile_pymc_model(logp, n_dim=3, freeze_model=False)
