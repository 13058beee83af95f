"""The dense-precision Gaussian inside the engine (nphip_model_dense_gaussian; BASELINE.json configs[1] read as a DENSE correlated
Gaussian, SURVEY.md §8d variant (ii)): its gradient is the engine's own fp64 matrix-core GEMM (nutpie_amd/csrc/dense_tile.h).

The matrix core accumulates the four products of one v_mfma_f64_16x16x4_f64 as a chain of fused multiply-adds, so the GEMM's
summation order is a contract (include/nphip_spec.h "dense gradient") the CPU oracle restates with std::fma — tolerance ZERO:
gradients, log-densities, draws and every statistic are compared bit for bit, as for the fused tridiagonal models."""
import numpy as np
import pytest

from nutpie_amd.gaussian import dense_precision
from tests.conftest import assert_trace_equal
from tests.test_gpu_parity import oracle_settings, run_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from nutpie_amd import _lib

    _lib.lib()
    return _lib


@pytest.mark.parametrize("n,dim", [(1, 1), (5, 3), (64, 64), (65, 63), (70, 100), (33, 257), (128, 1000), (300, 1023)])
@pytest.mark.parametrize("waves", [1, 4])
def test_gradient_gemm_is_the_contract_fma_chain(hip, oracle, n, dim, waves):
    """Ragged shapes on purpose: rows that end inside a fragment (dim % 16 != 0), odd row lengths (rows of the dense staging
    buffers then start on 8-byte boundaries), partial tiles in both directions, a single chain."""
    rng = np.random.default_rng(1000 * n + dim)
    P = dense_precision(dim, seed=3)
    mu = rng.normal(size=dim)
    x = rng.normal(size=(n, dim)) * 3
    g, lp = hip.test_dense_grad(x, P, mu, waves=waves)
    go, lpo = oracle.dense_grad(x, P, mu, waves=waves)
    assert np.array_equal(g, go), f"gradient: {int((g != go).sum())} of {g.size} elements differ"
    assert np.array_equal(lp, lpo)
    ref = -(x - mu) @ P                                   # and the restatement is the mathematics (free summation order: rounding)
    np.testing.assert_allclose(g, ref, rtol=0, atol=1e-11 * np.abs(ref).max())


def test_non_finite_rows_stay_in_their_row(hip, oracle):
    """A chain whose position is not finite (a diverged trajectory) must not leak into the gradients of its neighbours in the
    tile — in particular not through the zero-padded tail of the last k-step (0 * inf)."""
    dim, n = 100, 20
    P = dense_precision(dim, seed=4)
    x = np.random.default_rng(0).normal(size=(n, dim))
    x[7, 3] = np.inf
    x[11, 99] = np.nan
    g, lp = hip.test_dense_grad(x, P)
    go, lpo = oracle.dense_grad(x, P)
    ok = np.ones(n, bool)
    ok[[7, 11]] = False
    assert np.array_equal(g[ok], go[ok]) and np.array_equal(lp[ok], lpo[ok])
    assert not np.isfinite(g[7]).any() and not np.isfinite(lp[[7, 11]]).any()


def test_unsymmetric_matrix_is_refused(hip):
    P = dense_precision(5, seed=1)
    P[1, 3] += 1e-3
    with pytest.raises(ValueError, match="symmetric"):
        hip.DenseGaussianModel(P)


# launch-per-evaluation form (host_persist = 1: the two kernels behind the engine's device-callback path, with and without the HIP graph, in
# groups) and the RESIDENT form (default up to 1024 dimensions: the register-resident leaf with the launch-wide GEMM as its evaluation —
# clusters of workgroups that rendezvous inside the kernel; partial clusters, spare waves, chains that finish early, short launches)
@pytest.mark.parametrize("dim,chains,launch,mode", [
    (100, 8, {"host_persist": 1}, "launch-per-evaluation"), (100, 8, {"host_persist": 1, "graph_steps": -1}, "launch-per-evaluation"),
    (37, 5, {"host_persist": 1}, "launch-per-evaluation"), (300, 70, {"host_persist": 1}, "launch-per-evaluation"),
    (100, 16, {"host_groups": 2}, "launch-per-evaluation"), (1100, 6, {}, "launch-per-evaluation"),
    (100, 8, {}, "resident"), (37, 5, {}, "resident"), (100, 1, {}, "resident"), (300, 70, {"evals_per_launch": 7, "host_persist": 2}, "resident"),
    (1000, 6, {"host_persist": 2}, "resident"), (129, 1024, {}, "resident"), (257, 301, {"evals_per_launch": 64}, "resident"),
    (1024, 130, {"host_persist": 2}, "resident"),
    # (round 6: a job too small for its clusters — a member of a die-local cluster would compute three or more column tiles per round — takes
    #  the launch-per-evaluation form by itself; host_persist = 2 above forces the resident form onto such shapes)
    (1000, 6, {}, "launch-per-evaluation"), (1024, 130, {}, "launch-per-evaluation"), (300, 70, {}, "launch-per-evaluation")])
def test_dense_gaussian_bit_identical_to_the_oracle(hip, oracle, dim, chains, launch, mode):
    """A whole job — warm-up with step-size search and mass-matrix adaptation, then sampling — against oracle.sample_dense."""
    P = dense_precision(dim, seed=5, cond_lo=0.1, cond_hi=10)
    mu = np.linspace(-1, 1, dim)
    tune, draws = (60, 20) if dim < 1000 and chains < 1000 else (30, 8)
    info = {}
    got, W = run_engine(hip, hip.DenseGaussianModel(P, mu), chains=chains, tune=tune, draws=draws, seed=11, launch=launch, info=info)
    assert info["host_mode"] == mode
    want = oracle.sample_dense(oracle_settings(oracle, chains=chains, tune=tune, draws=draws, seed=11, W=W), P, mu)
    assert_trace_equal(got, want)


def test_resident_form_at_the_benchmark_shape(hip, oracle):
    """BASELINE.json configs[1] as a dense Gaussian: 1000 dimensions x 1024 chains — every CU holds a workgroup, every die two clusters."""
    P = dense_precision(1000)
    info = {}
    got, W = run_engine(hip, hip.DenseGaussianModel(P), chains=1024, tune=4, draws=2, seed=1, info=info)
    assert info["host_mode"] == "resident" and W == 1
    want = oracle.sample_dense(oracle_settings(oracle, chains=1024, tune=4, draws=2, seed=1, W=1), P)
    assert_trace_equal(got, want)


def test_dense_gaussian_through_sample(hip):
    """The front door: nutpie_amd.sample on the dense model recovers the covariance's leading moments."""
    import nutpie_amd

    dim = 20
    m = nutpie_amd.dense_gaussian(dim, seed=2, cond_lo=0.5, cond_hi=2)
    tr = nutpie_amd.sample(m, draws=400, tune=300, chains=64, seed=3, progress_bar=False)
    x = np.asarray(tr.posterior["x"]).reshape(-1, dim)
    cov = m.covariance()
    assert np.abs(x.mean(0)).max() < 5 * np.sqrt(np.diag(cov).max() / 2000)
    np.testing.assert_allclose(np.cov(x.T), cov, atol=0.15 * np.abs(cov).max())
