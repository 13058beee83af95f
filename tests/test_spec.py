"""The numerics contract (include/nphip_spec.h) as restated by the oracle: known-answer and accuracy tests."""
import math

import numpy as np
import pytest


def test_philox_random123_known_answers(oracle):
    # Random123 kat_vectors for philox4x32-10 (counter words, key words -> output)
    assert oracle.philox(0, 0, 0, 0, 0) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert oracle.philox(0xFFFFFFFFFFFFFFFF, *([0xFFFFFFFF] * 4)) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert oracle.philox(0x299F31D0A4093822, 0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344) == (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def _ulp_err(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


@pytest.mark.parametrize("fn,ref,x", [
    ("exp", np.exp, np.random.default_rng(1).uniform(-700, 700, 100000)),
    ("exp", np.exp, np.random.default_rng(2).uniform(-2, 2, 100000)),
    ("log", np.log, np.exp(np.random.default_rng(3).uniform(-300, 300, 100000))),
    ("log", np.log, 1 + np.random.default_rng(4).uniform(-0.01, 0.01, 100000)),
    ("log1p", np.log1p, np.random.default_rng(5).uniform(0, 1, 100000)),
])
def test_detmath_accuracy(oracle, fn, ref, x):
    y = oracle.detmath(fn, x)
    assert _ulp_err(y, ref(x)).max() <= 4.0


def test_detmath_special_values(oracle):
    assert oracle.detmath("exp", [-800.0, 710.0, 0.0]).tolist() == [0.0, math.inf, 1.0]
    assert oracle.detmath("exp", [-740.0])[0] == pytest.approx(math.exp(-740.0), rel=1e-3)
    y = oracle.detmath("log", [0.0, math.inf, 1.0, 5e-324])
    assert y[0] == -math.inf and y[1] == math.inf and y[2] == 0.0 and y[3] == pytest.approx(math.log(5e-324))
    assert np.isnan(oracle.detmath("log", [-1.0])[0])
    assert np.isnan(oracle.detmath("exp", [math.nan])[0])


def test_sincos2pi(oracle):
    u = np.random.default_rng(6).uniform(0, 1, 100000)
    s, c = oracle.detmath("sin2pi", u), oracle.detmath("cos2pi", u)
    ang = 2 * np.pi * u.astype(np.longdouble)
    assert np.abs(s - np.sin(ang).astype(np.float64)).max() < 4e-16
    assert np.abs(c - np.cos(ang).astype(np.float64)).max() < 4e-16
    # exact values at the quadrant boundaries
    assert oracle.detmath("sin2pi", [0.0, 0.25, 0.5, 0.75]).tolist() == [0.0, 1.0, 0.0, -1.0]
    assert oracle.detmath("cos2pi", [0.0, 0.25, 0.5, 0.75]).tolist() == [1.0, 0.0, -1.0, 0.0]


def test_logaddexp(oracle):
    rng = np.random.default_rng(7)
    for a, b in rng.uniform(-50, 50, size=(2000, 2)):
        assert oracle.logaddexp(a, b) == pytest.approx(np.logaddexp(a, b), rel=1e-14, abs=1e-14)
    assert oracle.logaddexp(1.5, 1.5) == 1.5 + math.log(2.0)
    assert oracle.logaddexp(-math.inf, 2.0) == 2.0
    assert oracle.logaddexp(0.0, -math.inf) == 0.0


def test_normals_are_standard_normal(oracle):
    from scipy import stats

    z = oracle.normals(123, 0, 0, 1, 200001)
    assert abs(z.mean()) < 4 / math.sqrt(z.size)
    assert abs(z.var() - 1) < 0.02
    assert stats.kstest(z, "norm").pvalue > 1e-3
    # streams: different chain / draw / purpose give different numbers, same inputs the same
    assert np.array_equal(z, oracle.normals(123, 0, 0, 1, 200001))
    assert not np.array_equal(z[:10], oracle.normals(123, 1, 0, 1, 10))
    assert not np.array_equal(z[:10], oracle.normals(123, 0, 1, 1, 10))
    assert not np.array_equal(z[:10], oracle.normals(123, 0, 0, 5, 10))
    assert not np.array_equal(z[:10], oracle.normals(124, 0, 0, 1, 10))
    # an odd length is a prefix of the even one (pairing does not depend on n)
    assert np.array_equal(oracle.normals(9, 2, 3, 1, 7), oracle.normals(9, 2, 3, 1, 8)[:7])


@pytest.mark.parametrize("waves", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 1000, 10000])
def test_dot_geometry(oracle, waves, n):
    rng = np.random.default_rng(n * 31 + waves)
    x, y = rng.normal(size=n), rng.normal(size=n)
    exact = math.fsum(x * y)
    got = oracle.dot(x, y, waves)
    assert got == pytest.approx(exact, rel=1e-12, abs=1e-12)


def test_dot_order_is_the_contract(oracle):
    # the summation order is part of the contract: for n <= 128 every element has its own accumulator,
    # so the result is the pairwise tree (x0*y0 + x1*y1) combined by the xor butterfly
    x = np.array([1e16, 1.0, -1e16, 1.0])
    y = np.ones(4)
    # lanes: l0 = 1e16 + 1 = 1e16 ; l1 = -1e16 + 1 = -1e16 ; first butterfly stage pairs lanes 0 and 1: 0
    assert oracle.dot(x, y, 1) == 0.0
    # different geometry (element 2,3 -> still lane 1): same here
    assert oracle.dot(x, y, 2) == 0.0
