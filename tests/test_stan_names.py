"""Known-answer tests of the Stan name parser and the column-major -> C reorder, restating the reference's
own unit tests (src/stan.rs:819-871 `transpose`, :873-1231 `parse_vars`)."""
import numpy as np
import pytest

from nutpie_amd.stan_names import expand_constrained, fortran_to_c_order, parse_stan_variables


def test_transpose_known_answers():
    assert fortran_to_c_order(np.arange(6.0), (2, 3)).tolist() == [0, 2, 4, 1, 3, 5]
    assert fortran_to_c_order(np.arange(6.0), (3, 2)).tolist() == [0, 3, 1, 4, 2, 5]
    assert fortran_to_c_order(np.arange(30.0), (2, 3, 5)).tolist() == [
        0, 6, 12, 18, 24, 2, 8, 14, 20, 26, 4, 10, 16, 22, 28, 1, 7, 13, 19, 25, 3, 9, 15, 21, 27, 5, 11, 17, 23, 29]
    assert fortran_to_c_order(np.arange(30.0), (5, 3, 2)).tolist() == [
        0, 15, 5, 20, 10, 25, 1, 16, 6, 21, 11, 26, 2, 17, 7, 22, 12, 27, 3, 18, 8, 23, 13, 28, 4, 19, 9, 24, 14, 29]
    # the generator named in the reference test: np.arange(n).reshape(shape, order="F").ravel()
    for shape in [(2, 3, 5), (4,), (3, 1, 2, 2)]:
        n = int(np.prod(shape))
        assert np.array_equal(fortran_to_c_order(np.arange(float(n)), shape), np.arange(float(n)).reshape(shape, order="F").ravel())
    # batched over draws
    batch = np.arange(2 * 3 * 6, dtype=float).reshape(2, 3, 6)
    out = fortran_to_c_order(batch, (2, 3))
    assert out.shape == (2, 3, 6) and np.array_equal(out[1, 2], batch[1, 2].reshape((2, 3), order="F").ravel())


def test_parse_vars_known_answers():
    assert parse_stan_variables("") == []
    (v,) = parse_stan_variables("x.1.1,x.2.1,x.3.1,x.1.2,x.2.2,x.3.2")
    assert v.name == "x" and v.shape == (3, 2) and (v.start, v.end) == (0, 6)
    with pytest.raises(ValueError, match="expected order"):
        parse_stan_variables("x.1.2,x.1.1,x.2.1,x.2.2,x.3.1,x.3.2")
    with pytest.raises(ValueError, match="expected order"):
        parse_stan_variables("x.1.2.real,x.1.2.imag")
    re_, im_ = parse_stan_variables("x.1.1.real,x.1.1.imag,x.2.1.real,x.2.1.imag,x.3.1.real,x.3.1.imag")
    assert re_.name == "x.real" and re_.shape == (3, 1) and im_.name == "x.imag" and im_.shape == (3, 1)
    assert (re_.start, re_.end, im_.start, im_.end) == (0, 3, 3, 6)
    (a,) = parse_stan_variables("alpha")
    assert a.name == "alpha" and a.shape == () and a.num_elements == 1
    abc = parse_stan_variables("alpha,beta,gamma")
    assert [v.name for v in abc] == ["alpha", "beta", "gamma"] and [v.start for v in abc] == [0, 1, 2]
    (t,) = parse_stan_variables("theta.1,theta.2,theta.3,theta.4")
    assert t.name == "theta" and t.shape == (4,) and t.num_elements == 4
    (c,) = parse_stan_variables("x:1:2.4:1.1,x:1:2.4:1.2,x:1:2.4:1.3")      # colons and dots inside the name
    assert c.name == "x:1:2.4:1" and c.shape == (3,)


def test_parse_vars_nested_tuples():
    names = """
        a, base, base_i, pair:1, pair:2, nested:1, nested:2:1, nested:2:2.real, nested:2:2.imag,
        arr_pair.1:1, arr_pair.1:2, arr_pair.2:1, arr_pair.2:2,
        arr_very_nested.1:1:1, arr_very_nested.1:1:2:1, arr_very_nested.1:1:2:2.real, arr_very_nested.1:1:2:2.imag, arr_very_nested.1:2,
        arr_2d_pair.1.1:1, arr_2d_pair.1.1:2,
        ultimate.1.1:1.1:1, ultimate.1.1:1.1:2.1, ultimate.1.1:1.1:2.2,
        ultimate.1.1:2.1.1, ultimate.1.1:2.2.1, ultimate.1.1:2.3.1, ultimate.1.1:2.4.1,
        ultimate.1.1:2.1.2, ultimate.1.1:2.2.2, ultimate.1.1:2.3.2, ultimate.1.1:2.4.2"""
    parsed = parse_stan_variables(names)
    want = ["a", "base", "base_i", "pair:1", "pair:2", "nested:1", "nested:2:1", "nested:2:2.real", "nested:2:2.imag",
            "arr_pair.1:1", "arr_pair.1:2", "arr_pair.2:1", "arr_pair.2:2", "arr_very_nested.1:1:1", "arr_very_nested.1:1:2:1",
            "arr_very_nested.1:1:2:2.real", "arr_very_nested.1:1:2:2.imag", "arr_very_nested.1:2", "arr_2d_pair.1.1:1",
            "arr_2d_pair.1.1:2", "ultimate.1.1:1.1:1", "ultimate.1.1:1.1:2", "ultimate.1.1:2"]
    assert [v.name for v in parsed] == want
    shapes = {v.name: v.shape for v in parsed}
    assert all(shapes[n] == () for n in want[:21])
    assert shapes["ultimate.1.1:1.1:2"] == (2,) and shapes["ultimate.1.1:2"] == (4, 2)
    assert parsed[-1].end == sum(v.num_elements for v in parsed)


def test_expand_constrained_layout():
    # tests/test_stan.py:209-249 ("memory order"): a matrix written by Stan column-major comes out [row, col]
    vars_ = parse_stan_variables("mu,m.1.1,m.2.1,m.1.2,m.2.2,m.1.3,m.2.3")
    flat = np.array([[9.0, 11, 21, 12, 22, 13, 23], [8.0, 110, 210, 120, 220, 130, 230]])   # two draws
    out = expand_constrained(flat, vars_)
    assert out["mu"].tolist() == [9.0, 8.0]
    assert out["m"].shape == (2, 2, 3)
    assert out["m"][0].tolist() == [[11, 12, 13], [21, 22, 23]]
    assert out["m"][1].tolist() == [[110, 120, 130], [210, 220, 230]]
