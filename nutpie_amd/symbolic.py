"""A model front-end that PRODUCES a device density: expressions -> reverse-mode gradient -> HIP source -> resident kernel.

The reference compiles a PyMC model by asking PyTensor for the graph of ``logp`` and its gradient, joining them into one
function of the flat unconstrained vector and handing that to a compiler (``python/nutpie/compile_pymc.py:523-871``); the
model's observed data stay *shared variables* that ``with_data`` swaps (``:140-166``, ``:239-269``).  PyMC and PyTensor are
not available on the target image, so this module is the part of that pipeline that does not need them, built for the GPU:

    m = Model()
    mu    = m.param("mu")                           # scalar
    sigma = m.param("sigma", lower=0.0)             # log-transformed, Jacobian added  (PyMC's LogTransform)
    a     = m.param("a", dim="county", size=85, zero_sum=True)     # PyMC's ZeroSumTransform
    y     = m.data("y", y_values, dim="obs")
    idx   = m.index("county_idx", county_values, dim="obs", into="county")
    m.add_logp(normal_lpdf(a, 0.0, 1.0).sum())
    m.add_logp(normal_lpdf(y, mu + a[idx], sigma).sum())
    m.deterministic("a_plus_mu", a + mu)            # (the parameters themselves are reported without asking)
    compiled = m.compile()                          # -> nutpie_amd.density.DensitySourceModel

``compile`` differentiates the expression graph symbolically (reverse mode, the gradient is a graph in the same IR),
schedules logp and gradient together into wave-wide loops — one wavefront evaluates one chain's density, lanes stride over
the elements of a *dimension* (a coordinate of the model: counties, observations) —, and prints the HIP device function
``nphip_density`` that :func:`nutpie_amd.from_density_source` compiles into the engine's resident NUTS kernel.

Also: ``m.matrix("X", values, dim="obs", cols="coef")`` and ``X @ beta`` (a design matrix: lowered to a sum over its columns, the
transposed product to one wave-wide sum per column); ``lower=`` / ``upper=`` / both (log and logit-interval transforms with their
Jacobians); densities ``normal``, ``halfnormal``, ``student_t``, ``cauchy``, ``halfcauchy``, ``exponential``, ``lognormal``, ``gamma``,
``bernoulli_logit``, ``poisson_log``; ``compile(waves_per_chain=...)`` (default: from the LDS the model needs); ``Model.profile()``
(cycle attribution of the generated code on the GPU).

The IR has five kinds of nodes: scalars; element-wise values over a dimension; ``Sum`` (dimension -> scalar);
``Gather`` (``v[idx]``: dimension A -> dimension B through an integer data array); ``SegSum`` (its transpose: for every
element of A the sum over the elements of B that point to it).  The transpose of a gather is evaluated without atomics and
in a fixed order: the adjoints are stored in LDS *grouped by target* (the host computes the grouping permutation once per
data set) and every lane sums a contiguous range.  Values a later loop needs from an earlier one (gather sources, adjoints)
live in per-chain LDS (the largest of them in device memory when the LDS is too small: models with tens of thousands of
observations); the model's data are staged into the workgroup's shared LDS once per launch while they fit.  Everything else is
recomputed where it is used.

``deterministic`` values and the constrained parameters (the reference's expand step: ``compile_pymc.py:601-666``) are generated
code too: a second device function, ``nphip_expand``, that the engine runs over the stored draws in one batched launch
(``nphip_model_set_device_expand``); values on a data dimension — whose size changes with ``with_data`` — and ``_expand_draws`` of
positions that did not come out of a sampler are evaluated with numpy from the same graph.
"""

from __future__ import annotations

import math
import os
import weakref
from typing import Any

import numpy as np

__all__ = ["Model", "Expr", "Matrix", "exp", "log", "log1p", "sqrt", "softplus", "sigmoid", "tanh", "expm1", "erf", "erfc", "sin", "cos", "atan", "lgamma", "digamma",
           "absolute", "sign", "select", "pad", "trunc", "where_lt", "elem", "stack", "normal_lpdf",
           "halfnormal_lpdf", "student_t_lpdf", "cauchy_lpdf", "halfcauchy_lpdf", "exponential_lpdf", "lognormal_lpdf", "gamma_lpdf",
           "inverse_gamma_lpdf", "beta_lpdf", "laplace_lpdf", "logistic_lpdf", "weibull_lpdf", "uniform_lpdf",
           "bernoulli_logit_lpmf", "binomial_logit_lpmf", "negative_binomial_log_lpmf", "poisson_log_lpmf", "dirichlet_lpdf", "flat_lpdf"]

_WAVE = 64
_SEG_BATCH = os.environ.get("NUTPIE_AMD_SEG_MODE", "select") != "loop"   # (developer switch: "loop" = plain loops over a segment)
_UNROLL = 4   # iterations of a loop whose reads are issued together (a lone wave waits out every access otherwise)


# --------------------------------------------------------------------------- IR
class Dim:
    """A named coordinate of the model: the range one wave-wide loop runs over."""

    def __init__(self, name: str, size, runtime_len: str | None = None, runtime_div: int = 1):
        self.name = name
        self.size = size                  # int, or None for a data dimension whose length comes from the data block
        self.runtime_len = runtime_len    # the data array whose length is the dimension's (times runtime_div: a matrix's rows)
        self.runtime_div = runtime_div

    def len_c(self) -> str:
        if self.size is not None:
            return str(self.size)
        return f"data.n_{self.runtime_len}" + (f" / {self.runtime_div}" if self.runtime_div != 1 else "")

    def len_py(self, data) -> int:
        return self.size if self.size is not None else int(np.asarray(data[self.runtime_len]).size) // self.runtime_div

    def __repr__(self):
        return f"Dim({self.name})"


class Index:
    """Integer data: for every element of ``dim`` the element of ``into`` it refers to."""

    def __init__(self, name: str, dim: Dim, into: Dim):
        self.name, self.dim, self.into = name, dim, into


class Matrix:
    """Float data with one row per element of ``dim`` and one column per element of ``cols`` (a design matrix).  ``X @ v`` with ``v``
    on ``cols`` is the linear predictor on ``dim``: a sum over the columns of (column x element of v), so that its transpose —
    the gradient with respect to ``v`` — is one wave-wide sum per column."""

    def __init__(self, name: str, dim: Dim, cols: Dim):
        self.name, self.dim, self.cols = name, dim, cols

    def column(self, c: int) -> "Expr":
        return Expr("datacol", (), self.dim, (self.name, int(c), self.cols.size))

    def __matmul__(self, v) -> "Expr":
        if not isinstance(v, Expr) or v.dim is not self.cols:
            raise ValueError(f"matrix {self.name!r} multiplies a vector on dimension {self.cols.name!r}")
        total = None
        for c in range(self.cols.size):
            term = self.column(c) * elem(v, c)
            total = term if total is None else total + term
        return total


class Expr:
    """A node of the expression graph.  ``dim`` is None for scalars.  Nodes are hash-consed by ``Model``-independent structural
    keys, so that the gradient graph shares its sub-expressions with the forward graph."""

    _table: "weakref.WeakValueDictionary[tuple, Expr]" = None   # (weak: the nodes of a discarded model are collected with it)
    _count = 0
    __array_ufunc__ = None     # numpy scalars defer to the operators below

    def __new__(cls, op: str, args: tuple = (), dim: Dim | None = None, payload: Any = None):
        key = (op, tuple(id(a) for a in args), id(dim) if dim is not None else None, payload if not isinstance(payload, (Index, Dim)) else id(payload))
        if Expr._table is None:
            Expr._table = weakref.WeakValueDictionary()
        hit = Expr._table.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.op, self.args, self.dim, self.payload = op, tuple(args), dim, payload
        self.id = Expr._count
        Expr._count += 1
        Expr._table[key] = self
        return self

    # ---- construction helpers
    @staticmethod
    def const(v) -> "Expr":
        return Expr("const", (), None, float(v))

    @staticmethod
    def wrap(v) -> "Expr":
        return v if isinstance(v, Expr) else Expr.const(v)

    def is_const(self, v=None):
        return self.op == "const" and (v is None or self.payload == v)

    def _bin(self, op, other, swap=False):
        a, b = Expr.wrap(self), Expr.wrap(other)
        if swap:
            a, b = b, a
        return _binary(op, a, b)

    __add__ = lambda s, o: s._bin("add", o)           # noqa: E731
    __radd__ = lambda s, o: s._bin("add", o, True)    # noqa: E731
    __sub__ = lambda s, o: s._bin("sub", o)           # noqa: E731
    __rsub__ = lambda s, o: s._bin("sub", o, True)    # noqa: E731
    __mul__ = lambda s, o: s._bin("mul", o)           # noqa: E731
    __rmul__ = lambda s, o: s._bin("mul", o, True)    # noqa: E731
    __truediv__ = lambda s, o: s._bin("div", o)       # noqa: E731
    __rtruediv__ = lambda s, o: s._bin("div", o, True)  # noqa: E731

    def __neg__(self):
        return _unary("neg", self)

    def __pow__(self, k):
        if k == 2:
            return self * self
        raise TypeError("only `** 2` is supported: write other powers with exp / log")

    def __getitem__(self, index: Index) -> "Expr":
        if not isinstance(index, Index):
            raise TypeError("an expression is indexed with a Model.index(...) array")
        if self.dim is not index.into:
            raise ValueError(f"index {index.name!r} points into dimension {index.into.name!r}, the expression lives on {self.dim.name if self.dim else 'no dimension'!r}")
        return Expr("gather", (self,), index.dim, index)

    def sum(self) -> "Expr":
        if self.dim is None:
            raise ValueError("sum() of a scalar")
        return Expr("sum", (self,), None, None)

    def max(self, constant: bool = False) -> "Expr":
        """The largest element.  ``constant``: a value the result does not depend on in exact arithmetic (the shift of a softmax or a
        log-sum-exp) — no gradient flows through it; otherwise the gradient goes to the element(s) that attain it."""
        if self.dim is None:
            return self
        return Expr("max", (self,), None, "constant" if constant else None)

    def __repr__(self):
        return f"<{self.op}#{self.id}{'@' + self.dim.name if self.dim else ''}>"


def _join(a: Expr, b: Expr) -> Dim | None:
    if a.dim is None:
        return b.dim
    if b.dim is None or a.dim is b.dim:
        return a.dim
    raise ValueError(f"operands live on different dimensions ({a.dim.name!r}, {b.dim.name!r}): index one into the other")


def _binary(op: str, a: Expr, b: Expr) -> Expr:
    # constant folding and the identities the gradient graph is full of
    if a.op == "const" and b.op == "const":
        x, y = a.payload, b.payload
        return Expr.const({"add": x + y, "sub": x - y, "mul": x * y, "div": x / y if y != 0.0 else math.copysign(math.inf, x)}[op])
    if op == "mul" and b.op == "const" and a.op != "const":
        a, b = b, a                                   # constants first
    if op == "add":
        if a.is_const(0.0):
            return b
        if b.is_const(0.0):
            return a
        if a is b:
            return _binary("mul", Expr.const(2.0), a)
    elif op == "sub":
        if b.is_const(0.0):
            return a
        if a.is_const(0.0):
            return _unary("neg", b)
    elif op == "mul":
        if a.is_const(1.0):
            return b
        if b.is_const(1.0):
            return a
        if a.is_const(0.0) or b.is_const(0.0):
            return Expr.const(0.0)
        if a.is_const(-1.0):
            return _unary("neg", b)
        if a.op == "const" and b.op == "mul" and b.args[0].op == "const":
            return _binary("mul", Expr.const(a.payload * b.args[0].payload), b.args[1])     # c1 (c2 x) = (c1 c2) x
        if a.op == "const" and b.op == "neg":
            return _binary("mul", Expr.const(-a.payload), b.args[0])
    elif op == "div":
        if b.is_const(1.0):
            return a
        if a.is_const(0.0):
            return a
        if b.op == "const" and b.payload != 0.0:
            return _binary("mul", a, Expr.const(1.0 / b.payload))     # (the numpy evaluation follows the same graph)
        if a.dim is not None and b.dim is None:
            # one division per evaluation instead of one per element
            return _binary("mul", a, _binary("div", Expr.const(1.0), b))
    return Expr(op, (a, b), _join(a, b))


def _digamma_py(x: float) -> float:
    """psi(x) the way the generated device code computes it (recurrence up to 10, then the asymptotic series)"""
    if x <= 0.0:
        if x == math.floor(x):
            return math.nan
        return _digamma_py(1.0 - x) - math.pi / math.tan(math.pi * x)
    r = 0.0
    while x < 10.0:
        r -= 1.0 / x
        x += 1.0
    f = 1.0 / (x * x)
    return r + math.log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))))


_UNARY_FOLD = {"neg": lambda v: -v, "exp": math.exp, "log": math.log, "log1p": math.log1p, "sqrt": math.sqrt,
               "softplus": lambda v: max(v, 0.0) + math.log1p(math.exp(-abs(v))), "sigmoid": lambda v: 1.0 / (1.0 + math.exp(-v)),
               "tanh": math.tanh, "expm1": math.expm1, "erf": math.erf, "erfc": math.erfc, "sin": math.sin, "cos": math.cos, "atan": math.atan,
               "lgamma": math.lgamma, "digamma": _digamma_py, "abs": abs, "sign": lambda v: float((v > 0) - (v < 0))}


def _unary(op: str, a) -> Expr:
    a = Expr.wrap(a)
    if a.op == "const":
        return Expr.const(_UNARY_FOLD[op](a.payload))
    if op == "neg" and a.op == "neg":
        return a.args[0]
    if op == "neg" and a.op == "mul" and a.args[0].op == "const":
        return _binary("mul", Expr.const(-a.args[0].payload), a.args[1])
    if op == "log" and a.op == "exp":      # log sigma of a log-transformed sigma: the raw parameter
        return a.args[0]
    return Expr(op, (a,), a.dim)


def exp(a):
    return _unary("exp", a)


def log(a):
    return _unary("log", a)


def log1p(a):
    return _unary("log1p", a)


def sqrt(a):
    return _unary("sqrt", a)


def softplus(a):
    """log(1 + e^a), evaluated as max(a, 0) + log1p(e^-|a|)."""
    return _unary("softplus", a)


def sigmoid(a):
    return _unary("sigmoid", a)


def tanh(a):
    return _unary("tanh", a)


def expm1(a):
    return _unary("expm1", a)


def erf(a):
    return _unary("erf", a)


def erfc(a):
    return _unary("erfc", a)


def sin(a):
    return _unary("sin", a)


def cos(a):
    return _unary("cos", a)


def atan(a):
    return _unary("atan", a)


def lgamma(a):
    return _unary("lgamma", a)


def digamma(a):
    return _unary("digamma", a)


def absolute(a):
    return _unary("abs", a)


def sign(a):
    return _unary("sign", a)


def _join3(c: Expr, a: Expr, b: Expr) -> Dim | None:
    d = None
    for v in (c, a, b):
        if v.dim is not None:
            if d is not None and v.dim is not d:
                raise ValueError(f"operands live on different dimensions ({d.name!r}, {v.dim.name!r})")
            d = v.dim
    return d


def select(c, a, b, strict: bool = True) -> Expr:
    """``a`` where ``c > 0`` (``strict``) resp. ``c >= 0``, else ``b`` — element-wise; the condition carries no gradient."""
    c, a, b = Expr.wrap(c), Expr.wrap(a), Expr.wrap(b)
    if c.op == "const":
        return a if (c.payload > 0.0 if strict else c.payload >= 0.0) else b
    if a is b:
        return a
    return Expr("sel_gt" if strict else "sel_ge", (c, a, b), _join3(c, a, b))


def pad(v: Expr, dim: Dim) -> Expr:
    """``v`` (on a shorter dimension of fixed size) on the first elements of ``dim``, zero on the rest."""
    v = Expr.wrap(v)
    if v.dim is None or v.dim is dim:
        return v
    if v.dim.size is None or dim.size is None or v.dim.size > dim.size:
        raise ValueError("pad() goes from a fixed-size dimension to a longer fixed-size dimension")
    return Expr("pad", (v,), dim, None)


def trunc(v: Expr, dim: Dim) -> Expr:
    """The first ``dim.size`` elements of ``v`` (on a longer dimension of fixed size), as a value on ``dim``."""
    v = Expr.wrap(v)
    if v.dim is None or v.dim is dim:
        return v
    if v.dim.size is None or dim.size is None or v.dim.size < dim.size:
        raise ValueError("trunc() goes from a fixed-size dimension to a shorter fixed-size dimension")
    if v.op == "pad" and v.args[0].dim is dim:
        return v.args[0]
    return Expr("trunc", (v,), dim, None)


def where_lt(dim: Dim, k: int, a, b) -> Expr:
    """``a`` on the first ``k`` elements of ``dim``, ``b`` on the rest."""
    a, b = Expr.wrap(a), Expr.wrap(b)
    for v in (a, b):
        if v.dim is not None and v.dim is not dim:
            raise ValueError("where_lt: operands must be scalars or live on `dim`")
    return Expr("where_lt", (a, b), dim, int(k))


def _bcast(a: Expr, dim: Dim) -> Expr:
    return a if a.dim is dim else Expr("bcast", (a,), dim, None)


def _dim_len(dim: Dim) -> Expr:
    return Expr.const(dim.size) if dim.size is not None else Expr("dimlen", (), None, dim)


def _segsum(e: Expr, index: Index) -> Expr:
    if e.is_const(0.0):
        return e
    return Expr("segsum", (_bcast(e, index.dim),), index.into, index)


def elem(v: Expr, c: int) -> Expr:
    """Element ``c`` of a vector on a fixed-size dimension, as a scalar."""
    if v.dim is None:
        return v
    if v.dim.size is None or not 0 <= c < v.dim.size:
        raise ValueError("elem() needs a vector on a fixed-size dimension and an index inside it")
    if v.op == "vparam" and c >= v.payload[1]:
        return Expr.const(0.0)            # the padding element of a zero-sum parameter
    if v.op == "stack":
        return v.args[c]
    if v.op == "bcast":
        return v.args[0]
    return Expr("elem", (v,), None, int(c))


def stack(scalars, dim: Dim) -> Expr:
    """The vector on ``dim`` whose elements are the given scalars."""
    scalars = [Expr.wrap(v) for v in scalars]
    if dim.size is None or len(scalars) != dim.size or any(v.dim is not None for v in scalars):
        raise ValueError("stack() needs one scalar per element of a fixed-size dimension")
    if all(v.is_const(0.0) for v in scalars):
        return Expr.const(0.0)
    return Expr("stack", tuple(scalars), dim, None)


# --------------------------------------------------------------------------- densities
_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def normal_lpdf(x, mu, sigma) -> Expr:
    x, mu, sigma = Expr.wrap(x), Expr.wrap(mu), Expr.wrap(sigma)
    z = (x - mu) / sigma
    return -0.5 * (z * z) - log(sigma) - _HALF_LOG_2PI


def halfnormal_lpdf(x, sigma) -> Expr:
    """x >= 0 (a ``lower=0`` parameter)."""
    x, sigma = Expr.wrap(x), Expr.wrap(sigma)
    z = x / sigma
    return -0.5 * (z * z) - log(sigma) + 0.5 * math.log(2.0 / math.pi)


def _is_number(v) -> bool:
    return isinstance(v, (int, float, np.integer, np.floating))


def student_t_lpdf(x, nu, mu, sigma) -> Expr:
    """``nu``: a Python number (its log-gamma terms are folded on the host) or an expression (a parameter: ``lgamma`` in the kernel)."""
    x, mu, sigma = Expr.wrap(x), Expr.wrap(mu), Expr.wrap(sigma)
    z = (x - mu) / sigma
    if _is_number(nu):
        nu = float(nu)
        c = math.lgamma(0.5 * (nu + 1.0)) - math.lgamma(0.5 * nu) - 0.5 * math.log(nu * math.pi)
        return c - log(sigma) - (0.5 * (nu + 1.0)) * log1p((z * z) / nu)
    nu = Expr.wrap(nu)
    c = lgamma(0.5 * (nu + 1.0)) - lgamma(0.5 * nu) - 0.5 * log(nu * math.pi)
    return c - log(sigma) - (0.5 * (nu + 1.0)) * log1p((z * z) / nu)


def cauchy_lpdf(x, mu, gamma) -> Expr:
    x, mu, gamma = Expr.wrap(x), Expr.wrap(mu), Expr.wrap(gamma)
    z = (x - mu) / gamma
    return -math.log(math.pi) - log(gamma) - log1p(z * z)


def halfcauchy_lpdf(x, gamma) -> Expr:
    """x >= 0 (a ``lower=0`` parameter)."""
    x, gamma = Expr.wrap(x), Expr.wrap(gamma)
    z = x / gamma
    return math.log(2.0 / math.pi) - log(gamma) - log1p(z * z)


def exponential_lpdf(x, rate) -> Expr:
    """x >= 0."""
    x, rate = Expr.wrap(x), Expr.wrap(rate)
    return log(rate) - rate * x


def lognormal_lpdf(x, mu, sigma) -> Expr:
    """x > 0."""
    x, mu, sigma = Expr.wrap(x), Expr.wrap(mu), Expr.wrap(sigma)
    lx = log(x)
    z = (lx - mu) / sigma
    return -0.5 * (z * z) - log(sigma) - lx - _HALF_LOG_2PI


def gamma_lpdf(x, alpha, beta) -> Expr:
    """x > 0; the shape ``alpha`` is a Python number (its log-gamma is folded on the host) or an expression, the rate ``beta`` an
    expression."""
    x, beta = Expr.wrap(x), Expr.wrap(beta)
    if _is_number(alpha):
        alpha = float(alpha)
        return alpha * log(beta) - math.lgamma(alpha) + (alpha - 1.0) * log(x) - beta * x
    alpha = Expr.wrap(alpha)
    return alpha * log(beta) - lgamma(alpha) + (alpha - 1.0) * log(x) - beta * x


def inverse_gamma_lpdf(x, alpha, beta) -> Expr:
    """x > 0; shape ``alpha`` (number or expression), scale ``beta``."""
    x, beta = Expr.wrap(x), Expr.wrap(beta)
    lg = math.lgamma(float(alpha)) if _is_number(alpha) else lgamma(Expr.wrap(alpha))
    alpha = float(alpha) if _is_number(alpha) else Expr.wrap(alpha)
    return alpha * log(beta) - lg - (alpha + 1.0) * log(x) - beta / x


def beta_lpdf(x, a, b) -> Expr:
    """0 < x < 1 (a ``lower=0, upper=1`` parameter); ``a``, ``b`` numbers or expressions."""
    x = Expr.wrap(x)
    if _is_number(a) and _is_number(b):
        a, b = float(a), float(b)
        const = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
        return const + (a - 1.0) * log(x) + (b - 1.0) * log1p(-x)
    a, b = Expr.wrap(a), Expr.wrap(b)
    return lgamma(a + b) - lgamma(a) - lgamma(b) + (a - 1.0) * log(x) + (b - 1.0) * log1p(-x)


def laplace_lpdf(x, mu, b) -> Expr:
    x, mu, b = Expr.wrap(x), Expr.wrap(mu), Expr.wrap(b)
    return -math.log(2.0) - log(b) - absolute(x - mu) / b


def logistic_lpdf(x, mu, s_) -> Expr:
    x, mu, s_ = Expr.wrap(x), Expr.wrap(mu), Expr.wrap(s_)
    z = (x - mu) / s_
    return -z - log(s_) - 2.0 * softplus(-z)


def weibull_lpdf(x, alpha, beta) -> Expr:
    """x > 0; shape ``alpha``, scale ``beta`` (PyMC's parametrisation)."""
    x, alpha, beta = Expr.wrap(x), Expr.wrap(alpha), Expr.wrap(beta)
    lz = log(x) - log(beta)
    return log(alpha) - log(beta) + (alpha - 1.0) * lz - exp(alpha * lz)


def uniform_lpdf(x, lower: float, upper: float) -> Expr:
    """lower < x < upper (a parameter with these bounds): the constant ``-log(upper - lower)``."""
    return Expr.const(-math.log(float(upper) - float(lower)))


def binomial_logit_lpmf(y, n, eta, log_binomial) -> Expr:
    """y successes of n trials (data), eta the logit, ``log_binomial`` = data holding log C(n, y)."""
    y, n, eta = Expr.wrap(y), Expr.wrap(n), Expr.wrap(eta)
    return y * eta - n * softplus(eta) + Expr.wrap(log_binomial)


def negative_binomial_log_lpmf(y, eta, phi, log_factorial) -> Expr:
    """y counts (data), eta = log mean, ``phi`` the over-dispersion (PyMC's ``alpha``: variance = mu + mu^2 / phi; number or expression),
    ``log_factorial`` = data holding lgamma(y + 1)."""
    y, eta = Expr.wrap(y), Expr.wrap(eta)
    if _is_number(phi):
        raise ValueError("negative_binomial_log_lpmf: phi is a parameter or data expression (lgamma(y + phi) depends on the observation)")
    phi = Expr.wrap(phi)
    lphi = log(phi)
    lse = softplus(eta - lphi) + lphi               # log(mu + phi)
    return lgamma(y + phi) - lgamma(phi) - Expr.wrap(log_factorial) + y * (eta - lse) + phi * (lphi - lse)


def dirichlet_lpdf(p, a) -> Expr:
    """Dirichlet log-density of a simplex-valued vector ``p`` (``Model.param(..., simplex=True)``): ``sum((a - 1) log p)`` plus the
    normalising constant when the concentration ``a`` is one number; with ``a`` a data vector on ``p``'s dimension the constant
    (which no parameter enters) is dropped."""
    p = Expr.wrap(p)
    if p.dim is None:
        raise ValueError("dirichlet_lpdf needs a vector")
    if isinstance(a, Expr):
        return ((a - 1.0) * log(p)).sum()
    a = float(a)
    n = p.dim.size
    const = math.lgamma(n * a) - n * math.lgamma(a) if n is not None else 0.0
    return ((a - 1.0) * log(p)).sum() + const if a != 1.0 else Expr.const(const)


def flat_lpdf(x) -> Expr:
    """An improper flat prior (``pm.Flat``): contributes nothing to the density — here for models that want to say so."""
    return Expr.const(0.0)


def bernoulli_logit_lpmf(y, eta) -> Expr:
    """y in {0, 1} (data), eta the logit."""
    y, eta = Expr.wrap(y), Expr.wrap(eta)
    return y * eta - softplus(eta)


def poisson_log_lpmf(y, eta, log_factorial) -> Expr:
    """y counts (data), eta = log rate, ``log_factorial`` = data holding lgamma(y + 1) (``Model.data`` of ``scipy.special.gammaln``)."""
    y, eta = Expr.wrap(y), Expr.wrap(eta)
    return y * eta - exp(eta) - Expr.wrap(log_factorial)


# --------------------------------------------------------------------------- reverse mode
_TWO_OVER_SQRT_PI = 2.0 / math.sqrt(math.pi)


def _topo(roots) -> list[Expr]:
    seen, order = set(), []
    stack = [(r, False) for r in roots]
    while stack:
        n, done = stack.pop()
        if done:
            order.append(n)
            continue
        if n.id in seen:
            continue
        seen.add(n.id)
        stack.append((n, True))
        for a in n.args:
            if a.id not in seen:
                stack.append((a, False))
    return order


def gradient(out: Expr, wrt: list[Expr]) -> list[Expr]:
    """d out / d wrt[k] as expressions (``out`` a scalar; every ``wrt[k]`` a parameter node).  The adjoint of a node lives on the
    node's dimension; a scalar-valued adjoint of a dimensioned node means the same value for every element."""
    if out.dim is not None:
        raise ValueError("the log-density must be a scalar: sum() the element-wise terms")
    order = _topo([out])
    adj: dict[int, Expr] = {out.id: Expr.const(1.0)}

    def reduce_to(e: Expr, from_dim, target: Expr) -> Expr:
        """the adjoint contribution ``e`` — one value per element of ``from_dim`` — folded onto ``target``'s dimension"""
        if target.dim is from_dim:
            return e
        if target.dim is None:
            return e.sum() if e.dim is not None else _dim_len(from_dim) * e
        raise AssertionError("dimension mismatch in the gradient")

    def acc(target: Expr, e: Expr):
        if e.is_const(0.0) or target.op in ("const", "data", "dimlen", "datacol"):
            return
        adj[target.id] = adj[target.id] + e if target.id in adj else e

    elem_adj: dict[int, dict[int, Expr]] = {}     # vector -> {element: adjoint of elem(vector, element)}
    for n in reversed(order):
        if n.id in elem_adj:      # every consumer of n has been visited: the adjoints of its extracted elements as one vector
            parts = elem_adj.pop(n.id)
            acc(n, stack([parts.get(c, Expr.const(0.0)) for c in range(n.dim.size)], n.dim))
        g = adj.get(n.id)
        if g is None or not n.args:
            continue
        if n.op == "elem":
            slot = elem_adj.setdefault(n.args[0].id, {})
            slot[n.payload] = slot[n.payload] + g if n.payload in slot else g
            continue
        if n.op == "stack":
            for c, a_ in enumerate(n.args):
                acc(a_, elem(g, c))
            continue
        d = n.dim
        a = n.args[0]
        b = n.args[1] if len(n.args) > 1 else None
        if n.op == "add":
            acc(a, reduce_to(g, d, a)); acc(b, reduce_to(g, d, b))
        elif n.op == "sub":
            acc(a, reduce_to(g, d, a)); acc(b, reduce_to(-g, d, b))
        elif n.op == "mul":
            acc(a, reduce_to(g * b, d, a)); acc(b, reduce_to(g * a, d, b))
        elif n.op == "div":
            acc(a, reduce_to(g / b, d, a)); acc(b, reduce_to(-(g * n) / b, d, b))
        elif n.op == "neg":
            acc(a, -g)
        elif n.op == "exp":
            acc(a, g * n)
        elif n.op == "log":
            acc(a, g / a)
        elif n.op == "log1p":
            acc(a, g / (1.0 + a))
        elif n.op == "sqrt":
            acc(a, g / (2.0 * n))
        elif n.op == "softplus":
            acc(a, g * sigmoid(a))
        elif n.op == "sigmoid":
            acc(a, g * (n * (1.0 - n)))
        elif n.op == "where_lt":
            acc(a, reduce_to(where_lt(d, n.payload, g, 0.0), d, a)); acc(b, reduce_to(where_lt(d, n.payload, 0.0, g), d, b))
        elif n.op in ("sel_gt", "sel_ge"):
            strict = n.op == "sel_gt"
            va, vb = n.args[1], n.args[2]
            acc(va, reduce_to(select(a, g, 0.0, strict), d, va)); acc(vb, reduce_to(select(a, 0.0, g, strict), d, vb))
        elif n.op == "tanh":
            acc(a, g * (1.0 - n * n))
        elif n.op == "expm1":
            acc(a, g * (n + 1.0))
        elif n.op == "erf":
            acc(a, g * (_TWO_OVER_SQRT_PI * exp(-(a * a))))
        elif n.op == "erfc":
            acc(a, g * (-_TWO_OVER_SQRT_PI * exp(-(a * a))))
        elif n.op == "sin":
            acc(a, g * cos(a))
        elif n.op == "cos":
            acc(a, -(g * sin(a)))
        elif n.op == "atan":
            acc(a, g / (1.0 + a * a))
        elif n.op == "lgamma":
            acc(a, g * digamma(a))
        elif n.op == "digamma":
            raise NotImplementedError("the derivative of digamma (trigamma) is not available")
        elif n.op == "abs":
            acc(a, g * sign(a))
        elif n.op == "sign":
            pass
        elif n.op == "pad":
            acc(a, trunc(g, a.dim) if g.dim is not None else g)
        elif n.op == "trunc":
            acc(a, pad(g, a.dim) if g.dim is not None else where_lt(a.dim, d.size, g, 0.0))
        elif n.op == "bcast":
            acc(a, reduce_to(g, d, a))
        elif n.op == "sum":
            acc(a, g)                      # the same scalar for every element
        elif n.op == "max":
            if n.payload != "constant":
                acc(a, select(a - n, g, 0.0, strict=False))
        elif n.op == "gather":
            acc(a, _segsum(g, n.payload))
        elif n.op == "segsum":
            acc(a, _bcast(g, n.payload.into)[n.payload])
        else:
            raise AssertionError(n.op)
    return [adj.get(w.id, Expr.const(0.0)) for w in wrt]


# --------------------------------------------------------------------------- numpy evaluation (expand step; the tests' checker)
_NP_UNARY = ("tanh", "expm1", "erf", "erfc", "sin", "cos", "atan", "lgamma", "digamma", "abs", "sign")


def _np_unary(op: str, a: np.ndarray) -> np.ndarray:
    if op in ("erf", "erfc", "lgamma", "digamma"):
        import scipy.special as sp

        return {"erf": sp.erf, "erfc": sp.erfc, "lgamma": sp.gammaln, "digamma": sp.digamma}[op](a)
    return {"tanh": np.tanh, "expm1": np.expm1, "sin": np.sin, "cos": np.cos, "atan": np.arctan, "abs": np.abs, "sign": np.sign}[op](a)


def evaluate(nodes: list[Expr], x: np.ndarray, data: dict[str, Any]) -> list[np.ndarray]:
    """Values of ``nodes`` for a block of positions ``x[N, D]``: scalars as ``[N]``, dimensioned nodes as ``[N, len]``."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    N = x.shape[0]
    val: dict[int, np.ndarray] = {}

    def dim_len(dim: Dim) -> int:
        return dim.len_py(data)

    def up(v, n):     # align a scalar [N] with a dimensioned operand [N, len]
        return v[:, None] if (n.dim is not None and v.ndim == 1) else v

    for n in _topo(nodes):
        a = val[n.args[0].id] if n.args else None
        b = val[n.args[1].id] if len(n.args) > 1 else None
        if n.op in ("add", "sub", "mul", "div", "where_lt"):
            a, b = up(a, n), up(b, n)
        with np.errstate(all="ignore"):
            if n.op == "const":
                v = np.full(N, n.payload)
            elif n.op == "sparam":
                v = x[:, n.payload]
            elif n.op == "vparam":
                off, nv = n.payload
                v = np.zeros((N, dim_len(n.dim)))
                v[:, :nv] = x[:, off:off + nv]
            elif n.op == "data":
                v = np.broadcast_to(np.asarray(data[n.payload], dtype=np.float64), (N, dim_len(n.dim)))
            elif n.op == "datacol":
                name, c, K = n.payload
                v = np.broadcast_to(np.asarray(data[name], dtype=np.float64).reshape(-1, K)[:, c], (N, dim_len(n.dim)))
            elif n.op == "elem":
                v = a[:, n.payload] if a.ndim == 2 else a
            elif n.op == "stack":
                v = np.stack([np.broadcast_to(val[x_.id], (N,)) for x_ in n.args], axis=1)
            elif n.op == "sdata":
                v = np.full(N, float(data[n.payload]))
            elif n.op == "dimlen":
                v = np.full(N, float(dim_len(n.payload)))
            elif n.op == "add":
                v = a + b
            elif n.op == "sub":
                v = a - b
            elif n.op == "mul":
                v = a * b
            elif n.op == "div":
                v = a / b
            elif n.op == "neg":
                v = -a
            elif n.op == "exp":
                v = np.exp(a)
            elif n.op == "log":
                v = np.log(a)
            elif n.op == "log1p":
                v = np.log1p(a)
            elif n.op == "sqrt":
                v = np.sqrt(a)
            elif n.op == "softplus":
                v = np.maximum(a, 0.0) + np.log1p(np.exp(-np.abs(a)))
            elif n.op == "sigmoid":
                v = 1.0 / (1.0 + np.exp(-a))
            elif n.op == "where_lt":
                L = dim_len(n.dim)
                v = np.where((np.arange(L) < n.payload)[None, :], np.broadcast_to(a, (N, L)), np.broadcast_to(b, (N, L)))
            elif n.op in ("sel_gt", "sel_ge"):
                c3 = [val[x_.id] for x_ in n.args]
                if n.dim is not None:
                    c3 = [np.broadcast_to(t[:, None] if t.ndim == 1 else t, (N, dim_len(n.dim))) for t in c3]
                v = np.where(c3[0] > 0 if n.op == "sel_gt" else c3[0] >= 0, c3[1], c3[2])
            elif n.op in _NP_UNARY:
                v = _np_unary(n.op, a)
            elif n.op == "pad":
                v = np.zeros((N, dim_len(n.dim)))
                v[:, :a.shape[1]] = a
            elif n.op == "trunc":
                v = a[:, :dim_len(n.dim)]
            elif n.op == "bcast":
                v = np.broadcast_to(a[:, None] if a.ndim == 1 else a, (N, dim_len(n.dim)))
            elif n.op == "sum":
                v = a.sum(axis=1)
            elif n.op == "max":
                v = a.max(axis=1)
            elif n.op == "gather":
                v = a[:, np.asarray(data[n.payload.name], dtype=np.int64)]
            elif n.op == "segsum":
                idx = np.asarray(data[n.payload.name], dtype=np.int64)
                v = np.zeros((N, dim_len(n.dim)))
                np.add.at(v, (slice(None), idx), a)
            else:
                raise AssertionError(n.op)
        val[n.id] = np.asarray(v, dtype=np.float64)
    return [val[n.id] for n in nodes]


# --------------------------------------------------------------------------- code generation
class _Out:
    """Where a generated function writes one of its outputs: what ``_Gen`` reads of a parameter node (``dim``, ``payload`` = offset
    or (offset, valid length)), for outputs that are not gradients — the rows of the generated expand function.  ``perm`` =
    (rows, columns) of a two-dimensional value that is stored transposed."""

    def __init__(self, dim, payload, perm=None):
        self.dim, self.payload, self.perm = dim, payload, perm


class _Gen:
    """logp + gradient -> the source of ``nphip_density``; or (``outputs``) a list of reported values -> ``nphip_expand``."""

    def __init__(self, model: "Model", logp: Expr, grads: list[Expr], waves: int = 1, profile: bool = False, outputs=None, fn_name: str = "nphip_density"):
        self.m = model
        self.threads = _WAVE * waves
        self.profile = profile      # cycle counters per section, added into data.prof__ (Model.profile)
        self.sections: list[str] = []
        self.spilled: list[Any] = []   # keys of `stored` that live in device memory (data.scratch__) instead of LDS: Model.compile decides
        self.logp = logp
        self.fn_name = fn_name
        self.params = model._params
        if outputs is not None:
            # (offset, expression, transposed-as) of every reported value: a row of the expand function's output
            self.out_scalar = [(_Out(None, off), e) for off, e, _ in outputs if e.dim is None]
            # (a raw parameter vector reports its free values only: a zero-sum / simplex parameter has one fewer than its dimension)
            self.out_vector = [(_Out(e.dim, (off, e.payload[1] if e.op == "vparam" else (e.dim.size if e.dim.size is not None else f"n_{e.dim.name}")), perm), e)
                               for off, e, perm in outputs if e.dim is not None]
        else:
            # gradient outputs: scalars by lane 0 at the end; vectors in a loop over their dimension
            self.out_scalar = [(p, g) for p, g in zip(self.params, grads) if p.dim is None]
            self.out_vector = [(p, _bcast(g, p.dim) if g.dim is None else g) for p, g in zip(self.params, grads) if p.dim is not None]
        roots = [logp] + [g for _, g in self.out_scalar] + [g for _, g in self.out_vector]
        self.order = _topo(roots)
        # names in the generated source count the nodes of THIS graph (node ids count every node ever made: the same model built
        # twice would print two different sources, and the library cache is keyed by the source)
        self.num: dict[int, int] = {n.id: k for k, n in enumerate(self.order)}
        self.level: dict[int, int] = {}
        for n in self.order:
            lv = max([self.level[a.id] for a in n.args], default=0)
            if n.op in ("sum", "max", "gather", "segsum") or (n.op in ("elem", "pad", "trunc") and n.args[0].op not in ("vparam", "data")):
                lv += 1
            self.level[n.id] = lv
        # what lives in per-chain LDS: sources of gathers (unless they are parameters or data, read in place), arguments of
        # segment sums (stored grouped by target) ...
        self.stored: dict[Any, tuple[str, Dim]] = {}
        for n in self.order:
            if n.op in ("gather", "pad", "trunc") and n.args[0].op not in ("vparam", "data"):
                self.stored[n.args[0].id] = ("plain", n.args[0].dim)
            elif n.op == "segsum":
                self.stored[("seg", n.args[0].id, n.payload.name)] = ("grouped", n.args[0].dim)
            elif n.op == "elem" and n.args[0].op not in ("vparam", "data", "stack"):
                self.stored[n.args[0].id] = ("plain", n.args[0].dim)     # a scalar read of one element of a computed vector
            elif n.op == "stack":
                self.stored[n.id] = ("scalars", n.dim)                   # written by the scalar code, element by element
        # ... and segment sums that more than one loop needs.  Element-wise values are recomputed in every loop that needs them
        # (a few operations on values that are read anyway); a segment sum is an inner loop over its range.
        evaluated: dict[int, set[int]] = {}     # segment sum -> the levels of the loops that evaluate it
        by_id = {n.id: n for n in self.order}
        for lv, roots in self.loop_roots().items():
            seen: set[int] = set()
            stack = list(roots)
            while stack:
                n = stack.pop()
                if n.dim is None or n.id in seen:
                    continue
                seen.add(n.id)
                if n.op == "stack" or (n.id in self.stored and self.level[n.id] < lv[1]):
                    continue               # read from LDS
                if n.op == "segsum":
                    evaluated.setdefault(n.id, set()).add(lv[1])
                    continue               # (its argument was stored by an earlier loop)
                if n.op in ("gather", "pad", "trunc"):
                    continue
                stack.extend(n.args)
        for nid, levels in evaluated.items():
            if len(levels) > 1:
                self.stored[nid] = ("plain", by_id[nid].dim)
        # Parameter vectors on a dimension short enough for ONE trip of its loops (size <= threads x unroll): every loop over the
        # dimension reads the same elements in the same lanes, so they are read once, before the first loop (a read of x[] at the
        # head of a loop is a round trip nothing else hides).
        self.hoisted: set[int] = set()
        reads: dict[int, set[tuple[int, int]]] = {}
        for key, roots in self.loop_roots().items():
            seen: set[int] = set()
            stack = list(roots)
            while stack:
                n = stack.pop()
                if n.dim is None or n.id in seen:
                    continue
                seen.add(n.id)
                if n.op == "vparam":
                    reads.setdefault(n.id, set()).add(key)
                if n.op in ("stack", "gather", "segsum", "pad", "trunc") or (n.id in self.stored and self.level[n.id] < key[1]):
                    continue
                stack.extend(n.args)
        for nid, loops in reads.items():
            d = by_id[nid].dim
            if len(loops) > 1 and d.size is not None and d.size <= self.threads * _UNROLL:
                self.hoisted.add(nid)

    def loop_roots(self) -> dict[tuple[int, int], list[Expr]]:
        """(id of the dimension, level) -> the nodes the loop has to produce: arguments of sums, stored values, gradient rows"""
        roots: dict[tuple[int, int], list[Expr]] = {}
        by_id = {n.id: n for n in self.order}
        for n in self.order:
            if n.op in ("sum", "max"):
                a = n.args[0]
                roots.setdefault((id(a.dim), self.level[a.id]), []).append(a)
        for key, (how, _) in self.stored.items():
            if how == "scalars":
                continue
            node = by_id[key[1] if isinstance(key, tuple) else key]
            roots.setdefault((id(node.dim), self.level[node.id]), []).append(node)
        for _, g in self.out_vector:
            roots.setdefault((id(g.dim), self.level[g.id]), []).append(g)
        return roots

    # ---- LDS layout
    def lds_layout(self):
        """[(key, dim)] in the order of allocation; the size of each entry is its dimension's length."""
        return list(self.stored.items())

    def source(self, emit_stage: bool = True) -> tuple[str, list]:
        m = self.m
        L: list[str] = []
        emit = L.append
        stage_src, shared_fields = m._stage_source()
        if emit_stage:
            emit(stage_src)
        if any(n.op == "digamma" for n in self.order):
            emit(_DIGAMMA_SOURCE)
        emit(f"__device__ double {self.fn_name}(const NphipData& data, int dim, const double* x, double* g, double* lds, const double* shared, int lane) {{")
        # dimension lengths, data pointers (shared LDS where staged, else global), LDS scratch
        for d in m._dims.values():
            emit(f"    const int n_{d.name} = {m._len_c(d)};")
        off_expr = "0"
        for name, kind, dim in shared_fields:
            n_c = _field_len_c(name, dim, m._matrix_cols.get(name, 1))
            if not m._staged:
                emit(f"    const {'double' if kind == 'double' else 'int'}* __restrict__ D_{name} = data.{name};")
            elif kind == "double":
                emit(f"    const auto D_{name} = NPHIP_LDS_CPTR(double, shared + ({off_expr}));")
                off_expr += f" + {n_c}"
            else:
                emit(f"    const auto D_{name} = NPHIP_LDS_CPTR(int, shared + ({off_expr}));")
                off_expr += f" + ({n_c} + 1) / 2"
        lds_off = "0"
        self.store_name: dict[Any, str] = {}
        if self.spilled:   # this chain's block of the device-memory scratch: the arrays that do not fit the LDS
            stride = " + ".join(f"(size_t)n_{self.stored[key][1].name}" for key in self.spilled)
            emit(f"    double* const scratch_ = (double*)data.scratch__ + (size_t)NPHIP_CHAIN_SLOT * ({stride});")
        g_off = "0"
        for k, (key, (_, dim)) in enumerate(self.stored.items()):
            self.store_name[key] = f"M{k}"
            if key in self.spilled:
                emit(f"    double* const M{k} = scratch_ + ({g_off});")
                g_off += f" + (size_t)n_{dim.name}"
            else:
                emit(f"    const auto M{k} = NPHIP_LDS_PTR(double, lds + ({lds_off}));")
                lds_off += f" + n_{dim.name}"
        self.spilled_names = {self.store_name[key] for key in self.spilled}
        by_id_ = {n.id: n for n in self.order}
        for nid in sorted(self.hoisted):
            n = by_id_[nid]
            off, nv = n.payload
            U = max(1, min(_UNROLL, -(-n.dim.size // self.threads)))
            for u in range(U):
                idx = f"(lane + {self.threads * u})"
                emit(f"    const double H{self.num[nid]}_{u} = ({idx} < {nv}) ? x[{off} + {idx}] : 0.0;")
        max_level = max(self.level.values(), default=0)
        done_scalar: set[int] = set()
        self.L = L
        if self.profile:   # eight evaluations, the last one timed: the first pays for instruction and data cache misses
            emit("    double result_ = 0.0;")
            emit("    for (int rep_ = 0; rep_ < 8; ++rep_) {")
            emit("    long long t_ = (long long)__builtin_readcyclecounter();")

        def mark(label: str):
            if self.profile:
                # (kept in registers until the evaluation is over: 1024 chains adding into one counter would be what is measured)
                emit('    __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");')
                emit(f"    const long long m{len(self.sections)}_ = (long long)__builtin_readcyclecounter();")
                emit("    __builtin_amdgcn_sched_barrier(0);")
                self.sections.append(label)

        def scalar_ready(n: Expr) -> bool:
            return all(a.dim is not None or a.id in done_scalar for a in n.args)

        for lv in range(max_level + 2):
            # scalars of this level (sums were produced by the loops of the level below)
            for n in self.order:
                if n.dim is None and n.op not in ("sum", "max") and self.level[n.id] == lv and n.id not in done_scalar:
                    assert scalar_ready(n), n
                    if n.op != "const":
                        emit(f"    const double s{self.num[n.id]} = {self.scalar_rhs(n)};")
                    done_scalar.add(n.id)
            stacks = [n for n in self.order if n.op == "stack" and self.level[n.id] == lv]
            if stacks:
                emit("    if (lane == 0) {")
                for n in stacks:
                    for c, a_ in enumerate(n.args):
                        emit(f"        {self.store_name[n.id]}[{c}] = {self.sref(a_)};")
                emit("    }")
                emit("    nphip_chain_barrier();")
            mark(f"scalars of level {lv}")
            # loops of this level, one per dimension that has something to produce here
            for d in m._dims.values():
                sums = [n for n in self.order if n.op in ("sum", "max") and n.args[0].dim is d and self.level[n.args[0].id] == lv]
                stores = []
                for key, (how, sd) in self.stored.items():
                    if sd is not d or how == "scalars":
                        continue
                    node_id = key[1] if isinstance(key, tuple) else key
                    if self.level[node_id] == lv:
                        stores.append((key, how))
                outs = [(p, g) for p, g in self.out_vector if p.dim is d and self.level[g.id] == lv]
                if not (sums or stores or outs):
                    continue
                self.loop(d, lv, sums, stores, outs, mark)
                for n in sums:
                    done_scalar.add(n.id)
        mark("tail scalars")
        emit("    if (lane == 0) {")
        for p, g in self.out_scalar:
            emit(f"        g[{p.payload}] = {self.sref(g)};")
        emit("    }")
        if self.profile:
            emit(f"    result_ = {self.sref(self.logp)};")
            emit("    if (lane == 0 && rep_ == 7) {")
            for k in range(len(self.sections)):
                emit(f"        atomicAdd((double*)data.prof__ + {k}, (double)(m{k}_ - {'t_' if k == 0 else f'm{k - 1}_'}));")
            emit("    }")
            emit("    nphip_chain_barrier();")
            emit("    }")
            emit("    return result_;")
        else:
            emit(f"    return {self.sref(self.logp)};")
        emit("}")
        return "\n".join(L), shared_fields

    # ---- scalars
    def sref(self, n: Expr) -> str:
        if n.op == "const":
            return _lit(n.payload)
        return f"s{self.num[n.id]}"

    def scalar_rhs(self, n: Expr) -> str:
        if n.op == "sparam":
            return f"x[{n.payload}]"
        if n.op == "sdata":
            return f"data.{n.payload}"
        if n.op == "dimlen":
            return f"(double)n_{n.payload.name}"
        if n.op == "elem":
            v, c = n.args[0], n.payload
            if v.op == "vparam":
                return f"x[{v.payload[0] + c}]"
            if v.op == "data":
                return f"D_{v.payload}[{c}]"
            return f"{self.store_name[v.id]}[{c}]"
        return _op_c(n.op, [self.sref(a) for a in n.args])

    def seg_cap(self, iname: str, u: int, T: int, U: int, n_acc: int) -> int:
        """how many elements of its range every lane reads up front in unrolled iteration ``u`` of a loop: the longest range that
        iteration meets in the model's data (at most 32, at most 64 values in flight)"""
        rows = self.m._data.get(iname + "__rows")
        if rows is None:
            return 0
        lens = np.diff(np.asarray(rows, dtype=np.int64))
        mine = lens[(np.arange(lens.size) % (T * U)) // T == u]
        longest = int(mine.max()) if mine.size else 0
        return 0 if longest < 3 else min(longest, 32, max(1, 64 // max(1, n_acc)))

    # ---- one wave-wide loop
    def loop(self, d: Dim, lv: int, sums, stores, outs, mark=lambda label: None):
        emit = self.L.append
        T = self.threads
        U = _UNROLL if d.size is None else max(1, min(_UNROLL, -(-d.size // T)))
        by_id = {n.id: n for n in self.order}
        emit(f"    // level {lv}, over {d.name}")
        for n in sums:
            emit(f"    double s{self.num[n.id]} = {'0.0' if n.op == 'sum' else '-INFINITY'};")
        emit(f"    for (int i0 = lane; i0 < n_{d.name}; i0 += {T * U}) {{")
        stages: list[list[str]] = [[], [], [], [], []]   # 0 direct reads, 1 dependent reads, 2 segment sums, 3 arithmetic, 4 stores / sums
        seg: dict[tuple, list[tuple[str, str]]] = {}     # (iteration, index) -> [(accumulator, array)]: one inner loop for all of them
        for u in range(U):
            memo: dict[int, str] = {}
            stages[0].append(f"        const int i_{u} = i0 + {T * u}, j_{u} = i_{u} < n_{d.name} ? i_{u} : 0;")

            def val(n: Expr, u=u, memo=memo) -> str:
                if n.dim is None:
                    return self.sref(n)
                if n.id in memo:
                    return memo[n.id]
                name = f"v{self.num[n.id]}_{u}"
                if n.id in self.stored and self.level[n.id] < lv:
                    stages[0].append(f"        const double {name} = {self.store_name[n.id]}[j_{u}];")
                elif n.op == "vparam" and n.id in self.hoisted:
                    name = f"H{self.num[n.id]}_{u}"       # (read before the first loop; lanes past the end hold 0 and are never used)
                elif n.op == "vparam":
                    off, nv = n.payload
                    full = d.size is not None and nv == d.size
                    stages[0].append(f"        const double {name} = " + (f"x[{off} + j_{u}];" if full else f"(j_{u} < {nv}) ? x[{off} + j_{u}] : 0.0;"))
                elif n.op == "data":
                    stages[0].append(f"        const double {name} = D_{n.payload}[j_{u}];")
                elif n.op == "datacol":
                    mname, c, K = n.payload
                    stages[0].append(f"        const double {name} = D_{mname}[j_{u} * {K} + {c}];")
                elif n.op == "stack":
                    stages[0].append(f"        const double {name} = {self.store_name[n.id]}[j_{u}];")
                elif n.op == "gather":
                    src, index = n.args[0], n.payload
                    iname = f"k{index.name}_{u}"
                    if ("idx", index.name) not in memo:
                        stages[0].append(f"        const int {iname} = D_{index.name}[j_{u}];")
                        memo[("idx", index.name)] = iname
                    if src.op == "vparam":
                        off, nv = src.payload
                        full = src.dim.size is not None and nv == src.dim.size
                        stages[1].append(f"        const double {name} = " + (f"x[{off} + {iname}];" if full else f"({iname} < {nv}) ? x[{off} + {iname}] : 0.0;"))
                    elif src.op == "data":
                        stages[1].append(f"        const double {name} = D_{src.payload}[{iname}];")
                    else:
                        stages[1].append(f"        const double {name} = {self.store_name[src.id]}[{iname}];")
                elif n.op in ("pad", "trunc"):
                    # the same element of a value on a shorter (pad: zero beyond its end) or longer (trunc) dimension
                    src = n.args[0]
                    inside = f"(j_{u} < n_{src.dim.name})" if n.op == "pad" else None
                    if src.op == "vparam":
                        off, nv = src.payload
                        stages[0].append(f"        const double {name} = (j_{u} < {nv}) ? x[{off} + j_{u}] : 0.0;")
                    else:
                        arr = f"D_{src.payload}" if src.op == "data" else self.store_name[src.id]
                        stages[0].append(f"        const double {name} = " + (f"{inside} ? {arr}[j_{u}] : 0.0;" if inside else f"{arr}[j_{u}];"))
                elif n.op == "segsum":
                    index = n.payload
                    arr = self.store_name[("seg", n.args[0].id, index.name)]
                    if (u, index.name) not in seg:
                        seg[(u, index.name)] = []
                        stages[0].append(f"        const int r0{index.name}_{u} = D_{index.name}__rows[j_{u}], "
                                         f"r1{index.name}_{u} = (i_{u} < n_{d.name}) ? D_{index.name}__rows[j_{u} + 1] : r0{index.name}_{u};")
                    seg[(u, index.name)].append((name, arr))
                elif n.op == "bcast":
                    name = self.sref(n.args[0])
                elif n.op == "where_lt":
                    a, b = val(n.args[0]), val(n.args[1])
                    stages[3].append(f"        const double {name} = (j_{u} < {n.payload}) ? {a} : {b};")
                else:
                    args = [val(a) for a in n.args]
                    stages[3].append(f"        const double {name} = {_op_c(n.op, args)};")
                memo[n.id] = name
                return name

            guard = f"if (i_{u} < n_{d.name}) "
            for key, how in stores:
                node = by_id[key[1] if isinstance(key, tuple) else key]
                v = val(node)
                if how == "grouped":
                    index_name = key[2]
                    pname = f"p{index_name}_{u}"
                    if ("pos", index_name) not in memo:
                        stages[0].append(f"        const int {pname} = D_{index_name}__pos[j_{u}];")
                        memo[("pos", index_name)] = pname
                    stages[4].append(f"        {guard}{self.store_name[key]}[{pname}] = {v};")
                else:
                    stages[4].append(f"        {guard}{self.store_name[key]}[i_{u}] = {v};")
            for n in sums:
                acc_ = f"s{self.num[n.id]}"
                stages[4].append(f"        {guard}{acc_} += {val(n.args[0])};" if n.op == "sum" else f"        {guard}{acc_} = fmax({acc_}, {val(n.args[0])});")
            for p, gexpr in outs:
                off, nv = p.payload
                perm = getattr(p, "perm", None)
                at = f"i_{u}" if perm is None else f"((i_{u} % {perm[1]}) * {perm[0]} + i_{u} / {perm[1]})"   # (row-major [rows][cols] stored as [cols][rows])
                stages[4].append(f"        if (i_{u} < {nv}) g[{off} + {at}] = {val(gexpr)};")
        # all segment sums of one index in one inner loop per unrolled iteration (walking the ranges of several iterations side by
        # side was measured: the extra compares cost more than the overlapped reads save — config 3: 44.7 -> 40.7 M leapfrogs/s)
        # The first `cap` elements of every range are read at once and added under a comparison (x + 0.0 == x: the same additions in
        # the same order as the loop, which stays for what a range has beyond `cap`): a loop with a trip count per lane pays an LDS
        # round trip per four elements and up to three lone iterations at its end.  `cap` is the longest range THIS iteration of
        # the loop meets in the data the model is compiled with (other data: still exact, the rest loop takes what is longer).
        # Reads past the end of a range are other ranges' elements or, at the end of the array, whatever follows in the workgroup's
        # LDS — never used; arrays in device memory keep the plain loop.  (config 3, same box: profiles/r5_generated_density_loop_ab.txt)
        for (u, iname), accs in seg.items():
            stages[2].append("        double " + ", ".join(f"{a} = 0.0" for a, _ in accs) + ";")
            cap = self.seg_cap(iname, u, T, U, len(accs)) if _SEG_BATCH and all(arr not in self.spilled_names for _, arr in accs) else 0
            r0, r1 = f"r0{iname}_{u}", f"r1{iname}_{u}"
            if cap:
                stages[2].append(f"        const int n{iname}_{u} = {r1} - {r0};")
                for t in range(cap):
                    stages[2].append("        const double " + ", ".join(f"{a}_k{t} = {arr}[{r0} + {t}]" for a, arr in accs) + ";")
                for t in range(cap):
                    stages[2].append("        " + " ".join(f"{a} += ({t} < n{iname}_{u}) ? {a}_k{t} : 0.0;" for a, _ in accs))
            stages[2].append("#pragma unroll 4")
            stages[2].append(f"        for (int k = {r0}{f' + {cap}' if cap else ''}; k < {r1}; ++k) {{ " + " ".join(f"{a} += {arr}[k];" for a, arr in accs) + " }")
        for st in stages:
            for line in st:
                emit(line)
        emit("    }")
        mark(f"loop over {d.name}, level {lv}")
        # the loop's sums over the wave, several at a time
        for n in sums:
            if n.op == "max":
                emit(f"    s{self.num[n.id]} = nphip_chain_max(s{self.num[n.id]});")
        ids = [f"s{self.num[n.id]}" for n in sums if n.op == "sum"]
        for k in range(0, len(ids), 4):
            grp = ids[k:k + 4]
            if len(grp) == 1:
                emit(f"    {grp[0]} = nphip_chain_sum({grp[0]});")
            else:
                emit(f"    nphip_chain_sum{len(grp)}({', '.join(grp)});")
        if stores:
            emit("    nphip_chain_barrier();")
        if ids or stores:
            mark(f"{len(ids)} sums" + (" + barrier" if stores else "") + f" after {d.name}, level {lv}")


def _field_len_c(name: str, dim: Dim, cols: int = 1) -> str:
    """C expression of the length of a data field (the ranges of a grouping have one more entry than the target dimension; a
    matrix has ``cols`` values per element)"""
    if cols != 1:
        return f"(n_{dim.name} * {cols})"
    return f"(n_{dim.name} + 1)" if name.endswith("__rows") else f"n_{dim.name}"


def _lit(v: float) -> str:
    if math.isinf(v):
        return "INFINITY" if v > 0 else "-INFINITY"
    if math.isnan(v):
        return "NAN"
    return f"({float(v).hex()})" if v < 0 else float(v).hex()


def _op_c(op: str, a: list[str]) -> str:
    if op == "add":
        return f"{a[0]} + {a[1]}"
    if op == "sub":
        return f"{a[0]} - {a[1]}"
    if op == "mul":
        return f"{a[0]} * {a[1]}"
    if op == "div":
        return f"{a[0]} / {a[1]}"
    if op == "neg":
        return f"-{a[0]}"
    if op in ("exp", "log", "log1p", "sqrt"):
        return f"{op}({a[0]})"
    if op == "softplus":
        return f"(fmax({a[0]}, 0.0) + log1p(exp(-fabs({a[0]}))))"
    if op == "sigmoid":
        return f"(1.0 / (1.0 + exp(-{a[0]})))"
    if op in ("tanh", "expm1", "erf", "erfc", "sin", "cos", "atan", "lgamma"):
        return f"{op}({a[0]})"
    if op == "digamma":
        return f"nphip_digamma({a[0]})"
    if op == "abs":
        return f"fabs({a[0]})"
    if op == "sign":
        return f"(double)(({a[0]} > 0.0) - ({a[0]} < 0.0))"
    if op == "sel_gt":
        return f"(({a[0]} > 0.0) ? {a[1]} : {a[2]})"
    if op == "sel_ge":
        return f"(({a[0]} >= 0.0) ? {a[1]} : {a[2]})"
    raise AssertionError(op)


_DIGAMMA_SOURCE = """#ifndef NPHIP_DIGAMMA_DEFINED
#define NPHIP_DIGAMMA_DEFINED
// psi(x): reflection for x <= 0, the recurrence psi(x) = psi(x + 1) - 1/x up to 10, then the asymptotic series (to 1e-16)
static __device__ double nphip_digamma(double x) {
    double r = 0.0;
    if (x <= 0.0) {
        if (x == floor(x)) return NAN;
        r = -3.14159265358979323846 / tan(3.14159265358979323846 * x);
        x = 1.0 - x;
    }
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
}
#endif
"""


# --------------------------------------------------------------------------- the model
class Model:
    """Collects parameters, data and log-density terms; ``compile()`` returns the sampler-ready model."""

    #: LDS a workgroup may spend on staged data (bytes); beyond it the density reads its data from global memory (L2)
    STAGE_LIMIT = 48 * 1024

    def __init__(self):
        self._dims: dict[str, Dim] = {}
        self._params: list[Expr] = []          # sparam / vparam nodes in the order of the flat vector
        self._param_names: list[str] = []
        self._n_dim = 0
        self._data: dict[str, Any] = {}
        self._data_fields: list[tuple[str, str, Dim | None]] = []   # (name, "double" | "int", dim) in declaration order
        self._indices: dict[str, Index] = {}
        self._matrix_cols: dict[str, int] = {}       # matrix data: values per row
        self._terms: list[Expr] = []
        self._det: list[tuple[str, Expr]] = []
        self._det_dims: dict[str, tuple[str, ...]] = {}       # reported values on a product of two dimensions: the order of the axes in the trace
        self._products: dict[str, tuple[Dim, Dim, Index, Index]] = {}   # product dimension -> (rows, columns, index to rows, index to columns)
        self._unconstrained: dict[str, tuple[str, int, int]] = {}       # parameter -> (name of its unconstrained value, offset, free values)
        self._transforms: dict[str, tuple] = {}      # parameter -> (kind, lower, upper, shape): what initial_point inverts
        self._initvals: dict[str, Any] = {}          # parameter -> the constrained value its chains start around (PyMC's `initval`)
        self._staged = True
        self._specialize = True     # lengths of data dimensions are compile-time constants of the generated source (compile(specialize=...))

    # ---- declarations
    def dim(self, name: str, size: int | None = None) -> Dim:
        d = self._dims.get(name)
        if d is None:
            if size is None:
                raise ValueError(f"dimension {name!r} is not known yet: give its size")
            d = self._dims[name] = Dim(name, int(size))
        elif size is not None and d.size is not None and d.size != int(size):
            raise ValueError(f"dimension {name!r} has size {d.size}, not {size}")
        return d

    def _data_dim(self, name: str, array_name: str, n: int, div: int = 1) -> Dim:
        d = self._dims.get(name)
        if d is None:
            d = self._dims[name] = Dim(name, None, runtime_len=array_name, runtime_div=div)   # its length is that of its first data array
        elif d.size is not None and d.size != n:
            raise ValueError(f"data on dimension {name!r} must have length {d.size}")
        elif d.size is None and d.len_py(self._data) != n:
            raise ValueError(f"data on dimension {name!r} must have length {d.len_py(self._data)}")
        return d

    def matrix(self, name: str, values, dim: str, cols: str) -> Matrix:
        """A float data matrix with one row per element of ``dim`` and one column per element of ``cols`` (a fixed-size dimension:
        the coefficients'): ``X @ beta`` is the linear predictor.  ``with_data`` can replace it (same number of columns)."""
        self._check_new_data(name)
        a = np.ascontiguousarray(values, dtype=np.float64)
        if a.ndim != 2:
            raise ValueError("a matrix is two-dimensional")
        cd = self.dim(cols, a.shape[1])
        if cd.size is None or cd.size != a.shape[1]:
            raise ValueError(f"matrix {name!r} needs {cd.size} columns")
        self._data[name] = a.reshape(-1)
        self._matrix_cols[name] = int(a.shape[1])
        d = self._data_dim(dim, name, a.shape[0], div=int(a.shape[1]))
        self._data_fields.append((name, "double", d))
        return Matrix(name, d, cd)

    def _constrain(self, raw: Expr, lower, upper) -> tuple[Expr, Expr | None]:
        """(value, log-Jacobian) of the default transforms: log (``lower`` only), logit-interval (both bounds)"""
        if lower is None and upper is None:
            return raw, None
        if upper is None:
            return (exp(raw) + lower if lower != 0.0 else exp(raw)), raw
        if lower is None:
            return upper - exp(raw), raw
        if not upper > lower:
            raise ValueError("upper must exceed lower")
        width = float(upper) - float(lower)
        # value = lower + width sigmoid(raw);  log |d value / d raw| = log width - softplus(raw) - softplus(-raw)
        return sigmoid(raw) * width + lower, math.log(width) - softplus(raw) - softplus(-raw)

    def product(self, rows: str, cols: str) -> Dim:
        """The dimension of the (rows x cols) values of a two-dimensional quantity, row-major — with the two index arrays that
        take an element to its row and to its column, so that sums along an axis and broadcasts are the IR's segment sums and
        gathers (``reduce`` / ``broadcast``)."""
        name = f"{rows}_x_{cols}"
        if name in self._products:
            return self._dims[name]
        r, c = self.dim(rows), self.dim(cols)
        if r.size is None or c.size is None:
            raise ValueError("a product needs two dimensions of fixed size")
        self.dim(name, r.size * c.size)
        e = np.arange(r.size * c.size)
        to_r = self.index(f"{name}_row", e // c.size, dim=name, into=rows)
        to_c = self.index(f"{name}_col", e % c.size, dim=name, into=cols)
        self._products[name] = (r, c, to_r, to_c)
        return self._dims[name]

    def reduce(self, expr: Expr, over: str) -> Expr:
        """Sum of a two-dimensional value along the axis ``over``: a vector on the other axis."""
        r, c, to_r, to_c = self._product_of(expr)
        if over == r.name:
            return _segsum(expr, to_c)
        if over == c.name:
            return _segsum(expr, to_r)
        raise ValueError(f"{over!r} is not an axis of this value")

    def broadcast(self, vec: Expr, rows: str, cols: str) -> Expr:
        """A vector on one of the two axes, repeated along the other: a value on the product dimension."""
        self.product(rows, cols)
        r, c, to_r, to_c = self._products[f"{rows}_x_{cols}"]
        if vec.dim is r:
            return vec[to_r]
        if vec.dim is c:
            return vec[to_c]
        raise ValueError("broadcast: the vector lives on neither axis")

    def _product_of(self, expr: Expr):
        if expr.dim is None or expr.dim.name not in self._products:
            raise ValueError("not a two-dimensional value (Model.param(..., dims=(rows, cols)))")
        return self._products[expr.dim.name]

    def param(self, name: str, dim: str | None = None, size: int | None = None, lower: float | None = None, upper: float | None = None,
              zero_sum: bool = False, simplex: bool = False, dims: tuple[str, str] | None = None, initval=None, ordered: bool = False) -> Expr:
        """A free parameter.  Scalar, or a vector over ``dim``.  ``lower`` / ``upper``: PyMC's default transforms — ``lower + exp(raw)``,
        ``upper - exp(raw)``, or ``lower + (upper - lower) sigmoid(raw)`` with both — with the log-Jacobian added to the density;
        ``zero_sum``: the vector sums to zero (``size - 1`` free values, PyMC's isometric ZeroSumTransform — no Jacobian term);
        ``simplex``: positive and sums to one (``size - 1`` free values: the softmax of the zero-sum extension of the free values —
        PyMC's SimplexTransform, the default transform of ``pm.Dirichlet`` — with its log-Jacobian).
        ``ordered``: an increasing vector (PyMC's ``ordered`` transform: the cut points of an ordinal regression) — the first element
        free, every further one the one before plus ``exp(raw)``, with the log-Jacobian ``sum(raw[1:])``; up to 64 elements (the prefix
        sums are a gather of the (i, j <= i) pairs and a segment sum: n (n + 1) / 2 terms).
        ``dims=(rows, cols)``: a two-dimensional parameter, a value on ``product(rows, cols)`` (row-major); with ``zero_sum`` every
        COLUMN sums to zero along ``rows`` (``pmd.ZeroSumNormal(core_dims=(rows,), dims=(rows, cols))``: ``(rows - 1) x cols`` free values)."""
        if name in self._param_names:
            raise ValueError(f"parameter {name!r} is defined twice")
        # ``initval``: the (constrained) value the chains start around — PyMC's support point of the variable; without one the
        # unconstrained value 0 (the support point of a Normal(0, .), a ZeroSumNormal, a uniform Dirichlet, a HalfNormal(1) ...)
        self._transforms[name] = ("zero_sum" if zero_sum else "simplex" if simplex else "ordered" if ordered else "bounds", lower, upper, dims)
        if initval is not None:
            self._initvals[name] = initval
        if dims is not None:
            return self._param_2d(name, dims, lower, upper, zero_sum, simplex)
        self._param_names.append(name)
        if dim is None:
            raw = Expr("sparam", (), None, self._n_dim)
            self._params.append(raw)
            self._unconstrained[name] = (name + ("_log__" if (lower is not None and upper is None) else "_interval__" if lower is not None else
                                                 "_upper__" if upper is not None else ""), self._n_dim, 1)
            self._n_dim += 1
            if zero_sum or simplex or ordered:
                raise ValueError("zero_sum / simplex / ordered need a vector parameter")
            value, jac = self._constrain(raw, lower, upper)
            if jac is not None:
                self._terms.append(jac)
            self._det.append((name, value))
            return value
        d = self.dim(dim, size)
        if d.size is None:
            raise ValueError("a parameter's dimension needs a fixed size")
        if zero_sum and simplex:
            raise ValueError("zero_sum and simplex exclude each other")
        n_free = d.size - 1 if (zero_sum or simplex) else d.size
        raw = Expr("vparam", (), d, (self._n_dim, n_free))
        self._params.append(raw)
        if ordered and (zero_sum or simplex or lower is not None or upper is not None or d.size > 64):
            raise ValueError("ordered excludes zero_sum, simplex and bounds, and takes up to 64 elements")
        self._unconstrained[name] = (name + ("_zerosum__" if zero_sum else "_simplex__" if simplex else "_ordered__" if ordered else "_log__" if (lower is not None and upper is None) else
                                             "_interval__" if lower is not None else "_upper__" if upper is not None else ""), self._n_dim, n_free)
        self._n_dim += n_free
        if zero_sum:
            if lower is not None or upper is not None:
                raise ValueError("zero_sum and bounds exclude each other")
            n = d.size
            s = raw.sum()            # (the padding element reads as 0)
            value = where_lt(d, n - 1, raw - s * (1.0 / (math.sqrt(n) + n)), -s * (1.0 / math.sqrt(n)))
        elif simplex:
            if lower is not None or upper is not None:
                raise ValueError("simplex and bounds exclude each other")
            # y = (raw, -sum raw);  value = softmax(y) = softmax(y + sum raw): z = raw + s on the free elements, 0 on the last
            # (shifted by the largest exponent, as PyMC's SimplexTransform does: exp cannot overflow during a warm-up excursion; the
            #  shift cancels in value and in the Jacobian, so no gradient flows through it)
            n = d.size
            s = raw.sum()
            z = where_lt(d, n - 1, raw + s, 0.0)
            shift = z.max(constant=True)
            e = exp(z - shift)
            total = e.sum()
            value = e / total
            self._terms.append(math.log(n) + n * s - n * (shift + log(total)))     # log |det d value[:n-1] / d raw|
        elif ordered:
            n = d.size
            step = where_lt(d, 1, raw, exp(raw))          # the first element itself, then the positive increments
            if n == 1:
                value = raw
            else:
                pairs = self.dim(f"{name}_pairs", n * (n + 1) // 2)
                src = np.array([j for i in range(n) for j in range(i + 1)])
                dst = np.array([i for i in range(n) for j in range(i + 1)])
                i_src = self.index(f"{name}_pair_src", src, dim=pairs.name, into=d.name)
                i_dst = self.index(f"{name}_pair_dst", dst, dim=pairs.name, into=d.name)
                value = _segsum(step[i_src], i_dst)
                self._terms.append(raw.sum() - elem(raw, 0))   # log |det d value / d raw| = sum of raw[1:]
        else:
            value, jac = self._constrain(raw, lower, upper)
            if jac is not None:
                self._terms.append(jac.sum())
        self._det.append((name, value))
        return value

    def _param_2d(self, name, dims, lower, upper, zero_sum, simplex):
        rows, cols = dims
        if simplex:
            raise ValueError("a simplex parameter is one-dimensional")
        prod = self.product(rows, cols)
        r, c, to_r, to_c = self._products[prod.name]
        self._param_names.append(name)
        n_free = (r.size - 1) * c.size if zero_sum else r.size * c.size
        raw = Expr("vparam", (), prod, (self._n_dim, n_free))     # (row-major: the free rows come first, the padding row reads as 0)
        self._params.append(raw)
        self._unconstrained[name] = (name + ("_zerosum__" if zero_sum else "_log__" if (lower is not None and upper is None) else
                                             "_interval__" if lower is not None else "_upper__" if upper is not None else ""), self._n_dim, n_free)
        self._n_dim += n_free
        if zero_sum:
            if lower is not None or upper is not None:
                raise ValueError("zero_sum and bounds exclude each other")
            n = r.size
            s = _segsum(raw, to_c)                   # per column: the sum of its free values
            sb = s[to_c]
            value = where_lt(prod, (n - 1) * c.size, raw - sb * (1.0 / (math.sqrt(n) + n)), -sb * (1.0 / math.sqrt(n)))
        else:
            value, jac = self._constrain(raw, lower, upper)
            if jac is not None:
                self._terms.append(jac.sum())
        self._det.append((name, value))
        return value

    def data(self, name: str, values, dim: str | None = None) -> Expr:
        """Observed / shared data: a float array over ``dim`` or (``dim=None``) one float.  ``with_data`` can replace it."""
        self._check_new_data(name)
        if dim is None:
            self._data[name] = float(values)
            self._data_fields.append((name, "sdouble", None))
            return Expr("sdata", (), None, name)
        a = np.ascontiguousarray(values, dtype=np.float64)
        if a.ndim != 1:
            raise ValueError("data arrays are one-dimensional")
        self._data[name] = a
        d = self._data_dim(dim, name, a.size)
        self._data_fields.append((name, "double", d))
        return Expr("data", (), d, name)

    def index(self, name: str, values, dim: str, into: str) -> Index:
        """Integer data: ``values[i]`` is the element of dimension ``into`` that element ``i`` of ``dim`` belongs to."""
        self._check_new_data(name)
        a = np.ascontiguousarray(values, dtype=np.int32)
        into_d = self.dim(into)
        if into_d.size is None:
            raise ValueError("the target of an index needs a fixed size")
        self._data[name] = a
        d = self._data_dim(dim, name, a.size)
        ix = Index(name, d, into_d)
        self._indices[name] = ix
        self._data_fields.append((name, "int", d))
        self._data_fields.append((name + "__pos", "int", d))
        self._data_fields.append((name + "__rows", "int", into_d))
        self._derive(name)
        return ix

    def _check_new_data(self, name):
        if not name.isidentifier() or "__" in name:
            raise ValueError(f"data name {name!r} must be an identifier without double underscores")
        if name in self._data:
            raise ValueError(f"data {name!r} is defined twice")

    def _derive(self, name, data=None):
        """the grouping of an index array: where each element sits when the elements are grouped by target, and the groups' ranges"""
        data = self._data if data is None else data
        ix = self._indices[name]
        a = np.asarray(data[name], dtype=np.int64)
        n = ix.into.size
        if a.size and (a.min() < 0 or a.max() >= n):
            raise ValueError(f"index {name!r} must lie in [0, {n})")
        counts = np.bincount(a, minlength=n)
        rows = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        order = np.argsort(a, kind="stable")
        pos = np.empty(a.size, dtype=np.int32)
        pos[order] = np.arange(a.size, dtype=np.int32)
        data[name + "__pos"] = pos
        data[name + "__rows"] = rows

    def unconstrain(self, name: str, value) -> np.ndarray:
        """The unconstrained coordinates of a parameter at a constrained value (the forward direction of PyMC's transforms:
        ``LogTransform``, ``IntervalTransform``, ``ZeroSumTransform.forward`` = ``extend_axis_rev``, ``SimplexTransform.forward``)."""
        kind, lower, upper, dims = self._transforms[name]
        _, off, n_free = self._unconstrained[name]
        v = np.asarray(value, dtype=np.float64)
        if kind == "zero_sum":
            if dims is not None:
                r, c = self._dims[dims[0]].size, self._dims[dims[1]].size
                v = np.broadcast_to(v, (r, c))
                norm = (-v[-1:] * math.sqrt(r)) / (math.sqrt(r) + r)           # per column (the zero-sum axis is `rows`)
                return (v[:-1] + norm).reshape(-1)
            n = n_free + 1
            v = np.broadcast_to(v, (n,))
            return v[:-1] + (-v[-1] * math.sqrt(n)) / (math.sqrt(n) + n)
        if kind == "simplex":
            v = np.broadcast_to(v, (n_free + 1,))
            lv = np.log(v)
            return lv[:-1] - lv.sum() / (n_free + 1)
        if kind == "ordered":
            v = np.broadcast_to(v, (n_free,)).astype(np.float64)
            with np.errstate(all="ignore"):       # (a value that does not increase gives NaN: initial_point refuses it by name)
                return np.concatenate([v[:1], np.log(np.diff(v))])
        v = np.broadcast_to(v, (n_free,)) if v.ndim <= 1 else v.reshape(-1)
        if lower is None and upper is None:
            return v.astype(np.float64)
        with np.errstate(all="ignore"):       # (a value outside the support gives NaN: initial_point refuses it by name)
            if upper is None:
                return np.log(v - lower)
            if lower is None:
                return np.log(upper - v)
            t = (v - lower) / (upper - lower)
            return np.log(t) - np.log1p(-t)

    def initial_point(self, overrides: dict[str, Any] | None = None) -> np.ndarray:
        """The model's support point on the unconstrained scale: every parameter at its ``initval`` (``overrides`` first; PyMC's
        ``initial_points`` / ``overrides`` of ``make_initial_point_fn``), 0 where none was given."""
        x = np.zeros(self._n_dim)
        vals = {**self._initvals, **(overrides or {})}
        for name, value in vals.items():
            if name not in self._unconstrained:
                raise ValueError(f"initial point for an unknown parameter {name!r}")
            _, off, n_free = self._unconstrained[name]
            u = np.asarray(self.unconstrain(name, value), dtype=np.float64).reshape(-1)
            if u.size != n_free or not np.all(np.isfinite(u)):
                raise ValueError(f"the initial value of {name!r} does not lie inside its support")
            x[off:off + n_free] = u
        return x

    def jittered_init(self, overrides: dict[str, Any] | None = None, jitter_rvs=None):
        """PyMC's default initial points (reference compile_pymc.py:593-602): the support point plus U(-1, 1) on the unconstrained
        coordinates of ``jitter_rvs`` (None: every parameter)."""
        from nutpie_amd.density import JitteredInit

        mask = np.zeros(self._n_dim)
        for name, (_, off, n_free) in self._unconstrained.items():
            if jitter_rvs is None or name in jitter_rvs:
                mask[off:off + n_free] = 1.0
        return JitteredInit(center=self.initial_point(overrides), jitter=mask)

    def add_logp(self, term) -> None:
        term = Expr.wrap(term)
        if term.dim is not None:
            raise ValueError("a log-density term must be a scalar: sum() it")
        self._terms.append(term)

    def deterministic(self, name: str, expr, dims: tuple[str, ...] | None = None) -> None:
        """A value to report in the trace besides the parameters (the reference's expanded variables).  ``dims``: for a
        two-dimensional value, the order of its axes in the trace (``(cols, rows)`` reports it transposed)."""
        if any(n == name for n, _ in self._det):
            raise ValueError(f"{name!r} is reported already")
        expr = Expr.wrap(expr)
        if dims is not None:
            r, c, _, _ = self._product_of(expr)
            if tuple(dims) not in ((r.name, c.name), (c.name, r.name)):
                raise ValueError(f"dims must be a permutation of ({r.name!r}, {c.name!r})")
            self._det_dims[name] = tuple(dims)
        self._det.append((name, expr))

    # ---- compilation
    @property
    def n_dim(self):
        return self._n_dim

    def logp_expr(self) -> Expr:
        if not self._terms:
            raise ValueError("the model has no log-density terms")
        total = self._terms[0]
        for t in self._terms[1:]:
            total = total + t
        return total

    def _len_c(self, d: Dim) -> str:
        """C expression of a dimension's length: a constant, also for a data dimension when the source is specialised to the data it is
        compiled with (constant loop bounds and LDS offsets: config 3 +9 % — with_data compiles again when a length changes)"""
        return str(d.len_py(self._data)) if (self._specialize and d.size is None) else d.len_c()

    def _stage_source(self):
        """``nphip_density_stage`` + the order of the staged fields: doubles first, then the 32-bit integers (each rounded up to a
        whole double)."""
        fields = [f for f in self._data_fields if f[1] == "double"] + [f for f in self._data_fields if f[1] == "int"]
        if not self._staged or not fields:
            return "", fields
        L = ["__device__ void nphip_density_stage(const NphipData& data, double* shared, int thread, int n_threads) {"]
        for d in self._dims.values():
            L.append(f"    const int n_{d.name} = {self._len_c(d)};")
        L.append("    double* at = shared;")
        for name, kind, dim in fields:
            n = _field_len_c(name, dim, self._matrix_cols.get(name, 1))
            if kind == "double":
                L.append(f"    for (int i = thread; i < {n}; i += n_threads) at[i] = data.{name}[i];")
                L.append(f"    at += {n};")
            else:
                L.append(f"    for (int i = thread; i < {n}; i += n_threads) ((int*)at)[i] = data.{name}[i];")
                L.append(f"    at += ({n} + 1) / 2;")
        L.append("}")
        return "\n".join(L), fields

    def _shared_doubles(self, data) -> int:
        if not self._staged:
            return 0
        total = 0
        for name, kind, _ in self._data_fields:
            if kind == "double":
                total += len(data[name])
            elif kind == "int":
                total += (len(data[name]) + 1) // 2
        return total

    def generate(self, waves_per_chain: int = 1):
        """-> (source, generator): the HIP source of the density and the object that knows its LDS layout."""
        logp = self.logp_expr()
        grads = gradient(logp, self._params)
        self._staged = 8 * self._shared_doubles_unconditional(self._data) <= self.STAGE_LIMIT
        gen = _Gen(self, logp, grads, waves_per_chain)
        src, _ = gen.source()
        return src, gen

    def _shared_doubles_unconditional(self, data):
        was, self._staged = self._staged, True
        try:
            return self._shared_doubles(data)
        finally:
            self._staged = was

    def profile(self, n_chains: int = 1024, waves_per_chain: int = 1, scale: float = 0.3, seed: int = 0):
        """Where an evaluation spends its cycles: the density is generated with a cycle counter (``s_memtime``) after every
        section, compiled, and run once for ``n_chains`` positions (1024 single-wave chains = one wave per SIMD, the situation
        of the sampler).  Returns ``[(section, mean cycles per evaluation)]``."""
        logp = self.logp_expr()
        grads = gradient(logp, self._params)
        had = "prof__" in self._data
        if not had:
            self._data["prof__"] = np.zeros(64)
            self._data_fields.append(("prof__", "raw", None))
        try:
            self._staged = 8 * self._shared_doubles_unconditional(self._data) <= self.STAGE_LIMIT
            gen = _Gen(self, logp, grads, waves_per_chain, profile=True)
            src, _ = gen.source()
            compiled = self._finish(src, gen, waves_per_chain=waves_per_chain)
            x = scale * np.random.default_rng(seed).normal(size=(n_chains, self._n_dim))
            compiled.logp_and_grad(x)                      # (first call: loads the library, pages in)
            _, _, data = compiled.logp_and_grad(x, return_data=True)
            cycles = data["prof__"][:len(gen.sections)] / n_chains
            return list(zip(gen.sections, cycles.tolist()))
        finally:
            if not had:
                del self._data["prof__"]
                self._data_fields.pop()

    #: LDS of a CU (bytes) and what the resident kernel itself takes of it: per wave of a workgroup a control block and a ring of
    #: (p, rho) summaries (4 KB per chunk of 128 dimensions), per chain the position and gradient rows — nphip_model_jit_density
    LDS_BYTES = 160 * 1024

    def _lds_budget(self, waves: int, data=None) -> int:
        """bytes of LDS one chain may use for its scratch with ``waves`` waves per chain"""
        nch = (self._n_dim + 127) // 128
        nv = (nch + waves - 1) // waves
        cpb, nwaves, ld = (4, 4, nv * 128) if waves == 1 else (1, waves, nv * 128 * waves)
        fixed = nwaves * 1200 + 1024 + nwaves * nv * 4096 + 64 + 16 * waves * nv + cpb * 2 * ld * 8
        shared = 8 * self._shared_doubles(self._data if data is None else data)
        return (self.LDS_BYTES - 2048 - fixed - shared) // cpb

    def _lds_budget_for(self, waves: int, data) -> int:
        return self._lds_budget(waves, data)

    def _lds_fits(self, gen, waves: int) -> bool:
        per_chain = 8 * sum(d.len_py(self._data) for key, (_, d) in gen.stored.items() if key not in gen.spilled)
        return per_chain <= self._lds_budget(waves)

    def _spill(self, gen, waves: int) -> None:
        """move the largest stored arrays to device memory until the rest fits the LDS"""
        by_size = sorted(gen.stored.items(), key=lambda kv: -kv[1][1].len_py(self._data))
        for key, (how, d) in by_size:
            if self._lds_fits(gen, waves):
                return
            if how != "scalars":       # (vectors assembled by the scalar code are tiny and written by one lane: they stay)
                gen.spilled.append(key)

    def compile(self, *, init="uniform", resident: bool = True, coords=None, dims=None, waves_per_chain: int | None = None,
                expanded_names=None, expanded_shapes=None, expand_fn=None, specialize: bool = True, var_names=None):
        """-> :class:`SymbolicModel` (a :class:`nutpie_amd.density.DensitySourceModel`).  ``waves_per_chain`` (1, 2, 4): wavefronts
        that evaluate one chain's density together — more than one pays with fewer chains than the device has SIMDs (1024), and a
        workgroup then holds ONE chain instead of four, i.e. a quarter of the per-chain LDS: the default (None) is one wave per
        chain unless the model's scratch (one double per observation and gathered value) only fits with more.

        ``var_names`` (reference ``compile_pymc.py:821-822``, ``tests/test_pymc.py:425-468``): which of the variables that are COMPUTED from a
        draw go to the trace — the deterministics and the constrained values of transformed parameters.  ``None``: all of them; ``[]``: none;
        ``["b"]``: only ``b``.  The free variables themselves (untransformed parameters under their own name, transformed ones under their
        unconstrained name) are always stored, as in the reference; a name that the model does not report is an error."""
        from nutpie_amd.density import from_density_source
        import copy

        # (the compiled model keeps a snapshot of the front-end: what is decided here — staging, specialisation, the LDS plan — belongs
        #  to THIS compilation, and the same Model may be compiled again with other options)
        self = copy.copy(self)
        self._data = dict(self._data)
        # ``specialize``: the lengths of the data arrays become constants of the source (with_data compiles again — a cached library
        # or a few seconds of hipcc — when one of them changes); False: one library for data of any length
        self._specialize = bool(specialize)
        self._compile_kw = dict(init=init, resident=resident, coords=coords, dims=dims, waves_per_chain=waves_per_chain, expanded_names=expanded_names,
                                expanded_shapes=expanded_shapes, expand_fn=expand_fn, specialize=specialize, var_names=var_names)
        if var_names is not None:
            wanted = {getattr(v, "name", v) for v in var_names}
            unknown = wanted - {n for n, _ in self._det}
            if unknown:
                raise KeyError(f"var_names: the model reports no variable named {sorted(unknown)}")
            transformed = {p for p, (u, _, _) in self._unconstrained.items() if u != p}
            free = set(self._param_names) - transformed          # (reported under their own name: the free variables of the reference's trace)
            self._det = [(n, e) for n, e in self._det if n in wanted or n in free]
        logp = self.logp_expr()
        grads = gradient(logp, self._params)
        self._staged = 8 * self._shared_doubles_unconditional(self._data) <= self.STAGE_LIMIT
        if waves_per_chain is None:
            for waves_per_chain in (1, 2, 4):
                gen = _Gen(self, logp, grads, waves_per_chain)
                if self._lds_fits(gen, waves_per_chain):
                    break
            else:
                # not even one chain per workgroup holds it: four chains per workgroup again, the large arrays in device memory
                waves_per_chain = 1
                gen = _Gen(self, logp, grads, 1)
        else:
            gen = _Gen(self, logp, grads, waves_per_chain)
        self._spill(gen, waves_per_chain)
        src, _ = gen.source()
        if isinstance(init, str) and init == "support_point":
            init = self.jittered_init()
        return self._finish(src, gen, init=init, resident=resident, coords=coords, dims=dims, waves_per_chain=waves_per_chain,
                            expanded_names=expanded_names, expanded_shapes=expanded_shapes, expand_fn=expand_fn)

    def _finish(self, src, gen, *, init="uniform", resident=True, coords=None, dims=None, waves_per_chain=1, expanded_names=None, expanded_shapes=None,
                expand_fn=None):
        from nutpie_amd.density import from_density_source

        dim_of = {k: d for k, d in self._dims.items()}
        stored_dims = [d for key, (_, d) in gen.stored.items() if key not in gen.spilled]
        spilled_dims = [gen.stored[key][1] for key in gen.spilled]

        def dim_len(d: Dim, data):
            return d.len_py(data)

        def lds_per_chain(data):
            return sum(dim_len(d, data) for d in stored_dims)

        fields = list(self._data_fields)
        staged = self._staged and any(k in ("double", "int") for _, k, _ in fields)

        def lds_shared(data):
            if not staged:
                return 0
            return sum(len(data[n]) if k == "double" else (len(data[n]) + 1) // 2 for n, k, _ in fields if k in ("double", "int"))

        det = list(self._det)
        # the unconstrained values of transformed parameters, under PyMC's names (b_log__, c_zerosum__, d_simplex__ ...): they go to
        # the trace's `unconstrained_posterior` group (reference sample.py:147-160; nutpie.sample(store_unconstrained=True))
        by_name = dict(zip(self._param_names, self._params))
        raw_names = []
        for pname, (uname, _, n_free) in self._unconstrained.items():
            if uname != pname:
                det.append((uname, by_name[pname]))
                raw_names.append(uname)
        names = [n for n, _ in det]
        nodes = [e for _, e in det]

        def shape_of(name, e, data=None):
            data = self._data if data is None else data
            if e.dim is None:
                return ()
            if e.op == "vparam" and name in raw_names:
                if e.dim.name in self._products:
                    r, c, _, _ = self._products[e.dim.name]
                    return (e.payload[1] // c.size, c.size)
                return (e.payload[1],)
            if e.dim.name in self._products:
                r, c, _, _ = self._products[e.dim.name]
                return (c.size, r.size) if self._det_dims.get(name) == (c.name, r.name) else (r.size, c.size)
            return (e.dim.len_py(data),)

        def dims_of(name, e):
            if e.op == "vparam" and name in raw_names:
                # an unconstrained value keeps its dimension unless the transform took an element away (tests/test_pymc.py:332-346)
                if e.dim.name in self._products:
                    r, c, _, _ = self._products[e.dim.name]
                    return ((r.name if e.payload[1] == r.size * c.size else name + "_dim"), c.name)
                return (e.dim.name if e.payload[1] == e.dim.size else name + "_dim",)
            if e.dim.name in self._products:
                r, c, _, _ = self._products[e.dim.name]
                return self._det_dims.get(name, (r.name, c.name))
            return (e.dim.name,)

        shapes = [shape_of(n, e) for n, e in det]
        # (what with_data needs to re-derive for new data: the shapes of values on a data dimension, and whether the LDS plan —
        #  staging, waves per chain, what was moved to device memory: decided here, from THIS data — still holds)
        self._shapes_for = lambda data: [shape_of(n, e, data) for n, e in det]
        self._plan_check = lambda data: (8 * lds_per_chain(data), self._lds_budget_for(waves_per_chain, data))
        auto_dims = {n: dims_of(n, e) for n, e in det if e.dim is not None}
        used_dims = {dn for ds in auto_dims.values() for dn in ds}
        auto_coords = {d.name: np.arange(d.size) for d in dim_of.values() if d.size is not None and d.name in used_dims and d.name not in self._products}
        transposed = {n for n, e in det if n not in raw_names and e.dim is not None and e.dim.name in self._products
                      and self._det_dims.get(n) == tuple(reversed(dims_of("", e)))}

        def expand(positions, /, **data):   # (positional-only: a data array may be called `x`)
            vals = evaluate(nodes, positions, data)
            out = {}
            for (n, e), v in zip(det, vals):
                if e.op == "vparam" and n in raw_names:
                    v = np.asarray(v)[:, :e.payload[1]]
                    out[n] = v.reshape(v.shape[0], *shape_of(n, e))
                    continue
                if e.dim is not None and e.dim.name in self._products:
                    r, c, _, _ = self._products[e.dim.name]
                    v = np.asarray(v).reshape(-1, r.size, c.size)
                    v = v.transpose(0, 2, 1) if n in transposed else v
                out[n] = v
            return out

        # the same values as a generated device function (the reference's expand step, compile_pymc.py:601-666, on the GPU): one row
        # of the flat expanded vector per draw, variables in declaration order, each in its trace layout
        offs, off = [], 0
        for (n, e), shp in zip(det, shapes):
            perm = None
            if n in transposed:
                r, c, _, _ = self._products[e.dim.name]
                perm = (r.size, c.size)
            offs.append((off, e, perm))
            off += int(np.prod(shp, dtype=np.int64)) if shp else 1
        fixed = all(e.dim is None or e.dim.size is not None for e in nodes)   # (values on a data dimension change size with the data: host expand)
        # A caller that names the reported variables itself (a traced torch density: one variable `x`, or the user's expand function):
        # with a host function there is no generated expand; without one the generated rows are re-read under the caller's shapes
        if expanded_names is not None:
            out_names, out_shapes = list(expanded_names), [tuple(int(v) for v in shp) for shp in expanded_shapes]
            if expand_fn is None and sum(int(np.prod(shp, dtype=np.int64)) if shp else 1 for shp in out_shapes) != off:
                raise ValueError("expanded_shapes must hold as many values as the model reports")
            if expand_fn is not None:
                fixed = False
        else:
            out_names, out_shapes = names, shapes
        egen = _Gen(self, Expr.const(0.0), [], waves_per_chain, outputs=offs, fn_name="nphip_expand") if (fixed and det) else None
        expand_src = ""
        if egen is not None:
            egen.spilled = []
            expand_src, _ = egen.source(emit_stage=False)
            e_dims = [d for key, (_, d) in egen.stored.items()]

            def expand_lds(data):
                return sum(dim_len(d, data) for d in e_dims)
        else:
            def expand_lds(data):
                return 0

        def scratch_per_chain(data):
            return sum(dim_len(d, data) for d in spilled_dims)

        if expanded_names is None:
            expand_host = expand
        elif expand_fn is not None:
            expand_host = expand_fn
        else:
            def expand_host(positions, /, **data):   # the model's own values in declaration order, re-read under the caller's names
                vals = expand(positions, **data)
                N = np.atleast_2d(np.asarray(positions)).shape[0]
                flat = np.concatenate([np.asarray(vals[n]).reshape(N, -1) for n in names], axis=1)
                out, at = {}, 0
                for n, shp in zip(out_names, out_shapes):
                    k = int(np.prod(shp, dtype=np.int64)) if shp else 1
                    out[n] = flat[:, at:at + k].reshape(N, *shp)
                    at += k
                return out

        data0 = {k: v for k, v in self._data.items() if k != "scratch__"}
        src = src + ("\n" + expand_src if expand_src else "")
        base = from_density_source(self._n_dim, src, data0, lds_doubles_per_chain=lds_per_chain, lds_doubles_shared=lds_shared if staged else 0,
                                   expand_lds_doubles=expand_lds if expand_src else 0,
                                   scratch_doubles_per_chain=scratch_per_chain if spilled_dims else 0,
                                   expand_fn=expand_host, expanded_names=out_names, expanded_shapes=out_shapes,
                                   coords={**(auto_coords if expanded_names is None else {}), **(coords or {})},
                                   dims={**(auto_dims if expanded_names is None else {}), **(dims or {})}, init=init, resident=resident, waves_per_chain=waves_per_chain,
                                   reparameterized_names=raw_names if expanded_names is None else None)
        import dataclasses

        return _symbolic_model_class()(**{f.name: getattr(base, f.name) for f in dataclasses.fields(base)}, _front=self)


_SYMBOLIC_MODEL = None


def _symbolic_model_class():
    global _SYMBOLIC_MODEL
    if _SYMBOLIC_MODEL is not None:
        return _SYMBOLIC_MODEL
    import dataclasses

    from nutpie_amd.density import DensitySourceModel

    @dataclasses.dataclass(frozen=True)
    class SymbolicModel(DensitySourceModel):
        """The compiled form of a :class:`Model`: a :class:`~nutpie_amd.density.DensitySourceModel` that also knows how the
        index arrays' groupings are derived, so that ``with_data`` can replace them."""

        _front: Any = None

        def with_data(self, **updates):
            f = self._front
            for k in updates:
                if k not in f._data or "__" in k:
                    raise ValueError(f"Unknown data variable: {k}")
            new = {**self._data}
            for k, v in updates.items():
                kind = next(kd for n, kd, _ in f._data_fields if n == k)
                if k in f._matrix_cols:
                    a = np.ascontiguousarray(v, dtype=np.float64)
                    if a.ndim != 2 or a.shape[1] != f._matrix_cols[k]:
                        raise ValueError(f"matrix {k!r} must keep its {f._matrix_cols[k]} columns")
                    new[k] = a.reshape(-1)
                    continue
                new[k] = float(v) if kind == "sdouble" else np.ascontiguousarray(v, dtype=np.int32 if kind == "int" else np.float64)
            for name in f._indices:
                f._derive(name, new)
            # arrays of one dimension must keep a common length
            for d in f._dims.values():
                if d.size is None:
                    n = d.len_py(new)
                    for name, kind, dd in f._data_fields:
                        rows = len(new[name]) // f._matrix_cols.get(name, 1) if kind in ("double", "int") else 0
                        if dd is d and kind in ("double", "int") and not name.endswith("__rows") and rows != n:
                            raise ValueError(f"data on dimension {d.name!r} must share one length ({name!r} has {rows}, {d.runtime_len!r} has {n})")
            # a source specialised to the lengths of its data: other lengths are another source (compiled now, or found in the cache)
            if f._specialize and any(d.size is None and d.len_py(new) != d.len_py(f._data) for d in f._dims.values()):
                import copy

                g = copy.copy(f)
                g._data = {k: v for k, v in new.items() if k != "scratch__"}
                return g.compile(**f._compile_kw)
            # what compile() decided from the ORIGINAL data: re-derive the shapes of values on a data dimension, and refuse — here,
            # not after the run — data that no longer fit the LDS plan (staging, waves per chain, spilled arrays)
            need, budget = f._plan_check(new)
            if need > budget:
                raise ValueError(f"the new data do not fit the LDS plan compile() made for the original data ({need} bytes of scratch per chain needed, "
                                 f"{max(budget, 0)} left beside the staged data): compile the model again with data of this size")
            return dataclasses.replace(self, _data=new, _shapes=[tuple(s_) for s_ in f._shapes_for(new)])

        def logp_and_grad_numpy(self, x):
            """The same graph evaluated with numpy (host; for checking a model, not for sampling)."""
            logp = f_logp = self._front.logp_expr()
            grads = gradient(f_logp, self._front._params)
            vals = evaluate([logp] + grads, x, self._data)
            N = vals[0].shape[0]
            g = np.zeros((N, self._front._n_dim))
            for p, v in zip(self._front._params, vals[1:]):
                if p.dim is None:
                    g[:, p.payload] = v
                else:
                    off, nv = p.payload
                    g[:, off:off + nv] = (v if v.ndim == 2 else np.broadcast_to(v[:, None], (N, nv)))[:, :nv]
            return vals[0], g

    _SYMBOLIC_MODEL = SymbolicModel
    return SymbolicModel


def __getattr__(name):
    if name == "SymbolicModel":
        return _symbolic_model_class()
    raise AttributeError(name)
