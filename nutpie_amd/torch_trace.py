"""A torch log-density -> the front-end's expression graph -> generated HIP density -> the model's own resident kernel.

The reference compiles a PyMC model by asking PyTensor for the graph of ``logp`` and lowering that graph with a compiler
(``python/nutpie/compile_pymc.py:668-871``; with the JAX / "Python callable" route the compiled thing is a traced array function,
``:410-520``).  PyTensor's ``mode="PYTORCH"`` linker turns the same graph into a torch function, and users of
``nutpie.compiled_pyfunc.from_pyfunc`` write torch or numpy functions directly.  On the GPU engine a torch function evaluated
eagerly costs one kernel launch per operation per gradient evaluation (config 3: 323 us per leapfrog, 0.35 M leapfrogs/s); this
module instead TRACES the function once (``torch.fx`` / ``make_fx``: the ATen operations of the forward pass only) and maps the
operations onto :mod:`nutpie_amd.symbolic`'s IR, which differentiates the graph symbolically and prints it as a HIP device
function that the engine calls in the middle of its register-resident leaf.

    model = nutpie_amd.from_torch_density(ndim, logp, compile=True)      # logp(x[chains, ndim]) -> [chains]
    model = trace(logp, ndim).compile()                                  # the same, spelled out

Representation.  Every traced tensor is a flat value in row-major order: ``_Sym(expr, shape)`` with ``expr`` living on the IR
dimension ``n<numel>`` (all tensors with the same number of elements share a loop range), or with a scalar ``expr`` when all of
its elements are equal (a broadcast scalar).  Operations that only MOVE data (``select``, ``slice``, ``index``, ``expand``,
``permute``, ``view`` ...) are applied to ``arange(numel).reshape(shape)``: the result is, for every output element, the flat
source element it reads — an identity (nothing to do), a prefix (``trunc``), a single element (``elem``) or an integer data array
(``Gather``; its transpose in the gradient is the IR's ``SegSum``).  Reductions along an axis are ``SegSum`` through the map from
input to output element; ``cat`` is ``pad`` / ``Gather`` under ``where_lt``.  Tensors the function closes over (the model's data)
are constants of the trace: operations on constants alone are evaluated eagerly, and a constant that meets a traced value
becomes a data array of the compiled model.

The position vector.  ``x[a:b]`` and ``x[k]`` (also through a leading batch axis of length one) become the model's parameters
when the pieces do not overlap; any other use of ``x`` makes the whole vector one parameter and the pieces gathers from it.

Anything that cannot be mapped raises :class:`UnsupportedTorchOp` naming the operation; ``from_torch_density(compile="auto")``
then falls back to the eager device callback and says why.
"""

from __future__ import annotations

import math
import operator
from typing import Any, Callable

import numpy as np

from nutpie_amd import symbolic as S
from nutpie_amd.symbolic import Expr

__all__ = ["trace", "traced_model", "UnsupportedTorchOp", "TraceResult"]


class UnsupportedTorchOp(NotImplementedError):
    """The traced function uses an operation (or a form of one) the IR has no counterpart for."""


class _NeedWholeVector(Exception):
    """the position vector is used in a way the partition into parameters cannot express: trace again with x as ONE parameter"""


def _numel(shape) -> int:
    return int(np.prod(shape, dtype=np.int64)) if len(shape) else 1


class _Sym:
    """a traced float tensor: ``expr`` on the dimension of ``numel(shape)`` elements (row-major), or a scalar for all of them"""

    __slots__ = ("expr", "shape")

    def __init__(self, expr: Expr, shape):
        self.expr, self.shape = expr, tuple(int(v) for v in shape)


class _Bool:
    """a traced boolean tensor: a tree of comparisons of traced values (``gt`` / ``ge`` of an expression against zero, ``not``,
    ``and``, ``or``)"""

    __slots__ = ("tree", "shape")

    def __init__(self, tree, shape):
        self.tree, self.shape = tree, tuple(int(v) for v in shape)


class _X:
    """the position vector (or a view of it that keeps all of its elements in order)"""

    __slots__ = ("shape",)

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in shape)


class TraceResult:
    """What :func:`trace` returns: the front-end model (``.model``, a :class:`nutpie_amd.symbolic.Model`) and ``.compile()``."""

    def __init__(self, model: S.Model, n_dim: int, whole: bool, n_ops: int):
        self.model, self.n_dim, self.whole_vector, self.n_ops = model, n_dim, whole, n_ops

    def compile(self, **kw):
        """-> the sampler-ready model (:class:`nutpie_amd.symbolic.SymbolicModel`); reports ONE variable ``x`` of shape ``(n_dim,)``
        unless ``expand_fn`` / ``expanded_names`` / ``expanded_shapes`` say otherwise (as :func:`nutpie_amd.from_torchfunc`)."""
        kw.setdefault("expanded_names", ["x"])
        kw.setdefault("expanded_shapes", [(self.n_dim,)])
        return self.model.compile(**kw)


# ----------------------------------------------------------------------------------------------------------------- the interpreter
class _Interp:
    def __init__(self, n_dim: int, whole: bool, data_names: list[str]):
        import torch

        self.torch = torch
        self.m = S.Model()
        self.n_dim = int(n_dim)
        self.whole = whole
        self.pieces: dict[tuple[int, int, bool], Expr] = {}      # (start, stop, scalar) -> parameter node
        self.x_expr: Expr | None = None
        self.n_const = 0
        self.n_index = 0
        self.const_cache: dict[tuple, Any] = {}
        self.index_cache: dict[tuple, S.Index] = {}
        self.data_names = list(data_names)
        self.named: dict[int, str] = {}         # id(tensor of a shared-data placeholder) -> its name in the model's data
        self._alive: list = []                  # (the tensors those ids belong to: an id is only unique while its object lives)
        if whole:
            d = self.dim(self.n_dim)
            self.x_expr = Expr("vparam", (), d, (0, self.n_dim)) if self.n_dim > 1 else Expr("sparam", (), None, 0)

    # ---- dimensions, data, indices
    def dim(self, n: int) -> S.Dim:
        return self.m.dim(f"n{int(n)}", int(n))

    def data(self, t, name: str | None = None) -> Expr:
        """a constant float tensor as a data array of the model (on the dimension of its number of elements)"""
        a = np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float64).reshape(-1)
        if a.size == 1 or (a.size > 1 and np.all(a == a[0])):
            return Expr.const(float(a[0]))           # (also +-inf: `where(inside, logp, -inf)` of a bounds check)
        if not np.all(np.isfinite(a)):
            raise UnsupportedTorchOp("a constant array with non-finite entries meets a traced value")
        key = (a.size, a.tobytes())
        hit = self.const_cache.get(key)
        if hit is not None:
            return hit
        if name is None:
            name = self.fresh_name(t)
        d = self.dim(a.size)
        e = self.m.data(name, a, dim=d.name)
        self.const_cache[key] = e
        return e

    def fresh_name(self, t=None) -> str:
        """the name of a data array: a shared-data placeholder keeps its keyword, anything else is numbered"""
        name = self.named.get(id(t))
        if name is not None and name not in self.m._data and name.isidentifier() and "__" not in name:
            return name
        name = f"c{self.n_const}"
        self.n_const += 1
        return name

    def index(self, values: np.ndarray, n_from: int, n_into: int) -> S.Index:
        a = np.ascontiguousarray(values, dtype=np.int32).reshape(-1)
        assert a.size == n_from
        key = (n_from, n_into, a.tobytes())
        hit = self.index_cache.get(key)
        if hit is None:
            hit = self.m.index(f"i{self.n_index}", a, dim=self.dim(n_from).name, into=self.dim(n_into).name)
            self.n_index += 1
            self.index_cache[key] = hit
        return hit

    # ---- the position vector
    def piece(self, start: int, stop: int, scalar: bool):
        if self.whole:
            raise AssertionError
        start, stop = int(start), int(stop)
        key = (start, stop, scalar)
        e = self.pieces.get(key)
        if e is None:
            for (a, b, _), _e in self.pieces.items():
                if a < stop and start < b and (a, b) != (start, stop):
                    raise _NeedWholeVector(f"x[{start}:{stop}] overlaps x[{a}:{b}]")
            if stop - start == 1:
                e = Expr("sparam", (), None, start)
                key = (start, stop, True)
            else:
                e = Expr("vparam", (), self.dim(stop - start), (start, stop - start))
            self.pieces[key] = e
        return _Sym(e, () if scalar else (stop - start,))

    def x_as_sym(self, xv: _X) -> _Sym:
        if not self.whole:
            raise _NeedWholeVector("x is used as a whole")
        return _Sym(self.x_expr, xv.shape)

    # ---- values
    def is_const(self, v) -> bool:
        return not isinstance(v, (_Sym, _Bool, _X))

    def sym(self, v, shape=None) -> _Sym:
        """any value as a traced float tensor"""
        torch = self.torch
        if isinstance(v, _Sym):
            return v
        if isinstance(v, _X):
            return self.x_as_sym(v)
        if isinstance(v, _Bool):
            return _Sym(self.where_tree(v.tree, Expr.const(1.0), Expr.const(0.0)), v.shape)
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bool:
                v = v.to(torch.float64)
            if v.dtype.is_complex:
                raise UnsupportedTorchOp("complex tensors")
            f = v.to(torch.float64)
            if id(v) in self.named:
                self.named[id(f)] = self.named[id(v)]
                self._alive.append(f)
            return _Sym(self.data(f), tuple(v.shape))
        if isinstance(v, (int, float, bool, np.floating, np.integer)):
            return _Sym(Expr.const(float(v)), ())
        raise UnsupportedTorchOp(f"a value of type {type(v).__name__} in the traced function")

    def shape_of(self, v):
        if isinstance(v, (_Sym, _Bool, _X)):
            return v.shape
        if isinstance(v, self.torch.Tensor):
            return tuple(v.shape)
        return ()

    # ---- data movement: apply `f` to the map of flat source elements
    def move(self, v, f: Callable) -> Any:
        torch = self.torch
        if self.is_const(v):
            return f(v)
        if isinstance(v, _X):
            v = self.x_as_sym(v)
        shape = v.shape
        n = _numel(shape)
        src = torch.arange(n, dtype=torch.int64).reshape(shape)
        out = f(src)
        if isinstance(out, (tuple, list)):
            return [self._moved(v, o, n) for o in out]
        return self._moved(v, out, n)

    def _moved(self, v, out, n):
        flat = out.reshape(-1).numpy()
        oshape = tuple(out.shape)
        if isinstance(v, _Bool):
            return _Bool(self._move_tree(v.tree, flat, n, oshape), oshape)
        return _Sym(self._move_expr(v.expr, flat, n), oshape)

    def _move_expr(self, e: Expr, flat: np.ndarray, n: int) -> Expr:
        k = flat.size
        if e.dim is None:
            return e                                              # all elements are equal
        if k == 1:
            return S.elem(e, int(flat[0]))
        if k == n and np.array_equal(flat, np.arange(n)):
            return e
        if k < n and np.array_equal(flat, np.arange(k)):
            return S.trunc(e, self.dim(k))
        if flat.min() < 0:
            raise UnsupportedTorchOp("an index outside the tensor")
        return e[self.index(flat, k, n)]

    def _move_tree(self, t, flat, n, oshape):
        if t[0] in ("gt", "ge"):
            return (t[0], self._move_expr(t[1], flat, n))
        if t[0] == "not":
            return ("not", self._move_tree(t[1], flat, n, oshape))
        return (t[0], self._move_tree(t[1], flat, n, oshape), self._move_tree(t[2], flat, n, oshape))

    def broadcast(self, v: _Sym, shape) -> Expr:
        """the expression of ``v`` broadcast to ``shape`` (a scalar expression stays one)"""
        if v.expr.dim is None or _numel(v.shape) == _numel(shape):
            return v.expr
        torch = self.torch
        return self.move(v, lambda t: torch.broadcast_to(t, shape)).expr

    def binary(self, a, b, f: Callable[[Expr, Expr], Expr]) -> _Sym:
        a, b = self.sym(a), self.sym(b)
        shape = tuple(self.torch.broadcast_shapes(a.shape, b.shape))
        return _Sym(f(self.broadcast(a, shape), self.broadcast(b, shape)), shape)

    def unary(self, a, f: Callable[[Expr], Expr]) -> _Sym:
        a = self.sym(a)
        return _Sym(f(a.expr), a.shape)

    # ---- booleans
    def compare(self, a, b, kind: str) -> _Bool:
        a, b = self.sym(a), self.sym(b)
        shape = tuple(self.torch.broadcast_shapes(a.shape, b.shape))
        ea, eb = self.broadcast(a, shape), self.broadcast(b, shape)
        if kind == "gt":
            return _Bool(("gt", ea - eb), shape)
        if kind == "lt":
            return _Bool(("gt", eb - ea), shape)
        if kind == "ge":
            return _Bool(("ge", ea - eb), shape)
        if kind == "le":
            return _Bool(("ge", eb - ea), shape)
        if kind == "eq":
            return _Bool(("and", ("ge", ea - eb), ("ge", eb - ea)), shape)
        if kind == "ne":
            return _Bool(("not", ("and", ("ge", ea - eb), ("ge", eb - ea))), shape)
        raise UnsupportedTorchOp(f"comparison {kind} of traced values")

    def as_bool(self, v) -> _Bool:
        torch = self.torch
        if isinstance(v, _Bool):
            return v
        if isinstance(v, torch.Tensor):
            f = v.to(torch.float64)
            return _Bool(("gt", self.data(f) - 0.5), tuple(v.shape))
        if isinstance(v, (bool, int)):
            return _Bool(("gt", Expr.const(1.0 if v else -1.0)), ())
        raise UnsupportedTorchOp("a traced float tensor used as a condition")

    def where_tree(self, t, a: Expr, b: Expr) -> Expr:
        if t[0] == "gt":
            return S.select(t[1], a, b, True)
        if t[0] == "ge":
            return S.select(t[1], a, b, False)
        if t[0] == "not":
            return self.where_tree(t[1], b, a)
        if t[0] == "and":
            return self.where_tree(t[1], self.where_tree(t[2], a, b), b)
        if t[0] == "or":
            return self.where_tree(t[1], a, self.where_tree(t[2], a, b))
        raise AssertionError(t[0])

    def where(self, c, a, b) -> _Sym:
        torch = self.torch
        a, b = self.sym(a), self.sym(b)
        shape = tuple(torch.broadcast_shapes(self.shape_of(c), a.shape, b.shape))
        ea, eb = self.broadcast(a, shape), self.broadcast(b, shape)
        if isinstance(c, torch.Tensor):       # a constant mask: all / none / a prefix / data
            mask = torch.broadcast_to(c.to(torch.bool), shape).reshape(-1).numpy()
            k = int(mask.sum())
            if k == mask.size:
                return _Sym(ea, shape)
            if k == 0:
                return _Sym(eb, shape)
            if mask[:k].all():
                return _Sym(S.where_lt(self.dim(mask.size), k, ea, eb), shape)
            if not mask[:mask.size - k].any():
                return _Sym(S.where_lt(self.dim(mask.size), mask.size - k, eb, ea), shape)
            return _Sym(S.select(self.data(mask.astype(np.float64)) - 0.5, ea, eb), shape)
        c = self.as_bool(c)
        tree = c.tree if _numel(c.shape) == _numel(shape) else self.move(c, lambda t: torch.broadcast_to(t, shape)).tree
        return _Sym(self.where_tree(tree, ea, eb), shape)

    # ---- reductions
    def sum(self, v, dims=None, keepdim=False) -> _Sym:
        torch = self.torch
        v = self.sym(v)
        shape = v.shape
        nd = len(shape)
        if dims is None or (isinstance(dims, (list, tuple)) and len(dims) == 0):
            dims = list(range(nd))
        if isinstance(dims, int):
            dims = [dims]
        dims = sorted({d % nd for d in dims}) if nd else []
        oshape_keep = tuple(1 if i in dims else s for i, s in enumerate(shape))
        oshape = oshape_keep if keepdim else tuple(s for i, s in enumerate(shape) if i not in dims)
        n_in, n_out = _numel(shape), _numel(oshape)
        if v.expr.dim is None:
            return _Sym(v.expr * float(n_in // max(n_out, 1)), oshape)
        if n_out == n_in:
            return _Sym(v.expr, oshape)
        if n_out == 1:
            return _Sym(v.expr.sum(), oshape)
        to = torch.broadcast_to(torch.arange(n_out, dtype=torch.int64).reshape(oshape_keep), shape).reshape(-1).numpy()
        return _Sym(S._segsum(v.expr, self.index(to, n_in, n_out)), oshape)

    # ---- cat
    def cat(self, parts: list, dim: int) -> Any:
        torch = self.torch
        parts = [p for p in parts if not (isinstance(p, torch.Tensor) and p.numel() == 0 and p.dim() == 1)]
        if all(self.is_const(p) for p in parts):
            return torch.cat(parts, dim)
        syms = [self.sym(p) for p in parts]
        # where every output element comes from: (piece, flat element of the piece)
        tags = [torch.arange(_numel(sv.shape), dtype=torch.int64).reshape(sv.shape) + (k << 40) for k, sv in enumerate(syms)]
        out = torch.cat(tags, dim)
        oshape = tuple(out.shape)
        flat = out.reshape(-1).numpy()
        which, at = flat >> 40, flat & ((1 << 40) - 1)
        n_out = flat.size
        d_out = self.dim(n_out)
        pos = [np.flatnonzero(which == k) for k in range(len(syms))]
        ks = [k for k in range(len(syms)) if pos[k].size]
        # a cat along the only non-trivial axis: every piece is one run of the output, in order -> nested `where_lt`; otherwise one
        # select per piece against a data mask
        runs = all(pos[k][-1] - pos[k][0] + 1 == pos[k].size for k in ks) and all(pos[a][-1] < pos[b][0] for a, b in zip(ks, ks[1:]))

        def placed(k) -> Expr:
            sv = syms[k]
            if sv.expr.dim is None:
                return sv.expr
            n_k = _numel(sv.shape)
            if runs and pos[k][0] == 0 and np.array_equal(at[pos[k]], np.arange(n_k)):
                return S.pad(sv.expr, d_out)
            return sv.expr[self.index(np.where(which == k, at, 0), n_out, n_k)]

        result = placed(ks[-1])
        for k in reversed(ks[:-1]):
            if runs:
                result = S.where_lt(d_out, int(pos[k][-1]) + 1, placed(k), result)
            else:
                result = S.select(self.data((which == k).astype(np.float64)) - 0.5, placed(k), result)
        return _Sym(result, oshape)

    # ---- products
    def matmul_like(self, a, b) -> Any:
        """torch.matmul semantics through broadcasting: (..., m, k) @ (..., k, n)"""
        torch = self.torch
        if self.is_const(a) and self.is_const(b):
            return torch.matmul(a, b)
        sa, sb = self.shape_of(a), self.shape_of(b)
        if len(sa) == 0 or len(sb) == 0:
            raise UnsupportedTorchOp("matmul with a zero-dimensional operand")
        # a data matrix times a traced vector with few columns: the IR's design-matrix form (one wave-wide sum per column in the
        # gradient); `v @ M` with M[k, n] data is the same product with the transposed matrix
        if self.is_const(b) and len(sb) == 2 and len(sa) in (1, 2) and _numel(sa) == sb[0] and not self.is_const(a):
            r = self._data_matrix_product(b.t(), a, (sb[1],) if len(sa) == 1 else (1, sb[1]), None)
            if r is not None:
                return r
        if self.is_const(a) and len(sa) == 2 and len(sb) in (1, 2) and _numel(sb) == sa[1] and not self.is_const(b):
            r = self._data_matrix_product(a, b, (sa[0],) if len(sb) == 1 else (sa[0], 1), a)
            if r is not None:
                return r
        av = a if self.is_const(a) else self.sym(a)
        bv = b if self.is_const(b) else self.sym(b)
        a1 = len(sa) == 1
        b1 = len(sb) == 1
        ua = self.move(av, lambda t: t.unsqueeze(0)) if a1 else av           # (1, k)
        ub = self.move(bv, lambda t: t.unsqueeze(-1)) if b1 else bv          # (k, 1)
        ua = self.move(ua, lambda t: t.unsqueeze(-1))                        # (..., m, k, 1)
        ub = self.move(ub, lambda t: t.unsqueeze(-3))                        # (..., 1, k, n)
        prod = self.binary(ua, ub, lambda x, y: x * y)                       # (..., m, k, n)
        out = self.sum(prod, [-2])                                           # (..., m, n)
        if a1:
            out = self.move(out, lambda t: t.squeeze(-2))
        if b1:
            out = self.move(out, lambda t: t.squeeze(-1))
        return out

    def _data_matrix_product(self, mat, vec, out_shape, original):
        torch = self.torch
        rows, k = int(mat.shape[0]), int(mat.shape[1])
        if not (1 < k <= 32 and rows > 1) or isinstance(vec, _X) and not self.whole:
            return None
        bv = self.sym(vec)
        if bv.expr.dim is None or not bool(torch.isfinite(mat).all()):
            return None
        name = self.fresh_name(original)
        m = self.m.matrix(name, mat.to(torch.float64).contiguous().cpu().numpy(), dim=self.dim(rows).name, cols=self.dim(k).name)
        return _Sym(m @ bv.expr, out_shape)

    # ---- scatter-adds: out = base + sum of source elements by target
    def scatter_add(self, base, source, target_of_source: np.ndarray) -> _Sym:
        base = self.sym(base)
        source = self.sym(source)
        n_out = _numel(base.shape)
        n_src = target_of_source.size
        se = source.expr
        if se.dim is None:
            se = S._bcast(se, self.dim(n_src))
        if n_out == 1:
            return _Sym(base.expr + se.sum(), base.shape)
        add = S._segsum(se, self.index(target_of_source, n_src, n_out))
        return _Sym(base.expr + add, base.shape)


def _scatter_replace(it: "_Interp", base, src, target_flat: np.ndarray) -> _Sym:
    """``base`` with the elements ``target_flat`` (one per element of ``src``, all different) replaced by ``src``"""
    torch = it.torch
    if len(np.unique(target_flat)) != target_flat.size:
        raise UnsupportedTorchOp("an indexed assignment with repeated indices (the result depends on the order of the writes; accumulate with index_add)")
    if isinstance(base, torch.Tensor):
        base = torch.nan_to_num(base.to(torch.float64), nan=0.0, posinf=0.0, neginf=0.0)      # (torch.empty: whatever was in memory)
    b = it.sym(base)
    n_out = _numel(b.shape)
    sv = it.sym(src)
    se = sv.expr
    n_src = target_flat.size
    if n_out == 1:
        return _Sym(se if se.dim is None else S.elem(se, 0), b.shape)
    if se.dim is None:
        se = S._bcast(se, it.dim(n_src)) if n_src > 1 else se
    mask = np.zeros(n_out)
    mask[target_flat] = 1.0
    if n_src == 1:
        placed = se                                     # one element: the scalar, selected by the mask
    else:
        placed = S._segsum(se, it.index(target_flat, n_src, n_out))
    if mask.all():
        return _Sym(placed, b.shape)
    return _Sym(S.select(it.data(mask) - 0.5, placed, b.expr), b.shape)


def _pow_const(e: Expr, c: float) -> Expr:
    if c == 0.0:
        return Expr.const(1.0)
    if c == 1.0:
        return e
    if c == 0.5:
        return S.sqrt(e)
    if c == -0.5:
        return 1.0 / S.sqrt(e)
    if c == float(int(c)) and abs(c) <= 2 ** 20:
        # every integer exponent by repeated squaring: exp(c log x) is NaN for a negative base, where torch (and the mathematics) has a value
        k = int(abs(c))
        result, base = None, e
        while k:
            if k & 1:
                result = base if result is None else result * base
            k >>= 1
            if k:
                base = base * base
        return result if c > 0 else 1.0 / result
    return S.exp(c * S.log(e))      # (a non-integer power: a negative base has none — NaN here as in torch)


def _provably_positive(e: Expr) -> bool:
    """conservative: exp(.), positive constants, and sums / products / quotients / square roots of such"""
    op = getattr(e, "op", None)
    if op == "const":
        try:
            return float(e.payload) > 0.0
        except Exception:
            return False
    if op == "exp":
        return True
    if op in ("add", "mul", "div"):
        return all(_provably_positive(a) for a in e.args)
    if op == "sqrt":
        return _provably_positive(e.args[0])
    return False


def _pow_traced(x: Expr, y: Expr) -> Expr:
    # x ** y with a traced exponent is exp(y log x): right for a positive base only (torch has values for negative bases with integer-valued
    # exponents, which cannot be told at trace time) — anything else is left to the eager path
    if not _provably_positive(x):
        raise UnsupportedTorchOp("pow with a traced exponent needs a base that is positive by construction (exp(.), a positive constant, sums / products of such)")
    return S.exp(y * S.log(x))


def _scalar(v) -> float:
    if hasattr(v, "item"):
        return float(v.item())
    return float(v)


def _run(gm, it: _Interp, x_shape, data_values: dict[str, Any]):
    import torch

    aten = torch.ops.aten
    env: dict[Any, Any] = {}
    placeholders = [n for n in gm.graph.nodes if n.op == "placeholder"]
    n_ops = 0

    def val(a):
        if isinstance(a, torch.fx.Node):
            return env[a]
        if isinstance(a, (list, tuple)):
            return type(a)(val(v) for v in a)
        return a

    def to_host(v):
        if isinstance(v, torch.Tensor):
            return v.detach().cpu()
        if isinstance(v, (list, tuple)):
            return type(v)(to_host(t) for t in v)
        return v

    def const_call(node, args, kwargs):
        # (constants live on the host whatever device the function was traced on)
        kwargs = {k: (torch.device("cpu") if k == "device" else v) for k, v in kwargs.items()}
        return to_host(node.target(*args, **kwargs))

    def any_traced(v) -> bool:
        if isinstance(v, (list, tuple)):
            return any(any_traced(t) for t in v)
        return isinstance(v, (_Sym, _Bool, _X))

    def x_piece(xv: _X, lo: int, hi: int, scalar: bool, out_shape):
        if it.whole:
            return None
        sv = it.piece(lo, hi, scalar)
        return _Sym(sv.expr, out_shape)

    for node in gm.graph.nodes:
        if node.op == "placeholder":
            k = placeholders.index(node)
            if k == 0:
                env[node] = _X(x_shape)
            else:
                name = it.data_names[k - 1]
                t = data_values[name]
                env[node] = t
                if isinstance(t, torch.Tensor):
                    it.named[id(t)] = name                            # keeps its name in the model's data when it is used as it is
                    it._alive.append(t)
            continue
        if node.op == "get_attr":
            t = getattr(gm, node.target)
            env[node] = to_host(t)
            continue
        if node.op == "output":
            out = val(node.args[0])
            if isinstance(out, (list, tuple)):
                if len(out) != 1:
                    raise UnsupportedTorchOp("the log-density must return one tensor")
                out = out[0]
            return out, n_ops
        if node.op != "call_function":
            raise UnsupportedTorchOp(f"fx node kind {node.op}")
        n_ops += 1
        tgt = node.target
        args, kwargs = val(node.args), val(dict(node.kwargs))
        if tgt is operator.getitem:
            env[node] = args[0][args[1]]
            continue
        if not any_traced(args) and not any_traced(list(kwargs.values())):
            env[node] = const_call(node, args, kwargs)               # constants only: evaluated now
            continue
        name = tgt.__name__ if hasattr(tgt, "__name__") else str(tgt)
        pkt = getattr(tgt, "overloadpacket", None)
        base = pkt.__name__ if pkt is not None else name
        if base.endswith("_copy") and base not in ("_to_copy", "lift_fresh_copy"):
            base = base[:-5]                      # (functionalised views: slice_copy, select_copy, view_copy, expand_copy ... move data like their views)
        a0 = args[0] if args else None

        # ---------------- the position vector: pieces become parameters
        if isinstance(a0, _X) and not it.whole:
            D = it.n_dim
            shp = a0.shape
            ax = [i for i, s in enumerate(shp) if s != 1]
            ax = ax[0] if ax else len(shp) - 1
            if base == "select":
                d_, idx = args[1] % len(shp), args[2]
                if d_ == ax and D > 0:
                    idx = idx % D
                    env[node] = x_piece(a0, idx, idx + 1, True, tuple(s for i, s in enumerate(shp) if i != d_))
                    continue
                if shp[d_] == 1:
                    env[node] = _X(tuple(s for i, s in enumerate(shp) if i != d_))
                    continue
            elif base == "slice":
                d_ = (args[1] if len(args) > 1 else 0) % len(shp)
                lo = args[2] if len(args) > 2 and args[2] is not None else 0
                hi = args[3] if len(args) > 3 and args[3] is not None else shp[d_]
                step = args[4] if len(args) > 4 else 1
                lo, hi, _ = slice(lo, hi, step).indices(shp[d_])
                if d_ == ax and step == 1:
                    if lo == 0 and hi == D:
                        env[node] = a0
                    elif hi > lo:
                        env[node] = x_piece(a0, lo, hi, False, tuple((hi - lo) if i == d_ else s for i, s in enumerate(shp)))
                    else:
                        raise UnsupportedTorchOp("an empty slice of x")
                    continue
                if d_ != ax and lo == 0 and hi == shp[d_] and step == 1:
                    env[node] = a0
                    continue
            elif base in ("split", "split_with_sizes", "unbind", "chunk"):
                probe = tgt(torch.arange(D, dtype=torch.int64).reshape(shp), *args[1:], **kwargs)
                outs = []
                ok = True
                for p in probe:
                    f = p.reshape(-1)
                    if f.numel() == 0 or not torch.equal(f, torch.arange(int(f[0]), int(f[0]) + f.numel())):
                        ok = False
                        break
                    outs.append((int(f[0]), int(f[0]) + f.numel(), tuple(p.shape)))
                if ok:
                    env[node] = [x_piece(a0, lo, hi, _numel(s) == 1 and base == "unbind", s) for lo, hi, s in outs]
                    continue
            elif base in ("view", "_unsafe_view", "reshape", "squeeze", "unsqueeze", "alias", "detach", "clone", "contiguous", "_to_copy", "lift_fresh_copy", "flatten", "expand"):
                probe = tgt(torch.zeros(shp, dtype=torch.float64), *args[1:], **kwargs)
                if probe.numel() == D and sum(1 for s in probe.shape if s != 1) <= 1:
                    env[node] = _X(tuple(probe.shape))
                    continue
            raise _NeedWholeVector(f"x is used by {name}")

        # ---------------- element-wise arithmetic
        B = it.binary
        U = it.unary

        def alpha_of():
            return _scalar(kwargs.get("alpha", 1))

        if base == "add":
            al = alpha_of()
            env[node] = B(args[0], args[1], lambda x, y: x + (y if al == 1 else al * y))
        elif base == "sub":
            al = alpha_of()
            env[node] = B(args[0], args[1], lambda x, y: x - (y if al == 1 else al * y))
        elif base == "rsub":
            al = alpha_of()
            env[node] = B(args[0], args[1], lambda x, y: y - (x if al == 1 else al * x))
        elif base == "mul":
            env[node] = B(args[0], args[1], lambda x, y: x * y)
        elif base in ("div", "true_divide"):
            if kwargs.get("rounding_mode") is not None:
                raise UnsupportedTorchOp("div with a rounding mode")
            env[node] = B(args[0], args[1], lambda x, y: x / y)
        elif base == "neg":
            env[node] = U(a0, lambda x: -x)
        elif base == "reciprocal":
            env[node] = U(a0, lambda x: 1.0 / x)
        elif base == "square":
            env[node] = U(a0, lambda x: x * x)
        elif base == "pow":
            if it.is_const(args[1]) and (not isinstance(args[1], torch.Tensor) or args[1].numel() == 1):
                c = _scalar(args[1])
                env[node] = U(a0, lambda x: _pow_const(x, c))
            elif it.is_const(a0) and (not isinstance(a0, torch.Tensor) or a0.numel() == 1):
                c = _scalar(a0)
                if c <= 0:
                    raise UnsupportedTorchOp("a non-positive constant to a traced power")
                env[node] = U(args[1], lambda y: S.exp(math.log(c) * y))
            else:
                env[node] = B(args[0], args[1], _pow_traced)
        elif base in ("sqrt", "exp", "log", "log1p", "sigmoid", "tanh", "expm1", "erf", "erfc", "sin", "cos", "atan", "lgamma", "digamma", "sign"):
            env[node] = U(a0, lambda x: S._unary(base, x))
        elif base == "abs":
            env[node] = U(a0, S.absolute)
        elif base == "rsqrt":
            env[node] = U(a0, lambda x: 1.0 / S.sqrt(x))
        elif base == "log2":
            env[node] = U(a0, lambda x: S.log(x) * (1.0 / math.log(2.0)))
        elif base == "log10":
            env[node] = U(a0, lambda x: S.log(x) * (1.0 / math.log(10.0)))
        elif base == "exp2":
            env[node] = U(a0, lambda x: S.exp(x * math.log(2.0)))
        elif base == "softplus":
            beta = _scalar(args[1] if len(args) > 1 else kwargs.get("beta", 1.0))
            thr = _scalar(args[2] if len(args) > 2 else kwargs.get("threshold", 20.0))
            env[node] = U(a0, lambda x: S.select(x * beta - thr, x, S.softplus(x * beta) * (1.0 / beta) if beta != 1.0 else S.softplus(x)))
        elif base == "log_sigmoid_forward":
            r = U(a0, lambda x: -S.softplus(-x))
            env[node] = (r, r)
        elif base == "logit":
            env[node] = U(a0, lambda x: S.log(x) - S.log1p(-x))
        elif base == "xlogy":
            env[node] = B(args[0], args[1], lambda x, y: x * S.log(y))
        elif base == "xlog1py":
            env[node] = B(args[0], args[1], lambda x, y: x * S.log1p(y))
        elif base == "relu":
            env[node] = U(a0, lambda x: S.select(x, x, 0.0))
        elif base in ("maximum", "fmax"):
            env[node] = B(args[0], args[1], lambda x, y: S.select(x - y, x, y, False))
        elif base in ("minimum", "fmin"):
            env[node] = B(args[0], args[1], lambda x, y: S.select(y - x, x, y, False))
        elif base in ("clamp", "clamp_min", "clamp_max", "clip"):
            lo = args[1] if len(args) > 1 else kwargs.get("min")
            hi = (args[2] if len(args) > 2 else kwargs.get("max")) if base in ("clamp", "clip") else None
            if base == "clamp_max":
                lo, hi = None, lo
            r = it.sym(a0)
            if lo is not None:
                r = B(r, lo, lambda x, y: S.select(x - y, x, y, False))
            if hi is not None:
                r = B(r, hi, lambda x, y: S.select(y - x, x, y, False))
            env[node] = r
        elif base in ("gt", "lt", "ge", "le", "eq", "ne"):
            env[node] = it.compare(args[0], args[1], base)
        elif base == "logaddexp":
            # max(a, b) + log1p(exp(-|a - b|))
            env[node] = B(args[0], args[1], lambda x, y: S.select(x - y, x, y, False) + S.log1p(S.exp(-S.absolute(x - y))))
        elif base in ("logical_not", "bitwise_not"):
            b_ = it.as_bool(a0)
            env[node] = _Bool(("not", b_.tree), b_.shape)
        elif base in ("logical_and", "bitwise_and", "logical_or", "bitwise_or"):
            l, r = it.as_bool(args[0]), it.as_bool(args[1])
            shape = tuple(torch.broadcast_shapes(l.shape, r.shape))
            lt = l.tree if _numel(l.shape) == _numel(shape) else it.move(l, lambda t: torch.broadcast_to(t, shape)).tree
            rt = r.tree if _numel(r.shape) == _numel(shape) else it.move(r, lambda t: torch.broadcast_to(t, shape)).tree
            env[node] = _Bool(("and" if "and" in base else "or", lt, rt), shape)
        elif base == "where":
            if len(args) != 3:
                raise UnsupportedTorchOp("where(condition) without values")
            env[node] = it.where(args[0], args[1], args[2])
        elif base == "masked_fill":
            env[node] = it.where(args[1], args[2], args[0])
        # ---------------- reductions and products
        elif base == "sum":
            env[node] = it.sum(a0, args[1] if len(args) > 1 else kwargs.get("dim"), args[2] if len(args) > 2 else kwargs.get("keepdim", False))
        elif base == "mean":
            dims = args[1] if len(args) > 1 else kwargs.get("dim")
            r = it.sum(a0, dims, args[2] if len(args) > 2 else kwargs.get("keepdim", False))
            cnt = _numel(it.shape_of(a0)) // max(_numel(r.shape), 1)
            env[node] = _Sym(r.expr * (1.0 / cnt), r.shape)
        elif base in ("var", "std"):
            # (sum of squared deviations from the mean) / (n - correction) over the given axes
            dims = args[1] if len(args) > 1 else kwargs.get("dim")
            corr = kwargs.get("correction", 1)
            if isinstance(dims, bool):       # the overload var(x, unbiased)
                corr, dims = (1 if dims else 0), None
            corr = 1 if corr is None else corr
            keep = kwargs.get("keepdim", False)
            m_ = it.sum(a0, dims, True)
            cnt = _numel(it.shape_of(a0)) // max(_numel(m_.shape), 1)
            if cnt - corr <= 0:
                raise UnsupportedTorchOp(f"{name} of {cnt} element(s) with correction {corr}")
            dev = B(a0, _Sym(m_.expr * (1.0 / cnt), m_.shape), lambda x, y: x - y)
            r = it.sum(_Sym(dev.expr * dev.expr, dev.shape), dims, keep)
            e_ = r.expr * (1.0 / (cnt - corr))
            env[node] = _Sym(S.sqrt(e_) if base == "std" else e_, r.shape)
        elif base == "linalg_vector_norm":
            ord_ = args[1] if len(args) > 1 else kwargs.get("ord", 2)
            dims = args[2] if len(args) > 2 else kwargs.get("dim")
            keep = args[3] if len(args) > 3 else kwargs.get("keepdim", False)
            v_ = it.sym(a0)
            if ord_ in (2, 2.0):
                r = it.sum(_Sym(v_.expr * v_.expr, v_.shape), dims, keep)
                env[node] = _Sym(S.sqrt(r.expr), r.shape)
            elif ord_ in (1, 1.0):
                r = it.sum(_Sym(S.absolute(v_.expr), v_.shape), dims, keep)
                env[node] = r
            else:
                raise UnsupportedTorchOp(f"{name} with ord = {ord_}")
        elif base in ("amax", "amin", "max", "min", "logsumexp", "_softmax", "_log_softmax", "softmax", "log_softmax"):
            # reductions by the maximum — over ALL elements of the tensor (one chain's vector): the IR's `max` is dimension -> scalar
            v_ = it.sym(a0)
            dims = args[1] if len(args) > 1 else kwargs.get("dim")
            if base in ("max", "min") and len(args) > 1:
                raise UnsupportedTorchOp(f"{name} along an axis (values and indices)")
            shp = v_.shape
            nd = len(shp)
            if dims is None or (isinstance(dims, (list, tuple)) and len(dims) == 0):
                dims = list(range(nd))
            if isinstance(dims, int):
                dims = [dims]
            dims = sorted({d_ % nd for d_ in dims}) if nd else []
            if _numel(tuple(s_ for i_, s_ in enumerate(shp) if i_ not in dims)) != 1:
                # along ONE short axis of a tensor with several rows (a multinomial logit: log_softmax of [observations, classes]): the
                # maximum of every row as a chain of selects over the axis' slices, then the shifted sums as segment sums
                if len(dims) != 1 or shp[dims[0]] > 32:
                    raise UnsupportedTorchOp(f"{name} along an axis of more than 32 elements of a tensor with several rows")
                ax = dims[0]
                keep = bool(args[2] if len(args) > 2 else kwargs.get("keepdim", False)) if base in ("amax", "amin", "logsumexp") else False
                sign = -1.0 if base in ("amin", "min") else 1.0
                cols = [it.move(v_, lambda t, k_=k_: t.select(ax, k_).unsqueeze(ax)) for k_ in range(shp[ax])]
                m_ = cols[0] if sign > 0 else _Sym(-cols[0].expr, cols[0].shape)
                for c_ in cols[1:]:
                    ce = c_.expr if sign > 0 else -c_.expr
                    m_ = _Sym(S.select(ce - m_.expr, ce, m_.expr, False), m_.shape)
                squeeze = (lambda sv: sv if keep else it.move(sv, lambda t: t.squeeze(ax)))
                if base in ("amax", "amin", "max", "min"):
                    env[node] = squeeze(_Sym(m_.expr if sign > 0 else -m_.expr, m_.shape))
                    continue
                sh = B(v_, m_, lambda x_, y_: x_ - y_)
                ex = U(sh, S.exp)
                tot = it.sum(ex, [ax], True)
                if base == "logsumexp":
                    env[node] = squeeze(B(m_, U(tot, S.log), lambda x_, y_: x_ + y_))
                elif base in ("_softmax", "softmax"):
                    env[node] = B(ex, tot, lambda x_, y_: x_ / y_)
                else:
                    env[node] = B(sh, U(tot, S.log), lambda x_, y_: x_ - y_)
                continue
            keep = bool(args[2] if len(args) > 2 else kwargs.get("keepdim", False)) if base in ("amax", "amin", "logsumexp") else False
            oshape = tuple(1 if i_ in dims else s_ for i_, s_ in enumerate(shp)) if keep else tuple(s_ for i_, s_ in enumerate(shp) if i_ not in dims)
            e = v_.expr
            if base in ("amax", "max"):
                env[node] = _Sym(e.max() if e.dim is not None else e, oshape)
            elif base in ("amin", "min"):
                env[node] = _Sym(-((-e).max()) if e.dim is not None else e, oshape)
            elif e.dim is None:      # all elements equal
                n_ = float(_numel(shp))
                env[node] = _Sym(e + math.log(n_), oshape) if base == "logsumexp" else _Sym(Expr.const(1.0 / n_) if "log" not in base else Expr.const(-math.log(n_)), shp)
            else:
                m_ = e.max(constant=True)            # the shift: cancels exactly, no gradient through it
                ex = S.exp(e - m_)
                tot = ex.sum()
                if base == "logsumexp":
                    env[node] = _Sym(m_ + S.log(tot), oshape)
                elif base in ("_softmax", "softmax"):
                    env[node] = _Sym(ex / tot, shp)
                else:
                    env[node] = _Sym((e - m_) - S.log(tot), shp)
        elif base in ("cumsum", "logcumsumexp") and base == "cumsum":
            # a prefix sum along a short axis (ordered cut points: PyMC's `ordered` transform is a cumsum of exponentials): every output element
            # sums the elements before it — a gather of the (i, j <= i) pairs and a segment sum back, n (n + 1) / 2 terms
            v_ = it.sym(a0)
            shp = v_.shape
            ax = (args[1] if len(args) > 1 else kwargs.get("dim")) % max(len(shp), 1)
            n_ax = shp[ax] if shp else 1
            if n_ax > 64:
                raise UnsupportedTorchOp("cumsum along an axis of more than 64 elements")
            if v_.expr.dim is None:
                ramp = torch.arange(1, n_ax + 1, dtype=torch.float64).reshape([n_ax if i_ == ax else 1 for i_ in range(len(shp))]).expand(shp)
                env[node] = B(v_, ramp, lambda x_, y_: x_ * y_)
            else:
                n_ = _numel(shp)
                flat = torch.arange(n_, dtype=torch.int64).reshape(shp)
                src, dst = [], []
                for i_ in range(n_ax):
                    for j_ in range(i_ + 1):
                        src.append(flat.select(ax, j_).reshape(-1))
                        dst.append(flat.select(ax, i_).reshape(-1))
                src, dst = torch.cat(src).numpy(), torch.cat(dst).numpy()
                pairs = v_.expr[it.index(src, src.size, n_)]
                env[node] = _Sym(S._segsum(pairs, it.index(dst, dst.size, n_)), shp)
        elif base == "dot" or base == "vdot":
            env[node] = it.sum(B(args[0], args[1], lambda x, y: x * y))
        elif base == "linalg_solve_triangular":
            # A X = B (left) or X A = B with a CONSTANT triangular A (the scale_tril of a MultivariateNormal): a product with A^-1
            A_, B_ = args[0], args[1]
            if not it.is_const(A_):
                raise UnsupportedTorchOp(f"{name} with a traced matrix (a covariance that depends on parameters)")
            eye = torch.eye(A_.shape[-1], dtype=A_.dtype)
            Ainv = torch.linalg.solve_triangular(A_, eye.expand_as(A_).contiguous(), upper=bool(kwargs.get("upper", False)),
                                                 left=True, unitriangular=bool(kwargs.get("unitriangular", False)))
            env[node] = it.matmul_like(Ainv, B_) if kwargs.get("left", True) else it.matmul_like(B_, Ainv)
        elif base in ("mv", "mm", "matmul", "bmm"):
            env[node] = it.matmul_like(args[0], args[1])
        elif base == "addmm" or base == "addmv":
            beta, al = _scalar(kwargs.get("beta", 1)), _scalar(kwargs.get("alpha", 1))
            prod = it.matmul_like(args[1], args[2])
            env[node] = B(args[0], prod, lambda x, y: (x if beta == 1 else beta * x) + (y if al == 1 else al * y))
        # ---------------- data movement
        elif base in ("view", "_unsafe_view", "reshape", "squeeze", "unsqueeze", "expand", "permute", "transpose", "t", "select", "slice", "narrow",
                      "flip", "diagonal", "alias", "flatten", "unflatten", "movedim", "swapaxes", "index_select", "roll", "repeat", "tril", "triu",
                      "split", "split_with_sizes", "unbind", "chunk", "as_strided", "expand_as", "view_as", "take"):
            if base in ("tril", "triu"):
                raise UnsupportedTorchOp(base)
            if any_traced(args[1:]) or any_traced(list(kwargs.values())):
                raise UnsupportedTorchOp(f"{name} with a traced index")
            env[node] = it.move(a0, lambda t: tgt(t, *args[1:], **kwargs))
        elif base == "index":
            idx = args[1]
            if any_traced(idx):
                raise UnsupportedTorchOp("indexing with a traced index")
            if any(isinstance(i_, torch.Tensor) and i_.dtype == torch.bool for i_ in idx if i_ is not None):
                pass
            env[node] = it.move(a0, lambda t: tgt(t, idx))
        elif base == "gather":
            if any_traced(args[1:]):
                raise UnsupportedTorchOp("gather with a traced index")
            env[node] = it.move(a0, lambda t: tgt(t, *args[1:], **kwargs))
        elif base in ("clone", "contiguous", "detach", "lift_fresh_copy", "lift_fresh", "alias", "positive", "_to_copy", "to", "type_as", "double", "float"):
            dt = kwargs.get("dtype")
            if isinstance(a0, _Bool) and (dt is None or dt == torch.bool):
                env[node] = a0
            elif dt is not None and not dt.is_floating_point:
                raise UnsupportedTorchOp(f"a traced value converted to {dt}")
            else:
                env[node] = it.sym(a0)
        elif base == "cat" or base == "concat" or base == "concatenate":
            env[node] = it.cat(list(args[0]), args[1] if len(args) > 1 else kwargs.get("dim", 0))
        elif base == "stack":
            d_ = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            parts = [it.move(p, lambda t: t.unsqueeze(d_ if d_ >= 0 else d_ + t.dim() + 1)) if not it.is_const(p) else p.unsqueeze(d_) for p in args[0]]
            env[node] = it.cat(parts, d_)
        elif base in ("zeros_like", "ones_like", "full_like", "empty_like", "new_zeros", "new_ones", "new_full", "new_empty"):
            shp = it.shape_of(a0)
            proxy = torch.zeros(shp, dtype=torch.float64)
            env[node] = tgt(proxy, *args[1:], **{k: v for k, v in kwargs.items() if k not in ("device", "pin_memory", "layout")})
        elif base == "index_add":
            if any_traced([args[1], args[2]]):
                raise UnsupportedTorchOp("index_add with a traced index")
            base_shape = it.shape_of(args[0])
            tmap = torch.arange(_numel(base_shape), dtype=torch.int64).reshape(base_shape).index_select(args[1], args[2])
            src = it.sym(args[3])
            al = alpha_of()
            if al != 1:
                src = _Sym(src.expr * al, src.shape)
            srcb = _Sym(it.broadcast(src, tuple(tmap.shape)), tuple(tmap.shape))
            env[node] = it.scatter_add(args[0], srcb, tmap.reshape(-1).numpy())
        elif base == "scatter_add":
            if any_traced([args[1], args[2]]):
                raise UnsupportedTorchOp("scatter_add with a traced index")
            base_shape = it.shape_of(args[0])
            index = args[2]
            tmap = torch.arange(_numel(base_shape), dtype=torch.int64).reshape(base_shape).gather(args[1], index)
            # element (i, j, ...) of src (restricted to index's shape) goes to base[..., index[i, j, ...], ...]
            src = it.move(it.sym(args[3]), lambda t: t[tuple(slice(0, s) for s in index.shape)])
            coords = torch.meshgrid(*[torch.arange(s) for s in index.shape], indexing="ij") if index.dim() else ()
            coords = list(coords)
            if index.dim():
                coords[args[1] % index.dim()] = index
                strides = torch.tensor([int(np.prod(base_shape[i + 1:], dtype=np.int64)) for i in range(len(base_shape))])
                tflat = sum(c * s for c, s in zip(coords, strides)).reshape(-1).numpy()
            else:
                tflat = np.zeros(1, dtype=np.int64)
            del tmap
            env[node] = it.scatter_add(args[0], src, tflat)
        elif base == "copy":
            src = it.sym(args[1])
            shp = it.shape_of(args[0])
            env[node] = _Sym(it.broadcast(src, shp), shp)
        elif base in ("slice_scatter", "select_scatter", "diagonal_scatter"):
            base_shape = it.shape_of(args[0])
            view = {"slice_scatter": torch.ops.aten.slice.Tensor, "select_scatter": torch.ops.aten.select.int, "diagonal_scatter": torch.ops.aten.diagonal.default}[base]
            tmap = view(torch.arange(_numel(base_shape), dtype=torch.int64).reshape(base_shape), *args[2:], **kwargs)
            src = it.sym(args[1])
            srcb = _Sym(it.broadcast(src, tuple(tmap.shape)), tuple(tmap.shape))
            env[node] = _scatter_replace(it, args[0], srcb, tmap.reshape(-1).numpy())
        elif base == "index_put" and not (args[3] if len(args) > 3 else kwargs.get("accumulate", False)):
            idx = args[1]
            if any_traced(idx):
                raise UnsupportedTorchOp("index_put with a traced index")
            base_shape = it.shape_of(args[0])
            tmap = torch.ops.aten.index.Tensor(torch.arange(_numel(base_shape), dtype=torch.int64).reshape(base_shape), idx)
            src = it.sym(args[2])
            srcb = _Sym(it.broadcast(src, tuple(tmap.shape)), tuple(tmap.shape))
            env[node] = _scatter_replace(it, args[0], srcb, tmap.reshape(-1).numpy())
        elif base == "index_put":
            idx = args[1]
            if any_traced(idx):
                raise UnsupportedTorchOp("index_put with a traced index")
            base_shape = it.shape_of(args[0])
            tmap = torch.ops.aten.index.Tensor(torch.arange(_numel(base_shape), dtype=torch.int64).reshape(base_shape), idx)
            src = it.sym(args[2])
            srcb = _Sym(it.broadcast(src, tuple(tmap.shape)), tuple(tmap.shape))
            env[node] = it.scatter_add(args[0], srcb, tmap.reshape(-1).numpy())
        else:
            raise UnsupportedTorchOp(f"{name} (no counterpart in the expression IR)")
    raise UnsupportedTorchOp("the traced graph has no output")


def trace(logp_fn: Callable, n_dim: int, *, batched: bool = True, shared_data: dict[str, Any] | None = None, example=None) -> TraceResult:
    """Trace ``logp_fn`` into a :class:`nutpie_amd.symbolic.Model`.

    ``logp_fn(x, **shared_data)``: with ``batched`` (the convention of :func:`nutpie_amd.from_torch_density`) ``x`` is ``[chains, n_dim]``
    and the result ``[chains]`` — traced with ONE chain, rows are independent —, otherwise ``x`` is ``[n_dim]`` and the result one
    number.  Only the forward pass is traced; the gradient is derived from the IR.  ``shared_data`` entries are passed as keyword
    arguments and keep their names as data arrays of the model; tensors the function closes over become anonymous data.
    Python control flow on the values of ``x`` is followed for the example point only (as with every tracer)."""
    import torch
    from torch.fx.experimental.proxy_tensor import make_fx

    n_dim = int(n_dim)
    shared = dict(shared_data or {})
    names = list(shared)
    tens = {}
    for k in names:
        v = shared[k]
        tens[k] = v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
    x_shape = (1, n_dim) if batched else (n_dim,)
    if example is None:
        example = 0.1 * torch.randn(x_shape, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    else:
        example = torch.as_tensor(np.asarray(example), dtype=torch.float64).reshape(x_shape)

    def fn(x, *data):
        return logp_fn(x, **dict(zip(names, data)))

    # (argument validation of torch.distributions asks for the VALUE of a comparison — data-dependent control flow no tracer can
    #  follow; the checks say nothing about the density itself)
    from torch.distributions import Distribution

    validate = Distribution._validate_args
    Distribution.set_default_validate_args(False)
    def run_make_fx(device):
        # (functionalised first: `y.mul_(2)`, `out[:3] = y`, `tot.index_add_(0, idx, a)` become out-of-place operations + scatters)
        from torch.func import functionalize

        with torch.no_grad():
            return make_fx(functionalize(fn, remove="mutations_and_views"))(example.to(device), *[tens[k].to(device) for k in names])

    try:
        try:
            gm = run_make_fx("cpu")
        except RuntimeError as e:
            # the function closes over tensors that live on the GPU (what the eager path needs): trace there, the constants of
            # the trace are brought to the host below
            if "same device" in str(e) and torch.cuda.is_available():
                gm = run_make_fx("cuda")
            else:
                raise
    except RuntimeError as e:
        if "tracing tensor" in str(e) or "data-dependent" in str(e):
            raise UnsupportedTorchOp("Python control flow on the values of x (" + str(e).split(" - ")[0][-60:] + ")") from e
        if "same device" in str(e):
            raise UnsupportedTorchOp("the function mixes tensors of several devices and no GPU is here to trace it on") from e
        raise
    finally:
        Distribution.set_default_validate_args(validate)
    gm.graph.eliminate_dead_code()
    last_err = None
    for whole in (False, True):
        it = _Interp(n_dim, whole, names)
        try:
            out, n_ops = _run(gm, it, x_shape, tens)
        except _NeedWholeVector as e:
            last_err = e
            continue
        if not isinstance(out, _Sym):
            if isinstance(out, _X):
                raise UnsupportedTorchOp("the log-density returns x itself")
            raise UnsupportedTorchOp("the log-density does not depend on x")
        if _numel(out.shape) != 1:
            raise UnsupportedTorchOp(f"the log-density must have one value per chain (traced shape {out.shape})")
        e = out.expr if out.expr.dim is None else S.elem(out.expr, 0)
        m = it.m
        _declare_parameters(m, it)
        m.add_logp(e)
        return TraceResult(m, n_dim, whole, n_ops)
    raise UnsupportedTorchOp(f"could not map the position vector: {last_err}")


def _declare_parameters(m: S.Model, it: _Interp) -> None:
    """the parameter nodes the trace created, in the order of the flat vector, gaps filled (their gradient is zero)"""
    D = it.n_dim
    if it.whole:
        nodes = [("x", it.x_expr, 0, D)]
    else:
        nodes, at = [], 0
        for (lo, hi, _), e in sorted(it.pieces.items(), key=lambda kv: kv[0][0]):
            if lo > at:
                nodes.append((f"x_{at}", Expr("vparam", (), it.dim(lo - at), (at, lo - at)) if lo - at > 1 else Expr("sparam", (), None, at), at, lo))
            nodes.append((f"x_{lo}", e, lo, hi))
            at = hi
        if at < D:
            nodes.append((f"x_{at}", Expr("vparam", (), it.dim(D - at), (at, D - at)) if D - at > 1 else Expr("sparam", (), None, at), at, D))
    for name, e, lo, hi in nodes:
        m._param_names.append(name)
        m._params.append(e)
        m._unconstrained[name] = (name, lo, hi - lo)
        m._det.append((name, e))
    m._n_dim = D


_TRACED_MODEL = None


def _traced_model_class():
    global _TRACED_MODEL
    if _TRACED_MODEL is not None:
        return _TRACED_MODEL
    import dataclasses

    @dataclasses.dataclass(frozen=True)
    class TracedTorchModel(S._symbolic_model_class()):
        """A torch log-density compiled into the engine (:func:`traced_model`).  ``with_data`` traces the function again with the
        new shared data — constants derived from the data are part of the trace — and finds the compiled library in the cache
        when only values changed (the library is keyed by the generated source)."""

        _retrace: Any = None
        _shared: Any = None

        def with_data(self, **updates):
            unknown = next((k for k in updates if k not in (self._shared or {})), None)
            if unknown is not None:
                raise ValueError(f"Unknown data variable: {unknown}")
            return self._retrace({**self._shared, **updates})

    _TRACED_MODEL = TracedTorchModel
    return TracedTorchModel


def traced_model(ndim: int, density_fn: Callable, *, batched: bool = True, shared_data: dict[str, Any] | None = None, expand_fn: Callable | None = None,
                 expanded_shapes=None, expanded_names=None, coords=None, dims=None, init="uniform", reparameterized_names=None,
                 waves_per_chain: int | None = None, resident: bool = True):
    """``density_fn`` traced and compiled: what :func:`nutpie_amd.from_torch_density` returns on its compiled path.  ``expand_fn``
    (numpy ``[N, ndim]`` -> dict of arrays, with ``expanded_names`` / ``expanded_shapes``) as in :func:`nutpie_amd.from_torchfunc`."""
    import dataclasses

    if (expand_fn is None) != (expanded_names is None):
        raise ValueError("expand_fn needs expanded_names and expanded_shapes")

    def build(shared):
        tr = trace(density_fn, ndim, batched=batched, shared_data=shared)
        user_expand = None
        if expand_fn is not None:
            def user_expand(positions, /, **_model_data):      # (the model's data arrays are the trace's, not the user's keywords)
                return expand_fn(np.asarray(positions), **shared)
        kw = dict(init=init, coords=coords, dims=dims, waves_per_chain=waves_per_chain, resident=resident)
        if expand_fn is not None:
            kw.update(expand_fn=user_expand, expanded_names=list(expanded_names), expanded_shapes=[tuple(s_) for s_ in expanded_shapes])
        try:
            base = tr.compile(**kw)
        except UnsupportedTorchOp:
            raise
        except NotImplementedError as e:   # (e.g. symbolic.gradient: an operation whose derivative the IR does not have)
            raise UnsupportedTorchOp(str(e)) from e
        cls = _traced_model_class()
        fields = {f.name: getattr(base, f.name) for f in dataclasses.fields(base)}
        if reparameterized_names is not None:
            fields["reparameterized_names"] = reparameterized_names
        return cls(**fields, _retrace=build, _shared=dict(shared))

    return build(dict(shared_data or {}))
