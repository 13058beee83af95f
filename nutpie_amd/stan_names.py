"""Stan output bookkeeping: flat parameter names -> variables, and column-major -> C order.

BridgeStan reports constrained parameters as one flat vector with names such as ``x.2.1`` (1-based
indices, first index fastest = column-major) and ``z.1.real`` / ``z.1.imag`` for complex entries; tuple
fields use ``:`` and are part of the name.  The reference parses those names and re-orders every block to
C order in Rust (``src/stan.rs:93-251`` ``params`` / ``determine_variable_shape`` and ``:671-711``
``fortran_to_c_order``; known-answer tests ``src/stan.rs:819-1231``).  This module restates that logic for
the HIP engine's trace hand-off, vectorised over all draws at once.
"""

from __future__ import annotations

from dataclasses import dataclass
from itertools import groupby

import numpy as np


@dataclass(frozen=True)
class StanVariable:
    name: str
    shape: tuple[int, ...]
    start: int   # offset of the block in the flat constrained vector
    end: int

    @property
    def num_elements(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64))


def _parse_one(var: str):
    """``name.i.j(.real|.imag)`` -> (name, is_complex, zero-based indices); parsed right to left like the reference."""
    indices = []
    remaining = var
    is_complex = False
    while True:
        idx = remaining.rfind(".")
        if idx < 0:
            break
        suffix = remaining[idx + 1:]
        if suffix in ("real", "imag"):
            is_complex = True
            remaining = remaining[:idx]
            continue
        if suffix.isdigit():
            one_based = int(suffix)
            if one_based < 1:
                raise ValueError("Invalid parameter index (must be > 0)")
            indices.append(one_based - 1)
            remaining = remaining[:idx]
        else:
            break  # not a number: part of the variable name
    indices.reverse()
    return remaining.strip(), is_complex, tuple(indices)


def _shape_of_group(name, group):
    is_complex = group[0][1]
    rank = len(group[0][2])
    shape = [0] * rank
    for _, cplx, idx in group:
        if cplx != is_complex:
            raise ValueError(f"Inconsistent complex flags for variable {name}")
        if len(idx) != rank:
            raise ValueError(f"Error while parsing stan variable {name}")
        shape = [max(a, b) for a, b in zip(shape, idx)]
    shape = [m + 1 for m in shape]
    # the entries must arrive in column-major (Fortran) order, real before imag
    expected = [0] * rank
    expect_imag = False
    for _, _, idx in group:
        if list(idx) != expected:
            raise ValueError("Stan returned data that was not in the expected order.")
        if is_complex:
            expect_imag = not expect_imag
        if not expect_imag:
            for i in range(rank):
                if expected[i] < shape[i] - 1:
                    expected[i] += 1
                    break
                expected[i] = 0
    return tuple(shape), is_complex


def parse_stan_variables(var_string: str) -> list[StanVariable]:
    """Comma-separated BridgeStan parameter names -> variables with shapes and flat offsets."""
    if var_string == "":
        return []
    parsed = [_parse_one(v) for v in var_string.split(",")]
    out: list[StanVariable] = []
    start = 0
    for name, grp in groupby(parsed, key=lambda t: t[0]):
        group = list(grp)
        shape, is_complex = _shape_of_group(name, group)
        size = int(np.prod(shape, dtype=np.int64))
        if is_complex:
            out.append(StanVariable(f"{name}.real", shape, start, start + size))
            start += size
            out.append(StanVariable(f"{name}.imag", shape, start, start + size))
        else:
            out.append(StanVariable(name, shape, start, start + size))
        start += size
    return out


def fortran_to_c_order(data, shape) -> np.ndarray:
    """One column-major block (or a batch ``[..., size]`` of blocks) re-ordered to C order, flattened."""
    data = np.asarray(data)
    shape = tuple(int(s) for s in shape)
    size = int(np.prod(shape, dtype=np.int64))
    if data.shape[-1] != size:
        raise ValueError("block size does not match shape")
    lead = data.shape[:-1]
    rank = len(shape)
    if rank <= 1:
        return data.copy()
    # column-major: the FIRST index is fastest -> view as reversed shape in C order, then reverse the axes
    blk = data.reshape(*lead, *shape[::-1])
    axes = tuple(range(len(lead))) + tuple(len(lead) + rank - 1 - i for i in range(rank))
    return np.ascontiguousarray(blk.transpose(axes)).reshape(*lead, size)


def expand_constrained(flat, variables: list[StanVariable]) -> dict[str, np.ndarray]:
    """``flat[..., n_constrained]`` (BridgeStan ``param_constrain`` output per draw) -> ``{name: [..., *shape]}`` in C order."""
    flat = np.asarray(flat)
    out = {}
    for v in variables:
        blk = fortran_to_c_order(flat[..., v.start:v.end], v.shape) if v.shape else flat[..., v.start:v.end]
        out[v.name] = blk.reshape(*flat.shape[:-1], *v.shape)
    return out


def c_order_permutation(variables: list[StanVariable]) -> np.ndarray:
    """``perm`` with ``out[j] = theta[perm[j]]``: BridgeStan's flat ``param_constrain`` output (every variable a column-major
    block) -> the same vector with every block in C order.  This is what ``nphip_model_set_bridgestan_expand`` applies to each
    draw natively (the reference: ``fortran_to_c_order`` per variable per draw, src/stan.rs:507-516, 671-711)."""
    total = variables[-1].end if variables else 0
    perm = np.arange(total, dtype=np.uint64)
    for v in variables:
        if len(v.shape) >= 2:
            perm[v.start:v.end] = v.start + fortran_to_c_order(np.arange(v.end - v.start), v.shape).astype(np.uint64)
    return perm
