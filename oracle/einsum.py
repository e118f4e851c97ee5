"""CPU restatement of the reference's Einsum operator.  TEST INFRASTRUCTURE ONLY (see oracle/ref.py's header:
only tests/, smoke() and bench.py's cpu_baseline leg may import anything under oracle/).

Follows the reference function by function, on numpy views where the reference uses TensorViews:
  * parser / validation      rten-shape-inference/src/einsum_parser.rs:68-166,191-275
  * path (2 terms per step)  src/ops/einsum.rs:566-692
  * one step                 src/ops/einsum.rs:124-363 (diagonals, lone-dim sums, mul / matmul lowering)
  * matmul lowering          src/ops/einsum.rs:449-537
  * reduce_sum               src/ops/reduce.rs:414-520,1101-1124 (slice packed in row-major order of the reduced dims,
                             vecmath::Sum in V-lane order -> oracle rto_simd_sum)
Arithmetic is delegated to the pinned C oracle: gemm_f32 (kc-blocked k-ordered chain), rto_simd_sum, f32 multiply.

Parity status: pinned against the reference's own test literals (src/ops/einsum.rs:705-1311 cases with literal expected
values, error strings, and the test_einsum_path table :1313-1498; parser table einsum_parser.rs:277-557) in
tests/test_einsum.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import ref

MAX_DIMS = 10  # einsum_parser.rs:241-245
INSERTED_M, INSERTED_N, MERGED_K = "<", ">", "*"  # einsum.rs:366-380


class EinsumError(ref.OpError):
    def __init__(self, kind, msg):
        super().__init__(f"{kind}: {msg}")
        self.kind, self.msg = kind, msg


# ------------------------------------------------------------------------------------------ parser
def _is_valid_term(term: str) -> bool:  # einsum_parser.rs:231-237
    if "..." in term:
        lhs, rhs = term.split("...", 1)
        return _is_valid_term(lhs) and "..." not in rhs and _is_valid_term(rhs)
    return all(c.isascii() and c.isalpha() for c in term)


def _strip_ws(s: str) -> str:
    return "".join(c for c in s if c not in " \t\n\r\x0c")  # is_ascii_whitespace


def parse(expr: str):
    """EinsumExpr::parse, einsum_parser.rs:68-103 -> (inputs, output)."""
    parts = [p.strip() for p in expr.strip().split("->", 1)]
    inputs = [_strip_ws(t) for t in parts[0].split(",")]
    if any(not _is_valid_term(t) for t in inputs):
        raise EinsumError("InvalidValue", "Input term is invalid")
    if len(parts) > 1:
        output = _strip_ws(parts[1])
    else:  # default_output, :191-228: "..." then the letters used exactly once, ASCII order
        letters = [c for t in inputs for c in t if c.isalpha()]
        output = ("..." if any("..." in t for t in inputs) else "") + "".join(sorted(c for c in set(letters) if letters.count(c) == 1))
    if not _is_valid_term(output):
        raise EinsumError("InvalidValue", "Output term is invalid")
    if any(c != "." and output.count(c) > 1 for c in output):
        raise EinsumError("InvalidValue", "Einsum output term contains repeated labels")
    if any(c.isalpha() and not any(c in t for t in inputs) for c in output):
        raise EinsumError("InvalidValue", "Einsum output term contains a label not present in any input term")
    return inputs, output


def validate_inputs(inputs, ndims) -> int:
    """EinsumExpr::validate_inputs, einsum_parser.rs:109-165 with the error mapping of einsum.rs:69-86."""
    if len(ndims) != len(inputs):
        raise EinsumError("InvalidValue", "Number of terms in Einsum equation does not match input tensor count")
    broadcast = None
    for term, ndim in zip(inputs, ndims):
        has = "..." in term
        non_b = len(term) - 3 if has else len(term)
        if not (ndim >= non_b if has else ndim == non_b):
            raise EinsumError("InvalidValue", "Einsum term dimension count does not match input tensor")
        if ndim > MAX_DIMS:
            raise EinsumError("UnsupportedValue", "Einsum input or term has too many dimensions")
        if has:
            if broadcast is None:
                broadcast = ndim - non_b
            elif broadcast != ndim - non_b:
                raise EinsumError("InvalidValue", "Number of broadcast dims does not match across inputs")
    return broadcast or 0


def expand_ellipsis(term: str, n: int) -> str:  # einsum_parser.rs:254-265
    if "..." in term:
        lhs, rhs = term.split("...", 1)
        return lhs + "".join(chr(ord("0") + i) for i in range(n)) + rhs
    return term


# ------------------------------------------------------------------------------------------ path
def _unique(term):
    return [c for i, c in enumerate(term) if c not in term[:i]]


def einsum_path(inputs, output, broadcast_ndim):
    """einsum.rs:605-692 -> [(lhs_term, lhs_src, rhs_term | None, rhs_src, step_output)]; src = input index or 'prev'."""
    out = expand_ellipsis(output, broadcast_ndim)
    terms = [expand_ellipsis(t, broadcast_ndim) for t in inputs]
    if len(terms) == 1:
        return [(terms[0], 0, None, None, out)]
    if len(terms) == 2:
        return [(terms[0], 0, terms[1], 1, out)]
    remaining = {}
    for t in terms:
        for d in _unique(t):
            if d not in out:
                remaining[d] = remaining.get(d, 0) + 1

    def subtract(t):
        for d in _unique(t):
            if d in remaining:
                remaining[d] -= 1

    def step_output(a, b):
        o = ""
        for d in a + b:
            if d not in o and (d in out or remaining.get(d, 0) > 0):
                o += d
        return o
    subtract(terms[0])
    subtract(terms[1])
    nxt = step_output(terms[0], terms[1])
    steps = [(terms[0], 0, terms[1], 1, nxt)]
    rest = terms[2:]
    for i, t in enumerate(rest):
        subtract(t)
        prev = nxt
        nxt = out if i == len(rest) - 1 else step_output(prev, t)
        steps.append((prev, "prev", t, i + 2, nxt))
    return steps


# ------------------------------------------------------------------------------------------ arithmetic leaves
def reduce_sum(x, axes, keep_dims=False, lanes=ref.LANES):
    """reduce(), reduce.rs:414-520 with OptimizedSumKernel (:1101-1106): each output element is vecmath::Sum of the slice
    spanned by the reduced axes (kept in their original relative order, :489-492), packed row-major when strided."""
    x = np.asarray(x, np.float32)
    if x.ndim == 0:
        return x.copy()
    axes = sorted({a + x.ndim if a < 0 else a for a in axes}) if axes else list(range(x.ndim))  # resolve_axes: sorted, deduplicated
    keep = [d for d in range(x.ndim) if d not in axes]
    xp = np.transpose(x, keep + axes)
    kshape = tuple(x.shape[d] for d in keep)
    rows = xp.reshape(int(np.prod(kshape, dtype=np.int64)), -1)  # packs (copies) when strided, same element order
    rows = np.ascontiguousarray(rows)
    out = np.empty(rows.shape[0], np.float32)
    L = ref.lib()
    for r in range(rows.shape[0]):
        out[r] = L.rto_simd_sum(rows[r].ctypes.data_as(C.c_void_p), C.c_int64(rows.shape[1]), C.c_int(lanes))
    if keep_dims:
        return out.reshape([1 if d in axes else x.shape[d] for d in range(x.ndim)])
    return out.reshape(kshape)


def reduce_mean(x, axes, keep_dims=False, lanes=ref.LANES):
    """reduce_mean, reduce.rs:523-541: Sum(slice) / slice.len() as f32 per output element (an empty slice: 0 / 0 = NaN)."""
    x = np.asarray(x, np.float32)
    s = reduce_sum(x, axes, keep_dims, lanes)
    if x.ndim == 0:
        return s
    ax = sorted({a + x.ndim if a < 0 else a for a in axes}) if axes else list(range(x.ndim))
    n = np.float32(int(np.prod([x.shape[a] for a in ax], dtype=np.int64)))
    with np.errstate(invalid="ignore", divide="ignore"):
        return (s / n).astype(np.float32)


def _matmul(a, b):
    """matmul(), src/ops/matmul.rs:208-385 on views (ref.matmul_f32 folds [A,M,K]x[K,N] into one GEMM like :266-297)."""
    pa, pb = a.shape[:-2], b.shape[:-2]
    if int(np.prod(pa, dtype=np.int64)) > 1 and int(np.prod(pb, dtype=np.int64)) == 1:
        pre = np.broadcast_shapes(pa, pb)
        out = ref.gemm_f32(np.ascontiguousarray(a).reshape(-1, a.shape[-1]), b.reshape(b.shape[-2:]))
        return out.reshape(*pre, a.shape[-2], b.shape[-1])
    pre = np.broadcast_shapes(pa, pb)
    out_shape = tuple(pre) + (a.shape[-2], b.shape[-1])
    if int(np.prod(out_shape, dtype=np.int64)) == 0:
        return np.zeros(out_shape, np.float32)
    ab = np.broadcast_to(a, tuple(pre) + a.shape[-2:])
    bb = np.broadcast_to(b, tuple(pre) + b.shape[-2:])
    out = np.empty(out_shape, np.float32)
    for idx in np.ndindex(*pre):
        if a.shape[-1] == 0:
            out[idx] = 0.0
        else:
            out[idx] = ref.gemm_f32(ab[idx], bb[idx])
    return out


# ------------------------------------------------------------------------------------------ one step
def _take_diagonals(term, x):  # einsum.rs:124-162
    shape, strides, uniq = [], [], ""
    for i, label in enumerate(term):
        if label in uniq:
            continue
        uniq += label
        size, st = x.shape[i], 0
        for k, other in enumerate(term):
            if other != label:
                continue
            if x.shape[k] != size:
                raise EinsumError("InvalidValue", "Dimension sizes for repeated labels in term do not match")
            st += x.strides[k]
        shape.append(size)
        strides.append(st)
    return uniq, np.lib.stride_tricks.as_strided(x, shape=shape, strides=strides, writeable=False)


def _sum_lone_dims(view, term, other, output, lanes):  # einsum.rs:168-190
    lone = [i for i, c in enumerate(term) if c not in other and c not in output]
    new_term = "".join(c for c in term if c in other or c in output)
    return (new_term, view) if not lone else (new_term, reduce_sum(view, lone, False, lanes))


def _reduced_dims(lhs, rhs, output):  # einsum.rs:227-235
    dims = []
    for c in lhs + rhs:
        if c not in output and c not in dims:
            dims.append(c)
    return dims


def _permute_insert(x, in_order, out_order):  # einsum.rs:414-442
    assert len(in_order) == x.ndim and all(in_order.count(c) == 1 and out_order.count(c) == 1 for c in in_order)
    perm = [in_order.index(c) for c in out_order if c in in_order]
    p = np.transpose(x, perm)
    for i, c in enumerate(out_order):
        if c not in in_order:
            p = np.expand_dims(p, i)
    return p


def _broadcast_size(a, b):  # einsum.rs:197-205
    if a == b:
        return a
    if a == 1 or b == 1:
        return b if a == 1 else a
    raise EinsumError("IncompatibleInputShapes", "Einsum label has different sizes in different terms")


def _expand_dim(view, size, from_end):  # einsum.rs:210-223
    dim = view.ndim - from_end
    if view.shape[dim] == size:
        return view
    shape = list(view.shape)
    shape[dim] = size
    return np.ascontiguousarray(np.broadcast_to(view, shape))


def _einsum_matmul(x, y, term1, term2, output, k):  # einsum.rs:449-537
    n = next((c for c in reversed(term2) if c not in term1), INSERTED_N)
    m = next((c for c in reversed(term1) if c not in term2), INSERTED_M)
    batch = ""
    for c in term1 + term2:
        if c not in (k, m, n) and c not in batch:
            batch += c
    out_order = batch + (m if m != INSERTED_M else "") + (n if n != INSERTED_N else "")
    xp = _permute_insert(x, term1, batch + m + k)
    yp = _permute_insert(y, term2, batch + k + n)
    ks = _broadcast_size(xp.shape[-1], yp.shape[-2])
    xp = _expand_dim(xp, ks, 1)
    yp = _expand_dim(yp, ks, 2)
    out = _matmul(xp, yp)
    if m == INSERTED_M:
        out = out.reshape(out.shape[:-2] + out.shape[-1:])
    if n == INSERTED_N:
        out = out.reshape(out.shape[:-1])
    if out_order == output:
        return out
    return np.ascontiguousarray(_permute_insert(out, out_order, output))


def _einsum_step(step, x, y, lanes):  # einsum.rs:238-364
    lhs, _, rhs, _, output = step
    lhs_term, x = _take_diagonals(lhs, x)
    if rhs is None:
        red = _reduced_dims(lhs_term, "", output)
        xp = _permute_insert(x, lhs_term, output + "".join(red))
        if not red:
            return np.array(xp, dtype=np.float32, order="C", copy=True)
        return reduce_sum(xp, list(range(xp.ndim - len(red), xp.ndim)), False, lanes)
    rhs_term, y = _take_diagonals(rhs, y)
    lhs_term, x = _sum_lone_dims(x, lhs_term, rhs_term, output, lanes)
    rhs_term, y = _sum_lone_dims(y, rhs_term, lhs_term, output, lanes)
    red = _reduced_dims(lhs_term, rhs_term, output)
    if len(red) == 1:
        return _einsum_matmul(x, y, lhs_term, rhs_term, output, red[0])
    common = output + "".join(red)
    xp = _permute_insert(x, lhs_term, common)
    yp = _permute_insert(y, rhs_term, common)
    if not red:
        return (xp * yp).astype(np.float32)  # mul(), binary_elementwise.rs: one rounded f32 product per element
    xs, ys = list(xp.shape), list(yp.shape)
    for i in range(xp.ndim - len(red), xp.ndim):
        xs[i] = ys[i] = _broadcast_size(xs[i], ys[i])
    xc = np.ascontiguousarray(np.broadcast_to(xp, xs))
    yc = np.ascontiguousarray(np.broadcast_to(yp, ys))
    start = xp.ndim - len(red)
    rsize = int(np.prod(xs[start:], dtype=np.int64))
    xc = xc.reshape(xs[:start] + [rsize])
    yc = yc.reshape(ys[:start] + [rsize])
    simplified = output + MERGED_K
    return _einsum_matmul(xc, yc, simplified, simplified, output, MERGED_K)


def einsum(equation: str, *inputs, lanes=ref.LANES):
    """einsum(), src/ops/einsum.rs:61-108."""
    inputs = [np.asarray(a, np.float32) for a in inputs]
    terms, output = parse(equation)
    b = validate_inputs(terms, [a.ndim for a in inputs])
    out = None
    for step in einsum_path(terms, output, b):
        x = out if step[1] == "prev" else inputs[step[1]]
        y = None if step[2] is None else (out if step[3] == "prev" else inputs[step[3]])
        out = _einsum_step(step, x, y, lanes)
    return np.asarray(out, np.float32)
