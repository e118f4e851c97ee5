"""Einsum on the device (host side of the reference's `Einsum` operator, src/ops/einsum.rs:26-108).

The reference evaluates an equation as a path of two-term steps, each lowered to diagonal views, ReduceSum, a broadcast
Mul or a (batched) MatMul on permuted TensorViews, copying views where its kernels need contiguous data.  Here a view is
(device buffer, shape, element strides) and every kernel on the path reads its operands THROUGH the strides:

  * permutes / inserted axes / diagonals / 1 -> n expansion are stride arithmetic only (einsum.rs:124-162,414-442);
  * ReduceSum runs on the strided view in place (`rten_hip_reduce_sum_strided_f32`, no packing, reduce.rs:470-505);
  * Mul is the stride-0 broadcast kernel (`rten_hip_binary_broadcast_f32`);
  * MatMul is ONE `rten_hip_gemm_f32` launch: M/K/N are element strides, batch labels become the descriptor's two batch
    levels after adjacent axes with compatible strides are merged (a copy happens only when more than two levels remain,
    when a GEMM axis is a broadcast, or where the reference's own algorithm needs the data re-laid -- the multi-label K
    merge, einsum.rs:310-344);
  * the final permutation into the output order (einsum.rs:531-536) is folded into the GEMM's C strides (ldc + the two C
    batch strides) whenever it keeps N innermost; otherwise it is the one copy (`rten_hip_copy_strided_b32`).

What is computed, in which order, is the reference's: the same path (einsum.rs:605-692), the same choice of M / N / batch
labels (:473-492), sums of lone labels before the product (:273-276), `[A, M, K] x [K, N]` folded into one GEMM
(matmul.rs:266-297), so results are bit-identical to the reference for everything but a GEMM with one row (its
ISA-dependent gemv path).  Validation order and messages are einsum_parser.rs:68-166 / einsum.rs:66-86.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as L
from .tensor import DeviceTensor

MAX_DIMS = 10  # einsum_parser.rs:241-245
INS_M, INS_N, MERGED_K = "<", ">", "*"  # einsum.rs:366-380
_WS = " \t\n\r\x0c"


def _err(kind, msg):
    from .ops import OpError
    return OpError(kind, msg)


# ------------------------------------------------------------------------------------------ equation -> path
def _valid_term(t: str) -> bool:
    head, dots, tail = t.partition("...")
    if dots and "..." in tail:
        return False
    return all(c.isascii() and c.isalpha() for c in head + tail)


def parse_equation(equation: str):
    """EinsumExpr::parse (einsum_parser.rs:68-103): -> ([input terms], output term)."""
    lhs, arrow, rhs = equation.strip().partition("->")
    terms = ["".join(c for c in t if c not in _WS) for t in lhs.strip().split(",")]
    if not all(_valid_term(t) for t in terms):
        raise _err("InvalidValue", "Input term is invalid")
    if arrow:
        out = "".join(c for c in rhs if c not in _WS)
    else:
        count = {}
        for t in terms:
            for c in t:
                if c != ".":
                    count[c] = count.get(c, 0) + 1
        out = ("..." if any("..." in t for t in terms) else "") + "".join(c for c in sorted(count) if count[c] == 1)
    if not _valid_term(out):
        raise _err("InvalidValue", "Output term is invalid")
    letters = out.replace(".", "")
    if len(set(letters)) != len(letters):
        raise _err("InvalidValue", "Einsum output term contains repeated labels")
    if any(not any(c in t for t in terms) for c in letters):
        raise _err("InvalidValue", "Einsum output term contains a label not present in any input term")
    return terms, out


def broadcast_ndim(terms, ndims) -> int:
    """EinsumExpr::validate_inputs (einsum_parser.rs:109-165), errors as einsum.rs:69-86."""
    if len(ndims) != len(terms):
        raise _err("InvalidValue", "Number of terms in Einsum equation does not match input tensor count")
    b = None
    for t, nd in zip(terms, ndims):
        dots = "..." in t
        named = len(t) - 3 * dots
        if (nd < named) if dots else (nd != named):
            raise _err("InvalidValue", "Einsum term dimension count does not match input tensor")
        if nd > MAX_DIMS:
            raise _err("UnsupportedValue", "Einsum input or term has too many dimensions")
        if dots:
            if b is not None and b != nd - named:
                raise _err("InvalidValue", "Number of broadcast dims does not match across inputs")
            b = nd - named
    return b or 0


def _expand(term, n):
    return term.replace("...", "".join(str(i) for i in range(n)))


def _dedup(s):
    return "".join(dict.fromkeys(s))


def plan_path(terms, out, bdims):
    """einsum_path (einsum.rs:605-692): -> [(lhs term, lhs source, rhs term | None, rhs source, step output)] where a
    source is an input index or -1 for the previous step's result."""
    out = _expand(out, bdims)
    terms = [_expand(t, bdims) for t in terms]
    if len(terms) <= 2:
        return [(terms[0], 0, terms[1] if len(terms) == 2 else None, 1, out)]
    pending = {}  # reduced label -> number of terms that still have to consume it
    for t in terms:
        for c in _dedup(t):
            if c not in out:
                pending[c] = pending.get(c, 0) + 1

    def consume(t):
        for c in _dedup(t):
            if c in pending:
                pending[c] -= 1

    def keep(a, b):
        return "".join(c for c in _dedup(a + b) if c in out or pending.get(c, 0) > 0)
    consume(terms[0])
    consume(terms[1])
    cur = keep(terms[0], terms[1])
    steps = [(terms[0], 0, terms[1], 1, cur)]
    for i in range(2, len(terms)):
        consume(terms[i])
        nxt = out if i == len(terms) - 1 else keep(cur, terms[i])
        steps.append((cur, -1, terms[i], i, nxt))
        cur = nxt
    return steps


# ------------------------------------------------------------------------------------------ strided device views
class View:
    """(buffer, shape, element strides).  Stride 0 = broadcast / inserted axis; a diagonal is the sum of its axes' strides."""

    __slots__ = ("t", "shape", "strides")

    def __init__(self, t: DeviceTensor, shape=None, strides=None):
        self.t = t
        self.shape = list(t.shape if shape is None else shape)
        if strides is None:
            strides, acc = [0] * len(self.shape), 1
            for d in range(len(self.shape) - 1, -1, -1):
                strides[d] = acc
                acc *= self.shape[d]
        self.strides = list(strides)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def is_contiguous(self):
        acc = 1
        for n, s in zip(reversed(self.shape), reversed(self.strides)):
            if n != 1 and s != acc:
                return False
            acc *= n
        return True

    def relabel(self, have: str, want: str) -> "View":
        """permute_and_insert_axes (einsum.rs:414-442): axes of `have` reordered as in `want`; labels missing from
        `have` become 1-sized axes."""
        assert len(have) == len(self.shape) and all(have.count(c) == 1 and want.count(c) == 1 for c in have)
        shape, strides = [], []
        for c in want:
            i = have.find(c)
            shape.append(self.shape[i] if i >= 0 else 1)
            strides.append(self.strides[i] if i >= 0 else 0)
        return View(self.t, shape, strides)

    def expanded(self, shape) -> "View":
        assert len(shape) == len(self.shape)
        st = [0 if (n == 1 and m != 1) else s for n, m, s in zip(self.shape, shape, self.strides)]
        return View(self.t, shape, st)


def _i64(v):
    return (C.c_int64 * max(len(v), 1))(*v)


def _merge_axes(shape, *stride_sets):
    """Drops 1-sized axes and merges neighbours (outer, inner) whose strides satisfy outer == inner * inner_size in every
    operand, so the kernels see the fewest index divisions."""
    dims = [(n, [st[i] for st in stride_sets]) for i, n in enumerate(shape) if n != 1]
    merged = []
    for n, st in dims:
        if merged and all(ps == s * n for ps, s in zip(merged[-1][1], st)):
            merged[-1] = (merged[-1][0] * n, st)
        else:
            merged.append((n, st))
    return [n for n, _ in merged], [[st[k] for _, st in merged] for k in range(len(stride_sets))]


def materialize(ctx, v: View, shape=None) -> DeviceTensor:
    """to_tensor / to_contiguous / expand_to: a contiguous tensor of `shape` (default: the view's own shape)."""
    v = v if shape is None else v.expanded(shape)
    out = DeviceTensor(ctx, v.shape, np.float32)
    if v.size:
        msh, (mst,) = _merge_axes(v.shape, v.strides)
        if len(msh) > 6:
            raise _err("UnsupportedValue", "Einsum view with more than 6 non-mergeable dims is not supported by the device path")
        ctx.call("rten_hip_copy_strided_b32", len(msh), _i64(msh), _i64(mst), v.t.vp, out.vp)
    return out


def reduce_sum(ctx, v: View, axes, mean: bool = False) -> DeviceTensor:
    """reduce_sum(view, axes, keep_dims = false), reduce.rs:414-520: kept axes in order; each output element sums its slice
    over the reduced axes in row-major order (the order the reference packs it in), 16-lane vecmath::Sum order."""
    axes = sorted(axes)
    keep = [d for d in range(len(v.shape)) if d not in axes]
    out = DeviceTensor(ctx, [v.shape[d] for d in keep], np.float32)
    if out.size:
        osh, (ost,) = _merge_axes([v.shape[d] for d in keep], [v.strides[d] for d in keep])
        ish, (ist,) = _merge_axes([v.shape[d] for d in axes], [v.strides[d] for d in axes])
        if len(osh) > 6 or len(ish) > 6:
            raise _err("UnsupportedValue", "Einsum reduction over more than 6 non-mergeable dims is not supported by the device path")
        ctx.call("rten_hip_reduce_mean_strided_f32" if mean else "rten_hip_reduce_sum_strided_f32", len(osh), _i64(osh), _i64(ost), len(ish), _i64(ish), _i64(ist), v.t.vp, out.vp)
    return out


def mul(ctx, a: View, b: View) -> DeviceTensor:
    """mul() with numpy broadcasting (binary_elementwise.rs:58-170) on two views of equal rank."""
    try:
        shape = list(np.broadcast_shapes(tuple(a.shape), tuple(b.shape)))
    except ValueError:
        raise _err("IncompatibleInputShapes", "Cannot broadcast inputs")
    out = DeviceTensor(ctx, shape, np.float32)
    if out.size:
        msh, (sa, sb) = _merge_axes(shape, a.expanded(shape).strides, b.expanded(shape).strides)
        if len(msh) > 6:
            raise _err("UnsupportedValue", "broadcasting over more than 6 dims is not supported by the device path")
        ctx.call("rten_hip_binary_broadcast_f32", 1, len(msh), _i64(msh), _i64(sa), _i64(sb), a.t.vp, b.t.vp, out.vp)
    return out


def matmul(ctx, a: View, b: View, into: View | None = None) -> DeviceTensor:
    """matmul() of [batch.., M, K] x [batch.., K, N] views of equal rank (src/ops/matmul.rs:208-385): numpy batch
    broadcasting; `[A, M, K] x [K, N]` is one GEMM of A*M rows (:266-297).

    `into`: a view of the caller's output tensor with this product's [batch.., M, N] axes (N contiguous): the GEMM then
    writes its rows straight into the permuted output through ldc and the C batch strides.  Returns None when those
    strides do not fit the descriptor's two batch levels (the caller falls back to product + copy)."""
    m, k, n = a.shape[-2], a.shape[-1], b.shape[-1]
    if k != b.shape[-2]:
        raise _err("IncompatibleInputShapes", "Columns of first matrix does not match rows of second matrix")
    try:
        pre = list(np.broadcast_shapes(tuple(a.shape[:-2]), tuple(b.shape[:-2])))
    except ValueError:
        raise _err("IncompatibleInputShapes", "Cannot broadcast shapes")
    if into is None:
        out = DeviceTensor(ctx, pre + [m, n], np.float32)
        cv = View(out)
    else:
        out, cv = into.t, into
    if out.size == 0:
        return out
    if k == 0:
        ctx.call("rten_hip_memset", out.vp, 0, C.c_size_t(out.nbytes))
        return out
    ldc = max(cv.strides[-2], n)  # a single row never uses it
    # a GEMM axis that is a broadcast (stride 0 with size > 1) is re-laid, as expand_dim does (einsum.rs:210-223)
    if any(s == 0 and sz > 1 for s, sz in zip(a.strides[-2:], a.shape[-2:])):
        a = View(materialize(ctx, a))
    if any(s == 0 and sz > 1 for s, sz in zip(b.strides[-2:], b.shape[-2:])):
        b = View(materialize(ctx, b))
    na = int(np.prod(a.shape[:-2], dtype=np.int64))
    nb = int(np.prod(b.shape[:-2], dtype=np.int64))
    if na > 1 and nb == 1 and len(_merge_axes(pre + [m], cv.strides[:-1])[0]) <= 1:
        # rows of all A matrices form one [A*M, K] matrix when the (batch.., M) axes merge into one stride; else re-lay A
        rsh, (rst,) = _merge_axes(a.shape[:-1], a.strides[:-1])
        if len(rsh) > 1:
            a = View(materialize(ctx, a))
            rsh, (rst,) = _merge_axes(a.shape[:-1], a.strides[:-1])
        crows = _merge_axes(pre + [m], cv.strides[:-1])[1][0]
        d = L.gemm_desc(na * m, n, k, rst[0] if rst else 0, a.strides[-1], b.strides[-2], b.strides[-1], max(crows[0], n) if crows else n)
        ctx.call("rten_hip_gemm_f32", C.byref(d), a.t.vp, b.t.vp, None, out.vp)
        return out
    ea, eb = a.expanded(pre + [m, k]), b.expanded(pre + [k, n])
    bsh, (sa, sb, sc) = _merge_axes(pre, ea.strides[:-2], eb.strides[:-2], cv.strides[:-2])
    if len(bsh) > 2:  # more batch levels than the descriptor has: re-lay the operand(s) that do not merge
        if len(_merge_axes(pre, ea.strides[:-2])[0]) > 1 or any(s == 0 for s in sa):
            ea = View(materialize(ctx, ea))
        if len(_merge_axes(pre, eb.strides[:-2])[0]) > 1 or any(s == 0 for s in sb):
            eb = View(materialize(ctx, eb))
        bsh, (sa, sb, sc) = _merge_axes(pre, ea.strides[:-2], eb.strides[:-2], cv.strides[:-2])
        if len(bsh) > 2:
            assert into is not None  # a contiguous C always merges with re-laid operands
            return None
    batch = int(np.prod(bsh, dtype=np.int64)) if bsh else 1
    if len(bsh) == 2:
        d = L.gemm_desc(m, n, k, ea.strides[-2], ea.strides[-1], eb.strides[-2], eb.strides[-1], ldc, batch, sa[0], sb[0],
                        sc[0], batch_inner=bsh[1], a_bsi=sa[1], b_bsi=sb[1], c_bsi=sc[1])
    else:
        d = L.gemm_desc(m, n, k, ea.strides[-2], ea.strides[-1], eb.strides[-2], eb.strides[-1], ldc, batch,
                        sa[0] if sa else 0, sb[0] if sb else 0, sc[0] if sc else 0)
    ctx.call("rten_hip_gemm_f32", C.byref(d), ea.t.vp, eb.t.vp, None, out.vp)
    return out


# ------------------------------------------------------------------------------------------ one step of the path
def _diagonals(term: str, v: View):
    """take_diagonals (einsum.rs:124-162): repeated labels of a term collapse to one axis whose stride is the sum."""
    labels = _dedup(term)
    shape, strides = [], []
    for c in labels:
        idx = [i for i, t in enumerate(term) if t == c]
        if any(v.shape[i] != v.shape[idx[0]] for i in idx):
            raise _err("InvalidValue", "Dimension sizes for repeated labels in term do not match")
        shape.append(v.shape[idx[0]])
        strides.append(sum(v.strides[i] for i in idx))
    return labels, View(v.t, shape, strides)


def _drop_lone(ctx, v: View, term: str, other: str, out: str):
    """sum_lone_dims (einsum.rs:168-190)."""
    lone = [i for i, c in enumerate(term) if c not in other and c not in out]
    kept = "".join(c for c in term if c in other or c in out)
    return (kept, v) if not lone else (kept, View(reduce_sum(ctx, v, lone)))


def _bsize(a, b):
    if a == b or b == 1:
        return a
    if a == 1:
        return b
    raise _err("IncompatibleInputShapes", "Einsum label has different sizes in different terms")


def _contract(ctx, x: View, y: View, tx: str, ty: str, out: str, kl: str) -> DeviceTensor:
    """einsum_matmul (einsum.rs:449-537)."""
    nl = next((c for c in reversed(ty) if c not in tx), INS_N)
    ml = next((c for c in reversed(tx) if c not in ty), INS_M)
    batch = "".join(c for c in _dedup(tx + ty) if c not in (kl, ml, nl))
    xv, yv = x.relabel(tx, batch + ml + kl), y.relabel(ty, batch + kl + nl)
    ks = _bsize(xv.shape[-1], yv.shape[-2])
    xv = xv.expanded(xv.shape[:-1] + [ks])
    yv = yv.expanded(yv.shape[:-2] + [ks, yv.shape[-1]])
    full = batch + ml + nl
    order = "".join(c for c in full if c not in (INS_M, INS_N))
    if order != out and (nl == INS_N or out[-1] == nl):
        # the output permutation keeps N innermost: the GEMM writes the permuted tensor directly (C strides), no copy
        size = {c: max(sx, sy) if min(sx, sy) == 1 else sx for c, sx, sy in zip(batch, xv.shape, yv.shape)}
        size[ml], size[nl] = xv.shape[-2], yv.shape[-1]
        final = DeviceTensor(ctx, [size[c] for c in out], np.float32)
        if matmul(ctx, xv, yv, into=View(final).relabel(out, full)) is not None:
            return final
    r = matmul(ctx, xv, yv)
    r = r.reshape([s for s, c in zip(r.shape, full) if c not in (INS_M, INS_N)])
    return r if order == out else materialize(ctx, View(r).relabel(order, out))


def run_step(ctx, step, x: View, y: View | None) -> DeviceTensor:
    """einsum_step (einsum.rs:238-364)."""
    tx, _, ty, _, out = step
    tx, x = _diagonals(tx, x)
    if ty is None:
        red = "".join(c for c in tx if c not in out)
        xv = x.relabel(tx, out + red)
        if not red:
            return materialize(ctx, xv)
        return reduce_sum(ctx, xv, list(range(len(out), len(out) + len(red))))
    ty, y = _diagonals(ty, y)
    tx, x = _drop_lone(ctx, x, tx, ty, out)
    ty, y = _drop_lone(ctx, y, ty, tx, out)
    red = "".join(c for c in _dedup(tx + ty) if c not in out)
    if len(red) == 1:
        return _contract(ctx, x, y, tx, ty, out, red)
    xv, yv = x.relabel(tx, out + red), y.relabel(ty, out + red)
    if not red:
        return mul(ctx, xv, yv)
    # several reduced labels: re-laid next to each other and merged into one K (einsum.rs:310-344)
    xs, ys = list(xv.shape), list(yv.shape)
    for i in range(len(out), len(out) + len(red)):
        xs[i] = ys[i] = _bsize(xs[i], ys[i])
    ksz = int(np.prod(xs[len(out):], dtype=np.int64))
    xc = View(materialize(ctx, xv, xs).reshape(xs[:len(out)] + [ksz]))
    yc = View(materialize(ctx, yv, ys).reshape(ys[:len(out)] + [ksz]))
    return _contract(ctx, xc, yc, out + MERGED_K, out + MERGED_K, out, MERGED_K)


def einsum(ctx, inputs, equation: str) -> DeviceTensor:
    """einsum() (einsum.rs:61-108)."""
    terms, out = parse_equation(equation)
    bdims = broadcast_ndim(terms, [len(t.shape) for t in inputs])
    result = None
    for step in plan_path(terms, out, bdims):
        x = result if step[1] < 0 else inputs[step[1]]
        y = None if step[2] is None else (result if step[3] < 0 else inputs[step[3]])
        result = run_step(ctx, step, View(x), None if y is None else View(y))
    return result
