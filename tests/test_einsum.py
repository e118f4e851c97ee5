"""Einsum / ReduceSum: the oracle pinned to the reference's own test tables, the device planner's host logic on a simulated
device (CPU), and GPU parity through the C ABI.

Reference tables restated here: src/ops/einsum.rs:705-1311 (`test_einsum`: equations, operands, expected values and
error strings), :1313-1498 (`test_einsum_path`), rten-shape-inference/src/einsum_parser.rs:277-557 (parser).

* not gpu: (1) oracle/einsum.py against the reference's literal expectations -- the operands are small integers, so every
  expected value is an exact integer that any summation order reproduces: equality is exact; (2) rten_amd/einsum.py's
  planner run on a SIMULATED device whose five kernels are numpy / oracle restatements of the C ABI contracts: every
  stride, merge and batch-level decision is checked bit for bit against the oracle without a GPU.
* gpu: the same tables plus seeded random shapes through librten_hip.so, bit-exact against the oracle; an evaluation
  that launched a GEMM with ONE row (the reference's ISA-dependent gemv path) is held to rtol 1e-5 instead.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import einsum as OE
from oracle import ref
from rten_amd import einsum as PE
from rten_amd import lib as L
from rten_amd import ops
from rten_amd.tensor import DeviceTensor


def arange(lo, hi, shape=None):
    a = np.arange(lo, hi, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


# operands of test_einsum (einsum.rs:714-804)
SCALAR = np.array(2.5, np.float32)
VEC_A, VEC_B = arange(1, 10), arange(1, 5)
MAT_A = arange(1, 7, (2, 3))
MAT_B = arange(1, 13, (3, 4))
MAT_C = (MAT_A.astype(np.float64) @ MAT_B.astype(np.float64)).astype(np.float32)  # matmul_ab: small integers, exact
SQUARE = arange(1, 10, (3, 3))
CUBE = arange(1, 28, (3, 3, 3))
BHWC = MAT_A.reshape(1, 1, 2, 3)
HCK = MAT_B.reshape(1, 3, 4)
IJK = np.zeros((10, 5, 8), np.float32)
ROW_1x3 = arange(1, 4, (1, 3))
MAT_4x3 = arange(1, 13, (4, 3))
EMPTY_0x3 = np.zeros((0, 3), np.float32)
ABF = arange(1, 25, (2, 3, 4))
FCD = arange(1, 121, (4, 5, 6))


def f64(eq, *xs):
    return np.einsum(eq.replace(" ", ""), *[np.asarray(x, np.float64) for x in xs]).astype(np.float32)


# (equation, operands, expected) -- expected is the reference's literal where it gives one, else the exact integer result
OK_CASES = [
    ("ij->ij", [MAT_A], MAT_A), ("i j -> i j", [MAT_A], MAT_A), ("ij->ji", [MAT_A], MAT_A.T), (" ij -> ji ", [MAT_A], MAT_A.T),
    ("ba", [MAT_A], MAT_A.T), ("ij->i", [MAT_A], MAT_A.sum(1)), ("abf->a", [ABF], ABF.sum((1, 2))), ("abf->f", [ABF], ABF.sum((0, 1))),
    ("i,j->ij", [VEC_A, VEC_B], np.outer(VEC_A, VEC_B)), ("ij,kl->ijkl", [MAT_A, MAT_B], MAT_A[:, :, None, None] * MAT_B[None, None]),
    ("a,b->ba", [VEC_A, VEC_B], np.outer(VEC_B, VEC_A)), ("ij,jk->ik", [MAT_A, MAT_B], MAT_C), ("ij,jk", [MAT_A, MAT_B], MAT_C),
    ("ji,kj->ik", [MAT_A.T, MAT_B.T], MAT_C), ("ij,jk->ki", [MAT_A, MAT_B], MAT_C.T),
    ("bhwc,hkc->bhwk", [BHWC, HCK.transpose(0, 2, 1)], MAT_C.reshape(1, 1, 2, 4)),
    ("mc,hck->hmk", [MAT_A, HCK], MAT_C.reshape(1, 2, 4)), ("mc,hck->khm", [MAT_A, HCK], MAT_C.reshape(1, 2, 4).transpose(2, 0, 1)),
    ("c,hck->hk", [MAT_A[0], HCK], MAT_C[:1]), ("abf,fcd->abcd", [ABF, FCD], f64("abf,fcd->abcd", ABF, FCD)),
    ("ij,ik->ik", [MAT_A, MAT_C], MAT_A.sum(1, keepdims=True) * MAT_C), ("ik,ij->ik", [MAT_C, MAT_A], MAT_A.sum(1, keepdims=True) * MAT_C),
    ("af,abf->a", [MAT_C, ABF], (MAT_C * ABF.sum(1)).sum(1)), ("i,i->", [VEC_A, VEC_A], np.float32(285.0)),
    ("ij,j->i", [MAT_A, MAT_B[:, 0]], MAT_C[:, 0]), ("j,jk->k", [MAT_A[0], MAT_B], MAT_C[0]),
    ("ij,ij->", [MAT_A, MAT_A], np.float32(91.0)), ("bhwc,bhwc->", [BHWC, BHWC], np.float32(91.0)), ("ij,ji->", [MAT_A, MAT_A.T], np.float32(91.0)),
    ("ij,ij->", [MAT_A[:1], MAT_A], (MAT_A[:1] * MAT_A).sum()), ("ij,ij->j", [ROW_1x3, MAT_4x3], np.float32([22, 52, 90])),
    ("ij,ij->j", [MAT_4x3, ROW_1x3], np.float32([22, 52, 90])), ("ij,ij->", [ROW_1x3, EMPTY_0x3], np.float32(0.0)),
    ("ij,j->", [MAT_A, MAT_B[:, 0]], (MAT_A * MAT_B[:, 0]).sum()), ("", [SCALAR], SCALAR), ("->", [SCALAR], SCALAR),
    ("C,MCN->MN", [MAT_A[0], HCK], MAT_C[:1]), ("IJK,IJK->K", [ABF, ABF], (ABF * ABF).sum((0, 1))), ("iI->Ii", [MAT_A], MAT_A.T),
    ("II->I", [SQUARE], np.float32([1, 5, 9])), ("aBc", [ABF], ABF.transpose(1, 0, 2)), ("I...J->J...I", [IJK], IJK.T),
    ("ii->i", [SQUARE], np.float32([1, 5, 9])), ("iii->i", [CUBE], np.float32([1, 14, 27])), ("ii->", [SQUARE], np.float32(15.0)),
    ("i,i,i->", [VEC_A, VEC_A, VEC_A], np.float32(2025.0)), ("...", [MAT_A], MAT_A), ("i...j->i...j", [MAT_A], MAT_A),
    ("i...j->j...i", [IJK], IJK.T), ("i...j", [IJK], IJK.transpose(1, 0, 2)), ("...i->...", [MAT_A], MAT_A.sum(-1)),
    ("f,fc...->c...", [MAT_B[0], FCD], f64("f,fcd->cd", MAT_B[0], FCD)), ("af,f...->a...", [MAT_C, FCD], f64("af,fcd->acd", MAT_C, FCD)),
]

ERR_CASES = [
    ("ij,jk->ik", [MAT_A], "InvalidValue", "Number of terms in Einsum equation does not match input tensor count"),
    ("", [], "InvalidValue", "Number of terms in Einsum equation does not match input tensor count"),
    ("i1j", [MAT_A], "InvalidValue", "Input term is invalid"), ("i.j", [MAT_A], "InvalidValue", "Input term is invalid"),
    ("i...j...", [MAT_A], "InvalidValue", "Input term is invalid"),
    ("ii->i", [MAT_A], "InvalidValue", "Dimension sizes for repeated labels in term do not match"),
    ("ij,jk->i.k", [MAT_A, MAT_B], "InvalidValue", "Output term is invalid"),
    ("ij,jk->IK", [MAT_A, MAT_B], "InvalidValue", "Einsum output term contains a label not present in any input term"),
    ("ij->ii", [MAT_A], "InvalidValue", "Einsum output term contains repeated labels"),
    ("ij", [VEC_A], "InvalidValue", "Einsum term dimension count does not match input tensor"),
    ("i...j", [VEC_A], "InvalidValue", "Einsum term dimension count does not match input tensor"),
    ("abcdefghijkl...", [np.zeros((1,) * 12, np.float32)], "UnsupportedValue", "Einsum input or term has too many dimensions"),
    ("...", [np.zeros((1,) * 11, np.float32)], "UnsupportedValue", "Einsum input or term has too many dimensions"),
    ("...,...->...", [VEC_A, MAT_A], "InvalidValue", "Number of broadcast dims does not match across inputs"),
    ("ij,jk->ik", [MAT_A, MAT_4x3], "IncompatibleInputShapes", "Einsum label has different sizes in different terms"),
]

# test_einsum_path (einsum.rs:1329-1492); source None = previous output
PATH_CASES = [
    ("i->i", 0, [("i", 0, None, None, "i")]),
    ("ij,jk->ik", 0, [("ij", 0, "jk", 1, "ik")]),
    ("ab,bc,cd,de->ea", 0, [("ab", 0, "bc", 1, "ac"), ("ac", None, "cd", 2, "ad"), ("ad", None, "de", 3, "ea")]),
    ("ab,cd,ef", 0, [("ab", 0, "cd", 1, "abcd"), ("abcd", None, "ef", 2, "abcdef")]),
    ("ii,j,i->", 0, [("ii", 0, "j", 1, "i"), ("i", None, "i", 2, "")]),
    ("ii,i,j->", 0, [("ii", 0, "i", 1, ""), ("", None, "j", 2, "")]),
    ("ii,i,i->i", 0, [("ii", 0, "i", 1, "i"), ("i", None, "i", 2, "i")]),
    ("i...j->j...i", 3, [("i012j", 0, None, None, "j012i")]),
    ("...i,...j,...k->...ijk", 2, [("01i", 0, "01j", 1, "01ij"), ("01ij", None, "01k", 2, "01ijk")]),
]

# parser table (einsum_parser.rs:277-400): equation -> (inputs, output)
PARSE_CASES = [("ij->ij", ["ij"], "ij"), ("ij,jk->ik", ["ij", "jk"], "ik"), (" i j , j k -> i k ", ["ij", "jk"], "ik"),
               ("ij,jk", ["ij", "jk"], "ik"), ("...ij", ["...ij"], "...ij"), ("i,i", ["i", "i"], ""), ("aBc", ["aBc"], "Bac"),
               ("", [""], ""), ("->", [""], "")]

# shapes beyond the reference's table: attention-style batched products, >2 batch levels, multi-label K, ragged sizes
RANDOM_CASES = [
    ("bhqd,bhkd->bhqk", [(2, 3, 17, 8), (2, 3, 19, 8)]), ("bhqk,bhkd->bhqd", [(2, 3, 17, 19), (2, 3, 19, 8)]),
    ("bqhd,bkhd->bhqk", [(2, 17, 3, 8), (2, 19, 3, 8)]), ("bij,bjk,bkl->bil", [(3, 4, 5), (3, 5, 6), (3, 6, 7)]),
    ("abcij,abcjk->abcik", [(2, 3, 2, 5, 7), (2, 3, 2, 7, 4)]), ("abcij,acjk->abcik", [(2, 3, 2, 5, 7), (2, 2, 7, 4)]),
    ("acbij,bajk->abcik", [(2, 2, 3, 5, 7), (3, 2, 7, 4)]), ("ijkl,klm->ijm", [(3, 4, 5, 6), (5, 6, 7)]), ("ijkl,jkl->i", [(3, 4, 5, 6), (4, 5, 6)]),
    ("ij,ij->i", [(33, 300), (33, 300)]), ("ik,jk->ij", [(70, 513), (45, 513)]), ("ij->j", [(257, 65)]), ("ijk->j", [(9, 70, 11)]),
    ("iij->j", [(6, 6, 40)]), ("iji->j", [(5, 33, 5)]), ("ii,ij->j", [(6, 6), (6, 9)]), ("bnd,bmd,bn->bm", [(2, 5, 8), (2, 6, 8), (2, 5)]),
    ("...ij,...jk->...ik", [(2, 3, 4, 5), (2, 3, 5, 6)]), ("i...,i...->...", [(7, 3, 4), (7, 3, 4)]), ("ab,cd->acbd", [(3, 4), (5, 6)]),
    ("bhwc,hkc->bhwk", [(2, 5, 6, 16), (5, 7, 16)]), ("ij,jk->ik", [(1, 300), (300, 5)]), ("abij,jk->abik", [(2, 3, 1, 40), (40, 5)]),
    ("ij,ij->j", [(1, 5), (300, 5)]), ("aij,ajk->aik", [(1, 4, 5), (3, 5, 6)]),
    # output permutations that keep N innermost go through the GEMM's C strides (no copy)
    ("bhqk,bkhd->bqhd", [(2, 3, 17, 19), (2, 19, 3, 8)]), ("mc,hck->mhk", [(5, 6), (4, 6, 7)]), ("abij,abjk->baik", [(2, 3, 4, 5), (2, 3, 5, 6)]),
    ("abcij,abcjk->cbaik", [(2, 3, 2, 5, 7), (2, 3, 2, 7, 4)]), ("ij,jk,kl->il", [(4, 5), (5, 6), (6, 7)]), ("bsc,chd->bshd", [(3, 10, 24), (24, 3, 8)]),
    ("ak,bk->ba", [(5, 9), (7, 9)]), ("abk,ck->cab", [(2, 3, 9), (4, 9)]),
]


# ------------------------------------------------------------------------------------------------ oracle pinning (CPU)
@pytest.mark.parametrize("eq,xs,want", OK_CASES, ids=[c[0] or "<empty>" for c in OK_CASES])
def test_oracle_matches_reference_table(eq, xs, want):
    got = OE.einsum(eq, *xs)
    want = np.asarray(want, np.float32)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(got, want), (eq, got, want)


@pytest.mark.parametrize("eq,xs,kind,msg", ERR_CASES, ids=[f"{c[0]}|{c[3][:24]}" for c in ERR_CASES])
def test_oracle_errors_match_reference(eq, xs, kind, msg):
    with pytest.raises(OE.EinsumError) as e:
        OE.einsum(eq, *xs)
    assert (e.value.kind, e.value.msg) == (kind, msg)


def test_oracle_path_and_parser_tables():
    for eq, b, want in PATH_CASES:
        terms, out = OE.parse(eq)
        got = [(l, None if ls == "prev" else ls, r, None if rs == "prev" else (rs if r is not None else None), o)
               for l, ls, r, rs, o in OE.einsum_path(terms, out, b)]
        assert got == want, (eq, got)
    for eq, terms, out in PARSE_CASES:
        assert OE.parse(eq) == (terms, out), eq


def test_oracle_reduce_sum_order():
    """reduce_sum == vecmath::Sum over the row-major packed slice; against f64 and the scalar 16-lane restatement."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((5, 37, 9)).astype(np.float32)
    for axes in ([2], [1], [0, 2], [0, 1, 2], [1, 2]):
        got = OE.reduce_sum(x, axes)
        np.testing.assert_allclose(got, x.astype(np.float64).sum(tuple(axes)), rtol=2e-5, atol=1e-5)
    row = rng.standard_normal(1000).astype(np.float32)
    acc = np.zeros(16, np.float32)  # one accumulator vector is the same order as four folded in sequence only for n < 64 ...
    short = row[:50]
    for i, v in enumerate(short):  # ... so pin the short-slice case by hand: lane l sums l, l+16, l+32, then lanes left to right
        acc[i % 16] = np.float32(acc[i % 16] + v)
    s = np.float32(0)
    for l in range(16):
        s = np.float32(s + acc[l])
    assert OE.reduce_sum(short.reshape(1, 50), [1])[0] == s


# ------------------------------------------------------------------------------------------------ planner on a simulated device (CPU)
class SimDevice:
    """The five C-ABI entry points the Einsum path uses, restated with numpy + the oracle's arithmetic on host memory.
    `call` has Context.call's signature, so rten_amd.einsum runs unmodified; kernel launches are recorded."""

    def __init__(self):
        self.heap, self.launches, self.h = {}, [], 1

    @staticmethod
    def _addr(p):
        return p if isinstance(p, int) else (p.value or 0) if p is not None else 0

    def _view(self, addr, shape, strides):
        shape, strides = [int(s) for s in shape], [int(s) for s in strides]
        if any(s == 0 for s in shape):
            return np.zeros(shape, np.float32)
        n = 1 + sum((s - 1) * st for s, st in zip(shape, strides))
        flat = np.ctypeslib.as_array((C.c_float * n).from_address(addr))
        return np.lib.stride_tricks.as_strided(flat, shape=shape, strides=[4 * s for s in strides])

    def call(self, name, *a):
        if name == "rten_hip_malloc":
            buf = np.full(int(a[0].value if hasattr(a[0], "value") else a[0]) + 64, 0xCD, np.uint8)  # poisoned: reads of unwritten memory show
            self.heap[buf.ctypes.data] = buf
            a[1]._obj.value = buf.ctypes.data
            return
        if name == "rten_hip_free":
            self.heap.pop(self._addr(a[0]), None)
            return
        if name in ("rten_hip_memcpy_h2d", "rten_hip_memcpy_d2h", "rten_hip_memcpy_d2d"):
            C.memmove(self._addr(a[0]), self._addr(a[1]), int(a[2].value if hasattr(a[2], "value") else a[2]))
            return
        if name == "rten_hip_memset":
            C.memset(self._addr(a[0]), int(a[1]), int(a[2].value if hasattr(a[2], "value") else a[2]))
            return
        self.launches.append((name, a))
        if name == "rten_hip_copy_strided_b32":
            nd, shape, st, x, y = a
            assert nd <= 6
            self._view(self._addr(y), list(shape)[:nd], _row_major(list(shape)[:nd]))[...] = self._view(self._addr(x), list(shape)[:nd], list(st)[:nd])
        elif name in ("rten_hip_reduce_sum_strided_f32", "rten_hip_reduce_mean_strided_f32"):
            no, osh, ost, ni, ish, ist, x, y = a
            assert no <= 6 and ni <= 6
            osh, ost, ish, ist = list(osh)[:no], list(ost)[:no], list(ish)[:ni], list(ist)[:ni]
            v = self._view(self._addr(x), osh + ish, ost + ist)
            out = self._view(self._addr(y), osh, _row_major(osh))
            red = OE.reduce_mean if "mean" in name else OE.reduce_sum
            out[...] = red(v, list(range(no, no + ni))) if ni else v
        elif name == "rten_hip_binary_broadcast_f32":
            op, nd, shape, sa, sb, x, z, y = a
            assert op == 1 and nd <= 6
            shape = list(shape)[:nd]
            self._view(self._addr(y), shape, _row_major(shape))[...] = self._view(self._addr(x), shape, list(sa)[:nd]) * self._view(self._addr(z), shape, list(sb)[:nd])
        elif name == "rten_hip_gemm_f32":
            d = a[0]._obj
            assert a[3] is None and d.alpha == 1.0 and d.beta == 0.0 and d.ldc >= d.n and d.k > 0
            assert min(d.a_rs, d.a_cs, d.b_rs, d.b_cs) >= 0
            assert not (d.a_rs == 0 and d.m > 1) and not (d.a_cs == 0 and d.k > 1) and not (d.b_rs == 0 and d.k > 1) and not (d.b_cs == 0 and d.n > 1)
            inner = d.batch_inner if d.batch_inner > 1 else 1
            for z in range(d.batch):
                zo, zi = divmod(z, inner)
                A = self._view(self._addr(a[1]) + 4 * (zo * d.a_bs + zi * d.a_bsi), [d.m, d.k], [d.a_rs, d.a_cs])
                B = self._view(self._addr(a[2]) + 4 * (zo * d.b_bs + zi * d.b_bsi), [d.k, d.n], [d.b_rs, d.b_cs])
                self._view(self._addr(a[4]) + 4 * (zo * d.c_bs + zi * d.c_bsi), [d.m, d.n], [d.ldc, 1])[...] = ref.gemm_f32(A, B)
        else:
            raise AssertionError(f"unexpected device call {name}")


def _row_major(shape):
    st, acc = [0] * len(shape), 1
    for i in range(len(shape) - 1, -1, -1):
        st[i] = acc
        acc *= int(shape[i])
    return st


def _random_operands(shapes, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(s).astype(np.float32) for s in shapes]


def _all_value_cases():
    for eq, xs, _ in OK_CASES:
        yield eq, xs
    for i, (eq, shapes) in enumerate(RANDOM_CASES):
        yield eq, _random_operands(shapes, 100 + i)


ALL_VALUE_CASES = list(_all_value_cases())


@pytest.mark.parametrize("eq,xs", ALL_VALUE_CASES, ids=[f"{i}:{c[0] or '<empty>'}" for i, c in enumerate(ALL_VALUE_CASES)])
def test_planner_on_simulated_device(eq, xs):
    sim = SimDevice()
    got = ops.Einsum(eq).run(sim, [DeviceTensor.from_numpy(sim, x) for x in xs])[0]
    want = OE.einsum(eq, *xs)
    assert tuple(got.shape) == want.shape
    g = got.numpy()
    assert np.array_equal(g, want) or (np.isnan(want).all() and np.isnan(g).all()), (eq, np.abs(g - want).max())


def test_planner_lowering_is_strided_not_copied():
    """The MI355X-side design claims of rten_amd/einsum.py: transposed / head-interleaved operands reach the GEMM as strides
    (no copy launches), and only the final output permutation is a copy."""
    def launches(eq, shapes):
        sim = SimDevice()
        ops.Einsum(eq).run(sim, [DeviceTensor.from_numpy(sim, x) for x in _random_operands(shapes, 1)])
        return [n for n, _ in sim.launches]
    assert launches("bqhd,bkhd->bhqk", [(2, 17, 3, 8), (2, 19, 3, 8)]) == ["rten_hip_gemm_f32"]  # [B,S,H,D] heads: two batch levels
    assert launches("ji,kj->ik", [(3, 2), (4, 3)]) == ["rten_hip_gemm_f32"]
    assert launches("ij,jk->ki", [(2, 3), (3, 4)]) == ["rten_hip_gemm_f32", "rten_hip_copy_strided_b32"]  # N not innermost
    assert launches("bhqk,bkhd->bqhd", [(2, 3, 17, 19), (2, 19, 3, 8)]) == ["rten_hip_gemm_f32"]  # permuted output through C strides
    assert launches("mc,hck->mhk", [(5, 6), (4, 6, 7)]) == ["rten_hip_gemm_f32"]
    assert launches("iij->j", [(6, 6, 40)]) == ["rten_hip_reduce_sum_strided_f32"]  # diagonal + reduction in place
    assert launches("ij,ik->ik", [(2, 3), (2, 4)]) == ["rten_hip_reduce_sum_strided_f32", "rten_hip_binary_broadcast_f32"]


@pytest.mark.parametrize("eq,xs,kind,msg", ERR_CASES, ids=[f"{c[0]}|{c[3][:24]}" for c in ERR_CASES])
def test_planner_errors_match_reference(eq, xs, kind, msg):
    sim = SimDevice()
    with pytest.raises(ops.OpError) as e:
        ops.Einsum(eq).run(sim, [DeviceTensor.from_numpy(sim, x) for x in xs])
    assert e.value == ops.OpError(kind, msg)
    assert not sim.launches  # validation precedes every launch


def test_planner_path_and_parser_tables():
    for eq, b, want in PATH_CASES:
        terms, out = PE.parse_equation(eq)
        got = [(l, None if ls < 0 else ls, r, None if (r is None or rs < 0) else rs, o) for l, ls, r, rs, o in PE.plan_path(terms, out, b)]
        assert got == want, (eq, got)
    for eq, terms, out in PARSE_CASES:
        assert PE.parse_equation(eq) == (terms, out), eq
    assert "Einsum" in ops.OpRegistry.with_all_ops().op_types() and "ReduceSum" in ops.OpRegistry.with_all_ops().op_types()


def test_reduce_sum_operator_on_simulated_device():
    x = _random_operands([(4, 70, 3, 5)], 9)[0]
    for axes, keep in (([1], True), ([-1, 1], False), (None, False), ([0, 3], True), ([2, 2, -2], False)):
        sim = SimDevice()
        got = ops.ReduceSum(axes=axes, keep_dims=keep).run(sim, [DeviceTensor.from_numpy(sim, x)])[0].numpy()
        want = OE.reduce_sum(x, axes, keep)
        assert got.shape == want.shape and np.array_equal(got, want)
        got = ops.ReduceMean(axes=axes, keep_dims=keep).run(sim, [DeviceTensor.from_numpy(sim, x)])[0].numpy()
        want = OE.reduce_mean(x, axes, keep)
        assert got.shape == want.shape and np.array_equal(got, want)
        np.testing.assert_allclose(got, x.astype(np.float64).mean(axis=tuple(sorted({a % 4 for a in axes})) if axes else None, keepdims=keep), rtol=1e-5, atol=1e-6)
    with pytest.raises(ops.OpError) as e:
        ops.ReduceSum(axes=[4]).run(SimDevice(), [DeviceTensor.from_numpy(SimDevice(), x)])
    assert e.value == ops.InvalidValue("Axis is invalid")
    # reduce.rs:1868-1872 (test_reduce_mean literals): mean over the last axis / all axes of [[1,2,3],[4,5,6]]... exact small integers
    lit = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    assert OE.reduce_mean(lit, [-1]).tolist() == [2.0, 5.0, 8.0] and OE.reduce_mean(lit, None).tolist() == 5.0


# ------------------------------------------------------------------------------------------------ GPU parity
class _GemvSpy:
    """Context proxy that notes whether an evaluation launched a GEMM with one row (reference gemv path: rtol 1e-5)."""

    def __init__(self, ctx):
        self._ctx, self.gemv = ctx, False

    def call(self, name, *a):
        if name == "rten_hip_gemm_f32" and a[0]._obj.m == 1:
            self.gemv = True
        return self._ctx.call(name, *a)

    def __getattr__(self, k):
        return getattr(self._ctx, k)


@pytest.mark.gpu
@pytest.mark.parametrize("eq,xs", ALL_VALUE_CASES, ids=[f"{i}:{c[0] or '<empty>'}" for i, c in enumerate(ALL_VALUE_CASES)])
def test_gpu_einsum_matches_oracle(ctx, eq, xs):
    spy = _GemvSpy(ctx)
    got = ops.Einsum(eq).run(spy, [DeviceTensor.from_numpy(ctx, x) for x in xs])[0].numpy()
    want = OE.einsum(eq, *xs)
    assert got.shape == want.shape
    if spy.gemv:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)  # M == 1: ISA-dependent order in the reference
    else:
        assert np.array_equal(got, want), (eq, np.abs(got - want).max())


@pytest.mark.gpu
def test_gpu_reduce_sum_strided_matches_oracle(ctx):
    """ReduceSum through strides (kept / reduced axes interleaved, diagonal, broadcast, slices of 1..3000 elements, empty)."""
    rng = np.random.default_rng(11)
    for shape, axes in (((4, 70, 3, 5), [1]), ((4, 70, 3, 5), [0, 3]), ((4, 70, 3, 5), [0, 1, 2, 3]), ((3000, 7), [0]), ((7, 3000), [1]),
                        ((129, 64), [1]), ((129, 65), [1]), ((5, 1, 9), [1]), ((6, 0, 4), [1]), ((2, 3, 4, 5, 6, 7), [1, 3, 5]),
                        # every branch of the 16-lane order on the four-rows-per-wave path (1..256 elements) and just above it
                        *[((37, n), [1]) for n in (1, 15, 16, 17, 63, 64, 65, 79, 80, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300)],
                        # contiguous slices of 64 / 128 / 256 with ragged row counts and kept prefix axes
                        ((1000, 256), [1]), ((3, 50, 64), [2]), ((4, 70, 128), [2]), ((2, 3, 5, 128), [3]), ((1, 128), [1]),
                        # column sums (coalesced kernel): whole and ragged groups of 16 columns, chunk / vector / tail boundaries,
                        # a kept prefix axis, two reduced axes of which the inner one is strided
                        *[((n, c), [0]) for n in (65, 128, 143, 144, 150, 1000) for c in (16, 17, 100)],
                        ((3, 200, 40), [1]), ((70, 3, 33), [0, 1]), ((9, 70, 5, 20), [1]), ((4096, 384), [0])):
        x = rng.standard_normal(shape).astype(np.float32)
        got = ops.ReduceSum(axes=axes, keep_dims=False).run(ctx, [DeviceTensor.from_numpy(ctx, x)])[0].numpy()
        want = OE.reduce_sum(x, axes, False)
        assert got.shape == want.shape and np.array_equal(got, want), (shape, axes)
        gm = ops.ReduceMean(axes=axes, keep_dims=True).run(ctx, [DeviceTensor.from_numpy(ctx, x)])[0].numpy()
        wm = OE.reduce_mean(x, axes, True)
        assert gm.shape == wm.shape and np.array_equal(gm, wm, equal_nan=True), (shape, axes)


@pytest.mark.gpu
def test_gpu_einsum_at_bert_attention_size(ctx):
    """BASELINE config-4 geometry (batch 32 x 12 heads x 128 tokens x 64) written as Einsum on the un-transposed [B,S,H,D]
    projections: one strided batched GEMM per product; checked against the oracle on a slice of heads (size-independent
    property: every (batch, head) block equals the oracle's evaluation of that block alone)."""
    rng = np.random.default_rng(3)
    q = rng.standard_normal((32, 128, 12, 64)).astype(np.float32)
    k = rng.standard_normal((32, 128, 12, 64)).astype(np.float32)
    s = ops.Einsum("bqhd,bkhd->bhqk").run(ctx, [DeviceTensor.from_numpy(ctx, q), DeviceTensor.from_numpy(ctx, k)])[0]
    assert s.shape == (32, 12, 128, 128)
    v = rng.standard_normal((32, 128, 12, 64)).astype(np.float32)
    o = ops.Einsum("bhqk,bkhd->bqhd").run(ctx, [s, DeviceTensor.from_numpy(ctx, v)])[0].numpy()
    sn = s.numpy()
    for b, h in ((0, 0), (7, 5), (31, 11)):
        want_s = OE.einsum("qd,kd->qk", q[b, :, h], k[b, :, h])
        assert np.array_equal(sn[b, h], want_s)
        assert np.array_equal(o[b, :, h], OE.einsum("qk,kd->qd", want_s, v[b, :, h]))


# ------------------------------------------------------------------------------------------------ seeded random equations
def _random_equation(rng):
    """2-3 operands over labels a..f with sizes 1..6: random ranks, repeated labels (diagonals), labels summed in one or several
    terms, random output subset in random order; `broadcast` variants shrink one occurrence of a label to size 1."""
    labels = "abcdef"
    size = {c: int(rng.integers(1, 7)) for c in labels}
    terms = []
    for _ in range(int(rng.integers(1, 4))):
        r = int(rng.integers(0, 5))
        t = "".join(rng.choice(list(labels), r, replace=True)) if rng.random() < 0.25 else "".join(rng.choice(list(labels), min(r, 6), replace=False))
        terms.append(t)
    used = sorted(set("".join(terms)))
    out = "".join(rng.permutation([c for c in used if rng.random() < 0.5])) if used else ""
    shapes = [[size[c] for c in t] for t in terms]
    broadcast = False
    if rng.random() < 0.2 and len(terms) > 1:
        for ti, t in enumerate(terms):  # a label that also occurs in another term, once in this one: make it 1 here
            cands = [i for i, c in enumerate(t) if t.count(c) == 1 and any(c in o for j, o in enumerate(terms) if j != ti)]
            if cands:
                shapes[ti][int(rng.choice(cands))] = 1
                broadcast = True
                break
    return ",".join(terms) + "->" + out, shapes, broadcast


RANDOM_EQUATIONS = [_random_equation(np.random.default_rng(1000 + i)) for i in range(160)]


@pytest.mark.parametrize("case", range(len(RANDOM_EQUATIONS)))
def test_random_equations_oracle_and_planner(case):
    """oracle vs numpy.einsum in f64 (the oracle's semantics beyond the reference's table) and planner-on-simulated-device vs
    oracle bit for bit."""
    eq, shapes, broadcast = RANDOM_EQUATIONS[case]
    xs = _random_operands(shapes, 5000 + case)
    want = OE.einsum(eq, *xs)
    if not broadcast:  # numpy's einsum has its own broadcasting rules for 1-sized axes: compare only label-consistent cases
        f = np.einsum(eq, *[x.astype(np.float64) for x in xs])
        assert want.shape == f.shape
        np.testing.assert_allclose(want, f, rtol=1e-4, atol=1e-4)
    sim = SimDevice()
    got = ops.Einsum(eq).run(sim, [DeviceTensor.from_numpy(sim, x) for x in xs])[0]
    assert tuple(got.shape) == want.shape and np.array_equal(got.numpy(), want), eq


@pytest.mark.gpu
def test_gpu_random_equations_match_oracle(ctx):
    bad = []
    for eq, shapes, _ in RANDOM_EQUATIONS:
        xs = _random_operands(shapes, 7000 + len(eq))
        spy = _GemvSpy(ctx)
        got = ops.Einsum(eq).run(spy, [DeviceTensor.from_numpy(ctx, x) for x in xs])[0].numpy()
        want = OE.einsum(eq, *xs)
        ok = got.shape == want.shape and (np.allclose(got, want, rtol=1e-5, atol=1e-6) if spy.gemv else np.array_equal(got, want))
        if not ok:
            bad.append(eq)
    assert not bad, bad
