"""ctypes front-end of the CPU oracle (oracle/rten_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``rten_amd``) never does.  Each function forwards to the C
restatement, which cites the reference file:line it follows.

Parity status: pinned against the reference's own golden vectors (tests/golden/) -- the Rust
reference cannot be built in this image (no cargo/rustc), so ``oracle/_ref`` does not exist.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librten_oracle.so")

LANES = 16  # SIMD lane count whose reduction order the oracle reproduces (AVX-512 host)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "rten_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.rto_exp_f32.restype = C.c_float
        _lib.rto_exp_f32.argtypes = [C.c_float]
        _lib.rto_erf_f32.restype = C.c_float
        _lib.rto_erf_f32.argtypes = [C.c_float]
        _lib.rto_gelu_f32.restype = C.c_float
        _lib.rto_gelu_f32.argtypes = [C.c_float]
        _lib.rto_simd_sum.restype = C.c_float
        _lib.rto_strerror.restype = C.c_char_p
        _lib.rto_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


i64 = C.c_int64


# ---------------------------------------------------------------- RNG (rng.rs, reduced_range_rng.rs)
class XorShiftRng:
    """Clone of rten_tensor::rng::XorShiftRng (rten-tensor/src/rng.rs:6-66)."""

    def __init__(self, seed: int):
        self.state = C.c_uint64(seed)

    def _fill(self, fn, n, dtype):
        out = np.empty(n, dtype=dtype)
        getattr(lib(), fn)(C.byref(self.state), i64(n), _p(out))
        return out

    def f32(self, n):
        return self._fill("rto_rng_f32", n, np.float32)

    def u8(self, n, reduced=False):
        return self._fill("rto_rng_u8_reduced" if reduced else "rto_rng_u8", n, np.uint8)

    def i8(self, n, reduced=False):
        return self._fill("rto_rng_i8_reduced" if reduced else "rto_rng_i8", n, np.int8)

    def i32(self, n):
        return self._fill("rto_rng_i32", n, np.int32)


# ---------------------------------------------------------------- shapes
class OpError(Exception):
    """Mirrors rten OpError::InvalidValue(msg) for the padding calculator."""


def calc_output_size_and_padding(in_size, kernel, strides, padding, dilations=(1, 1), ceil_mode=False):
    """src/ops/pooling.rs:139-159.  padding: "same" or [top, left, bottom, right]."""
    same = 1 if isinstance(padding, str) and padding.lower() == "same" else 0
    pads = (i64 * 4)(*([0, 0, 0, 0] if same else list(padding)))
    out = (i64 * 2)()
    opads = (i64 * 4)()
    rc = lib().rto_calc_output_size_and_padding(
        i64(in_size[0]), i64(in_size[1]), i64(kernel[0]), i64(kernel[1]), i64(strides[0]), i64(strides[1]),
        C.c_int(same), pads, i64(dilations[0]), i64(dilations[1]), C.c_int(1 if ceil_mode else 0), out, opads)
    if rc:
        raise OpError(lib().rto_strerror(rc).decode())
    return int(out[0]), int(out[1]), [int(v) for v in opads]


# ---------------------------------------------------------------- f32 GEMM / conv
BIAS_NONE, BIAS_PER_ROW, BIAS_PER_COL = 0, 1, 2


def gemm_f32(a, b, c=None, alpha=1.0, beta=0.0, bias=None, bias_kind=BIAS_NONE):
    """C = alpha*A@B + beta*C (+bias) in the reference's accumulation order.  `a`, `b` may be any
    2-D float32 numpy views (strides are honoured, nothing is copied)."""
    assert a.dtype == np.float32 and b.dtype == np.float32 and a.ndim == 2 and b.ndim == 2
    M, K = a.shape
    K2, N = b.shape
    assert K == K2
    if c is None:
        out = np.full((M, N), np.nan, dtype=np.float32) if beta == 0.0 else np.zeros((M, N), np.float32)
    else:
        out = np.ascontiguousarray(c, dtype=np.float32).copy()
    bias = None if bias is None else _f32(bias)
    es = 4
    lib().rto_gemm_f32(i64(M), i64(N), i64(K), _p(a), i64(a.strides[0] // es), i64(a.strides[1] // es),
                       _p(b), i64(b.strides[0] // es), i64(b.strides[1] // es), _p(out), i64(N),
                       C.c_float(alpha), C.c_float(beta), _p(bias), C.c_int(bias_kind if bias is not None else 0))
    return out


def set_gemv_threads(t: int):
    """Thread count the reference's gemv path is assumed to run with (column blocks of max(128, ceil(N / t)) columns); 0 = enough
    threads for 128-column blocks everywhere (the default)."""
    lib().rto_set_gemv_threads(i64(t))


def set_gemv_enabled(on: bool):
    """False: M == 1 products take the blocked GEMM order too (the reference with prepacked weights)."""
    lib().rto_set_gemv_enabled(C.c_int(1 if on else 0))


def matmul_f32(a, b, alpha=1.0, bias=None):
    """numpy.matmul-style batched product following src/ops/matmul.rs:208-385 (bias = per column)."""
    a = _f32(a)
    b = _f32(b)
    if a.ndim > 2 and b.ndim == 2:  # matmul.rs:266-297: flatten to one GEMM
        out = gemm_f32(a.reshape(-1, a.shape[-1]), b, alpha=alpha, bias=bias, bias_kind=BIAS_PER_COL)
        return out.reshape(*a.shape[:-1], b.shape[-1])
    if a.ndim == 2 and b.ndim == 2:
        return gemm_f32(a, b, alpha=alpha, bias=bias, bias_kind=BIAS_PER_COL)
    pre = np.broadcast_shapes(a.shape[:-2], b.shape[:-2])
    ab = np.broadcast_to(a, pre + a.shape[-2:]).reshape(-1, *a.shape[-2:])
    bb = np.broadcast_to(b, pre + b.shape[-2:]).reshape(-1, *b.shape[-2:])
    out = np.empty((ab.shape[0], a.shape[-2], b.shape[-1]), np.float32)
    for i in range(ab.shape[0]):
        out[i] = gemm_f32(ab[i], bb[i], alpha=alpha, bias=bias, bias_kind=BIAS_PER_COL)
    return out.reshape(*pre, a.shape[-2], b.shape[-1])


def conv2d_f32(x, w, bias=None, pads=(0, 0, 0, 0), strides=(1, 1), dilations=(1, 1), groups=1,
               residual=None, relu=False):
    """src/ops/conv.rs:124-365 (+ optional Add / Relu as the following graph ops)."""
    x = _f32(x)
    w = _f32(w)
    N, Cin, H, W = x.shape
    O, Cg, kh, kw = w.shape
    oh, ow, fp = calc_output_size_and_padding((H, W), (kh, kw), strides, pads, dilations)
    y = np.empty((N, O, oh, ow), np.float32)
    bias = None if bias is None else _f32(bias)
    residual = None if residual is None else _f32(residual)
    lib().rto_conv2d_f32(i64(N), i64(Cin), i64(H), i64(W), i64(O), i64(kh), i64(kw), (i64 * 4)(*fp),
                         (i64 * 2)(*strides), (i64 * 2)(*dilations), i64(groups), _p(x), _p(w), _p(bias),
                         _p(residual), C.c_int(1 if relu else 0), _p(y), i64(oh), i64(ow))
    return y


# ---------------------------------------------------------------- integer GEMM / conv
PAD_ZERO_POINT, PAD_RAW0_I8, PAD_RAW0_U8 = 0, 1, 2


def _is_signed(a):
    assert a.dtype in (np.int8, np.uint8), a.dtype
    return 1 if a.dtype == np.int8 else 0


def conv_transpose_output_size_and_padding(in_hw, k_hw, padding, strides, dilations=(1, 1), output_padding=(0, 0)):
    """src/ops/conv_transpose.rs:144-224.  padding: "same" or [top, left, bottom, right].  Returns (oh, ow, pads)."""
    if 0 in strides:
        raise ValueError("Strides must be > 0")
    if 0 in dilations:
        raise ValueError("Dilations must be > 0")
    if 0 in k_hw:
        raise ValueError("Kernel size must be > 0")
    if 0 in in_hw:
        raise ValueError("Input width and height must be > 0")
    ke = [(k - 1) * d + 1 for k, d in zip(k_hw, dilations)]
    full = [(i - 1) * s + k + op for i, s, k, op in zip(in_hw, strides, ke, output_padding)]
    if isinstance(padding, str):
        out = [i * s for i, s in zip(in_hw, strides)]
        pad = [f - o for f, o in zip(full, out)]
        if min(pad) < 0:
            raise ValueError("Input is too small")
        return out[0], out[1], [pad[0] // 2, pad[1] // 2, -(-pad[0] // 2), -(-pad[1] // 2)]
    if len(padding) != 4:
        raise ValueError("Wrong number of pad values")
    oh, ow = full[0] - padding[0] - padding[2], full[1] - padding[1] - padding[3]
    if oh < 0 or ow < 0:
        raise ValueError("Input is too small")
    return oh, ow, list(padding)


def matmul_nbits_f32(lhs, quant, scales):
    """MatMulNBits (src/ops/matmul/contrib.rs:21-106, rten-gemm/src/block_quant.rs).  lhs [..., rows, K] f32, quant [N, K/bs, bs/2] u8,
    scales [N, K/bs] f32.  rows == 1 follows the AVX-512 vector path, rows > 1 the f32 GEMM on the dequantised matrix."""
    lhs = _f32(lhs)
    quant = np.ascontiguousarray(quant, np.uint8)
    scales = _f32(scales).reshape(quant.shape[0], quant.shape[1])
    N, kb, half = quant.shape
    bs = half * 2
    rows, K = lhs.shape[-2], lhs.shape[-1]
    assert K == kb * bs
    batch = int(np.prod(lhs.shape[:-2], dtype=np.int64))
    y = np.empty(lhs.shape[:-1] + (N,), np.float32)
    rc = lib().rto_matmul_nbits_f32(i64(batch), i64(rows), i64(K), i64(N), i64(bs), _p(lhs), _p(quant), _p(scales), _p(y))
    assert rc == 0
    return y


def dequantize_4bit(quant, scales):
    """[K, N] f32 matrix a block-quantised RHS stands for (packing.rs:300-312)."""
    quant = np.ascontiguousarray(quant, np.uint8)
    N, kb, half = quant.shape
    scales = _f32(scales).reshape(N, kb)
    out = np.empty((kb * half * 2, N), np.float32)
    lib().rto_dequantize_4bit(i64(N), i64(kb * half * 2), i64(half * 2), _p(quant), _p(scales), _p(out))
    return out


def quantize_4bit_blocks(w, block_size):
    """Symmetric 4-bit block quantisation of a [K, N] f32 matrix into the MatMulNBits layout (zero point 8), the way ONNX Runtime's
    MatMul4BitsQuantizer does for is_symmetric=True: scale = -absmax_signed / 8 per block, q = clip(round(w / scale) + 8, 0, 15).
    Test / tooling helper -- not a reference restatement."""
    w = _f32(w)
    K, N = w.shape
    assert K % block_size == 0
    blocks = w.T.reshape(N, K // block_size, block_size)
    idx = np.abs(blocks).argmax(axis=2)
    peak = np.take_along_axis(blocks, idx[..., None], axis=2)[..., 0]
    scales = (peak / np.float32(-8.0)).astype(np.float32)
    inv = np.where(scales != 0, np.float32(1.0) / np.where(scales != 0, scales, 1), 0).astype(np.float32)
    q = np.clip(np.rint(blocks * inv[..., None]) + 8, 0, 15).astype(np.uint8)
    packed = (q[..., 0::2] | (q[..., 1::2] << 4)).astype(np.uint8)
    return packed, scales


def conv_transpose2d_f32(x, w, bias=None, padding=(0, 0, 0, 0), strides=(1, 1), dilations=(1, 1), groups=1, output_padding=(0, 0)):
    """src/ops/conv_transpose.rs:226-412.  x [N,C,H,W], w [C, O/g, kh, kw]."""
    x, w = _f32(x), _f32(w)
    N, Cc, H, W = x.shape
    _, Og, kh, kw = w.shape
    oh, ow, pads = conv_transpose_output_size_and_padding((H, W), (kh, kw), padding, strides, dilations, output_padding)
    y = np.empty((N, Og * groups, oh, ow), np.float32)
    bias = None if bias is None else _f32(bias)
    rc = lib().rto_conv_transpose2d_f32(i64(N), i64(Cc), i64(H), i64(W), i64(Og), i64(kh), i64(kw), (i64 * 4)(*pads), (i64 * 2)(*strides),
                                        (i64 * 2)(*dilations), i64(groups), _p(x), _p(w), _p(bias), _p(y), i64(oh), i64(ow))
    assert rc == 0
    return y


def gemm_int8(a, b, a_zp=None, b_zp=None, c=None):
    """sum_k (A - a_zp[m]) (B - b_zp[n]) -> i32 (rten-gemm/src/kernels/generic.rs:274-366)."""
    M, K = a.shape
    _, N = b.shape
    out = np.zeros((M, N), np.int32) if c is None else np.ascontiguousarray(c, np.int32).copy()
    azp = None if a_zp is None else np.ascontiguousarray(np.atleast_1d(a_zp), dtype=a.dtype)
    bzp = None if b_zp is None else np.ascontiguousarray(np.atleast_1d(b_zp), dtype=b.dtype)
    lib().rto_gemm_int8(i64(M), i64(N), i64(K), _p(a), C.c_int(_is_signed(a)), i64(a.strides[0]), i64(a.strides[1]),
                        _p(b), C.c_int(_is_signed(b)), i64(b.strides[0]), i64(b.strides[1]), _p(out), i64(N),
                        _p(azp), i64(0 if azp is None else azp.size), _p(bzp), i64(0 if bzp is None else bzp.size),
                        C.c_int(0 if c is None else 1))
    return out


def conv2d_int8(x, w, x_zp=0, w_zp=None, pads=(0, 0, 0, 0), strides=(1, 1), dilations=(1, 1), groups=1,
                pad_mode=PAD_RAW0_I8):
    """src/ops/conv.rs:421-476; x u8|i8 NCHW, w i8|u8 OIHW, scalar x_zp, scalar or [O] w_zp."""
    x = np.ascontiguousarray(x)
    w = np.ascontiguousarray(w)
    N, Cin, H, W = x.shape
    O, Cg, kh, kw = w.shape
    oh, ow, fp = calc_output_size_and_padding((H, W), (kh, kw), strides, pads, dilations)
    y = np.empty((N, O, oh, ow), np.int32)
    wzp = None if w_zp is None else np.ascontiguousarray(np.atleast_1d(w_zp), dtype=w.dtype)
    lib().rto_conv2d_int8(i64(N), i64(Cin), i64(H), i64(W), i64(O), i64(kh), i64(kw), (i64 * 4)(*fp),
                          (i64 * 2)(*strides), (i64 * 2)(*dilations), i64(groups), _p(x), C.c_int(_is_signed(x)),
                          _p(w), C.c_int(_is_signed(w)), C.c_int32(int(x_zp)), _p(wzp),
                          i64(0 if wzp is None else wzp.size), C.c_int(pad_mode), _p(y), i64(oh), i64(ow))
    return y


def cast_scale(x_i32, scale):
    x = np.ascontiguousarray(x_i32, np.int32)
    s = _f32(np.atleast_1d(scale))
    y = np.empty(x.shape, np.float32)
    lib().rto_cast_scale(i64(x.size), _p(x), _p(s), i64(s.size), _p(y))
    return y


def dynamic_quantize_linear(x):
    """src/ops/quantize.rs:352-436 -> (u8 tensor, scale f32, zero_point u8)."""
    x = _f32(x)
    y = np.empty(x.shape, np.uint8)
    scale = C.c_float()
    zp = C.c_uint8()
    lib().rto_dynamic_quantize_linear(i64(x.size), _p(x), _p(y), C.byref(scale), C.byref(zp))
    return y, np.float32(scale.value), np.uint8(zp.value)


# ---------------------------------------------------------------- elementwise
def _unary(fn, x):
    x = _f32(x)
    y = np.empty_like(x)
    getattr(lib(), fn)(i64(x.size), _p(x), _p(y))
    return y


def gelu(x):
    return _unary("rto_gelu", x)


def erf(x):
    return _unary("rto_erf", x)


def exp(x):
    return _unary("rto_exp", x)


def tanh(x):
    """rten-vecmath/src/tanh.rs: Tanh (the pooler activation of BERT)."""
    return _unary("rto_tanh", x)


def relu(x):
    return _unary("rto_relu", x)


def add(a, b):
    a = _f32(a)
    b = _f32(b)
    y = np.empty_like(a)
    lib().rto_add(i64(a.size), _p(a), _p(b), i64(b.size), _p(y))
    return y


# ---------------------------------------------------------------- row-wise
def softmax(x, addend=None, add_div=1, add_mod=None, flush_nan=False, lanes=LANES):
    """Softmax along the last axis (rten-vecmath/src/softmax.rs:60-100).  `addend` is added first
    (AddSoftmax, src/ops/attention.rs:30-68): row r uses addend row (r // add_div) % add_mod."""
    x = _f32(x)
    rows = x.size // x.shape[-1] if x.size else 0
    cols = x.shape[-1]
    y = np.empty_like(x)
    if addend is not None:
        addend = _f32(addend)
        if add_mod is None:
            add_mod = addend.size // cols
    lib().rto_softmax(i64(rows), i64(cols), _p(x), _p(addend), i64(add_div), i64(add_mod or 1), _p(y),
                      C.c_int(1 if flush_nan else 0), C.c_int(lanes))
    return y


def layer_norm(x, gamma=None, beta=None, gamma_scalar=1.0, beta_scalar=0.0, eps=1e-5, lanes=LANES):
    """LayerNormalization over the last axis (src/ops/norm.rs:456-529)."""
    x = _f32(x)
    cols = x.shape[-1]
    rows = x.size // cols
    y = np.empty_like(x)
    g = None if gamma is None else _f32(gamma)
    b = None if beta is None else _f32(beta)
    lib().rto_layer_norm(i64(rows), i64(cols), _p(x), _p(g), _p(b), C.c_float(gamma_scalar), C.c_float(beta_scalar),
                         C.c_float(eps), _p(y), C.c_int(lanes))
    return y


def batch_norm(x, scale, bias, mean, var, eps=1e-5):
    x = _f32(x)
    N, Cc = x.shape[:2]
    inner = x.size // (N * Cc)
    y = np.empty_like(x)
    lib().rto_batch_norm(i64(N), i64(Cc), i64(inner), _p(x), _p(_f32(scale)), _p(_f32(bias)), _p(_f32(mean)),
                         _p(_f32(var)), C.c_float(eps), _p(y))
    return y


# ---------------------------------------------------------------- pooling
def _pool(x, kernel, strides, pads, is_max, count_include_pad=False, ceil_mode=False):
    x = _f32(x)
    N, Cc, H, W = x.shape
    oh, ow, fp = calc_output_size_and_padding((H, W), kernel, strides, pads, (1, 1), ceil_mode)
    y = np.empty((N, Cc, oh, ow), np.float32)
    lib().rto_pool2d(i64(N), i64(Cc), i64(H), i64(W), i64(kernel[0]), i64(kernel[1]), i64(strides[0]), i64(strides[1]),
                     i64(fp[0]), i64(fp[1]), i64(oh), i64(ow), _p(x), _p(y), C.c_int(is_max),
                     C.c_int(1 if count_include_pad else 0))
    return y


def max_pool(x, kernel, strides, pads=(0, 0, 0, 0), ceil_mode=False):
    return _pool(x, kernel, strides, pads, 1, ceil_mode=ceil_mode)


def average_pool(x, kernel, strides, pads=(0, 0, 0, 0), count_include_pad=False, ceil_mode=False):
    return _pool(x, kernel, strides, pads, 0, count_include_pad, ceil_mode)


def global_average_pool(x, lanes=LANES):
    x = _f32(x)
    N, Cc = x.shape[:2]
    inner = x.size // (N * Cc)
    y = np.empty((N, Cc) + (1,) * (x.ndim - 2), np.float32)
    lib().rto_global_avg_pool(i64(N * Cc), i64(inner), _p(x), _p(y), C.c_int(lanes))
    return y


# ---------------------------------------------------------------- attention
def sdpa(q, k, v, mask=None, scale=None, lanes=LANES, flush_nan=True):
    """softmax(scale*Q K^T + mask) V per (batch, head) (src/ops/attention.rs:518-626).
    q:[B,H,S,D] k:[B,H,T,D] v:[B,H,T,Dv]; mask: None, [B,1,1,T] or [B,1,S,T] additive f32."""
    q = _f32(q)
    k = _f32(k)
    v = _f32(v)
    B, H, S, D = q.shape
    T = k.shape[2]
    Dv = v.shape[3]
    if scale is None:
        scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))  # f32 arithmetic, attention.rs:659-670
    out = np.empty((B, H, S, Dv), np.float32)
    mask_rs = 0
    if mask is not None:
        mask = _f32(mask)
        assert mask.shape[0] == B and mask.shape[1] == 1 and mask.shape[3] == T
        mask_rs = T if mask.shape[2] == S and S > 1 else 0
    lib().rto_sdpa(i64(B * H), i64(S), i64(T), i64(D), i64(Dv), _p(q), _p(k), _p(v), _p(mask), i64(H), i64(mask_rs),
                   C.c_float(scale), _p(out), C.c_int(lanes), C.c_int(1 if flush_nan else 0))
    return out


def num_threads() -> int:
    return int(lib().rto_num_threads())
