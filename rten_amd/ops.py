"""Host-side mirror of RTen's operator interface for the hot path (src/operator.rs:486-613, src/ops/*).

Each class mirrors the reference operator of the same name: same attribute names, same input order,
same validation and the same `OpError` messages (asserted verbatim by the reference's tests, e.g.
src/ops/conv.rs:1182-1268, src/ops/matmul.rs:1284-1333).  Validation runs on the host before any
launch; the arithmetic is done by librten_hip.so through the C ABI (include/rten_hip.h) on
device-resident tensors.  There is no CPU fallback: without the HIP library / an MI355X every `run`
raises.

The Rust reference cannot be compiled in this image (no cargo), so this mirror stands where the Rust
`Operator` impls of INTEGRATION.md would; the registry below mirrors `OpRegistry`
(src/op_registry.rs:25-72).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import lib as L
from .lib import Context
from .tensor import DeviceTensor


class OpError(Exception):
    """Mirrors rten::ops::OpError (src/operator.rs:116-144)."""

    def __init__(self, kind: str, msg: str = ""):
        super().__init__(f"{kind}({msg!r})" if msg else kind)
        self.kind = kind
        self.msg = msg

    def __eq__(self, other):
        return isinstance(other, OpError) and (self.kind, self.msg) == (other.kind, other.msg)

    __hash__ = Exception.__hash__


def InvalidValue(m):
    return OpError("InvalidValue", m)


def IncompatibleInputShapes(m):
    return OpError("IncompatibleInputShapes", m)


def UnsupportedValue(m):
    return OpError("UnsupportedValue", m)


UnsupportedType = OpError("UnsupportedType")
MissingInputs = OpError("MissingInputs")


def _vp(t):
    return None if t is None else C.c_void_p(t.ptr)


def _require(inputs, i):
    if i >= len(inputs) or inputs[i] is None:
        raise MissingInputs
    return inputs[i]


def _get(inputs, i):
    return inputs[i] if i < len(inputs) else None


def _want(t, dtype):
    if t.dtype != np.dtype(dtype):
        raise OpError("InputCastFailed", f"expected {np.dtype(dtype).name} tensor")
    return t


def calc_output_size_and_padding(ctx, in_size, kernel, strides, padding, dilations=(1, 1), ceil_mode=False):
    """src/ops/pooling.rs:139-159 via the C ABI.  padding: "same" or [top, left, bottom, right]."""
    same = isinstance(padding, str)
    if not same and len(padding) != 4:
        raise InvalidValue("Expected 4 padding values")
    pads = (C.c_int32 * 4)(*([0, 0, 0, 0] if same else [int(p) for p in padding]))
    out = (C.c_int32 * 2)()
    opads = (C.c_int32 * 4)()
    msg = C.c_char_p()
    rc = ctx.lib.rten_hip_calc_output_size_and_padding(
        in_size[0], in_size[1], kernel[0], kernel[1], strides[0], strides[1], 1 if same else 0, pads,
        dilations[0], dilations[1], 1 if ceil_mode else 0, out, opads, C.byref(msg))
    if rc:
        raise InvalidValue(msg.value.decode())
    return out[0], out[1], [opads[i] for i in range(4)]


class Operator:
    """Mirror of `trait Operator` (src/operator.rs:486-613): name(), run(ctx, inputs) -> [outputs]."""

    def name(self) -> str:
        return type(self).__name__

    def max_inputs(self):
        return None

    def run(self, ctx: Context, inputs):
        raise NotImplementedError

    # weight staging hook (prepack_inputs / prepack, operator.rs:586-601)
    def prepack_inputs(self):
        return []


# ------------------------------------------------------------------------------------------ Conv
class Conv(Operator):
    """src/ops/conv.rs:367-403.  inputs: X [N,C,H,W] (or [N,C,W]), W [O,C/g,kh,kw], bias [O]?
    Extra (backend fusion of the following graph ops): `fuse_relu`, and a 4th input =
    residual tensor added before the Relu."""

    def __init__(self, groups=1, dilations=(1, 1), padding=(0, 0, 0, 0), strides=(1, 1), fuse_relu=False):
        self.groups = groups
        self.dilations = list(dilations)
        self.padding = padding
        self.strides = list(strides)
        self.fuse_relu = fuse_relu
        self._packed = {}

    def max_inputs(self):
        return 4

    def prepack_inputs(self):
        return [1]

    def _geometry(self, ctx, xs, ws):
        if len(xs) != 4:
            raise InvalidValue("input must have 4 dims (NCHW)")
        if len(ws) != 4:
            raise InvalidValue("kernel must have 4 dims (OCHW)")
        if len(self.strides) != 2:
            raise InvalidValue("expected 2 stride values")
        if len(self.dilations) != 2:
            raise InvalidValue("expected 2 dilation values")
        n, c, h, w = xs
        o, kc, kh, kw = ws
        oh, ow, pads = calc_output_size_and_padding(ctx, (h, w), (kh, kw), self.strides, self.padding, self.dilations)
        if self.groups == 0:
            raise InvalidValue("Group count must be > 0")
        if c % self.groups != 0:
            raise InvalidValue("Input channel count not divisible by groups")
        if c // self.groups != kc:
            raise IncompatibleInputShapes("Input channels (per group) does not match kernel input channels")
        if o % self.groups != 0:
            raise InvalidValue("Output channel count not divisible by groups")
        d = L.Conv2dDesc(n, c, h, w, o, kh, kw, (C.c_int32 * 4)(*pads), self.strides[0], self.strides[1],
                         self.dilations[0], self.dilations[1], self.groups, oh, ow)
        return d

    def prepack(self, ctx, weight: DeviceTensor, desc=None):
        """Stage weights once (GPU analogue of PrepackedInput, operator.rs:25-66)."""
        if desc is None:
            o, kc, kh, kw = weight.shape
            desc = L.Conv2dDesc(1, kc * self.groups, max(kh, 1), max(kw, 1), o, kh, kw, (C.c_int32 * 4)(0, 0, 0, 0),
                                1, 1, 1, 1, self.groups, 1, 1)
        nbytes = ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(desc))
        packed = DeviceTensor(ctx, (nbytes // 4,), np.float32)
        ctx.call("rten_hip_conv2d_f32_prepack", C.byref(desc), weight.vp, packed.vp)
        return packed

    def _as_2d(self, x, w):
        """1-D convolution: expand to 2-D and remove the extra axis from the result (conv.rs:142-182; views, no copies)."""
        if len(w.shape) != 3:
            raise InvalidValue("kernel must have 3 dims (OCW)")
        if not isinstance(self.padding, str):
            if len(self.padding) != 2:
                raise InvalidValue("expected 2 pad values")
            pad2 = [0, self.padding[0], 0, self.padding[1]]
        else:
            pad2 = "same"
        if len(self.strides) != 1:
            raise InvalidValue("expected 1 stride value")
        if len(self.dilations) != 1:
            raise InvalidValue("expected 1 dilation value")
        op = Conv(self.groups, [1, self.dilations[0]], pad2, [1, self.strides[0]], self.fuse_relu)
        n, c, wd = x.shape
        return op, x.view((n, c, 1, wd)), w.view((w.shape[0], w.shape[1], 1, w.shape[2]))

    def run(self, ctx, inputs, packed_weight: DeviceTensor | None = None, out: DeviceTensor | None = None):
        x = _want(_require(inputs, 0), np.float32)
        w = _want(_require(inputs, 1), np.float32)
        bias = _get(inputs, 2)
        residual = _get(inputs, 3)
        if len(x.shape) == 3:
            op, x2, w2 = self._as_2d(x, w)
            res2 = residual.view((residual.shape[0], residual.shape[1], 1, residual.shape[2])) if residual is not None else None
            y = op.run(ctx, [x2, w2, bias, res2], packed_weight=packed_weight)[0]
            return [y.view((y.shape[0], y.shape[1], y.shape[3]))]
        d = self._geometry(ctx, x.shape, w.shape)
        if bias is not None and bias.shape[0] != d.o:
            raise IncompatibleInputShapes("bias.size(0) != out_channels")
        y = out if out is not None else DeviceTensor(ctx, (d.n, d.o, d.out_h, d.out_w), np.float32)
        flags = (L.CONV_RELU if self.fuse_relu else 0) | (L.CONV_RESIDUAL if residual is not None else 0)
        wt = packed_weight if packed_weight is not None else w
        ctx.call("rten_hip_conv2d_f32", C.byref(d), x.vp, wt.vp, 1 if packed_weight is not None else 0, _vp(bias),
                 _vp(residual), flags, y.vp)
        return [y]


class ConvTranspose(Operator):
    """src/ops/conv_transpose.rs:414-458 (op), :226-412 (conv_transpose).  inputs: X [N,C,H,W] (or [N,C,W]), W [C, O/g, kh, kw], bias [O]?"""

    def __init__(self, padding=(0, 0, 0, 0), groups=1, strides=(1, 1), dilations=(1, 1), output_padding=None):
        self.padding = padding
        self.groups = groups
        self.strides = list(strides)
        self.dilations = list(dilations)
        self.output_padding = None if output_padding is None else list(output_padding)

    def max_inputs(self):
        return 3

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        w = _want(_require(inputs, 1), np.float32)
        bias = _get(inputs, 2)
        if len(x.shape) == 3:  # 1-D: expand to 2-D, remove the extra axis from the result (conv_transpose.rs:237-291)
            if len(w.shape) != 3:
                raise InvalidValue("kernel must have 3 dims (OCW)")
            if not isinstance(self.padding, str):
                if len(self.padding) != 2:
                    raise InvalidValue("expected 2 pad values")
                pad2 = [0, self.padding[0], 0, self.padding[1]]
            else:
                pad2 = self.padding
            if len(self.strides) != 1:
                raise InvalidValue("expected 1 stride value")
            if len(self.dilations) != 1:
                raise InvalidValue("expected 1 dilation value")
            if self.output_padding is not None and len(self.output_padding) != 1:
                raise InvalidValue("expected 1 output_padding value")
            op = ConvTranspose(pad2, self.groups, [1, self.strides[0]], [1, self.dilations[0]], [0, self.output_padding[0]] if self.output_padding else None)
            y = op.run(ctx, [x.view((x.shape[0], x.shape[1], 1, x.shape[2])), w.view((w.shape[0], w.shape[1], 1, w.shape[2])), bias])[0]
            return [y.view((y.shape[0], y.shape[1], y.shape[3]))]
        if self.groups == 0:
            raise InvalidValue("Group count must be > 0")
        if len(x.shape) != 4:
            raise InvalidValue("input must have 4 dims (NCHW)")
        if len(w.shape) != 4:
            raise InvalidValue("kernel must have 4 dims (COHW)")
        n, c, h, wd = x.shape
        kc, og, kh, kw = w.shape
        o = og * self.groups
        if bias is not None and bias.shape[0] != o:
            raise IncompatibleInputShapes("bias.size(0) != out_channels")
        if c != kc:
            raise IncompatibleInputShapes("Input channels does not match kernel input channels")
        if kc % self.groups != 0:
            raise InvalidValue("Input channel count not divisible by groups")
        if len(self.strides) != 2:
            raise InvalidValue("expected 2 stride values")
        if len(self.dilations) != 2:
            raise InvalidValue("expected 2 dilation values")
        if self.output_padding is not None and len(self.output_padding) != 2:
            raise InvalidValue("expected 2 output_padding values")
        op_h, op_w = self.output_padding or (0, 0)
        same = isinstance(self.padding, str)
        if not same and len(self.padding) != 4:
            raise InvalidValue("Wrong number of pad values")
        pads = (C.c_int32 * 4)(*([0, 0, 0, 0] if same else [int(p) for p in self.padding]))
        out_hw, out_pads, msg = (C.c_int32 * 2)(), (C.c_int32 * 4)(), C.c_char_p()
        rc = ctx.lib.rten_hip_conv_transpose_output_size(h, wd, kh, kw, self.strides[0], self.strides[1], 1 if same else 0, pads, self.dilations[0], self.dilations[1],
                                                         op_h, op_w, out_hw, out_pads, C.byref(msg))
        if rc:
            raise InvalidValue(msg.value.decode() if msg.value else "")
        d = L.Conv2dDesc(n, c, h, wd, o, kh, kw, out_pads, self.strides[0], self.strides[1], self.dilations[0], self.dilations[1], self.groups, out_hw[0], out_hw[1])
        y = DeviceTensor(ctx, (n, o, out_hw[0], out_hw[1]), np.float32)
        ctx.call("rten_hip_conv_transpose2d_f32", C.byref(d), x.vp, w.vp, _vp(bias), y.vp)
        return [y]


class ConvInteger(Operator):
    """src/ops/conv.rs:478-526.  inputs: X u8|i8, W i8|u8, x_zero_point (scalar)?, w_zero_point (scalar|[O])?"""

    def __init__(self, groups=1, dilations=(1, 1), padding=(0, 0, 0, 0), strides=(1, 1), pad_mode=L.PAD_RAW0_I8):
        self._conv = Conv(groups, dilations, padding, strides)
        self.pad_mode = pad_mode

    def max_inputs(self):
        return 4

    def _desc(self, ctx, x, w, x_zp, w_zp):
        if x.dtype not in (np.uint8, np.int8) or w.dtype not in (np.uint8, np.int8):
            raise UnsupportedType
        o = w.shape[0] if len(w.shape) >= 1 else 0
        if x_zp is not None and x_zp.size != 1:
            raise InvalidValue("input zero point must be a scalar")
        w_zp_len = _zp_len(w_zp, o, "w")  # zero_point_to_vec, matmul.rs:513-531
        d = self._conv._geometry(ctx, x.shape, w.shape)
        return L.Conv2dInt8Desc(d, 1 if x.dtype == np.int8 else 0, 1 if w.dtype == np.int8 else 0, w_zp_len, self.pad_mode)

    def run(self, ctx, inputs, scale=None, bias=None, residual=None, relu=False, out=None):
        x = _require(inputs, 0)
        w = _require(inputs, 1)
        x_zp, w_zp = _get(inputs, 2), _get(inputs, 3)
        di = self._desc(ctx, x, w, x_zp, w_zp)
        d = di.conv
        y = out if out is not None else DeviceTensor(ctx, (d.n, d.o, d.out_h, d.out_w), np.float32 if scale is not None else np.int32)
        flags = (L.CONV_RELU if relu else 0) | (L.CONV_RESIDUAL if residual is not None else 0)
        ctx.call("rten_hip_conv2d_int8", C.byref(di), x.vp, w.vp, _vp(x_zp), _vp(w_zp), _vp(scale), _vp(bias),
                 _vp(residual), flags, y.vp)
        return [y]


class ConvIntegerToFloat(Operator):
    """src/ops/conv.rs:552-587.  inputs: X, W, x_zp, w_zp, scale (scalar).
    Backend fusion extras: bias (the following Add of a [1,O,1,1] constant), residual, relu."""

    def __init__(self, conv: ConvInteger, fuse_relu=False):
        self.conv = conv
        self.fuse_relu = fuse_relu

    def max_inputs(self):
        return 7

    def run(self, ctx, inputs, out=None):
        scale = _want(_require(inputs, 4), np.float32)
        if scale.size != 1:
            raise InvalidValue("scale should be a scalar")
        return self.conv.run(ctx, inputs[:4], scale=scale, bias=_get(inputs, 5), residual=_get(inputs, 6),
                             relu=self.fuse_relu, out=out)


# ------------------------------------------------------------------------------------------ MatMul family
def _broadcast_shapes(a, b):
    try:
        return tuple(np.broadcast_shapes(tuple(a), tuple(b)))
    except ValueError:
        return None


def _matmul(ctx, a: DeviceTensor, b: DeviceTensor, bias=None, alpha=None, act=L.ACT_NONE, b_transposed=False):
    """numpy.matmul rules, src/ops/matmul.rs:208-385.  Contiguous inputs; `b_transposed` reads B as
    [..., N, K] (TransformInputs / transB folded into strides, matmul.rs:47-48, fusions.rs:1066)."""
    ash, bsh = list(a.shape), list(b.shape)
    if len(ash) < 1 or len(bsh) < 1:
        raise InvalidValue("Inputs must have >= 1 dimensions")
    a_vec, b_vec = len(ash) == 1, len(bsh) == 1
    if a_vec:
        ash = [1] + ash
    if b_vec:
        bsh = bsh + [1] if not b_transposed else [1] + bsh
    if b_transposed:
        bsh = bsh[:-2] + [bsh[-1], bsh[-2]]
    m, k = ash[-2], ash[-1]
    kb, n = bsh[-2], bsh[-1]
    if k != kb:
        raise IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix")
    pre = _broadcast_shapes(ash[:-2], bsh[:-2])
    if pre is None:
        raise IncompatibleInputShapes("Cannot broadcast shapes")
    out_shape = list(pre) + [m, n]
    na = int(np.prod(ash[:-2], dtype=np.int64))
    nb = int(np.prod(bsh[:-2], dtype=np.int64))
    b_rs, b_cs = (1, k) if b_transposed else (n, 1)
    y = DeviceTensor(ctx, out_shape, np.float32)
    if int(np.prod(out_shape, dtype=np.int64)) > 0:
        if na > 1 and nb == 1:  # matmul.rs:266-297: one [A*M, K] x [K, N] GEMM
            d = L.gemm_desc(na * m, n, k, k, 1, b_rs, b_cs, n, alpha=alpha if alpha is not None else 1.0,
                            bias_kind=L.BIAS_PER_COL if bias is not None else L.BIAS_NONE, act=act)
        else:
            batch = int(np.prod(pre, dtype=np.int64)) if len(pre) else 1
            if na not in (1, batch) or nb not in (1, batch):
                raise UnsupportedValue("partial batch broadcasting is not supported by the device path")
            d = L.gemm_desc(m, n, k, k, 1, b_rs, b_cs, n, batch, m * k if na > 1 else 0, k * n if nb > 1 else 0, m * n,
                            alpha=alpha if alpha is not None else 1.0,
                            bias_kind=L.BIAS_PER_COL if bias is not None else L.BIAS_NONE, act=act)
        ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, b.vp, _vp(bias), y.vp)
    if a_vec:
        out_shape.pop(-2)
    if b_vec:
        out_shape.pop(-1)
    return y.reshape(out_shape) if (a_vec or b_vec) else y


class MatMul(Operator):
    """src/ops/matmul.rs:387-428"""

    def max_inputs(self):
        return 2

    def run(self, ctx, inputs):
        a = _want(_require(inputs, 0), np.float32)
        b = _want(_require(inputs, 1), np.float32)
        return [_matmul(ctx, a, b)]


class FusedMatMul(Operator):
    """src/ops/matmul.rs:455-510: MatMul + bias (per column) + alpha.  `act` is a backend fusion extra
    (GELU epilogue for the BERT FFN)."""

    def __init__(self, alpha=None, act=L.ACT_NONE, transpose_b=False):
        self.alpha = alpha
        self.act = act
        self.transpose_b = transpose_b

    def max_inputs(self):
        return 3

    def run(self, ctx, inputs):
        a = _want(_require(inputs, 0), np.float32)
        b = _want(_require(inputs, 1), np.float32)
        bias = _get(inputs, 2)
        if bias is not None and len(bias.shape) != 1:
            raise OpError("InputCastFailed", "expected tensor with 1 dims")
        return [_matmul(ctx, a, b, bias=bias, alpha=self.alpha, act=self.act, b_transposed=self.transpose_b)]


class Gemm(Operator):
    """src/ops/matmul.rs:106-156: c = alpha * (a b) + beta * c with transA / transB."""

    def __init__(self, alpha=1.0, beta=1.0, transpose_a=False, transpose_b=False):
        self.alpha, self.beta, self.transpose_a, self.transpose_b = alpha, beta, transpose_a, transpose_b

    def max_inputs(self):
        return 3

    def run(self, ctx, inputs):
        a = _want(_require(inputs, 0), np.float32)
        b = _want(_require(inputs, 1), np.float32)
        c = _get(inputs, 2)
        if len(a.shape) != 2:
            raise InvalidValue("a must have 2 dims")
        if len(b.shape) != 2:
            raise InvalidValue("b must have 2 dims")
        m, k = (a.shape[1], a.shape[0]) if self.transpose_a else a.shape
        kb, n = (b.shape[1], b.shape[0]) if self.transpose_b else b.shape
        if k != kb:
            raise IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix")
        a_rs, a_cs = (1, a.shape[1]) if self.transpose_a else (a.shape[1], 1)
        b_rs, b_cs = (1, b.shape[1]) if self.transpose_b else (b.shape[1], 1)
        beta = self.beta if c is not None else 0.0
        if c is not None and beta != 0.0 and _broadcast_shapes(tuple(c.shape), (m, n)) != (m, n):
            raise IncompatibleInputShapes("Cannot broadcast c to output shape")
        y = DeviceTensor(ctx, (m, n), np.float32)
        if c is not None and beta != 0.0:
            # expand_to(c, out_shape) then gemm with beta (matmul.rs:63-82)
            cs = tuple(c.shape)
            if cs == (m, n):
                ctx.call("rten_hip_memcpy_d2d", y.vp, c.vp, C.c_size_t(y.nbytes))
            else:  # row / column / scalar broadcast: materialise via add to zeros
                ctx.call("rten_hip_memset", y.vp, 0, C.c_size_t(y.nbytes))
                if cs in ((n,), (1, n)) or c.size == 1:
                    ctx.call("rten_hip_add_f32", m * n, y.vp, c.vp, c.size, y.vp)
                elif cs == (m, 1):
                    ctx.call("rten_hip_add_channel_bias_f32", 1, m, n, y.vp, c.vp, y.vp)
                else:
                    raise IncompatibleInputShapes("Cannot broadcast c to output shape")
        if m * n > 0:
            d = L.gemm_desc(m, n, k, a_rs, a_cs, b_rs, b_cs, n, alpha=self.alpha, beta=beta)
            ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, b.vp, None, y.vp)
        return [y]


def _zp_len(zp, expected, what):
    if zp is None:
        return 0
    if len(zp.shape) == 0:
        return 1
    if len(zp.shape) == 1:
        if zp.shape[0] != expected:
            raise InvalidValue("Zero point has incorrect size")
        return expected if expected != 1 else 1
    raise UnsupportedValue("Only scalar or vector zero points are supported")


def _broadcast_prefix(a_prefix, b_prefix):
    """broadcast_shapes (rten-tensor) of two batch prefixes; None when they do not broadcast."""
    n = max(len(a_prefix), len(b_prefix))
    ap = [1] * (n - len(a_prefix)) + list(a_prefix)
    bp = [1] * (n - len(b_prefix)) + list(b_prefix)
    out = []
    for x, y in zip(ap, bp):
        if x != y and x != 1 and y != 1:
            return None
        out.append(max(x, y) if min(x, y) != 0 else 0)
    return out, ap, bp


class MatMulInteger(Operator):
    """src/ops/matmul.rs:582-700 (matmul_integer -> matmul_impl :208-385).  inputs: A u8|i8 [..., M, K], B i8|u8 [..., K, N],
    a_zero_point? (scalar or [M]), b_zero_point? (scalar or [N]).  All of matmul_impl's forms: vector operands, the
    `[A.., M, K] x [K, N]` collapse with the row zero points cycled (:266-280), and batched / broadcast prefixes
    (batched_gemm_uninit, :302-372).  `packed_b`: the RHS staged at load by `prepack` (Operator::prepack, :696-705)."""

    def max_inputs(self):
        return 4

    def prepack_inputs(self):
        return [1]

    def prepack(self, ctx, b):
        """Stage a constant 2-D RHS once (PackedBMatrix).  Returns None when the staged kernel does not cover the shape."""
        if b.dtype not in (np.uint8, np.int8) or len(b.shape) != 2:
            return None
        k, n = b.shape
        nbytes = ctx.lib.rten_hip_gemm_int8_packed_bytes(k, n)
        if not nbytes:
            return None
        packed = DeviceTensor(ctx, [nbytes], np.uint8)
        ctx.call("rten_hip_gemm_int8_prepack", k, n, b.vp, n, 1, 1 if b.dtype == np.int8 else 0, packed.vp)
        packed.packed_for = (k, n, b.dtype)
        return packed

    def run(self, ctx, inputs, scale=None, packed_b=None):
        a = _require(inputs, 0)
        b = _require(inputs, 1)
        if a.dtype not in (np.uint8, np.int8) or b.dtype not in (np.uint8, np.int8):
            raise UnsupportedType
        a_zp, b_zp = _get(inputs, 2), _get(inputs, 3)
        ash, bsh = list(a.shape), list(b.shape)
        a_rows = ash[-2] if len(ash) > 1 else 1  # matmul_integer, matmul.rs:598-607
        b_cols = bsh[-1] if len(bsh) > 1 else 1
        azl = _zp_len(a_zp, a_rows, "a")
        bzl = _zp_len(b_zp, b_cols, "b")
        if len(ash) < 1 or len(bsh) < 1:
            raise InvalidValue("Inputs must have >= 1 dimensions")
        a_is_vec, b_is_vec = len(ash) == 1, len(bsh) == 1  # numpy.matmul rules (matmul.rs:232-240)
        if a_is_vec:
            ash = [1] + ash
        if b_is_vec:
            bsh = bsh + [1]
        m, k = ash[-2], ash[-1]
        kb, n = bsh[-2], bsh[-1]
        if k != kb:
            raise IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix")
        bc = _broadcast_prefix(ash[:-2], bsh[:-2])
        if bc is None:
            raise IncompatibleInputShapes("Cannot broadcast shapes")
        out_prefix, ap, bp = bc
        out_shape = out_prefix + [m, n]
        sl = 0
        if scale is not None:
            if len(scale.shape) > 1:
                raise InvalidValue("scale should have rank 0 or 1")
            sl = 1 if scale.size == 1 else scale.size
            if sl > 1 and sl != n:
                raise IncompatibleInputShapes("Scale length does not match tensor columns")
        final_shape = list(out_shape)
        if a_is_vec:
            del final_shape[-2]
        if b_is_vec:
            del final_shape[-1]
        y = DeviceTensor(ctx, final_shape, np.float32 if scale is not None else np.int32)
        if y.size == 0:
            return [y]
        num_a = int(np.prod(ash[:-2], dtype=np.int64))
        num_b = int(np.prod(bsh[:-2], dtype=np.int64))
        a_s, b_s = 1 if a.dtype == np.int8 else 0, 1 if b.dtype == np.int8 else 0
        use_packed = packed_b is not None and num_b == 1 and getattr(packed_b, "packed_for", None) == (k, n, b.dtype)
        bptr = packed_b.vp if use_packed else b.vp

        def call(mm, a_zp_len, batch, a_bs, b_bs, c_bs, a_off=0, b_off=0, c_off=0):
            d = L.GemmInt8Desc(mm, n, k, k, 1, n, 1, n, a_s, b_s, a_zp_len, bzl, sl, batch, a_bs, b_bs, c_bs, 1 if use_packed else 0)
            itm = y.dtype.itemsize
            ctx.call("rten_hip_gemm_int8", C.byref(d), C.c_void_p(a.ptr + a_off), bptr if use_packed else C.c_void_p(b.ptr + b_off),
                     _vp(a_zp), _vp(b_zp), _vp(scale), C.c_void_p(y.ptr + c_off * itm))

        if num_b == 1:
            # [A.., M, K] x [K, N] as one [A*M, K] x [K, N] product; row r uses a_zp[r % M] (matmul.rs:259-296)
            call(num_a * m, azl, 1, 0, 0, 0)
            return [y]
        # batched products over the broadcast prefix (matmul.rs:302-372)
        nb = int(np.prod(out_prefix, dtype=np.int64))
        if num_a == 1 or ap == out_prefix:
            a_mat = 0 if num_a == 1 else m * k
            if num_b == 1 or bp == out_prefix:
                call(m, azl, nb, a_mat, 0 if num_b == 1 else k * n, m * n)
                return [y]
        # general broadcast: one product per output matrix
        a_str = np.array([0 if d == 1 else int(np.prod(ap[i + 1:], dtype=np.int64)) for i, d in enumerate(ap)], np.int64)
        b_str = np.array([0 if d == 1 else int(np.prod(bp[i + 1:], dtype=np.int64)) for i, d in enumerate(bp)], np.int64)
        for z, idx in enumerate(np.ndindex(*out_prefix)):
            ia, ib = int(np.dot(idx, a_str)), int(np.dot(idx, b_str))
            call(m, azl, 1, 0, 0, 0, a_off=ia * m * k, b_off=ib * k * n, c_off=z * m * n)
        return [y]


class MatMulIntegerToFloat(Operator):
    """src/ops/matmul.rs:775-810.  inputs: A, B, a_zp, b_zp, scale (scalar or [N])."""

    def __init__(self):
        self.matmul = MatMulInteger()

    def max_inputs(self):
        return 5

    def prepack_inputs(self):
        return [1]

    def prepack(self, ctx, b):
        return self.matmul.prepack(ctx, b)

    def run(self, ctx, inputs, packed_b=None):
        scale = _want(_require(inputs, 4), np.float32)
        return self.matmul.run(ctx, inputs[:4], scale=scale, packed_b=packed_b)


class MatMulNBits(Operator):
    """com.microsoft MatMulNBits, src/ops/matmul/contrib.rs:119-196 (op), :21-106 (matmul_nbits).
    inputs: A f32 [..., M, K]; B u8 [N, K/bs, bs/2] (4-bit, even element in the low nibble, zero point 8); scales f32 [N, K/bs] or 1-D.
    accuracy_level 4 (AccuracyLevel::Int8) is an opt-in approximation the reference may decline (contrib.rs:102-108): always Float here."""

    def __init__(self, bits=4, block_size=32, accuracy_level=0):
        self.bits = bits
        self.block_size = block_size
        self.accuracy_level = accuracy_level

    def max_inputs(self):
        return 3

    def run(self, ctx, inputs):
        lhs = _want(_require(inputs, 0), np.float32)
        rhs = _want(_require(inputs, 1), np.uint8)
        if len(rhs.shape) != 3:
            raise OpError("InputCastFailed", "expected tensor with 3 dims")
        scales = _want(_require(inputs, 2), np.float32)
        n = rhs.shape[0]
        if len(scales.shape) == 1:  # earlier versions of the spec used 1-D scales (contrib.rs:152-164)
            k = lhs.shape[-1] if len(lhs.shape) >= 1 else 1
            k_blocks = k // self.block_size if self.block_size else 0
            if scales.size != n * k_blocks:
                raise InvalidValue("Expected 1D `scales` size to match columns * block_size")
        elif len(scales.shape) != 2:
            raise InvalidValue("Expected `scales` to have one or two dims")
        if len(inputs) > 3:
            raise UnsupportedValue("zero_points, g_idx and bias inputs are unsupported")
        if len(lhs.shape) < 2:
            raise InvalidValue("A input must have at least 2 dims")
        if self.bits not in (4, 8):  # BlockQuantizedMatrix::new (block_quant.rs:690-708)
            raise UnsupportedValue("Unsupported bits-per-element")
        bs = rhs.shape[2] * (8 // self.bits)
        if bs < 16 or bs & (bs - 1):
            raise UnsupportedValue("Unsupported K block size")
        rows, k = lhs.shape[-2], lhs.shape[-1]
        if k != rhs.shape[1] * bs:
            raise IncompatibleInputShapes("Columns of first matrix does not match rows of second matrix")
        if self.bits != 4:
            raise UnsupportedValue("MatMulNBits: only 4-bit elements (GemmError::QuantBitsNotSupported, block_quant.rs:77-79)")
        if len(scales.shape) == 2 and tuple(scales.shape) != (n, rhs.shape[1]):
            raise IncompatibleInputShapes("scales shape does not match the quantised matrix")
        batch = int(np.prod(lhs.shape[:-2], dtype=np.int64))
        y = DeviceTensor(ctx, list(lhs.shape[:-1]) + [n], np.float32)
        ctx.call("rten_hip_matmul_nbits_f32", batch, rows, k, n, bs, lhs.vp, rhs.vp, scales.vp, y.vp)
        return [y]


# ------------------------------------------------------------------------------------------ norm / softmax
def _resolve_axis(ndim, axis):
    if axis < -ndim or axis >= ndim:
        raise InvalidValue("Axis is invalid")
    return axis + ndim if axis < 0 else axis


class Softmax(Operator):
    """src/ops/norm.rs:842-900; any axis (a non-last axis is moved last and back, like normalize_lanes)."""

    def __init__(self, axis=-1, flush_nans_to_zero=False):
        self.axis = axis
        self.flush_nans_to_zero = flush_nans_to_zero

    def max_inputs(self):
        return 1

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        nd = len(x.shape)
        ax = _resolve_axis(nd, self.axis)
        flush = 1 if self.flush_nans_to_zero else 0
        if ax != nd - 1:
            # normalize_lanes (src/ops/norm.rs:705-754): move the axis last, make contiguous, apply, move it back
            if nd > 6:
                raise UnsupportedValue("Softmax over a non-last axis of more than 6 dims is not supported by the device path")
            fwd = [i for i in range(nd) if i != ax] + [ax]
            back = [fwd.index(i) for i in range(nd)]
            tshape = [x.shape[i] for i in fwd]
            t, u, y = DeviceTensor(ctx, tshape, np.float32), DeviceTensor(ctx, tshape, np.float32), DeviceTensor(ctx, x.shape, np.float32)
            if x.size:
                cols = x.shape[ax]
                i64 = lambda v: (C.c_int64 * len(v))(*v)
                i32 = lambda v: (C.c_int32 * len(v))(*v)
                ctx.call("rten_hip_transpose_b32", nd, i64(list(x.shape)), i32(fwd), x.vp, t.vp)
                ctx.call("rten_hip_softmax_f32", x.size // cols, cols, t.vp, None, 1, 1, flush, u.vp)
                ctx.call("rten_hip_transpose_b32", nd, i64(tshape), i32(back), u.vp, y.vp)
            return [y]
        y = DeviceTensor(ctx, x.shape, np.float32)
        cols = x.shape[-1]
        ctx.call("rten_hip_softmax_f32", x.size // max(cols, 1), cols, x.vp, None, 1, 1, flush, y.vp)
        return [y]


class AddSoftmax(Operator):
    """src/ops/attention.rs:70-156: Softmax(Add(qk, m), axis=-1); `m` broadcast to qk."""

    def __init__(self, flush_nans_to_zero=False):
        self.flush_nans_to_zero = flush_nans_to_zero

    def max_inputs(self):
        return 2

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        y_in = _want(_require(inputs, 1), np.float32)
        qk, m = (x, y_in) if x.size > y_in.size else (y_in, x)
        if _broadcast_shapes(qk.shape, m.shape) != tuple(qk.shape):
            if _broadcast_shapes(qk.shape, m.shape) is None:
                raise IncompatibleInputShapes("Cannot broadcast inputs")
            raise UnsupportedValue("device AddSoftmax needs one input to have the output shape")
        cols = qk.shape[-1]
        rows = qk.size // max(cols, 1)
        msh = (1,) * (len(qk.shape) - len(m.shape)) + tuple(m.shape)
        if msh[-1] != cols:
            raise UnsupportedValue("mask must be contiguous along the softmax axis")
        # Broadcast patterns expressible as addend_row = (row // div) % mod:
        #   same shape; all-ones; prefix [B,1,1] (mask [B,1,1,S] vs scores [B,H,S,S]); suffix [1,1,S].
        # [B,1,S] (prefix + suffix) is run as one launch per leading index.
        lead = msh[:-1]
        qlead = tuple(qk.shape[:-1])
        flush = 1 if self.flush_nans_to_zero else 0
        out = DeviceTensor(ctx, qk.shape, np.float32)

        def prod(t):
            return int(np.prod(t, dtype=np.int64)) if len(t) else 1

        def pattern(ld, ql):
            if ld == ql:
                return 1, max(prod(ql), 1)
            if all(v == 1 for v in ld):
                return max(prod(ql), 1), 1
            i = len(ld)
            while i > 0 and ld[i - 1] == 1:
                i -= 1
            if ld[:i] == ql[:i]:
                return prod(ql[i:]), prod(ql[:i])
            j = 0
            while j < len(ld) and ld[j] == 1:
                j += 1
            if ld[j:] == ql[j:]:
                return 1, prod(ql[j:])
            return None

        pat = pattern(lead, qlead)
        if pat is not None:
            ctx.call("rten_hip_softmax_f32", rows, cols, qk.vp, m.vp, pat[0], pat[1], flush, out.vp)
            return [out]
        if len(lead) >= 2 and lead[0] == qlead[0]:
            sub = pattern(lead[1:], qlead[1:])
            if sub is not None:
                rows_b = prod(qlead[1:])
                m_rows_b = prod(lead[1:])
                for b in range(qlead[0]):
                    ctx.call("rten_hip_softmax_f32", rows_b, cols, C.c_void_p(qk.ptr + b * rows_b * cols * 4),
                             C.c_void_p(m.ptr + b * m_rows_b * cols * 4), sub[0], sub[1], flush,
                             C.c_void_p(out.ptr + b * rows_b * cols * 4))
                return [out]
        raise UnsupportedValue("unsupported mask broadcast pattern on the device path")


class LayerNormalization(Operator):
    """src/ops/norm.rs:531-580.  inputs: X, scale, bias?"""

    def __init__(self, axis=-1, epsilon=None):
        self.axis = axis
        self.epsilon = epsilon

    def max_inputs(self):
        return 3

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        scale = _want(_require(inputs, 1), np.float32)
        bias = _get(inputs, 2)
        ax = _resolve_axis(len(x.shape), self.axis)
        norm_shape = tuple(x.shape[ax:])
        cols = int(np.prod(norm_shape, dtype=np.int64))
        gamma = beta = None
        gs, bs = 1.0, 0.0
        if scale.size == 1 and len(scale.shape) == 0:
            gs = float(scale.numpy().reshape(()))
        else:
            if _broadcast_shapes(scale.shape, norm_shape) != norm_shape:
                raise InvalidValue("`scale` is not broadcastable to normalized axes of input")
            if int(np.prod(scale.shape, dtype=np.int64)) != cols:
                raise UnsupportedValue("device path needs scale already expanded to the normalized shape")
            gamma = scale
        if bias is not None:
            if bias.size == 1 and len(bias.shape) == 0:
                bs = float(bias.numpy().reshape(()))
            else:
                if _broadcast_shapes(bias.shape, norm_shape) != norm_shape:
                    raise InvalidValue("`bias` is not broadcastable to normalized axes of input")
                if int(np.prod(bias.shape, dtype=np.int64)) != cols:
                    raise UnsupportedValue("device path needs bias already expanded to the normalized shape")
                beta = bias
        y = DeviceTensor(ctx, x.shape, np.float32)
        eps = 1e-5 if self.epsilon is None else self.epsilon
        ctx.call("rten_hip_layer_norm_f32", x.size // max(cols, 1), cols, x.vp, _vp(gamma), _vp(beta), gs, bs, eps, y.vp)
        return [y]


class BatchNormalization(Operator):
    """src/ops/norm.rs:194-290.  inputs: X, scale, bias, mean, var"""

    def __init__(self, epsilon=1e-5):
        self.epsilon = epsilon

    def max_inputs(self):
        return 5

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        scale, bias, mean, var = (_require(inputs, i) for i in range(1, 5))
        if len(x.shape) < 1:
            raise InvalidValue("Input must have at least 1 dim")
        chans = x.shape[1] if len(x.shape) >= 2 else 1
        for t, nm in ((scale, "scale"), (bias, "bias"), (mean, "mean"), (var, "var")):
            if t.shape[0] != chans:
                raise IncompatibleInputShapes(f"{nm}.size(0) != channels")
        n = x.shape[0]
        inner = x.size // max(n * chans, 1)
        y = DeviceTensor(ctx, x.shape, np.float32)
        ctx.call("rten_hip_batch_norm_f32", n, chans, inner, x.vp, scale.vp, bias.vp, mean.vp, var.vp, self.epsilon, y.vp)
        return [y]


# ------------------------------------------------------------------------------------------ elementwise
class _Unary(Operator):
    fn = ""

    def max_inputs(self):
        return 1

    def run(self, ctx, inputs, in_place=False):
        x = _want(_require(inputs, 0), np.float32)
        y = x if in_place else DeviceTensor(ctx, x.shape, np.float32)
        ctx.call(self.fn, x.size, x.vp, y.vp)
        return [y]


class Relu(_Unary):
    """src/ops/unary_elementwise.rs:611-613"""
    fn = "rten_hip_relu_f32"


class Gelu(_Unary):
    """src/ops/unary_elementwise.rs:399-420 (exact erf form)"""
    fn = "rten_hip_gelu_f32"


class Erf(_Unary):
    fn = "rten_hip_erf_f32"


class _Binary(Operator):
    """binary_elementwise.rs:58-170,476-495: numpy broadcasting.  Equal shapes and trailing-dims / per-channel broadcasts take
    the flat 16-byte kernels; anything else goes through the general stride-0 kernel (rten_hip_binary_broadcast_f32)."""
    fn = ""
    opcode = 0
    commutative = True

    def max_inputs(self):
        return 2

    def run(self, ctx, inputs, in_place=False):
        a = _want(_require(inputs, 0), np.float32)
        b = _want(_require(inputs, 1), np.float32)
        if self.commutative and b.size > a.size:  # largest input is the in-place candidate (graph.rs:977-990)
            a, b = b, a
        bshape = _broadcast_shapes(a.shape, b.shape)
        if bshape is None:
            raise IncompatibleInputShapes("Cannot broadcast inputs")
        if bshape == tuple(a.shape):
            bsh = (1,) * (len(a.shape) - len(b.shape)) + tuple(b.shape)
            k = 0
            while k < len(bsh) and bsh[k] == 1:
                k += 1
            if tuple(bsh[k:]) == tuple(a.shape[k:]):  # trailing-dims broadcast (b == a's trailing block)
                y = a if in_place else DeviceTensor(ctx, a.shape, np.float32)
                ctx.call(self.fn, a.size, a.vp, b.vp, b.size, y.vp)
                return [y]
            if self.fn == "rten_hip_add_f32" and len(a.shape) >= 2 and bsh[1] == a.shape[1] and b.size == a.shape[1]:  # [1,C,1,1]
                y = a if in_place else DeviceTensor(ctx, a.shape, np.float32)
                inner = a.size // (a.shape[0] * a.shape[1])
                ctx.call("rten_hip_add_channel_bias_f32", a.shape[0], a.shape[1], inner, a.vp, b.vp, y.vp)
                return [y]
        nd = len(bshape)
        if nd > 6:
            raise UnsupportedValue("broadcasting over more than 6 dims is not supported by the device path")

        def strides(shape):
            sh, st, acc = (1,) * (nd - len(shape)) + tuple(shape), [0] * nd, 1
            for i in range(nd - 1, -1, -1):
                st[i] = 0 if sh[i] == 1 else acc
                acc *= sh[i]
            return st
        y = DeviceTensor(ctx, bshape, np.float32)
        i64 = lambda v: (C.c_int64 * max(len(v), 1))(*v)
        if y.size:
            ctx.call("rten_hip_binary_broadcast_f32", self.opcode, nd, i64(list(bshape)), i64(strides(a.shape)), i64(strides(b.shape)), a.vp, b.vp, y.vp)
        return [y]


class Add(_Binary):
    """src/ops/binary_elementwise.rs:476-495"""
    fn, opcode = "rten_hip_add_f32", 0


class Mul(_Binary):
    fn, opcode = "rten_hip_mul_f32", 1


class Sub(_Binary):
    fn, opcode, commutative = "rten_hip_sub_f32", 2, False


class Div(_Binary):
    fn, opcode, commutative = "rten_hip_div_f32", 3, False


class Transpose(Operator):
    """src/ops/layout.rs:669+: perm None = reverse the axes; 4-byte element types."""

    def __init__(self, perm=None):
        self.perm = perm

    def max_inputs(self):
        return 1

    def run(self, ctx, inputs):
        x = _require(inputs, 0)
        if x.dtype.itemsize != 4:
            raise UnsupportedType
        nd = len(x.shape)
        perm = list(range(nd - 1, -1, -1)) if self.perm is None else [p + nd if p < 0 else p for p in self.perm]
        if sorted(perm) != list(range(nd)):
            raise InvalidValue("Permutation is invalid")
        y = DeviceTensor(ctx, [x.shape[p] for p in perm], x.dtype)
        if nd > 6:
            raise UnsupportedValue("transpose of more than 6 dims is not supported by the device path")
        ctx.call("rten_hip_transpose_b32", nd, (C.c_int64 * max(nd, 1))(*x.shape), (C.c_int32 * max(nd, 1))(*perm), x.vp, y.vp)
        return [y]


# ------------------------------------------------------------------------------------------ pooling
class _Pool(Operator):
    fn = ""

    def __init__(self, kernel_size, padding=(0, 0, 0, 0), strides=None, ceil_mode=False, count_include_pad=False):
        self.kernel_size = list(kernel_size)
        self.padding = padding
        self.strides = list(strides) if strides is not None else [1] * len(self.kernel_size)
        self.ceil_mode = ceil_mode
        self.count_include_pad = count_include_pad

    def max_inputs(self):
        return 1

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        spatial = max(len(x.shape) - 2, 0)
        if len(self.kernel_size) != spatial:
            raise InvalidValue("kernel_size len does not match spatial dims")
        if len(self.strides) != spatial:
            raise InvalidValue("strides len does not match spatial dims")
        if spatial != 2:
            raise UnsupportedValue("Only inputs with 1 or 2 spatial dims are supported")
        n, c, h, w = x.shape
        oh, ow, pads = calc_output_size_and_padding(ctx, (h, w), self.kernel_size, self.strides, self.padding, (1, 1), self.ceil_mode)
        d = L.Pool2dDesc(n, c, h, w, self.kernel_size[0], self.kernel_size[1], self.strides[0], self.strides[1],
                         (C.c_int32 * 4)(*pads), oh, ow, 1 if self.count_include_pad else 0)
        y = DeviceTensor(ctx, (n, c, oh, ow), np.float32)
        ctx.call(self.fn, C.byref(d), x.vp, y.vp)
        return [y]


class MaxPool(_Pool):
    """src/ops/pooling.rs:602-660"""
    fn = "rten_hip_max_pool2d_f32"


class AveragePool(_Pool):
    """src/ops/pooling.rs:419-475"""
    fn = "rten_hip_average_pool2d_f32"


class GlobalAveragePool(Operator):
    """src/ops/pooling.rs:516-545"""

    def max_inputs(self):
        return 1

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        if len(x.shape) < 2:
            raise InvalidValue("Input must have at least 2 dims")
        n, c = x.shape[:2]
        inner = x.size // max(n * c, 1)
        y = DeviceTensor(ctx, (n, c) + (1,) * (len(x.shape) - 2), np.float32)
        ctx.call("rten_hip_global_average_pool_f32", n * c, inner, x.vp, y.vp)
        return [y]


class Flatten(Operator):
    """src/ops/layout.rs:264: zero-copy view on device."""

    def __init__(self, axis=1):
        self.axis = axis

    def run(self, ctx, inputs):
        x = _require(inputs, 0)
        ax = _resolve_axis(len(x.shape) + 1, self.axis) if self.axis != len(x.shape) else self.axis
        outer = int(np.prod(x.shape[:ax], dtype=np.int64))
        return [x.reshape(outer, -1)]


# ------------------------------------------------------------------------------------------ quantization
class DynamicQuantizeLinear(Operator):
    """src/ops/quantize.rs:438-478 -> [y u8, y_scale f32 scalar, y_zero_point u8 scalar] (all on device)."""

    def max_inputs(self):
        return 1

    def run(self, ctx, inputs):
        x = _want(_require(inputs, 0), np.float32)
        y = DeviceTensor(ctx, x.shape, np.uint8)
        scale = DeviceTensor(ctx, (), np.float32)
        zp = DeviceTensor(ctx, (), np.uint8)
        ctx.call("rten_hip_dynamic_quantize_linear", x.size, x.vp, y.vp, scale.vp, zp.vp)
        return [y, scale, zp]


# ------------------------------------------------------------------------------------------ attention
class Attention(Operator):
    """ONNX Attention restricted to the BERT path (src/ops/attention.rs:645-905): 4-D Q/K/V
    [B,H,S,D], optional additive float mask [B,1,1,T] or [B,1,S,T]; no KV cache / GQA / causal."""

    def __init__(self, scale=None):
        self.scale = scale

    def max_inputs(self):
        return 4

    def run(self, ctx, inputs):
        q = _want(_require(inputs, 0), np.float32)
        k = _want(_require(inputs, 1), np.float32)
        v = _want(_require(inputs, 2), np.float32)
        mask = _get(inputs, 3)
        if len(q.shape) != 4 or len(k.shape) != 4 or len(v.shape) != 4:
            raise UnsupportedValue("device Attention needs 4-D Q/K/V")
        B, H, S, D = q.shape
        T, Dv = k.shape[2], v.shape[3]
        if k.shape[:2] != (B, H) or v.shape[:3] != (B, H, T) or k.shape[3] != D:
            raise IncompatibleInputShapes("Q/K/V shapes are inconsistent")
        # `1.0 / (head_size as f32).sqrt()` in f32 arithmetic (attention.rs:659-670)
        scale = self.scale if self.scale is not None else float(np.float32(1.0) / np.sqrt(np.float32(D)))
        mbs, mrs = 0, 0
        if mask is not None:
            if mask.dtype != np.float32:
                raise UnsupportedValue("device Attention supports additive float masks")
            if tuple(mask.shape) == (B, 1, 1, T):
                mbs, mrs = T, 0
            elif tuple(mask.shape) == (B, 1, S, T):
                mbs, mrs = S * T, T
            else:
                raise UnsupportedValue("mask must be [B,1,1,T] or [B,1,S,T]")
        out = DeviceTensor(ctx, (B, H, S, Dv), np.float32)
        d = L.SdpaDesc(B, H, S, T, D, Dv, H * S * D, S * D, D, H * T * D, T * D, D, H * T * Dv, T * Dv, Dv,
                       H * S * Dv, S * Dv, Dv, mbs, mrs, scale, 1)
        ctx.call("rten_hip_sdpa_f32", C.byref(d), q.vp, k.vp, v.vp, _vp(mask), out.vp)
        return [out]


class Gather(Operator):
    """Gather along axis 0 of a 2-D f32 table (embedding lookup; src/ops/gather.rs).  inputs: table [R,W], ids i32."""

    def __init__(self, axis=0):
        self.axis = axis

    def max_inputs(self):
        return 2

    def run(self, ctx, inputs):
        table = _want(_require(inputs, 0), np.float32)
        ids = _want(_require(inputs, 1), np.int32)
        if self.axis != 0 or len(table.shape) != 2:
            raise UnsupportedValue("device Gather supports axis 0 of a 2-D table")
        out = DeviceTensor(ctx, tuple(ids.shape) + (table.shape[1],), np.float32)
        ctx.call("rten_hip_gather_rows_f32", ids.size, table.shape[1], table.shape[0], table.vp, ids.vp, out.vp)
        return [out]


class ReduceSum(Operator):
    """src/ops/reduce.rs:1126-1165 (f32): axes attribute or second input already resolved by the caller into `axes`;
    the slice of every output element is summed in place through the view's strides, 16-lane vecmath::Sum order."""
    mean = False

    def __init__(self, axes=None, keep_dims=True, noop_with_empty_axes=False):
        self.axes, self.keep_dims, self.noop_with_empty_axes = axes, keep_dims, noop_with_empty_axes

    def max_inputs(self):
        return 2

    def run(self, ctx, inputs):
        from . import einsum as E
        x = _want(_require(inputs, 0), np.float32)
        nd = len(x.shape)
        if not self.axes and self.noop_with_empty_axes:
            return [E.materialize(ctx, E.View(x))]
        axes = []
        for a in (self.axes if self.axes else range(nd)):
            if a < -nd or a >= nd:
                raise InvalidValue("Axis is invalid")
            axes.append(a + nd if a < 0 else a)
        axes = sorted(set(axes))  # resolve_axes sorts and dedups (src/ops/mod.rs:259-271)
        y = E.reduce_sum(ctx, E.View(x), axes, mean=self.mean) if nd else E.materialize(ctx, E.View(x))  # rank 0: Sum = Mean = x (slice of one)
        if self.keep_dims:
            y = y.reshape([1 if d in axes else x.shape[d] for d in range(nd)])
        return [y]


class ReduceMean(ReduceSum):
    """src/ops/reduce.rs:523-580: vecmath::Sum(slice) / len(slice) as f32 (an empty slice gives NaN)."""
    mean = True


class Einsum(Operator):
    """src/ops/einsum.rs:21-108: any number of f32 inputs, `equation` attribute.  The planner and the strided lowering
    are in rten_amd/einsum.py."""

    def __init__(self, equation: str):
        self.equation = equation

    def max_inputs(self):
        return None

    def run(self, ctx, inputs):
        from . import einsum as E
        xs = [_want(_require(inputs, i), np.float32) for i in range(len(inputs))]
        return [E.einsum(ctx, xs, self.equation)]


class OpRegistry:
    """Mirror of OpRegistry (src/op_registry.rs:25-72): op_type -> operator class for the hot path."""

    def __init__(self):
        self._ops = {}

    @classmethod
    def with_all_ops(cls):
        r = cls()
        for op in (Conv, ConvTranspose, ConvInteger, ConvIntegerToFloat, MatMul, FusedMatMul, Gemm, MatMulInteger, MatMulIntegerToFloat, MatMulNBits,
                   Softmax, AddSoftmax, LayerNormalization, BatchNormalization, Relu, Gelu, Erf, Add, Mul, Sub, Div, Transpose, MaxPool,
                   AveragePool, GlobalAveragePool, Flatten, DynamicQuantizeLinear, Attention, Gather, ReduceSum, ReduceMean, Einsum):
            r.register_op(op)
        return r

    def register_op(self, op_cls):
        self._ops[op_cls.__name__] = op_cls

    def get(self, op_type: str):
        return self._ops.get(op_type)

    def op_types(self):
        return sorted(self._ops)
