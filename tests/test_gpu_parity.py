"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (DESIGN.md section "Parity"):
  * integer / byte work (int8 GEMM + conv, DQL codes, zero points): bit-exact
  * f32 GEMM / conv (M > 1), element-wise ops, pooling: bit-exact vs the oracle's restatement of the
    reference's accumulation order
  * reductions (softmax, LayerNorm, GlobalAveragePool): bit-exact vs oracle lanes=16 (AVX-512 order);
    tolerance 1e-6 relative vs other lane counts
  * M == 1 GEMM (reference gemv path): tolerance 1e-5 relative (tests state it where used)
"""
import ctypes as C

import numpy as np
import pytest

from oracle import ref
from rten_amd import lib as L
from rten_amd import ops
from rten_amd.tensor import DeviceTensor

pytestmark = pytest.mark.gpu


def dev(ctx, a):
    return DeviceTensor.from_numpy(ctx, a)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        # +0 / -0 compare equal; NaNs must coincide
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        if not same.all():
            idx = np.argwhere(~same)[0]
            raise AssertionError(f"{(~same).sum()} of {a.size} elements differ; first at {tuple(idx)}: {a[tuple(idx)]!r} vs {b[tuple(idx)]!r}; "
                                 f"max abs diff {np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))}")
    else:
        assert np.array_equal(a, b), f"{(a != b).sum()} of {a.size} elements differ"


def gpu_gemm(ctx, a, b, c=None, alpha=1.0, beta=0.0, bias=None, bias_kind=0, act=0, variant=None):
    M, K = a.shape
    N = b.shape[1]
    ad, bd = dev(ctx, a), dev(ctx, b)  # contiguous copies of whatever layout `a`/`b` views have
    # preserve the logical strides of the views
    a_rs, a_cs = (a.strides[0] // 4, a.strides[1] // 4) if a.size else (K, 1)
    b_rs, b_cs = (b.strides[0] // 4, b.strides[1] // 4) if b.size else (N, 1)
    if not (a.flags.c_contiguous or a.T.flags.c_contiguous):
        raise AssertionError("test helper supports plain or transposed views")
    if a.T.flags.c_contiguous and not a.flags.c_contiguous:
        ad = dev(ctx, np.ascontiguousarray(a.T))
    if b.T.flags.c_contiguous and not b.flags.c_contiguous:
        bd = dev(ctx, np.ascontiguousarray(b.T))
    cd = dev(ctx, c if c is not None else np.full((M, N), np.nan, np.float32))
    biasd = dev(ctx, bias) if bias is not None else None
    d = L.gemm_desc(M, N, K, a_rs, a_cs, b_rs, b_cs, N, alpha=alpha, beta=beta, bias_kind=bias_kind if bias is not None else 0, act=act)
    if variant is not None:
        ctx.set_gemm_variant(variant)
    ctx.call("rten_hip_gemm_f32", C.byref(d), ad.vp, bd.vp, biasd.vp if biasd else None, cd.vp)
    ctx.set_gemm_variant(-1)
    return cd.numpy()


# ------------------------------------------------------------------------------------------ f32 GEMM
@pytest.mark.parametrize("m", [2, 8, 10, 64, 80, 130])
def test_gemm_f32_bit_exact_size_matrix(ctx, m):
    # size matrix of rten-gemm/src/tests.rs:336-362 (rows x depth x cols)
    rng = ref.XorShiftRng(1234 + m)
    for k in (0, 1, 2, 20, 256, 300, 700):
        for n in (1, 2, 4, 5, 8, 129, 1024, 1025):
            a = rng.f32(m * k).reshape(m, k) - 0.5
            b = rng.f32(k * n).reshape(k, n) - 0.5
            bits_equal(gpu_gemm(ctx, a, b), ref.gemm_f32(a, b))


def test_gemm_f32_all_tile_variants_agree(ctx):
    rng = ref.XorShiftRng(5)
    a = rng.f32(200 * 520).reshape(200, 520) - 0.5
    b = rng.f32(520 * 264).reshape(520, 264) - 0.5
    want = ref.gemm_f32(a, b)
    for v in range(ctx.lib.rten_hip_num_gemm_variants()):
        bits_equal(gpu_gemm(ctx, a, b, variant=v), want)


def test_gemm_f32_options_bit_exact(ctx):
    rng = ref.XorShiftRng(7)
    a = rng.f32(72 * 300).reshape(72, 300) - 0.5
    b = rng.f32(300 * 132).reshape(300, 132) - 0.5
    c = rng.f32(72 * 132).reshape(72, 132)
    br, bc = rng.f32(72), rng.f32(132)
    for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (0.5, 0.0), (0.5, 2.0), (0.125, 1.0)):
        bits_equal(gpu_gemm(ctx, a, b, c=c, alpha=alpha, beta=beta), ref.gemm_f32(a, b, c=c, alpha=alpha, beta=beta))
    # beta == 0 never reads C (NaN-poisoned): tests.rs:632-674
    assert not np.isnan(gpu_gemm(ctx, a, b, c=np.full((72, 132), np.nan, np.float32), beta=0.0)).any()
    bits_equal(gpu_gemm(ctx, a, b, bias=br, bias_kind=L.BIAS_PER_ROW), ref.gemm_f32(a, b, bias=br, bias_kind=ref.BIAS_PER_ROW))
    bits_equal(gpu_gemm(ctx, a, b, bias=bc, bias_kind=L.BIAS_PER_COL), ref.gemm_f32(a, b, bias=bc, bias_kind=ref.BIAS_PER_COL))
    # transposed operands are strides (tests.rs:523-569); all four loader combinations
    at, bt = np.ascontiguousarray(a.T).T, np.ascontiguousarray(b.T).T
    want = ref.gemm_f32(a, b)
    for aa, bb in ((a, b), (at, b), (a, bt), (at, bt)):
        bits_equal(gpu_gemm(ctx, aa, bb), want)
    # fused activations == the separate ops
    bits_equal(gpu_gemm(ctx, a, b, bias=bc, bias_kind=L.BIAS_PER_COL, act=L.ACT_RELU), ref.relu(ref.gemm_f32(a, b, bias=bc, bias_kind=ref.BIAS_PER_COL)))
    bits_equal(gpu_gemm(ctx, a, b, bias=bc, bias_kind=L.BIAS_PER_COL, act=L.ACT_GELU), ref.gelu(ref.gemm_f32(a, b, bias=bc, bias_kind=ref.BIAS_PER_COL)))


def test_gemm_f32_split_k_bit_exact(ctx):
    # exact split-K on the generic GEMM path (every operand layout, alpha/beta/bias/act in the fixup epilogue);
    # the default automatic plan splits skinny products such as the ResNet classifier [32, 2048] x [2048, 1000]
    rng = ref.XorShiftRng(11)
    try:
        for (m, k, n) in ((32, 2048, 1000), (70, 700, 130), (2, 1030, 5)):
            a = rng.f32(m * k).reshape(m, k) - 0.5
            b = rng.f32(k * n).reshape(k, n) - 0.5
            c = rng.f32(m * n).reshape(m, n)
            bc = rng.f32(n)
            at, bt = np.ascontiguousarray(a.T).T, np.ascontiguousarray(b.T).T
            want = ref.gemm_f32(a, b)
            want2 = ref.gelu(ref.gemm_f32(a, b, c=c, alpha=0.5, beta=2.0, bias=bc, bias_kind=ref.BIAS_PER_COL))
            for mode, groups in ((3, 1), (2, 2), (2, 3), (2, 64), (1, 4)):
                ctx.call("rten_hip_set_gemm_split", mode, groups)
                for aa, bb in ((a, b), (at, b), (a, bt), (at, bt)):
                    bits_equal(gpu_gemm(ctx, aa, bb), want)
                bits_equal(gpu_gemm(ctx, a, bt, c=c, alpha=0.5, beta=2.0, bias=bc, bias_kind=L.BIAS_PER_COL, act=L.ACT_GELU), want2)
                for v in (0, 3):
                    bits_equal(gpu_gemm(ctx, a, b, variant=v), want)
    finally:
        ctx.call("rten_hip_set_gemm_split", 3, 1)


def test_gemm_f32_one_row_follows_the_reference_gemv_order(ctx):
    """M == 1: the reference takes its gemv kernels (rten-gemm/src/lib.rs:876-891, simd_generic.rs:14-197) whose order is not the
    blocked GEMM's; rten_hip_gemm_f32 follows them bit for bit (row-major / transposed / strided B, alpha / beta / bias / activation,
    left-over columns of a column block, the stated reference thread count), and with the order switched off (prepacked weights in
    the reference) a one-row product is a row of the blocked result."""
    rng = ref.XorShiftRng(11)
    for (k, n) in ((1, 1), (7, 5), (8, 32), (20, 45), (530, 40), (2048, 1000), (33, 300), (600, 137)):
        a = rng.f32(k).reshape(1, k) - 0.5
        b = rng.f32(k * n).reshape(k, n) - 0.5
        bt = np.ascontiguousarray(b.T).T  # unit ROW stride: the transposed kernel
        c = rng.f32(n).reshape(1, n) - 0.5
        bias = rng.f32(n) - 0.5
        for bb in (b, bt):
            bits_equal(gpu_gemm(ctx, a, bb), ref.gemm_f32(a, bb))
            bits_equal(gpu_gemm(ctx, a, bb, c=c, alpha=0.5, beta=2.0, bias=bias, bias_kind=L.BIAS_PER_COL),
                       ref.gemm_f32(a, bb, c=c, alpha=0.5, beta=2.0, bias=bias, bias_kind=ref.BIAS_PER_COL))
            bits_equal(gpu_gemm(ctx, a, bb, c=c, alpha=1.0, beta=1.0), ref.gemm_f32(a, bb, c=c, alpha=1.0, beta=1.0))
            bits_equal(gpu_gemm(ctx, a, bb, bias=bias, bias_kind=L.BIAS_PER_COL, act=L.ACT_GELU), ref.gelu(ref.gemm_f32(a, bb, bias=bias, bias_kind=ref.BIAS_PER_COL)))
            bits_equal(gpu_gemm(ctx, a, bb, alpha=0.25), ref.gemm_f32(a, bb, alpha=0.25))
        # against the truth, by tolerance (what round 1 could claim)
        want = a.astype(np.float64) @ b.astype(np.float64)
        assert np.abs(gpu_gemm(ctx, a, b) - want).max() <= 1e-5 * max((np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)).max(), 1e-30)
    # neither stride is 1: the fallback kernel (a strided view staged as a wider matrix on the device)
    k, n = 37, 50
    wide = rng.f32(2 * k * 3 * n).reshape(2 * k, 3 * n) - 0.5
    bs = wide[::2, ::3]
    a = rng.f32(k).reshape(1, k) - 0.5
    wd, ad = dev(ctx, wide), dev(ctx, a)
    cd = dev(ctx, np.full((1, n), np.nan, np.float32))
    d = L.gemm_desc(1, n, k, k, 1, 2 * 3 * n, 3, n)
    ctx.call("rten_hip_gemm_f32", C.byref(d), ad.vp, wd.vp, None, cd.vp)
    bits_equal(cd.numpy(), ref.gemm_f32(a, bs))
    # the reference's thread count is part of the order (which columns are left over in a column block) ...
    k, n = 300, 1000
    a = rng.f32(k).reshape(1, k) - 0.5
    b = rng.f32(k * n).reshape(k, n) - 0.5
    try:
        for threads in (4, 3, 1):
            ctx.call("rten_hip_set_gemv_order", 1, threads)
            ref.set_gemv_threads(threads)
            for bb in (b, np.ascontiguousarray(b.T).T):
                bits_equal(gpu_gemm(ctx, a, bb), ref.gemm_f32(a, bb))
        # ... and switched off, a one-row product is the blocked GEMM's row
        ctx.call("rten_hip_set_gemv_order", 0, 0)
        bits_equal(gpu_gemm(ctx, a, b), ref.gemm_f32(np.concatenate([a, a]), b)[:1])
    finally:
        ctx.call("rten_hip_set_gemv_order", 1, 0)
        ref.set_gemv_threads(0)
    # batched one-row products (one gemv per batch element, matmul.rs:299-385)
    qa = rng.f32(3 * 1 * 40).reshape(3, 1, 40) - 0.5
    qb = rng.f32(3 * 40 * 24).reshape(3, 40, 24) - 0.5
    got = ops.MatMul().run(ctx, [DeviceTensor.from_numpy(ctx, qa), DeviceTensor.from_numpy(ctx, qb)])[0].numpy()
    bits_equal(got, ref.matmul_f32(qa, qb))


def test_resnet50_batch1_logits_bit_exact(ctx):
    """BASELINE configs[0] (batch 1): the classifier is a one-row Gemm(transB), i.e. the reference's transposed gemv kernel."""
    from oracle import models as omodels
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    net = resnet50.ResNet50(ctx, batch=1, weights=w)
    net.upload_weights()
    x = ref.XorShiftRng(8).f32(3 * 224 * 224).reshape(1, 3, 224, 224)
    net.x.upload(x)
    net.forward()
    bits_equal(net.logits.numpy(), omodels.resnet50_forward(net.specs, w, x))
    net.capture()
    net.logits.upload(np.zeros((1, 1000), np.float32))
    net.run()
    bits_equal(net.logits.numpy(), omodels.resnet50_forward(net.specs, w, x))


def test_matmul_ops_broadcast_and_batched(ctx):
    # src/ops/matmul.rs:1096-1183 shape/broadcast matrix (the device-supported subset)
    rng = ref.XorShiftRng(13)

    def r(*s):
        return rng.f32(int(np.prod(s))).reshape(s) - 0.5
    for ash, bsh in (((3, 10), (10, 8)), ((2, 3, 10), (10, 8)), ((4, 5, 10), (4, 10, 8)), ((2, 2, 5, 10), (10, 8)), ((10,), (10, 8)), ((3, 10), (10,))):
        a, b = r(*ash), r(*bsh)
        got = ops.MatMul().run(ctx, [dev(ctx, a), dev(ctx, b)])[0].numpy()
        want = ref.matmul_f32(a, b) if a.ndim > 1 and b.ndim > 1 else None
        if want is None:
            want = (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
        elif a.shape[-2] == 1 or (a.ndim == 2 and a.shape[0] == 1):
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
        else:
            bits_equal(got, want)
    # FusedMatMul: bias + alpha (matmul.rs:1242-1281)
    a, b, bias = r(6, 12, 40), r(40, 24), r(24)
    got = ops.FusedMatMul(alpha=0.125).run(ctx, [dev(ctx, a), dev(ctx, b), dev(ctx, bias)])[0].numpy()
    bits_equal(got, ref.matmul_f32(a, b, alpha=0.125, bias=bias))
    # Gemm op: transB + C broadcast with beta (ResNet fc), matmul.rs:32-104
    x, w, cb = r(32, 200), r(50, 200), r(50)
    got = ops.Gemm(alpha=1.0, beta=1.0, transpose_b=True).run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, cb)])[0].numpy()
    want = ref.gemm_f32(x, w.T, c=np.broadcast_to(cb, (32, 50)).astype(np.float32), alpha=1.0, beta=1.0)
    bits_equal(got, want)


# ------------------------------------------------------------------------------------------ f32 conv
def gpu_conv(ctx, x, w, bias=None, pads=(0, 0, 0, 0), strides=(1, 1), dilations=(1, 1), groups=1, residual=None, relu=False, prepack=True, variant=None):
    op = ops.Conv(groups=groups, dilations=dilations, padding=list(pads), strides=strides, fuse_relu=relu)
    xd, wd = dev(ctx, x), dev(ctx, w)
    packed = None
    if prepack:
        d = op._geometry(ctx, x.shape, w.shape)
        packed = op.prepack(ctx, wd, d)
    ins = [xd, wd, dev(ctx, bias) if bias is not None else None, dev(ctx, residual) if residual is not None else None]
    if variant is not None:
        ctx.set_gemm_variant(variant)
    y = op.run(ctx, ins, packed_weight=packed)[0].numpy()
    ctx.set_gemm_variant(-1)
    return y


CONV_CASES = [
    # N, C, H, W, O, kh, kw, pads, strides, dil, groups   (src/ops/conv.rs:885-1319 sweeps)
    (2, 8, 9, 11, 6, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1),
    (1, 4, 12, 12, 8, 3, 3, (0, 1, 2, 1), (2, 2), (1, 1), 2),
    (3, 6, 10, 7, 6, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1),
    (2, 16, 8, 8, 12, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1),      # pointwise, P % 4 == 0 -> dense vectorised path
    (2, 16, 7, 7, 12, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1),      # pointwise, P = 49 -> gather path
    (2, 16, 8, 8, 8, 1, 1, (0, 0, 0, 0), (2, 2), (1, 1), 1),       # 1x1 stride-2 downsample
    (1, 3, 32, 32, 16, 7, 7, (3, 3, 3, 3), (2, 2), (1, 1), 1),     # stem-like, K = 147
    (1, 2, 14, 14, 2, 3, 3, (2, 2, 2, 2), (1, 1), (2, 2), 1),      # dilation
    (2, 4, 6, 6, 4, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 4),        # depthwise shape through the generic path
    (1, 40, 9, 9, 70, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1),      # K = 360 > 256: two depth blocks
    (5, 1, 5, 5, 1, 5, 5, (2, 2, 2, 2), (1, 1), (1, 1), 1),
    (1, 8, 1, 20, 5, 1, 3, (0, 1, 0, 1), (1, 1), (1, 1), 1),       # 1-D conv expanded to 2-D (conv.rs:142-182)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_f32_bit_exact(ctx, case):
    N, Cc, H, W, O, kh, kw, pads, strides, dil, groups = case
    rng = ref.XorShiftRng(1234)
    x = rng.f32(N * Cc * H * W).reshape(N, Cc, H, W) - 0.5
    w = rng.f32(O * (Cc // groups) * kh * kw).reshape(O, Cc // groups, kh, kw) - 0.5
    b = rng.f32(O) - 0.5
    want = ref.conv2d_f32(x, w, b, pads=pads, strides=strides, dilations=dil, groups=groups)
    for prepack in (True, False):
        bits_equal(gpu_conv(ctx, x, w, b, pads, strides, dil, groups, prepack=prepack), want)
    res = rng.f32(want.size).reshape(want.shape) - 0.5
    bits_equal(gpu_conv(ctx, x, w, b, pads, strides, dil, groups, residual=res, relu=True),
               ref.conv2d_f32(x, w, b, pads=pads, strides=strides, dilations=dil, groups=groups, residual=res, relu=True))
    bits_equal(gpu_conv(ctx, x, w, None, pads, strides, dil, groups, relu=True),
               ref.conv2d_f32(x, w, None, pads=pads, strides=strides, dilations=dil, groups=groups, relu=True))


def test_conv_1d_expands_to_2d(ctx):
    # conv.rs:142-182: [N, C, W] x [O, C/g, kw] runs as the 2-D convolution with H = kh = 1
    rng = ref.XorShiftRng(8)
    x = rng.f32(2 * 8 * 20).reshape(2, 8, 20) - 0.5
    w = rng.f32(5 * 8 * 3).reshape(5, 8, 3) - 0.5
    b = rng.f32(5) - 0.5
    got = ops.Conv(padding=[1, 2], strides=[2], dilations=[1]).run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, b)])[0].numpy()
    want = ref.conv2d_f32(x[:, :, None, :], w[:, :, None, :], b, pads=(0, 1, 0, 2), strides=(1, 2))[:, :, 0, :]
    bits_equal(got, want)
    with pytest.raises(ops.OpError, match="expected 1 stride value"):
        ops.Conv(padding=[1, 2], strides=[1, 1], dilations=[1]).run(ctx, [dev(ctx, x), dev(ctx, w)])
    with pytest.raises(ops.OpError, match="kernel must have 3 dims"):
        ops.Conv(padding=[0, 0], strides=[1], dilations=[1]).run(ctx, [dev(ctx, x), dev(ctx, w[:, :, None, :])])


def test_conv_transpose(ctx):
    # literals of test_conv_transpose / _padding / _1d (conv_transpose.rs:603-760) and bit-exact sweeps of the cases the
    # reference's own tests run (groups, dilations, output_padding, uneven padding, empty input ranges)
    x = np.array([1.0, 2.0, 3.0, 4.0], np.float32).reshape(1, 1, 2, 2)
    k = np.array([0.1, 0.2, 0.3, 0.4], np.float32).reshape(1, 1, 2, 2)
    got = ops.ConvTranspose(strides=[2, 2]).run(ctx, [dev(ctx, x), dev(ctx, k)])[0].numpy()
    np.testing.assert_allclose(got.ravel(), [0.1, 0.2, 0.2, 0.4, 0.3, 0.4, 0.6, 0.8, 0.3, 0.6, 0.4, 0.8, 0.9, 1.2, 1.2, 1.6], rtol=1e-6)
    got = ops.ConvTranspose(strides=[2, 2]).run(ctx, [dev(ctx, x), dev(ctx, k), dev(ctx, np.array([1.234], np.float32))])[0].numpy()
    np.testing.assert_allclose(got.ravel(), np.array([0.1, 0.2, 0.2, 0.4, 0.3, 0.4, 0.6, 0.8, 0.3, 0.6, 0.4, 0.8, 0.9, 1.2, 1.2, 1.6]) + 1.234, rtol=1e-6)
    np.testing.assert_allclose(ops.ConvTranspose(padding=[1, 1, 1, 1], strides=[2, 2]).run(ctx, [dev(ctx, x), dev(ctx, k)])[0].numpy().ravel(), [0.4, 0.6, 0.6, 0.4], rtol=1e-6)
    assert ops.ConvTranspose(padding="same", strides=[2, 2]).run(ctx, [dev(ctx, x), dev(ctx, k)])[0].shape == (1, 1, 4, 4)
    got = ops.ConvTranspose(padding=[0, 0], strides=[2], dilations=[1]).run(ctx, [dev(ctx, x.reshape(1, 1, 4)[:, :, :2].copy()), dev(ctx, k.reshape(1, 1, 4)[:, :, :2].copy())])[0].numpy()
    np.testing.assert_allclose(got.ravel(), [0.1, 0.2, 0.2, 0.4], rtol=1e-6)
    rng = np.random.default_rng(12)
    for (N, Cc, H, W, og, kh, kw, pads, strides, dil, groups, opad) in ((2, 6, 5, 7, 4, 3, 2, (1, 0, 2, 1), (2, 3), (1, 2), 2, (1, 0)), (1, 4, 8, 8, 3, 4, 4, (1, 1, 1, 1), (2, 2), (1, 1), 1, (0, 0)),
                                                                          (3, 3, 1, 9, 5, 1, 3, (0, 2, 0, 2), (1, 1), (1, 1), 3, (0, 0)), (1, 300, 6, 6, 8, 2, 2, (0, 0, 0, 0), (2, 2), (1, 1), 1, (1, 1)),
                                                                          (1, 2, 3, 3, 2, 3, 3, (2, 2, 2, 2), (3, 3), (1, 1), 1, (0, 0))):
        x = rng.random((N, Cc, H, W), dtype=np.float32) - 0.5
        w = rng.random((Cc, og, kh, kw), dtype=np.float32) - 0.5
        b = rng.random(og * groups, dtype=np.float32) - 0.5
        op = ops.ConvTranspose(padding=list(pads), groups=groups, strides=list(strides), dilations=list(dil), output_padding=list(opad))
        bits_equal(op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, b)])[0].numpy(), ref.conv_transpose2d_f32(x, w, b, pads, strides, dil, groups, opad))
        bits_equal(op.run(ctx, [dev(ctx, x), dev(ctx, w)])[0].numpy(), ref.conv_transpose2d_f32(x, w, None, pads, strides, dil, groups, opad))
    with pytest.raises(ops.OpError, match="Input channels does not match kernel input channels"):
        ops.ConvTranspose().run(ctx, [dev(ctx, x), dev(ctx, rng.random((5, 2, 3, 3), dtype=np.float32))])
    with pytest.raises(ops.OpError, match="Input is too small"):
        ops.ConvTranspose(padding=[9, 9, 9, 9]).run(ctx, [dev(ctx, x), dev(ctx, w)])


def test_matmul_nbits(ctx):
    # the reference's seeded cases (block_quant.rs:940-1072) plus wider sweeps of both paths, bit-exact against the oracle
    def case(n_rows, n_cols, n_blocks, bs, batch=(), seed=1234):
        rng = ref.XorShiftRng(seed)
        lead = int(np.prod(batch, dtype=np.int64))
        lhs = (rng.f32(lead * n_rows * n_blocks * bs) - np.float32(0.5)).reshape(*batch, n_rows, n_blocks * bs)
        quant = rng.u8(n_cols * n_blocks * (bs // 2)).reshape(n_cols, n_blocks, bs // 2)
        scales = rng.f32(n_cols * n_blocks).reshape(n_cols, n_blocks)
        return lhs, quant, scales

    cases = [(1, 3, max(128 // bs, 1), bs, ()) for bs in (16, 32, 64, 128, 256)]
    cases += [(1, 1, 128 // 16 + 1, 16, ()),     # one vector step + a scalar tail
              (1, 37, 24, 32, ()),               # K = 768: six 128-element steps, ragged column count
              (1, 9, 7, 16, (3,)),               # batch of vectors, K = 112: tail only
              (1, 16, 3, 64, (2, 2)),            # K = 192: one step + 64-element tail
              (1, 5, 70, 128, ()),               # K = 8960: crosses the 8192-element LDS chunk
              (1, 8, 33, 256, ()),               # K = 8448
              (1, 8192, 10, 32, ()), (1, 16384 + 5, 3, 128, ()), (1, 4100, 9, 16, (2,)),  # wide matrices: the 4- and 8-slots-per-lane launches
              (4, 21, 6, 32, ()), (7, 130, 3, 128, (2,)), (33, 64, 2, 16, ())]  # rows > 1: the f32 GEMM on the expanded weights
    for n_rows, n_cols, n_blocks, bs, batch in cases:
        lhs, quant, scales = case(n_rows, n_cols, n_blocks, bs, batch)
        got = ops.MatMulNBits(4, bs).run(ctx, [dev(ctx, lhs), dev(ctx, quant), dev(ctx, scales)])[0].numpy()
        bits_equal(got, ref.matmul_nbits_f32(lhs, quant, scales))
    # 1-D scales (contrib.rs:152-164) and K == 0
    lhs, quant, scales = case(1, 4, 2, 32)
    got = ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, lhs), dev(ctx, quant), dev(ctx, scales.reshape(-1))])[0].numpy()
    bits_equal(got, ref.matmul_nbits_f32(lhs, quant, scales))
    z = ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, np.zeros((2, 1, 0), np.float32)), dev(ctx, np.zeros((3, 0, 16), np.uint8)), dev(ctx, np.zeros((3, 0), np.float32))])[0]
    assert z.shape == (2, 1, 3) and not z.numpy().any()
    # error behaviour of the op (contrib.rs:29-61, 141-178)
    with pytest.raises(ops.OpError, match="A input must have at least 2 dims"):
        ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, lhs.reshape(-1)), dev(ctx, quant), dev(ctx, scales)])
    with pytest.raises(ops.OpError, match="Columns of first matrix does not match rows of second matrix"):
        ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, lhs[:, :32].copy()), dev(ctx, quant), dev(ctx, scales)])
    with pytest.raises(ops.OpError, match="Unsupported K block size"):
        ops.MatMulNBits(4, 8).run(ctx, [dev(ctx, lhs), dev(ctx, quant.reshape(4, 8, 4).copy()), dev(ctx, np.zeros((4, 8), np.float32))])
    with pytest.raises(ops.OpError, match="Unsupported bits-per-element"):
        ops.MatMulNBits(2, 32).run(ctx, [dev(ctx, lhs), dev(ctx, quant), dev(ctx, scales)])
    with pytest.raises(ops.OpError, match="Expected `scales` to have one or two dims"):
        ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, lhs), dev(ctx, quant), dev(ctx, scales.reshape(1, 4, 2))])
    with pytest.raises(ops.OpError, match="zero_points, g_idx and bias inputs are unsupported"):
        ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, lhs), dev(ctx, quant), dev(ctx, scales), dev(ctx, np.zeros(4, np.uint8))])


def test_matmul_nbits_decoder_sized(ctx):
    # an LLM-decoder-sized projection (K = N = 4096, block 32) quantised from real-valued weights: vector path bit-exact against the
    # oracle and close to the f64 product with the dequantised matrix; the 16-row prefill path within the f32 GEMM tolerance
    rng = np.random.default_rng(21)
    w = (rng.standard_normal((4096, 1024)) * 0.02).astype(np.float32)
    quant, scales = ref.quantize_4bit_blocks(w, 32)
    deq = ref.dequantize_4bit(quant, scales).astype(np.float64)
    x = rng.standard_normal((1, 4096)).astype(np.float32)
    got = ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, x), dev(ctx, quant), dev(ctx, scales)])[0].numpy()
    bits_equal(got, ref.matmul_nbits_f32(x, quant, scales))
    want = x.astype(np.float64) @ deq
    assert np.max(np.abs(got - want)) <= 1e-4 * np.max(np.abs(want)) + 1e-5
    xs = rng.standard_normal((16, 4096)).astype(np.float32)
    got = ops.MatMulNBits(4, 32).run(ctx, [dev(ctx, xs), dev(ctx, quant), dev(ctx, scales)])[0].numpy()
    bits_equal(got, ref.matmul_nbits_f32(xs, quant, scales))


def test_conv_f32_reference_literals(ctx):
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_literals.json")))["conv"]
    k = np.array(g["kernel"], np.float32).reshape(1, 1, 3, 3)
    x = np.array(g["input"], np.float32).reshape(1, 1, 3, 3)
    np.testing.assert_allclose(gpu_conv(ctx, x, k, pads=(1, 1, 1, 1)).ravel(), g["expected_same_padding"], atol=1e-4)
    np.testing.assert_allclose(gpu_conv(ctx, x, k).ravel(), g["expected_no_padding"], atol=1e-4)
    np.testing.assert_allclose(gpu_conv(ctx, x, k, bias=np.array([1.0], np.float32)).ravel(), g["expected_with_bias_1"], atol=1e-4)


@pytest.mark.parametrize("shape", [(64, 64, 56, 3, 1, 1), (256, 64, 56, 1, 1, 0), (128, 128, 28, 3, 1, 1), (512, 256, 28, 1, 2, 0),
                                   (256, 256, 14, 3, 1, 1), (2048, 512, 7, 1, 1, 0), (512, 512, 7, 3, 1, 1), (64, 3, 224, 7, 2, 3)])
def test_conv_f32_resnet_layer_shapes_all_variants(ctx, shape):
    # real ResNet-50 layer geometries at batch 2, every tile variant
    O, Cc, H, k, s, pad = shape
    rng = ref.XorShiftRng(77)
    x = rng.f32(2 * Cc * H * H).reshape(2, Cc, H, H) - 0.5
    w = (rng.f32(O * Cc * k * k).reshape(O, Cc, k, k) - 0.5) * 0.1
    b = rng.f32(O) - 0.5
    want = ref.conv2d_f32(x, w, b, pads=(pad,) * 4, strides=(s, s), relu=True)
    for v in range(ctx.lib.rten_hip_num_gemm_variants()):  # 0..3 LDS-DMA pipeline, 4..7 register-staged
        bits_equal(gpu_conv(ctx, x, w, b, (pad,) * 4, (s, s), relu=True, variant=v), want)
    try:
        ctx.call("rten_hip_set_gemm_order", 1)  # tiles walk n fastest
        for v in (0, 3, 7, 11, 24, 25):
            bits_equal(gpu_conv(ctx, x, w, b, (pad,) * 4, (s, s), relu=True, variant=v), want)
    finally:
        ctx.call("rten_hip_set_gemm_order", 0)


@pytest.mark.parametrize("shape", [(128, 128, 28, 3, 1, 1), (256, 256, 14, 3, 1, 1), (512, 512, 7, 3, 1, 1), (512, 2048, 7, 1, 1, 0),
                                   (128, 512, 28, 1, 1, 0), (40, 70, 9, 3, 1, 1),
                                   (64, 32, 96, 3, 1, 1)])  # last: 288 tiles of 64x64 -> tail mode splits only tiles 256..287
def test_conv_f32_split_k_bit_exact(ctx, shape):
    # exact split-K: K cut at the reference's depth-block boundaries (rten-gemm/src/lib.rs:630-633), partial sums
    # added in block order -> same bits as the unsplit chain, with bias / residual / relu in the fixup epilogue
    O, Cc, H, k, s, pad = shape
    rng = ref.XorShiftRng(99)
    x = rng.f32(2 * Cc * H * H).reshape(2, Cc, H, H) - 0.5
    w = (rng.f32(O * Cc * k * k).reshape(O, Cc, k, k) - 0.5) * 0.1
    b = rng.f32(O) - 0.5
    want0 = ref.conv2d_f32(x, w, b, pads=(pad,) * 4, strides=(s, s), relu=True)
    res = rng.f32(want0.size).reshape(want0.shape) - 0.5
    want = ref.conv2d_f32(x, w, b, pads=(pad,) * 4, strides=(s, s), residual=res, relu=True)
    nblk = (Cc * k * k + 255) // 256
    try:
        for v in (0, 1, 2, 3, 24, 25, 26, 30):  # 24..26: one wave per 64x64 tile (gemm_f32_wave.hip); 30: image patches (gemm_f32_patch.hip)
            for mode in (1, 2):
                for groups in sorted({2, 3, nblk}):
                    ctx.call("rten_hip_set_gemm_split", mode, groups)
                    for order in ((0, 1, 2, 3) if v in (3, 24) else (0, 3)):  # workgroup -> tile orders
                        ctx.call("rten_hip_set_gemm_order", order)
                        bits_equal(gpu_conv(ctx, x, w, b, (pad,) * 4, (s, s), residual=res, relu=True, variant=v), want)
                    ctx.call("rten_hip_set_gemm_order", 0)
        ctx.call("rten_hip_set_gemm_split", 2, 2)
        bits_equal(gpu_conv(ctx, x, w, None, (pad,) * 4, (s, s), variant=3),
                   ref.conv2d_f32(x, w, None, pads=(pad,) * 4, strides=(s, s)))
    finally:
        ctx.call("rten_hip_set_gemm_split", 3, 1)
        ctx.call("rten_hip_set_gemm_order", 0)


def test_conv_f32_grouped_split_k_bit_exact(ctx):
    # grouped convolution (grid.y = group) with K = 576 per group: split-K slabs are indexed per group
    rng = ref.XorShiftRng(123)
    x = rng.f32(2 * 128 * 12 * 12).reshape(2, 128, 12, 12) - 0.5
    w = (rng.f32(96 * 64 * 3 * 3).reshape(96, 64, 3, 3) - 0.5) * 0.1
    b = rng.f32(96) - 0.5
    want = ref.conv2d_f32(x, w, b, pads=(1, 1, 1, 1), groups=2, relu=True)
    try:
        for v in (3, 7, 1, 24):
            for mode, groups in ((0, 1), (2, 2), (2, 3), (1, 3)):
                ctx.call("rten_hip_set_gemm_split", mode, groups)
                bits_equal(gpu_conv(ctx, x, w, b, (1, 1, 1, 1), (1, 1), (1, 1), 2, relu=True, variant=v), want)
    finally:
        ctx.call("rten_hip_set_gemm_split", 3, 1)


# ------------------------------------------------------------------------------------------ int8
@pytest.fixture(params=[0, 1], ids=["i8staged", "i8generic"])
def i8path(request, ctx):
    """Both int8 implementations: k-contiguous staging + LDS-DMA kernel (automatic) and the generic byte-gather kernel."""
    ctx.call("rten_hip_set_int8_path", request.param)
    yield request.param
    ctx.call("rten_hip_set_int8_path", 0)


@pytest.mark.parametrize("adt,bdt", [(np.uint8, np.int8), (np.uint8, np.uint8), (np.int8, np.int8), (np.int8, np.uint8)])
def test_matmul_integer_bit_exact(ctx, i8path, adt, bdt):
    # src/ops/matmul.rs:1365-1750
    rng = ref.XorShiftRng(1234)
    for (m, n, k) in ((1, 1, 1), (5, 7, 3), (16, 33, 64), (70, 130, 300), (64, 64, 128), (3, 1000, 17), (200, 260, 520), (130, 64, 1)):
        a = (rng.u8(m * k) if adt == np.uint8 else rng.i8(m * k)).reshape(m, k)
        b = (rng.u8(k * n) if bdt == np.uint8 else rng.i8(k * n)).reshape(k, n)
        zps = [(None, None), (np.array(3, adt), np.array(5, bdt)),
               ((rng.u8(m) if adt == np.uint8 else rng.i8(m)), (rng.u8(n) if bdt == np.uint8 else rng.i8(n)))]
        for a_zp, b_zp in zps:
            ins = [dev(ctx, a), dev(ctx, b), dev(ctx, a_zp) if a_zp is not None else None, dev(ctx, b_zp) if b_zp is not None else None]
            got = ops.MatMulInteger().run(ctx, ins)[0].numpy()
            bits_equal(got, ref.gemm_int8(a, b, a_zp, b_zp))
    # MatMulIntegerToFloat with scalar and per-column scale (matmul.rs:718-729)
    a = rng.u8(20 * 48).reshape(20, 48)
    b = rng.i8(48 * 12, reduced=True).reshape(48, 12)
    azp, bzp = np.array(7, np.uint8), rng.i8(12, reduced=True)
    acc = ref.gemm_int8(a, b, azp, bzp)
    for scale in (np.array(0.02, np.float32), (rng.f32(12) * 0.1).astype(np.float32)):
        got = ops.MatMulIntegerToFloat().run(ctx, [dev(ctx, a), dev(ctx, b), dev(ctx, azp), dev(ctx, bzp), dev(ctx, scale)])[0].numpy()
        bits_equal(got, ref.cast_scale(acc, scale))


@pytest.mark.parametrize("xdt,wdt", [(np.uint8, np.int8), (np.uint8, np.uint8), (np.int8, np.int8), (np.int8, np.uint8)])
def test_conv_integer_bit_exact(ctx, i8path, xdt, wdt):
    # src/ops/conv.rs:1370-1527 (all four signedness combos) + padded cases for every pad_mode
    rng = ref.XorShiftRng(42)
    for (N, Cc, H, W, O, k, pads, strides, groups) in ((2, 4, 7, 6, 6, 3, (0, 0, 0, 0), (1, 1), 1), (1, 8, 9, 9, 4, 3, (1, 1, 1, 1), (2, 2), 2),
                                                        (2, 16, 7, 7, 20, 1, (0, 0, 0, 0), (1, 1), 1), (1, 3, 20, 20, 8, 7, (3, 3, 3, 3), (2, 2), 1),
                                                        (3, 40, 13, 11, 70, 3, (1, 0, 2, 1), (1, 1), 1), (2, 64, 14, 14, 130, 3, (1, 1, 1, 1), (1, 1), 1),
                                                        (2, 128, 8, 8, 32, 1, (0, 0, 0, 0), (2, 2), 1)):
        x = (rng.u8(N * Cc * H * W) if xdt == np.uint8 else rng.i8(N * Cc * H * W)).reshape(N, Cc, H, W)
        wn = O * (Cc // groups) * k * k
        w = (rng.u8(wn, reduced=True) if wdt == np.uint8 else rng.i8(wn, reduced=True)).reshape(O, Cc // groups, k, k)
        x_zp = np.array(rng.u8(1)[0] if xdt == np.uint8 else rng.i8(1)[0], xdt)
        for w_zp in (None, np.array(2, wdt), (rng.u8(O, reduced=True) if wdt == np.uint8 else rng.i8(O, reduced=True))):
            for pm in (L.PAD_ZERO_POINT, L.PAD_RAW0_I8, L.PAD_RAW0_U8):
                op = ops.ConvInteger(groups=groups, padding=list(pads), strides=strides, pad_mode=pm)
                got = op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, x_zp), dev(ctx, w_zp) if w_zp is not None else None])[0].numpy()
                want = ref.conv2d_int8(x, w, x_zp=int(x_zp), w_zp=w_zp, pads=pads, strides=strides, groups=groups, pad_mode=pm)
                bits_equal(got, want)


def test_conv_integer_to_float_fused_epilogue(ctx, i8path):
    # test_conv_integer_to_float (conv.rs:1530-1604) + the ort-quantized graph tail: Add(bias) -> Add(residual) -> Relu
    rng = ref.XorShiftRng(8)
    x = rng.u8(2 * 8 * 10 * 10).reshape(2, 8, 10, 10)
    w = rng.i8(12 * 8 * 3 * 3, reduced=True).reshape(12, 8, 3, 3)
    x_zp, scale = np.array(120, np.uint8), np.array(0.0123, np.float32)
    bias = rng.f32(12) - 0.5
    acc = ref.conv2d_int8(x, w, x_zp=120, pads=(1, 1, 1, 1), pad_mode=ref.PAD_RAW0_I8)
    f = ref.cast_scale(acc, scale)
    res = rng.f32(f.size).reshape(f.shape) - 0.5
    op = ops.ConvIntegerToFloat(ops.ConvInteger(padding=[1, 1, 1, 1]))
    got = op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, x_zp), None, dev(ctx, scale)])[0].numpy()
    bits_equal(got, f)
    op = ops.ConvIntegerToFloat(ops.ConvInteger(padding=[1, 1, 1, 1]), fuse_relu=True)
    got = op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, x_zp), None, dev(ctx, scale), dev(ctx, bias), dev(ctx, res)])[0].numpy()
    want = ref.relu(ref.add(f + bias[None, :, None, None], res))
    bits_equal(got, want)


def test_dynamic_quantize_linear_bit_exact(ctx):
    rng = ref.XorShiftRng(3)
    for x in (rng.f32(100000) * 6 - 3, rng.f32(4097), -rng.f32(1001), np.zeros(64, np.float32), np.array([-234.56], np.float32),
              np.arange(-0.1, 0.1, 0.001, dtype=np.float32), (rng.f32(3 * 64 * 28 * 28) - 0.2).reshape(3, 64, 28, 28)):
        y, s, z = ops.DynamicQuantizeLinear().run(ctx, [dev(ctx, x)])
        qy, qs, qz = ref.dynamic_quantize_linear(x)
        assert s.numpy().reshape(()) == qs and z.numpy().reshape(()) == qz
        bits_equal(y.numpy(), qy)
    y, s, z = ops.DynamicQuantizeLinear().run(ctx, [dev(ctx, np.zeros((0,), np.float32))])
    assert float(s.numpy()) == 1.0 and int(z.numpy()) == 0


def test_int8_resnet_block_chain(ctx, i8path):
    # DQL -> ConvIntegerToFloat(+bias, relu) chained on device: scale = x_scale * w_scale computed by a device Mul
    rng = ref.XorShiftRng(17)
    x = (rng.f32(2 * 16 * 14 * 14) - 0.3).reshape(2, 16, 14, 14)
    w = rng.i8(24 * 16 * 3 * 3, reduced=True).reshape(24, 16, 3, 3)
    w_scale, bias = np.array(0.004, np.float32), rng.f32(24) - 0.5
    xq, xs, xz = ops.DynamicQuantizeLinear().run(ctx, [dev(ctx, x)])
    sc = ops.Mul().run(ctx, [xs.reshape(1), dev(ctx, w_scale.reshape(1))])[0]
    got = ops.ConvIntegerToFloat(ops.ConvInteger(padding=[1, 1, 1, 1]), fuse_relu=True).run(
        ctx, [xq, dev(ctx, w), xz, None, sc.reshape(()), dev(ctx, bias)])[0].numpy()
    q, s, z = ref.dynamic_quantize_linear(x)
    acc = ref.conv2d_int8(q, w, x_zp=int(z), pads=(1, 1, 1, 1), pad_mode=ref.PAD_RAW0_I8)
    want = ref.relu(ref.cast_scale(acc, np.float32(s * w_scale)) + bias[None, :, None, None])
    bits_equal(got, want)


@pytest.mark.parametrize("pm", [L.PAD_ZERO_POINT, L.PAD_RAW0_I8, L.PAD_RAW0_U8])
def test_dql_staged_conv_bit_exact(ctx, pm):
    # DynamicQuantizeLinear fused with the consumer's activation staging + prepacked weights == the separate ops
    rng = ref.XorShiftRng(23)
    for (N, Cc, H, W, O, k, pad, stride) in ((2, 20, 9, 11, 24, 3, 1, 1), (1, 64, 14, 14, 70, 3, 1, 2), (3, 3, 16, 16, 8, 7, 3, 2), (2, 32, 7, 7, 16, 1, 0, 1)):
        x = (rng.f32(N * Cc * H * W) - 0.4).reshape(N, Cc, H, W) * 3
        w = rng.i8(O * Cc * k * k, reduced=True).reshape(O, Cc, k, k)
        w_scale, bias = np.array([0.003], np.float32), rng.f32(O) - 0.5
        oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        cd = L.Conv2dDesc(N, Cc, H, W, O, k, k, (C.c_int32 * 4)(pad, pad, pad, pad), stride, stride, 1, 1, 1, oh, ow)
        d = L.Conv2dInt8Desc(cd, 0, 1, 0, pm, 1, 1)
        staged = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d)),), np.uint8)
        packed = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_int8_packed_bytes(C.byref(d)),), np.uint8)
        xd, wd, bd, wsd = dev(ctx, x), dev(ctx, w), dev(ctx, bias), dev(ctx, w_scale)
        xs, xz, sc = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8), DeviceTensor(ctx, (1,), np.float32)
        out = DeviceTensor(ctx, (N, O, oh, ow), np.float32)
        ctx.call("rten_hip_conv2d_int8_prepack", C.byref(d), wd.vp, packed.vp)
        if k == 1:  # the following Mul(x_scale, w_scale) either folded into the staging kernel or as its own op
            ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), xd.vp, staged.vp, xs.vp, xz.vp, None, None)
            ctx.call("rten_hip_mul_f32", 1, xs.vp, wsd.vp, 1, sc.vp)
        else:
            ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), xd.vp, staged.vp, xs.vp, xz.vp, wsd.vp, sc.vp)
        ctx.call("rten_hip_conv2d_int8", C.byref(d), staged.vp, packed.vp, xz.vp, None, sc.vp, bd.vp, None, L.CONV_RELU, out.vp)
        q, s, z = ref.dynamic_quantize_linear(x)
        assert xs.numpy()[0] == s and xz.numpy()[0] == z
        acc = ref.conv2d_int8(q, w, x_zp=int(z), pads=(pad,) * 4, strides=(stride,) * 2, pad_mode=pm)
        want = ref.relu(ref.cast_scale(acc, np.float32(np.float32(s) * w_scale[0])) + bias[None, :, None, None])
        bits_equal(out.numpy(), want)
        # per-output-channel weight scales: ConvInteger -> Cast -> Mul([1,O,1,1]) (the form the reference leaves unfused)
        wsv = (rng.f32(O) * 0.01 + 0.001).astype(np.float32)
        scv = DeviceTensor(ctx, (O,), np.float32)
        wsd = dev(ctx, wsv)  # (named: a temporary would be freed before the launch reads it)
        ctx.call("rten_hip_mul_f32", O, wsd.vp, xs.vp, 1, scv.vp)
        dv = L.Conv2dInt8Desc(cd, 0, 1, 0, pm, 1, 1, O)
        ctx.call("rten_hip_conv2d_int8", C.byref(dv), staged.vp, packed.vp, xz.vp, None, scv.vp, bd.vp, None, 0, out.vp)
        sv = (wsv * np.float32(s)).astype(np.float32)
        want = (acc.astype(np.float32) * sv[None, :, None, None]) + bias[None, :, None, None]
        bits_equal(out.numpy(), want)
        ctx.call("rten_hip_set_int8_path", 1)  # generic kernel, plain operands
        dg = L.Conv2dInt8Desc(cd, 0, 1, 0, pm, 0, 0, O)
        qd = dev(ctx, q)
        ctx.call("rten_hip_conv2d_int8", C.byref(dg), qd.vp, wd.vp, xz.vp, None, scv.vp, bd.vp, None, 0, out.vp)
        ctx.sync()
        ctx.call("rten_hip_set_int8_path", 0)
        bits_equal(out.numpy(), want)


# ------------------------------------------------------------------------------------------ row-wise / element-wise / pooling
@pytest.mark.parametrize("cols", [1, 3, 6, 16, 17, 64, 100, 128, 129, 200, 256, 257, 384, 768, 1000, 1024, 1500, 3000])
def test_softmax_bit_exact_avx512_order(ctx, cols):
    rng = ref.XorShiftRng(cols)
    x = (rng.f32(37 * cols).reshape(37, cols) - 0.5) * 8
    got = ops.Softmax(axis=-1).run(ctx, [dev(ctx, x)])[0].numpy()
    bits_equal(got, ref.softmax(x, lanes=16))
    np.testing.assert_allclose(got, ref.softmax(x, lanes=4), rtol=1e-6, atol=0)


def test_softmax_other_axes(ctx):
    # normalize_lanes (norm.rs:705-754): the axis is moved last, the lanes processed, and moved back
    rng = ref.XorShiftRng(31)
    x = (rng.f32(3 * 5 * 7 * 4).reshape(3, 5, 7, 4) - 0.5) * 6
    for axis in (0, 1, 2, -2, -4, 3):
        got = ops.Softmax(axis=axis).run(ctx, [dev(ctx, x)])[0].numpy()
        want = np.moveaxis(ref.softmax(np.ascontiguousarray(np.moveaxis(x, axis, -1)), lanes=16), -1, axis)
        bits_equal(got, want)


def test_add_softmax_bert_mask(ctx):
    rng = ref.XorShiftRng(2)
    qk = (rng.f32(2 * 12 * 128 * 128).reshape(2, 12, 128, 128) - 0.5) * 4
    mask = np.where(rng.f32(2 * 128).reshape(2, 1, 1, 128) > 0.2, 0.0, -np.inf).astype(np.float32)
    got = ops.AddSoftmax(flush_nans_to_zero=True).run(ctx, [dev(ctx, qk), dev(ctx, mask)])[0].numpy()
    bits_equal(got, ref.softmax(qk, addend=mask, add_div=12 * 128, add_mod=2, flush_nan=True))
    m2 = rng.f32(2 * 128 * 128).reshape(2, 1, 128, 128)
    # [B,1,S,S] mask: per-batch addend rows
    want = np.stack([ref.softmax(qk[b], addend=m2[b], add_div=1, add_mod=128) for b in range(2)])
    got = np.stack([ops.AddSoftmax().run(ctx, [dev(ctx, qk[b:b + 1]), dev(ctx, m2[b:b + 1])])[0].numpy()[0] for b in range(2)])
    bits_equal(got, want)
    # all -inf rows: NaN unless flushed (attention.rs:1088-1106)
    ninf = np.full((1, 3), -np.inf, np.float32)
    assert np.isnan(ops.AddSoftmax().run(ctx, [dev(ctx, ninf), dev(ctx, np.zeros((1, 3), np.float32))])[0].numpy()).all()
    assert ops.AddSoftmax(flush_nans_to_zero=True).run(ctx, [dev(ctx, ninf), dev(ctx, np.zeros((1, 3), np.float32))])[0].numpy().tolist() == [[0, 0, 0]]


@pytest.mark.parametrize("cols", [2, 10, 64, 100, 256, 768, 1024, 2000])
def test_layer_norm_bit_exact_avx512_order(ctx, cols):
    rng = ref.XorShiftRng(cols + 1)
    x = (rng.f32(29 * cols).reshape(29, cols) - 0.5) * 3
    g, b = rng.f32(cols) + 0.5, rng.f32(cols) - 0.5
    for gamma, beta, gs, bs in ((g, b, 1.0, 0.0), (g, None, 1.0, 0.0), (None, None, 2.0, 0.5)):
        ins = [dev(ctx, x), dev(ctx, gamma) if gamma is not None else dev(ctx, np.array(gs, np.float32)),
               (dev(ctx, beta) if beta is not None else (dev(ctx, np.array(bs, np.float32)) if bs else None))]
        got = ops.LayerNormalization(axis=-1).run(ctx, ins)[0].numpy()
        bits_equal(got, ref.layer_norm(x, gamma, beta, gs, bs, lanes=16))
        np.testing.assert_allclose(got, ref.layer_norm(x, gamma, beta, gs, bs, lanes=4), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("cols", [10, 100, 768, 1024, 2000])
def test_add_layer_norm_fused_bit_exact(ctx, cols):
    # Add -> LayerNormalization as one kernel == the two operators, also when the output aliases an input (BERT's residual stream)
    rng = ref.XorShiftRng(cols + 7)
    x = (rng.f32(29 * cols).reshape(29, cols) - 0.5) * 3
    r = (rng.f32(29 * cols).reshape(29, cols) - 0.5) * 2
    g, b = rng.f32(cols) + 0.5, rng.f32(cols) - 0.5
    want = ref.layer_norm(ref.add(x, r), g, b, 1.0, 0.0, eps=1e-12, lanes=16)
    xd, rd, gd, bd = dev(ctx, x), dev(ctx, r), dev(ctx, g), dev(ctx, b)
    out = DeviceTensor(ctx, (29, cols), np.float32)
    ctx.call("rten_hip_add_layer_norm_f32", 29, cols, xd.vp, rd.vp, gd.vp, bd.vp, 1.0, 0.0, 1e-12, out.vp)
    bits_equal(out.numpy(), want)
    ctx.call("rten_hip_add_layer_norm_f32", 29, cols, xd.vp, rd.vp, gd.vp, bd.vp, 1.0, 0.0, 1e-12, rd.vp)  # in place over the addend
    bits_equal(rd.numpy(), want)


def test_layer_norm_reference_literals(ctx):
    import json, os
    for c in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_literals.json")))["layer_norm"]["cases"]:
        x = np.array(c["input"], np.float32)
        if "scale" not in c:
            continue
        got = ops.LayerNormalization(axis=-1).run(ctx, [dev(ctx, x), dev(ctx, np.array(c["scale"], np.float32)), dev(ctx, np.array(c["bias"], np.float32))])[0].numpy()
        np.testing.assert_allclose(got, c["expected"], atol=1e-4)


def test_elementwise_bit_exact(ctx):
    rng = ref.XorShiftRng(9)
    x = np.concatenate([(rng.f32(100003) - 0.5) * 12, np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 88.0, -88.0], np.float32)])
    bits_equal(ops.Gelu().run(ctx, [dev(ctx, x)])[0].numpy(), ref.gelu(x))
    bits_equal(ops.Erf().run(ctx, [dev(ctx, x)])[0].numpy(), ref.erf(x))
    r = ops.Relu().run(ctx, [dev(ctx, x)])[0].numpy()
    bits_equal(r, ref.relu(x))
    assert r[-5] == 0.0  # relu(NaN) == 0 (f32::max semantics, unary_elementwise.rs:611-613)
    y = (rng.f32(x.size) - 0.5).astype(np.float32)
    bits_equal(ops.Add().run(ctx, [dev(ctx, x), dev(ctx, y)])[0].numpy(), x + y)
    a4 = rng.f32(2 * 6 * 5 * 5).reshape(2, 6, 5, 5)
    cb = rng.f32(6).reshape(1, 6, 1, 1)
    bits_equal(ops.Add().run(ctx, [dev(ctx, a4), dev(ctx, cb)])[0].numpy(), a4 + cb)
    bits_equal(ops.Add().run(ctx, [dev(ctx, a4), dev(ctx, a4[0, 0])])[0].numpy(), a4 + a4[0, 0])
    sc, bi, me, va = rng.f32(6) + 0.5, rng.f32(6), rng.f32(6), rng.f32(6) + 0.1
    bits_equal(ops.BatchNormalization().run(ctx, [dev(ctx, a4), dev(ctx, sc), dev(ctx, bi), dev(ctx, me), dev(ctx, va)])[0].numpy(),
               ref.batch_norm(a4, sc, bi, me, va))
    acc = rng.i32(1000).reshape(10, 100)
    for s in (np.array([0.37], np.float32), rng.f32(100)):
        out = DeviceTensor(ctx, acc.shape, np.float32)
        accd, sd = dev(ctx, acc), dev(ctx, s)  # keep the device buffers alive across the async launch
        ctx.call("rten_hip_cast_scale", acc.size, accd.vp, sd.vp, s.size, out.vp)
        bits_equal(out.numpy(), ref.cast_scale(acc, s))


def test_pooling_bit_exact(ctx):
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_literals.json")))["pool"]
    x4 = np.array(g["input4"], np.float32).reshape(1, 1, 4, 4)
    for c in g["max"]:
        np.testing.assert_allclose(ops.MaxPool(c["kernel"], strides=c["strides"]).run(ctx, [dev(ctx, x4)])[0].numpy()[0, 0], c["expected"], rtol=1e-6)
    for c in g["average"]:
        np.testing.assert_allclose(ops.AveragePool(c["kernel"], strides=c["strides"]).run(ctx, [dev(ctx, x4)])[0].numpy()[0, 0], c["expected"], rtol=1e-5)
    rng = ref.XorShiftRng(4)
    x = rng.f32(3 * 5 * 23 * 19).reshape(3, 5, 23, 19) - 0.5
    for k, s, p, ceil in (((3, 3), (2, 2), (1, 1, 1, 1), False), ((2, 2), (2, 2), (0, 0, 0, 0), True), ((3, 2), (1, 2), (1, 0, 1, 1), False)):
        bits_equal(ops.MaxPool(k, padding=list(p), strides=s, ceil_mode=ceil).run(ctx, [dev(ctx, x)])[0].numpy(), ref.max_pool(x, k, s, p, ceil))
        for cip in (False, True):
            bits_equal(ops.AveragePool(k, padding=list(p), strides=s, ceil_mode=ceil, count_include_pad=cip).run(ctx, [dev(ctx, x)])[0].numpy(),
                       ref.average_pool(x, k, s, p, cip, ceil))
    # the stem pool's form (3x3, stride 2, left padding 1, even width): paired 8-byte loads + the neighbour lane's value; planes whose work items
    # do not fill whole workgroups (the workgroup size follows the plane)
    for shape, p in (((2, 3, 20, 22), (1, 1, 1, 1)), ((1, 2, 112, 112), (1, 1, 1, 1)), ((1, 3, 18, 140), (0, 1, 1, 0)), ((2, 2, 9, 6), (1, 1, 0, 1))):
        xe = rng.f32(int(np.prod(shape))).reshape(shape) - 0.5
        bits_equal(ops.MaxPool((3, 3), padding=list(p), strides=(2, 2)).run(ctx, [dev(ctx, xe)])[0].numpy(), ref.max_pool(xe, (3, 3), (2, 2), p, False))
        for cip in (False, True):
            bits_equal(ops.AveragePool((3, 3), padding=list(p), strides=(2, 2), count_include_pad=cip).run(ctx, [dev(ctx, xe)])[0].numpy(),
                       ref.average_pool(xe, (3, 3), (2, 2), p, cip, False))
    for inner in ((7, 7), (1, 1), (4, 4), (9, 9), (30, 30)):
        xg = rng.f32(2 * 10 * inner[0] * inner[1]).reshape(2, 10, *inner)
        bits_equal(ops.GlobalAveragePool().run(ctx, [dev(ctx, xg)])[0].numpy(), ref.global_average_pool(xg, lanes=16))


def test_sdpa_bit_exact(ctx):
    rng = ref.XorShiftRng(21)
    for (B, H, S, D) in ((2, 3, 16, 8), (2, 12, 128, 64), (1, 2, 40, 24)):
        q = rng.f32(B * H * S * D).reshape(B, H, S, D) - 0.5
        k = rng.f32(B * H * S * D).reshape(B, H, S, D) - 0.5
        v = rng.f32(B * H * S * D).reshape(B, H, S, D) - 0.5
        mask = np.where(rng.f32(B * S).reshape(B, 1, 1, S) > 0.3, 0.0, -np.inf).astype(np.float32)
        for m in (None, mask):
            ins = [dev(ctx, q), dev(ctx, k), dev(ctx, v)] + ([dev(ctx, m)] if m is not None else [])
            bits_equal(ops.Attention().run(ctx, ins)[0].numpy(), ref.sdpa(q, k, v, mask=m, lanes=16))


@pytest.mark.parametrize("path", [0, 1], ids=["fused", "composed"])
def test_sdpa_head64_shapes_bit_exact(ctx, path):
    # head size 64: the fused attention kernel (key length <= 128) and the composed GEMM / softmax / GEMM path, same bits.
    # Ragged query / key lengths, both mask forms, fully masked rows with and without the NaN flush.
    rng = ref.XorShiftRng(31)
    ctx.call("rten_hip_set_sdpa_path", path)
    try:
        for (B, H, S, T) in ((1, 2, 40, 40), (2, 3, 200, 100), (1, 1, 128, 128), (2, 2, 33, 7), (1, 2, 64, 129)):
            q = rng.f32(B * H * S * 64).reshape(B, H, S, 64) - 0.5
            k = rng.f32(B * H * T * 64).reshape(B, H, T, 64) - 0.5
            v = rng.f32(B * H * T * 64).reshape(B, H, T, 64) - 0.5
            m1 = np.where(rng.f32(B * T).reshape(B, 1, 1, T) > 0.3, 0.0, -np.inf).astype(np.float32)
            m1[0, 0, 0, :] = -np.inf  # a fully masked batch item: NaN rows unless flushed
            m2 = ((rng.f32(B * S * T).reshape(B, 1, S, T) - 0.5) * 4).astype(np.float32)
            for m in (None, m1, m2):
                for flush in (True, False):
                    mbs, mrs = (0, 0) if m is None else ((T, 0) if m.shape[2] == 1 else (S * T, T))
                    d = L.SdpaDesc(B, H, S, T, 64, 64, H * S * 64, S * 64, 64, H * T * 64, T * 64, 64, H * T * 64, T * 64, 64, H * S * 64, S * 64, 64,
                                   mbs, mrs, 0.125, 1 if flush else 0)
                    out = DeviceTensor(ctx, (B, H, S, 64), np.float32)
                    qd, kd, vd = dev(ctx, q), dev(ctx, k), dev(ctx, v)
                    md = dev(ctx, m) if m is not None else None
                    ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, md.vp if md is not None else None, out.vp)
                    bits_equal(out.numpy(), ref.sdpa(q, k, v, mask=m, scale=0.125, lanes=16, flush_nan=flush))
    finally:
        ctx.call("rten_hip_set_sdpa_path", 0)


# ------------------------------------------------------------------------------------------ end to end
def test_resnet50_end_to_end_bit_exact_and_graph_replay(ctx):
    from oracle import models as omodels
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    net = resnet50.ResNet50(ctx, batch=2, weights=w)
    net.upload_weights()
    x = ref.XorShiftRng(1234).f32(2 * 3 * 224 * 224).reshape(2, 3, 224, 224)  # U[0,1) like rten-cli (input_generator.rs:139-144)
    net.x.upload(x)
    net.forward()
    logits = net.logits.numpy()
    want, acts = omodels.resnet50_forward(net.specs, w, x, return_activations=True)
    for name in ("stem", "pool", "s0b0out", "s1b3out", "s2b5out", net.specs[-1]["dst"]):
        if name in (net.specs[-1]["dst"],):  # earlier buffers have been recycled by the static plan
            bits_equal(net.bufs[name].numpy(), acts[name])
    bits_equal(logits, want)
    assert np.array_equal(np.argsort(-logits, 1)[:, :5], np.argsort(-want, 1)[:, :5])
    # hipGraph replay == eager
    net.capture()
    net.logits.upload(np.zeros_like(logits))
    net.run()
    bits_equal(net.logits.numpy(), want)
    # autotuned variants leave the result unchanged
    net.graph = None
    net.autotune(reps=1)
    net.forward()
    bits_equal(net.logits.numpy(), want)
    # projection shortcuts on a second context (stream): eager and as parallel hipGraph branches
    net.concurrent = True
    net.logits.upload(np.zeros_like(logits))
    net.forward()
    bits_equal(net.logits.numpy(), want)
    net.capture()
    net.logits.upload(np.zeros_like(logits))
    net.run()
    bits_equal(net.logits.numpy(), want)


def test_resnet50_int8_end_to_end_bit_exact(ctx):
    # BASELINE configs[2]: dynamically quantized ResNet-50 (DynamicQuantizeLinear -> ConvIntegerToFloat chain per conv)
    from oracle import models as omodels
    from rten_amd.workloads import resnet50, resnet50_int8
    w = resnet50.make_weights()
    net = resnet50_int8.ResNet50Int8(ctx, batch=2, weights=w)
    net.upload_weights()
    x = ref.XorShiftRng(99).f32(2 * 3 * 224 * 224).reshape(2, 3, 224, 224)
    net.x.upload(x)
    net.forward()
    got = net.logits.numpy()
    qw = omodels.quantize_weights_int8(w)
    for name in ("stem", "s2b1c2", "fc"):
        assert np.array_equal(qw[name][0], net.q[name][0]) and qw[name][1] == net.q[name][1]
    want = omodels.resnet50_int8_forward(net.specs, qw, x)
    bits_equal(got, want)
    net.capture()
    net.logits.upload(np.zeros_like(got))
    net.run()
    bits_equal(net.logits.numpy(), want)
    # two-sweep DynamicQuantizeLinear everywhere (no producer-side statistics): same bits
    net.graph, net.producer_stats = None, False
    net.logits.upload(np.zeros_like(got))
    net.forward()
    bits_equal(net.logits.numpy(), want)
    # the generic int8 kernel gives the same bits through the plain (unpacked) operator path
    ctx.call("rten_hip_set_int8_path", 1)
    try:
        l = net.specs[5]
        wq, ws, b = net.q[l["name"]]
        xin = (ref.XorShiftRng(5).f32(int(np.prod(net.shapes[l["src"]]))) - 0.3).reshape(net.shapes[l["src"]])
        q, s, z = ref.dynamic_quantize_linear(xin)
        op = ops.ConvInteger(padding=[l["pad"]] * 4, strides=(l["stride"],) * 2)
        acc = op.run(ctx, [dev(ctx, q), dev(ctx, wq), dev(ctx, np.array(z, np.uint8)), None])[0].numpy()
        bits_equal(acc, ref.conv2d_int8(q, wq, x_zp=int(z), pads=(l["pad"],) * 4, strides=(l["stride"],) * 2, pad_mode=ref.PAD_RAW0_I8))
    finally:
        ctx.call("rten_hip_set_int8_path", 0)


def test_resnet50_batch32_batch_independence(ctx):
    # BASELINE config 2 at full size: every image of a batch-32 run equals the oracle's batch-1 run of that image
    from oracle import models as omodels
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    net = resnet50.ResNet50(ctx, batch=32, weights=w)
    net.upload_weights()
    x = ref.XorShiftRng(99).f32(32 * 3 * 224 * 224).reshape(32, 3, 224, 224)
    net.x.upload(x)
    net.forward()
    logits = net.logits.numpy()
    assert np.isfinite(logits).all()
    for i in (0, 13, 31):
        bits_equal(logits[i:i + 1], omodels.resnet50_forward(net.specs, w, x[i:i + 1]) if False else omodels.resnet50_forward(net.specs, w, np.concatenate([x[i:i + 1], x[i:i + 1]]))[:1])


def test_bert_encoder_bit_exact(ctx):
    # BASELINE config 4 topology (reduced width/depth so the oracle finishes in seconds) + one full-width layer
    from oracle import models as omodels
    from rten_amd.workloads import bert
    for cfg, B, S in ((bert.BertConfig(hidden=128, heads=4, layers=2, ffn=256, vocab=1000, max_pos=64), 3, 40),
                      (bert.BertConfig(hidden=768, heads=12, layers=1, ffn=3072, vocab=2000, max_pos=128), 2, 128)):
        w = bert.make_weights(cfg)
        rng = np.random.default_rng(5)
        ids = rng.integers(0, cfg.vocab, (B, S))
        tts = rng.integers(0, 2, (B, S))
        am = np.ones((B, S), np.float32)
        am[0, S - 7:] = 0  # padded tail on one sequence
        net = bert.Bert(ctx, cfg, B, S, w)
        net.set_inputs(ids, am, tts)
        got = net.forward().numpy()
        want = omodels.bert_forward(cfg, w, ids, am, tts)
        bits_equal(got, want)
        net.capture()
        net.run()
        bits_equal(net.x.numpy(), want)
        net.graph = None
        net.autotune(reps=1)  # per-shape tile variants leave the bits unchanged
        bits_equal(net.forward().numpy(), want)


def test_binary_ops_numpy_broadcasting(ctx):
    # binary_elementwise.rs:58-170: every numpy broadcast form, through the operator layer
    rng = ref.XorShiftRng(5)
    x = rng.f32(2 * 3 * 4 * 5).reshape(2, 3, 4, 5) - 0.5
    others = [rng.f32(5) + 1, (rng.f32(3) + 1).reshape(3, 1, 1), (rng.f32(2 * 5) + 1).reshape(2, 1, 1, 5), np.float32(1.5).reshape(()), (rng.f32(4) + 1).reshape(4, 1),
              rng.f32(2 * 3 * 4 * 5).reshape(2, 3, 4, 5) + 1]
    for o in others:
        o = np.asarray(o, np.float32)
        for op, fn in ((ops.Add, np.add), (ops.Mul, np.multiply), (ops.Sub, np.subtract), (ops.Div, np.divide)):
            bits_equal(op().run(ctx, [dev(ctx, x), dev(ctx, o)])[0].numpy(), fn(x, o))
            bits_equal(op().run(ctx, [dev(ctx, o), dev(ctx, x)])[0].numpy(), fn(o, x))
    a, b = rng.f32(3).reshape(3, 1), rng.f32(4).reshape(1, 4)
    bits_equal(ops.Sub().run(ctx, [dev(ctx, a), dev(ctx, b)])[0].numpy(), a - b)  # neither input has the output shape
    with pytest.raises(ops.OpError, match="Cannot broadcast"):
        ops.Add().run(ctx, [dev(ctx, rng.f32(6).reshape(2, 3)), dev(ctx, rng.f32(8).reshape(2, 4))])
    for perm in (None, (0, 2, 1, 3), (3, 0, 1, 2), (-1, 1, 2, 0)):
        bits_equal(ops.Transpose(perm).run(ctx, [dev(ctx, x)])[0].numpy(), np.transpose(x, perm))


def test_layout_and_broadcast_ops_bit_exact(ctx):
    """Transpose / general broadcasting / Sub / Div (src/ops/layout.rs, binary_elementwise.rs): pure data movement and
    single IEEE operations, so numpy is the exact reference."""
    rng = ref.XorShiftRng(21)
    i64 = lambda v: (C.c_int64 * len(v))(*v)
    x = (rng.f32(3 * 4 * 5 * 7) - 0.5).reshape(3, 4, 5, 7)
    xd = dev(ctx, x)
    for perm in ((0, 2, 1, 3), (0, 2, 3, 1), (3, 2, 1, 0), (0, 1, 2, 3), (1, 0, 3, 2)):
        want = np.ascontiguousarray(x.transpose(perm))
        out = DeviceTensor(ctx, want.shape, np.float32)
        ctx.call("rten_hip_transpose_b32", 4, i64(x.shape), (C.c_int32 * 4)(*perm), xd.vp, out.vp)
        bits_equal(out.numpy(), want)
    with pytest.raises(L.HipError, match="Permutation is invalid"):
        ctx.call("rten_hip_transpose_b32", 4, i64(x.shape), (C.c_int32 * 4)(0, 1, 1, 3), xd.vp, xd.vp)
    # flat forms: equal shapes and trailing-dims broadcast
    y = (rng.f32(x.size) - 0.5).reshape(x.shape) + 2.0
    yd, out = dev(ctx, y), DeviceTensor(ctx, x.shape, np.float32)
    for name, fn in (("rten_hip_sub_f32", np.subtract), ("rten_hip_div_f32", np.divide)):
        ctx.call(name, x.size, xd.vp, yd.vp, y.size, out.vp)
        bits_equal(out.numpy(), fn(x, y))
        row = dev(ctx, y[0, 0, 0])
        ctx.call(name, x.size, xd.vp, row.vp, 7, out.vp)
        bits_equal(out.numpy(), fn(x, y[0, 0, 0]))
    # general broadcasting: scalar on the left, [B,1,1,T] masks, middle-axis broadcast
    cases = [(np.float32(1.0).reshape(()), x), (x, (rng.f32(3 * 7) - 0.5).reshape(3, 1, 1, 7)), ((rng.f32(4 * 5) + 1).reshape(4, 5, 1), x),
             ((rng.f32(5) + 1).reshape(5, 1), (rng.f32(3 * 7)).reshape(3, 1, 1, 7))]
    for a, b in cases:
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        oshape = np.broadcast_shapes(a.shape, b.shape)
        nd = len(oshape)

        def strides(t):
            s, acc, shp = [0] * nd, 1, (1,) * (nd - t.ndim) + t.shape
            for i in range(nd - 1, -1, -1):
                s[i] = 0 if shp[i] == 1 else acc
                acc *= shp[i]
            return s
        ad, bd = dev(ctx, a.reshape(-1) if a.ndim else a.reshape(1)), dev(ctx, b.reshape(-1) if b.ndim else b.reshape(1))
        out = DeviceTensor(ctx, oshape, np.float32)
        for op, fn in enumerate((np.add, np.multiply, np.subtract, np.divide)):
            ctx.call("rten_hip_binary_broadcast_f32", op, nd, i64(oshape), i64(strides(a)), i64(strides(b)), ad.vp, bd.vp, out.vp)
            with np.errstate(all="ignore"):
                bits_equal(out.numpy(), fn(a, b).astype(np.float32))
