"""Round-3 parity tests (all through the C ABI, on the GPU).

  * rten_hip_conv2d_int8_qout: ConvIntegerToFloat with the DynamicQuantizeLinear of its single consumer in the epilogue (one launch,
    grid-wide min / max all-gather) against the two operators run separately -- staged codes byte for byte (border
    pieces included), scale, zero point, scale product, optional f32 output -- and int8 ResNet-50 at batch 32 with those launches on.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import ref
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor

pytestmark = pytest.mark.gpu


def dev(ctx, a):
    return DeviceTensor.from_numpy(ctx, a)


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        if not same.all():
            idx = tuple(np.argwhere(~same)[0])
            raise AssertionError(f"{(~same).sum()} of {a.size} elements differ; first at {idx}: {a[idx]!r} vs {b[idx]!r}")
    else:
        assert np.array_equal(a, b), f"{(a != b).sum()} of {a.size} elements differ"


def _conv_desc(n, c, h, w, o, k, stride, pad, pad_mode=L.PAD_RAW0_I8, packed=1, staged=1, scale_len=1):
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    return L.Conv2dInt8Desc(L.Conv2dDesc(n, c, h, w, o, k, k, (C.c_int32 * 4)(pad, pad, pad, pad), stride, stride, 1, 1, 1, oh, ow), 0, 1, 0, pad_mode,
                            packed, staged, scale_len), oh, ow


def staged_image(codes_u8, zero_point, pads, pad_mode):
    """numpy restatement of the staged int8 activation layout the fast kernels read (csrc/int8_fast.hip `ConvGeom`):
    [N][ceil(C / 16)][H + pad_top + pad_bottom][W + pad_left + pad_right][16 B] of SIGNED bytes (u8 code ^ 0x80); the border holds the padded-tap
    value of the pad mode -- zero point (in the signed domain), raw 0 after the u8 -> i8 shift, or raw u8 0; channels past C are 0
    everywhere (border included).
    With it a test can compare staged bytes against oracle.ref.dynamic_quantize_linear directly instead of against another HIP path."""
    n, c, h, w = codes_u8.shape
    pt, pl, pb, pr = pads
    cb = (c + 15) // 16
    border = {L.PAD_ZERO_POINT: (int(zero_point) - 128) & 0xFF, L.PAD_RAW0_I8: 0, L.PAD_RAW0_U8: 0x80}[pad_mode]
    chan_fill = np.where(np.arange(cb * 16) < c, border, 0).astype(np.uint8).reshape(cb, 1, 1, 16)  # border bytes: pad value on real channels, 0 on padding channels
    img = np.broadcast_to(chan_fill, (n, cb, h + pt + pb, w + pl + pr, 16)).copy()
    body = np.zeros((n, cb * 16, h, w), np.uint8)
    body[:, :c] = codes_u8 ^ 0x80
    img[:, :, pt:pt + h, pl:pl + w, :] = body.reshape(n, cb, 16, h, w).transpose(0, 1, 3, 4, 2)
    return img.reshape(-1)


QOUT_CASES = [
    # n, c, h, w | producer o, k, stride, pad | consumer o2, k2, stride2, pad2 | consumer pad mode, per-channel scale, residual + f32 output
    (4, 64, 56, 56, 64, 1, 1, 0, 64, 3, 1, 1, L.PAD_RAW0_I8, False, False),      # c1 -> c2 of stage 0
    (3, 64, 28, 28, 64, 3, 1, 1, 256, 1, 1, 0, L.PAD_RAW0_I8, False, False),     # c2 -> c3
    (32, 256, 14, 14, 128, 1, 1, 0, 128, 3, 2, 1, L.PAD_ZERO_POINT, False, False),  # stride-2 consumer, dynamic border value
    (8, 128, 28, 28, 128, 3, 2, 1, 512, 1, 1, 0, L.PAD_RAW0_I8, False, True),    # stride-2 producer, residual, f32 output kept as well
    (32, 2048, 7, 7, 512, 1, 1, 0, 512, 3, 1, 1, L.PAD_RAW0_U8, False, False),   # under-filled launch: four k-groups per workgroup
    (1, 48, 9, 5, 80, 3, 1, 1, 32, 3, 1, 1, L.PAD_ZERO_POINT, True, True),       # ragged everything, per-channel scale
    (32, 64, 56, 56, 64, 3, 1, 1, 256, 1, 1, 0, L.PAD_RAW0_I8, False, False),    # BASELINE size: 784 tiles of 64 x 128, all resident
    (2, 16, 6, 6, 16, 1, 1, 0, 16, 1, 1, 0, L.PAD_RAW0_I8, False, False),        # one workgroup
]


@pytest.mark.parametrize("case", QOUT_CASES, ids=[f"case{i}" for i in range(len(QOUT_CASES))])
def test_conv2d_int8_qout_matches_the_separate_operators(ctx, case):
    n, c, h, w, o, k, stride, pad, o2, k2, stride2, pad2, pad_mode2, per_ch, with_res = case
    rng = ref.XorShiftRng(4321 + n * 7 + c)
    lib = ctx.lib
    sb, gb = lib.rten_hip_minmax_stats_bytes(), lib.rten_hip_grid_sync_bytes()
    d, oh, ow = _conv_desc(n, c, h, w, o, k, stride, pad, scale_len=o if per_ch else 1)
    d2, _, _ = _conv_desc(n, o, oh, ow, o2, k2, stride2, pad2, pad_mode=pad_mode2)
    # the producer's own quantized input (staged) and prepacked weights
    xf = (rng.f32(n * c * h * w).reshape(n, c, h, w) - 0.3).astype(np.float32)
    staged_in = DeviceTensor(ctx, (lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d)),), np.uint8)
    xs, xz, sc = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8), DeviceTensor(ctx, (o if per_ch else 1,), np.float32)
    ws = dev(ctx, (rng.f32(o if per_ch else 1) * 0.01 + 0.002).astype(np.float32))
    xfd = dev(ctx, xf)  # (named: a temporary would be freed before the launch reads it)
    ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), xfd.vp, staged_in.vp, xs.vp, xz.vp, None, None)
    ctx.call("rten_hip_mul_f32", o if per_ch else 1, ws.vp, xs.vp, 1, sc.vp)
    wq = rng.i8(o * c * k * k, reduced=True).reshape(o, c, k, k)
    packed = DeviceTensor(ctx, (lib.rten_hip_conv2d_int8_packed_bytes(C.byref(d)),), np.uint8)
    wqd = dev(ctx, wq)
    ctx.call("rten_hip_conv2d_int8_prepack", C.byref(d), wqd.vp, packed.vp)
    ctx.sync()
    bias = dev(ctx, rng.f32(o) - 0.5)
    res = dev(ctx, rng.f32(n * o * oh * ow).reshape(n, o, oh, ow) - 0.5) if with_res else None
    flags = L.CONV_RELU | (L.CONV_RESIDUAL if with_res else 0)
    ws2 = dev(ctx, np.array([0.0173], np.float32))
    nb = lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d2))
    assert nb > 0

    # (a) the two operators: ConvIntegerToFloat (+ producer statistics), then DynamicQuantizeLinear into the consumer's staged layout
    y_a = DeviceTensor(ctx, (n, o, oh, ow), np.float32)
    st_a = DeviceTensor(ctx, (sb,), np.uint8)
    ctx.call("rten_hip_minmax_stats_reset", st_a.vp, 1)
    ctx.call("rten_hip_conv2d_int8_stats", C.byref(d), staged_in.vp, packed.vp, xz.vp, None, sc.vp, bias.vp, res.vp if res else None, flags, y_a.vp, st_a.vp)
    staged_a = dev(ctx, np.full(nb, 0xEE, np.uint8))
    xs_a, xz_a, pr_a = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8), DeviceTensor(ctx, (1,), np.float32)
    ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d2), y_a.vp, st_a.vp, staged_a.vp, xs_a.vp, xz_a.vp, ws2.vp, pr_a.vp)
    ctx.sync()

    # (a') ... and the CPU oracle on its own, so that this test does not rest on another HIP path: the producer's input codes, the convolution in
    # the reference's operator order (ConvInteger -> Cast -> Mul -> Add(bias) -> Add(residual) -> Relu), DynamicQuantizeLinear of the result, and
    # its codes laid out as the consumer's staged image
    q_in, s_in, z_in = ref.dynamic_quantize_linear(xf)
    acc = ref.conv2d_int8(q_in, wq, x_zp=int(z_in), pads=(pad,) * 4, strides=(stride,) * 2, pad_mode=L.PAD_RAW0_I8)
    scale_vec = (ws.numpy() * np.float32(s_in)).astype(np.float32)
    y_ref = acc.astype(np.float32) * (scale_vec[None, :, None, None] if per_ch else scale_vec[0])
    y_ref = y_ref + bias.numpy()[None, :, None, None]
    if with_res:
        y_ref = y_ref + res.numpy()
    y_ref = ref.relu(y_ref.astype(np.float32))
    bits_equal(y_a.numpy(), y_ref)
    q2, s2, z2 = ref.dynamic_quantize_linear(y_ref)
    assert xs_a.numpy()[0] == s2 and xz_a.numpy()[0] == z2 and pr_a.numpy()[0] == np.float32(np.float32(s2) * ws2.numpy()[0])
    want_img = staged_image(q2, z2, (pad2,) * 4, pad_mode2)
    assert np.array_equal(staged_a.numpy()[: want_img.size], want_img), "two-operator staged image differs from the oracle's codes re-laid by numpy"

    # (b) one launch; twice in a row (the barrier block must come back zeroed), with and without the f32 output
    sync = DeviceTensor(ctx, (gb,), np.uint8)
    ctx.call("rten_hip_grid_sync_reset", sync.vp, 1)
    ctx.sync()
    sync0 = sync.numpy()
    for rep, want_y in enumerate((with_res, False, True)):
        st_b = DeviceTensor(ctx, (sb,), np.uint8)
        ctx.call("rten_hip_minmax_stats_reset", st_b.vp, 1)
        staged_b = dev(ctx, np.full(nb, 0xEE, np.uint8))
        y_b = dev(ctx, np.zeros((n, o, oh, ow), np.float32))
        xs_b, xz_b, pr_b = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8), DeviceTensor(ctx, (1,), np.float32)
        ctx.call("rten_hip_conv2d_int8_qout", C.byref(d), staged_in.vp, packed.vp, xz.vp, None, sc.vp, bias.vp, res.vp if res else None, flags,
                 y_b.vp if want_y else None, st_b.vp, sync.vp, C.byref(d2), staged_b.vp, xs_b.vp, xz_b.vp, ws2.vp, pr_b.vp)
        ctx.sync()
        to = C.c_int32(-1)
        ctx.call("rten_hip_grid_sync_timeouts", sync.vp, 1, C.byref(to))
        assert to.value == 0, "the launch gave up waiting for its grid"
        assert np.array_equal(sync.numpy(), sync0), "exchange block not left as it was found"
        assert np.array_equal(xs_a.numpy().view(np.uint32), xs_b.numpy().view(np.uint32)) and np.array_equal(xz_a.numpy(), xz_b.numpy())
        assert np.array_equal(pr_a.numpy().view(np.uint32), pr_b.numpy().view(np.uint32))
        a, b = staged_a.numpy(), staged_b.numpy()
        assert np.array_equal(a, b), f"rep {rep}: {(a != b).sum()} of {a.size} staged bytes differ (first at {np.argwhere(a != b)[0]})"
        assert np.array_equal(b[: want_img.size], want_img) and xs_b.numpy()[0] == s2 and xz_b.numpy()[0] == z2  # the one-launch form against the ORACLE
        if want_y:
            bits_equal(y_b.numpy(), y_a.numpy())
        else:
            assert not y_b.numpy().any()

    # (c) round 6: the RECOMPUTE form of the same entry point (sync == NULL): a statistics-only pass, then the convolution again writing the codes.  No
    # exchange block, nothing to fit; covered for the form without a residual (UNSUPPORTED otherwise -> the two operators)
    for rep, want_y in enumerate((False, True, False)):
        st_c = DeviceTensor(ctx, (sb,), np.uint8)
        ctx.call("rten_hip_minmax_stats_reset", st_c.vp, 1)
        staged_c = dev(ctx, np.full(nb, 0xEE, np.uint8))
        y_c = dev(ctx, np.zeros((n, o, oh, ow), np.float32))
        xs_c, xz_c, pr_c = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8), DeviceTensor(ctx, (1,), np.float32)
        rc = lib.rten_hip_conv2d_int8_qout(ctx.h, C.byref(d), staged_in.vp, packed.vp, xz.vp, None, sc.vp, bias.vp, res.vp if res else None, flags,
                                           y_c.vp if want_y else None, st_c.vp, None, C.byref(d2), staged_c.vp, xs_c.vp, xz_c.vp, ws2.vp, pr_c.vp)
        ctx.sync()
        if with_res:
            assert rc == L.ERR_UNSUPPORTED
            break
        assert rc == 0, ctx.last_error() if hasattr(ctx, "last_error") else rc
        assert np.array_equal(xs_a.numpy().view(np.uint32), xs_c.numpy().view(np.uint32)) and np.array_equal(xz_a.numpy(), xz_c.numpy())
        assert np.array_equal(pr_a.numpy().view(np.uint32), pr_c.numpy().view(np.uint32))
        a, c_ = staged_a.numpy(), staged_c.numpy()
        assert np.array_equal(a, c_), f"recompute form, rep {rep}: {(a != c_).sum()} of {a.size} staged bytes differ (first at {np.argwhere(a != c_)[0]})"
        assert np.array_equal(st_a.numpy(), st_c.numpy())  # the statistics block as the two-operator sequence leaves it (a second reader of the f32 output)
        if want_y:
            bits_equal(y_c.numpy(), y_a.numpy())
        else:
            assert not y_c.numpy().any()


def test_conv2d_int8_qout_refuses_a_grid_that_cannot_be_resident(ctx):
    """A launch with more workgroups than the device holds at once must be refused (UNSUPPORTED), never attempted."""
    lib = ctx.lib
    n, c, h, w, o = 64, 64, 112, 112, 256  # 256 x 802816 outputs: 12544 tiles of 128 x 128
    d, oh, ow = _conv_desc(n, c, h, w, o, 1, 1, 0)
    d2, _, _ = _conv_desc(n, o, oh, ow, 64, 1, 1, 0)
    sb, gb = lib.rten_hip_minmax_stats_bytes(), lib.rten_hip_grid_sync_bytes()
    staged_in = DeviceTensor(ctx, (lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d)),), np.uint8)
    packed = DeviceTensor(ctx, (lib.rten_hip_conv2d_int8_packed_bytes(C.byref(d)),), np.uint8)
    staged_out = DeviceTensor(ctx, (lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d2)),), np.uint8)
    st, sync = DeviceTensor(ctx, (sb,), np.uint8), DeviceTensor(ctx, (gb,), np.uint8)
    ctx.call("rten_hip_grid_sync_reset", sync.vp, 1)
    one = dev(ctx, np.ones(1, np.float32))
    z = dev(ctx, np.zeros(1, np.uint8))
    rc = lib.rten_hip_conv2d_int8_qout(ctx.h, C.byref(d), staged_in.vp, packed.vp, z.vp, None, one.vp, None, None, 0, None, st.vp, sync.vp, C.byref(d2), staged_out.vp,
                                       one.vp, z.vp, None, None)
    assert rc == L.ERR_UNSUPPORTED
    ctx.sync()


def test_resnet50_int8_batch32_quantized_output_launches(ctx):
    """BASELINE configs[2] with the c1 -> c2 -> c3 edges of every bottleneck block quantized in the producing conv's epilogue: logits
    bit-identical to the oracle, eager and as a replayed hipGraph, and to the runner with the feature off."""
    from oracle import models as omodels
    from rten_amd.workloads import resnet50_int8
    from tests import baseline_oracle as bo
    w = bo.resnet_weights()
    net = resnet50_int8.ResNet50Int8(ctx, batch=32, weights=w)
    net.upload_weights()
    x = bo.resnet_input()
    net.x.upload(x)
    want = bo.resnet50_int8_logits()
    net.fused_qout = True
    net.forward()
    bits_equal(net.logits.numpy(), want)
    assert net.qout_timeouts() == 0
    # 32 c1 -> c2 -> c3 edges + 15 block outputs that feed the next block's quantizer (the last block's feeds the pooling); 12 of those are
    # also a residual and keep their f32 tensor (the last block of a stage is read only through the next stage's quantizer).  The launches
    # of stage 0 / 1 block outputs (1568 / 784 tiles of 128 x 128) and s1b0c1 do not fit the device at once and keep the two launches
    assert len(net.qout_next) == 47 and len(net.qout_keeps_f32) == 12, (len(net.qout_next), len(net.qout_keeps_f32))
    assert len(net._qout_off) <= 9, sorted(net._qout_off)
    net.capture()
    for _ in range(3):
        net.logits.upload(np.zeros_like(want))
        net.run()
        bits_equal(net.logits.numpy(), want)
    assert net.qout_timeouts() == 0
    net.graph, net.fused_qout = None, False
    net.logits.upload(np.zeros_like(want))
    net.forward()
    bits_equal(net.logits.numpy(), want)
    # the other pad mode goes through the in-kernel border fill with the dynamic zero point
    net2 = resnet50_int8.ResNet50Int8(ctx, batch=32, weights=w, pad_mode=L.PAD_ZERO_POINT)
    net2.upload_weights()
    net2.x.upload(x)
    net2.fused_qout = True
    net2.forward()
    bits_equal(net2.logits.numpy(), omodels.resnet50_int8_forward(net.specs, omodels.quantize_weights_int8(w), x, pad_mode=ref.PAD_ZERO_POINT))
    assert net2.qout_timeouts() == 0


def test_chained_resnet50_with_one_image_chains_keeps_the_blocked_classifier(ctx):
    """Batch 5 as 4 chains = sub-batches of 2, 1, 1, 1 images: the one-image chains must NOT run the classifier in the reference's
    one-row (gemv) order -- the reference's product has 5 rows -- so the logits stay the single-chain / oracle bits."""
    from oracle import models as omodels
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    x = ref.XorShiftRng(78).f32(5 * 3 * 224 * 224).reshape(5, 3, 224, 224)
    net = resnet50.ChainedResNet50(ctx, 5, w, chains=4)
    assert net.sizes == [2, 1, 1, 1]
    net.upload_weights()
    net.x.upload(x)
    net.forward()
    net.sync()
    bits_equal(net.logits.numpy(), omodels.resnet50_forward(net.specs, w, x))


@pytest.mark.parametrize("hd", [32, 64, 128])
def test_sdpa_fused_general_shapes_bit_exact(ctx, hd):
    """The one-kernel attention path beyond BERT-base's shape (missing 2): head size 32 / 64 / 128, up to 512 keys walked in
    chunks of 128, PV across the reference's depth-block boundary at 256 keys -- against the oracle's sdpa (sdpa_head,
    src/ops/attention.rs:518-562) and against the composed GEMM / softmax / GEMM path, bit for bit; ragged lengths, both mask forms, fully
    masked rows with and without the NaN flush."""
    rng = ref.XorShiftRng(97 + hd)
    scale = np.float32(1.0 / np.sqrt(hd))
    for (B, H, S, T) in ((1, 2, 40, 200), (2, 2, 130, 256), (1, 1, 128, 257), (1, 2, 33, 300), (1, 1, 64, 512), (2, 1, 200, 384), (1, 3, 17, 129), (2, 2, 70, 100)):
        q = rng.f32(B * H * S * hd).reshape(B, H, S, hd) - 0.5
        k = rng.f32(B * H * T * hd).reshape(B, H, T, hd) - 0.5
        v = rng.f32(B * H * T * hd).reshape(B, H, T, hd) - 0.5
        m1 = np.where(rng.f32(B * T).reshape(B, 1, 1, T) > 0.3, 0.0, -np.inf).astype(np.float32)
        m1[0, 0, 0, :] = -np.inf
        m2 = ((rng.f32(B * S * T).reshape(B, 1, S, T) - 0.5) * 4).astype(np.float32)
        qd, kd, vd = dev(ctx, q), dev(ctx, k), dev(ctx, v)
        for m in (None, m1, m2):
            md = dev(ctx, m) if m is not None else None
            for flush in (True, False):
                mbs, mrs = (0, 0) if m is None else ((T, 0) if m.shape[2] == 1 else (S * T, T))
                d = L.SdpaDesc(B, H, S, T, hd, hd, H * S * hd, S * hd, hd, H * T * hd, T * hd, hd, H * T * hd, T * hd, hd, H * S * hd, S * hd, hd,
                               mbs, mrs, float(scale), 1 if flush else 0)
                want = ref.sdpa(q, k, v, mask=m, scale=scale, lanes=16, flush_nan=flush)
                outs = []
                for path in (2, 1):  # 2 = the one-kernel form wherever it covers the shape (the automatic choice keeps it to <= 128 keys, where it is the faster one)
                    ctx.call("rten_hip_set_sdpa_path", path)
                    try:
                        out = DeviceTensor(ctx, (B, H, S, hd), np.float32)
                        ctx.profile_reset()
                        ctx.profile(True)
                        ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, md.vp if md is not None else None, out.vp)
                        ctx.sync()
                        ctx.profile(False)
                        kernels = {r["kernel"] for r in ctx.profile_report()}
                    finally:
                        ctx.call("rten_hip_set_sdpa_path", 0)
                    one_kernel = "sdpa_fused_kernel" if (hd == 64 and T <= 128) else "sdpa_fused_general_kernel"  # (BERT-base's own shape has its own kernel)
                    assert (one_kernel in kernels) == (path == 2), kernels  # the shape really took the path under test
                    outs.append(out.numpy())
                bits_equal(outs[0], want)
                bits_equal(outs[1], want)


PRODUCT_CASES = [
    # n, c, h, w | consumer k, stride, pad | pad mode | statistics from the producer (True) or the quantizer's own sweep (False) | products
    (32, 64, 56, 56, 1, 1, 0, L.PAD_RAW0_I8, True, 2),     # the max-pool's output, read by a stage's shortcut and first 1x1
    (4, 256, 56, 56, 1, 2, 0, L.PAD_RAW0_I8, False, 2),    # strided shortcut + 1x1 of stage 1, own sweep
    (2, 48, 9, 7, 3, 1, 1, L.PAD_ZERO_POINT, True, 4),     # ragged, padded, four readers
    (3, 2048, 7, 7, 1, 1, 0, L.PAD_RAW0_U8, False, 3),     # small-map form of the staging kernel
    (1, 16, 5, 5, 3, 1, 1, L.PAD_RAW0_I8, True, 0),        # no product at all
]


@pytest.mark.parametrize("case", PRODUCT_CASES, ids=[f"case{i}" for i in range(len(PRODUCT_CASES))])
def test_quantize_staged_products_matches_the_quantizer_plus_separate_muls(ctx, case):
    n, c, h, w, k, stride, pad, pad_mode, from_stats, count = case
    rng = ref.XorShiftRng(977 + n + c)
    lib = ctx.lib
    d, _, _ = _conv_desc(n, c, h, w, 32, k, stride, pad, pad_mode=pad_mode)
    xf = (rng.f32(n * c * h * w).reshape(n, c, h, w) * 3.0 - 0.8).astype(np.float32)
    xfd = dev(ctx, xf)
    nb = lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d))
    assert nb > 0
    st = None
    if from_stats:  # a statistics block as a producing launch leaves it: here the max-pool form over a 1x1 window (the identity)
        st = DeviceTensor(ctx, (lib.rten_hip_minmax_stats_bytes(),), np.uint8)
        ctx.call("rten_hip_minmax_stats_reset", st.vp, 1)
        pd = L.Pool2dDesc(n, c, h, w, 1, 1, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), h, w, 0)
        ycopy = DeviceTensor(ctx, (n, c, h, w), np.float32)
        ctx.call("rten_hip_max_pool2d_f32_stats", C.byref(pd), xfd.vp, ycopy.vp, st.vp)
        ctx.sync()
        bits_equal(ycopy.numpy(), xf)
    muls_np = [np.array([0.004 + 0.013 * i], np.float32) for i in range(count)]
    muls = [dev(ctx, m) for m in muls_np]
    # (a) the quantizer (own sweep: the operator as the reference runs it) and one Mul launch per reader
    staged_a = dev(ctx, np.full(nb, 0xEE, np.uint8))
    xs_a, xz_a = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8)
    ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), xfd.vp, staged_a.vp, xs_a.vp, xz_a.vp, None, None)
    prods_a = [DeviceTensor(ctx, (1,), np.float32) for _ in range(count)]
    for m, p in zip(muls, prods_a):
        ctx.call("rten_hip_mul_f32", 1, xs_a.vp, m.vp, 1, p.vp)
    # (b) one launch
    staged_b = dev(ctx, np.full(nb, 0xEE, np.uint8))
    xs_b, xz_b = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8)
    prods_b = [dev(ctx, np.array([-1.0], np.float32)) for _ in range(count)]
    mul_ptrs = (C.c_void_p * max(count, 1))(*[m.ptr for m in muls])
    out_ptrs = (C.c_void_p * max(count, 1))(*[p.ptr for p in prods_b])
    ctx.call("rten_hip_dynamic_quantize_linear_staged_products", C.byref(d), xfd.vp, st.vp if st is not None else None, staged_b.vp, xs_b.vp, xz_b.vp, count,
             mul_ptrs if count else None, out_ptrs if count else None)
    ctx.sync()
    assert np.array_equal(xs_a.numpy().view(np.uint32), xs_b.numpy().view(np.uint32)) and np.array_equal(xz_a.numpy(), xz_b.numpy())
    a, b = staged_a.numpy(), staged_b.numpy()
    assert np.array_equal(a, b), f"{(a != b).sum()} of {a.size} staged bytes differ"
    for i in range(count):
        want = (xs_a.numpy() * muls_np[i]).astype(np.float32)  # one f32 multiply
        assert np.array_equal(prods_a[i].numpy().view(np.uint32), want.view(np.uint32))
        assert np.array_equal(prods_b[i].numpy().view(np.uint32), want.view(np.uint32)), f"product {i}"
    # the oracle's DynamicQuantizeLinear on the same tensor: scale and zero point
    q_ref, s_ref, z_ref = ref.dynamic_quantize_linear(xf)
    assert np.float32(s_ref).view(np.uint32) == xs_b.numpy().view(np.uint32)[0] and int(z_ref) == int(xz_b.numpy()[0])
    # ... and its codes, re-laid by numpy into the consumer's staged image: the staged bytes stand on the oracle, not on another HIP path
    want_img = staged_image(q_ref, z_ref, (pad,) * 4, pad_mode)
    assert np.array_equal(b[: want_img.size], want_img), f"{(b[: want_img.size] != want_img).sum()} staged bytes differ from the oracle's codes"


def test_quantize_staged_products_rejects_more_than_four_and_null_entries(ctx):
    d, _, _ = _conv_desc(1, 16, 4, 4, 16, 1, 1, 0)
    x = dev(ctx, np.ones((1, 16, 4, 4), np.float32))
    staged = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d)),), np.uint8)
    xs, xz = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8)
    five = (C.c_void_p * 5)(*[xs.ptr] * 5)
    rc = ctx.lib.rten_hip_dynamic_quantize_linear_staged_products(ctx.h, C.byref(d), x.vp, None, staged.vp, xs.vp, xz.vp, 5, five, five)
    assert rc == L.ERR_UNSUPPORTED
    two = (C.c_void_p * 2)(xs.ptr, None)
    rc = ctx.lib.rten_hip_dynamic_quantize_linear_staged_products(ctx.h, C.byref(d), x.vp, None, staged.vp, xs.vp, xz.vp, 2, two, two)
    assert rc == L.ERR_INVALID_VALUE
    rc = ctx.lib.rten_hip_dynamic_quantize_linear_staged_products(ctx.h, C.byref(d), x.vp, None, staged.vp, xs.vp, xz.vp, 1, None, None)
    assert rc == L.ERR_INVALID_VALUE


@pytest.mark.parametrize("geom", [(32, 64, 112, 112, 3, 2, 1), (2, 5, 9, 7, 2, 2, 0), (1, 3, 13, 11, 3, 1, 1), (2, 4, 7, 5, 5, 3, 2)],
                         ids=["resnet_stem_pool", "k2s2", "k3s1", "k5s3"])
def test_max_pool_stats_same_values_and_the_statistics_the_quantizer_would_sweep(ctx, geom):
    n, c, h, w, k, stride, pad = geom
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    rng = ref.XorShiftRng(31 + h)
    x = (rng.f32(n * c * h * w).reshape(n, c, h, w) * 2.0 - 1.2).astype(np.float32)
    xd = dev(ctx, x)
    pd = L.Pool2dDesc(n, c, h, w, k, k, stride, stride, (C.c_int32 * 4)(pad, pad, pad, pad), oh, ow, 0)
    y_a, y_b = DeviceTensor(ctx, (n, c, oh, ow), np.float32), DeviceTensor(ctx, (n, c, oh, ow), np.float32)
    st = DeviceTensor(ctx, (ctx.lib.rten_hip_minmax_stats_bytes(),), np.uint8)
    ctx.call("rten_hip_minmax_stats_reset", st.vp, 1)
    ctx.call("rten_hip_max_pool2d_f32", C.byref(pd), xd.vp, y_a.vp)
    ctx.call("rten_hip_max_pool2d_f32_stats", C.byref(pd), xd.vp, y_b.vp, st.vp)
    ctx.sync()
    bits_equal(y_b.numpy(), y_a.numpy())
    bits_equal(y_a.numpy(), ref.max_pool(x, (k, k), (stride, stride), (pad, pad, pad, pad)))
    # the statistics, read the way their consumer reads them: DynamicQuantizeLinear into a 1x1 consumer's staged layout
    d, _, _ = _conv_desc(n, c, oh, ow, 16, 1, 1, 0)
    nb = ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d))
    outs = []
    for stats in (None, st):
        staged = dev(ctx, np.full(nb, 0xEE, np.uint8))
        xs, xz = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8)
        ctx.call("rten_hip_dynamic_quantize_linear_staged_products", C.byref(d), y_a.vp, stats.vp if stats is not None else None, staged.vp, xs.vp, xz.vp, 0, None, None)
        ctx.sync()
        outs.append((staged.numpy(), xs.numpy().view(np.uint32), xz.numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    # against the oracle alone: DynamicQuantizeLinear of the pooled tensor, codes re-laid by numpy
    q_ref, s_ref, z_ref = ref.dynamic_quantize_linear(ref.max_pool(x, (k, k), (stride, stride), (pad, pad, pad, pad)))
    want_img = staged_image(q_ref, z_ref, (0, 0, 0, 0), L.PAD_RAW0_I8)
    assert np.array_equal(outs[1][0][: want_img.size], want_img) and outs[1][1][0] == np.float32(s_ref).view(np.uint32) and int(outs[1][2][0]) == int(z_ref)


def test_max_pool_stats_needs_a_statistics_block(ctx):
    pd = L.Pool2dDesc(1, 1, 4, 4, 2, 2, 2, 2, (C.c_int32 * 4)(0, 0, 0, 0), 2, 2, 0)
    x, y = dev(ctx, np.ones((1, 1, 4, 4), np.float32)), DeviceTensor(ctx, (1, 1, 2, 2), np.float32)
    assert ctx.lib.rten_hip_max_pool2d_f32_stats(ctx.h, C.byref(pd), x.vp, y.vp, None) == L.ERR_INVALID_VALUE
