"""Randomised geometry sweeps (fixed seeds) of the hot-path kernels against the CPU oracle: the directed cases in
test_gpu_parity.py follow the reference's own test tables; these cover the space between them -- ragged sizes, asymmetric
padding, stride / dilation / group combinations, K straddling the depth-block size, every launch plan the tuner may pick."""
import numpy as np
import pytest

from oracle import ref
from rten_amd import lib as L
from rten_amd import ops
from tests.test_gpu_parity import bits_equal, dev, gpu_conv, gpu_gemm

pytestmark = pytest.mark.gpu


def _conv_geometry(rng, max_c=48, max_o=72, max_hw=18):
    while True:
        groups = int(rng.choice([1, 1, 1, 2, 3, 4]))
        cg, og = int(rng.integers(1, max(2, max_c // groups))), int(rng.integers(1, max(2, max_o // groups)))
        kh, kw = int(rng.choice([1, 1, 2, 3, 3, 5])), int(rng.choice([1, 1, 2, 3, 3, 5]))
        strides = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        dil = (int(rng.choice([1, 1, 1, 2])), int(rng.choice([1, 1, 1, 2])))
        pads = tuple(int(v) for v in rng.integers(0, 3, 4))
        H, W = int(rng.integers(1, max_hw)), int(rng.integers(1, max_hw))
        if H + pads[0] + pads[2] >= dil[0] * (kh - 1) + 1 and W + pads[1] + pads[3] >= dil[1] * (kw - 1) + 1:
            return int(rng.integers(1, 4)), cg * groups, H, W, og * groups, kh, kw, pads, strides, dil, groups


@pytest.mark.parametrize("seed", range(6))
def test_conv_f32_random_geometry(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(12):
        N, Cc, H, W, O, kh, kw, pads, strides, dil, groups = _conv_geometry(rng)
        x = rng.random((N, Cc, H, W), dtype=np.float32) - 0.5
        w = rng.random((O, Cc // groups, kh, kw), dtype=np.float32) - 0.5
        bias = (rng.random(O, dtype=np.float32) - 0.5) if rng.random() < 0.7 else None
        relu = bool(rng.random() < 0.5)
        want = ref.conv2d_f32(x, w, bias, pads=pads, strides=strides, dilations=dil, groups=groups)
        res = (rng.random(want.shape, dtype=np.float32) - 0.5) if rng.random() < 0.4 else None
        want = ref.conv2d_f32(x, w, bias, pads=pads, strides=strides, dilations=dil, groups=groups, residual=res, relu=relu)
        nvar = ctx.lib.rten_hip_num_gemm_variants()
        variant = None if rng.random() < 0.3 else int(rng.integers(0, nvar))
        mode, g, order = int(rng.integers(0, 7)), int(rng.integers(1, 5)), int(rng.integers(0, 4))  # mode 4 = thin-tile tail plan, 5 / 6 = persistent
        ctx.call("rten_hip_set_gemm_split", mode, g)
        ctx.call("rten_hip_set_gemm_order", order)
        try:
            got = gpu_conv(ctx, x, w, bias, pads, strides, dil, groups, residual=res, relu=relu, prepack=bool(rng.random() < 0.5), variant=variant)
        finally:
            ctx.call("rten_hip_set_gemm_split", 3, 1)
            ctx.call("rten_hip_set_gemm_order", 0)
        try:
            bits_equal(got, want)
        except AssertionError as e:
            raise AssertionError(f"conv N={N} C={Cc} H={H} W={W} O={O} k={kh}x{kw} pads={pads} strides={strides} dil={dil} groups={groups} "
                                 f"variant={variant} split=({mode},{g}) order={order}: {e}") from None


@pytest.mark.parametrize("seed", range(4))
def test_conv_f32_random_deep_k(ctx, seed):
    """K = C * kh * kw up to a few depth blocks of 256: the exact split-K plans and the fixup replay."""
    rng = np.random.default_rng(2000 + seed)
    for _ in range(6):
        C_, O = int(rng.integers(60, 200)), int(rng.integers(1, 140))
        k = int(rng.choice([1, 3]))
        H, W = int(rng.integers(3, 15)), int(rng.integers(3, 15))
        N = int(rng.integers(1, 5))
        p = k // 2
        x = rng.random((N, C_, H, W), dtype=np.float32) - 0.5
        w = rng.random((O, C_, k, k), dtype=np.float32) - 0.5
        bias = rng.random(O, dtype=np.float32) - 0.5
        want = ref.conv2d_f32(x, w, bias, pads=(p, p, p, p), relu=True)
        nblk = (C_ * k * k + 255) // 256
        for mode in (0, 1, 2, 3, 4, 5, 6):
            g = int(rng.integers(1, nblk + 2))
            ctx.call("rten_hip_set_gemm_split", mode, g)
            try:
                got = gpu_conv(ctx, x, w, bias, (p, p, p, p), relu=True, variant=int(rng.integers(0, ctx.lib.rten_hip_num_gemm_variants())))
            finally:
                ctx.call("rten_hip_set_gemm_split", 3, 1)
            bits_equal(got, want)


@pytest.mark.parametrize("seed", range(4))
def test_gemm_f32_random(ctx, seed):
    rng = np.random.default_rng(3000 + seed)
    for _ in range(16):
        M, K, N = int(rng.integers(2, 200)), int(rng.integers(1, 700)), int(rng.integers(1, 200))
        a = rng.random((M, K), dtype=np.float32) - 0.5
        b = rng.random((K, N), dtype=np.float32) - 0.5
        if rng.random() < 0.4:
            a = np.ascontiguousarray(a.T).T
        if rng.random() < 0.4:
            b = np.ascontiguousarray(b.T).T
        alpha = float(rng.choice([1.0, 0.5, -1.25]))
        beta = float(rng.choice([0.0, 1.0, 0.75]))
        c = (rng.random((M, N), dtype=np.float32) - 0.5) if beta != 0.0 else None
        kind = int(rng.choice([0, 1, 2]))
        bias = None if kind == 0 else rng.random(M if kind == L.BIAS_PER_ROW else N, dtype=np.float32)
        want = ref.gemm_f32(a, b, c=c, alpha=alpha, beta=beta, bias=bias, bias_kind=kind)
        ctx.call("rten_hip_set_gemm_split", int(rng.integers(0, 4)), int(rng.integers(1, 4)))
        try:
            variant = int(rng.integers(0, 16))
            got = gpu_gemm(ctx, a, b, c=c, alpha=alpha, beta=beta, bias=bias, bias_kind=kind, variant=variant)
        finally:
            ctx.call("rten_hip_set_gemm_split", 3, 1)
        try:
            bits_equal(got, want)
        except AssertionError as e:
            raise AssertionError(f"gemm M={M} K={K} N={N} aT={not a.flags.c_contiguous} bT={not b.flags.c_contiguous} alpha={alpha} beta={beta} bias_kind={kind} variant={variant}: {e}") from None


@pytest.mark.parametrize("seed", range(4))
def test_conv_integer_random_geometry(ctx, seed):
    rng = np.random.default_rng(4000 + seed)
    for path in (0, 1):
        ctx.call("rten_hip_set_int8_path", path)
        try:
            for _ in range(8):
                N, Cc, H, W, O, kh, kw, pads, strides, dil, groups = _conv_geometry(rng, max_c=70, max_o=140)
                xdt, wdt = (np.uint8, np.int8) if rng.random() < 0.6 else (rng.choice([np.uint8, np.int8]), rng.choice([np.uint8, np.int8]))
                x = rng.integers(0, 256, (N, Cc, H, W)).astype(np.uint8).view(xdt)
                w = rng.integers(-64, 65, (O, Cc // groups, kh, kw)).astype(np.int8).view(np.int8).astype(wdt if wdt == np.int8 else np.int16).astype(wdt) if wdt == np.int8 \
                    else rng.integers(0, 128, (O, Cc // groups, kh, kw)).astype(np.uint8)
                x_zp = np.array(rng.integers(0, 256), np.uint8).view(xdt).reshape(())
                w_zp = None if rng.random() < 0.5 else (np.array(3, wdt) if rng.random() < 0.5 else rng.integers(0, 5, O).astype(wdt))
                pm = int(rng.choice([L.PAD_ZERO_POINT, L.PAD_RAW0_I8, L.PAD_RAW0_U8]))
                op = ops.ConvInteger(groups=groups, dilations=dil, padding=list(pads), strides=strides, pad_mode=pm)
                got = op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, x_zp), dev(ctx, w_zp) if w_zp is not None else None])[0].numpy()
                want = ref.conv2d_int8(x, w, x_zp=int(x_zp), w_zp=w_zp, pads=pads, strides=strides, dilations=dil, groups=groups, pad_mode=pm)
                try:
                    bits_equal(got, want)
                except AssertionError as e:
                    raise AssertionError(f"path={path} N={N} C={Cc} H={H} W={W} O={O} k={kh}x{kw} pads={pads} strides={strides} dil={dil} groups={groups} "
                                         f"x={np.dtype(xdt)} w={np.dtype(wdt)} w_zp={'none' if w_zp is None else w_zp.shape} pad_mode={pm}: {e}") from None
        finally:
            ctx.call("rten_hip_set_int8_path", 0)


@pytest.mark.parametrize("seed", range(3))
def test_matmul_integer_random(ctx, seed):
    rng = np.random.default_rng(5000 + seed)
    for path in (0, 1):
        ctx.call("rten_hip_set_int8_path", path)
        try:
            for _ in range(10):
                M, K, N = int(rng.integers(1, 150)), int(rng.integers(1, 400)), int(rng.integers(1, 150))
                adt, bdt = rng.choice([np.uint8, np.int8]), rng.choice([np.uint8, np.int8])
                a = rng.integers(0, 256, (M, K)).astype(np.uint8).view(adt)
                b = rng.integers(0, 256, (K, N)).astype(np.uint8).view(bdt)
                a_zp = None if rng.random() < 0.3 else (rng.integers(0, 256, () if rng.random() < 0.5 else (M,)).astype(np.uint8).view(adt))
                b_zp = None if rng.random() < 0.3 else (rng.integers(0, 256, () if rng.random() < 0.5 else (N,)).astype(np.uint8).view(bdt))
                want = ref.gemm_int8(a, b, a_zp, b_zp)
                got = ops.MatMulInteger().run(ctx, [dev(ctx, a), dev(ctx, b), dev(ctx, a_zp) if a_zp is not None else None,
                                                    dev(ctx, b_zp) if b_zp is not None else None])[0].numpy()
                bits_equal(got, want)
        finally:
            ctx.call("rten_hip_set_int8_path", 0)


@pytest.mark.parametrize("seed", range(5))
def test_depthwise_conv_f32_random_geometry(ctx, seed):
    """groups == C == O: the reference's depthwise kernel (conv/depthwise.rs) -- bias-first accumulator, separate multiply and add,
    padded taps skipped -- replayed per output element; window sizes with and without the unrolled instantiations."""
    rng = np.random.default_rng(6000 + seed)
    for _ in range(14):
        C_ = int(rng.integers(1, 40))
        kh, kw = [(3, 3), (5, 5), (2, 2), (1, 3), (3, 1), (1, 1), (7, 7)][int(rng.integers(0, 7))]
        strides = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        dil = (int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2])))
        pads = tuple(int(v) for v in rng.integers(0, 4, 4))
        H, W = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        if rng.random() < 0.4:  # steer some cases onto the four-outputs-per-thread 3x3 kernel (out_w % 4 == 0, dilation 1, stride_w 1 or 2)
            kh, kw, dil = 3, 3, (1, 1)
            strides = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
            ow = 4 * int(rng.integers(1, 7))
            W = (ow - 1) * strides[1] + 3 - pads[1] - pads[3]
            if W < 1:
                continue
        if H + pads[0] + pads[2] < dil[0] * (kh - 1) + 1 or W + pads[1] + pads[3] < dil[1] * (kw - 1) + 1:
            continue
        if (kh, kw) == (1, 1) and strides == (1, 1) and pads == (0, 0, 0, 0) and C_ == 1:
            continue  # that one is the pointwise GEMM (groups == 1)
        N = int(rng.integers(1, 4))
        x = rng.random((N, C_, H, W), dtype=np.float32) - 0.5
        w = rng.random((C_, 1, kh, kw), dtype=np.float32) - 0.5
        bias = (rng.random(C_, dtype=np.float32) - 0.5) if rng.random() < 0.7 else None
        relu = bool(rng.random() < 0.5)
        want = ref.conv2d_f32(x, w, bias, pads=pads, strides=strides, dilations=dil, groups=C_)
        res = (rng.random(want.shape, dtype=np.float32) - 0.5) if rng.random() < 0.4 else None
        want = ref.conv2d_f32(x, w, bias, pads=pads, strides=strides, dilations=dil, groups=C_, residual=res, relu=relu)
        got = gpu_conv(ctx, x, w, bias, pads, strides, dil, C_, residual=res, relu=relu, prepack=bool(rng.random() < 0.5))
        try:
            bits_equal(got, want)
        except AssertionError as e:
            raise AssertionError(f"depthwise N={N} C={C_} H={H} W={W} k={kh}x{kw} pads={pads} strides={strides} dil={dil}: {e}") from None


def test_depthwise_conv_integer_ignores_the_padding_quirk(ctx):
    """Integer depthwise geometries skip padded taps in the reference (conv/depthwise.rs:148-190): the platform's im2col pad
    quirk (pad_mode) must not show, on either int8 path."""
    rng = np.random.default_rng(7000)
    for path in (0, 1):
        ctx.call("rten_hip_set_int8_path", path)
        try:
            for (C_, H, W, k, pads, strides) in ((6, 9, 8, 3, (1, 1, 1, 1), (1, 1)), (16, 7, 7, 3, (1, 0, 2, 1), (2, 2)), (3, 12, 5, 5, (2, 2, 2, 2), (1, 1))):
                x = rng.integers(0, 256, (2, C_, H, W)).astype(np.uint8)
                w = rng.integers(-64, 65, (C_, 1, k, k)).astype(np.int8)
                x_zp = np.array(rng.integers(1, 255), np.uint8)
                want = ref.conv2d_int8(x, w, x_zp=int(x_zp), pads=pads, strides=strides, groups=C_, pad_mode=L.PAD_ZERO_POINT)
                for pm in (L.PAD_ZERO_POINT, L.PAD_RAW0_I8, L.PAD_RAW0_U8):
                    assert np.array_equal(ref.conv2d_int8(x, w, x_zp=int(x_zp), pads=pads, strides=strides, groups=C_, pad_mode=pm), want)
                    op = ops.ConvInteger(groups=C_, padding=list(pads), strides=strides, pad_mode=pm)
                    got = op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, x_zp), None])[0].numpy()
                    bits_equal(got, want)
        finally:
            ctx.call("rten_hip_set_int8_path", 0)


@pytest.mark.parametrize("seed", range(2))
def test_pooling_random_geometry(ctx, seed):
    rng = np.random.default_rng(8000 + seed)
    done = 0
    while done < 25:
        k = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
        s = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        p = tuple(int(v) for v in rng.integers(0, 3, 4))
        if p[0] >= k[0] or p[2] >= k[0] or p[1] >= k[1] or p[3] >= k[1]:
            continue  # a window must not lie entirely in the padding
        N, C_, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 7)), int(rng.integers(1, 26)), int(rng.integers(1, 26))
        if H + p[0] + p[2] < k[0] or W + p[1] + p[3] < k[1]:
            continue
        ceil = bool(rng.random() < 0.3)
        x = rng.random((N, C_, H, W), dtype=np.float32) - 0.5
        try:
            want_max = ref.max_pool(x, k, s, p, ceil)
        except Exception:
            continue  # geometry the reference rejects
        bits_equal(ops.MaxPool(k, padding=list(p), strides=s, ceil_mode=ceil).run(ctx, [dev(ctx, x)])[0].numpy(), want_max)
        for cip in (False, True):
            bits_equal(ops.AveragePool(k, padding=list(p), strides=s, ceil_mode=ceil, count_include_pad=cip).run(ctx, [dev(ctx, x)])[0].numpy(),
                       ref.average_pool(x, k, s, p, cip, ceil))
        done += 1
    for (nc, inner) in ((1, 1), (7, 49), (130, 64), (5, 65), (3, 1000), (260, 13)):
        xg = rng.random((nc, 1, inner, 1), dtype=np.float32).reshape(1, nc, inner, 1) - 0.5
        bits_equal(ops.GlobalAveragePool().run(ctx, [dev(ctx, xg)])[0].numpy(), ref.global_average_pool(xg, lanes=16))
