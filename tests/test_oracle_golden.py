"""Pins the CPU oracle (oracle/rten_oracle.c) against the reference's own golden vectors
(tests/golden/reference_literals.json, each entry citing the reference test it was copied from)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import ref

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_literals.json")))


def eq_1e4(a, b):  # expect_eq_1e4, src/ops/mod.rs:407-412
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=0, atol=1e-4)


def expect_equal(a, b):  # rten-tensor/src/test_util.rs:47-71
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.all(np.abs(a - b) <= 1e-8 + 1e-5 * np.abs(b)), np.abs(a - b).max()


def test_rng_streams():
    g = G["rng"]
    assert ref.XorShiftRng(g["seed"]).f32(10).tolist() == [np.float32(v) for v in g["f32"]]
    assert ref.XorShiftRng(g["seed"]).i8(10).tolist() == g["i8"]
    assert ref.XorShiftRng(g["seed"]).u8(10).tolist() == g["u8"]
    assert ref.XorShiftRng(g["seed"]).i32(10).tolist() == g["i32"]
    r = ref.XorShiftRng(1234)
    for _ in range(50):  # reduced_range_rng.rs:63-73
        assert -64 <= int(r.i8(1, reduced=True)[0]) <= 63
        assert 0 <= int(r.u8(1, reduced=True)[0]) <= 127


def test_conv_literals():
    g = G["conv"]
    k = np.array(g["kernel"], np.float32).reshape(1, 1, 3, 3)
    x = np.array(g["input"], np.float32).reshape(1, 1, 3, 3)
    eq_1e4(ref.conv2d_f32(x, k, pads=(1, 1, 1, 1)).ravel(), g["expected_same_padding"])
    eq_1e4(ref.conv2d_f32(x, k).ravel(), g["expected_no_padding"])
    eq_1e4(ref.conv2d_f32(x, k, bias=np.array([1.0], np.float32)).ravel(), g["expected_with_bias_1"])
    # Padding::Same == [1,1,1,1] here (conv.rs:845-883)
    oh, ow, pads = ref.calc_output_size_and_padding((3, 3), (3, 3), (1, 1), "same")
    assert (oh, ow, pads) == (3, 3, [1, 1, 1, 1])


def test_conv_depthwise_literal_and_operation_order():
    """Depthwise convolutions take the reference's own path (conv.rs:269-284 -> conv/depthwise.rs): the literal of
    test_conv_depthwise (conv.rs:990-1031), and the operation sequence itself -- accumulator starts at the bias, one rounded
    multiply and one add per in-bounds tap in (k_y, k_x) order, padded taps skipped -- against a scalar numpy loop."""
    g = G["conv_depthwise"]
    x = np.array(g["input"], np.float32).reshape(g["input_shape"])
    k = np.array(g["kernel"], np.float32).reshape(g["kernel_shape"])
    b = np.array(g["bias"], np.float32)
    want = np.array(g["expected_without_bias"], np.float32) + b
    eq_1e4(ref.conv2d_f32(x, k, b, groups=3).ravel(), want)
    rng = ref.XorShiftRng(77)
    for (C, H, W, kh, kw, pads, strides, dil) in ((5, 9, 7, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1)), (3, 8, 8, 3, 3, (0, 0, 1, 1), (2, 2), (1, 1)),
                                                  (4, 10, 6, 5, 3, (2, 1, 2, 1), (1, 2), (2, 1)), (2, 6, 6, 1, 1, (0, 0, 0, 0), (2, 2), (1, 1))):
        x = rng.f32(2 * C * H * W).reshape(2, C, H, W) - 0.5
        w = rng.f32(C * kh * kw).reshape(C, 1, kh, kw) - 0.5
        bias = rng.f32(C) - 0.5
        got = ref.conv2d_f32(x, w, bias, pads=pads, strides=strides, dilations=dil, groups=C)
        want = np.empty_like(got)
        for n in range(2):
            for c in range(C):
                for oy in range(got.shape[2]):
                    for ox in range(got.shape[3]):
                        acc = np.float32(bias[c])
                        for ky in range(kh):
                            iy = oy * strides[0] + ky * dil[0] - pads[0]
                            if not 0 <= iy < H:
                                continue
                            for kx in range(kw):
                                ix = ox * strides[1] + kx * dil[1] - pads[1]
                                if 0 <= ix < W:
                                    acc = np.float32(acc + np.float32(x[n, c, iy, ix] * w[c, 0, ky, kx]))
                        want[n, c, oy, ox] = acc
        assert np.array_equal(got.view(np.int32), want.view(np.int32))


def test_conv_transpose_literals():
    """conv_transpose.rs:603-760 (test_conv_transpose, _padding, _1d) + output-size cases + torch as an independent restatement."""
    x = np.array([1.0, 2.0, 3.0, 4.0], np.float32).reshape(1, 1, 2, 2)
    k = np.array([0.1, 0.2, 0.3, 0.4], np.float32).reshape(1, 1, 2, 2)
    eq_1e4(ref.conv_transpose2d_f32(x, k, strides=(2, 2)).ravel(), [0.1, 0.2, 0.2, 0.4, 0.3, 0.4, 0.6, 0.8, 0.3, 0.6, 0.4, 0.8, 0.9, 1.2, 1.2, 1.6])
    eq_1e4(ref.conv_transpose2d_f32(x, k, padding=(1, 1, 1, 1), strides=(2, 2)).ravel(), [0.4, 0.6, 0.6, 0.4])
    assert ref.conv_transpose2d_f32(x, k, padding="same", strides=(2, 2)).shape == (1, 1, 4, 4)
    eq_1e4(ref.conv_transpose2d_f32(x[:, :, :1, :], k[:, :, :1, :], strides=(1, 2)).ravel(), [0.1, 0.2, 0.2, 0.4])  # the 1-D case as 2-D
    assert ref.conv_transpose_output_size_and_padding((5, 5), (3, 3), "same", (2, 2)) == (10, 10, [0, 0, 1, 1])
    with pytest.raises(ValueError, match="Input is too small"):
        ref.conv_transpose_output_size_and_padding((1, 1), (1, 1), [2, 2, 2, 2], (1, 1))
    import torch
    rng = np.random.default_rng(0)
    x = rng.random((2, 6, 5, 7), dtype=np.float32) - 0.5
    w = rng.random((6, 4, 3, 2), dtype=np.float32) - 0.5
    b = rng.random(8, dtype=np.float32)
    got = ref.conv_transpose2d_f32(x, w, b, padding=(1, 0, 2, 1), strides=(2, 3), dilations=(1, 2), groups=2, output_padding=(1, 0))
    want = torch.nn.functional.conv_transpose2d(torch.tensor(x), torch.tensor(w), torch.tensor(b), stride=(2, 3), dilation=(1, 2), groups=2, output_padding=(1, 0)).numpy()
    np.testing.assert_allclose(got, want[:, :, 1:want.shape[2] - 2, 0:want.shape[3] - 1], rtol=1e-5, atol=1e-6)


def test_layer_norm_literals():
    for c in G["layer_norm"]["cases"]:
        x = np.array(c["input"], np.float32)
        if c["axis"] == -2:
            xs = x.reshape(x.shape[0], -1)
            n = xs.shape[-1]
            y = ref.layer_norm(xs, gamma=np.full(n, c["scale_full"], np.float32), beta=np.full(n, c["bias_full"], np.float32))
        elif "scale_scalar" in c:
            y = ref.layer_norm(x, gamma_scalar=c["scale_scalar"], beta_scalar=c.get("bias_scalar", 0.0))
        else:
            y = ref.layer_norm(x, gamma=np.array(c["scale"], np.float32), beta=np.array(c["bias"], np.float32))
        eq_1e4(y.reshape(np.array(c["expected"]).shape), c["expected"])


def test_softmax_literals():
    for lanes in (4, 8, 16):
        for c in G["softmax"]["cases"]:
            eq_1e4(ref.softmax(np.array(c["input"], np.float32), lanes=lanes), c["expected"])
    # all -inf: NaN unless flushed (attention.rs:1088-1106)
    x = np.full((1, 3), -np.inf, np.float32)
    assert np.isnan(ref.softmax(x)).all()
    assert ref.softmax(x, flush_nan=True).tolist() == [[0.0, 0.0, 0.0]]
    assert ref.softmax(np.zeros((0, 4), np.float32)).shape == (0, 4)


def test_add_softmax_equals_add_then_softmax():
    # test_add_softmax (attention.rs:1003-1088): AddSoftmax == Softmax(Add(qk, m))
    rng = ref.XorShiftRng(1234)
    qk = rng.f32(1 * 8 * 32 * 32).reshape(1, 8, 32, 32)
    m = rng.f32(32 * 32).reshape(1, 1, 32, 32)
    fused = ref.softmax(qk, addend=m, add_div=1, add_mod=32)
    plain = ref.softmax(ref.add(qk, m))
    assert np.array_equal(fused, plain)


def test_cast_scale_literals():
    for c in G["cast_scale"]["cases"]:
        y = ref.cast_scale(np.array(c["input"], np.int32), np.array(c["scale"], np.float32))
        assert y.tolist() == c["expected"]


def test_pool_literals():
    g = G["pool"]
    x = np.array(g["input4"], np.float32).reshape(1, 1, 4, 4)
    for c in g["average"]:
        expect_equal(ref.average_pool(x, c["kernel"], c["strides"])[0, 0], c["expected"])
    for c in g["max"]:
        expect_equal(ref.max_pool(x, c["kernel"], c["strides"])[0, 0], c["expected"])
    xp = np.broadcast_to(np.array(g["padding_input"], np.float32), (1, 5, 4, 4)).copy()
    y = ref.average_pool(xp, (2, 2), (2, 2), (1, 1, 1, 1))
    for ch in range(5):
        eq_1e4(y[0, ch], g["padding_expected"])
    y = ref.average_pool(xp, (2, 2), (2, 2), (1, 1, 1, 1), count_include_pad=True)
    eq_1e4(y[0, 0], g["padding_expected_include_pad"])
    ga = g["global_average"]
    for lanes in (4, 16):
        expect_equal(ref.global_average_pool(np.array(ga["input"], np.float32).reshape(ga["shape"]), lanes=lanes).ravel(), ga["expected"])


def test_output_size_literals():
    for c in G["output_size"]["cases"]:
        args = dict(in_size=c.get("in_size", [5, 5]), kernel=c.get("kernel", [3, 3]), strides=c.get("strides", [1, 1]),
                    padding=c.get("padding", [0, 0, 0, 0]), dilations=c.get("dilations", [1, 1]), ceil_mode=c.get("ceil", False))
        if "error" in c:
            with pytest.raises(ref.OpError, match=c["error"]):
                ref.calc_output_size_and_padding(**args)
        else:
            oh, ow, pads = ref.calc_output_size_and_padding(**args)
            assert [oh, ow, pads] == c["expected"]


def test_erf_gelu_exp_error_bounds():
    xs = np.arange(-6.0, 6.0, 1e-3, dtype=np.float32)
    truth = np.array([math.erf(float(v)) for v in xs])
    err = np.abs(ref.erf(xs).astype(np.float64) - truth).max()
    assert err <= G["erf"]["max_abs_error"] * 1.05 + 6e-8, err  # erf.rs:127-156 (libm::erff is itself ~0.5 ULP off)
    g_truth = 0.5 * xs.astype(np.float64) * (1 + np.array([math.erf(float(v) / math.sqrt(2)) for v in xs]))
    assert np.abs(ref.gelu(xs) - g_truth).max() < 3e-6
    # Exp: <= 1 ULP vs a correctly rounded exp (exp.rs:30-31)
    xe = np.linspace(-87.0, 88.0, 20001, dtype=np.float32)
    got = ref.exp(xe)
    want = np.exp(xe.astype(np.float64))
    ulp = np.abs(got.astype(np.float64) - want) / np.spacing(want.astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 1.0 + 1e-6, ulp.max()
    assert ref.exp(np.array([104.0, -104.0, 0.0], np.float32)).tolist() == [np.inf, 0.0, 1.0]


def test_dynamic_quantize_linear():
    # quantize.rs:704-767: dequantisation error bound + spec examples; exact algebra restated below
    for x in (np.array([-234.56], np.float32), np.array([234.56], np.float32), np.arange(-0.1, 0.1, 0.01, dtype=np.float32),
              ref.XorShiftRng(5).f32(1000) * 6 - 3):
        q, scale, zp = ref.dynamic_quantize_linear(x)
        deq = (q.astype(np.float32) - np.float32(zp)) * scale
        assert np.abs(deq - x).max() <= scale / 2 + 1e-6
        # independent numpy restatement of the operator definition (ONNX spec formula, quantize.rs:365-383)
        mn, mx = min(x.min(), np.float32(0)), max(x.max(), np.float32(0))
        s = np.float32(mx - mn) / np.float32(255)
        z = np.clip(np.rint(np.clip(np.float32(0) - mn / s, 0, 255)), 0, 255).astype(np.uint8)
        assert scale == s and zp == z
        want = np.clip(np.rint(x * (np.float32(1) / s)).astype(np.int64) + int(z), 0, 255).astype(np.uint8)
        assert np.array_equal(q, want)
    q, scale, zp = ref.dynamic_quantize_linear(np.zeros((0,), np.float32))
    assert (scale, zp) == (1.0, 0)
    q, scale, zp = ref.dynamic_quantize_linear(np.zeros((7,), np.float32))  # all zero -> scale 0, codes 0
    assert scale == 0.0 and zp == 0 and not q.any()


def reference_gemm_f64(a, b, alpha=1.0, beta=0.0, c=None, bias=None, bias_kind=0):
    acc = a.astype(np.float64) @ b.astype(np.float64) * alpha
    if c is not None and beta != 0:
        acc = acc + beta * c
    if bias is not None:
        acc = acc + (bias[:, None] if bias_kind == 1 else bias[None, :])
    return acc


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (2, 5, 20), (8, 8, 256), (10, 1025, 300), (64, 65, 257), (80, 4, 1), (16, 1024, 0), (0, 5, 3)])
def test_gemm_f32_vs_f64_truth(m, n, k):
    # size matrix in the spirit of rten-gemm/src/tests.rs:336-362; tolerance: f32 accumulation error bound
    rng = ref.XorShiftRng(1234)
    a = rng.f32(m * k).reshape(m, k) - 0.5
    b = rng.f32(k * n).reshape(k, n) - 0.5
    got = ref.gemm_f32(a, b)
    want = reference_gemm_f64(a, b)
    assert got.shape == (m, n)
    if got.size:
        assert np.abs(got - want).max() <= 1e-6 * max(k, 1)


def test_gemm_f32_options():
    rng = ref.XorShiftRng(7)
    a = rng.f32(20 * 300).reshape(20, 300) - 0.5
    b = rng.f32(300 * 33).reshape(300, 33) - 0.5
    c = rng.f32(20 * 33).reshape(20, 33)
    bias_r, bias_c = rng.f32(20), rng.f32(33)
    for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (0.5, 0.0), (0.5, 2.0)):
        got = ref.gemm_f32(a, b, c=c, alpha=alpha, beta=beta)
        np.testing.assert_allclose(got, reference_gemm_f64(a, b, alpha, beta, c), rtol=0, atol=5e-5)
    # beta == 0 must not read C (NaN-poisoned output, tests.rs:632-674)
    got = ref.gemm_f32(a, b, c=np.full((20, 33), np.nan, np.float32), beta=0.0)
    assert not np.isnan(got).any()
    np.testing.assert_allclose(ref.gemm_f32(a, b, bias=bias_r, bias_kind=ref.BIAS_PER_ROW), reference_gemm_f64(a, b, bias=bias_r, bias_kind=1), atol=5e-5)
    np.testing.assert_allclose(ref.gemm_f32(a, b, bias=bias_c, bias_kind=ref.BIAS_PER_COL), reference_gemm_f64(a, b, bias=bias_c, bias_kind=2), atol=5e-5)
    # transposed / strided operands give bit-identical results (strides only change addressing, tests.rs:523-569)
    assert np.array_equal(ref.gemm_f32(np.ascontiguousarray(a.T).T, np.ascontiguousarray(b.T).T), ref.gemm_f32(a, b))


def test_gemm_f32_accumulation_order_is_kc_blocked():
    # depth blocks of 256 with k-ordered fma chains (rten-gemm/src/lib.rs:630-633, simd_generic.rs:326-344)
    rng = ref.XorShiftRng(99)
    K = 600
    a = rng.f32(K).reshape(1 + 0, K)
    a = np.vstack([a, a])  # M = 2 (avoid the gemv path semantics note)
    b = rng.f32(K * 3).reshape(K, 3)
    bias = np.array([0.25, -0.5], np.float32)
    got = ref.gemm_f32(a, b, bias=bias, bias_kind=ref.BIAS_PER_ROW)
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.fmaf.restype = ctypes.c_float
    libm.fmaf.argtypes = [ctypes.c_float] * 3
    for j in range(3):
        total = None
        for k0 in range(0, K, 256):
            acc = np.float32(0)
            for k in range(k0, min(k0 + 256, K)):
                acc = np.float32(libm.fmaf(float(a[0, k]), float(b[k, j]), float(acc)))
            total = np.float32(acc + bias[0]) if total is None else np.float32(total + acc)
        assert got[0, j] == total


def reference_gemm_int(a, b, a_zp, b_zp):
    az = np.zeros(a.shape[0], np.int64) if a_zp is None else np.broadcast_to(np.asarray(a_zp, np.int64), (a.shape[0],))
    bz = np.zeros(b.shape[1], np.int64) if b_zp is None else np.broadcast_to(np.asarray(b_zp, np.int64), (b.shape[1],))
    r = (a.astype(np.int64) - az[:, None]) @ (b.astype(np.int64) - bz[None, :])
    return ((r + 2**31) % 2**32 - 2**31).astype(np.int32)


@pytest.mark.parametrize("adt,bdt", [(np.uint8, np.int8), (np.uint8, np.uint8), (np.int8, np.int8), (np.int8, np.uint8)])
def test_gemm_int8_all_signedness(adt, bdt):
    # src/ops/matmul.rs:1365-1750 (16 cases + 4 signedness macros vs reference_matmul_integer)
    rng = ref.XorShiftRng(1234)
    for (m, n, k) in ((1, 1, 1), (5, 7, 3), (16, 33, 64), (8, 4, 300)):
        a = (rng.u8(m * k) if adt == np.uint8 else rng.i8(m * k)).reshape(m, k)
        b = (rng.u8(k * n) if bdt == np.uint8 else rng.i8(k * n)).reshape(k, n)
        for a_zp, b_zp in ((None, None), (np.array([3], adt), np.array([5], bdt)),
                           ((rng.u8(m) if adt == np.uint8 else rng.i8(m)), (rng.u8(n) if bdt == np.uint8 else rng.i8(n)))):
            got = ref.gemm_int8(a, b, a_zp, b_zp)
            assert np.array_equal(got, reference_gemm_int(a, b, a_zp, b_zp))


def test_conv_int8_against_integer_definition():
    import torch
    rng = ref.XorShiftRng(11)
    x = rng.u8(2 * 4 * 7 * 6).reshape(2, 4, 7, 6)
    w = rng.i8(6 * 2 * 3 * 3, reduced=True).reshape(6, 2, 3, 3)
    w_zp = rng.i8(6, reduced=True)
    # no padding: all pad modes agree (the reference's own int8 conv tests use Padding::zero, conv.rs:1505)
    outs = [ref.conv2d_int8(x, w, x_zp=9, w_zp=w_zp, groups=2, strides=(2, 1), pad_mode=pm) for pm in (0, 1, 2)]
    want = torch.nn.functional.conv2d(torch.tensor(x.astype(np.float64) - 9),
                                      torch.tensor(w.astype(np.float64)) - torch.tensor(w_zp.astype(np.float64)).view(6, 1, 1, 1),
                                      stride=(2, 1), groups=2).numpy()
    for o in outs:
        assert np.array_equal(o, want.astype(np.int32))
    # with padding: ZERO_POINT == ONNX semantics; RAW0_I8 == padding with u8 value 128
    zp_out = ref.conv2d_int8(x, w, x_zp=9, w_zp=w_zp, groups=2, pads=(1, 1, 1, 1), pad_mode=ref.PAD_ZERO_POINT)
    xpad = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=9)
    want = torch.nn.functional.conv2d(torch.tensor(xpad - 9), torch.tensor(w.astype(np.float64)) - torch.tensor(w_zp.astype(np.float64)).view(6, 1, 1, 1), groups=2).numpy()
    assert np.array_equal(zp_out, want.astype(np.int32))
    raw_out = ref.conv2d_int8(x, w, x_zp=9, w_zp=w_zp, groups=2, pads=(1, 1, 1, 1), pad_mode=ref.PAD_RAW0_I8)
    xpad = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=128)
    want = torch.nn.functional.conv2d(torch.tensor(xpad - 9), torch.tensor(w.astype(np.float64)) - torch.tensor(w_zp.astype(np.float64)).view(6, 1, 1, 1), groups=2).numpy()
    assert np.array_equal(raw_out, want.astype(np.int32))


def test_conv_f32_vs_torch_sweep():
    import torch
    rng = ref.XorShiftRng(3)
    cases = [dict(N=2, C=8, H=9, W=11, O=6, k=(3, 3), pads=(1, 1, 1, 1), strides=(1, 1), dil=(1, 1), groups=1),
             dict(N=1, C=4, H=12, W=12, O=8, k=(3, 3), pads=(0, 1, 2, 1), strides=(2, 2), dil=(1, 1), groups=2),
             dict(N=3, C=6, H=10, W=7, O=6, k=(1, 1), pads=(0, 0, 0, 0), strides=(1, 1), dil=(1, 1), groups=1),
             dict(N=1, C=3, H=16, W=16, O=4, k=(7, 7), pads=(3, 3, 3, 3), strides=(2, 2), dil=(1, 1), groups=1),
             dict(N=1, C=2, H=14, W=14, O=2, k=(3, 3), pads=(2, 2, 2, 2), strides=(1, 1), dil=(2, 2), groups=1)]
    for c in cases:
        x = rng.f32(c["N"] * c["C"] * c["H"] * c["W"]).reshape(c["N"], c["C"], c["H"], c["W"]) - 0.5
        w = rng.f32(c["O"] * (c["C"] // c["groups"]) * c["k"][0] * c["k"][1]).reshape(c["O"], c["C"] // c["groups"], *c["k"]) - 0.5
        b = rng.f32(c["O"])
        got = ref.conv2d_f32(x, w, b, pads=c["pads"], strides=c["strides"], dilations=c["dil"], groups=c["groups"])
        pt, pl, pb, pr = c["pads"]
        xp = torch.nn.functional.pad(torch.tensor(x), (pl, pr, pt, pb))
        want = torch.nn.functional.conv2d(xp, torch.tensor(w), torch.tensor(b), stride=c["strides"], dilation=c["dil"], groups=c["groups"]).numpy()
        expect_equal(got, want) if False else np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def test_sdpa_vs_torch():
    import torch
    rng = ref.XorShiftRng(21)
    q = rng.f32(2 * 3 * 16 * 8).reshape(2, 3, 16, 8) - 0.5
    k = rng.f32(2 * 3 * 16 * 8).reshape(2, 3, 16, 8) - 0.5
    v = rng.f32(2 * 3 * 16 * 8).reshape(2, 3, 16, 8) - 0.5
    mask = np.where(rng.f32(2 * 16).reshape(2, 1, 1, 16) > 0.3, 0.0, -np.inf).astype(np.float32)
    got = ref.sdpa(q, k, v, mask=mask)
    want = torch.nn.functional.scaled_dot_product_attention(torch.tensor(q), torch.tensor(k), torch.tensor(v), attn_mask=torch.tensor(mask)).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


# ---- MatMulNBits: the reference pins its optimised paths structurally (rten-gemm/src/block_quant.rs:940-1072): same seeded
# inputs, optimised result == naive f32 loop under expect_equal.  Restated here for the oracle's two paths.
def _naive_block_quant_gemm(lhs, quant, scales):
    """reference_gemm_f32_with_block_quantized_rhs (block_quant.rs:820-849): acc += lhs * ((q - 8) as f32 * scale), k ascending, in f32."""
    m, k = lhs.shape
    n, kb, half = quant.shape
    bs = half * 2
    out = np.zeros((m, n), np.float32)
    for row in range(m):
        for col in range(n):
            acc = np.float32(0)
            for ki in range(k):
                byte = int(quant[col, ki // bs, (ki % bs) // 2])
                elem = (byte & 0x0F) if ki % 2 == 0 else (byte >> 4)
                deq = np.float32(elem - 8) * scales[col, ki // bs]
                acc = np.float32(acc + np.float32(lhs[row, ki] * deq))
            out[row, col] = acc
    return out


def _block_quant_case(n_rows, n_cols, n_blocks, block_size):
    rng = ref.XorShiftRng(1234)
    lhs = rng.f32(n_rows * n_blocks * block_size).reshape(n_rows, n_blocks * block_size)
    quant = rng.u8(n_cols * n_blocks * (block_size // 2)).reshape(n_cols, n_blocks, block_size // 2)
    scales = rng.f32(n_cols * n_blocks).reshape(n_cols, n_blocks)
    return lhs, quant, scales


@pytest.mark.parametrize("block_size", [16, 32, 64, 128, 256])
def test_matmul_nbits_vector_cases(block_size):
    # block_quant.rs:957-966: one row, three columns, max(128 / bs, 1) blocks
    lhs, quant, scales = _block_quant_case(1, 3, max(128 // block_size, 1), block_size)
    got = ref.matmul_nbits_f32(lhs[None], quant, scales)[0]
    want = _naive_block_quant_gemm(lhs, quant, scales)
    assert np.all(np.abs(got - want) <= 1e-8 + 1e-5 * np.abs(want)), (got, want)  # expect_equal (test_util.rs:47-93)


def test_matmul_nbits_vector_main_and_tail():
    # block_quant.rs:968-979: block 16, 128 / 16 + 1 blocks: one vector step plus a 16-element scalar tail
    lhs, quant, scales = _block_quant_case(1, 1, 128 // 16 + 1, 16)
    got = ref.matmul_nbits_f32(lhs[None], quant, scales)[0]
    want = _naive_block_quant_gemm(lhs, quant, scales)
    assert np.all(np.abs(got - want) <= 1e-8 + 1e-5 * np.abs(want))


def test_matmul_nbits_empty_k_and_matrix_path():
    out = ref.matmul_nbits_f32(np.zeros((1, 1, 0), np.float32), np.zeros((1, 0, 16), np.uint8), np.zeros((1, 0), np.float32))
    assert out.shape == (1, 1, 1) and out[0, 0, 0] == 0.0  # block_quant.rs:82-86
    # rows > 1 (contrib.rs:86-100, test_matmul_nbits :270-330): the f32 GEMM on the dequantised matrix
    lhs, quant, scales = _block_quant_case(5, 7, 4, 32)
    got = ref.matmul_nbits_f32(lhs, quant, scales)
    want = _naive_block_quant_gemm(lhs, quant, scales)
    assert np.all(np.abs(got - want) <= 1e-8 + 1e-5 * np.abs(want))
    deq = ref.dequantize_4bit(quant, scales)
    np.testing.assert_array_equal(got, ref.gemm_f32(lhs, deq))


def test_quantize_4bit_blocks_round_trip():
    rng = np.random.default_rng(5)
    w = rng.standard_normal((128, 24)).astype(np.float32)
    packed, scales = ref.quantize_4bit_blocks(w, 32)
    assert packed.shape == (24, 4, 16) and scales.shape == (24, 4)
    deq = ref.dequantize_4bit(packed, scales)
    step = np.repeat(np.abs(scales), 32, axis=1).T  # [K, N]: one quantisation step per element (the +7 side clips, so allow a full step)
    assert np.all(np.abs(deq - w) <= step * 1.0001 + 1e-6)


def test_real_reference_check_recorded():
    """The oracle is pinned to the reference's unit-test literals only until somebody with a Rust toolchain runs the three
    commands of tools/make_rten_golden.md and commits their report; then every recorded `max diff` must be zero."""
    path = os.path.join(os.path.dirname(__file__), "golden", "rten_check.txt")
    if not os.path.exists(path):
        pytest.xfail("tests/golden/rten_check.txt absent: no Rust toolchain in this image (tools/make_rten_golden.md)")
    import re
    diffs = [float(d) for d in re.findall(r"max diff ([0-9.eE+-]+)", open(path).read())]
    assert len(diffs) >= 3 and all(d == 0.0 for d in diffs), diffs


def _fma(a, b, c):
    """f32 fused multiply-add through 80-bit intermediates (product exact; a double rounding would need a 64-bit tie)."""
    return np.float32(np.longdouble(a) * np.longdouble(b) + np.longdouble(c))


def _gemv_emulation(a, b, out0, alpha, beta, bias, threads=0):
    """The reference's vector-matrix product written a second time, straight from rten-gemm/src/lib.rs:668-747 and
    kernels/simd_generic.rs:14-197 (AVX-512: 16 lanes, 32-column tiles), with numpy scalars."""
    f = np.float32
    K, N = b.shape
    rs, cs = b.strides[0] // 4, b.strides[1] // 4
    out = None if out0 is None else out0.astype(np.float32).copy()
    res = np.zeros(N, np.float32)
    cb = 128 if threads == 0 else max(128, -(-N // threads))
    kb = 512 if rs == 1 else 8
    for c0 in range(0, N, cb):
        nc = min(cb, N - c0)
        for c in range(nc):
            col = c0 + c
            eff_beta = f(beta)
            o = f(0) if out is None else out[col]
            for k0 in range(0, K, kb):
                depth = min(kb, K - k0)
                ak, bk = a[k0:k0 + depth], b[k0:k0 + depth, col]
                if rs == 1 and c < nc // 8 * 8:
                    lanes = [f(0)] * 16
                    dt = depth // 16 * 16
                    for d in range(0, dt, 16):
                        for l in range(16):
                            lanes[l] = _fma(ak[d + l], bk[d + l], lanes[l])
                    w = 8
                    while w >= 1:
                        for l in range(w):
                            lanes[l] = f(lanes[l] + lanes[l + w])
                        w //= 2
                    acc = lanes[0]
                    for k in range(dt, depth):
                        acc = _fma(ak[k], bk[k], acc)
                    o = f(f(alpha) * acc) if eff_beta == 0 else f(f(f(alpha) * acc) + f(eff_beta * o))
                elif rs == 1 or cs != 1:
                    acc = f(0)
                    for k in range(depth):
                        acc = _fma(ak[k], bk[k], acc)
                    acc = f(acc * f(alpha))
                    o = acc if eff_beta == 0 else f(acc + f(eff_beta * o))
                elif c < nc // 32 * 32:
                    acc = f(0)
                    for k in range(depth):
                        acc = _fma(ak[k], bk[k], acc)
                    if alpha != 1.0:
                        acc = f(acc * f(alpha))
                    o = acc if eff_beta == 0 else (f(o + acc) if eff_beta == 1 else _fma(o, eff_beta, acc))
                else:
                    acc = f(0)
                    for k in range(depth):
                        acc = f(acc + f(ak[k] * bk[k]))
                    tmp = f(0) if eff_beta == 0 else o
                    o = f(f(eff_beta * tmp) + f(acc * f(alpha)))
                eff_beta = f(1)
            if bias is not None:
                o = f(o + (bias[col] if len(bias) > 1 else bias[0]))
            res[col] = o
    return res


@pytest.mark.parametrize("layout", ["rowmajor", "transposed", "strided"])
def test_gemv_path_restated_twice(layout):
    """M == 1 takes the reference's gemv kernels (not the blocked GEMM order): the C restatement against a second, scalar one."""
    rng = ref.XorShiftRng(31)
    for (K, N) in ((1, 1), (7, 5), (8, 32), (20, 45), (530, 40), (2048, 137), (33, 300)):
        a = rng.f32(K) - 0.5
        if layout == "rowmajor":
            b = rng.f32(K * N).reshape(K, N) - 0.5
        elif layout == "transposed":
            b = (rng.f32(K * N).reshape(N, K) - 0.5).T
        else:
            b = (rng.f32(K * N * 6).reshape(K * 2, N * 3) - 0.5)[::2, ::3]
        c = rng.f32(N) - 0.5
        bias = rng.f32(N) - 0.5
        for alpha, beta, use_bias in ((1.0, 0.0, False), (1.0, 1.0, True), (0.5, 2.0, True), (0.25, 0.0, False)):
            got = ref.gemm_f32(a.reshape(1, K), b, c=c.reshape(1, N) if beta != 0 else None, alpha=alpha, beta=beta,
                               bias=bias if use_bias else None, bias_kind=ref.BIAS_PER_COL if use_bias else ref.BIAS_NONE)
            want = _gemv_emulation(a, b, c if beta != 0 else None, alpha, beta, bias if use_bias else None)
            assert np.array_equal(got.ravel().view(np.uint32), want.view(np.uint32)), (layout, K, N, alpha, beta)
    # the thread-count assumption is observable: left-over columns of a 250-column block take the scalar kernel
    a = rng.f32(300) - 0.5
    b = rng.f32(300 * 1000).reshape(300, 1000) - 0.5
    base = ref.gemm_f32(a.reshape(1, -1), b)
    try:
        ref.set_gemv_threads(4)
        other = ref.gemm_f32(a.reshape(1, -1), b)
        assert np.array_equal(other.ravel().view(np.uint32), _gemv_emulation(a, b, None, 1.0, 0.0, None, threads=4).view(np.uint32))
        assert not np.array_equal(other, base) and np.allclose(other, base, rtol=1e-4, atol=1e-5)
        ref.set_gemv_enabled(False)  # "prepacked B": the blocked order, one fma chain per 256-deep block
        blocked = ref.gemm_f32(a.reshape(1, -1), b)
        two = ref.gemm_f32(np.stack([a, a]), b)
        assert np.array_equal(blocked.ravel().view(np.uint32), two[0].view(np.uint32))
    finally:
        ref.set_gemv_threads(0)
        ref.set_gemv_enabled(True)
    assert np.array_equal(ref.gemm_f32(np.stack([a, a]), b)[1].view(np.uint32), blocked.ravel().view(np.uint32))


def test_one_row_products_inside_sdpa_and_conv_transpose_take_the_gemv_order():
    """sdpa_head (src/ops/attention.rs:518-562) and conv_transpose (src/ops/conv_transpose.rs:376-383) call gemm on UNPACKED operands, so a single query
    row / a one-row kernel matrix takes gemm_impl's vector-matrix branch (rten-gemm/src/lib.rs:876-891) like any other one-row product: the oracle's
    composite equals the composition of its own one-row GEMM (which makes that choice), softmax and one-row GEMM -- and, where the two orders differ in
    the last bits, NOT the blocked chain."""
    rng = ref.XorShiftRng(99)
    differs = 0
    for (T, D) in ((40, 64), (129, 32), (700, 128)):
        q = rng.f32(D).reshape(1, 1, 1, D) - 0.5
        k = rng.f32(T * D).reshape(1, 1, T, D) - 0.5
        v = rng.f32(T * D).reshape(1, 1, T, D) - 0.5
        m = ((rng.f32(T).reshape(1, 1, 1, T) - 0.5) * 4).astype(np.float32)
        got = ref.sdpa(q, k, v, mask=m, scale=0.125)
        scores = ref.gemm_f32(q[0, 0], k[0, 0].T, alpha=0.125)            # one row: the gemv order (transposed B)
        p = ref.softmax(scores, addend=m.reshape(1, T), flush_nan=True)
        want = ref.gemm_f32(p, v[0, 0])                                      # one row: the gemv order (row-major B)
        expect_equal(got[0, 0], want)
        ref.set_gemv_enabled(False)
        try:
            blocked = ref.sdpa(q, k, v, mask=m, scale=0.125)
        finally:
            ref.set_gemv_enabled(True)
        differs += int((blocked.view(np.int32) != got.view(np.int32)).any())
    assert differs > 0, "the vector-matrix order and the blocked order agreed on every case: the test does not discriminate"
    # ConvTranspose with a one-row kernel matrix (O_g = kh = kw = 1)
    x = rng.f32(2 * 300 * 5 * 7).reshape(2, 300, 5, 7) - 0.5
    w = rng.f32(300).reshape(300, 1, 1, 1) - 0.5
    got = ref.conv_transpose2d_f32(x, w, None, (0, 0, 0, 0), (1, 1))
    for n in range(2):
        expect_equal(got[n, 0].reshape(1, 35), ref.gemm_f32(w.reshape(1, 300), x[n].reshape(300, 35)) + np.float32(0))
