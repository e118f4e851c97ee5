"""Round-6 kernels against the oracle / numpy: the cross-workgroup K split of the int8 convolution kernel (KS), the streaming LayerNormalization (rows in
sequence per wave), Tanh, and the layout / logic operators behind rten_hip_elementwise_nd / rten_hip_gather_axis_b32 / rten_hip_copy_rows_b32."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref
from rten_amd import lib as L, ops
from rten_amd.tensor import DeviceTensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def dev(ctx, a):
    return DeviceTensor.from_numpy(ctx, a)


def _bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    same = a.view(np.int32) == b.view(np.int32)
    assert same.all(), f"{(~same).sum()} of {same.size} differ, first at {tuple(np.argwhere(~same)[0])}"


KS_CASES = [(2, 128, 14, 14, 70, 3, (1, 1, 1, 1), (1, 1)),    # K 1152: 18 k-tiles; ragged rows (M = 70)
            (3, 256, 7, 7, 96, 3, (1, 1, 1, 1), (1, 1)),      # K 2304: 36 k-tiles; N = 147 (ragged columns)
            (2, 1024, 5, 5, 64, 1, (0, 0, 0, 0), (1, 1)),     # pointwise, K 1024: 16 k-tiles
            (1, 512, 9, 9, 130, 3, (1, 1, 1, 1), (2, 2)),     # stride 2, K 4608: 72 k-tiles
            (2, 2048, 3, 3, 200, 1, (0, 0, 0, 0), (1, 1))]    # K 2048, 32 k-tiles, 18 columns


def ks_cases_main():
    """(runs in a child process whose RTEN_I8_KS selects the split: the knob is read once per process)"""
    c = L.Context(0)
    for (N, Cc, H, W, O, k, pads, strides) in KS_CASES:
        rng = ref.XorShiftRng(17 + Cc + O)
        x = rng.u8(N * Cc * H * W).reshape(N, Cc, H, W)
        w = rng.i8(O * Cc * k * k, reduced=True).reshape(O, Cc, k, k)
        x_zp, scale = np.array(117, np.uint8), np.array(0.0071, np.float32)
        bias = rng.f32(O) - 0.5
        acc = ref.conv2d_int8(x, w, x_zp=117, pads=pads, strides=strides, pad_mode=ref.PAD_RAW0_I8)
        want_plain = ref.cast_scale(acc, scale)
        want_relu = ref.relu(want_plain + bias[None, :, None, None])
        c.profile_reset()
        c.profile(True)
        op = ops.ConvIntegerToFloat(ops.ConvInteger(padding=list(pads), strides=strides))
        ins = [dev(c, x), dev(c, w), dev(c, x_zp), None, dev(c, scale)]
        got = op.run(c, ins)[0].numpy()
        c.sync()
        c.profile(False)
        kernels = [r["kernel"] for r in c.profile_report()]
        assert any(",ks>" in n for n in kernels), kernels  # the split form is what ran
        _bits(got, want_plain)
        op_r = ops.ConvIntegerToFloat(ops.ConvInteger(padding=list(pads), strides=strides), fuse_relu=True)
        first = None
        for _ in range(40):  # whichever workgroup arrives last: the same bits
            g = op_r.run(c, ins + [dev(c, bias)])[0].numpy()
            if first is None:
                first = g
                _bits(g, want_relu)
            else:
                assert np.array_equal(g.view(np.int32), first.view(np.int32))
        # a residual keeps the unsplit kernel (the split form does not carry the residual tile): the oracle's bits either way
        res = rng.f32(want_plain.size).reshape(want_plain.shape) - 0.5
        g = op_r.run(c, ins + [dev(c, bias), dev(c, res)])[0].numpy()
        _bits(g, ref.relu(ref.add(want_plain + bias[None, :, None, None], res)))
    c.close()
    print("KS CASES OK")


@pytest.mark.parametrize("setting", ["2,3", "4,3", "3,1", "2,0"])
def test_int8_cross_workgroup_k_split_is_bit_exact_and_deterministic(setting):
    """The KS kernels (a measurement knob since they lose on every ResNet layer: profiles/r09/int8_cross_workgroup_k_split.txt): `parts` workgroups per tile sum
    disjoint slices of K, the last arrival adds the parked partials.  Integer sums: the oracle's bits, launch after launch, whichever workgroup arrives
    last; bias, Relu and the output statistics ride the last arrival's epilogue."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTEN_I8_KS=setting, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", "from tests.test_gpu_round6 import ks_cases_main; ks_cases_main()"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "KS CASES OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


@pytest.mark.parametrize("rows,cols", [(16384, 768), (16387, 768), (17001, 200), (16385, 1024), (16500, 129), (20000, 512), (4096, 768)])
def test_layer_norm_rows_in_sequence_is_bit_exact(ctx, rows, cols):
    """Launches with at least 16384 rows of 129..1024 columns run layer_norm_stream_kernel (a wave owns four consecutive rows, the next row's loads in flight
    under the current row's reductions; smaller launches keep one row per wave -- profiles/r09/layer_norm_rows.txt): the reference's 16-lane order per row,
    ragged last waves, column tails, with and without the fused residual Add, in place."""
    rng = ref.XorShiftRng(rows + cols)
    x = (rng.f32(rows * cols).reshape(rows, cols) - 0.5) * 3
    r = (rng.f32(rows * cols).reshape(rows, cols) - 0.5) * 2
    g, b = rng.f32(cols) + 0.5, rng.f32(cols) - 0.5
    xd, rd, gd, bd = dev(ctx, x), dev(ctx, r), dev(ctx, g), dev(ctx, b)
    out = DeviceTensor(ctx, (rows, cols), np.float32)
    ctx.profile_reset()
    ctx.profile(True)
    ctx.call("rten_hip_layer_norm_f32", rows, cols, xd.vp, gd.vp, bd.vp, 1.0, 0.0, 1e-5, out.vp)
    ctx.sync()
    ctx.profile(False)
    _bits(out.numpy(), ref.layer_norm(x, g, b, 1.0, 0.0, eps=1e-5, lanes=16))
    ctx.call("rten_hip_layer_norm_f32", rows, cols, xd.vp, gd.vp, None, 1.0, 0.0, 1e-5, out.vp)  # scale only
    _bits(out.numpy(), ref.layer_norm(x, g, None, 1.0, 0.0, eps=1e-5, lanes=16))
    ctx.call("rten_hip_layer_norm_f32", rows, cols, xd.vp, None, None, 2.0, 0.5, 1e-5, out.vp)  # scalar scale / bias
    _bits(out.numpy(), ref.layer_norm(x, None, None, 2.0, 0.5, eps=1e-5, lanes=16))
    want = ref.layer_norm(ref.add(x, r), g, b, 1.0, 0.0, eps=1e-12, lanes=16)
    ctx.call("rten_hip_add_layer_norm_f32", rows, cols, xd.vp, rd.vp, gd.vp, bd.vp, 1.0, 0.0, 1e-12, out.vp)
    _bits(out.numpy(), want)
    ctx.call("rten_hip_add_layer_norm_f32", rows, cols, xd.vp, rd.vp, gd.vp, bd.vp, 1.0, 0.0, 1e-12, rd.vp)  # in place over the addend (BERT's residual stream)
    _bits(rd.numpy(), want)


def test_tanh_is_the_reference_polynomial_and_exp_forms(ctx):
    rng = ref.XorShiftRng(21)
    x = np.concatenate([(rng.f32(200003) - 0.5) * 20, (rng.f32(50000) - 0.5) * 1.2, (rng.f32(5000) - 0.5) * 1e-3,
                        np.array([0.0, -0.0, 0.0004, 0.00040001, 0.55, 0.55000001, 9.02, 9.0199995, -9.02, 50.0, -50.0, 104.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40], np.float32)]).astype(np.float32)
    y = DeviceTensor(ctx, x.shape, np.float32)
    ctx.call("rten_hip_tanh_f32", x.size, dev(ctx, x).vp, y.vp)
    got, want = y.numpy(), ref.tanh(x)
    same = (got.view(np.int32) == want.view(np.int32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (x[~same][:5], got[~same][:5], want[~same][:5])
    fin = np.isfinite(x) & (x != 0)
    t = np.tanh(x[fin].astype(np.float64)).astype(np.float32)
    ulp = np.abs(got[fin].view(np.int32).astype(np.int64) - t.view(np.int32).astype(np.int64))
    assert ulp.max() <= 3  # the reference's own bound (tanh.rs: MAX_TANH_ERROR_ULPS)


def _strides(shape, out):
    st, acc = [0] * len(out), 1
    for i in range(len(shape) - 1, -1, -1):
        st[len(out) - len(shape) + i] = 0 if shape[i] == 1 else acc
        acc *= shape[i]
    return st


def _ew(ctx, op, a, b=None, c=None, y_dtype=np.int32):
    dt = {np.dtype(np.float32): 0, np.dtype(np.int32): 1, np.dtype(np.uint8): 2, np.dtype(np.int8): 3}
    shapes = [t.shape for t in (a, b, c) if t is not None]
    out = np.broadcast_shapes(*shapes)
    nd = len(out)
    I64 = C.c_int64 * max(nd, 1)
    arr = lambda v: I64(*(list(v) + [0] * (max(nd, 1) - len(v))))  # noqa: E731
    y = DeviceTensor(ctx, out, y_dtype)
    da, db, dc = dev(ctx, a), dev(ctx, b) if b is not None else None, dev(ctx, c) if c is not None else None  # (alive until y.numpy() below)
    ctx.call("rten_hip_elementwise_nd", op, nd, arr(out), da.vp, dt[a.dtype], arr(_strides(a.shape, out)), db.vp if db else None, dt[b.dtype] if b is not None else 0,
             arr(_strides(b.shape, out)) if b is not None else None, dc.vp if dc else None, arr(_strides(c.shape, out)) if c is not None else None, y.vp, dt[np.dtype(y_dtype)])
    res = y.numpy()
    del da, db, dc
    return res


def test_elementwise_nd_cast_logic_compare_where(ctx):
    """src/ops/convert.rs (Rust `as`), binary_elementwise.rs (booleans are int32 0 / 1; numpy broadcasting), where_op."""
    rng = np.random.default_rng(4)
    f = np.array([1.9, -1.9, 0.0, -0.0, 3e10, -3e10, np.nan, 255.7, -128.9, 127.5, 1e-3], np.float32)
    assert _ew(ctx, 0, f, y_dtype=np.int32).tolist() == [1, -1, 0, 0, 2147483647, -2147483648, 0, 255, -128, 127, 0]
    assert _ew(ctx, 0, f, y_dtype=np.uint8).tolist() == [1, 0, 0, 0, 255, 0, 0, 255, 0, 127, 0]
    assert _ew(ctx, 0, f, y_dtype=np.int8).tolist() == [1, -1, 0, 0, 127, -128, 0, 127, -128, 127, 0]
    i = np.array([0, 1, -1, 255, 256, 300, -129, 2147483647], np.int32)
    assert _ew(ctx, 0, i, y_dtype=np.uint8).tolist() == (i & 0xff).astype(np.uint8).tolist()
    assert _ew(ctx, 0, i, y_dtype=np.int8).tolist() == i.astype(np.int8).tolist()
    np.testing.assert_array_equal(_ew(ctx, 0, np.arange(256, dtype=np.uint8), y_dtype=np.float32), np.arange(256, dtype=np.float32))
    np.testing.assert_array_equal(_ew(ctx, 0, np.arange(-128, 128).astype(np.int8), y_dtype=np.int32), np.arange(-128, 128, dtype=np.int32))
    a = rng.integers(-3, 4, (2, 1, 5, 1)).astype(np.int32)
    b = rng.integers(-3, 4, (3, 1, 7)).astype(np.int32)
    for op, fn in ((2, lambda x, y: (x != 0) & (y != 0)), (3, lambda x, y: (x != 0) | (y != 0)), (4, lambda x, y: (x != 0) ^ (y != 0)), (5, np.equal), (6, np.less),
                   (7, np.less_equal), (8, np.greater), (9, np.greater_equal), (11, np.add), (12, np.subtract), (13, np.multiply)):
        np.testing.assert_array_equal(_ew(ctx, op, a, b), fn(a, b).astype(np.int32), err_msg=str(op))
    np.testing.assert_array_equal(_ew(ctx, 1, a), (a == 0).astype(np.int32))
    d = np.where(b == 0, 1, b).astype(np.int32)
    np.testing.assert_array_equal(_ew(ctx, 14, a, d), np.trunc(a / d).astype(np.int32))  # truncation toward zero
    fa, fb = rng.standard_normal((4, 1, 6)).astype(np.float32), rng.standard_normal((5, 1)).astype(np.float32)
    fa[0, 0, 0] = np.nan
    for op, fn in ((5, np.equal), (6, np.less), (9, np.greater_equal)):
        np.testing.assert_array_equal(_ew(ctx, op, fa, fb), fn(fa, fb).astype(np.int32))
    cond = rng.integers(0, 2, (2, 1, 8, 1)).astype(np.int32) * 7
    xs, ys = np.array([0.0], np.float32), np.array([-3.4028234663852886e38], np.float32)
    got = _ew(ctx, 10, cond, xs, ys, y_dtype=np.float32)
    np.testing.assert_array_equal(got, np.where(cond != 0, xs, ys))
    xi, yi = rng.integers(-9, 9, (1, 3, 1, 4)).astype(np.int32), rng.integers(-9, 9, (8, 1)).astype(np.int32)
    np.testing.assert_array_equal(_ew(ctx, 10, cond, xi, yi), np.where(cond != 0, xi, yi))


def test_gather_axis_and_copy_rows(ctx):
    rng = np.random.default_rng(5)
    data = rng.standard_normal((3, 5, 7)).astype(np.float32)
    for axis, ids in ((1, np.array([4, 0, -1, 2], np.int32)), (0, np.array([[2, 0], [1, -3]], np.int32)), (2, np.array(3, np.int32))):
        outer, inner = int(np.prod(data.shape[:axis])), int(np.prod(data.shape[axis + 1:]))
        want = np.take(data, ids, axis=axis)
        y = DeviceTensor(ctx, want.shape, np.float32)
        dd, di = dev(ctx, data), dev(ctx, ids.reshape(-1))  # (kept alive until the result is read)
        ctx.call("rten_hip_gather_axis_b32", outer, data.shape[axis], inner, ids.size, dd.vp, di.vp, y.vp)
        np.testing.assert_array_equal(y.numpy(), want)
    a, b = rng.integers(0, 100, (4, 3, 5)).astype(np.int32), rng.integers(0, 100, (4, 2, 5)).astype(np.int32)
    out = DeviceTensor(ctx, (4, 5, 5), np.int32)
    da, db = dev(ctx, a), dev(ctx, b)
    ctx.call("rten_hip_copy_rows_b32", 4, 15, da.vp, 15, out.vp, 25)
    ctx.call("rten_hip_copy_rows_b32", 4, 10, db.vp, 10, C.c_void_p(out.ptr + 15 * 4), 25)
    np.testing.assert_array_equal(out.numpy(), np.concatenate([a, b], axis=1))


@pytest.mark.parametrize("m,k,n", [(16, 2304, 1000), (33, 4096, 4096), (8, 11008, 512), (64, 4096, 1000)])
def test_small_m_streaming_gemm_beyond_eight_depth_blocks(ctx, m, k, n):
    """gemm_f32_smallm_kernel is the automatic path for every one-batch GEMM with M <= 64, and its fold loop takes the parked depth blocks
    eight at a time -- K = 2304 (9 blocks), 4096 (16) and 11008 (43: LLM projections) exercise the second and later passes, which the round-5 tests
    (K <= 2048) never reached.  The oracle's blocked chain, bit for bit, for the explicit variant and the automatic choice."""
    from tests.test_gpu_round5 import _gemm
    rng = ref.XorShiftRng(m + k + n)
    a = rng.f32(m * k).reshape(m, k) - 0.5
    b = rng.f32(k * n).reshape(k, n) - 0.5
    bias = rng.f32(n) - 0.5
    want = ref.gemm_f32(a, b, bias=bias, bias_kind=ref.BIAS_PER_COL)
    bt = np.ascontiguousarray(b.T).T
    _bits(_gemm(ctx, a, bt, bias=bias, bias_kind=L.BIAS_PER_COL, variant=31), want)
    _bits(_gemm(ctx, a, b, bias=bias, bias_kind=L.BIAS_PER_COL), want)


def test_measurement_only_knobs_are_rejected_by_the_product_library(ctx):
    """rten_hip_set_gemm_order bit 3 (relaxed split-K: not the reference's order) exists in -DRTEN_ABLATION builds only; a plan file cannot switch the
    bit-exactness contract off.  rten_hip_tuning_restore puts every knob back even when one of the saved values is refused."""
    assert ctx.lib.rten_hip_set_gemm_order(ctx.h, 8) == L.ERR_INVALID_VALUE
    assert ctx.lib.rten_hip_set_gemm_order(ctx.h, 3) == 0
    saved = (C.c_int32 * 8)()
    ctx.call("rten_hip_tuning_save", saved)
    assert saved[3] == 3
    bad = (C.c_int32 * 8)(*saved)
    bad[3] = 8          # refused ...
    bad[7] = 1          # ... but the knobs after it are still restored
    assert ctx.lib.rten_hip_tuning_restore(ctx.h, bad) == L.ERR_INVALID_VALUE
    now = (C.c_int32 * 8)()
    ctx.call("rten_hip_tuning_save", now)
    assert now[7] == 1
    saved[3] = 0
    saved[7] = 0
    ctx.call("rten_hip_tuning_restore", saved)


@pytest.mark.parametrize("shape", [(2, 3, 128, 128, 64), (2, 2, 64, 128, 64), (2, 2, 40, 72, 64), (1, 2, 48, 200, 32)])
def test_sdpa_reads_one_shared_mask_row_out_of_an_expanded_mask(ctx, shape):
    """rten_hip_sdpa_f32 with mask_row_stride = 0 and mask_batch_stride = S * T: the mask operand is a [B, 1, S, T] tensor whose S rows are copies of one row
    (an exporter's expanded padding mask; the executor knows it from Tensor::uniform_dims) and only the first row of each batch item is read.  Same bits as the
    [B, 1, S, T] form and as the compact [B, 1, 1, T] form, on the 16-query kernel (first shape), the 32-query kernel, the general one-kernel form and the
    composed GEMM / softmax / GEMM path."""
    B, H, S, T, D = shape
    rng = ref.XorShiftRng(S + T + D)
    q = rng.f32(B * H * S * D).reshape(B, H, S, D) - 0.5
    k = rng.f32(B * H * T * D).reshape(B, H, T, D) - 0.5
    v = rng.f32(B * H * T * D).reshape(B, H, T, D) - 0.5
    row = np.where(rng.f32(B * T).reshape(B, 1, 1, T) > 0.3, 0.0, -3.4028234663852886e38).astype(np.float32)
    full = np.ascontiguousarray(np.broadcast_to(row, (B, 1, S, T)))
    qd, kd, vd = (DeviceTensor.from_numpy(ctx, a) for a in (q, k, v))
    strides = (H * S * D, S * D, D, H * T * D, T * D, D, H * T * D, T * D, D, H * S * D, S * D, D)
    scale = float(np.float32(1.0 / np.sqrt(D)))
    want = ref.sdpa(q, k, v, mask=row, scale=scale, lanes=16, flush_nan=False)
    outs = []
    for path in (0, 1):  # automatic (a one-kernel form where one covers the shape), composed only
        ctx.call("rten_hip_set_sdpa_path", path)
        try:
            for (m, mbs, mrs) in ((row, T, 0), (full, S * T, T), (full, S * T, 0)):
                out = DeviceTensor(ctx, (B, H, S, D), np.float32)
                d = L.SdpaDesc(B, H, S, T, D, D, *strides, mbs, mrs, scale, 0)
                md = DeviceTensor.from_numpy(ctx, m)
                ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, md.vp, out.vp)
                outs.append(out.numpy())
        finally:
            ctx.call("rten_hip_set_sdpa_path", 0)
    for got in outs:
        _bits(got, want)
    # a batch stride that is neither T nor S * T is still refused
    d = L.SdpaDesc(B, H, S, T, D, D, *strides, 2 * T, 0, scale, 0)
    out = DeviceTensor(ctx, (B, H, S, D), np.float32)
    md = DeviceTensor.from_numpy(ctx, full)
    rc = ctx.lib.rten_hip_sdpa_f32(ctx.h, C.byref(d), qd.vp, kd.vp, vd.vp, md.vp, out.vp)
    assert rc == (L.ERR_UNSUPPORTED if S != 2 else 0), rc


def test_co_run_measurement_on_borrowed_contexts():
    """rten_amd/workloads/corun.py (what tools/tune_corun.py sweeps with and what bench.py reports as `roofline.dominant_kernel.co_run`): the same layer on two
    streams at once, on contexts the caller keeps -- a positive time per launch for two plans of one layer, the networks' activations in place (the layer's output
    equals a plain launch of the same layer), and the borrowed contexts still usable afterwards."""
    from rten_amd.workloads.corun import CoRun
    ctxs = [L.Context(0), L.Context(0)]
    try:
        cr = CoRun(2, batch=2, plan={}, ctxs=ctxs)
        fams = cr.families()
        key = max(fams, key=lambda k: sum(cr.flops(l["name"]) for _, l in fams[k]))
        idx, l = fams[key][1]
        t_a = cr.measure(idx, [3, 0, 1, 0], reps=4, rounds=2)
        t_b = cr.measure(idx, [27, 2, 3, 0], reps=4, rounds=2)
        assert 1.0 < t_a < 1e5 and 1.0 < t_b < 1e5, (t_a, t_b)
        outs = []
        for net in cr.nets:  # both plans are bit-identical launch forms of the same convolution on the same inputs
            net.variants[l["name"]] = (3, 0, 1, 0)
            net._conv(l)
            net.ctx.sync()
            a = net._act(l["dst"]).numpy().copy()
            net.variants[l["name"]] = (27, 2, 3, 0)
            net._conv(l)
            net.ctx.sync()
            _bits(net._act(l["dst"]).numpy(), a)
            outs.append(a)
        assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0
        cr.close()
        x = dev(ctxs[0], np.arange(16, dtype=np.float32))  # the borrowed contexts live on
        assert np.array_equal(x.numpy(), np.arange(16, dtype=np.float32))
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("pm", [L.PAD_ZERO_POINT, L.PAD_RAW0_I8, L.PAD_RAW0_U8])
def test_few_channel_packed_int8_convolution_is_bit_exact(ctx, pm):
    """Convolutions with C <= 4 input channels take the PACKED staging form (int8_fast.hip, conv_geom: a 16-byte chunk = 16 / C kernel columns x C channels of one
    kernel row, per output column; the stem of ResNet-50 goes from K = 784 to 224): every channel count, kernel widths that fill a chunk exactly / partly / need
    several, strides 1-3, dilation, asymmetric padding on both axes, both signednesses of both operands with no / a scalar / a per-channel weight zero point on the
    plain path (the unused bytes of a chunk are zero on both operands, so every term of the zero-point algebra is exact), and DynamicQuantizeLinear's staged image +
    prepacked weights on the product's path -- against the oracle's ConvInteger under the three padding conventions."""
    rng = ref.XorShiftRng(pm + 5)
    cases = ((2, 3, 23, 20, 8, 7, 7, (3, 3, 3, 3), (2, 2), (1, 1)), (1, 1, 9, 30, 5, 3, 11, (1, 5, 1, 5), (1, 2), (1, 1)), (2, 4, 12, 13, 70, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1)),
             (3, 2, 8, 17, 6, 1, 9, (0, 2, 0, 3), (1, 3), (1, 1)), (1, 3, 32, 32, 64, 7, 7, (3, 3, 3, 3), (2, 2), (1, 1)), (2, 4, 10, 9, 4, 5, 2, (2, 0, 1, 1), (2, 1), (1, 1)),
             (1, 3, 6, 40, 3, 2, 6, (0, 0, 1, 0), (1, 4), (1, 1)), (2, 3, 14, 19, 6, 3, 5, (2, 4, 2, 4), (1, 1), (2, 2)), (1, 2, 9, 21, 5, 2, 3, (1, 3, 0, 3), (1, 2), (1, 3)))
    for (N, Cc, H, W, O, kh, kw, pads, strides, dil) in cases:
        oh = (H + pads[0] + pads[2] - dil[0] * (kh - 1) - 1) // strides[0] + 1
        ow = (W + pads[1] + pads[3] - dil[1] * (kw - 1) - 1) // strides[1] + 1
        for xdt, wdt in ((np.uint8, np.int8), (np.int8, np.uint8), (np.uint8, np.uint8), (np.int8, np.int8)):  # plain operands: the u8 / i8 -> packed image staging launch
            x = (rng.u8(N * Cc * H * W) if xdt == np.uint8 else rng.i8(N * Cc * H * W)).reshape(N, Cc, H, W)
            wn = O * Cc * kh * kw
            w = (rng.u8(wn, reduced=True) if wdt == np.uint8 else rng.i8(wn, reduced=True)).reshape(O, Cc, kh, kw)
            x_zp = np.array(rng.u8(1)[0] if xdt == np.uint8 else rng.i8(1)[0], xdt)
            op = ops.ConvInteger(dilations=dil, padding=list(pads), strides=strides, pad_mode=pm)
            for w_zp in (None, np.array(3, wdt), (rng.u8(O, reduced=True) if wdt == np.uint8 else rng.i8(O, reduced=True))):
                got = op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, x_zp), dev(ctx, w_zp) if w_zp is not None else None])[0].numpy()
                _bits(got, ref.conv2d_int8(x, w, x_zp=int(x_zp), w_zp=w_zp, pads=pads, strides=strides, dilations=dil, pad_mode=pm))
        if dil != (1, 1):
            continue
        # the product's path: DynamicQuantizeLinear writing the packed image, prepacked weights, ConvIntegerToFloat + bias + Relu
        w = rng.i8(O * Cc * kh * kw, reduced=True).reshape(O, Cc, kh, kw)
        xf = (rng.f32(N * Cc * H * W) - 0.4).reshape(N, Cc, H, W) * 3
        w_scale, bias = np.array([0.003], np.float32), rng.f32(O) - 0.5
        cd = L.Conv2dDesc(N, Cc, H, W, O, kh, kw, (C.c_int32 * 4)(*pads), strides[0], strides[1], 1, 1, 1, oh, ow)
        d = L.Conv2dInt8Desc(cd, 0, 1, 0, pm, 1, 1)
        staged = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d)),), np.uint8)
        packed = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_int8_packed_bytes(C.byref(d)),), np.uint8)
        xd, wd, bd, wsd = dev(ctx, xf), dev(ctx, w), dev(ctx, bias), dev(ctx, w_scale)
        xs, xz, sc = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8), DeviceTensor(ctx, (1,), np.float32)
        out = DeviceTensor(ctx, (N, O, oh, ow), np.float32)
        ctx.call("rten_hip_conv2d_int8_prepack", C.byref(d), wd.vp, packed.vp)
        ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), xd.vp, staged.vp, xs.vp, xz.vp, wsd.vp, sc.vp)
        ctx.call("rten_hip_conv2d_int8", C.byref(d), staged.vp, packed.vp, xz.vp, None, sc.vp, bd.vp, None, L.CONV_RELU, out.vp)
        q, s, z = ref.dynamic_quantize_linear(xf)
        assert xs.numpy()[0] == s and xz.numpy()[0] == z
        acc = ref.conv2d_int8(q, w, x_zp=int(z), pads=pads, strides=strides, pad_mode=pm)
        _bits(out.numpy(), ref.relu(ref.cast_scale(acc, np.float32(np.float32(s) * w_scale[0])) + bias[None, :, None, None]))
        # the packed image has fewer chunks per output pixel than the 16-channel-block form has taps, and is what staged_bytes sizes
        nch = -(-kw // (16 // Cc))
        assert nch < kw and ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d)) >= N * (H + pads[0] + pads[2]) * ow * nch * 16


def test_int8_workgroup_tile_knob_changes_time_only(ctx):
    """rten_hip_set_int8_tile (ABI v7: what a launch plan's per-layer entry for an int8 convolution step sets around the step): every tile computes the same integer
    sums -- ConvIntegerToFloat with bias / residual / Relu on a stage-2-like and a stage-0-like shape, and MatMulInteger; out-of-range values are refused and the
    previous value is reported."""
    rng = ref.XorShiftRng(99)
    prev = C.c_int32(123)
    assert ctx.lib.rten_hip_set_int8_tile(ctx.h, 4, None) == L.ERR_INVALID_VALUE and ctx.lib.rten_hip_set_int8_tile(ctx.h, -2, None) == L.ERR_INVALID_VALUE
    ctx.call("rten_hip_set_int8_tile", 2, C.byref(prev))
    assert prev.value == -1
    ctx.call("rten_hip_set_int8_tile", -1, C.byref(prev))
    assert prev.value == 2
    for (N, Cc, H, W, O, k, pad, stride) in ((2, 64, 14, 14, 130, 3, 1, 1), (3, 32, 20, 20, 64, 1, 0, 1), (1, 128, 9, 9, 256, 3, 1, 2)):
        x = rng.u8(N * Cc * H * W).reshape(N, Cc, H, W)
        w = rng.i8(O * Cc * k * k, reduced=True).reshape(O, Cc, k, k)
        x_zp, scale, bias = np.array(117, np.uint8), np.array(0.01, np.float32), rng.f32(O) - 0.5
        oh = (H + 2 * pad - k) // stride + 1
        res = (rng.f32(N * O * oh * oh) - 0.5).reshape(N, O, oh, oh)
        want = None
        for tile in (-1, 0, 1, 2, 3):
            ctx.call("rten_hip_set_int8_tile", tile, None)
            try:
                op = ops.ConvIntegerToFloat(ops.ConvInteger(padding=[pad] * 4, strides=(stride, stride)), fuse_relu=True)
                got = op.run(ctx, [dev(ctx, x), dev(ctx, w), dev(ctx, x_zp), None, dev(ctx, scale), dev(ctx, bias), dev(ctx, res)])[0].numpy()
            finally:
                ctx.call("rten_hip_set_int8_tile", -1, None)
            if want is None:
                acc = ref.conv2d_int8(x, w, x_zp=int(x_zp), pads=(pad,) * 4, strides=(stride, stride), pad_mode=ref.PAD_RAW0_I8)
                want = ref.relu(ref.cast_scale(acc, scale) + bias[None, :, None, None] + res)
            _bits(got, want)


@pytest.mark.gpu
def test_two_pointwise_convolutions_in_one_launch_are_bit_exact():
    """rten_hip_conv2d_f32_pair (ABI v8: a bottleneck block's expand layer -- bias, residual Add, Relu -- and the next block's reduce layer in ONE launch, the second
    reading the first's output from LDS): both outputs bit-identical to the oracle's two convolutions (src/ops/conv.rs:248-284) and to two rten_hip_conv2d_f32
    launches, on ragged column counts (pixels per image not a multiple of the workgroup's columns, a last tile past the end), with / without residual, bias and
    activations, M1 in {64, 128, 256}, M2 in {64, 128}; unsupported shapes are refused, not computed."""
    ctx = L.Context(0)
    try:
        rng = ref.XorShiftRng(4242)
        cases = ((2, 64, 14, 14, 256, 64, True, True, True, True), (3, 64, 10, 6, 128, 128, True, False, True, True), (1, 64, 28, 28, 256, 128, False, True, False, True),
                 (5, 64, 6, 6, 64, 64, True, True, True, False), (2, 64, 56, 56, 256, 64, True, True, True, True),
                 # without a residual fewer requests follow a weight chunk's DMA: the kernel's counted wait must not count on them (a first version did: a race that only
                 # showed under load) -- a launch with many workgroups per compute unit, several times
                 (4, 64, 56, 56, 256, 64, False, True, True, True), (4, 64, 56, 56, 256, 128, False, False, False, True))
        for (N, C1, H, W, M1, M2, res, relu1, bias, relu2) in cases:
            x = (rng.f32(N * C1 * H * W) - 0.5).reshape(N, C1, H, W)
            w1 = (rng.f32(M1 * C1) - 0.5).reshape(M1, C1, 1, 1) * 0.2
            w2 = (rng.f32(M2 * M1) - 0.5).reshape(M2, M1, 1, 1) * 0.1
            b1 = rng.f32(M1) - 0.5 if bias else None
            b2 = rng.f32(M2) - 0.5 if bias else None
            r = (rng.f32(N * M1 * H * W) - 0.5).reshape(N, M1, H, W) if res else None
            want1 = ref.conv2d_f32(x, w1, b1, residual=r, relu=relu1)
            want2 = ref.conv2d_f32(want1, w2, b2, relu=relu2)
            d1 = L.Conv2dDesc(N, C1, H, W, M1, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, H, W)
            d2 = L.Conv2dDesc(N, M1, H, W, M2, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, H, W)
            assert ctx.lib.rten_hip_conv2d_f32_pair_supported(C.byref(d1), C.byref(d2)) == 1
            xd, w1d, w2d = dev(ctx, x), dev(ctx, w1), dev(ctx, w2)
            p1 = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d1)) // 4,), np.float32)
            p2 = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d2)) // 4,), np.float32)
            ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d1), w1d.vp, p1.vp)
            ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d2), w2d.vp, p2.vp)
            b1d, b2d, rd = (dev(ctx, b1) if bias else None), (dev(ctx, b2) if bias else None), (dev(ctx, r) if res else None)
            y1, y2 = DeviceTensor(ctx, (N, M1, H, W), np.float32), DeviceTensor(ctx, (N, M2, H, W), np.float32)
            f1 = (L.CONV_RELU if relu1 else 0) | (L.CONV_RESIDUAL if res else 0)
            f2 = L.CONV_RELU if relu2 else 0
            ctx.call("rten_hip_conv2d_f32_pair", C.byref(d1), xd.vp, p1.vp, b1d.vp if bias else None, rd.vp if res else None, f1, y1.vp,
                     C.byref(d2), p2.vp, b2d.vp if bias else None, f2, y2.vp)
            _bits(y1.numpy(), want1)
            _bits(y2.numpy(), want2)
            for _ in range(4 if not res else 1):  # (again, back to back: every launch gives the same bits)
                ctx.call("rten_hip_conv2d_f32_pair", C.byref(d1), xd.vp, p1.vp, b1d.vp if bias else None, rd.vp if res else None, f1, y1.vp,
                         C.byref(d2), p2.vp, b2d.vp if bias else None, f2, y2.vp)
            _bits(y1.numpy(), want1)
            _bits(y2.numpy(), want2)
            # ... and the two separate launches (what the pair replaces)
            s1, s2 = DeviceTensor(ctx, (N, M1, H, W), np.float32), DeviceTensor(ctx, (N, M2, H, W), np.float32)
            ctx.call("rten_hip_conv2d_f32", C.byref(d1), xd.vp, p1.vp, 1, b1d.vp if bias else None, rd.vp if res else None, f1, s1.vp)
            ctx.call("rten_hip_conv2d_f32", C.byref(d2), s1.vp, p2.vp, 1, b2d.vp if bias else None, None, f2, s2.vp)
            _bits(y1.numpy(), s1.numpy())
            _bits(y2.numpy(), s2.numpy())
        # shapes without a one-launch form
        for (C1, M1, M2, k) in ((128, 256, 64, 1), (64, 512, 64, 1), (64, 256, 256, 1), (64, 256, 64, 3)):
            d1 = L.Conv2dDesc(1, C1, 8, 8, M1, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, 8, 8)
            d2 = L.Conv2dDesc(1, M1, 8, 8, M2, k, k, (C.c_int32 * 4)(k // 2, k // 2, k // 2, k // 2), 1, 1, 1, 1, 1, 8, 8)
            assert ctx.lib.rten_hip_conv2d_f32_pair_supported(C.byref(d1), C.byref(d2)) == 0
            a = DeviceTensor(ctx, (1 << 16,), np.float32)
            assert ctx.lib.rten_hip_conv2d_f32_pair(ctx.h, C.byref(d1), a.vp, a.vp, None, None, 0, a.vp, C.byref(d2), a.vp, None, 0, a.vp) == L.ERR_UNSUPPORTED
    finally:
        ctx.close()


@pytest.mark.gpu
def test_stem_convolution_direct_form_is_bit_exact(ctx):
    """GEMM variant 32 (gemm_f32_stem.hip): 3-channel 7x7 stride-2 convolutions with prepacked weights as a direct implicit GEMM over an image patch in LDS --
    bit-identical to the oracle (src/ops/conv.rs:124-365) and to variant 3 on ResNet's stem shape, ragged outputs (tiles past the image's edge), asymmetric and zero
    padding, fewer than 64 / odd output channel counts, with and without bias / Relu; shapes outside its form run as variant 3."""
    rng = ref.XorShiftRng(777)
    cases = ((2, 224, 224, 64, (3, 3, 3, 3), True, True), (1, 50, 45, 64, (3, 3, 3, 3), True, False), (3, 33, 70, 24, (2, 3, 3, 2), False, True), (1, 21, 19, 7, (0, 0, 0, 0), True, True),
             (2, 64, 64, 61, (3, 3, 2, 2), True, True))
    for (N, H, W, O, pads, bias, relu) in cases:
        oh, ow = (H + pads[0] + pads[2] - 7) // 2 + 1, (W + pads[1] + pads[3] - 7) // 2 + 1
        x = (rng.f32(N * 3 * H * W) - 0.5).reshape(N, 3, H, W)
        w = (rng.f32(O * 3 * 49) - 0.5).reshape(O, 3, 7, 7) * 0.2
        b = rng.f32(O) - 0.5 if bias else None
        want = ref.conv2d_f32(x, w, b, pads=pads, strides=(2, 2), relu=relu)
        d = L.Conv2dDesc(N, 3, H, W, O, 7, 7, (C.c_int32 * 4)(*pads), 2, 2, 1, 1, 1, oh, ow)
        xd, wd, bd = dev(ctx, x), dev(ctx, w), (dev(ctx, b) if bias else None)
        pk = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d)) // 4,), np.float32)
        ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d), wd.vp, pk.vp)
        outs = []
        for variant in (32, 3):
            ctx.set_gemm_variant(variant)
            y = DeviceTensor(ctx, (N, O, oh, ow), np.float32)
            ctx.call("rten_hip_conv2d_f32", C.byref(d), xd.vp, pk.vp, 1, bd.vp if bias else None, None, L.CONV_RELU if relu else 0, y.vp)
            outs.append(y.numpy())
        ctx.set_gemm_variant(-1)
        _bits(outs[0], want)
        _bits(outs[1], want)
    # not its form (5x5 kernel; four channels; unpacked weights): variant 32 runs them as variant 3
    for (Cc, k, packed) in ((3, 5, True), (4, 7, True), (3, 7, False)):
        x = (rng.f32(1 * Cc * 20 * 20) - 0.5).reshape(1, Cc, 20, 20)
        w = (rng.f32(8 * Cc * k * k) - 0.5).reshape(8, Cc, k, k)
        oh = (20 + 2 * (k // 2) - k) // 2 + 1
        want = ref.conv2d_f32(x, w, None, pads=(k // 2,) * 4, strides=(2, 2))
        d = L.Conv2dDesc(1, Cc, 20, 20, 8, k, k, (C.c_int32 * 4)(*((k // 2,) * 4)), 2, 2, 1, 1, 1, oh, oh)
        xd, wd = dev(ctx, x), dev(ctx, w)
        pk = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d)) // 4,), np.float32)
        ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d), wd.vp, pk.vp)
        ctx.set_gemm_variant(32)
        y = DeviceTensor(ctx, (1, 8, oh, oh), np.float32)
        ctx.call("rten_hip_conv2d_f32", C.byref(d), xd.vp, pk.vp if packed else wd.vp, 1 if packed else 0, None, None, 0, y.vp)
        ctx.set_gemm_variant(-1)
        _bits(y.numpy(), want)



@pytest.mark.gpu
def test_pair_with_the_shortcut_convolution_inside_is_bit_exact():
    """rten_hip_conv2d_f32_pair_shortcut (ABI v8): a stage's FIRST bottleneck block -- the expand layer's residual is the shortcut layer, a 64-channel pointwise
    convolution of the block's input, computed in the same launch chunk by chunk instead of being written and read back -- with the next block's reduce layer behind
    it.  Both outputs bit-identical to the oracle's three convolutions (shortcut, expand + Add + Relu, reduce) on ragged column counts, with / without biases, run
    several times (every launch the same bits); shapes outside the form are refused."""
    ctx = L.Context(0)
    try:
        rng = ref.XorShiftRng(9191)
        for (N, H, W, M1, bias, relu1, relu2) in ((2, 14, 14, 256, True, True, True), (3, 10, 6, 128, False, True, False), (4, 56, 56, 256, True, True, True), (1, 6, 6, 64, True, False, True)):
            x = (rng.f32(N * 64 * H * W) - 0.5).reshape(N, 64, H, W)
            xd = (rng.f32(N * 64 * H * W) - 0.5).reshape(N, 64, H, W)
            w1 = (rng.f32(M1 * 64) - 0.5).reshape(M1, 64, 1, 1) * 0.2
            wds = (rng.f32(M1 * 64) - 0.5).reshape(M1, 64, 1, 1) * 0.2
            w2 = (rng.f32(64 * M1) - 0.5).reshape(64, M1, 1, 1) * 0.1
            b1, bds, b2 = ((rng.f32(M1) - 0.5, rng.f32(M1) - 0.5, rng.f32(64) - 0.5) if bias else (None, None, None))
            short = ref.conv2d_f32(xd, wds, bds)
            want1 = ref.conv2d_f32(x, w1, b1, residual=short, relu=relu1)
            want2 = ref.conv2d_f32(want1, w2, b2, relu=relu2)
            mk = lambda c, o: L.Conv2dDesc(N, c, H, W, o, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, H, W)  # noqa: E731
            d1, dsd, d2 = mk(64, M1), mk(64, M1), mk(M1, 64)
            assert ctx.lib.rten_hip_conv2d_f32_pair_shortcut_supported(C.byref(d1), C.byref(dsd), C.byref(d2)) == 1
            dv = [dev(ctx, a) for a in (x, xd, w1, wds, w2)]
            pk = []
            for d, wt in ((d1, dv[2]), (dsd, dv[3]), (d2, dv[4])):
                t = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d)) // 4,), np.float32)
                ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d), wt.vp, t.vp)
                pk.append(t)
            bv = [dev(ctx, a) if bias else None for a in (b1, bds, b2)]
            y1, y2 = DeviceTensor(ctx, (N, M1, H, W), np.float32), DeviceTensor(ctx, (N, 64, H, W), np.float32)
            for _ in range(3):
                ctx.call("rten_hip_conv2d_f32_pair_shortcut", C.byref(d1), dv[0].vp, pk[0].vp, bv[0].vp if bias else None, C.byref(dsd), dv[1].vp, pk[1].vp, bv[1].vp if bias else None,
                         L.CONV_RELU if relu1 else 0, y1.vp, C.byref(d2), pk[2].vp, bv[2].vp if bias else None, L.CONV_RELU if relu2 else 0, y2.vp)
                _bits(y1.numpy(), want1)
                _bits(y2.numpy(), want2)
        d1, dsd, d2 = mk(64, 256), mk(128, 256), mk(256, 64)  # a shortcut of 128 channels: no form
        assert ctx.lib.rten_hip_conv2d_f32_pair_shortcut_supported(C.byref(d1), C.byref(dsd), C.byref(d2)) == 0
        d2 = mk(256, 128)  # ... nor a reduce layer of 128 channels
        assert ctx.lib.rten_hip_conv2d_f32_pair_shortcut_supported(C.byref(d1), C.byref(mk(64, 256)), C.byref(d2)) == 0
    finally:
        ctx.close()
