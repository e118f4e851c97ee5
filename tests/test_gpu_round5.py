"""Round-5 kernels against the oracle: the streaming 3x3 / stride-2 max-pool (16-byte row loads + the left neighbour's lane, statistics folded per
workgroup), and the executor-level changes that move bits through other launches (dropped all-zero weight zero points)."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


def _bits(a, b):
    assert a.shape == b.shape
    same = (a.view(np.int32) == b.view(np.int32))
    assert same.all(), f"{(~same).sum()} of {same.size} differ, first at {tuple(np.argwhere(~same)[0])}"


@pytest.mark.parametrize("shape,pads", [((2, 3, 8, 8), (1, 1, 1, 1)), ((1, 5, 9, 12), (1, 1, 0, 1)), ((3, 2, 30, 20), (1, 1, 1, 0)), ((1, 70, 14, 16), (1, 1, 1, 1)),
                                         ((2, 64, 112, 112), (1, 1, 1, 1)), ((1, 1, 7, 4), (1, 1, 1, 1)), ((5, 7, 23, 36), (1, 1, 0, 0))])
def test_streaming_max_pool_bits_and_statistics(ctx, shape, pads):
    """Geometries that take maxpool3x3s2_stream_kernel (W % 4 == 0, out_w == W / 2, one leading padding row and column): ragged row groups (out_h not
    a multiple of 4), rows cut by the bottom edge with and without padding, one 16-byte group per row, thread counts that do not fill the last
    workgroup, planes that start mid-wave (lane 0 fetches its own left neighbour), NaN / -inf inputs (the sign of a zero maximum of +0 and -0 is the host libm's choice in the oracle: not tested) -- bit-identical to the oracle; and the
    statistics block equals the min / max of the result."""
    n, c, h, w = shape
    rng = ref.XorShiftRng(7 + h * w)
    x = (rng.f32(n * c * h * w).reshape(shape) * 4.0 - 2.0).astype(np.float32)
    x.reshape(-1)[5::193] = np.nan
    x.reshape(-1)[11::211] = -np.inf
    oh = (h + pads[0] + pads[2] - 3) // 2 + 1
    ow = (w + pads[1] + pads[3] - 3) // 2 + 1
    assert ow * 2 == w  # (the streaming form's condition: otherwise this test would exercise the round-4 kernel)
    want = ref.max_pool(x, (3, 3), (2, 2), pads)
    pd = L.Pool2dDesc(n, c, h, w, 3, 3, 2, 2, (C.c_int32 * 4)(*pads), oh, ow, 0)
    xd = DeviceTensor.from_numpy(ctx, x)
    y_a, y_b = DeviceTensor(ctx, want.shape, np.float32), DeviceTensor(ctx, want.shape, np.float32)
    st = DeviceTensor(ctx, (ctx.lib.rten_hip_minmax_stats_bytes(),), np.uint8)
    ctx.call("rten_hip_minmax_stats_reset", st.vp, 1)
    ctx.call("rten_hip_max_pool2d_f32", C.byref(pd), xd.vp, y_a.vp)
    ctx.call("rten_hip_max_pool2d_f32_stats", C.byref(pd), xd.vp, y_b.vp, st.vp)
    ctx.sync()
    _bits(y_a.numpy(), want)
    _bits(y_b.numpy(), want)
    # the statistics as ordered uints: 256 minima then 256 maxima (csrc/quantize.h); NaNs never enter (fminf / fmaxf drop them)
    raw = st.numpy().view(np.uint32)

    def ord2f(u):
        u = np.where(u & 0x80000000, u & 0x7fffffff, ~u).astype(np.uint32)
        return u.view(np.float32)
    mins, maxs = ord2f(raw[:256][raw[:256] != 0xffffffff]), ord2f(raw[256:512][raw[256:512] != 0])
    finite = want[~np.isnan(want)]
    assert mins.min() == finite.min() and maxs.max() == finite.max()


# ------------------------------------------------------------------------------------------ ADVICE round 4, as tests
def _small_cnn_case(batch=6):
    from rten_amd import onnx_writer as ow
    model_bytes, w = ow.small_cnn_f32()
    x = np.random.default_rng(3).standard_normal((batch, 3, 16, 16)).astype(np.float32)
    a = ref.conv2d_f32(x, w["c1"][0], w["c1"][1], pads=(1, 1, 1, 1), strides=(2, 2), relu=True)
    p = ref.max_pool(a, (2, 2), (2, 2))
    s = ref.conv2d_f32(p, w["c2"][0], w["c2"][1], residual=p, relu=True)
    g = ref.global_average_pool(s).reshape(batch, -1)
    want = ref.gemm_f32(g, w["fc"][0].T, c=np.broadcast_to(w["fc"][1], (batch, 5)).astype(np.float32), alpha=1.0, beta=1.0)  # (as tests/test_graph_executor.py)
    return model_bytes, x, want


def test_model_abi_reports_why_and_replans(ctx):
    """Load errors carry text (there is no model object to ask), a keyed plan without the chain's sub-batch size is an error, a plan that matches no step
    a warning; rten_hip_model_plan_json / _set_plan round trip and re-capture with the same bits; a scalar-output / dim-0-changing graph is refused for
    chains > 1 instead of writing past its buffer (ADVICE round 4, low + medium 1)."""
    import json
    from rten_amd import onnx_writer as ow
    model_bytes, x, want = _small_cnn_case()
    with pytest.raises(L.HipError) as e:
        L.Model(ctx, b"\x08\x01garbage", None, 1)
    assert "model_load" in str(e.value) and len(str(e.value)) > 12, str(e.value)
    with pytest.raises(L.HipError) as e:
        L.Model(ctx, model_bytes, "{not json", 1)
    assert "plan file" in str(e.value), str(e.value)
    # keyed plan that does not name the chain's sub-batch size (6 rows as 2 chains of 3): error at prepare, with text
    m = L.Model(ctx, model_bytes, json.dumps({"8": {"c1": [3, 0, 1, 0]}}), 2)
    try:
        m.bind_input("x", x.shape)
        with pytest.raises(L.HipError) as e:
            m.prepare()
        assert "keyed by sub-batch size" in str(e.value), str(e.value)
    finally:
        m.close()
    # a plan that names no step of this graph: loads, prepares, warns
    m = L.Model(ctx, model_bytes, json.dumps({"no_such_step": [3, 0, 1, 0]}), 1)
    try:
        p = m.bind_input("x", x.shape)
        m.prepare()
        assert "matched no step" in m.warning, m.warning
        xt = DeviceTensor(ctx, x.shape, np.float32, ptr=p, keepalive=m)
        xt.upload(x)
        # plan export, re-plan (every conv AND the classifier Gemm on other variants), re-capture: same bits
        outs = []
        for plan in (None, {"c1": [3, 0, 1, 1], "c2": [27, 0, 1, 0], "fc": [2, 3, 1, 0]}):
            if plan is not None:
                m.set_plan(json.dumps(plan))
                m.prepare()
                ran = json.loads(m.plan_json())
                assert ran[str(x.shape[0])]["c1"] == [3, 0, 1, 1] and ran[str(x.shape[0])]["fc"] == [2, 3, 1, 0], ran
                assert m.planned_steps == 3 and m.warning == ""
            m.run(inputs_written_on_caller_stream=True)
            m.sync()
            optr, oshape = m.output(0)
            outs.append(DeviceTensor(ctx, oshape, np.float32, ptr=optr, keepalive=m).numpy())
        _bits(outs[0], want)
        _bits(outs[1], want)
    finally:
        m.close()


def test_executor_leaves_a_borrowed_context_as_it_found_it(ctx):
    """Chain 0 of a model runs on the CALLER's context: after load / prepare / run / profile the caller's tuning knobs are what the caller set, not the
    defaults."""
    import json
    model_bytes, x, want = _small_cnn_case(batch=4)
    ctx.call("rten_hip_set_gemm_variant_override", 2)
    ctx.call("rten_hip_set_gemm_split", 1, 3)
    ctx.call("rten_hip_set_gemm_order", 1)
    ctx.call("rten_hip_set_gemv_order", 0, 7)
    before = (C.c_int32 * 8)()
    ctx.call("rten_hip_tuning_save", before)
    try:
        m = L.Model(ctx, model_bytes, json.dumps({"c1": [3, 0, 1, 0], "c2": [27, 0, 1, 0]}), 3)  # 4 rows as chains of 2 + 1 + 1: lone-row chains flip the gemv order
        try:
            p = m.bind_input("x", x.shape)
            m.prepare()
            DeviceTensor(ctx, x.shape, np.float32, ptr=p, keepalive=m).upload(x)
            m.run(inputs_written_on_caller_stream=True)
            m.sync()
            m.profile_pass(1)
            after = (C.c_int32 * 8)()
            ctx.call("rten_hip_tuning_save", after)
            assert list(after) == list(before), (list(before), list(after))
        finally:
            m.close()
    finally:
        ctx.call("rten_hip_set_gemm_variant_override", -1)
        ctx.call("rten_hip_set_gemm_split", 3, 1)
        ctx.call("rten_hip_set_gemm_order", 0)
        ctx.call("rten_hip_set_gemv_order", 1, 0)


def test_scratch_a_live_graph_replays_from_is_not_freed(ctx):
    """A captured hipGraph holds the context's auxiliary scratch pointer (the composed attention path keeps its score tensor there).  A later eager call
    on the same context that needs a larger buffer must not free the one the graph replays from: the replay still gives the
    first result."""
    rng = ref.XorShiftRng(77)

    def case(B, H, S, T):
        q = rng.f32(B * H * S * 64).reshape(B, H, S, 64) - 0.5
        k = rng.f32(B * H * T * 64).reshape(B, H, T, 64) - 0.5
        v = rng.f32(B * H * T * 64).reshape(B, H, T, 64) - 0.5
        d = L.SdpaDesc(B, H, S, T, 64, 64, H * S * 64, S * 64, 64, H * T * 64, T * 64, 64, H * T * 64, T * 64, 64, H * S * 64, S * 64, 64, 0, 0, 0.125, 0)
        return d, [DeviceTensor.from_numpy(ctx, t) for t in (q, k, v)], DeviceTensor(ctx, (B, H, S, 64), np.float32), ref.sdpa(q, k, v, scale=0.125, lanes=16, flush_nan=False)
    ctx.call("rten_hip_set_sdpa_path", 1)  # the composed path: GEMM -> softmax -> GEMM through the auxiliary scratch
    try:
        d1, (q1, k1, v1), o1, want1 = case(1, 2, 40, 200)
        run1 = lambda: ctx.call("rten_hip_sdpa_f32", C.byref(d1), q1.vp, k1.vp, v1.vp, None, o1.vp)
        run1()
        ctx.sync()
        _bits(o1.numpy(), want1)
        ctx.graph_begin()
        run1()
        g = ctx.graph_end()
        try:
            d2, (q2, k2, v2), o2, want2 = case(8, 16, 384, 384)  # 8 * 16 * 384 * 384 * 4 B = 75 MB of scores: larger than the first buffer
            ctx.call("rten_hip_sdpa_f32", C.byref(d2), q2.vp, k2.vp, v2.vp, None, o2.vp)
            ctx.sync()
            _bits(o2.numpy(), want2)
            for _ in range(3):
                o1.upload(np.zeros(o1.shape, np.float32))
                ctx.graph_launch(g)
                ctx.sync()
                _bits(o1.numpy(), want1)
        finally:
            ctx.graph_destroy(g)
    finally:
        ctx.call("rten_hip_set_sdpa_path", 0)


def _gemm(ctx, a, b, c=None, alpha=1.0, beta=0.0, bias=None, bias_kind=0, act=0, variant=-1):
    """a [m, k], b [k, n]; an operand that is not C-contiguous is uploaded as its (contiguous) transpose and described by strides."""
    m, k = a.shape
    n = b.shape[1]
    out = DeviceTensor.from_numpy(ctx, c if c is not None else np.full((m, n), np.nan, np.float32))
    if a.flags.c_contiguous:
        ad = DeviceTensor.from_numpy(ctx, a); a_rs, a_cs = k, 1
    else:
        ad = DeviceTensor.from_numpy(ctx, np.ascontiguousarray(a.T)); a_rs, a_cs = 1, m
    if b.flags.c_contiguous:
        bd = DeviceTensor.from_numpy(ctx, b); b_rs, b_cs = n, 1
    else:
        bd = DeviceTensor.from_numpy(ctx, np.ascontiguousarray(b.T)); b_rs, b_cs = 1, k
    biasd = DeviceTensor.from_numpy(ctx, bias) if bias is not None else None
    d = L.gemm_desc(m, n, k, a_rs, a_cs, b_rs, b_cs, n, alpha=alpha, beta=beta, bias_kind=bias_kind, act=act)
    ctx.set_gemm_variant(variant)
    try:
        ctx.call("rten_hip_gemm_f32", C.byref(d), ad.vp, bd.vp, biasd.vp if biasd else None, out.vp)
    finally:
        ctx.set_gemm_variant(-1)
    return out.numpy()


@pytest.mark.parametrize("m", [2, 5, 16, 17, 32, 33, 48, 64])
def test_small_m_streaming_gemm_is_the_blocked_chain(ctx, m):
    """gemm_f32_smallm_kernel (M <= 64, one batch; variant 31 and the automatic choice): one wave per 16x16 block per depth block, block sums parked and
    folded by the last arrival in depth-block order -- the bits of rten-gemm's blocked chain (lib.rs:1008-1013, 1221-1255) for one / two / four row
    blocks, ragged rows and columns, one depth block / a ragged last one / the classifier's eight, 16-byte and scalar loaders of either operand
    (all four transposition combinations), alpha / beta / both bias kinds / fused activations."""
    rng = ref.XorShiftRng(900 + m)
    for k in (3, 64, 256, 258, 700, 2048):
        for n in (1, 16, 33, 100, 1000):
            if k * n > 300000 and (m not in (32, 33)):
                continue
            a = rng.f32(m * k).reshape(m, k) - 0.5
            b = rng.f32(k * n).reshape(k, n) - 0.5
            want = ref.gemm_f32(a, b)
            at, bt = np.ascontiguousarray(a.T).T, np.ascontiguousarray(b.T).T
            for aa, bb in ((a, b), (a, bt), (at, b), (at, bt)):
                _bits(_gemm(ctx, aa, bb, variant=31), want)
            _bits(_gemm(ctx, a, bt), want)  # what -1 picks
    k, n = 700, 100
    a = rng.f32(m * k).reshape(m, k) - 0.5
    b = rng.f32(k * n).reshape(k, n) - 0.5
    bt = np.ascontiguousarray(b.T).T
    c = rng.f32(m * n).reshape(m, n)
    br, bc = rng.f32(m), rng.f32(n)
    for alpha, beta in ((1.0, 1.0), (0.5, 0.0), (0.5, 2.0)):
        _bits(_gemm(ctx, a, bt, c=c, alpha=alpha, beta=beta, variant=31), ref.gemm_f32(a, b, c=c, alpha=alpha, beta=beta))
    _bits(_gemm(ctx, a, bt, bias=br, bias_kind=L.BIAS_PER_ROW, variant=31), ref.gemm_f32(a, b, bias=br, bias_kind=ref.BIAS_PER_ROW))
    _bits(_gemm(ctx, a, bt, bias=bc, bias_kind=L.BIAS_PER_COL, act=L.ACT_RELU, variant=31), ref.relu(ref.gemm_f32(a, b, bias=bc, bias_kind=ref.BIAS_PER_COL)))
    _bits(_gemm(ctx, a, bt, c=c, alpha=0.5, beta=2.0, bias=bc, bias_kind=L.BIAS_PER_COL, act=L.ACT_GELU, variant=31),
          ref.gelu(ref.gemm_f32(a, b, c=c, alpha=0.5, beta=2.0, bias=bc, bias_kind=ref.BIAS_PER_COL)))
    assert not np.isnan(_gemm(ctx, a, bt, variant=31)).any()  # beta == 0 never reads C (the output starts as NaN)


@pytest.mark.parametrize("layout", ["bhsd", "bshd"])
def test_sixteen_query_attention_kernel_bits(ctx, layout):
    """sdpa_fused16_kernel (head 64, 128 keys, s a multiple of 64, no mask or a [B, 1, 1, T] mask): 16-query waves on v_mfma_f32_16x16x4_f32, key rows of a
    score block permuted so that the probabilities feed PV^T from the registers they were computed in -- against the oracle's sdpa
    (src/ops/attention.rs:518-626 order: d-ordered scores, * scale, + mask, 16-lane ordered softmax sums, key-ordered PV), for one to four
    64-query tiles, both memory layouts (heads contiguous, and BERT's [B, S, H * 64] projections), -inf masks, a fully masked batch item with
    and without the NaN flush."""
    rng = ref.XorShiftRng(77)
    T = 128
    for (B, H, S) in ((1, 1, 64), (2, 3, 128), (1, 2, 192), (2, 12, 256)):
        q = rng.f32(B * H * S * 64).reshape(B, H, S, 64) - 0.5
        k = rng.f32(B * H * T * 64).reshape(B, H, T, 64) - 0.5
        v = rng.f32(B * H * T * 64).reshape(B, H, T, 64) - 0.5
        m1 = np.where(rng.f32(B * T).reshape(B, 1, 1, T) > 0.3, 0.0, -np.inf).astype(np.float32)
        m1[0, 0, 0, :] = -np.inf
        m0 = ((rng.f32(B * T).reshape(B, 1, 1, T) - 0.5) * 6).astype(np.float32)
        for m in (None, m0, m1):
            for flush in (True, False):
                if layout == "bhsd":
                    qd, kd, vd = (DeviceTensor.from_numpy(ctx, a) for a in (q, k, v))
                    out = DeviceTensor(ctx, (B, H, S, 64), np.float32)
                    strides = (H * S * 64, S * 64, 64, H * T * 64, T * 64, 64, H * T * 64, T * 64, 64, H * S * 64, S * 64, 64)
                else:
                    qd, kd, vd = (DeviceTensor.from_numpy(ctx, np.ascontiguousarray(a.transpose(0, 2, 1, 3))) for a in (q, k, v))
                    out = DeviceTensor(ctx, (B, S, H, 64), np.float32)
                    strides = (S * H * 64, 64, H * 64, T * H * 64, 64, H * 64, T * H * 64, 64, H * 64, S * H * 64, 64, H * 64)
                d = L.SdpaDesc(B, H, S, T, 64, 64, *strides, 0 if m is None else T, 0, 0.125, 1 if flush else 0)
                md = DeviceTensor.from_numpy(ctx, m) if m is not None else None
                ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, md.vp if md is not None else None, out.vp)
                got = out.numpy() if layout == "bhsd" else out.numpy().transpose(0, 2, 1, 3)
                want = ref.sdpa(q, k, v, mask=m, scale=0.125, lanes=16, flush_nan=flush)
                assert got.shape == want.shape
                gi, wi = np.ascontiguousarray(got).view(np.int32), want.view(np.int32)
                nan_both = np.isnan(got) & np.isnan(want)  # (NaN payloads are not part of the contract)
                assert ((gi == wi) | nan_both).all(), (B, H, S, None if m is None else "mask", flush, int(((gi != wi) & ~nan_both).sum()))


@pytest.mark.parametrize("shape,pt,pb", [((2, 3, 8, 8), 1, 1), ((1, 5, 9, 12), 0, 1), ((3, 2, 30, 20), 1, 0), ((1, 70, 14, 16), 1, 1), ((2, 144, 56, 56), 1, 1),
                                          ((1, 1, 7, 4), 1, 1), ((5, 7, 23, 36), 0, 0), ((1, 2, 3, 260), 2, 2)])
def test_streaming_depthwise_3x3_bits(ctx, shape, pt, pb):
    """Geometries that take depthwise3x3s1_stream_kernel (3 x 3, stride 1, one padding column either side, W % 4 == 0): ragged row groups, any top / bottom
    padding, one 16-byte group per row, planes that start mid-wave (the first / last lane of a wave fetches its own neighbour column), rows longer than a
    wave, thread counts that do not fill the last workgroup; with / without bias, residual Add and Relu, prepacked and plain weights -- bit-identical to the
    reference's depthwise sequence (conv/depthwise.rs:95-146: accumulator = bias, one rounded multiply and one add per in-bounds tap in (k_y, k_x) order)."""
    from tests.test_gpu_parity import gpu_conv
    n, c, h, w = shape
    rng = np.random.default_rng(11 + h * w + c)
    x = rng.random(shape, dtype=np.float32) - 0.5
    wt = rng.random((c, 1, 3, 3), dtype=np.float32) - 0.5
    pads = (pt, 1, pb, 1)
    for bias, with_res, relu, prepack in ((None, False, False, False), (rng.random(c, dtype=np.float32) - 0.5, True, True, True), (rng.random(c, dtype=np.float32) - 0.5, False, True, False)):
        want = ref.conv2d_f32(x, wt, bias, pads=pads, strides=(1, 1), dilations=(1, 1), groups=c)
        res = (rng.random(want.shape, dtype=np.float32) - 0.5) if with_res else None
        want = ref.conv2d_f32(x, wt, bias, pads=pads, strides=(1, 1), dilations=(1, 1), groups=c, residual=res, relu=relu)
        _bits(gpu_conv(ctx, x, wt, bias, pads, (1, 1), (1, 1), c, residual=res, relu=relu, prepack=prepack), want)


@pytest.mark.parametrize("case", [
    # N, C, H, W, O_g, kh, kw, pads, stride, groups, output_padding
    (2, 8, 5, 7, 6, 4, 4, (1, 1, 1, 1), 2, 1, (0, 0)),       # the common 4x4 / 2 upsampler, ragged channel block (6 of 16)
    (1, 64, 28, 28, 32, 4, 4, (1, 1, 1, 1), 2, 1, (0, 0)),   # the microbenchmark's layer (one image)
    (2, 4, 9, 33, 20, 3, 3, (0, 1, 2, 0), 2, 1, (1, 0)),      # odd kernel at stride 2: row / column classes with different tap counts, output padding, > 16 s columns
    (1, 12, 6, 6, 40, 2, 2, (0, 0, 0, 0), 2, 3, (0, 0)),      # kernel == stride (no overlap), three groups, 40 channels per group (three 16-blocks of four)
    (3, 16, 7, 5, 16, 3, 3, (1, 1, 1, 1), 1, 2, (0, 0)),      # stride 1 (a flipped convolution), two groups
    (1, 4, 3, 40, 5, 5, 3, (2, 0, 1, 1), 1, 1, (0, 0)),       # stride 1, 5x3 window, rows wider than one 16-column block
    (2, 8, 4, 4, 64, 1, 1, (0, 0, 0, 0), 2, 1, (1, 1)),       # 1x1 kernel at stride 2: most outputs receive no tap (bias only)
])
def test_fused_conv_transpose_bits(ctx, case):
    """conv_transpose_fused_kernel (no dilation, stride 1 or 2, C_g a multiple of 4): per output element bias, then for (k_y, k_x) ascending the c-ordered chain
    of the tap that lands on it, added separately -- the reference's GEMM + col2im (conv_transpose.rs:80-142, 226-412) without the column matrix; with and
    without bias."""
    from rten_amd import ops
    n, c, h, w, og, kh, kw, pads, s, groups, opad = case
    rng = np.random.default_rng(5 + h * w + c + og)
    x = rng.random((n, c, h, w), dtype=np.float32) - 0.5
    wt = rng.random((c, og, kh, kw), dtype=np.float32) - 0.5
    b = rng.random(og * groups, dtype=np.float32) - 0.5
    op = ops.ConvTranspose(padding=list(pads), groups=groups, strides=[s, s], dilations=[1, 1], output_padding=list(opad))
    for bias in (b, None):
        ins = [DeviceTensor.from_numpy(ctx, x), DeviceTensor.from_numpy(ctx, wt)] + ([DeviceTensor.from_numpy(ctx, bias)] if bias is not None else [])
        _bits(op.run(ctx, ins)[0].numpy(), ref.conv_transpose2d_f32(x, wt, bias, pads, (s, s), (1, 1), groups, opad))


def test_one_query_attention_and_one_row_conv_transpose_follow_the_vector_matrix_order(ctx):
    """One query row (a decoder step) makes both of sdpa_head's products one-row products of unpacked operands: the reference's gemm_impl takes its
    vector-matrix kernels there (rten-gemm/src/lib.rs:876-891), and so does rten_hip_sdpa_f32 (composed path on the GEMM entry that makes that choice;
    the one-kernel forms replay the blocked chain and are not used).  Same for ConvTranspose with a one-row kernel matrix.  With the gemv order switched
    off (prepacked-weights semantics) both return to the blocked chain."""
    rng = ref.XorShiftRng(123)
    for (B, H, T, D) in ((2, 3, 40, 64), (1, 2, 129, 32), (1, 1, 700, 128), (2, 12, 128, 64)):
        q = rng.f32(B * H * D).reshape(B, H, 1, D) - 0.5
        k = rng.f32(B * H * T * D).reshape(B, H, T, D) - 0.5
        v = rng.f32(B * H * T * D).reshape(B, H, T, D) - 0.5
        m = np.where(rng.f32(B * T).reshape(B, 1, 1, T) > 0.3, 0.0, -np.inf).astype(np.float32)
        for mask in (None, m):
            d = L.SdpaDesc(B, H, 1, T, D, D, H * D, D, D, H * T * D, T * D, D, H * T * D, T * D, D, H * D, D, D, 0 if mask is None else T, 0, 0.125, 1)
            qd, kd, vd = (DeviceTensor.from_numpy(ctx, a) for a in (q, k, v))
            md = DeviceTensor.from_numpy(ctx, mask) if mask is not None else None
            out = DeviceTensor(ctx, (B, H, 1, D), np.float32)
            ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, md.vp if md is not None else None, out.vp)
            _bits(out.numpy(), ref.sdpa(q, k, v, mask=mask, scale=0.125, lanes=16, flush_nan=True))
            ctx.call("rten_hip_set_gemv_order", 0, 0)
            ref.set_gemv_enabled(False)
            try:
                ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, md.vp if md is not None else None, out.vp)
                _bits(out.numpy(), ref.sdpa(q, k, v, mask=mask, scale=0.125, lanes=16, flush_nan=True))
            finally:
                ctx.call("rten_hip_set_gemv_order", 1, 0)
                ref.set_gemv_enabled(True)
    from rten_amd import ops
    x = rng.f32(2 * 300 * 5 * 7).reshape(2, 300, 5, 7) - 0.5
    w = rng.f32(300).reshape(300, 1, 1, 1) - 0.5
    got = ops.ConvTranspose(strides=[1, 1]).run(ctx, [DeviceTensor.from_numpy(ctx, x), DeviceTensor.from_numpy(ctx, w)])[0].numpy()
    _bits(got, ref.conv_transpose2d_f32(x, w, None, (0, 0, 0, 0), (1, 1)))
