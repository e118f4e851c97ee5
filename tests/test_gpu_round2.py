"""Round-2 parity tests (all through the C ABI, on the GPU):

  * BASELINE-size configs against the CPU oracle: int8 ResNet-50 at batch 32 (configs[2]; runner and ONNX executor, producer
    statistics on and off) and BERT-base 12 layers x batch 32 x 128 tokens (configs[3]).
  * MatMulInteger: every batching form of matmul_impl (src/ops/matmul.rs:208-385) incl. the cycled row zero points of a
    collapsed batched LHS (:266-280) and a batched / broadcast RHS; the prepacked RHS (Operator::prepack, :696-705).
  * the boundary: one context shared by two host threads (Model::run(&self)), RCCL communicator + broadcast behind the ABI.
  * graph-level guards of the fused steps (run-time shape checks with the unfused sequence as the fallback).
"""
import ctypes as C
import threading

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
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        if not same.all():
            idx = tuple(np.argwhere(~same)[0])
            raise AssertionError(f"{(~same).sum()} of {a.size} elements differ; first at {idx}: {a[idx]!r} vs {b[idx]!r}")
    else:
        assert np.array_equal(a, b), f"{(a != b).sum()} of {a.size} elements differ"


# ------------------------------------------------------------------------------------------ BASELINE-size configs
def test_resnet50_int8_batch32_bit_exact_runner(ctx):
    """BASELINE configs[2] at the benchmarked size.  DynamicQuantizeLinear statistics are per WHOLE tensor incl. the batch
    (src/ops/quantize.rs:397-419) and the producer-side min/max fold (rten_hip_conv2d_int8_stats, 256 slots) sees 16x more
    workgroups than at batch 2, so batch 32 is its own case -- not a property of the batch-2 test."""
    from oracle import models as omodels
    from rten_amd.workloads import resnet50_int8
    from tests import baseline_oracle as bo
    w = bo.resnet_weights()
    net = resnet50_int8.ResNet50Int8(ctx, batch=32, weights=w)
    net.upload_weights()
    x = bo.resnet_input()  # bench.py's rank-0 batch
    net.x.upload(x)
    want = bo.resnet50_int8_logits()
    net.forward()
    bits_equal(net.logits.numpy(), want)
    # two-sweep DynamicQuantizeLinear everywhere (no producer-side statistics): same bits
    net.producer_stats = False
    net.logits.upload(np.zeros_like(want))
    net.forward()
    bits_equal(net.logits.numpy(), want)
    # hipGraph replay with producer statistics (what bench.py --config int8 times)
    net.producer_stats = True
    net.capture()
    net.logits.upload(np.zeros_like(want))
    net.run()
    bits_equal(net.logits.numpy(), want)
    # pointwise layers with the quantizer fused into the GEMM's loader (rten_hip_conv2d_int8_dql): same bits, eager and replayed
    net.graph, net.fused_dql = None, True
    net.logits.upload(np.zeros_like(want))
    net.forward()
    bits_equal(net.logits.numpy(), want)
    net.capture()
    net.logits.upload(np.zeros_like(want))
    net.run()
    bits_equal(net.logits.numpy(), want)
    net.graph, net.fused_dql = None, False
    # the other pad modes differ from the x86 default on a padded network: the mode is observable, i.e. it IS an assumption
    net2 = resnet50_int8.ResNet50Int8(ctx, batch=32, weights=w, pad_mode=L.PAD_ZERO_POINT)
    net2.upload_weights()
    net2.x.upload(x)
    net2.forward()
    assert not np.array_equal(net2.logits.numpy(), want)
    bits_equal(net2.logits.numpy(), omodels.resnet50_int8_forward(net.specs, omodels.quantize_weights_int8(w), x, pad_mode=ref.PAD_ZERO_POINT))


def test_resnet50_int8_batch32_bit_exact_onnx_executor(tmp_path):
    from rten_amd import onnx_writer as ow
    from tests import baseline_oracle as bo
    from tests.test_graph_executor import _run_model
    w, x, want = bo.resnet_weights(), bo.resnet_input(), bo.resnet50_int8_logits()
    got, log = _run_model(tmp_path, ow.resnet50_int8(w), x, "logits", "--graph", "-n", "2")
    assert "49 DynamicQuantizeLinear write the staged layout directly" in log and "Captured the plan into a hipGraph" in log
    assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))
    got2, _ = _run_model(tmp_path, ow.resnet50_int8(w), x, "logits", "--no-fuse")
    assert np.array_equal(got2.view(np.int32), want.ravel().view(np.int32))


def test_bert_base_full_size_bit_exact(ctx):
    """BASELINE configs[3] as benchmarked: 12 layers, hidden 768, 12 heads, batch 32 x 128 tokens, ragged attention masks."""
    from rten_amd.workloads import bert
    from tests import baseline_oracle as bo
    B, S = 32, 128
    cfg, w, ids, am, tts, want = bo.bert_base_case(B, S)
    net = bert.Bert(ctx, cfg, B, S, w)
    net.set_inputs(ids, am, tts)
    got = net.forward().numpy()
    bits_equal(got, want)
    net.capture()
    net.run()
    bits_equal(net.x.numpy(), want)


# ------------------------------------------------------------------------------------------ MatMulInteger forms (a7, a15)
def _mmi_ref(a, b, a_zp, b_zp):
    """matmul_integer by definition: sum_k (A - a_zp[row]) (B - b_zp[col]) in wrapping i32, numpy.matmul shape rules."""
    a64, b64 = a.astype(np.int64), b.astype(np.int64)
    if a_zp is not None:
        az = np.asarray(a_zp).astype(np.int64)
        a64 = a64 - (az.reshape(-1, 1) if az.ndim == 1 and a.ndim > 1 else az)
    if b_zp is not None:
        b64 = b64 - np.asarray(b_zp).astype(np.int64)
    return np.matmul(a64, b64).astype(np.int64).astype(np.int32)


@pytest.mark.parametrize("a_dt,b_dt", [(np.uint8, np.int8), (np.uint8, np.uint8), (np.int8, np.int8), (np.int8, np.uint8)])
def test_matmul_integer_batching_forms(ctx, a_dt, b_dt):
    rng = np.random.default_rng(7)

    def rnd(dt, shape):
        info = np.iinfo(dt)
        return rng.integers(info.min, info.max + 1, shape).astype(dt)
    # (a shape, b shape, a_zp kind, b_zp kind): test_matmul_integer's table (matmul.rs:1365-1500) widened with prefixes
    cases = [((5, 70), (70, 9), None, None), ((5, 70), (70, 9), "scalar", "scalar"), ((5, 70), (70, 9), "vec", "vec"),
             ((3, 5, 70), (70, 9), "vec", "vec"),              # [A, M, K] x [K, N]: row zero points cycle with period M
             ((2, 3, 4, 33), (33, 17), "vec", "scalar"),
             ((3, 5, 70), (3, 70, 9), "vec", "vec"),           # batched RHS
             ((5, 70), (4, 70, 9), "vec", "vec"),              # LHS broadcast over a batched RHS
             ((2, 1, 5, 40), (1, 3, 40, 6), "vec", "vec"),     # general broadcast of both prefixes
             ((70,), (70, 9), "scalar", "vec"),                # vector x matrix
             ((5, 70), (70,), "vec", "scalar"),                # matrix x vector
             ((3, 160, 300), (300, 130), "vec", "vec")]        # several tiles, K tail, cycled zero points
    for ash, bsh, az_kind, bz_kind in cases:
        a, b = rnd(a_dt, ash), rnd(b_dt, bsh)
        m = ash[-2] if len(ash) > 1 else 1
        n = bsh[-1] if len(bsh) > 1 else 1
        a_zp = None if az_kind is None else (rnd(a_dt, ()) if az_kind == "scalar" else rnd(a_dt, (m,)))
        b_zp = None if bz_kind is None else (rnd(b_dt, ()) if bz_kind == "scalar" else rnd(b_dt, (n,)))
        want = _mmi_ref(a, b, a_zp, b_zp)
        for path in (0, 1):  # staged LDS-DMA kernel and the generic kernel
            ctx.call("rten_hip_set_int8_path", path)
            try:
                got = ops.MatMulInteger().run(ctx, [dev(ctx, a), dev(ctx, b), None if a_zp is None else dev(ctx, a_zp),
                                                    None if b_zp is None else dev(ctx, b_zp)])[0].numpy()
            finally:
                ctx.call("rten_hip_set_int8_path", 0)
            bits_equal(got, want)
    # MatMulIntegerToFloat on a collapsed batched LHS with a per-column scale
    a, b = rnd(a_dt, (3, 6, 50)), rnd(b_dt, (50, 11))
    a_zp, b_zp, sc = rnd(a_dt, (6,)), rnd(b_dt, (11,)), rng.random(11, dtype=np.float32) + 0.01
    got = ops.MatMulIntegerToFloat().run(ctx, [dev(ctx, a), dev(ctx, b), dev(ctx, a_zp), dev(ctx, b_zp), dev(ctx, sc)])[0].numpy()
    bits_equal(got, ref.cast_scale(_mmi_ref(a, b, a_zp, b_zp).reshape(-1, 11), sc).reshape(3, 6, 11))
    # errors of the reference, verbatim
    with pytest.raises(ops.OpError) as e:
        ops.MatMulInteger().run(ctx, [dev(ctx, rnd(a_dt, (3, 5, 7))), dev(ctx, rnd(b_dt, (2, 7, 4)))])
    assert e.value == ops.IncompatibleInputShapes("Cannot broadcast shapes")
    with pytest.raises(ops.OpError) as e:
        ops.MatMulInteger().run(ctx, [dev(ctx, rnd(a_dt, (3, 5, 7))), dev(ctx, rnd(b_dt, (7, 4))), dev(ctx, rnd(a_dt, (3,)))])
    assert e.value == ops.InvalidValue("Zero point has incorrect size")


def test_matmul_integer_prepacked_rhs(ctx):
    """Operator::prepack for input 1 (matmul.rs:696-705): the RHS staged once gives the bits of the per-call staging, for
    signed and unsigned weights, plain and transposed sources, K tails and M / N that are not tile multiples."""
    rng = np.random.default_rng(3)
    for (m, k, n) in ((32, 2048, 1000), (4096, 768, 768), (77, 130, 65), (1, 64, 64)):
        for b_dt in (np.int8, np.uint8):
            a = rng.integers(0, 256, (m, k)).astype(np.uint8)
            info = np.iinfo(b_dt)
            b = rng.integers(info.min, info.max + 1, (k, n)).astype(b_dt)
            a_zp = np.array(rng.integers(0, 256), np.uint8)
            b_zp = rng.integers(info.min, info.max + 1, (n,)).astype(b_dt)
            op = ops.MatMulInteger()
            bd = dev(ctx, b)
            packed = op.prepack(ctx, bd)
            assert packed is not None and packed.size == ctx.lib.rten_hip_gemm_int8_packed_bytes(k, n)
            ins = [dev(ctx, a), bd, dev(ctx, a_zp), dev(ctx, b_zp)]
            got = op.run(ctx, ins, packed_b=packed)[0].numpy()
            bits_equal(got, op.run(ctx, ins)[0].numpy())
            bits_equal(got, _mmi_ref(a, b, a_zp, b_zp))
    # a transposed source ([N, K] weights read as B[k, n] through strides: the ResNet classifier's layout)
    wq = rng.integers(-64, 65, (1000, 2048)).astype(np.int8)
    a = rng.integers(0, 256, (32, 2048)).astype(np.uint8)
    nb = ctx.lib.rten_hip_gemm_int8_packed_bytes(2048, 1000)
    packed = DeviceTensor(ctx, [nb], np.uint8)
    wqd, ad, zpd = dev(ctx, wq), dev(ctx, a), dev(ctx, np.array(7, np.uint8))  # (named: temporaries would be freed before the launches read them)
    ctx.call("rten_hip_gemm_int8_prepack", 2048, 1000, wqd.vp, 1, 2048, 1, packed.vp)
    d = L.GemmInt8Desc(32, 1000, 2048, 2048, 1, 0, 0, 1000, 0, 1, 1, 0, 0, 1, 0, 0, 0, 1)
    y = DeviceTensor(ctx, [32, 1000], np.int32)
    ctx.call("rten_hip_gemm_int8", C.byref(d), ad.vp, packed.vp, zpd.vp, None, None, y.vp)
    bits_equal(y.numpy(), _mmi_ref(a, np.ascontiguousarray(wq.T), np.array(7, np.uint8), None))
    # a prepacked RHS cannot be batched
    d.batch, d.b_bs = 2, 5
    with pytest.raises(L.HipError, match="single matrix"):
        ctx.call("rten_hip_gemm_int8", C.byref(d), ad.vp, packed.vp, zpd.vp, None, None, y.vp)


# ------------------------------------------------------------------------------------------ the boundary
def test_one_context_shared_by_two_host_threads(ctx):
    """Model::run(&self) may be entered by several host threads (src/model.rs:308-550; call site src/graph.rs:782): two threads
    drive different operators (f32 conv with split-K scratch, int8 conv with staging scratch, softmax) through ONE context at
    the same time; every result must equal the single-threaded one bit for bit, and each thread reads its own error text."""
    rng = ref.XorShiftRng(77)
    x = rng.f32(4 * 64 * 28 * 28).reshape(4, 64, 28, 28) - 0.5
    w = (rng.f32(128 * 64 * 9).reshape(128, 64, 3, 3) - 0.5) * 0.1
    b = rng.f32(128) - 0.5
    xq = rng.u8(4 * 32 * 14 * 14).reshape(4, 32, 14, 14)
    wq = rng.i8(64 * 32 * 9, reduced=True).reshape(64, 32, 3, 3)
    sm = rng.f32(512 * 128).reshape(512, 128) - 0.5
    conv = ops.Conv(padding=[1, 1, 1, 1], fuse_relu=True)
    convi = ops.ConvIntegerToFloat(ops.ConvInteger(padding=[1, 1, 1, 1]))
    ins_f = [dev(ctx, x), dev(ctx, w), dev(ctx, b)]
    ins_i = [dev(ctx, xq), dev(ctx, wq), dev(ctx, np.array(121, np.uint8)), None, dev(ctx, np.array(0.01, np.float32))]
    smd = dev(ctx, sm)
    want_f = conv.run(ctx, ins_f)[0].numpy()
    want_i = convi.run(ctx, ins_i)[0].numpy()
    want_s = ops.Softmax(axis=-1).run(ctx, [smd])[0].numpy()
    errors, msgs = [], {}

    def worker(kind):
        try:
            for it in range(40):
                if kind == 0:
                    bits = conv.run(ctx, ins_f)[0].numpy()
                    if not np.array_equal(bits.view(np.int32), want_f.view(np.int32)):
                        errors.append(("f32 conv", it))
                else:
                    bits = convi.run(ctx, ins_i)[0].numpy()
                    if not np.array_equal(bits.view(np.int32), want_i.view(np.int32)):
                        errors.append(("int8 conv", it))
                    s = ops.Softmax(axis=-1).run(ctx, [smd])[0].numpy()
                    if not np.array_equal(s.view(np.int32), want_s.view(np.int32)):
                        errors.append(("softmax", it))
            # per-thread error text: each thread provokes its own failure and must read its own message
            d = L.GemmInt8Desc(4, 4, 4, 4, 1, 4, 1, 4, 0, 1, 3 if kind == 0 else 0, 0, 7 if kind == 1 else 0)
            try:
                ctx.call("rten_hip_gemm_int8", C.byref(d), ins_i[0].vp, ins_i[1].vp, ins_i[2].vp, None, ins_i[4].vp, ins_f[0].vp)
            except L.HipError as e:
                msgs[kind] = e.msg
        except Exception as e:  # noqa: BLE001
            errors.append((kind, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors[:5]
    assert msgs[0] == "Zero point has incorrect size" and msgs[1] == "Scale length does not match tensor columns"
    # a capture holds the context: the other thread's launch waits for graph_end instead of being recorded into the graph
    order = []
    ctx.graph_begin()

    def late():
        conv.run(ctx, ins_f)
        order.append("other thread ran")
    t = threading.Thread(target=late)
    t.start()
    ctx.call("rten_hip_relu_f32", smd.size, smd.vp, smd.vp)
    import time
    time.sleep(0.3)
    order.append("capture ends")
    g = ctx.graph_end()
    t.join(timeout=60)
    assert order == ["capture ends", "other thread ran"]
    ctx.graph_destroy(g)


def test_rccl_communicator_behind_the_abi(ctx):
    """rten_hip_comm_*: RCCL bound directly by the library (dlopen).  A 1-rank communicator on the one GPU of this box: the
    id / init / broadcast / destroy sequence a Rust host runs next to Model::load.  (N > 1 is the driver's 8-GPU bench.)"""
    uid = L.Comm.unique_id(ctx)
    assert len(uid) == 128 and any(uid)
    try:
        comm = L.Comm(ctx, uid, 1, 0)
    except L.HipError as e:
        # Seen once in ~10 sessions on a freshly leased box: RCCL's own initialisation ("unhandled cuda error" inside ncclCommInitRank, before any
        # of this library's code runs on the communicator) fails and succeeds on the next attempt.  One retry with a fresh id; a second failure is a failure.
        print("ncclCommInitRank failed once, retrying:", e)
        uid = L.Comm.unique_id(ctx)
        comm = L.Comm(ctx, uid, 1, 0)
    ws, rk = C.c_int32(), C.c_int32()
    assert ctx.lib.rten_hip_comm_world_size(comm.h, C.byref(ws), C.byref(rk)) == 0 and (ws.value, rk.value) == (1, 0)
    arena = np.random.default_rng(0).integers(0, 256, 1 << 20).astype(np.uint8)
    t = dev(ctx, arena)
    comm.broadcast(t.ptr, t.nbytes, root=0)
    ctx.sync()
    assert np.array_equal(t.numpy(), arena)
    with pytest.raises(L.HipError, match="root out of range"):
        comm.broadcast(t.ptr, t.nbytes, root=1)
    comm.close()
    with pytest.raises(L.HipError, match="rank < world_size"):
        L.Comm(ctx, uid, 2, 2)


# ------------------------------------------------------------------------------------------ graph guards (fused steps)
def test_graph_conv_add_of_a_broadcast_constant_is_not_taken_as_a_residual(tmp_path):
    """Conv -> Add(const [1,O,1,1]) -> Relu (the exporter's explicit-bias form) and Conv -> Add([O,1,1]): the Add's other operand
    does not have the conv's output shape, so the step must run Conv, broadcasting Add, Relu -- never index it as a residual."""
    from rten_amd import onnx_writer as ow
    from tests.test_graph_executor import _run_model
    rng = np.random.default_rng(2)
    w = (rng.standard_normal((8, 3, 3, 3)) * 0.2).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    x = rng.standard_normal((2, 3, 12, 12)).astype(np.float32)
    for cshape in ((1, 8, 1, 1), (8, 1, 1)):
        c = rng.standard_normal(cshape).astype(np.float32)
        nodes = [ow.node("Conv", ["x", "w", "b"], ["c0"], name="conv", kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1]),
                 ow.node("Add", ["c0", "c"], ["s0"], name="add"), ow.node("Relu", ["s0"], ["y"], name="relu")]
        m = ow.model(nodes, [ow.value_info("x", ow.FLOAT, ["batch", 3, 12, 12])], [ow.value_info("y", ow.FLOAT, ["batch", 8, 12, 12])],
                     [ow.tensor("w", w), ow.tensor("b", b), ow.tensor("c", c)])
        want = ref.relu(ref.conv2d_f32(x, w, b, pads=(1, 1, 1, 1)) + c)
        for extra in ((), ("--no-fuse",)):
            got, _ = _run_model(tmp_path, m, x, "y", *extra)
            assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32))


def test_graph_conv_integer_per_channel_and_odd_scales(tmp_path):
    """ConvInteger -> Cast -> Mul(scale): a per-channel [1,O,1,1] scale runs (fused, exactly as Cast -> Mul would), a scale of
    any other broadcastable shape runs unfused -- neither fails with "scale should be a scalar" (ConvIntegerToFloatFusion
    leaves non-scalar scales unfused, fusions.rs:1037-1047)."""
    from rten_amd import onnx_writer as ow
    from tests.test_graph_executor import _run_model
    rng = np.random.default_rng(4)
    wq = rng.integers(-64, 65, (8, 4, 3, 3)).astype(np.int8)
    x = rng.random((2, 4, 10, 10), dtype=np.float32)
    for sshape in ((1, 8, 1, 1), (8, 1, 1), (1, 1, 1, 10), (1,)):
        sc = (rng.random(sshape, dtype=np.float32) * 0.01 + 0.001).astype(np.float32)
        nodes = [ow.node("DynamicQuantizeLinear", ["x"], ["xq", "xs", "xz"], name="dql"),
                 ow.node("ConvInteger", ["xq", "wq", "xz"], ["acc"], name="conv", kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1]),
                 ow.node("Cast", ["acc"], ["f"], name="cast", to=ow.FLOAT), ow.node("Mul", ["f", "sc"], ["y"], name="mul")]
        m = ow.model(nodes, [ow.value_info("x", ow.FLOAT, ["batch", 4, 10, 10])], [ow.value_info("y", ow.FLOAT, ["batch", 8, 10, 10])],
                     [ow.tensor("wq", wq), ow.tensor("sc", sc)])
        q, s, z = ref.dynamic_quantize_linear(x)
        acc = ref.conv2d_int8(q, wq, x_zp=int(z), pads=(1, 1, 1, 1), pad_mode=ref.PAD_RAW0_I8)
        want = acc.astype(np.float32) * sc
        for extra in ((), ("--no-fuse",)):
            got, _ = _run_model(tmp_path, m, x, "y", *extra)
            assert np.array_equal(got.view(np.int32), want.ravel().view(np.int32)), (sshape, extra)


# ------------------------------------------------------------------------------------------ f32 conv launch plans (round 2)
def test_conv_f32_thin_tail_and_16x16_plans_bit_exact(ctx):
    """The persistent plan (split mode 5: num_cus x groups workgroups walking tile lists), the thin-tile tail plan (split mode 4: whole rounds of 256 tiles + 16x64 tiles on v_mfma_f32_16x16x4_f32), the
    fragments-first variants (16..19) and the 16x16x4 variants (20..23) on geometries that leave a partial round: bits of the
    oracle's k-ordered chain, for 1x1 (dense) and 3x3 (gather) layers, K below and above one depth block, with the fused
    bias / residual / Relu epilogue."""
    from tests.test_gpu_parity import gpu_conv
    rng = ref.XorShiftRng(4242)
    cases = [  # N, C, H, W, O, k, pad, stride  -> tiles of 64x64 = ceil(O/64) * ceil(N*OH*OW/64)
        (9, 64, 56, 56, 64, 1, 0, 1),     # 441 tiles: one whole round + 185
        (6, 40, 57, 57, 64, 3, 1, 1),     # K = 360 (two depth blocks), ragged width, 305 tiles
        (8, 300, 29, 29, 130, 1, 0, 1),   # M tail (130 rows = 3 row tiles: the whole rounds hold 85 column tiles), K = 300
        (12, 70, 60, 60, 128, 3, 1, 2),   # strided 3x3, K = 630 (three depth blocks), 338 tiles
        (10, 96, 55, 55, 100, 1, 0, 1),   # K = 96 (lean kernel: K % 32 == 0, one depth block), M tail, 948 tiles
        (7, 64, 41, 41, 72, 3, 1, 1),     # K = 576 (lean kernel, three depth blocks), gather with ragged rows, 368 tiles
        (9, 320, 30, 30, 64, 1, 0, 2),    # 1x1 stride 2 (tap-masked gather with one tap), K = 320
    ]
    for (N, C_, H, W, O, k, pad, stride) in cases:
        x = rng.f32(N * C_ * H * W).reshape(N, C_, H, W) - 0.5
        w = (rng.f32(O * C_ * k * k).reshape(O, C_, k, k) - 0.5) * 0.2
        b = rng.f32(O) - 0.5
        oh = (H + 2 * pad - k) // stride + 1
        res = rng.f32(N * O * oh * oh).reshape(N, O, oh, oh) - 0.5
        want = ref.conv2d_f32(x, w, b, pads=(pad,) * 4, strides=(stride,) * 2, residual=res, relu=True)
        for variant, mode, groups in [(v, m, 1) for v in (3, 2, 1, 0, 15, 19, 16, 23, 22, 21, 20) for m in (0, 4)] + \
                                     [(v, 5, r) for v in (3, 2, 1, 0, 23, 22, 21, 20) for r in (1, 2, 3)] + [(3, 6, r) for r in (1, 2, 3)]:
            if True:
                ctx.call("rten_hip_set_gemm_split", mode, groups)
                try:
                    got = gpu_conv(ctx, x, w, b, (pad,) * 4, (stride,) * 2, residual=res, relu=True, prepack=True, variant=variant)
                finally:
                    ctx.call("rten_hip_set_gemm_split", 3, 1)
                try:
                    bits_equal(got, want)
                except AssertionError as e:
                    raise AssertionError(f"conv N={N} C={C_} {H}x{W} O={O} k={k} variant={variant} split mode={mode} groups={groups}: {e}") from None
    # the 16x16x4 GEMM path (row-major A, BERT projection form) incl. alpha / beta / per-column bias
    M, K, Nn = 200, 300, 136
    a, bm_ = rng.f32(M * K).reshape(M, K) - 0.5, rng.f32(K * Nn).reshape(K, Nn) - 0.5
    c0, bias = rng.f32(M * Nn).reshape(M, Nn) - 0.5, rng.f32(Nn) - 0.5
    from tests.test_gpu_parity import gpu_gemm
    want = ref.gemm_f32(a, bm_, c=c0, alpha=0.5, beta=0.75, bias=bias, bias_kind=L.BIAS_PER_COL)
    for variant in (20, 21, 22, 23, 16, 19):
        bits_equal(gpu_gemm(ctx, a, bm_, c=c0.copy(), alpha=0.5, beta=0.75, bias=bias, bias_kind=L.BIAS_PER_COL, variant=variant), want)
    for variant in (0, 3, 20, 23):  # persistent plan on the plain GEMM form (row-major A), fewer tiles than workgroups
        ctx.call("rten_hip_set_gemm_split", 5, 2)
        try:
            bits_equal(gpu_gemm(ctx, a, bm_, c=c0.copy(), alpha=0.5, beta=0.75, bias=bias, bias_kind=L.BIAS_PER_COL, variant=variant), want)
        finally:
            ctx.call("rten_hip_set_gemm_split", 3, 1)


def test_golden_package_for_the_real_reference(tmp_path):
    """tools/make_rten_golden.py: the files a maintainer feeds to `rten --check-outputs` -- the expected outputs written by the HIP
    backend are the oracle's, bit for bit, and the package checks against itself with rten_hip_run (same flags as rten-cli)."""
    import subprocess
    import sys
    from safetensors.numpy import load_file
    from tests.test_graph_executor import ROOT, build_cli
    import os
    outs = {}
    for source in ("hip", "oracle"):
        d = tmp_path / source
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_rten_golden.py"), "--out", str(d), "--batch", "1", "--source", source,
                            "--models", "resnet50_int8"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs[source] = load_file(str(d / "resnet50_int8.expected.safetensors"))["logits"]
    bits_equal(outs["hip"], outs["oracle"])
    d = tmp_path / "oracle"
    r = subprocess.run([build_cli(), "-s", "batch=1", "-i", str(d / "resnet50_int8.inputs.safetensors"), "--check-outputs",
                        str(d / "resnet50_int8.expected.safetensors"), "--max-diff", "0", str(d / "resnet50_int8.onnx")], capture_output=True, text=True)
    assert r.returncode == 0 and "max diff 0" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_chained_resnet50_bit_exact(ctx):
    """The batch as independent sub-batch chains on their own streams (what bench.py times by default): eager, hipGraph replay
    on every stream placement, after the stand-alone + co-run autotune -- always the oracle's logits, bit for bit; and at the
    full batch of 32 the 4-chain logits equal the single-chain ones."""
    from oracle import models as omodels
    from rten_amd.workloads import resnet50
    w = resnet50.make_weights()
    x = ref.XorShiftRng(77).f32(5 * 3 * 224 * 224).reshape(5, 3, 224, 224)
    net = resnet50.ChainedResNet50(ctx, 5, w, chains=2)  # uneven split: 3 + 2 images
    assert net.sizes == [3, 2]
    net.upload_weights()
    net.x.upload(x)
    net.forward()
    ctx.sync()
    want = omodels.resnet50_forward(net.specs, w, x)
    bits_equal(net.logits.numpy(), want)
    net.autotune(reps=1, top=3, corun_reps=2)
    assert set(net.cotune) == {l["name"] for l in net.specs}
    net.capture()
    rows = net.tune_placement(steps=2)
    assert len(rows) >= net.POOL - 1 and net.place in [r[0] for r in rows] and len(set(net.place)) == net.chains
    for place, _ in rows:
        net.place = place
        net.logits.upload(np.zeros_like(want))
        net.run()
        ctx.sync()
        bits_equal(net.logits.numpy(), want)
    rep = net.profile_pass(1)
    assert any(r["kernel"].startswith("igemm_f32") for r in rep)
    # plan tables round-trip through the keyed form bench.py --save-plan / --load-plan uses
    table = net.plan_table()
    assert set(table) == {"3", "2"}
    net.variants = table
    assert net.nets[1].variants == {k: tuple(v) for k, v in table["2"].items()}
    # full batch: 4 chains == 1 chain
    x32 = ref.XorShiftRng(5).f32(32 * 3 * 224 * 224).reshape(32, 3, 224, 224)
    one = resnet50.ResNet50(ctx, 32, w, arena_ptr=net.arena.ptr, arena_keepalive=net.arena)
    one.x.upload(x32)
    one.forward()
    four = resnet50.ChainedResNet50(ctx, 32, w, chains=4, arena_ptr=net.arena.ptr, arena_keepalive=net.arena)
    four.x.upload(x32)
    four.capture()
    four.run()
    ctx.sync()
    bits_equal(four.logits.numpy(), one.logits.numpy())


def test_resnet50_int8_shortcuts_on_second_stream(ctx):
    """int8 ResNet-50 with the projection shortcuts on a second stream (double-buffered quantized input, own cast_scale slot):
    eager and as parallel hipGraph branches, bit-identical to the oracle."""
    from oracle import models as omodels
    from rten_amd.workloads import resnet50, resnet50_int8
    w = resnet50.make_weights()
    x = ref.XorShiftRng(4242).f32(3 * 3 * 224 * 224).reshape(3, 3, 224, 224)
    net = resnet50_int8.ResNet50Int8(ctx, batch=3, weights=w)
    net.upload_weights()
    net.x.upload(x)
    want = omodels.resnet50_int8_forward(net.specs, omodels.quantize_weights_int8(w), x)
    net.concurrent = True
    for _ in range(3):  # repeated: a missing join shows up as a race between passes
        net.logits.upload(np.zeros_like(want))
        net.forward()
        ctx.sync()
        bits_equal(net.logits.numpy(), want)
    net.capture()
    for _ in range(3):
        net.logits.upload(np.zeros_like(want))
        net.run()
        ctx.sync()
        bits_equal(net.logits.numpy(), want)


def test_split_k_last_arrival_fold_is_deterministic(ctx):
    """Split-K producers fold their tile in the same launch: whichever workgroup arrives last replays the ordered fold.  The
    arrival order varies from launch to launch, the bits must not: 300 launches of a stage-3 3x3 conv (K = 2304, every tile cut
    into 9 / 3 / 2 K groups) and of the batch-1 classifier GEMM, each compared with the oracle's unsplit chain."""
    from tests.test_gpu_parity import gpu_gemm
    rng = ref.XorShiftRng(2024)
    x = rng.f32(2 * 256 * 14 * 14).reshape(2, 256, 14, 14) - 0.5
    w = (rng.f32(256 * 256 * 9).reshape(256, 256, 3, 3) - 0.5) * 0.05
    b = rng.f32(256) - 0.5
    res = rng.f32(2 * 256 * 14 * 14).reshape(2, 256, 14, 14) - 0.5
    want = ref.conv2d_f32(x, w, b, pads=(1, 1, 1, 1), residual=res, relu=True)
    op = ops.Conv(padding=[1, 1, 1, 1], fuse_relu=True)
    xd, wd, bd, rd = (DeviceTensor.from_numpy(ctx, a) for a in (x, w, b, res))
    packed = op.prepack(ctx, wd, op._geometry(ctx, x.shape, w.shape))
    a1 = rng.f32(2048).reshape(1, 2048) - 0.5
    b1 = rng.f32(2048 * 1000).reshape(2048, 1000) - 0.5
    want1 = None
    try:
        for mode, groups in ((2, 9), (2, 3), (2, 2), (1, 4), (3, 1)):
            ctx.call("rten_hip_set_gemm_split", mode, groups)
            for it in range(60):
                y = op.run(ctx, [xd, wd, bd, rd], packed_weight=packed)[0]
                if it % 10 == 9 or it == 0:
                    bits_equal(y.numpy(), want)
                y.free()
            got1 = gpu_gemm(ctx, a1, b1)
            if want1 is None:
                want1 = got1
            bits_equal(got1, want1)  # same bits under every split plan (M == 1 parity itself is by tolerance)
    finally:
        ctx.call("rten_hip_set_gemm_split", 3, 1)


def test_conv2d_int8_dql_matches_the_separate_operators(ctx):
    """rten_hip_conv2d_int8_dql (DynamicQuantizeLinear inside the integer GEMM's loader) against DynamicQuantizeLinear (staged) +
    ConvIntegerToFloat on the same f32 tensor and statistics: outputs, output statistics and the quantizer's own scale / zero point,
    for every tile shape the dispatcher picks, scalar and per-channel weight scales, residual + Relu, ragged pixel counts."""
    rng = ref.XorShiftRng(777)
    sb = ctx.lib.rten_hip_minmax_stats_bytes()
    for (n, c, h, w, o, per_ch) in ((2, 64, 56, 56, 256, False), (32, 256, 14, 14, 1024, False), (3, 128, 7, 7, 512, True), (32, 64, 7, 7, 64, False),
                                    (1, 192, 5, 3, 40, True), (32, 512, 28, 28, 128, False)):
        # producer: an int8 conv whose f32 output (+ statistics) is the tensor under test
        x0 = rng.u8(n * 16 * h * w).reshape(n, 16, h, w)
        w0 = rng.i8(c * 16, reduced=True).reshape(c, 16, 1, 1)
        d0 = L.Conv2dInt8Desc(L.Conv2dDesc(n, 16, h, w, c, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, h, w), 0, 1, 0, L.PAD_RAW0_I8, 0, 0, 1)
        stats = DeviceTensor(ctx, (2 * sb,), np.uint8)
        ctx.call("rten_hip_minmax_stats_reset", stats.vp, 2)
        st_in, st_a, st_b = C.c_void_p(stats.ptr), None, None
        t = DeviceTensor(ctx, (n, c, h, w), np.float32)
        zp0, sc0 = DeviceTensor.from_numpy(ctx, np.array([7], np.uint8)), DeviceTensor.from_numpy(ctx, np.array([0.013], np.float32))
        x0d, w0d = DeviceTensor.from_numpy(ctx, x0), DeviceTensor.from_numpy(ctx, w0)  # (named: temporaries would be freed before the launch reads them)
        ctx.call("rten_hip_conv2d_int8_stats", C.byref(d0), x0d.vp, w0d.vp, zp0.vp, None, sc0.vp, None, None, 0,
                 t.vp, st_in)
        # consumer weights
        wq = rng.i8(o * c, reduced=True).reshape(o, c, 1, 1)
        ws = (rng.f32(o if per_ch else 1) * 0.01 + 0.001).astype(np.float32)
        bias = rng.f32(o) - 0.5
        res = rng.f32(n * o * h * w).reshape(n, o, h, w) - 0.5
        d = L.Conv2dInt8Desc(L.Conv2dDesc(n, c, h, w, o, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, h, w), 0, 1, 0, L.PAD_RAW0_I8, 1, 1, o if per_ch else 1)
        packed = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_int8_packed_bytes(C.byref(d)),), np.uint8)
        wqd = DeviceTensor.from_numpy(ctx, wq)
        ctx.call("rten_hip_conv2d_int8_prepack", C.byref(d), wqd.vp, packed.vp)
        wsd, bd, rd = DeviceTensor.from_numpy(ctx, ws), DeviceTensor.from_numpy(ctx, bias), DeviceTensor.from_numpy(ctx, res)
        outs = []
        for fused in (False, True):
            y = DeviceTensor(ctx, (n, o, h, w), np.float32)
            ost = DeviceTensor(ctx, (sb,), np.uint8)
            ctx.call("rten_hip_minmax_stats_reset", ost.vp, 1)
            xs, xz = DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8)
            flags = L.CONV_RELU | L.CONV_RESIDUAL
            if fused:
                ctx.call("rten_hip_conv2d_int8_dql", C.byref(d), t.vp, st_in, packed.vp, wsd.vp, bd.vp, rd.vp, flags, y.vp, ost.vp, xs.vp, xz.vp)
            else:
                staged = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d)),), np.uint8)
                sc = DeviceTensor(ctx, (o if per_ch else 1,), np.float32)
                ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), t.vp, st_in, staged.vp, xs.vp, xz.vp, None, None)
                ctx.call("rten_hip_mul_f32", o if per_ch else 1, wsd.vp, xs.vp, 1, sc.vp)
                ctx.call("rten_hip_conv2d_int8_stats", C.byref(d), staged.vp, packed.vp, xz.vp, None, sc.vp, bd.vp, rd.vp, flags, y.vp, ost.vp)
            ctx.sync()
            outs.append((y.numpy(), ost.numpy(), xs.numpy(), xz.numpy()))
        bits_equal(outs[0][0], outs[1][0])
        assert np.array_equal(outs[0][2].view(np.uint32), outs[1][2].view(np.uint32)) and np.array_equal(outs[0][3], outs[1][3])
        # statistics blocks hold the same min / max (which slot a workgroup hits may differ)
        k = sb // 8
        a0, a1 = outs[0][1].view(np.uint32), outs[1][1].view(np.uint32)
        assert a0[:k].min() == a1[:k].min() and a0[k:].max() == a1[k:].max()
