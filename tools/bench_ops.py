#!/usr/bin/env python3
"""Per-kernel microbench: every hot-path kernel class of SURVEY.md section 8 against the roofline that bounds it.

Not the driver's contract (that is bench.py); this writes one JSON document with a row per kernel:
    {"op": ..., "shape": ..., "us": ..., "bound": "hbm"|"mfma", "achieved": ..., "unit": "GB/s"|"TFLOP/s", "frac": ...}
Sizes are the ones the BASELINE configs put on these kernels (ResNet-50 batch 32, BERT-base batch 32 x 128).

    python tools/bench_ops.py [--reps 20] > gpurun_out/ops.json
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L  # noqa: E402
from rten_amd.tensor import DeviceTensor  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)
F32_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
I8_PEAK_TOPS = 5033.0   # v_mfma_i32_32x32x32_i8: 32768 MAC / 16 cycles per CU -> 256 CUs x 2.4 GHz (dense)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the op name (e.g. Integer)")
    ap.add_argument("--cpu-baseline", action="store_true", help="add bench.py's CPU-oracle timing (cpu_baseline leg) beside every row")
    args = ap.parse_args()
    ctx = L.Context(0)
    rng = np.random.default_rng(0)
    rows = []

    def dev(a):
        return DeviceTensor.from_numpy(ctx, np.ascontiguousarray(a))

    def empty(shape, dt=np.float32):
        return DeviceTensor(ctx, shape, dt)

    def timeit(fn):
        # warm the clocks on THIS kernel first (the chip re-clocks within tens of milliseconds of a change of load; a cold
        # first batch reads 10-15 % slow on the MFMA rows), then keep the best of three timed batches
        import time
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:
            for _ in range(args.reps):
                fn()
            ctx.sync()
        best = 1e30
        for _ in range(3):
            ctx.timer_start(3)
            for _ in range(args.reps):
                fn()
            ctx.timer_stop(3)
            best = min(best, ctx.timer_ms(3) / args.reps * 1e3)
        return best  # us

    def want(op):
        return args.only.lower() in op.lower()

    def hbm(op, shape, fn, nbytes):
        if not want(op):
            return
        us = timeit(fn)
        gbs = nbytes / us / 1e3
        rows.append({"op": op, "shape": shape, "us": round(us, 2), "bound": "hbm", "achieved": round(gbs, 1), "unit": "GB/s",
                     "peak": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 3), "algorithmic_bytes": int(nbytes)})

    def mfma(op, shape, fn, flops, peak, unit):
        if not want(op):
            return
        us = timeit(fn)
        t = flops / us / 1e6
        rows.append({"op": op, "shape": shape, "us": round(us, 2), "bound": "mfma", "achieved": round(t, 2), "unit": unit,
                     "peak": peak, "frac": round(t / peak, 3), "algorithmic_ops": int(flops)})

    # ---- element-wise / row-wise kernels on ResNet stage-0 and BERT activations
    n_act = 32 * 256 * 56 * 56
    x = dev(rng.standard_normal(n_act, dtype=np.float32))
    x2 = dev(rng.standard_normal(n_act, dtype=np.float32))
    y = empty((n_act,))
    hbm("Relu", f"n={n_act}", (lambda: ctx.call("rten_hip_relu_f32", n_act, x.vp, y.vp)), 8.0 * n_act)
    hbm("Add", f"n={n_act}", (lambda: ctx.call("rten_hip_add_f32", n_act, x.vp, x2.vp, n_act, y.vp)), 12.0 * n_act)
    n_ffn = 4096 * 3072
    hbm("Gelu", f"n={n_ffn}", (lambda: ctx.call("rten_hip_gelu_f32", n_ffn, x.vp, y.vp)), 8.0 * n_ffn)
    hbm("Erf", f"n={n_ffn}", (lambda: ctx.call("rten_hip_erf_f32", n_ffn, x.vp, y.vp)), 8.0 * n_ffn)
    r, c = 32 * 12 * 128, 128
    hbm("Softmax", f"rows={r} cols={c}", (lambda: ctx.call("rten_hip_softmax_f32", r, c, x.vp, None, 1, 1, 0, y.vp)), 8.0 * r * c)
    r, c = 4096, 768
    g, b = dev(np.ones(c, np.float32)), dev(np.zeros(c, np.float32))
    hbm("LayerNormalization", f"rows={r} cols={c}",
        (lambda: ctx.call("rten_hip_layer_norm_f32", r, c, x.vp, g.vp, b.vp, C.c_float(1.0), C.c_float(0.0), C.c_float(1e-12), y.vp)), 8.0 * r * c)
    # reference point for the short row-wise launches: a device copy of the same 12.6 MB (the fraction of the HBM peak ANY 6 us launch can
    # reach is bounded by its ramp-up and drain; LayerNormalization is to be read against this row, not against 8 TB/s)
    hbm("copy, LayerNormalization's size (reference point)", f"{4 * r * c} B", (lambda: ctx.call("rten_hip_memcpy_d2d", y.vp, x.vp, C.c_size_t(4 * r * c))), 8.0 * r * c)
    u8 = empty((n_act,), np.uint8)
    sc, zp = empty((1,)), empty((1,), np.uint8)
    hbm("DynamicQuantizeLinear", f"n={n_act}", (lambda: ctx.call("rten_hip_dynamic_quantize_linear", n_act, x.vp, u8.vp, sc.vp, zp.vp)), 9.0 * n_act)
    xi = dev(rng.integers(-1000, 1000, n_act).astype(np.int32))
    hbm("cast_scale", f"n={n_act}", (lambda: ctx.call("rten_hip_cast_scale", n_act, xi.vp, sc.vp, 1, y.vp)), 8.0 * n_act)
    pd = L.Pool2dDesc(32, 64, 112, 112, 3, 3, 2, 2, (C.c_int32 * 4)(1, 1, 1, 1), 56, 56, 0)
    n_in, n_out = 32 * 64 * 112 * 112, 32 * 64 * 56 * 56
    hbm("MaxPool 3x3/2", "32x64x112x112", (lambda: ctx.call("rten_hip_max_pool2d_f32", C.byref(pd), x.vp, y.vp)), 4.0 * (n_in + n_out))
    st = empty((ctx.lib.rten_hip_minmax_stats_bytes(),), np.uint8)
    ctx.call("rten_hip_minmax_stats_reset", st.vp, 1)
    hbm("MaxPool 3x3/2 + statistics for the following DynamicQuantizeLinear", "32x64x112x112",
        (lambda: ctx.call("rten_hip_max_pool2d_f32_stats", C.byref(pd), x.vp, y.vp, st.vp)), 4.0 * (n_in + n_out))
    hbm("GlobalAveragePool", "32x2048x7x7", (lambda: ctx.call("rten_hip_global_average_pool_f32", 32 * 2048, 49, x.vp, y.vp)), 4.0 * 32 * 2048 * 50)

    # depthwise 3x3 (MobileNet-style: 32 x 144 x 56 x 56) and a 2x upsampling ConvTranspose (32 x 64 x 28 x 28 -> 32 x 32 x 56 x 56, 4x4 / 2)
    cdw = 144
    xdw, wdw, bdw = dev(rng.standard_normal((32, cdw, 56, 56), dtype=np.float32)), dev(rng.standard_normal((cdw, 1, 3, 3), dtype=np.float32)), dev(np.zeros(cdw, np.float32))
    ydw = empty((32, cdw, 56, 56))
    ddw = L.Conv2dDesc(32, cdw, 56, 56, cdw, 3, 3, (C.c_int32 * 4)(1, 1, 1, 1), 1, 1, 1, 1, cdw, 56, 56)
    hbm("Conv depthwise 3x3", f"32x{cdw}x56x56", (lambda: ctx.call("rten_hip_conv2d_f32", C.byref(ddw), xdw.vp, wdw.vp, 0, bdw.vp, None, 0, ydw.vp)), 8.0 * 32 * cdw * 56 * 56)
    xct, wct, bct = dev(rng.standard_normal((32, 64, 28, 28), dtype=np.float32)), dev(rng.standard_normal((64, 32, 4, 4), dtype=np.float32)), dev(np.zeros(32, np.float32))
    yct = empty((32, 32, 56, 56))
    dct = L.Conv2dDesc(32, 64, 28, 28, 32, 4, 4, (C.c_int32 * 4)(1, 1, 1, 1), 2, 2, 1, 1, 1, 56, 56)
    mfma("ConvTranspose 4x4/2", "32x64x28x28 -> 32", (lambda: ctx.call("rten_hip_conv_transpose2d_f32", C.byref(dct), xct.vp, wct.vp, bct.vp, yct.vp)),
         2.0 * 32 * 64 * 28 * 28 * 32 * 16, F32_PEAK_TF, "TFLOP/s")

    # ---- MatMulNBits on LLM-decoder projections (4-bit blocks of 32): decode (1 row, bound by streaming the packed weights) on a
    # 4096 x 4096 attention projection and a 4096 x 14336 FFN up-projection, and a 128-row prefill
    for (kq, nq, rows_list) in ((4096, 4096, (1, 128)), (4096, 14336, (1,))):
        bsq = 32
        wq, wsc = dev(rng.integers(0, 256, (nq, kq // bsq, bsq // 2)).astype(np.uint8)), dev(rng.random((nq, kq // bsq), dtype=np.float32) * 0.01)
        for rows_q in rows_list:
            xq, yq = dev(rng.standard_normal((rows_q, kq), dtype=np.float32)), empty((rows_q, nq))
            fn = (lambda xq=xq, yq=yq, rows_q=rows_q, kq=kq, nq=nq, wq=wq, wsc=wsc: ctx.call("rten_hip_matmul_nbits_f32", 1, rows_q, kq, nq, bsq, xq.vp, wq.vp, wsc.vp, yq.vp))
            if rows_q == 1:
                hbm("MatMulNBits decode", f"1x{kq}x{nq} q4/{bsq}", fn, nq * kq / 2 + 4.0 * (nq * kq // bsq + kq + nq))
            else:
                mfma("MatMulNBits prefill", f"{rows_q}x{kq}x{nq} q4/{bsq}", fn, 2.0 * rows_q * kq * nq, F32_PEAK_TF, "TFLOP/s")

    # ---- ReduceSum through strides and Einsum (src/ops/reduce.rs, src/ops/einsum.rs): a contiguous last-axis sum, a column sum over
    # the strided axis (nothing packed), and the two attention products written as Einsum on un-transposed [B, S, H, D]
    # projections (one strided two-level batched GEMM each; the second also pays the output permutation copy)
    from rten_amd import ops as _ops
    ctx.enable_pool(True)  # operator-level rows: outputs and intermediates come from the buffer pool, as under a graph executor
    xr = dev(rng.standard_normal((32 * 12 * 128, 128), dtype=np.float32))
    rs_last, rs_first = _ops.ReduceSum(axes=[1], keep_dims=False), _ops.ReduceSum(axes=[0], keep_dims=False)

    def graphed(fn):  # run once (fills the pool), capture one evaluation into a hipGraph, time replays of it
        fn()  # results return to the pool at once, so the captured evaluation allocates nothing
        fn()
        ctx.sync()
        ctx.graph_begin()
        keep = fn()  # held by the closure: the graph's output buffer is not handed out again while it is replayed
        g = ctx.graph_end()
        return lambda keep=keep: ctx.graph_launch(g)
    # (these two straight through the C entry, like the other single-kernel rows: the replay of a ONE-kernel hipGraph has a ~9.6 us floor of its
    # own, which is what rounds 2-3 reported for the last-axis sum whatever its kernel did)
    i64 = lambda *v: (C.c_int64 * len(v))(*v)
    yr = empty((49152,))
    hbm("ReduceSum last axis", "49152x128", (lambda: ctx.call("rten_hip_reduce_sum_strided_f32", 1, i64(49152), i64(128), 1, i64(128), i64(1), xr.vp, yr.vp)),
        4.0 * (49152 * 128 + 49152))
    xc = dev(rng.standard_normal((4096, 3072), dtype=np.float32))
    yc = empty((3072,))
    hbm("ReduceSum strided axis", "4096x3072 -> 3072", (lambda: ctx.call("rten_hip_reduce_sum_strided_f32", 1, i64(3072), i64(1), 1, i64(4096), i64(3072), xc.vp, yc.vp)),
        4.0 * (4096 * 3072 + 3072))
    Be, Se, He, De = 32, 128, 12, 64
    qe, ke, ve = (dev(rng.standard_normal((Be, Se, He, De), dtype=np.float32)) for _ in range(3))
    pe = dev(rng.standard_normal((Be, He, Se, Se), dtype=np.float32))
    es, ec = _ops.Einsum("bqhd,bkhd->bhqk"), _ops.Einsum("bhqk,bkhd->bqhd")
    mfma("Einsum bqhd,bkhd->bhqk", f"{Be}x{Se}x{He}x{De}", graphed(lambda: es.run(ctx, [qe, ke])), 2.0 * Be * He * Se * Se * De, F32_PEAK_TF, "TFLOP/s")
    mfma("Einsum bhqk,bkhd->bqhd", f"{Be}x{He}x{Se}x{Se}", graphed(lambda: ec.run(ctx, [pe, ve])), 2.0 * Be * He * Se * Se * De, F32_PEAK_TF, "TFLOP/s")
    ctx.enable_pool(False)

    # ---- f32 GEMM on BERT-base shapes (batch 32 x 128 tokens)
    for (m, k, n, act, name) in ((4096, 768, 768, 0, "MatMul proj"), (4096, 768, 3072, L.ACT_GELU, "MatMul FFN1 + Gelu"), (4096, 3072, 768, 0, "MatMul FFN2")):
        a, w, bias, out = dev(rng.standard_normal((m, k), dtype=np.float32)), dev(rng.standard_normal((k, n), dtype=np.float32)), dev(np.zeros(n, np.float32)), empty((m, n))
        d = L.gemm_desc(m, n, k, k, 1, n, 1, n, bias_kind=L.BIAS_PER_COL, act=act)
        mfma(name, f"{m}x{k}x{n}", (lambda: ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, w.vp, bias.vp, out.vp)), 2.0 * m * k * n, F32_PEAK_TF, "TFLOP/s")
    # ---- few rows against a big weight matrix (ResNet-50's classifier, Gemm transB = 1; an LLM-decoder projection at 16 rows): bound by streaming B
    for (m, k, n, trans_b, name) in ((32, 2048, 1000, True, "Gemm classifier"), (16, 4096, 4096, False, "MatMul 16 rows")):
        a, w, bias, out = dev(rng.standard_normal((m, k), dtype=np.float32)), dev(rng.standard_normal((n, k) if trans_b else (k, n), dtype=np.float32)), dev(np.zeros(n, np.float32)), empty((m, n))
        d = L.gemm_desc(m, n, k, k, 1, 1 if trans_b else n, k if trans_b else 1, n, bias_kind=L.BIAS_PER_COL)
        hbm(name, f"{m}x{k}x{n}", (lambda: ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, w.vp, bias.vp, out.vp)), 4.0 * (m * k + k * n + m * n + n))
        ctx.set_gemm_variant(3)
        hbm(name + " (64x64 tiles, round 4)", f"{m}x{k}x{n}", (lambda: ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, w.vp, bias.vp, out.vp)), 4.0 * (m * k + k * n + m * n + n))
        ctx.set_gemm_variant(-1)
    # attention core, 12 heads x 64 on [B, S, H*D] projections (strided heads)
    B, S, H, D = 32, 128, 12, 64
    q, kk, v, o = (dev(rng.standard_normal((B, S, H * D), dtype=np.float32)) for _ in range(4))
    sd = L.SdpaDesc(B, H, S, S, D, D, S * H * D, D, H * D, S * H * D, D, H * D, S * H * D, D, H * D, S * H * D, D, H * D, 0, 0, 0.125, 0)
    mfma("sdpa (QK^T, softmax, PV)", f"b={B} h={H} s={S} d={D}", (lambda: ctx.call("rten_hip_sdpa_f32", C.byref(sd), q.vp, kk.vp, v.vp, None, o.vp)),
         4.0 * B * H * S * S * D, F32_PEAK_TF, "TFLOP/s")

    # the general one-kernel form (round 3): head 128 x 512 keys (2 query tiles per head), head 64 x 384 keys, head 32 x 256 keys
    for (B2, H2, S2, T2, D2) in ((8, 16, 256, 512, 128), (16, 12, 384, 384, 64), (32, 8, 256, 256, 32), (16, 12, 256, 256, 64), (8, 16, 256, 256, 128), (32, 8, 128, 128, 128),
                                 (32, 16, 128, 128, 32), (16, 12, 256, 200, 64)):
        q2, o2 = dev(rng.standard_normal((B2, H2, S2, D2), dtype=np.float32)), empty((B2, H2, S2, D2))
        k2, v2 = dev(rng.standard_normal((B2, H2, T2, D2), dtype=np.float32)), dev(rng.standard_normal((B2, H2, T2, D2), dtype=np.float32))
        sd2 = L.SdpaDesc(B2, H2, S2, T2, D2, D2, H2 * S2 * D2, S2 * D2, D2, H2 * T2 * D2, T2 * D2, D2, H2 * T2 * D2, T2 * D2, D2, H2 * S2 * D2, S2 * D2, D2, 0, 0,
                         float(1.0 / np.sqrt(D2)), 0)
        for path, label in ((2, "one kernel"), (1, "composed: GEMM, softmax, GEMM")):
            def fn(sd2=sd2, q2=q2, k2=k2, v2=v2, o2=o2):
                ctx.call("rten_hip_sdpa_f32", C.byref(sd2), q2.vp, k2.vp, v2.vp, None, o2.vp)
            ctx.call("rten_hip_set_sdpa_path", path)
            mfma(f"sdpa general ({label})", f"b={B2} h={H2} s={S2} t={T2} d={D2}", fn, 4.0 * B2 * H2 * S2 * T2 * D2, F32_PEAK_TF, "TFLOP/s")
            ctx.call("rten_hip_set_sdpa_path", 0)

    # ---- int8 GEMM / conv (u8 activations x i8 weights)
    for (m, k, n) in ((4096, 768, 768), (4096, 768, 3072)):
        a = dev(rng.integers(0, 255, (m, k)).astype(np.uint8)); w = dev(rng.integers(-127, 127, (k, n)).astype(np.int8))
        az, wz, out = dev(np.array(128, np.uint8)), dev(np.zeros(n, np.int8)), empty((m, n), np.int32)
        d = L.GemmInt8Desc(m, n, k, k, 1, n, 1, n, 0, 1, 1, n, 0)
        mfma("MatMulInteger", f"{m}x{k}x{n}", (lambda: ctx.call("rten_hip_gemm_int8", C.byref(d), a.vp, w.vp, az.vp, wz.vp, None, out.vp)), 2.0 * m * k * n, I8_PEAK_TOPS, "TOP/s")
        nb = ctx.lib.rten_hip_gemm_int8_packed_bytes(k, n)  # constant RHS staged once at load (Operator::prepack): the per-call staging of B disappears
        if nb:
            packed = empty((nb,), np.uint8)
            ctx.call("rten_hip_gemm_int8_prepack", k, n, w.vp, n, 1, 1, packed.vp)
            dp = L.GemmInt8Desc(m, n, k, k, 1, n, 1, n, 0, 1, 1, n, 0, 1, 0, 0, 0, 1)
            mfma("MatMulInteger (prepacked RHS)", f"{m}x{k}x{n}", (lambda: ctx.call("rten_hip_gemm_int8", C.byref(dp), a.vp, packed.vp, az.vp, wz.vp, None, out.vp)),
                 2.0 * m * k * n, I8_PEAK_TOPS, "TOP/s")
    for (o_, c_, hw, k_, s_, p_, name) in ((64, 64, 56, 3, 1, 1, "s0 3x3"), (256, 256, 14, 3, 1, 1, "s2 3x3"), (256, 64, 56, 1, 1, 0, "s0 1x1 expand")):
        xq = dev(rng.integers(0, 255, (32, c_, hw, hw)).astype(np.uint8)); wq = dev(rng.integers(-127, 127, (o_, c_, k_, k_)).astype(np.int8))
        xz, wz = dev(np.array(128, np.uint8)), dev(np.zeros(o_, np.int8))
        oh = (hw + 2 * p_ - k_) // s_ + 1
        out = empty((32, o_, oh, oh), np.int32)
        cd = L.Conv2dDesc(32, c_, hw, hw, o_, k_, k_, (C.c_int32 * 4)(p_, p_, p_, p_), s_, s_, 1, 1, 1, oh, oh)
        d = L.Conv2dInt8Desc(cd, 0, 1, o_, 1)
        mfma(f"ConvInteger {name}", f"32x{c_}x{hw}x{hw} -> {o_}, k{k_}",
             (lambda: ctx.call("rten_hip_conv2d_int8", C.byref(d), xq.vp, wq.vp, xz.vp, wz.vp, None, None, None, 0, out.vp)),
             2.0 * 32 * o_ * c_ * k_ * k_ * oh * oh, I8_PEAK_TOPS, "TOP/s")

    if args.cpu_baseline:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench  # the oracle is only ever touched through bench.py's cpu_baseline leg
        base = bench.cpu_op_baselines()
        for r in rows:
            key = r["op"] if r["op"] in base else f"{r['op']} {r['shape']}"
            if key in base:
                r["cpu_baseline"] = base[key]
                r["speedup_vs_cpu_port"] = round(base[key]["us"] / r["us"], 1)
    print(json.dumps({"device": ctx.device_info(), "reps": args.reps, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
