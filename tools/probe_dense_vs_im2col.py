#!/usr/bin/env python3
"""Upper bound for cheaper im2col gathers: time each 3x3 ResNet layer against a 1x1 conv of the same GEMM shape
(M = O, K = 9C, N = batch*H*W) whose B operand is a dense 16-byte-DMA matrix."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor

ctx = L.Context(0)
rng = np.random.default_rng(0)
def run(desc, x, w, b, y, plans, reps=5):
    best = (1e9, None)
    for (v, m, g) in plans:
        ctx.set_gemm_variant(v); ctx.call("rten_hip_set_gemm_split", m, g)
        for _ in range(2):
            ctx.call("rten_hip_conv2d_f32", C.byref(desc), x.vp, w.vp, 1, b.vp, None, L.CONV_RELU, y.vp)
        ctx.timer_start(1)
        for _ in range(reps):
            ctx.call("rten_hip_conv2d_f32", C.byref(desc), x.vp, w.vp, 1, b.vp, None, L.CONV_RELU, y.vp)
        ctx.timer_stop(1)
        t = ctx.timer_ms(1) / reps * 1e3
        if t < best[0]: best = (t, (v, m, g))
    ctx.set_gemm_variant(-1); ctx.call("rten_hip_set_gemm_split", 3, 1)
    return best
for (O, Cc, H) in ((64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)):
    N = 32
    nblk = (Cc * 9 + 255) // 256
    plans = [(v, 0, 1) for v in (0, 1, 2, 3, 9, 11)] + [(v, m, g) for v in (1, 3) for m in (1, 2) for g in sorted({2, 3, nblk})]
    out = []
    for (c, k, pad) in ((Cc, 3, 1), (Cc * 9, 1, 0)):
        d = L.Conv2dDesc(N, c, H, H, O, k, k, (C.c_int32 * 4)(pad, pad, pad, pad), 1, 1, 1, 1, 1, H, H)
        x = DeviceTensor.from_numpy(ctx, rng.standard_normal((N, c, H, H), dtype=np.float32))
        w = DeviceTensor.from_numpy(ctx, rng.standard_normal((O, c, k, k), dtype=np.float32))
        b = DeviceTensor.from_numpy(ctx, np.zeros(O, np.float32))
        packed = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d)) // 4,), np.float32)
        ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d), w.vp, packed.vp)
        y = DeviceTensor(ctx, (N, O, H, H), np.float32)
        out.append(run(d, x, packed, b, y, plans))
    fl = 2.0 * O * Cc * 9 * H * H * N
    print(f"O={O} C={Cc} H={H}: 3x3 im2col {out[0][0]:6.1f} us {out[0][1]} ({fl/out[0][0]/1e6:5.1f} TF/s) | dense 1x1, K=9C {out[1][0]:6.1f} us {out[1][1]} ({fl/out[1][0]/1e6:5.1f} TF/s)", flush=True)
