#!/usr/bin/env python3
"""rten_hip_conv2d_f32_pair against the two launches it replaces, on ResNet-50's stage-0 shapes at batch 32 (expand 64 -> 256 with residual + Relu, then the next
block's reduce 256 -> 64 / 256 -> 128), alone and under self-co-run (the same sequence on N streams at once: the default schedule's condition).

    python tools/probe_conv_pair.py [--lanes 4] [--batch 32]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--plan1", default="0,0,1,1", help="variant,split mode,K groups,order of the expand layer's own launch")
    ap.add_argument("--plan2", default="22,0,1,1")
    args = ap.parse_args()
    from rten_amd import lib as L
    from rten_amd.tensor import DeviceTensor
    N, H, W = args.batch, 56, 56
    rng = np.random.default_rng(7)
    for M2 in (tuple(int(t) for t in os.environ.get("PAIR_M2", "64,128").split(","))):
        ctxs = [L.Context(0) for _ in range(args.lanes)]
        sets = []
        d1 = L.Conv2dDesc(N, 64, H, W, 256, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, H, W)
        d2 = L.Conv2dDesc(N, 256, H, W, M2, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, H, W)
        for ctx in ctxs:
            x = DeviceTensor.from_numpy(ctx, np.maximum(rng.standard_normal((N, 64, H, W), dtype=np.float32), 0))
            r = DeviceTensor.from_numpy(ctx, np.maximum(rng.standard_normal((N, 256, H, W), dtype=np.float32), 0))
            w1 = DeviceTensor.from_numpy(ctx, rng.standard_normal((256, 64, 1, 1), dtype=np.float32) * 0.1)
            w2 = DeviceTensor.from_numpy(ctx, rng.standard_normal((M2, 256, 1, 1), dtype=np.float32) * 0.05)
            b1 = DeviceTensor.from_numpy(ctx, rng.standard_normal(256, dtype=np.float32))
            b2 = DeviceTensor.from_numpy(ctx, rng.standard_normal(M2, dtype=np.float32))
            p1 = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d1)) // 4,), np.float32)
            p2 = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d2)) // 4,), np.float32)
            ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d1), w1.vp, p1.vp)
            ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d2), w2.vp, p2.vp)
            y1, y2 = DeviceTensor(ctx, (N, 256, H, W), np.float32), DeviceTensor(ctx, (N, M2, H, W), np.float32)
            z1, z2 = DeviceTensor(ctx, (N, 256, H, W), np.float32), DeviceTensor(ctx, (N, M2, H, W), np.float32)
            sets.append(dict(ctx=ctx, x=x, r=r, p1=p1, p2=p2, b1=b1, b2=b2, y1=y1, y2=y2, z1=z1, z2=z2, keep=(w1, w2)))
        F = L.CONV_RELU | L.CONV_RESIDUAL

        def plan(ctx, spec):
            v, sm, kg, order = (int(t) for t in spec.split(","))
            ctx.call("rten_hip_set_gemm_variant_override", v)
            ctx.call("rten_hip_set_gemm_split", sm, kg)
            ctx.call("rten_hip_set_gemm_order", order)

        def separate(s):
            c = s["ctx"]
            plan(c, args.plan1)
            c.call("rten_hip_conv2d_f32", C.byref(d1), s["x"].vp, s["p1"].vp, 1, s["b1"].vp, s["r"].vp, F, s["z1"].vp)
            plan(c, args.plan2)
            c.call("rten_hip_conv2d_f32", C.byref(d2), s["z1"].vp, s["p2"].vp, 1, s["b2"].vp, None, L.CONV_RELU, s["z2"].vp)

        def pair(s):
            s["ctx"].call("rten_hip_conv2d_f32_pair", C.byref(d1), s["x"].vp, s["p1"].vp, s["b1"].vp, s["r"].vp, F, s["y1"].vp, C.byref(d2), s["p2"].vp, s["b2"].vp, L.CONV_RELU, s["y2"].vp)

        for s in sets:
            separate(s); pair(s)
            s["ctx"].sync()
        same = all(np.array_equal(s["y1"].numpy().view(np.int32), s["z1"].numpy().view(np.int32)) and np.array_equal(s["y2"].numpy().view(np.int32), s["z2"].numpy().view(np.int32)) for s in sets[:1])
        print(f"# M2 = {M2}: pair bit-identical to the two launches: {same}", flush=True)

        def measure(fn, streams):
            graphs = []
            use = sets[:streams]
            try:
                for s in use:
                    s["ctx"].graph_begin()
                    for _ in range(args.reps):
                        fn(s)
                    graphs.append((s["ctx"], s["ctx"].graph_end()))
                best = 1e30
                for _ in range(4):
                    t0 = time.perf_counter()
                    for c, g in graphs:
                        c.graph_launch(g)
                    for s in use:
                        s["ctx"].sync()
                    best = min(best, (time.perf_counter() - t0) / (args.reps * streams) * 1e6)
                return best
            finally:
                for c, g in graphs:
                    c.graph_destroy(g)

        fl = 2.0 * N * H * W * (64 * 256 + 256 * M2)
        for streams in sorted({1, args.lanes}):
            a, b = measure(separate, streams), measure(pair, streams)
            print(f"M2 {M2:4d}  streams {streams}:  two launches {a:7.1f} us ({fl / a / 1e6:6.1f} TF/s)   one launch {b:7.1f} us ({fl / b / 1e6:6.1f} TF/s)   {100 * (b / a - 1):+.1f} %", flush=True)
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
