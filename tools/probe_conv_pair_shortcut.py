#!/usr/bin/env python3
"""rten_hip_conv2d_f32_pair_shortcut against what it replaces (the shortcut layer's own launch + rten_hip_conv2d_f32_pair reading its output as the residual) on the
shapes of ResNet-50's first block at batch 32, alone and under four-stream self-co-run; outputs compared bit for bit.      python tools/probe_conv_pair_shortcut.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from rten_amd import lib as L  # noqa: E402
from rten_amd.tensor import DeviceTensor  # noqa: E402

N, H, W, LANES, REPS = 32, 56, 56, 4, 12
rng = np.random.default_rng(7)
mk = lambda c, o: L.Conv2dDesc(N, c, H, W, o, 1, 1, (C.c_int32 * 4)(0, 0, 0, 0), 1, 1, 1, 1, 1, H, W)  # noqa: E731
d1, ds, d2 = mk(64, 256), mk(64, 256), mk(256, 64)
sets = []
for _ in range(LANES):
    ctx = L.Context(0)
    f = lambda *s, k=1.0: DeviceTensor.from_numpy(ctx, (rng.standard_normal(s, dtype=np.float32) * k))  # noqa: E731
    s = dict(ctx=ctx, x=f(N, 64, H, W), xd=f(N, 64, H, W), b1=f(256), bd=f(256), b2=f(64))
    for name, d, k in (("p1", d1, 0.1), ("pd", ds, 0.1), ("p2", d2, 0.05)):
        w = f(d.o, d.c, 1, 1, k=k)
        s[name] = DeviceTensor(ctx, (ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d)) // 4,), np.float32)
        ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d), w.vp, s[name].vp)
    for n, c in (("r", 256), ("y1", 256), ("z1", 256), ("y2", 64), ("z2", 64)):
        s[n] = DeviceTensor(ctx, (N, c, H, W), np.float32)
    sets.append(s)


def two(s):
    c = s["ctx"]
    c.call("rten_hip_set_gemm_variant_override", 0); c.call("rten_hip_set_gemm_split", 0, 1); c.call("rten_hip_set_gemm_order", 0)
    c.call("rten_hip_conv2d_f32", C.byref(ds), s["xd"].vp, s["pd"].vp, 1, s["bd"].vp, None, 0, s["r"].vp)
    c.call("rten_hip_conv2d_f32_pair", C.byref(d1), s["x"].vp, s["p1"].vp, s["b1"].vp, s["r"].vp, L.CONV_RELU | L.CONV_RESIDUAL, s["z1"].vp, C.byref(d2), s["p2"].vp, s["b2"].vp, L.CONV_RELU, s["z2"].vp)


def one(s):
    s["ctx"].call("rten_hip_conv2d_f32_pair_shortcut", C.byref(d1), s["x"].vp, s["p1"].vp, s["b1"].vp, C.byref(ds), s["xd"].vp, s["pd"].vp, s["bd"].vp, L.CONV_RELU, s["y1"].vp,
                  C.byref(d2), s["p2"].vp, s["b2"].vp, L.CONV_RELU, s["y2"].vp)


for s in sets:
    two(s); one(s)
    s["ctx"].sync()
s = sets[0]
print("# bit-identical to the shortcut's own launch + the pair:", np.array_equal(s["y1"].numpy().view(np.int32), s["z1"].numpy().view(np.int32)) and np.array_equal(s["y2"].numpy().view(np.int32), s["z2"].numpy().view(np.int32)), flush=True)


def measure(fn, streams):
    graphs, use = [], sets[:streams]
    try:
        for s in use:
            s["ctx"].graph_begin()
            for _ in range(REPS):
                fn(s)
            graphs.append((s["ctx"], s["ctx"].graph_end()))
        best = 1e30
        for _ in range(4):
            t0 = time.perf_counter()
            for c, g in graphs:
                c.graph_launch(g)
            for s in use:
                s["ctx"].sync()
            best = min(best, (time.perf_counter() - t0) / (REPS * streams) * 1e6)
        return best
    finally:
        for c, g in graphs:
            c.graph_destroy(g)


for streams in (1, LANES):
    a, b = measure(two, streams), measure(one, streams)
    print(f"streams {streams}:  shortcut launch + pair {a:7.1f} us   one launch {b:7.1f} us   {100 * (b / a - 1):+.1f} %", flush=True)
