#!/usr/bin/env python3
"""How much of each ResNet-50 layer's time is tile quantisation?  Dense GEMMs of the layers' (M, K) with the column count
swept around the layer's own N (in units of one round of 256 CUs x 64 columns), 64x64 LDS-DMA tiles, no split-K: the time
per column count is a staircase if whole rounds are the granule.  Prints us and TF/s per point (GPU box only)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor

ctx = L.Context(0)
rng = np.random.default_rng(0)
LAYERS = [("s0 3x3", 64, 576, 100352), ("s0 c1", 64, 256, 100352), ("s1 c1", 128, 256, 100352), ("s1 3x3", 128, 1152, 25088),
          ("s2 3x3", 256, 2304, 6272), ("s2 c3", 1024, 256, 6272), ("s3 3x3", 512, 4608, 1568), ("s3 c3", 2048, 512, 1568)]
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bm, bn = {0: (128, 128), 1: (128, 64), 2: (64, 128), 3: (64, 64)}[variant & 3]
for name, m, k, n in LAYERS:
    tiles_m = (m + bm - 1) // bm
    per_round = 256 // tiles_m * bn if tiles_m <= 256 else bn  # columns that make one round of 256 tiles
    base = n // per_round
    a = DeviceTensor.from_numpy(ctx, rng.standard_normal((k, m), dtype=np.float32))  # k-major A (prepacked conv weights)
    nmax = (base + 2) * per_round
    w = DeviceTensor.from_numpy(ctx, rng.standard_normal((k, nmax), dtype=np.float32))
    out = DeviceTensor(ctx, (m, nmax), np.float32)
    pts = sorted({max(per_round, (base - 1) * per_round), base * per_round, n, (base + 1) * per_round, base * per_round + per_round // 8,
                  base * per_round + per_round // 4, base * per_round + per_round // 2})
    row = []
    for nn in pts:
        d = L.gemm_desc(m, nn, k, 1, m, nmax, 1, nmax)
        ctx.set_gemm_variant(variant); ctx.call("rten_hip_set_gemm_split", 0, 1)
        f = lambda: ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, w.vp, None, out.vp)
        for _ in range(3): f()
        ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_start(3)
            for _ in range(8): f()
            ctx.timer_stop(3)
            best = min(best, ctx.timer_ms(3) / 8 * 1e3)
        row.append((nn, best))
    ctx.set_gemm_variant(-1); ctx.call("rten_hip_set_gemm_split", 3, 1)
    print(f"{name:7s} M={m:4d} K={k:4d} tile {bm}x{bn} round={per_round} cols: " +
          "  ".join(f"N={nn}({nn / per_round:.3f}r) {t:6.1f}us {2.0 * m * k * nn / t / 1e6:5.1f}TF" + ("*" if nn == n else "") for nn, t in row), flush=True)
