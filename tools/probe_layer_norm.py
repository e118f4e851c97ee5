#!/usr/bin/env python3
"""LayerNormalization / Add+LayerNormalization at BERT-base's size ([4096, 768]) and a few others, stand-alone, under this process's RTEN_LN_ROWS setting
(rows per wave of the streaming form; 0 = one row per wave; unset = the launcher's rule), beside a copy of the same bytes.
    for r in 0 "" 2 3 4 8; do RTEN_LN_ROWS=$r python tools/probe_layer_norm.py; done"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor

ctx = L.Context(0)
rng = np.random.default_rng(0)
out = []
for rows, cols in ((4096, 768), (4096, 1024), (16384, 768), (4096, 512)):
    x, r = (DeviceTensor.from_numpy(ctx, rng.standard_normal((rows, cols), dtype=np.float32)) for _ in range(2))
    g, b = (DeviceTensor.from_numpy(ctx, rng.standard_normal(cols, dtype=np.float32)) for _ in range(2))
    y = DeviceTensor(ctx, (rows, cols), np.float32)

    def timed(f, reps=50):
        f(); ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_start(1)
            for _ in range(reps):
                f()
            ctx.timer_stop(1)
            best = min(best, ctx.timer_ms(1) / reps * 1e3)
        return best
    ln = timed(lambda: ctx.call("rten_hip_layer_norm_f32", rows, cols, x.vp, g.vp, b.vp, 1.0, 0.0, 1e-12, y.vp))
    aln = timed(lambda: ctx.call("rten_hip_add_layer_norm_f32", rows, cols, x.vp, r.vp, g.vp, b.vp, 1.0, 0.0, 1e-12, y.vp))
    cp = timed(lambda: ctx.call("rten_hip_memcpy_d2d", y.vp, x.vp, rows * cols * 4))
    by = rows * cols * 4
    out.append(f"{rows}x{cols}: LN {ln:5.2f} us ({2 * by / ln / 1e3 / 8000:.3f} of 8 TB/s) Add+LN {aln:5.2f} us ({3 * by / aln / 1e3 / 8000:.3f}) copy {cp:5.2f} us ({2 * by / cp / 1e3 / 8000:.3f})")
print(f"RTEN_LN_ROWS={os.environ.get('RTEN_LN_ROWS', '(rule)'):6s} | " + " | ".join(out), flush=True)
