#!/usr/bin/env python3
"""int8 ResNet-50, batch 32: the integer convolution launches that the cross-workgroup K split (KS kernels, csrc/int8_fast.hip) may take, timed stand-alone
under the RTEN_I8_KS setting of this process ("<parts>[,<tile>]"; unset = the dispatcher's rule; "0" = off), with a hash of every output so that the
settings can be compared bit for bit.   for s in 0 "" 2,3 4,3 ...; do RTEN_I8_KS=$s python tools/probe_int8_ks.py; done"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L
from rten_amd.workloads import resnet50_int8

ctx = L.Context(0)
net = resnet50_int8.ResNet50Int8(ctx, 32)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((32, 3, 224, 224), dtype=np.float32))
net.forward()
ctx.sync()
REPS = 30
LAYERS = os.environ.get("LAYERS", "s1b1c1,s1b1c2,s2b0c1,s2b0c2,s2b1c1,s2b1c2,s3b0c1,s3b0c2,s3b1c1,s3b1c2").split(",")
setting = os.environ.get("RTEN_I8_KS", "(rule)")
total = 0.0
row = []
for l in net.specs:
    name = l["name"]
    if name not in LAYERS:
        continue
    d = net.idesc[name]
    cv = d.conv
    src = net._act(l["src"])
    st = net.stats.get(l["src"])
    ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), src.vp, st, net.staged.vp, net.xs.vp, net.xz.vp, net.ws[name].vp, net.sc.vp)
    flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
    args = (C.byref(d), net.staged.vp, net.wq[name].vp, net.xz.vp, None, net.sc.vp, net.bq[name].vp, net._act(l["res"]).vp if l["res"] else None, flags, net._act(l["dst"]).vp)
    f = lambda: ctx.call("rten_hip_conv2d_int8_stats", *args, net.stats[l["dst"]])
    f()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start(1)
        for _ in range(REPS):
            f()
        ctx.timer_stop(1)
        best = min(best, ctx.timer_ms(1) / REPS * 1e3)
    sha = hashlib.sha256(net._act(l["dst"]).numpy().tobytes()).hexdigest()[:10]
    ops = 2.0 * cv.o * cv.c * cv.kh * cv.kw * cv.out_h * cv.out_w * cv.n
    total += best
    row.append(f"{name} {best:5.1f}us {ops / best / 1e6:4.0f}T {sha}")
print(f"KS={setting:7s} sum {total:6.1f} us | " + " | ".join(row), flush=True)
