#!/usr/bin/env python3
"""int8 ResNet-50, batch 32: per layer, the DynamicQuantizeLinear staging kernel and the integer conv kernel timed separately
(HIP-event timers around 20 back-to-back launches), with the HBM bytes each one has to move and the rate that is."""
import ctypes as C
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
REPS = 20


def timed(fn):
    fn()
    ctx.timer_start(1)
    for _ in range(REPS):
        fn()
    ctx.timer_stop(1)
    return ctx.timer_ms(1) / REPS * 1e3


tq = tc = tf = tsep = tqo = tqo_sep = 0.0
print(f"{'layer':9s} {'O':>4s} {'C':>4s} k s {'HxW':>7s} | {'quant us':>8s} {'GB/s':>6s} | {'conv us':>8s} {'GB/s':>6s} {'TOP/s':>6s}  tiles(128x128)")
ONLY = os.environ.get("LAYERS")
for l in net.specs:
    name = l["name"]
    if ONLY and name not in ONLY.split(","):
        continue
    d = net.idesc[name]
    cv = d.conv
    src = net._act(l["src"])
    st = net.stats.get(l["src"])
    in_elems = cv.n * cv.c * cv.h * cv.w
    out_elems = cv.n * cv.o * cv.out_h * cv.out_w
    staged_bytes = ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(d))
    if st is not None:
        q = lambda: ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), src.vp, st, net.staged.vp, net.xs.vp, net.xz.vp, net.ws[name].vp, net.sc.vp)
    else:
        q = lambda: ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), src.vp, net.staged.vp, net.xs.vp, net.xz.vp, net.ws[name].vp, net.sc.vp)
    us_q = timed(q)
    flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
    args = (C.byref(d), net.staged.vp, net.wq[name].vp, net.xz.vp, None, net.sc.vp, net.bq[name].vp, net._act(l["res"]).vp if l["res"] else None, flags,
            net._act(l["dst"]).vp)
    us_c = timed(lambda: ctx.call("rten_hip_conv2d_int8_stats", *args, net.stats[l["dst"]]))
    qb = in_elems * 4 + staged_bytes
    cb = staged_bytes + out_elems * 4 * (2 if l["res"] else 1) + cv.o * cv.c * cv.kh * cv.kw
    ops = 2.0 * cv.o * cv.c * cv.kh * cv.kw * cv.out_h * cv.out_w * cv.n
    tiles = ((cv.o + 127) // 128) * ((cv.n * cv.out_h * cv.out_w + 127) // 128)
    tq += us_q
    tc += us_c
    fused = ""
    if st is not None and cv.kh == 1 and cv.stride_h == 1 and not any(cv.pads) and cv.c % 64 == 0:
        us_f = timed(lambda: ctx.call("rten_hip_conv2d_int8_dql", C.byref(d), src.vp, st, net.wq[name].vp, net.ws[name].vp, net.bq[name].vp,
                                      net._act(l["res"]).vp if l["res"] else None, flags, net._act(l["dst"]).vp, net.stats[l["dst"]], None, None))
        tf += us_f
        tsep += us_q + us_c
        fused = f" | fused quantize+conv {us_f:6.1f} us (separate {us_q + us_c:6.1f})"
    nxt = net.qout_next.get(name)
    if nxt is not None:  # quantized-output launch (conv + the consumer's DynamicQuantizeLinear in one launch) vs the two launches
        nd = net.idesc[nxt["name"]]
        ost, oxs, oxz = net.qsets[1]
        qo = lambda: ctx.lib.rten_hip_conv2d_int8_qout(ctx.h, C.byref(d), net.staged.vp, net.wq[name].vp, net.xz.vp, None, net.sc.vp, net.bq[name].vp,
                                                       net._act(l["res"]).vp if l["res"] else None, flags, net._act(l["dst"]).vp if name in net.qout_keeps_f32 else None,
                                                       net.stats[l["dst"]], net.syncs[name], C.byref(nd),
                                                       ost.vp, oxs.vp, oxz.vp, net.ws[nxt["name"]].vp, net.scs[1].vp)
        if qo() == 0:
            us_qo = timed(qo)
            us_nq = timed(lambda: ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(nd), net._act(l["dst"]).vp, net.stats[l["dst"]], ost.vp, oxs.vp,
                                           oxz.vp, net.ws[nxt["name"]].vp, net.scs[1].vp))
            tqo += us_qo
            tqo_sep += us_c + us_nq
            fused += f" | qout {us_qo:6.1f} us (conv + consumer's quantize {us_c + us_nq:6.1f})"
        else:
            fused += " | qout: grid not resident at once"
    print(f"{name:9s} {cv.o:4d} {cv.c:4d} {cv.kh} {cv.stride_h} {cv.h:3d}x{cv.w:<3d} | {us_q:8.1f} {qb / us_q / 1e3:6.0f} | {us_c:8.1f} {cb / us_c / 1e3:6.0f} {ops / us_c / 1e6:6.0f}  {tiles}{fused}")
print(f"sum: quantize {tq / 1e3:.3f} ms, conv {tc / 1e3:.3f} ms; pointwise layers fused {tf / 1e3:.3f} ms vs separate {tsep / 1e3:.3f} ms; "
      f"quantized-output launches {tqo / 1e3:.3f} ms vs conv + consumer's quantize {tqo_sep / 1e3:.3f} ms; time-outs {net.qout_timeouts()}")
