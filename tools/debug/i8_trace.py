"""One launch of selected int8 conv layers of ResNet-50 (batch 32) with the kernel's cycle stamps printed (debug build of int8_fast.hip only)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rten_amd import lib as L
from rten_amd.workloads import resnet50_int8
ctx = L.Context(0)
net = resnet50_int8.ResNet50Int8(ctx, 32)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((32, 3, 224, 224), dtype=np.float32))
net.forward(); ctx.sync()
for name in sys.argv[1:]:
    l = next(s for s in net.specs if s["name"] == name)
    d = net.idesc[name]
    src, st = net._act(l["src"]), net.stats.get(l["src"])
    ctx.call("rten_hip_dynamic_quantize_linear_staged_stats" if st is not None else "rten_hip_dynamic_quantize_linear_staged", C.byref(d), src.vp, *((st,) if st is not None else ()),
             net.staged.vp, net.xs.vp, net.xz.vp, net.ws[name].vp, net.sc.vp)
    flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
    args = (C.byref(d), net.staged.vp, net.wq[name].vp, net.xz.vp, None, net.sc.vp, net.bq[name].vp, net._act(l["res"]).vp if l["res"] else None, flags, net._act(l["dst"]).vp)
    for _ in range(3):
        ctx.call("rten_hip_conv2d_int8_stats", *args, net.stats[l["dst"]])
    ctx.sync()
    print("----", name, flush=True)
    ctx.timer_start(1)
    ctx.call("rten_hip_conv2d_int8_stats", *args, net.stats[l["dst"]])
    ctx.timer_stop(1)
    print(f"{name}: launch {ctx.timer_ms(1)*1e3:.1f} us", flush=True)
