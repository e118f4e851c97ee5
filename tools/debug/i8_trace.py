"""One launch of selected int8 conv layers of ResNet-50 (batch 32) with the kernel's cycle stamps printed: the per-phase table of DESIGN.md section 7.2.

Needs a library whose int8_fast.hip was compiled with -DRTEN_TRACE (the stamps are compiled out of the product build):
    cd rten_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DRTEN_TRACE -c int8_fast.hip -o /tmp/int8_trace.o
    cd .. && hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/trace.so $(ls _build/*.o | grep -v int8_fast.o) /tmp/int8_trace.o
    RTEN_HIP_LIBRARY=$PWD/_ab/trace.so python tools/debug/i8_trace.py s0b0c3 s1b1c2 s2b1c1 s3b1c2      (on the GPU box: through gpurun)
One workgroup (id 8) of every launch prints its cycle counts per phase (shader clock, s_memtime)."""
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
