#!/usr/bin/env python3
"""The dynamically quantized ResNet-50 layer by layer UNDER SELF-CO-RUN (round 6): every conv layer's launch sequence (its DynamicQuantizeLinear staging launch +
ConvIntegerToFloat, or the quantize-on-load form) on N streams at once, as tools/tune_corun.py does for the f32 graph -- microseconds per layer over all streams,
the layer's algorithmic HBM bytes over that, and where the 4-lane step's time sits.  The sum predicts the 4-lane step of `python bench.py --config int8`.

    python tools/probe_int8_corun.py [--lanes 4]
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
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from rten_amd import lib as L
    from rten_amd.workloads import resnet50, resnet50_int8
    weights = resnet50.make_weights()
    ctxs = [L.Context(0) for _ in range(args.lanes)]
    nets = []
    for i, ctx in enumerate(ctxs):
        kw = {} if i == 0 else dict(i8_arena_ptr=nets[0].i8_arena.ptr, i8_arena_keepalive=nets[0].i8_arena)
        net = resnet50_int8.ResNet50Int8(ctx, args.batch, weights, **kw)
        if i == 0:
            net.upload_weights()
            ctx.sync()
        net.x.upload(np.random.default_rng(1234 + i).random((args.batch, 3, 224, 224), dtype=np.float32))
        net.forward()  # activations and the statistics block of every tensor exist (buffers are reused: a layer's input holds SOME activation of its size)
        ctx.sync()
        nets.append(net)
    specs, descs = nets[0].specs, nets[0].descs

    def sequence(net, l, form):
        name, d = l["name"], net.idesc[l["name"]]
        src, (staged, xs, xz) = net._act(l["src"]), net.qsets[0]
        st = net.stats.get(l["src"])
        flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
        res = net._act(l["res"]).vp if l["res"] else None
        c = net.ctx
        if form == "dql":
            c.call("rten_hip_conv2d_int8_dql", C.byref(d), src.vp, st, net.wq[name].vp, net.ws[name].vp, net.bq[name].vp, res, flags, net._act(l["dst"]).vp, net.stats[l["dst"]], None, None)
            return
        if form in ("both", "quant"):
            if st is not None:
                c.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), src.vp, st, staged.vp, xs.vp, xz.vp, net.ws[name].vp, net.sc.vp)
            else:
                c.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), src.vp, staged.vp, xs.vp, xz.vp, net.ws[name].vp, net.sc.vp)
        if form in ("both", "conv"):
            c.call("rten_hip_conv2d_int8_stats", C.byref(d), staged.vp, net.wq[name].vp, xz.vp, None, net.sc.vp, net.bq[name].vp, res, flags, net._act(l["dst"]).vp, net.stats[l["dst"]])

    def measure(l, form):
        graphs = []
        try:
            for net in nets:
                sequence(net, l, "both" if form != "dql" else "dql")
            for c in ctxs:
                c.sync()
            for net in nets:
                net.ctx.graph_begin()
                for _ in range(args.reps):
                    sequence(net, l, form)
                graphs.append((net.ctx, net.ctx.graph_end()))
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                for c, g in graphs:
                    c.graph_launch(g)
                for c in ctxs:
                    c.sync()
                best = min(best, (time.perf_counter() - t0) / (args.reps * len(nets)) * 1e6)
            return best
        finally:
            for c, g in graphs:
                c.graph_destroy(g)

    fams = {}
    for l in specs:
        d = descs[l["name"]]
        fams.setdefault((d.o, d.c, d.kh, d.stride_h, d.h, bool(l["res"])), []).append(l)
    tot = {"quant": 0.0, "conv": 0.0, "both": 0.0, "best": 0.0}
    tot_bytes = 0.0
    print(f"# {args.lanes} streams; us per layer over all streams: staging launch alone / convolution alone / both in sequence / quantize-on-load form (pointwise stride-1 layers); GB/s = algorithmic bytes of the pair over `both`")
    for key, members in sorted(fams.items(), key=lambda kv: -len(kv[1])):
        l = members[1] if len(members) > 1 else members[0]
        d = descs[l["name"]]
        in_e, out_e = d.n * d.c * d.h * d.w, d.n * d.o * d.out_h * d.out_w
        by = 5.0 * in_e + 1.0 * in_e + d.o * d.c * d.kh * d.kw + 4.0 * out_e + (4.0 * out_e if l["res"] else 0.0)
        q, cv, both = measure(l, "quant"), measure(l, "conv"), measure(l, "both")
        cvd = d
        dql_ok = cvd.kh == 1 and cvd.kw == 1 and cvd.stride_h == 1 and not any(cvd.pads) and cvd.c % 64 == 0 and nets[0].stats.get(l["src"]) is not None
        dql = measure(l, "dql") if dql_ok else None
        best = min(both, dql) if dql is not None else both
        n = len(members)
        tot["quant"] += q * n; tot["conv"] += cv * n; tot["both"] += both * n; tot["best"] += best * n
        tot_bytes += by * n
        tag = f"O{key[0]} C{key[1]} k{key[2]} s{key[3]} {key[4]}x{key[4]}{' +res' if key[5] else ''} x{n}"
        print(f"{tag:36s} {l['name']:7s} quant {q:6.1f}  conv {cv:6.1f}  both {both:6.1f}  dql {('%6.1f' % dql) if dql is not None else '     -'} | {by / 1e6:7.1f} MB  {by / both / 1e3:6.0f} GB/s"
              f" | 2MNK {2.0 * d.o * d.c * d.kh * d.kw * d.n * d.out_h * d.out_w / cv / 1e6:6.1f} TOP/s", flush=True)
    print(f"# sums over the 53 layers (us): staging {tot['quant']:.0f}, convolutions {tot['conv']:.0f}, pairs in sequence {tot['both']:.0f}, best form per layer {tot['best']:.0f}; "
          f"algorithmic bytes {tot_bytes / 1e9:.2f} GB -> {tot_bytes / tot['both'] / 1e6:.2f} TB/s over the pairs")


if __name__ == "__main__":
    main()
