#!/usr/bin/env python3
"""Per-layer workgroup tiles of the dynamically quantized ResNet-50 chosen UNDER SELF-CO-RUN (round 6; tools/tune_corun.py is the f32 counterpart): every conv layer's
int8 launch -- the convolution on its staged input, or the quantize-on-load form for the layers the base plan lists under "fused_dql" -- on N streams at once, for the
backend's per-shape rule and the four tiles (rten_hip_set_int8_tile), microseconds per launch over all streams.  Writes the base plan + a "<layer>": [tile, 0, 1, 0]
entry for every layer whose best tile beats the rule by the margin.
    python tools/tune_corun_int8.py [--lanes 4] [--plan profiles/plans/int8_lanes.json] [--out profiles/plans/experiments/int8_lanes_tiles.json]"""
import argparse
import ctypes as C
import json
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
    ap.add_argument("--margin", type=float, default=0.02)
    ap.add_argument("--plan", default=os.path.join(ROOT, "profiles", "plans", "int8_lanes.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "plans", "experiments", "int8_lanes_tiles.json"))
    args = ap.parse_args()
    from rten_amd import lib as L
    from rten_amd.workloads import resnet50, resnet50_int8
    base = json.load(open(args.plan))
    dql_layers = set(base.get("fused_dql", []))
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
        net.forward()
        ctx.sync()
        nets.append(net)
    specs, descs = nets[0].specs, nets[0].descs

    def launch(net, l, form):
        name, d = l["name"], net.idesc[l["name"]]
        src, (staged, xs, xz) = net._act(l["src"]), net.qsets[0]
        st = net.stats.get(l["src"])
        flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
        res = net._act(l["res"]).vp if l["res"] else None
        c = net.ctx
        if form == "stage":
            if st is not None:
                c.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), src.vp, st, staged.vp, xs.vp, xz.vp, net.ws[name].vp, net.sc.vp)
            else:
                c.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), src.vp, staged.vp, xs.vp, xz.vp, net.ws[name].vp, net.sc.vp)
        elif form == "dql":
            c.call("rten_hip_conv2d_int8_dql", C.byref(d), src.vp, st, net.wq[name].vp, net.ws[name].vp, net.bq[name].vp, res, flags, net._act(l["dst"]).vp, net.stats[l["dst"]], None, None)
        else:
            c.call("rten_hip_conv2d_int8_stats", C.byref(d), staged.vp, net.wq[name].vp, xz.vp, None, net.sc.vp, net.bq[name].vp, res, flags, net._act(l["dst"]).vp, net.stats[l["dst"]])

    def measure(l, form, tile):
        graphs = []
        try:
            for net in nets:
                net.ctx.call("rten_hip_set_int8_tile", tile, None)
                launch(net, l, "stage")  # this layer's codes in the staged buffer (and a warm run of the timed form: scratch growth outside the capture)
                launch(net, l, form)
            for c in ctxs:
                c.sync()
            for net in nets:
                net.ctx.graph_begin()
                for _ in range(args.reps):
                    launch(net, l, form)
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
            for net in nets:
                net.ctx.call("rten_hip_set_int8_tile", -1, None)

    fams = {}
    for l in specs:
        d = descs[l["name"]]
        fams.setdefault((d.o, d.c, d.kh, d.stride_h, d.h, bool(l["res"]), l["name"] in dql_layers), []).append(l)
    out = dict(base)
    t_rule = t_best = 0.0
    names = {-1: "rule", 0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64"}
    print(f"# {args.lanes} streams, base plan {os.path.relpath(args.plan, ROOT)}; us per launch over all streams: the backend's rule and the four tiles")
    for key, members in sorted(fams.items(), key=lambda kv: -len(kv[1])):
        l = members[1] if len(members) > 1 else members[0]
        form = "dql" if key[6] else "conv"
        row = {t: min(measure(l, form, t), measure(l, form, t)) for t in (-1, 0, 1, 2, 3)}
        bt = min((v, t) for t, v in row.items() if t >= 0)
        pick = bt[1] if bt[0] < row[-1] * (1 - args.margin) else -1
        t_rule += row[-1] * len(members)
        t_best += (row[pick]) * len(members)
        tag = f"O{key[0]} C{key[1]} k{key[2]} s{key[3]} {key[4]}x{key[4]}{' +res' if key[5] else ''}{' (quantize on load)' if key[6] else ''} x{len(members)}"
        print(f"{tag:52s} {l['name']:7s} " + "  ".join(f"{names[t]} {row[t]:5.1f}" for t in (-1, 0, 1, 2, 3)) + f"  -> {names[pick]}", flush=True)
        for m in members:
            if pick >= 0:
                out[m["name"]] = [pick, 0, 1, 0]
            else:
                out.pop(m["name"], None)
    print(f"# sums over the layers (us): rule {t_rule:.0f}, chosen {t_best:.0f}")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"))


if __name__ == "__main__":
    main()
