#!/usr/bin/env python3
"""Run selected ResNet-50 conv layers in isolation (for rocprofv3 --pmc / per-variant A/B timing).

    python tools/layer_probe.py --layers s0b0c2,s2b1c2 --variants 0,3 --reps 5
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib  # noqa: E402
from rten_amd.workloads import resnet50  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="s0b0c2,s0b0c3,s2b1c1,s2b1c2,s3b1c2")
    ap.add_argument("--variants", default="0,3", help="comma list of variant or variant:splitmode:groups")
    ap.add_argument("--plan", default=None, help="JSON plan file written by bench.py --save-plan: probe each layer with its tuned plan")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    ctx = lib.Context(0)
    net = resnet50.ResNet50(ctx, args.batch)
    net.upload_weights()
    net.x.upload(np.random.default_rng(0).random(net.shapes["x"], dtype=np.float32))
    net.forward()  # fill every activation buffer with realistic data
    ctx.sync()
    by_name = {l["name"]: l for l in net.specs}
    for name in args.layers.split(","):
        l = by_name[name]
        d = net.descs[name]
        fl = 2.0 * d.o * d.c * d.kh * d.kw * d.out_h * d.out_w * d.n
        plans = args.variants.split(",")
        if args.plan:
            import json
            plans = [":".join(str(x) for x in json.load(open(args.plan))[name])]
        for vs in plans:
            f = [int(x) for x in vs.split(":")]
            v = tuple(f) if len(f) >= 3 else (f[0], 0, 1)
            net.variants[name] = v
            net._conv(l)
            ctx.timer_start(2)
            for _ in range(args.reps):
                net._conv(l)
            ctx.timer_stop(2)
            ms = ctx.timer_ms(2) / args.reps
            print(f"{name} plan{v}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s  (M={d.o} K={d.c*d.kh*d.kw} N={d.n*d.out_h*d.out_w})", flush=True)


if __name__ == "__main__":
    main()
