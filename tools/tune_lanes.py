#!/usr/bin/env python3
"""Launch-plan tuning UNDER THE SCHEDULE THAT IS TIMED: ResNet-50 f32, batch 32, one chain per replica, `--lanes` replicas side by side (bench.py's default).

The committed one-chain plan (profiles/plans/f32_1chain.json) was tuned layer by layer, stand-alone.  With a second replica's launches filling the idle
compute units, the best plan of a layer can differ (split-K plans exist to fill a chip that a lone launch leaves idle).  This tool does coordinate descent
over the distinct convolution SHAPES of the network (layers of one shape share a plan): for each shape it tries a fixed candidate list, measuring whole-model
throughput with every replica running (rten_hip_model_set_plan + re-prepare between measurements), and keeps a candidate only if it beats the incumbent by
more than the noise margin, twice.  Writes the plan as a plan file.

    python tools/tune_lanes.py --lanes 2 --out gpurun_out/f32_1chain_lanes2.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rten_amd import lib as L, onnx_writer  # noqa: E402
from rten_amd.tensor import DeviceTensor  # noqa: E402
from rten_amd.workloads import resnet50  # noqa: E402

BATCH = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--margin", type=float, default=0.003, help="a candidate must beat the incumbent by this fraction, in two measurements")
    ap.add_argument("--big-tiles", action="store_true", help="add 128x128 / 128x64 / 64x128 tiles (three and four LDS stages, fragments-first) to every shape's candidates")
    ap.add_argument("--plan", default=os.path.join(ROOT, "profiles", "plans", "f32_1chain.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "f32_1chain_lanes.json"))
    args = ap.parse_args()
    weights = resnet50.make_weights()
    onnx_bytes = onnx_writer.resnet50_f32(weights)
    plan = json.load(open(args.plan))
    ctxs = [L.Context(0) for _ in range(args.lanes)]
    models = [L.Model(ctxs[0], onnx_bytes, json.dumps(plan), 1)]
    models += [models[0].clone(c) for c in ctxs[1:]]
    x = np.random.default_rng(1234).random((BATCH, 3, 224, 224), dtype=np.float32)
    for m, c in zip(models, ctxs):
        p = m.bind_input("x", x.shape)
        m.prepare()
        DeviceTensor(c, x.shape, np.float32, ptr=p, keepalive=m).upload(x)
        c.sync()

    def measure(pl):
        text = json.dumps(pl)
        for m in models:
            m.set_plan(text)
            m.prepare()
        for i in range(2 * args.lanes):
            models[i % args.lanes].run(join=False)
        for m in models:
            m.sync()
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            for i in range(args.steps):
                models[i % args.lanes].run(join=False)
            for m in models:
                m.sync()
            best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
        return best

    _, descs = resnet50.layer_geometry(BATCH)
    fams = {}
    for l in resnet50.conv_specs():
        d = descs[l["name"]]
        fams.setdefault((d.o, d.c, d.kh, d.stride_h, d.h, bool(l["res"])), []).append(l["name"])

    def candidates(key):
        o, c, k, s, h, res = key
        nblk = (c * k * k + 255) // 256
        cands = [[27, 0, 1, 0], [27, 0, 1, 1], [3, 0, 1, 0], [3, 0, 1, 1], [19, 0, 1, 0], [2, 0, 1, 0], [1, 0, 1, 0]]
        if args.big_tiles:  # larger tiles run closer to the matrix pipe's rate in their k-loop and lose it to tile quantisation when a launch is alone:
            cands += [[0, 0, 1, 0], [0, 0, 1, 1], [1, 0, 1, 1], [2, 0, 1, 1], [12, 0, 1, 0], [13, 0, 1, 0], [14, 0, 1, 0], [16, 0, 1, 0], [17, 0, 1, 0], [18, 0, 1, 0]]  # with a second replica filling the idle compute units that may flip
        if nblk > 1:
            for g in sorted({2, 3, 4, 5, 6, nblk} & set(range(2, nblk + 1))):
                cands += [[3, 1, g, 0], [27, 1, g, 0]]
            for g in sorted({nblk, max(2, nblk // 2), max(2, nblk // 3)}):
                cands += [[27, 2, g, 0], [27, 2, g, 3], [3, 2, g, 2]]
        return cands

    base = measure(plan)
    print(f"# {args.lanes} lanes, {args.steps} steps per measurement; incumbent plan {os.path.relpath(args.plan, ROOT)}: {base:.4f} ms per batch", flush=True)
    cur = base
    for ps in range(args.passes):
        for key, names in sorted(fams.items(), key=lambda kv: -len(kv[1])):
            inc = plan[names[0]]
            row = []
            best_c, best_ms = None, cur
            for cand in candidates(key):
                if cand == inc:
                    continue
                trial = dict(plan)
                for n in names:
                    trial[n] = cand
                try:
                    ms = measure(trial)
                except L.HipError:
                    continue  # a plan the kernel family does not offer for this shape
                row.append((ms, cand))
                if ms < best_ms * (1 - args.margin):
                    ms2 = measure(trial)  # confirm
                    if ms2 < cur * (1 - args.margin):
                        best_c, best_ms = cand, min(ms, ms2)
            tag = f"O{key[0]} C{key[1]} k{key[2]} s{key[3]} {key[4]}x{key[4]}{' +res' if key[5] else ''} x{len(names)}"
            top = " ".join(f"{c}={ms:.4f}" for ms, c in sorted(row)[:4])
            if best_c is not None:
                for n in names:
                    plan[n] = best_c
                cur = best_ms
                print(f"{tag:40s} {inc} -> {best_c}: {cur:.4f} ms | {top}", flush=True)
            else:
                print(f"{tag:40s} keeps {inc} | {top}", flush=True)
    final = measure(plan)
    print(f"# final {final:.4f} ms per batch (incumbent {base:.4f}, {100 * (base - final) / base:.2f} % faster)", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(plan, open(args.out, "w"))
    for m in reversed(models):
        m.close()


if __name__ == "__main__":
    main()
