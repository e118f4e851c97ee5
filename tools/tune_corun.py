#!/usr/bin/env python3
"""Per-layer launch plans chosen UNDER SELF-CO-RUN (round 6): the default schedule of bench.py runs three replicas ("lanes") side by side, so a layer's launch never
has the chip to itself -- whatever its last partial round of tiles leaves idle, another replica's launch fills.  tools/tune_lanes.py measures a candidate by the
whole-model throughput, where one layer family (<= 14 % of the FLOPs) changes the step by less than the run-to-run noise.  Here the SAME layer runs on three streams at
once (three runner networks sharing one weight arena, each on its REAL input activations of that layer -- the operand data matters: the matrix pipe and the data paths
draw less power on zeros, tools/probes/kloop2.hip), each stream a captured hipGraph of REPS launches, and the figure is launches per second over all three streams.

    python tools/tune_corun.py [--lanes 3] [--out profiles/plans/experiments/f32_corun3.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # every stream a hardware queue of its own, as in bench.py (must be set before the HIP runtime starts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--plan", default=os.path.join(ROOT, "profiles", "plans", "f32_lanes.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "plans", "experiments", "f32_corun3.json"))
    ap.add_argument("--only", default=None, help="comma-separated layer names (representatives) to restrict the sweep to")
    ap.add_argument("--margin", type=float, default=0.01)
    ap.add_argument("--exclude", default="", help="comma-separated kernel variants to leave out of the candidates (e.g. 28,29,31: the barrier-free 32x32-wave forms and the small-M alias, which measure like variant 3 on every dense layer)")
    ap.add_argument("--full", action="store_true", help="every plan of the runner's candidate list (all tile variants, thin-tail, persistent, lean, every split form) instead of the short list")
    args = ap.parse_args()
    from rten_amd import lib as L
    from rten_amd.workloads.corun import CoRun
    incumbent = json.load(open(args.plan))
    cr = CoRun(args.lanes, args.batch, incumbent)
    nets, specs, descs, fams = cr.nets, cr.specs, cr.descs, cr.families()

    def candidates(key):
        o, c, k, s, h, res = key
        nblk = (c * k * k + 255) // 256
        tiles = [27, 3, 1, 2, 0, 13, 14, 12]  # 64x64 (2 / 3 stages), 128x64, 64x128, 128x128, and the four-stage forms of the larger ones
        if o < 128:
            tiles = [27, 3, 2, 14]
        cands = [[v, 0, 1, od] for v in tiles for od in (0, 1)]
        if nblk > 1:
            for g in sorted({2, 3, 4, 6, nblk, max(2, nblk // 2), max(2, nblk // 3)} & set(range(2, nblk + 1))):
                for v in tiles:
                    cands.append([v, 1, g, 0])
                    cands += [[v, 2, g, 0], [v, 2, g, 3]]
        return cands

    def measure(idx, l, plan):
        return cr.measure(idx, plan, reps=args.reps)

    out_plan = dict(incumbent)
    only = set(args.only.split(",")) if args.only else None
    total_inc = total_new = 0.0
    for key, members in sorted(fams.items(), key=lambda kv: -len(kv[1])):
        idx, l = members[1] if len(members) > 1 else members[0]  # (a stage's first block reads another producer: take a later member when there is one)
        if only and l["name"] not in only:
            continue
        cr.position(idx)  # this layer's real input (the runner reuses activation buffers: the pass stops right before the layer)
        d = descs[l["name"]]
        flops = cr.flops(l["name"])
        inc = list(incumbent[l["name"]])
        try:
            t_inc = min(measure(idx, l, inc), measure(idx, l, inc))
        except L.HipError as e:
            print(f"{l['name']}: incumbent failed: {e}", flush=True)
            continue
        rows = []
        skip = {int(v) for v in args.exclude.split(",") if v}
        for cand in ([list(p) for p in nets[0].candidate_plans(l)] if args.full else candidates(key)):
            if cand == inc or cand[0] in skip:
                continue
            try:
                rows.append((measure(idx, l, cand), cand))
            except L.HipError:
                continue
        rows.sort()
        best_t, best_c = t_inc, inc
        for t, cand in rows[:3]:  # confirm the leaders
            t2 = min(t, measure(idx, l, cand))
            if t2 < best_t * (1 - args.margin):
                best_t, best_c = t2, cand
        tag = f"O{key[0]} C{key[1]} k{key[2]} s{key[3]} {key[4]}x{key[4]}{' +res' if key[5] else ''} x{len(members)}"
        top = " ".join(f"{c}={t:.1f}" for t, c in rows[:6])
        print(f"{tag:38s} {l['name']:7s} incumbent {inc} {t_inc:6.1f} us ({flops / t_inc / 1e6:5.1f} TF/s) -> {best_c} {best_t:6.1f} us ({flops / best_t / 1e6:5.1f}) | {top}", flush=True)
        for _, m in members:
            out_plan[m["name"]] = best_c
            net_l = m
        total_inc += t_inc * len(members)
        total_new += best_t * len(members)
        for net in nets:
            net.variants[l["name"]] = tuple(inc)
    print(f"# sum over layers (co-run us per launch x layers): incumbent {total_inc:.0f} us, chosen {total_new:.0f} us ({100 * (total_inc - total_new) / max(total_inc, 1e-9):.1f} % less)", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out_plan, open(args.out, "w"))


if __name__ == "__main__":
    main()
