#!/usr/bin/env python3
"""Probe: does running the batch as several independent chains (sub-batches on their own streams) hide the tile-quantisation
tails of the conv kernels?  A kernel whose tile count is not a multiple of the CU count leaves most CUs idle during its
last partial round (tools/probe_quantization.py: 6.125 rounds cost 7); a second chain's kernel can use those CUs because nothing
orders the two chains.  Weights are shared (one arena), every chain has its own activations, plan and hipGraph.

    python tools/probe_two_chains.py [--steps 40]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--chains", default="1,2,4")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--config", default="f32")
    ap.add_argument("--one-graph", action="store_true", help="capture all chains into ONE hipGraph (fork / join on the first chain's stream)")
    ap.add_argument("--sizes", default=None, help="explicit sub-batch sizes, e.g. 10,8,8,6 (overrides --chains)")
    ap.add_argument("--stagger", action="store_true", help="put the chains out of phase: chain i first runs the first i/chains of the layers once")
    ap.add_argument("--no-split", action="store_true", help="tune without split-K plans (less total work when chains overlap)")
    args = ap.parse_args()
    from rten_amd import lib
    from rten_amd.workloads import resnet50, resnet50_int8
    weights = resnet50.make_weights()
    x = np.random.default_rng(1234).random((args.batch, 3, 224, 224), dtype=np.float32)
    ref_logits = None
    for chains in ([len(args.sizes.split(","))] if args.sizes else [int(c) for c in args.chains.split(",")]):
        sub = args.batch // chains
        sizes = [sub + (1 if i < args.batch - sub * chains else 0) for i in range(chains)]  # uneven split when chains does not divide the batch
        if args.sizes:
            sizes = [int(x) for x in args.sizes.split(",")]
            assert sum(sizes) == args.batch
        starts = [sum(sizes[:i]) for i in range(chains)]
        ctxs = [lib.Context(0) for _ in range(chains)]
        nets = []
        for i, ctx in enumerate(ctxs):
            sub = sizes[i]
            if args.config == "int8":
                kw = {} if i == 0 else dict(i8_arena_ptr=nets[0].i8_arena.ptr, i8_arena_keepalive=nets[0].i8_arena)
                net = resnet50_int8.ResNet50Int8(ctx, sub, weights, **kw)
            else:
                kw = {} if i == 0 else dict(arena_ptr=nets[0].arena.ptr, arena_keepalive=nets[0].arena)
                net = resnet50.ResNet50(ctx, sub, weights, **kw)
            if i == 0:
                net.upload_weights()
                ctx.sync()
            net.x.upload(x[starts[i]:starts[i] + sub])
            nets.append(net)
        t0 = time.perf_counter()
        tuned = {}
        for net in nets:
            if net.batch not in tuned:
                if args.no_split:
                    orig = net.candidate_plans
                    net.candidate_plans = lambda l, orig=orig: [p for p in orig(l) if p[1] not in (1, 2, 3)]
                net.autotune(reps=3)
                tuned[net.batch] = dict(net.variants)
            net.variants = dict(tuned[net.batch])
        tune_s = time.perf_counter() - t0
        graph = None
        if args.one_graph and chains > 1:
            for net in nets:
                net.forward()  # warm-up: scratch allocations
            for ctx in ctxs:
                ctx.sync()
            ctxs[0].graph_begin()
            for net in nets[1:]:
                net.ctx.wait(ctxs[0])  # fork
                net.forward()
            nets[0].forward()
            for net in nets[1:]:
                ctxs[0].wait(net.ctx)  # join
            graph = ctxs[0].graph_end()
        else:
            for net in nets:
                net.capture()
        for ctx in ctxs:
            ctx.sync()

        stagger_graphs = []
        if args.stagger and chains > 1 and not graph:
            nl = len(nets[0].specs)
            for i, net in enumerate(nets):
                if i == 0:
                    stagger_graphs.append(None)
                    continue
                net.ctx.sync()
                net.ctx.graph_begin()
                net.forward(upto=i * nl // chains)
                stagger_graphs.append(net.ctx.graph_end())

        def run(n):
            for net, g in zip(nets, stagger_graphs):
                if g:
                    net.ctx.graph_launch(g)  # inside the timed region: the offset is paid for
            for _ in range(n):
                if graph:
                    ctxs[0].graph_launch(graph)
                else:
                    for net in nets:
                        net.run()
            for ctx in ctxs:
                ctx.sync()
        run(10)
        best, trials = 1e30, []
        for _ in range(4):
            t0 = time.perf_counter()
            run(args.steps)
            trials.append(round((time.perf_counter() - t0) / args.steps * 1e3, 3))
            best = min(best, trials[-1] * 1e-3)
        logits = np.concatenate([net.logits.numpy() for net in nets])
        if ref_logits is None:
            ref_logits = logits
        same = bool(np.array_equal(logits.view(np.int32), ref_logits.view(np.int32)))
        print(json.dumps({"config": args.config, "chains": chains, "sub_batches": sizes, "ms_per_step": round(best * 1e3, 4), "images_per_s": round(args.batch / best, 1),
                          "trials_ms": trials, "tune_s": round(tune_s, 1), "one_graph": bool(graph), "stagger": bool(stagger_graphs), "no_split": args.no_split, "logits_bit_identical_to_first": same}), flush=True)
        for net in nets:
            if net.graph:
                net.ctx.graph_destroy(net.graph)
        del nets, ctxs


if __name__ == "__main__":
    main()
