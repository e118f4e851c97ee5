#!/usr/bin/env python3
"""Secondary benchmark: BASELINE configs[3] (BERT-base encoder, f32, batch 32 x 128 tokens) on one MI355X.
Prints one JSON line in bench.py's format (bench.py itself stays on configs[1])."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L  # noqa: E402
from rten_amd.workloads import bert  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seq", type=int, default=128)
args = ap.parse_args()
ctx = L.Context(0)
cfg = bert.BertConfig()
net = bert.Bert(ctx, cfg, args.batch, args.seq)
rng = np.random.default_rng(0)
net.set_inputs(rng.integers(0, cfg.vocab, (args.batch, args.seq)), np.ones((args.batch, args.seq), np.float32), np.zeros((args.batch, args.seq), np.int64))
tune = net.autotune()
net.capture()
for _ in range(args.warmup):
    net.run()
ctx.sync()
t0 = time.perf_counter()
for _ in range(args.steps):
    net.run()
ctx.sync()
el = time.perf_counter() - t0
fl = bert.flops_per_sequence(cfg, args.seq) * args.batch
ctx.profile_reset(); ctx.profile(True)
g, net.graph = net.graph, None
for _ in range(args.steps):
    net.forward()
ctx.sync(); ctx.profile(False); net.graph = g
rep = ctx.profile_report()
gem = [r for r in rep if r["kernel"].startswith("igemm_f32")]
ms = sum(r["ms"] for r in gem); gfl = sum(r["flops"] for r in gem)
print(json.dumps({"metric": "sequences/sec, BERT-base encoder f32, batch 32 x 128 tokens", "value": round(args.batch * args.steps / el, 2), "unit": "sequences/s",
                  "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "dtype": "f32",
                  "data": "synthetic", "config": {"workload": "BERT-base (12 layers, hidden 768, 12 heads) encoder forward, random-init weights (BASELINE configs[3])",
                                                  "gflop_per_step": round(fl / 1e9, 1), "whole_model_tflops": round(fl / (el / args.steps) / 1e12, 2)},
                  "roofline": {"bound": "mfma", "kernel": "igemm_f32 family (projections, FFN, attention GEMMs)", "achieved": round(gfl / (ms * 1e-3) / 1e12, 2), "peak": 157.3,
                               "unit": "TFLOP/s", "frac": round(gfl / (ms * 1e-3) / 1e12 / 157.3, 4), "kernel_ms_per_step": round(ms / args.steps, 4),
                               "all_kernels_ms_per_step": round(sum(r["ms"] for r in rep) / args.steps, 4)},
                  "autotuned_variants": {f"n={n},k={k}": net.variants[(n, k)] for (n, k) in net.variants},
                  "kernels": {r["kernel"]: round(r["ms"] / args.steps, 4) for r in sorted(rep, key=lambda r: -r["ms"])[:12]}}))
