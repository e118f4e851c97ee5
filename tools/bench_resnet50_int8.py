#!/usr/bin/env python3
"""Secondary benchmark: BASELINE configs[2] (dynamically quantized int8 ResNet-50, batch 32) on one MI355X.
Prints one JSON line in bench.py's format (not the driver's headline; bench.py stays on configs[1])."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L  # noqa: E402
from rten_amd.workloads import resnet50, resnet50_int8  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
ctx = L.Context(0)
net = resnet50_int8.ResNet50Int8(ctx, args.batch)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((args.batch, 3, 224, 224), dtype=np.float32))
net.capture()
for _ in range(args.warmup):
    net.run()
ctx.sync()
t0 = time.perf_counter()
for _ in range(args.steps):
    net.run()
ctx.sync()
el = time.perf_counter() - t0
ctx.profile_reset(); ctx.profile(True)
g, net.graph = net.graph, None
for _ in range(args.steps):
    net.forward()
ctx.sync(); ctx.profile(False); net.graph = g
rep = [r for r in ctx.profile_report() if "i8" in r["kernel"]]
ms = sum(r["ms"] for r in rep); ops = sum(r["flops"] for r in rep)
print(json.dumps({"metric": "inferences/sec, ResNet-50 int8 (dynamic quantization) batch 32", "value": round(args.batch * args.steps / el, 2), "unit": "inferences/s",
                  "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
                  "dtype": "u8 x i8 -> i32", "data": "synthetic", "config": {"workload": "ResNet-50 int8, DynamicQuantizeLinear -> ConvIntegerToFloat per conv (BASELINE configs[2])"},
                  "roofline": {"bound": "mfma", "kernel": "igemm_i8_fast_kernel (all tiles)", "achieved": round(ops / (ms * 1e-3) / 1e12, 2) if ms else None, "peak": 5033.0, "unit": "TOP/s",
                               "frac": round(ops / (ms * 1e-3) / 1e12 / 5033.0, 4) if ms else None, "kernel_ms_per_step": round(ms / args.steps, 4)}}))
