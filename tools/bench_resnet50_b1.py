#!/usr/bin/env python3
"""Secondary benchmark: BASELINE configs[0] (ResNet-50 batch 1 latency; f32, or the dynamically quantized int8 graph with --config int8) on one MI355X.
One JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L  # noqa: E402
from rten_amd.workloads import resnet50, resnet50_int8  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--config", choices=("f32", "int8"), default="f32")
args = ap.parse_args()
ctx = L.Context(0)
net = resnet50_int8.ResNet50Int8(ctx, args.batch) if args.config == "int8" else resnet50.ResNet50(ctx, args.batch)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((args.batch, 3, 224, 224), dtype=np.float32))
net.autotune(reps=5)
net.capture()
for _ in range(args.warmup):
    net.run()
ctx.sync()
lat = []
for _ in range(args.steps):
    t0 = time.perf_counter()
    net.run()
    ctx.sync()
    lat.append((time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter()
for _ in range(args.steps):
    net.run()
ctx.sync()
el = time.perf_counter() - t0
dt = "u8 x i8 -> i32 (f32 between layers)" if args.config == "int8" else "f32"
print(json.dumps({"metric": f"ResNet-50 {args.config} batch {args.batch}: p50 latency and back-to-back throughput", "p50_latency_ms": round(float(np.median(lat)), 4),
                  "value": round(args.batch * args.steps / el, 1), "unit": "inferences/s", "ms_per_step_back_to_back": round(el / args.steps * 1e3, 4),
                  "n_gpus": 1, "dtype": dt, "data": "synthetic", "config": {"workload": "ResNet-50 v1.5 %s, 224x224, batch %d (BASELINE configs[0]), hipGraph replay" % (args.config, args.batch)}}))
