#!/usr/bin/env python3
"""Per-launch times of the int8 ResNet-50 pipeline (eager, HIP-event profile of the backend): one line per kernel class and
per conv layer.  RTEN_HIP_DEBUG tuning bits apply (e.g. 512 = six LDS stages for the 64x64 int8 tile)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L
from rten_amd.workloads import resnet50_int8

ctx = L.Context(0)
net = resnet50_int8.ResNet50Int8(ctx, 32)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((32, 3, 224, 224), dtype=np.float32))
for _ in range(3):
    net.forward()
ctx.sync()
ctx.profile_reset(); ctx.profile(True)
for _ in range(10):
    net.forward()
ctx.sync(); ctx.profile(False)
rep = sorted(ctx.profile_report(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rep)
print(f"eager profile: {tot / 10:.3f} ms of kernel time per step")
for r in rep:
    print(f"  {r['kernel']:44s} {r['launches'] // 10:3d} launches/step  {r['ms'] / r['launches'] * 1e3:7.1f} us avg  {r['ms'] / 10:7.3f} ms/step  {r['bytes'] / max(r['ms'], 1e-9) / 1e6:7.1f} GB/s")
net.capture()
for _ in range(10):
    net.run()
ctx.sync()
ctx.timer_start(1)
for _ in range(50):
    net.run()
ctx.timer_stop(1)
print(f"hipGraph replay: {ctx.timer_ms(1) / 50:.3f} ms per step")
