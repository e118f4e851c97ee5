#!/usr/bin/env python3
"""ResNet-50's stem under its plan entries [27, 0, 1, 0] (the generic gather, two-stage 64x64 ring: the committed choice) and [32, 0, 1, 0] (the direct form of
gemm_f32_stem.hip), alone and under four-stream self-co-run (rten_amd/workloads/corun.py).      python tools/probe_stem.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from rten_amd.workloads.corun import CoRun  # noqa: E402

for streams in (1, 4):
    cr = CoRun(streams, 32, {})
    idx = [i for i, l in enumerate(cr.specs) if l["name"] == "stem"][0]
    fl = cr.flops("stem")
    for plan in ([27, 0, 1, 0], [3, 0, 1, 0], [32, 0, 1, 0]):
        us = min(cr.measure(idx, plan) for _ in range(3))
        print(f"streams {streams}  plan {plan}: {us:7.1f} us per launch  {fl / us / 1e6:6.1f} TF/s", flush=True)
    cr.close()
