#!/usr/bin/env python3
"""Probe: exact split-K with the ordered fold in a separate fixup launch (split mode 2) against the same fold done by the
last-arriving producer workgroup of every tile (mode 4: device-scope ticket + release / acquire fences, no extra launch).
Measured on MI355X: mode 4 is 25-70 % slower on every ResNet-50 layer at batch 32 and at batch 1 -- e.g. batch 1
s3b1c2 16.0 vs 26.2 us, batch 32 s2b1c2 84.5 vs 141.4 us -- so the fixup launch stays (profiles/r03/inkernel_fold_probe.txt)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L  # noqa: E402
from rten_amd.workloads import resnet50  # noqa: E402

for batch in (32, 1):
    ctx = L.Context(0)
    net = resnet50.ResNet50(ctx, batch)
    net.upload_weights()
    net.x.upload(np.random.default_rng(1).random((batch, 3, 224, 224), dtype=np.float32))
    net.forward()
    ctx.sync()
    for name in ("s1b1c2", "s2b1c2", "s3b1c1", "s3b1c2"):
        l = next(x for x in net.specs if x["name"] == name)
        res = {}
        for plan in net.candidate_plans(l) + [(v, 4, g, o) for (v, m, g, o) in net.candidate_plans(l) if m == 2 and (v < 4 or v >= 12) and o in (0, 2)]:
            net.variants[name] = plan
            net._conv(l)
            ms = 1e30
            for _ in range(2):
                ctx.timer_start(1)
                for _ in range(5):
                    net._conv(l)
                ctx.timer_stop(1)
                ms = min(ms, ctx.timer_ms(1) / 5)
            res[plan] = ms
        best = {m: min((ms, p) for p, ms in res.items() if p[1] == m) for m in (0, 2, 4)}
        print(f"batch {batch} {name}: unsplit {best[0][0]*1e3:.1f} us {best[0][1]}, split + fixup launch {best[2][0]*1e3:.1f} us {best[2][1]}, "
              f"split + in-kernel fold {best[4][0]*1e3:.1f} us {best[4][1]}")
