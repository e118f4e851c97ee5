import sys, numpy as np
sys.path.insert(0, ".")
from rten_amd import lib as L
from rten_amd.workloads import resnet50
ctx = L.Context(0)
net = resnet50.ResNet50(ctx, 1)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((1, 3, 224, 224), dtype=np.float32))
net.autotune(reps=5)
for _ in range(5): net.forward()
ctx.sync()
ctx.profile_reset(); ctx.profile(True)
for _ in range(20): net.forward()
ctx.sync(); ctx.profile(False)
rep = ctx.profile_report()
tot = sum(r["ms"] for r in rep)
print("sum of kernel time per forward: %.1f us over %d launches" % (tot / 20 * 1e3, sum(r["launches"] for r in rep) / 20))
for r in sorted(rep, key=lambda r: -r["ms"])[:14]:
    print("%-46s x%3d  %7.1f us  avg %5.1f us" % (r["kernel"][:46], r["launches"] / 20, r["ms"] / 20 * 1e3, r["ms"] / r["launches"] * 1e3))
