"""Do the four sub-batch chains of the f32 ResNet-50 step lose time by running in lockstep?   (GPU box)   python tools/debug/f32_stagger.py

The chains are symmetric and free-running, so all four execute the SAME layer at the same moment: the memory-heavy stage-0 layers of one
chain (s0 c3: 231 MB for 3.3 GFLOP at batch 32 -- 5.2 TB/s when run alone) meet the memory-heavy layers of the other three, and so do the
MFMA-heavy layers.  Here chain c's launches are enqueued `c * offset` later than chain 0's (all K launches of a chain are enqueued at once; the
chains never synchronise with each other, so the offset persists), and the steady-state time per step is taken from the difference of a 2K
and a K step run (the fill / drain of the stagger cancels)."""
import argparse, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rten_amd import lib as L
from rten_amd.workloads import resnet50

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=4)
ap.add_argument("--steps", type=int, default=40)
args = ap.parse_args()
ctx = L.Context(0)
chains = args.chains
net = resnet50.ChainedResNet50(ctx, 32, resnet50.make_weights(), chains=chains, pool=chains)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((32, 3, 224, 224), dtype=np.float32))
net.variants = json.load(open(os.path.join(ROOT, "profiles", "plans", f"f32_{chains}chains.json")))
net.capture()
used = [net.pool[p] for p in net.place]


def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass


def run(k, offset_us):
    net.sync()
    t0 = time.perf_counter()
    for i, (g, c) in enumerate(zip(net.graph, used)):
        if i and offset_us:
            spin(offset_us)
        for _ in range(k):
            c.graph_launch(g)
    net.sync()
    return (time.perf_counter() - t0) * 1e3


run(10, 0)
K = args.steps
for off in (0, 100, 200, 350, 500, 675, 900, 1350):
    rows = []
    for _ in range(3):
        a, b = run(K, off), run(2 * K, off)
        rows.append(((b - a) / K, a / K))
    best = min(rows)
    print(f"offset {off:5d} us between chains: steady {best[0]:.4f} ms/step   (K={K} run incl. fill/drain {best[1]:.4f} ms/step)", flush=True)
