"""GEMM launch-variant sweep for the two attention products written as Einsum on [B, S, H, D] projections (BERT-base size):
prints us per hipGraph replay for the automatic plan (-1) and each variant override.  Measured (profiles/r04): 20.6-22.8 us /
18.4-30.6 us across variants, automatic plan 22.7 / 19.3 us -- the short-K (64) batched product is not variant-sensitive."""
import sys; sys.path.insert(0, ".")
import numpy as np
from rten_amd import lib as L, ops
from rten_amd.tensor import DeviceTensor
ctx = L.Context(0); ctx.enable_pool(True)
rng = np.random.default_rng(0)
B, S, H, D = 32, 128, 12, 64
q, k, v = (DeviceTensor.from_numpy(ctx, rng.standard_normal((B, S, H, D), dtype=np.float32)) for _ in range(3))
p = DeviceTensor.from_numpy(ctx, rng.standard_normal((B, H, S, S), dtype=np.float32))
def timeit(fn, reps=30):
    fn(); fn(); ctx.sync()
    ctx.graph_begin(); keep = fn(); g = ctx.graph_end()
    for _ in range(50): ctx.graph_launch(g)
    ctx.sync(); best = 1e9
    for _ in range(3):
        ctx.timer_start(3)
        for _ in range(reps): ctx.graph_launch(g)
        ctx.timer_stop(3); best = min(best, ctx.timer_ms(3) / reps * 1e3)
    return best
es, ec = ops.Einsum("bqhd,bkhd->bhqk"), ops.Einsum("bhqk,bkhd->bqhd")
for var in [-1] + list(range(16)):
    try:
        ctx.set_gemm_variant(var)
        print(var, round(timeit(lambda: es.run(ctx, [q, k])), 2), round(timeit(lambda: ec.run(ctx, [p, v])), 2), flush=True)
    except Exception as e:
        print(var, "err", str(e)[:80], flush=True)
