import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, ".")
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
ctx = L.Context(0)
rng = np.random.default_rng(0)
for (m, k, n) in ((4096, 768, 768), (4096, 768, 2304), (4096, 768, 3072), (4096, 3072, 768), (1024, 768, 3072), (512, 1024, 1024)):
    a = DeviceTensor.from_numpy(ctx, rng.standard_normal((m, k), dtype=np.float32)); w = DeviceTensor.from_numpy(ctx, rng.standard_normal((k, n), dtype=np.float32))
    b = DeviceTensor.from_numpy(ctx, np.zeros(n, np.float32)); out = DeviceTensor(ctx, (m, n), np.float32)
    d = L.gemm_desc(m, n, k, k, 1, n, 1, n, bias_kind=L.BIAS_PER_COL)
    res = {}
    for v in list(range(16)) + [-1] + list(range(16)) + [-1]:  # two passes: the first also warms the clocks; keep the best per plan
        for o in (0, 1):
            ctx.set_gemm_variant(v); ctx.call("rten_hip_set_gemm_order", o)
            f = lambda: ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, w.vp, b.vp, out.vp)
            f(); ctx.sync()
            best = 1e9
            for _ in range(2):
                ctx.timer_start(3)
                for _ in range(10): f()
                ctx.timer_stop(3)
                best = min(best, ctx.timer_ms(3) / 10 * 1e3)
            res[(v, o)] = min(best, res.get((v, o), 1e9))
    ctx.set_gemm_variant(-1); ctx.call("rten_hip_set_gemm_order", 0)
    fl = 2.0 * m * k * n
    top = sorted(res.items(), key=lambda kv: kv[1])[:4]
    print(f"{m}x{k}x{n}: default {res[(-1,0)]:.1f} us ({fl/res[(-1,0)]/1e6:.1f} TF/s) default+order1 {res[(-1,1)]:.1f}; best " + ", ".join(f"v{v}o{o}={t:.1f}us({fl/t/1e6:.1f})" for (v, o), t in top))
