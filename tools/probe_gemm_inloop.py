"""Round 6: where does the production 64x64 f32 GEMM kernel lose against its own k-loop probe (tools/probes/kloop2.hip: 140-143 TF/s at 4-6 workgroups per CU)?
One kernel (variant 3 / 27 / 1 / 0), dense operands, three questions: (a) operand DATA (zeros / constants / random) -> power / clock; (b) K long enough that the
per-tile prologue / epilogue vanish -> the in-loop rate on real addresses; (c) A layout (row-major [M][K] = A_K4, k-major [K][M] = A_M4)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
ctx = L.Context(0)
rng = np.random.default_rng(0)


def time_gemm(m, k, n, a_np, w_np, kmajor, variants=(3, 27, 1, 0), reps=8):
    a = DeviceTensor.from_numpy(ctx, a_np); w = DeviceTensor.from_numpy(ctx, w_np)
    out = DeviceTensor(ctx, (m, n), np.float32)
    d = L.gemm_desc(m, n, k, 1, m, n, 1, n) if kmajor else L.gemm_desc(m, n, k, k, 1, n, 1, n)
    res = {}
    for rnd in range(2):
        for v in variants:
            ctx.set_gemm_variant(v)
            f = lambda: ctx.call("rten_hip_gemm_f32", C.byref(d), a.vp, w.vp, None, out.vp)
            f(); ctx.sync()
            ctx.timer_start(3)
            for _ in range(reps): f()
            ctx.timer_stop(3)
            res[v] = min(res.get(v, 1e9), ctx.timer_ms(3) / reps * 1e3)
    ctx.set_gemm_variant(-1)
    fl = 2.0 * m * k * n
    return " ".join(f"v{v}: {t:8.1f} us {fl / t / 1e6:6.1f} TF/s |" for v, t in res.items())


for (m, k, n) in ((4096, 768, 3072), (4096, 3072, 768), (4096, 12288, 3072), (8192, 4096, 8192)):
    for kmajor in (False, True):
        for data in ("random", "zeros", "const"):
            shape_a = (k, m) if kmajor else (m, k)
            if data == "random":
                a_np, w_np = rng.standard_normal(shape_a, dtype=np.float32), rng.standard_normal((k, n), dtype=np.float32)
            elif data == "zeros":
                a_np, w_np = np.zeros(shape_a, np.float32), np.zeros((k, n), np.float32)
            else:
                a_np, w_np = np.full(shape_a, 0.0115, np.float32), np.full((k, n), 0.0115, np.float32)
            print(f"{m}x{k}x{n} A {'k-major' if kmajor else 'row-major'} {data:6s}: " + time_gemm(m, k, n, a_np, w_np, kmajor), flush=True)
