"""One launch of rten_hip_gemm_int8 (MatMulInteger, prepacked RHS) with the kernel's cycle stamps printed -- the GEMM-form companion of i8_trace.py
(same -DRTEN_TRACE build: see its docstring).  Usage: RTEN_HIP_LIBRARY=$PWD/rten_amd/_ab/trace.so python tools/debug/i8_trace_gemm.py [m k n] ..."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor

ctx = L.Context(0)
rng = np.random.default_rng(3)
shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)] or [(4096, 768, 3072), (4096, 768, 768)]
for (m, k, n) in shapes:
    a = DeviceTensor.from_numpy(ctx, rng.integers(0, 255, (m, k)).astype(np.uint8))
    w = DeviceTensor.from_numpy(ctx, rng.integers(-127, 127, (k, n)).astype(np.int8))
    az, wz = DeviceTensor.from_numpy(ctx, np.array(128, np.uint8)), DeviceTensor.from_numpy(ctx, np.zeros(n, np.int8))
    out = DeviceTensor(ctx, (m, n), np.int32)
    packed = DeviceTensor(ctx, (ctx.lib.rten_hip_gemm_int8_packed_bytes(k, n),), np.uint8)
    ctx.call("rten_hip_gemm_int8_prepack", k, n, w.vp, n, 1, 1, packed.vp)
    dp = L.GemmInt8Desc(m, n, k, k, 1, n, 1, n, 0, 1, 1, n, 0, 1, 0, 0, 0, 1)
    for wzp in (wz.vp,):
        for _ in range(3):
            ctx.call("rten_hip_gemm_int8", C.byref(dp), a.vp, packed.vp, az.vp, wzp, None, out.vp)
        ctx.sync()
        print("----", (m, k, n), "weight zero points:", "per column (zeros)" if wzp else "none", flush=True)
        ctx.timer_start(1)
        ctx.call("rten_hip_gemm_int8", C.byref(dp), a.vp, packed.vp, az.vp, wzp, None, out.vp)
        ctx.timer_stop(1)
        print(f"whole call (A staging + GEMM): {ctx.timer_ms(1) * 1e3:.1f} us", flush=True)
