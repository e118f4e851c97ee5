import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
ctx = L.Context(0)
def run(hd, S, T):
    rng = ref.XorShiftRng(5)
    scale = np.float32(1.0 / np.sqrt(hd))
    q = rng.f32(S * hd).reshape(1, 1, S, hd) - 0.5
    k = rng.f32(T * hd).reshape(1, 1, T, hd) - 0.5
    v = rng.f32(T * hd).reshape(1, 1, T, hd) - 0.5
    qd, kd, vd = (DeviceTensor.from_numpy(ctx, a) for a in (q, k, v))
    d = L.SdpaDesc(1, 1, S, T, hd, hd, S * hd, S * hd, hd, T * hd, T * hd, hd, T * hd, T * hd, hd, S * hd, S * hd, hd, 0, 0, float(scale), 1)
    want = ref.sdpa(q, k, v, mask=None, scale=scale, lanes=16, flush_nan=True)
    out = DeviceTensor(ctx, (1, 1, S, hd), np.float32)
    ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, None, out.vp)
    ctx.sync()
    got = out.numpy()
    bad = got != want
    print(f"hd={hd} S{S} T{T}: {bad.sum()} of {bad.size} differ maxabs {np.abs(got-want).max():.3g}")
for hd in (128, 64):
    for T in (129, 130, 255, 256, 257, 383, 384, 385, 400, 448, 480, 511, 512):
        run(hd, 128, T)
