import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
ctx = L.Context(0)
def run(hd, S, T, mode):
    rng = ref.XorShiftRng(5)
    B = H = 1
    scale = np.float32(1.0 / np.sqrt(hd))
    q = rng.f32(S * hd).reshape(1, 1, S, hd) - 0.5
    k = rng.f32(T * hd).reshape(1, 1, T, hd) - 0.5
    v = rng.f32(T * hd).reshape(1, 1, T, hd) - 0.5
    if mode == "v1": v[:] = 1
    if mode == "q0": q[:] = 0
    if mode == "vidx": v[:] = np.arange(T, dtype=np.float32).reshape(1, 1, T, 1)
    qd, kd, vd = (DeviceTensor.from_numpy(ctx, a) for a in (q, k, v))
    d = L.SdpaDesc(B, H, S, T, hd, hd, H * S * hd, S * hd, hd, H * T * hd, T * hd, hd, H * T * hd, T * hd, hd, H * S * hd, S * hd, hd, 0, 0, float(scale), 1)
    want = ref.sdpa(q, k, v, mask=None, scale=scale, lanes=16, flush_nan=True)
    out = DeviceTensor(ctx, (B, H, S, hd), np.float32)
    ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, None, out.vp)
    ctx.sync()
    got = out.numpy()
    bad = got != want
    print(f"hd={hd} S{S} T{T} {mode}: {bad.sum()} of {bad.size} differ; got[0,0,0,:4]={got[0,0,0,:4]} want={want[0,0,0,:4]}; got[0,0,5,64:68]={got[0,0,min(5,S-1),hd//2:hd//2+4]} want={want[0,0,min(5,S-1),hd//2:hd//2+4]}")
for mode in ("rand", "v1", "q0", "vidx"):
    run(128, 128, 257, mode)
run(128, 128, 384, "q0")
run(128, 32, 300, "rand")
