import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
ctx = L.Context(0)
def run(hd, B, H, S, T, path):
    rng = ref.XorShiftRng(97 + hd)
    scale = np.float32(1.0 / np.sqrt(hd))
    q = rng.f32(B * H * S * hd).reshape(B, H, S, hd) - 0.5
    k = rng.f32(B * H * T * hd).reshape(B, H, T, hd) - 0.5
    v = rng.f32(B * H * T * hd).reshape(B, H, T, hd) - 0.5
    qd, kd, vd = (DeviceTensor.from_numpy(ctx, a) for a in (q, k, v))
    d = L.SdpaDesc(B, H, S, T, hd, hd, H * S * hd, S * hd, hd, H * T * hd, T * hd, hd, H * T * hd, T * hd, hd, H * S * hd, S * hd, hd, 0, 0, float(scale), 1)
    want = ref.sdpa(q, k, v, mask=None, scale=scale, lanes=16, flush_nan=True)
    ctx.call("rten_hip_set_sdpa_path", path)
    out = DeviceTensor(ctx, (B, H, S, hd), np.float32)
    ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, None, out.vp)
    ctx.sync()
    ctx.call("rten_hip_set_sdpa_path", 0)
    got = out.numpy()
    bad = got != want
    rows = sorted(set(np.argwhere(bad)[:, 2].tolist()))
    cols = sorted(set(np.argwhere(bad)[:, 3].tolist()))
    print(f"hd={hd} B{B} H{H} S{S} T{T} path={path}: {bad.sum()} of {bad.size} differ; rows {rows[:8]}..{rows[-3:] if rows else ''} ({len(rows)}) cols {cols[:6]}.. ({len(cols)}) maxrel {np.abs((got-want)/np.maximum(np.abs(want),1e-9)).max() if bad.any() else 0:.3g}")
for hd in (32, 128):
    for T in (200, 256, 257, 300, 384, 512):
        for path in (0, 1):
            run(hd, 1, 1, 128, T, path)
run(64, 1, 1, 128, 257, 1)
run(64, 1, 1, 40, 257, 1)
