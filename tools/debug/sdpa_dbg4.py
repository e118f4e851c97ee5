import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
ctx = L.Context(0)
hd, S = 128, 128
for T in (300, 480, 496):
    q = np.zeros((1, 1, S, hd), np.float32)
    k = np.zeros((1, 1, T, hd), np.float32)
    v = np.zeros((1, 1, T, hd), np.float32)
    v[0, 0, :, :] = np.arange(T, dtype=np.float32).reshape(T, 1) + 1   # out = mean(t+1) if every key is used once
    # per-key probe: column d carries an indicator of key subset d (key t contributes to column t % hd)
    v2 = np.zeros((1, 1, T, hd), np.float32)
    for t in range(T):
        v2[0, 0, t, t % hd] = 1.0
    for name, vv in (("ramp", v), ("onehot", v2)):
        qd, kd, vd = (DeviceTensor.from_numpy(ctx, a) for a in (q, k, vv))
        d = L.SdpaDesc(1, 1, S, T, hd, hd, S * hd, S * hd, hd, T * hd, T * hd, hd, T * hd, T * hd, hd, S * hd, S * hd, hd, 0, 0, 1.0, 1)
        out = DeviceTensor(ctx, (1, 1, S, hd), np.float32)
        ctx.call("rten_hip_sdpa_f32", C.byref(d), qd.vp, kd.vp, vd.vp, None, out.vp)
        ctx.sync()
        got = out.numpy()[0, 0]
        if name == "ramp":
            print(f"T={T} ramp: got rows 0,1,40,127 col0 = {got[[0,1,40,127],0]} want {(T+1)/2}")
        else:
            cnt = np.rint(got * T).astype(int)   # how many keys with t % hd == d were counted, per (row, d)
            want = np.array([len(range(dd, T, hd)) for dd in range(hd)])
            bad_d = np.argwhere((cnt != want[None, :]).any(axis=0)).ravel()
            print(f"T={T} onehot: columns with wrong key counts: {bad_d[:20]} ... ({len(bad_d)}); e.g. row0 counts {cnt[0, bad_d[:10]]} want {want[bad_d[:10]]}; rows affected {np.argwhere((cnt != want[None,:]).any(axis=1)).ravel()[:10]}")
