#!/usr/bin/env python3
"""Where the fused attention kernel's time goes (BERT-base: 32 x 12 heads x 128 x 128 x 64, the merged [T, 3H] QKV layout, additive [B,1,1,T] mask):
the launch timed stand-alone under the kernel's ablation bits (RTEN_HIP_DEBUG >> 24: 1 no mask, 2 no exp, 4 no PV MFMAs, 8 no QK^T MFMAs, 16 no global
loads, 32 no stores -- WRONG results, timing only) and with the mask row staged in LDS (default) or fetched per lane (bit 0x400000).

    python tools/probe_sdpa.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib as L  # noqa: E402
from rten_amd.tensor import DeviceTensor  # noqa: E402

B, S, NH, DH = 32, 128, 12, 64
H = NH * DH
FL = 4.0 * B * NH * S * S * DH


def run(debug, reps=50):
    os.environ["RTEN_HIP_DEBUG"] = str(debug)
    ctx = L.Context(0)
    rng = np.random.default_rng(0)
    qkv = DeviceTensor.from_numpy(ctx, rng.standard_normal((B * S, 3 * H), dtype=np.float32))
    out = DeviceTensor(ctx, (B * S, H), np.float32)
    am = np.ones((B, S), np.float32)
    am[::3, 100:] = 0
    mask = DeviceTensor.from_numpy(ctx, ((1.0 - am) * np.finfo(np.float32).min).astype(np.float32).reshape(B, 1, 1, S))
    sd = L.SdpaDesc(B, NH, S, S, DH, DH, S * 3 * H, DH, 3 * H, S * 3 * H, DH, 3 * H, S * 3 * H, DH, 3 * H, S * H, DH, H, S, 0, 0.125, 0)
    q, k, v = (C.c_void_p(qkv.ptr + i * H * 4) for i in range(3))
    res = {}
    for label, m in (("mask", mask.vp), ("no mask", None)):
        fn = lambda: ctx.call("rten_hip_sdpa_f32", C.byref(sd), q, k, v, m, out.vp)
        fn()
        ctx.sync()
        best = 1e30
        for _ in range(3):
            ctx.timer_start(1)
            for _ in range(reps):
                fn()
            ctx.timer_stop(1)
            best = min(best, ctx.timer_ms(1) / reps)
        res[label] = best * 1e3
    ctx.close()
    return res


def main():
    rows = [("product kernel (16-query waves, 768 workgroups: sdpa_fused16_kernel)", 0), ("32-query waves, FULL-shape instantiation, mask row in LDS", 0x200000),
            ("32-query waves, general instantiation (per-lane row / key tests)", 0x800000), ("general + mask fetched per lane (round-4 form)", 0xC00000)]
    # ablation of the 16-query kernel (its ABL instantiation; bits 24..31)
    abl = [("ABL instantiation, no bit effective (256 = unused bit)", 256), ("- mask add", 1), ("- exp", 2), ("- PV MFMAs", 4), ("- QK^T MFMAs", 8), ("- both MFMA phases", 12),
           ("- global loads", 16), ("- stores", 32), ("- loads - stores", 48), ("- softmax arithmetic (all of phase 2)", 64), ("- LDS staging writes", 128),
           ("- loads - stores - staging writes", 176), ("- everything but the MFMAs", 240), ("- MFMAs - softmax (loads, staging, LDS reads, stores left)", 76),
           ("- everything", 255)]
    if "--old-ladder" in sys.argv:  # the 32-query kernel's ladder (profiles/r08/sdpa_ablation.txt)
        rows += [(l, (b << 24) | 0x200000) for l, b in (("32q: ABL, no bit effective", 64), ("32q: - exp", 2), ("32q: - both MFMA phases", 12), ("32q: - loads - stores", 48), ("32q: - everything", 63))]
    else:
        rows += [(l, b << 24) for l, b in abl]
    print(f"# fused attention, b={B} heads={NH} s=t={S} d={DH} (the 32-query form: 384 workgroups of 256 threads on 256 compute units; ablation rows are that form); {FL/1e9:.2f} GFLOP -> {FL/157.3e12*1e6:.1f} us at the f32 MFMA peak")
    if "--quick" in sys.argv:
        rows = rows[:3]
    for label, dbg in rows:
        r = run(dbg)
        print(f"{label:62s} mask {r['mask']:6.1f} us ({FL/r['mask']/1e6/157.3:.3f})   no mask {r['no mask']:6.1f} us ({FL/r['no mask']/1e6/157.3:.3f})", flush=True)


if __name__ == "__main__":
    main()
