"""Which kernels does variant 30 (image patches) launch on the ResNet-50 3x3 geometries, are the bits the oracle's, and what does a layer cost
against variants 3 / 27 at batch 32?   (GPU box)   python tools/debug/patch_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
from oracle import ref
import ctypes as C

ctx = L.Context(0)
rng = np.random.default_rng(5)


def conv(x, w, b, variant, split=(3, 1), reps=0, relu=True):
    N, Cc, H, W = x.shape
    O = w.shape[0]
    d = L.Conv2dDesc(N, Cc, H, W, O, 3, 3, (C.c_int32 * 4)(1, 1, 1, 1), 1, 1, 1, 1, 1, H, W)
    xd, wd, bd = DeviceTensor.from_numpy(ctx, x), DeviceTensor.from_numpy(ctx, w), DeviceTensor.from_numpy(ctx, b)
    nb = ctx.lib.rten_hip_conv2d_f32_packed_bytes(C.byref(d))
    pk = DeviceTensor(ctx, (nb // 4,), np.float32)
    ctx.call("rten_hip_conv2d_f32_prepack", C.byref(d), wd.vp, pk.vp)
    y = DeviceTensor(ctx, (N, O, H, W), np.float32)
    ctx.call("rten_hip_set_gemm_variant_override", variant)
    ctx.call("rten_hip_set_gemm_split", *split)
    flags = L.CONV_RELU if relu else 0
    ctx.call("rten_hip_conv2d_f32", C.byref(d), xd.vp, pk.vp, 1, bd.vp, None, flags, y.vp)
    ctx.sync()
    us = None
    if reps:
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.call("rten_hip_conv2d_f32", C.byref(d), xd.vp, pk.vp, 1, bd.vp, None, flags, y.vp)
        ctx.sync()
        us = (time.perf_counter() - t0) / reps * 1e6
    ctx.call("rten_hip_set_gemm_variant_override", -1)
    ctx.call("rten_hip_set_gemm_split", 3, 1)
    return y.numpy(), us


ctx.profile(True)
for (O, Cc, H) in ((64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7), (24, 16, 9)):
    x = rng.standard_normal((2, Cc, H, H)).astype(np.float32)
    w = (rng.standard_normal((O, Cc, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    want = ref.conv2d_f32(x, w, b, pads=(1, 1, 1, 1), relu=True)
    for split in ((0, 1), (2, 2), (1, 3)):
        ctx.profile_reset()
        got, _ = conv(x, w, b, 30, split)
        names = [str(r.get("kernel") or r.get("name") or r) for r in ctx.profile_report() if "igemm" in str(r)]
        print(f"O={O} C={Cc} H={H} split={split}: bit-exact={np.array_equal(got.view(np.int32), want.view(np.int32))}  kernels={names}", flush=True)
ctx.profile(False)
for (O, Cc, H) in ((64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)):
    x = rng.standard_normal((32, Cc, H, H)).astype(np.float32)
    w = (rng.standard_normal((O, Cc, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    nblk = (Cc * 9 + 255) // 256
    row = []
    for split in ((0, 1), (1, min(5, nblk)), (2, min(9, nblk))):
        for v in (3, 27, 30):
            _, us = conv(x, w, b, v, split, reps=30)
            row.append(f"v{v}m{split[0]}g{split[1]}={us:6.1f}")
    print(f"O={O} C={Cc} H={H} batch 32 us (incl. launch): " + "  ".join(row), flush=True)
