"""Wall-clock timeline of every workgroup of one ResNet-50 f32 step (batch 32), from a -DRTEN_TRACE build of gemm_f32.hip.

    cd rten_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DRTEN_TRACE -c gemm_f32.hip -o /tmp/tr/gemm_f32_trace.o
    cd .. && hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/trace.so $(ls _build/*.o | grep -v gemm_f32.o) /tmp/tr/gemm_f32_trace.o
    RTEN_HIP_LIBRARY=$PWD/rten_amd/_ab/trace.so python tools/debug/f32_trace.py [--chains 4] [--out gpurun_out/f32_trace]     (GPU box)

Every workgroup of igemm_f32_dma_kernel appends {kernel id, compute unit, s_memrealtime at entry / prologue issued / first k-tile landed /
k-loop done / epilogue issued / stores drained, M, K, N, grid, k-trips}.  The launch plan is the committed one with the non-DMA variants
mapped onto their LDS-DMA counterparts (only that kernel carries stamps).  Prints, per layer and for the whole step: time in each phase,
slot occupancy, and -- per compute unit -- the share of the step during which at least one workgroup is inside its k-loop (the only phase
that feeds the matrix pipe).  What comes out of it is in DESIGN.md (f32 section)."""
import argparse, ctypes as C, json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rten_amd import lib as L
from rten_amd.tensor import DeviceTensor
from rten_amd.workloads import resnet50

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=4)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "f32_trace"))
ap.add_argument("--plan", default=None)
ap.add_argument("--keep-variants", action="store_true", help="do not map the non-DMA kernel variants onto the DMA kernel (their workgroups are then missing from the trace)")
args = ap.parse_args()

ctx = L.Context(0)
lib = ctx.lib
lib.rten_hip_debug_trace_set.restype = C.c_int32
lib.rten_hip_debug_trace_set.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
lib.rten_hip_debug_trace_count.restype = C.c_int32
lib.rten_hip_debug_trace_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]

chains = args.chains
weights = resnet50.make_weights()
net = resnet50.ChainedResNet50(ctx, 32, weights, chains=chains, pool=chains) if chains > 1 else resnet50.ResNet50(ctx, 32, weights)
net.upload_weights()
net.x.upload(np.random.default_rng(1234).random((32, 3, 224, 224), dtype=np.float32))
ctx.sync()
plan_path = args.plan or os.path.join(ROOT, "profiles", "plans", f"f32_{chains}chain{'s' if chains > 1 else ''}.json")
plan = json.load(open(plan_path))


def to_dma(p):
    v, mode, g, o = (list(p) + [0])[:4]
    if not args.keep_variants:
        if 4 <= v < 12 or 16 <= v < 24:
            v = v & 3  # same tile shape on the 3-stage LDS-DMA kernel
        if mode in (4, 5, 6):
            mode, g = 0, 1
    return [v, mode, g, o]


if chains > 1:
    net.variants = {b: {k: to_dma(p) for k, p in t.items()} for b, t in plan.items()}
else:
    net.variants = {k: tuple(to_dma(p)) for k, p in plan.items()}
# the launches carry the record buffer in their arguments from the moment they are enqueued (a captured launch for every replay): hand it over BEFORE the capture
CAP = 1 << 21
buf = DeviceTensor(ctx, (CAP * 16,), np.uint64)
ctx.call("rten_hip_memset", buf.vp, 0, C.c_size_t(CAP * 128))
assert lib.rten_hip_debug_trace_set(ctx.h, buf.vp, CAP) == 0
net.capture()
import time
for _ in range(20):
    net.run()
(net.sync() if chains > 1 else ctx.sync())
t0 = time.perf_counter()
for _ in range(50):
    net.run()
(net.sync() if chains > 1 else ctx.sync())
step_ms = (time.perf_counter() - t0) / 50 * 1e3
print(f"[trace] chains={chains} plan={os.path.relpath(plan_path, ROOT)} (DMA-mapped) step (stamps compiled in, records written) {step_ms:.4f} ms", flush=True)

ctx.call("rten_hip_memset", buf.vp, 0, C.c_size_t(CAP * 128))  # drop the records of the warm-up replays: every non-zero record below is from the STEPS replays that follow
ctx.sync()
STEPS = 4
t0 = time.perf_counter()
for _ in range(STEPS):
    net.run()
(net.sync() if chains > 1 else ctx.sync())
traced_ms = (time.perf_counter() - t0) / STEPS * 1e3
n = C.c_uint32(0)
assert lib.rten_hip_debug_trace_count(ctx.h, C.byref(n)) == 0
lib.rten_hip_debug_trace_set(ctx.h, None, 0)
n = min(n.value, CAP)
rec = buf.numpy()[: n * 16].reshape(n, 16)
rec = rec[rec[:, 2] != 0].copy()
print(f"[trace] {n} slots handed out, {len(rec)} written by the replayed launches (last of {STEPS} steps); step with the buffer on {traced_ms:.4f} ms", flush=True)
STEPS = 1
os.makedirs(os.path.dirname(args.out), exist_ok=True)
np.savez_compressed(args.out + f"_{chains}ch.npz", rec=rec[:, :14])
if os.path.getsize(args.out + f"_{chains}ch.npz") > 24 << 20:
    os.remove(args.out + f"_{chains}ch.npz")  # gpurun merges at most 64 MiB back

# ---- the shader clock while the kernels ran: s_memtime cycles over s_memrealtime ticks (10 ns) of every workgroup's life
cyc = (rec[:, 14] & ((1 << 40) - 1)).astype(np.float64); tick = (rec[:, 14] >> 40).astype(np.float64)
okc = tick > 100  # (workgroups that lived at least a microsecond)
if okc.any():
    mhz = cyc[okc] / tick[okc] * 100.0
    print(f"[trace] shader clock over {int(okc.sum())} workgroups: median {np.median(mhz):.0f} MHz, 10th / 90th percentile {np.percentile(mhz, 10):.0f} / {np.percentile(mhz, 90):.0f} MHz "
          f"(peak figures assume 2400 MHz: at the median clock the f32 MFMA peak is {157.3 * np.median(mhz) / 2400:.1f} TFLOP/s)", flush=True)
# ---- analysis (100 MHz stamps -> us)
T = rec[:, 2:8].astype(np.float64) / 100.0
t_begin = T[:, 0].min()
T -= t_begin
span = T[:, 5].max()
kid = (rec[:, 0] & 0xffffffff).astype(np.int64)
hw = (rec[:, 1] & 0xffffffff).astype(np.int64)
xcc = (rec[:, 1] >> 32).astype(np.int64) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)  # cu_id | sh_id | se_id | xcc
M = (rec[:, 8] & 0xffffffff).astype(np.int64); K = (rec[:, 8] >> 32).astype(np.int64)
N = (rec[:, 9] & 0xffffffff).astype(np.int64); grid = (rec[:, 9] >> 32).astype(np.int64)
trips = (rec[:, 10] & 0xffffffff).astype(np.int64)
Cptr = rec[:, 12]
bm = kid & 0xff; bn = (kid >> 8) & 0xff; mode = (kid >> 16) & 0xf
print(f"[trace] {len(np.unique(cu))} distinct compute units seen; span of the {STEPS} steps {span:.1f} us ({span / STEPS:.1f} us per step)")
pro, first, loop, epi, drain = T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 4] - T[:, 3], T[:, 5] - T[:, 4]
life = T[:, 5] - T[:, 0]
flops = 2.0 * bm * bn * trips * 16
print(f"[trace] workgroup-time: total {life.sum() / STEPS:.0f} us per step = {life.sum() / span:.2f} workgroups resident on average ({life.sum() / span / len(np.unique(cu)):.2f} per CU)")
for nm, a in (("prologue (entry -> first DMA issued)", pro), ("first k-tile wait", first), ("k-loop", loop), ("fold + epilogue issue", epi), ("store drain", drain)):
    print(f"[trace]   {nm:38s} {a.sum() / life.sum() * 100:5.1f} % of workgroup-time, mean {a.mean():6.2f} us, p90 {np.percentile(a, 90):6.2f} us")

# per-CU: union of k-loop intervals, and of lifetimes
def union_len(iv):
    iv = iv[np.argsort(iv[:, 0])]
    tot, cs, ce = 0.0, iv[0, 0], iv[0, 1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs
busy_loop, busy_any, conc = [], [], []
for c in np.unique(cu):
    m = cu == c
    busy_loop.append(union_len(np.stack([T[m, 2], T[m, 3]], 1)) / span)
    busy_any.append(union_len(np.stack([T[m, 0], T[m, 5]], 1)) / span)
    conc.append(loop[m].sum() / max(union_len(np.stack([T[m, 2], T[m, 3]], 1)), 1e-9))
print(f"[trace] per CU: some workgroup resident {np.mean(busy_any) * 100:.1f} % of the time; some workgroup INSIDE its k-loop {np.mean(busy_loop) * 100:.1f} % "
      f"(min {np.min(busy_loop) * 100:.1f}, max {np.max(busy_loop) * 100:.1f}); k-loops overlapping while any runs: {np.mean(conc):.2f}")
tot_fl = flops.sum() / STEPS
print(f"[trace] tile FLOPs per step (padded tiles) {tot_fl / 1e9:.1f} G; matrix-pipe rate while a CU has a k-loop running: "
      f"{tot_fl * STEPS / (np.mean(busy_loop) * span * 1e-6) / 1e12:.1f} TF/s chip-equivalent")

# per layer (identified by output pointer + shape), first step only
key = np.stack([Cptr, M.astype(np.uint64), K.astype(np.uint64), N.astype(np.uint64)], 1)
uniq, inv = np.unique(key, axis=0, return_inverse=True)
rows = []
for u in range(len(uniq)):
    m = inv == u
    st = T[m, 0].min()
    rows.append((st, u))
rows.sort()
print("[layer] M K N tile mode grid | wgs/launch | launch span us (first..last end) | per-wg: pro first loop epi drain us | loop us per trip | TF/s over span")
seen = set()
for st, u in rows:
    m = inv == u
    Mv, Kv, Nv = int(uniq[u][1]), int(uniq[u][2]), int(uniq[u][3])
    # split the records of this (buffer, shape) into launches by time gaps: one launch per step per chain-buffer
    ts = np.sort(T[m, 0])
    nl = STEPS
    wg = m.sum() / nl
    idx = np.where(m)[0]
    order = idx[np.argsort(T[idx, 0])]
    first_launch = order[: int(round(wg))]
    sp0, sp1 = T[first_launch, 0].min(), T[first_launch, 5].max()
    fl = 2.0 * Mv * Kv * Nv
    print(f"[layer] {Mv:5d} {Kv:5d} {Nv:6d} {int(bm[first_launch[0]])}x{int(bn[first_launch[0]])} m{int(mode[first_launch[0]])} g{int(grid[first_launch[0]]):5d} | {wg:7.1f} | "
          f"{sp1 - sp0:7.1f} | {pro[first_launch].mean():5.2f} {first[first_launch].mean():5.2f} {loop[first_launch].mean():6.2f} {epi[first_launch].mean():5.2f} {drain[first_launch].mean():5.2f} | "
          f"{(loop[first_launch] / np.maximum(trips[first_launch], 1)).mean() * 1e3:6.0f} ns | {fl / ((sp1 - sp0) * 1e-6) / 1e12:6.1f}")
