#!/usr/bin/env python3
"""bench.py -- ResNet-50 f32, batch 32 per GPU, on the HIP backend (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one forward pass of ResNet-50 (53 convs + maxpool + global-avg-pool + fc) over one batch of
32 synthetic 224x224 images that is already resident in HBM.  One process per GPU; batches are
independent, so the path shards with no data-path collective (weak scaling: 32 images per GPU); the only
collective is the one-time RCCL broadcast of the prepacked weight arena from rank 0 at load.

Rank 0 prints ONE JSON line: metric/value (whole-job inferences/s), roofline of the dominant kernel
(measured live with HIP events on the backend's stream in an instrumented pass over the same K steps),
and cpu_baseline (the CPU oracle -- a port of the reference algorithm -- timed on this host's cores on a
bounded sample; N=1 only).  At N=1 the line also carries `secondary`: the int8 ResNet-50 (configs[2]) and BERT-base (configs[3])
harnesses run in child processes after the headline measurement.  Defaults: K = 50, W = 20 (the chip needs about 20 ms of
load to settle its clocks; a run with the driver's own K / W is timed exactly as given).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X v_mfma_f32_32x32x2_f32 peak (MI355X_MICROARCH.md)
BATCH_PER_GPU = 32


def cpu_baseline(specs, weights, budget_s=12.0):
    """Times the CPU oracle (reference algorithm port, OpenMP) on a bounded sample of the same workload."""
    from oracle import models as omodels
    from oracle import ref
    threads = ref.num_threads()
    x = ref.XorShiftRng(7).f32(2 * 3 * 224 * 224).reshape(2, 3, 224, 224)
    t0 = time.perf_counter()
    omodels.resnet50_forward(specs, weights, x)
    t_first = time.perf_counter() - t0
    reps = int(max(1, min(16, budget_s / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        omodels.resnet50_forward(specs, weights, x)
    dt = time.perf_counter() - t0
    imgs = 2 * reps
    return {"value": round(imgs / dt, 3), "unit": "inferences/s", "cores": threads, "kind": "port",
            "sample": f"{imgs} images (batch 2 x {reps} forward passes) of the same ResNet-50 graph through oracle/rten_oracle.c "
                      f"({threads} OpenMP threads, {dt:.1f} s)"}


def cpu_op_baselines():
    """cpu_baseline leg for the per-kernel table (tools/bench_ops.py --cpu-baseline): the CPU oracle timed on a bounded
    sample of each kernel's shape, scaled linearly to the full shape.  Returns {op: {"us": ..., "sample": ...}}."""
    from oracle import ref
    rng = np.random.default_rng(0)
    out = {}

    def t(fn, reps=3):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e6

    def put(op, us, scale, sample):
        out[op] = {"us": round(us * scale, 1), "cores": ref.num_threads(), "kind": "port", "sample": f"{sample}, scaled x{scale:g}"}

    n = 1 << 22
    x, x2 = rng.standard_normal(n, dtype=np.float32), rng.standard_normal(n, dtype=np.float32)
    full = 32 * 256 * 56 * 56
    put("Relu", t(lambda: ref.relu(x)), full / n, f"n={n}")
    put("Add", t(lambda: ref.add(x, x2)), full / n, f"n={n}")
    put("Gelu", t(lambda: ref.gelu(x)), 4096 * 3072 / n, f"n={n}")
    put("Erf", t(lambda: ref.erf(x)), 4096 * 3072 / n, f"n={n}")
    put("DynamicQuantizeLinear", t(lambda: ref.dynamic_quantize_linear(x)), full / n, f"n={n}")
    xi = rng.integers(-1000, 1000, n).astype(np.int32)
    put("cast_scale", t(lambda: ref.cast_scale(xi, np.float32(0.01))), full / n, f"n={n}")
    sm = x[:4096 * 128].reshape(4096, 128)
    put("Softmax", t(lambda: ref.softmax(sm)), 12, "rows=4096 cols=128")
    ln = x[:4096 * 768].reshape(4096, 768)
    g, b = np.ones(768, np.float32), np.zeros(768, np.float32)
    put("LayerNormalization", t(lambda: ref.layer_norm(ln, g, b, eps=1e-12)), 1, "rows=4096 cols=768")
    mp = x[:2 * 64 * 112 * 112].reshape(2, 64, 112, 112)
    put("MaxPool 3x3/2", t(lambda: ref.max_pool(mp, (3, 3), (2, 2), (1, 1, 1, 1))), 16, "2x64x112x112")
    gp = x[:32 * 2048 * 49].reshape(32, 2048, 7, 7)
    put("GlobalAveragePool", t(lambda: ref.global_average_pool(gp)), 1, "32x2048x7x7")
    for (m, k, nn, name) in ((4096, 768, 768, "MatMul proj"), (4096, 768, 3072, "MatMul FFN1 + Gelu"), (4096, 3072, 768, "MatMul FFN2")):
        a, w, bias = rng.standard_normal((512, k), dtype=np.float32), rng.standard_normal((k, nn), dtype=np.float32), np.zeros(nn, np.float32)
        if "Gelu" in name:
            put(name, t(lambda: ref.gelu(ref.matmul_f32(a, w, bias=bias)), 2), 8, f"512x{k}x{nn}")
        else:
            put(name, t(lambda: ref.matmul_f32(a, w, bias=bias), 2), 8, f"512x{k}x{nn}")
    q, kk, v = (rng.standard_normal((2, 12, 128, 64), dtype=np.float32) for _ in range(3))
    put("sdpa (QK^T, softmax, PV)", t(lambda: ref.sdpa(q, kk, v, scale=0.125), 2), 16, "b=2 h=12 s=128 d=64")
    for (m, k, nn) in ((4096, 768, 768), (4096, 768, 3072)):
        a, w = rng.integers(0, 255, (512, k)).astype(np.uint8), rng.integers(-127, 127, (k, nn)).astype(np.int8)
        put(f"MatMulInteger {m}x{k}x{nn}", t(lambda: ref.gemm_int8(a, w, np.array(128, np.uint8), None), 2), 8, f"512x{k}x{nn}")
    for (o_, c_, hw, k_, p_, name) in ((64, 64, 56, 3, 1, "s0 3x3"), (256, 256, 14, 3, 1, "s2 3x3"), (256, 64, 56, 1, 0, "s0 1x1 expand")):
        xq, wq = rng.integers(0, 255, (2, c_, hw, hw)).astype(np.uint8), rng.integers(-127, 127, (o_, c_, k_, k_)).astype(np.int8)
        put(f"ConvInteger {name}", t(lambda: ref.conv2d_int8(xq, wq, x_zp=128, pads=(p_,) * 4), 2), 16, f"2x{c_}x{hw}x{hw}")
    from oracle import einsum as oe
    rs = x[:4096 * 128].reshape(4096, 128)
    put("ReduceSum last axis", t(lambda: oe.reduce_sum(rs, [1])), 12, "4096x128")
    rc = x[:512 * 3072].reshape(512, 3072)
    put("ReduceSum strided axis", t(lambda: oe.reduce_sum(rc, [0])), 8, "512x3072 -> 3072")
    qe, ke = (rng.standard_normal((2, 128, 12, 64), dtype=np.float32) for _ in range(2))
    pe = rng.standard_normal((2, 12, 128, 128), dtype=np.float32)
    put("Einsum bqhd,bkhd->bhqk", t(lambda: oe.einsum("bqhd,bkhd->bhqk", qe, ke), 2), 16, "2x128x12x64")
    put("Einsum bhqk,bkhd->bqhd", t(lambda: oe.einsum("bhqk,bkhd->bqhd", pe, ke), 2), 16, "2x12x128x128")
    return out


def secondary_configs():
    """The other single-GPU configs of BASELINE.json, measured by their own harnesses in child processes after the headline
    run (same backend, same box): reported for the record, never part of `value`.  A failure is recorded, not raised."""
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for key, script in (("resnet50_int8_b32", "bench_resnet50_int8.py"), ("bert_base_f32_b32_s128", "bench_bert.py"),
                        ("resnet50_f32_b1_latency", "bench_resnet50_b1.py")):
        try:
            p = subprocess.run([sys.executable, os.path.join(root, "tools", script)], capture_output=True, text=True, timeout=420, cwd=root)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            res[key] = {k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "p50_latency_ms", "ms_per_step_back_to_back", "dtype") if k in j}
            if key.startswith("bert") and "roofline" in j:
                res[key]["gemm_family_tflops"] = j["roofline"].get("achieved")
                res[key]["gemm_family_frac_of_f32_mfma_peak"] = j["roofline"].get("frac")
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layer-table", action="store_true", help="print the per-layer autotune table to stderr")
    ap.add_argument("--save-plan", default=None, help="write the autotuned per-layer plans (variant, split mode, groups) as JSON")
    ap.add_argument("--load-plan", default=None, help="use per-layer plans from a JSON file instead of autotuning (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of BASELINE configs[2] (int8 ResNet-50) and configs[3] (BERT-base) reported under \"secondary\"")
    ap.add_argument("--concurrent", action="store_true", help="run the projection shortcuts on a second stream (parallel graph branches)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RTEN_DIST_BACKEND=gloo lets the rank != 0 path (arena received by broadcast) be exercised with several ranks on ONE
        # GPU (RCCL refuses two ranks per device); the driver's multi-GPU runs use the default, nccl (= RCCL over xGMI).
        backend = os.environ.get("RTEN_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)

    from rten_amd import lib
    from rten_amd.workloads import resnet50
    from rten_amd.sharding import broadcast_weight_arena

    ctx = lib.Context(local_rank)  # no CPU fallback: raises if the HIP extension / MI355X is missing
    weights = resnet50.make_weights()
    # weight arena lives in a torch allocation so RCCL can broadcast it
    net = None
    arena_t = None
    if world > 1:
        tmp = resnet50.ResNet50(ctx, BATCH_PER_GPU, weights)
        nbytes = tmp.arena_bytes
        del tmp
        arena_t = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{local_rank}")
        net = resnet50.ResNet50(ctx, BATCH_PER_GPU, weights, arena_ptr=arena_t.data_ptr(), arena_keepalive=arena_t)
        if rank == 0:
            net.upload_weights()
        ctx.sync()
        broadcast_weight_arena(arena_t, src=0)  # RCCL over xGMI, once
        torch.cuda.synchronize()
    else:
        net = resnet50.ResNet50(ctx, BATCH_PER_GPU, weights)
        net.upload_weights()

    # each rank gets its own (independent) synthetic batch, resident in HBM before timing starts
    x = np.random.default_rng(1234 + rank).random((BATCH_PER_GPU, 3, 224, 224), dtype=np.float32)
    net.x.upload(x)
    ctx.sync()

    table = None
    net.concurrent = args.concurrent
    if args.load_plan:
        net.variants = {k: tuple(v) for k, v in json.load(open(args.load_plan)).items()}
        args.no_autotune = True
    if not args.no_autotune:
        table = net.autotune(reps=3)
        if args.save_plan and rank == 0:
            json.dump({k: list(v) for k, v in net.variants.items()}, open(args.save_plan, "w"))
        if args.layer_table and rank == 0:
            for l in net.specs:
                d = net.descs[l["name"]]
                fl = 2.0 * d.o * d.c * d.kh * d.kw * d.out_h * d.out_w * d.n
                row = table[l["name"]]
                best_ms = min(ms for _, ms in row)
                nosplit = " ".join(f"v{p[0]}={ms*1e3:6.1f}" for p, ms in row if p[1] == 0 and p[3] == 0)
                split = sorted(((ms, p) for p, ms in row if p[1] != 0 or p[3] != 0))[:4]
                print(f"[layer] {l['name']:8s} O={d.o:4d} C={d.c:4d} k={d.kh} s={d.stride_h} {d.h:3d}->{d.out_h:3d} us: {nosplit}"
                      + " | split " + " ".join(f"v{p[0]}m{p[1]}g{p[2]}o{p[3]}={ms*1e3:6.1f}" for ms, p in split)
                      + f"  best={net.variants[l['name']]} {fl / (best_ms * 1e-3) / 1e12:6.1f} TF/s", file=sys.stderr)
    if not args.no_graph:
        net.capture()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        net.run()
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.run()
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- p50 latency per batch (separate pass, host-timed per step)
    lat = []
    for _ in range(min(args.steps, 20)):
        t1 = time.perf_counter()
        net.run()
        ctx.sync()
        lat.append((time.perf_counter() - t1) * 1e3)
    p50 = float(np.median(lat))

    # ---- roofline of the dominant kernel: instrumented eager pass over the same K steps (HIP events per launch
    #      on the backend's stream).  Kept out of the timed region so `value` is not perturbed.
    roof = None
    if rank == 0:
        ctx.profile_reset()
        ctx.profile(True)
        saved_graph, net.graph = net.graph, None
        saved_conc, net.concurrent = net.concurrent, False  # serialised launches: clean per-kernel durations
        for _ in range(args.steps):
            net.forward()
        ctx.sync()
        ctx.profile(False)
        net.graph, net.concurrent = saved_graph, saved_conc
        rep = ctx.profile_report()
        conv = [r for r in rep if r["kernel"].startswith("igemm_f32")]
        tot_ms = sum(r["ms"] for r in rep)
        if conv:
            dom = max(conv, key=lambda r: r["ms"])
            fam_ms = sum(r["ms"] for r in conv)
            fam_fl = sum(r["flops"] for r in conv)
            roof = {"bound": "mfma", "kernel": dom["kernel"],
                    "achieved": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 3), "peak": F32_MATRIX_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / F32_MATRIX_PEAK_TFLOPS, 4),
                    "traffic": None, "traffic_source": None,
                    "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2), "launches": dom["launches"],
                    "flops_per_launch": dom["flops"] / max(dom["launches"], 1),
                    "kernel_share_of_step": round(dom["ms"] / max(tot_ms, 1e-9), 4),
                    "igemm_family": {"achieved": round(fam_fl / (fam_ms * 1e-3) / 1e12, 3),
                                     "frac": round(fam_fl / (fam_ms * 1e-3) / 1e12 / F32_MATRIX_PEAK_TFLOPS, 4),
                                     "share_of_step": round(fam_ms / max(tot_ms, 1e-9), 4),
                                     "variants": {r["kernel"]: {"launches": r["launches"], "ms": round(r["ms"], 4),
                                                                "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2)} for r in conv}}}

    if rank == 0 and roof:
        # HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected separately by
        # tools/gpu/traffic.sh over the same tuned plan -- counters cannot be collected inside the timed run)
        import glob
        for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "hbm_traffic_per_kernel.json")), reverse=True):
            t = json.load(open(path)).get("kernels", {}).get(roof["kernel"].replace(" ", ""))
            if t:
                roof["traffic"] = t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"]
                roof["traffic_source"] = os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))
                break
    if rank == 0:
        global_batch = BATCH_PER_GPU * n_gpus
        value = global_batch * args.steps / elapsed
        out = {
            "metric": "inferences/sec, ResNet-50 f32 batch 32 per GPU",
            "value": round(value, 2), "unit": "inferences/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "p50_latency_ms": round(p50, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ResNet-50 v1.5 f32 inference, 224x224, batch 32 per GPU (BASELINE configs[1]); "
                                   "synthetic He-normal BN-folded weights (seed 1234), inputs U[0,1) resident in HBM",
                       "global_batch": global_batch, "parallelism": f"batch-shard x{n_gpus} (weights RCCL-broadcast once)" if n_gpus > 1 else "single GPU",
                       "launch": "eager" if args.no_graph else "hipGraph replay", "autotuned_tiles": bool(net.variants), "shortcut_branch": "second stream" if net.concurrent else "main stream",
                       "gflop_per_image": round((resnet50.conv_flops_per_image() + 2 * 2048 * 1000) / 1e9, 3),
                       "device": ctx.device_info()},
            "roofline": roof,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net.specs, weights)
        else:
            out["cpu_baseline"] = None
        if n_gpus == 1 and world == 1 and not args.no_secondary:
            out["secondary"] = secondary_configs()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()  # rank 0's instrumented pass / JSON line happen before any rank tears the group down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
