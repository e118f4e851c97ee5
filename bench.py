#!/usr/bin/env python3
"""bench.py -- ResNet-50 f32, batch 32 per GPU, on the HIP backend (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W [--config f32|int8] [--lanes L] [--chains C] [--via-runner]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` always means N ranks, one per GPU: without a launcher the script spawns them itself (torch.distributed.run);
under a launcher it refuses to run when WORLD_SIZE != N.  `--config int8` runs the dynamically quantized graph
(BASELINE configs[2]; with --gpus 8: configs[4]) with its own roofline / cpu_baseline objects.

What is timed is the PRODUCT path: the model as ONNX bytes through the C++ plan executor behind the C ABI (rten_hip_model_load_ex / _prepare / _run:
what a Rust host binds); `--via-runner` times round 4's hand-planned Python runner instead (A/B).
A step = one forward pass of ResNet-50 (53 convs + maxpool + global-avg-pool + fc) over one batch of 32 synthetic 224x224 images that is already
resident in HBM: one hipGraph replay per chain.  Steps are independent batches, so consecutive steps go round robin to `--lanes` REPLICAS of the model
(rten_hip_model_clone: own stream, buffers and hipGraphs, one shared weight arena) and overlap on the device -- f32: one whole-batch chain per replica,
4 lanes under per-layer plans chosen under co-run (profiles/plans/f32_lanes.json, tools/tune_corun.py); int8 (whose quantizers span the batch: no sub-batch chains): 4 lanes.  `ms_per_step` / `value` are therefore THROUGHPUT figures over the K timed
steps (both synchronisation points cover every stream of every lane); `ms_per_step_joined_every_step` and `p50_latency_ms` are ONE batch on ONE replica.
`--chains C` alone gives round 4's schedule (one replica, C sub-batch chains).  One process per GPU; batches are independent, so the path shards with no
data-path collective (weak scaling: 32 images per GPU); the only collective is the one-time RCCL broadcast of the model's weight arena from rank 0.

Rank 0 prints ONE COMPACT JSON line (at most 4096 bytes: `compact_line`; the driver keeps a bounded tail of stdout, and round 5's 23 KB line came back
unparsed) as the LAST thing on stdout: the contract fields, `roofline` -- f32: achieved / frac = the conv FLOPs of one batch over the TIMED step (every kernel
and gap included) with the dominant kernel's stand-alone figures (HIP events per launch, an instrumented eager pass outside the timed region) as
`dominant_kernel` -- plus, under lanes, `dominant_kernel.co_run`: the dominant layer family on `lanes` streams at once, the state a launch of the timed schedule
runs in and the way the lanes plan's entries were chosen (rten_amd/workloads/corun.py) -- and its HBM traffic from the committed PMC pass when that pass ran the same launch plan; int8: the dominant kernel against the HBM peak plus
the whole step against the graph's HBM floor (`step`) --, `cpu_baseline` (the CPU oracle -- a port of the reference algorithm -- timed on this host's cores on a
bounded sample; N=1 only) and, at N=1, a few numbers per `secondary` config: the f32 batch as 4 chains on one replica (the latency-optimal schedule:
`p50_latency_ms_4chains`), the int8 ResNet-50 (configs[2]), BERT-base (configs[3]) and the batch-1 latencies (configs[0]), each run in a child process after the
headline measurement.  The FULL record (per-shape table `roofline.shapes`, per-variant tables, notes, per-rank lists, the PCIe-inclusive pass, every secondary
line whole) goes to the detail file the line names (`detail`: gpurun_out/bench_detail.json, or --detail-file).
`ms_per_step` is, in every round-6 line, wall time of the K timed steps / K on the default schedule (lanes: consecutive batches overlap), i.e. a THROUGHPUT
figure; the time of ONE batch alone is `ms_per_step_joined_every_step` / `p50_latency_ms` (DESIGN.md section 6 fixes these definitions): one replica with nothing
beside it under the plan a single replica is given (f32_1chain.json; the lanes plan is chosen for company and is slower alone: `lanes_plan_alone`).
The per-layer launch plan is the one committed under profiles/plans/ (`--autotune` re-tunes: rank 0 tunes, the plan is broadcast; `config.launch_plan` names
what ran).  Defaults: K = 50, W = 20 (the chip needs about 20 ms of load to settle its clocks; a run with the driver's own K / W is timed exactly as given).
"""
import argparse
import json
import os
import sys
import time

# Every stream of the process gets a hardware queue of its own (the runtime's default maps streams onto 4 queues, and two of the f32
# runner's sub-batch chains on one queue serialise: 3.45 vs 2.79 ms per step, profiles/r05/chains_probe.txt).  Must be set before the HIP
# runtime starts, i.e. before torch / librten_hip.so are loaded; with it no stream-placement search is needed.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Control-flow test mode (tests/test_bench_world8.py): RTEN_BENCH_RECORDING=1 together with --recording-test swaps the device context for one that records launches instead of
# issuing them, keeps every tensor torch touches on the CPU and takes the gloo backend -- the script's rank / shard / plan / broadcast / aggregation
# logic runs unchanged at any world size without a GPU.  Numbers printed in this mode mean nothing and say so (`data`: "recording").
DRY = os.environ.get("RTEN_BENCH_RECORDING") == "1"

F32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X v_mfma_f32_32x32x2_f32 peak (MI355X_MICROARCH.md)
I8_MATRIX_PEAK_TOPS = 5033.0    # dense i8 MFMA: 2x the bf16 rate (MI355X_MICROARCH.md, "Matrix cores")
HBM_PEAK_GBS = 8000.0           # HBM3E spec (6.29 TB/s measured with a float4 copy)
BATCH_PER_GPU = 32
# Consecutive batches on independent replicas of the model ("lanes", --lanes): measured in session r5e (profiles/r08/lanes.txt) -- int8 1 / 2 / 3 / 4 lanes:
# 1.502 / 1.052 / 0.971 / 0.950 ms per batch; f32 one chain x 2 lanes 2.494 ms against 2.677 ms for one replica running the batch as 4 sub-batch chains
INT8_DEFAULT_LANES = 4
F32_DEFAULT_LANES = 4   # ... of ONE chain each (whole-batch launches), unless --chains is given.  Round 6, per-layer plans chosen under co-run (f32_lanes.json), same box, two
#                         alternating rounds (profiles/r10/f32_lanes_sweep_corun_plan.txt): 2 lanes 2.55 / 2.56 ms, 3 lanes 2.42 / 2.43, 4 lanes 2.39 / 2.40, 5 lanes 2.46 / 2.45


LINE_LIMIT = 4096   # the driver keeps a bounded tail of stdout: the final line must fit it whole (round 5's 23 KB line came back `parsed: null`)


def _sanitize(o):
    """JSON has no NaN / Infinity: they become null (json.dumps would print the bare words, which strict parsers reject)."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {str(k): _sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_sanitize(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return _sanitize(o.item())
    return o


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _compact_roofline(r):
    if not isinstance(r, dict):
        return None
    c = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"))
    c.setdefault("traffic", None)
    if isinstance(r.get("dominant_kernel"), dict):  # f32: stand-alone figures of the dominant kernel (HIP events per launch) beside the whole-step fraction
        c["dominant_kernel"] = _pick(r["dominant_kernel"], ("achieved", "frac", "avg_launch_us", "launches"))
        if isinstance(r["dominant_kernel"].get("co_run"), dict):  # ... and of its layer family under the schedule that is timed: the same layer on `streams` streams at once
            c["dominant_kernel"]["co_run"] = _pick(r["dominant_kernel"]["co_run"], ("layer", "streams", "avg_launch_us", "achieved", "frac"))
    elif "avg_launch_us" in r:
        c["avg_launch_us"] = r["avg_launch_us"]
    if isinstance(r.get("step"), dict):             # int8: the whole timed step against the HBM floor of the graph
        c["step"] = _pick(r["step"], ("algorithmic_bytes", "achieved", "frac"))
    if isinstance(r.get("mfma"), dict):
        c["mfma_frac"] = r["mfma"].get("frac")
    for k in ("gemm_family", "fused_attention", "rowwise"):  # BERT: the three kernel classes, fraction of their own bound
        if isinstance(r.get(k), dict):
            c[k + "_frac"] = r[k].get("frac")
    return c


def _compact_cpu(c):
    if not isinstance(c, dict):
        return None
    o = _pick(c, ("value", "unit", "cores", "kind"))
    if "sample" in c:
        o["sample"] = str(c["sample"])[:120]
    if isinstance(c.get("other_cpu_implementation"), dict):
        o["pytorch_cpu_value"] = c["other_cpu_implementation"].get("value")
    return o


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: every contract field, the roofline / cpu_baseline objects and a few numbers per secondary config -- at most
    LINE_LIMIT bytes.  Everything else (`roofline.shapes`, per-variant tables, notes, per-rank lists, the PCIe-inclusive pass) lives in the detail file."""
    cfg = out.get("config") or {}
    lp = cfg.get("launch_plan") or {}
    lanes = (cfg.get("batch_lanes") or {}).get("lanes", lp.get("lanes"))
    chains = (cfg.get("batch_chains") or {}).get("chains", lp.get("chains"))
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_joined_every_step", "p50_latency_ms",
                                "p50_latency_ms_4chains", "lanes_plan_alone", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:260], "path": cfg.get("path"), "lanes": lanes, "chains": chains,
                      "global_batch": cfg.get("global_batch"), "parallelism": cfg.get("parallelism"), "launch": cfg.get("launch"),
                      "launch_plan": _pick(lp, ("source", "sha16", "identical_on_all_ranks"))}
    rk = out.get("ranks")
    if isinstance(rk, dict):
        line["ranks"] = {"world_size": rk.get("world_size"), "dist_backend": rk.get("dist_backend"), "weight_broadcast_world": rk.get("weight_broadcast_world"),
                         "ms_per_step_per_rank": rk.get("ms_per_step_per_rank"), "distinct_plans": len(set(map(str, rk.get("plan_sha16_per_rank") or [])))}
    line["roofline"] = _compact_roofline(out.get("roofline"))
    line["cpu_baseline"] = _compact_cpu(out.get("cpu_baseline"))
    if isinstance(out.get("pcie_inclusive"), dict):
        line["pcie_inclusive_ms_per_step"] = out["pcie_inclusive"].get("ms_per_step")
    if isinstance(out.get("secondary"), dict):
        sec = {}
        for name, s in out["secondary"].items():
            if not isinstance(s, dict):
                continue
            if "error" in s:
                sec[name] = {"error": str(s["error"])[:160]}
                continue
            e = _pick(s, ("value", "unit", "ms_per_step", "ms_per_step_joined_every_step", "ms_per_step_back_to_back", "p50_latency_ms", "dtype"))
            scfg = s.get("config") or {}
            e["lanes"] = (scfg.get("batch_lanes") or {}).get("lanes", scfg.get("lanes", (scfg.get("launch_plan") or {}).get("lanes")))
            e["plan_sha16"] = (scfg.get("launch_plan") or {}).get("sha16")
            if s.get("roofline"):
                e["roofline"] = _compact_roofline(s["roofline"])
            if s.get("cpu_baseline"):
                e["cpu_baseline"] = _pick(s["cpu_baseline"], ("value", "unit", "cores", "kind"))
            sec[name] = {k: v for k, v in e.items() if v is not None}
        line["secondary"] = sec
    if detail_path:
        line["detail"] = detail_path
    line = _sanitize(line)
    # the size is a CONTRACT: shed optional parts, least important first, until the line fits
    text = json.dumps(line, separators=(",", ":"), allow_nan=False)
    for shed in (("secondary", "*", "cpu_baseline", "unit"), ("secondary", "*", "dtype"), ("cpu_baseline", "sample"), ("secondary", "*", "roofline", "kernel"),
                 ("ranks", "ms_per_step_per_rank"), ("config", "workload"), ("secondary",)):
        if len(text) <= LINE_LIMIT:
            break
        tgt = [line]
        for key in shed[:-1]:
            tgt = [v for t in tgt if isinstance(t, dict) for v in (t.values() if key == "*" else [t.get(key)])]
        for t in tgt:
            if isinstance(t, dict):
                t.pop(shed[-1], None)
        text = json.dumps(line, separators=(",", ":"), allow_nan=False)
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def emit(out, args, key=None):
    """Writes the full record to the detail file and prints the compact line (last thing on stdout)."""
    path = args.detail_file or os.path.join(ROOT, "gpurun_out", f"bench_detail{'_' + key if key else ''}.json")
    rel = None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(_sanitize(out), f, indent=1)
        rel = os.path.relpath(os.path.abspath(path), ROOT)
        if rel.startswith(".."):
            rel = os.path.abspath(path)
    except OSError as e:  # a read-only tree loses the detail, never the line
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    sys.stdout.flush()
    print(compact_line(out, rel), flush=True)


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


def torch_cpu_reference(specs, weights, budget_s=8.0, int8=False):
    """A COMPETENT CPU number beside the oracle's: the same ResNet-50 graph through PyTorch's CPU kernels (oneDNN) on this
    host's cores.  It is NOT the reference (RTen's own CPU path cannot be built here: no Rust toolchain) and not the parity
    checker -- only context for `cpu_baseline.value`, which times a plain restatement of the reference's algorithm."""
    try:
        import torch
        import torch.nn.functional as F
        torch.set_grad_enabled(False)
        tw = {k: (torch.from_numpy(w), torch.from_numpy(b)) for k, (w, b) in weights.items()}
        x = torch.rand(32, 3, 224, 224)

        def fwd(x):
            acts = {"x": x}
            for i, l in enumerate(specs):
                w, b = tw[l["name"]]
                y = F.conv2d(acts[l["src"]], w, b, stride=l["stride"], padding=l["pad"])
                if l["res"]:
                    y = y + acts[l["res"]]
                if l["relu"]:
                    y = F.relu(y)
                acts[l["dst"]] = y
                if i == 0:
                    acts["pool"] = F.max_pool2d(y, 3, 2, 1)
            g = acts[specs[-1]["dst"]].mean((2, 3))
            return F.linear(g, tw["fc"][0], tw["fc"][1])
        fwd(x)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget_s and reps < 20:
            fwd(x)
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(32 * reps / dt, 1), "unit": "inferences/s", "cores": torch.get_num_threads(), "kind": "pytorch-cpu (oneDNN), f32; not the reference",
                "sample": f"{32 * reps} images (batch 32 x {reps} forward passes, {dt:.1f} s)"}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}


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


def cpu_baseline_int8(specs, weights, budget_s=12.0):
    """cpu_baseline leg of --config int8: the CPU oracle's dynamically quantized ResNet-50 (port of the reference algorithm:
    DynamicQuantizeLinear -> ConvInteger -> cast_scale -> Add [-> Add] [-> Relu] per conv) on a bounded sample."""
    from oracle import models as omodels
    from oracle import ref
    threads = ref.num_threads()
    q = omodels.quantize_weights_int8(weights)
    x = ref.XorShiftRng(7).f32(2 * 3 * 224 * 224).reshape(2, 3, 224, 224)
    t0 = time.perf_counter()
    omodels.resnet50_int8_forward(specs, q, x)
    t_first = time.perf_counter() - t0
    reps = int(max(1, min(16, budget_s / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        omodels.resnet50_int8_forward(specs, q, x)
    dt = time.perf_counter() - t0
    imgs = 2 * reps
    return {"value": round(imgs / dt, 3), "unit": "inferences/s", "cores": threads, "kind": "port",
            "sample": f"{imgs} images (batch 2 x {reps} forward passes) of the same dynamically quantized ResNet-50 graph through oracle/rten_oracle.c "
                      f"({threads} OpenMP threads, {dt:.1f} s)"}


def int8_algorithmic_bytes(net, as_launched=False):
    """HBM bytes one forward pass of the dynamically quantized graph must move, per DESIGN.md section 7 / SURVEY 8(d): every
    conv output is an f32 tensor of the graph (4 B write), its consumer's DynamicQuantizeLinear reads it (4 B; one
    quantization per distinct tensor) and writes u8 codes (1 B), every conv reads those codes once (1 B) plus its weights,
    a residual Add reads 4 B.  The min/max sweep of DynamicQuantizeLinear is NOT counted (the producer's epilogue
    accumulates it), nor are the staged image's padding bytes: this is the floor, not what the kernels happen to move.
    as_launched: the floor of the launch sequence actually run -- a quantized-output launch (rten_hip_conv2d_int8_qout) writes the
    consumer's codes itself, so the quantizer's 4 B read disappears, and the 4 B f32 write too unless a residual needs the tensor."""
    qout = (set(net.qout_next) - net._qout_off) if (as_launched and getattr(net, "fused_qout", False)) else set()
    by_dst = {l["dst"]: l["name"] for l in net.specs}
    total, quantized = 0.0, set()
    for l in net.specs:
        d = net.descs[l["name"]]
        in_elems = d.n * d.c * d.h * d.w
        out_elems = d.n * d.o * d.out_h * d.out_w
        if l["src"] not in quantized:
            quantized.add(l["src"])
            total += 1.0 * in_elems if by_dst.get(l["src"]) in qout else 5.0 * in_elems  # quantize: (f32 read +) u8 write
        total += 1.0 * in_elems              # conv reads the codes
        total += d.o * d.c * d.kh * d.kw     # i8 weights
        if not (l["name"] in qout and l["name"] not in net.qout_keeps_f32):
            total += 4.0 * out_elems         # f32 output
        if l["res"]:
            total += 4.0 * out_elems         # residual read
    p = net.pool_desc
    total += 4.0 * p.n * p.c * (p.h * p.w + p.out_h * p.out_w)      # MaxPool
    last = net.shapes[net.specs[-1]["dst"]]
    total += 4.0 * last[0] * last[1] * (last[2] * last[3] + 1)      # GlobalAveragePool
    total += 2048.0 * net.num_classes + 8.0 * last[0] * net.num_classes + 9.0 * last[0] * 2048  # classifier
    return total


def dominant_co_run(plan, streams, ctxs=None):
    """The convolution family with the largest share of the FLOPs (stage 2's 3x3 layers) under its committed plan entry, timed the way the lanes plans are chosen
    (rten_amd/workloads/corun.py): `streams` runner networks sharing one weight arena run the SAME layer at once on their real input activations, a captured graph
    of 12 launches per stream; microseconds per launch over all streams, and 2 M N K of the layer over that."""
    import gc
    from rten_amd.workloads.corun import CoRun
    cr = CoRun(streams, BATCH_PER_GPU, plan, ctxs=ctxs)  # (on the lanes' own contexts: their streams already hold a hardware queue each)
    try:
        fams = cr.families()
        key = max(fams, key=lambda k: sum(cr.flops(l["name"]) for _, l in fams[k]))
        idx, l = fams[key][1] if len(fams[key]) > 1 else fams[key][0]
        entry = list(plan[l["name"]])
        us = min(cr.measure(idx, entry) for _ in range(2))
        fl = cr.flops(l["name"])
        return {"layer": l["name"], "layers_of_this_family": len(fams[key]), "share_of_conv_flops": round(fl * len(fams[key]) / sum(cr.flops(x["name"]) for x in cr.specs), 4),
                "plan": entry, "streams": streams, "avg_launch_us": round(us, 2), "achieved": round(fl / us / 1e6, 2), "unit": "TFLOP/s",
                "frac": round(fl / us / 1e6 / F32_MATRIX_PEAK_TFLOPS, 4),
                "what": "the same layer on `streams` streams at once (each on its real input activations, a captured graph of 12 launches per stream): time per launch over "
                        "all streams -- the state a launch of the timed lanes schedule runs in, and how the plan's entries were chosen (tools/tune_corun.py)"}
    finally:
        cr.close()
        gc.collect()


def secondary_configs():
    """The other single-GPU configs of BASELINE.json, measured in child processes after the headline run (same backend, same
    box): reported for the record, never part of `value`.  The int8 ResNet-50 (configs[2], named in BASELINE's metric) is a
    full bench line of its own -- `python bench.py --config int8` -- with its roofline and cpu_baseline objects."""
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    res = {}
    # the latency-optimal f32 schedule beside the throughput one: ONE replica running the batch as 4 sub-batch chains (round 4's schedule, its own committed
    # plan), in a process of its own (idle lane streams beside the four chains cost it 25 %: 3.5 vs 2.7-2.8 ms)
    jobs = (("resnet50_f32_b32_4chains_1lane", [os.path.join(root, "bench.py"), "--chains", "4", "--lanes", "1", "--no-secondary", "--no-cpu-baseline", "--no-shapes",
                                                "--detail-file", os.path.join(root, "gpurun_out", "bench_detail_f32_4chains.json")]),
            ("resnet50_int8_b32", [os.path.join(root, "bench.py"), "--config", "int8", "--no-secondary"]),
            ("bert_base_f32_b32_s128", [os.path.join(root, "tools", "bench_bert.py")]),
            ("resnet50_f32_b1_latency", [os.path.join(root, "tools", "bench_resnet50_b1.py")]),
            ("resnet50_int8_b1_latency", [os.path.join(root, "tools", "bench_resnet50_b1.py"), "--config", "int8"]))
    for key, cmd in jobs:
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
            p = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=420, cwd=root, env=env)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            if j.get("detail"):  # a child that prints a compact line of its own (bench.py --config int8) keeps its full record in a file
                dpath = j["detail"] if os.path.isabs(j["detail"]) else os.path.join(root, j["detail"])
                if os.path.exists(dpath):
                    j = json.load(open(dpath))
            res[key] = {k: j[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_joined_every_step", "p50_latency_ms", "ms_per_step_back_to_back",
                                          "dtype", "config", "roofline", "cpu_baseline") if k in j}
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


def int8_graph_floor_bytes(batch, qout=(), num_classes=1000):
    """int8_algorithmic_bytes() from host arithmetic alone (the executor path has no runner object): the HBM floor of the dynamically quantized graph
    as the reference runs it -- every conv output an f32 tensor (4 B write), one DynamicQuantizeLinear per distinct tensor (4 B read + 1 B codes), every
    conv reads its codes (1 B) and weights, a residual Add reads 4 B; max-pool, global average pool and the classifier as their operands.
    `qout`: the floor of the launch sequence actually run -- a quantized-output launch writes its consumer's codes itself, so the quantizer's 4 B read
    disappears, and the 4 B f32 write too unless a residual Add needs the tensor."""
    from rten_amd.workloads import resnet50
    shapes, descs = resnet50.layer_geometry(batch)
    specs = resnet50.conv_specs()
    by_dst = {l["dst"]: l["name"] for l in specs}
    residuals = {l["res"] for l in specs if l["res"]}
    qout = set(qout)
    total, quantized = 0.0, set()
    for l in specs:
        d = descs[l["name"]]
        in_elems, out_elems = d.n * d.c * d.h * d.w, d.n * d.o * d.out_h * d.out_w
        if l["src"] not in quantized:
            quantized.add(l["src"])
            total += 1.0 * in_elems if by_dst.get(l["src"]) in qout else 5.0 * in_elems
        total += 1.0 * in_elems + d.o * d.c * d.kh * d.kw
        if not (l["name"] in qout and l["dst"] not in residuals):
            total += 4.0 * out_elems
        if l["res"]:
            total += 4.0 * out_elems
    n, c, h, w = shapes["stem"]
    ph = shapes["pool"][2]
    total += 4.0 * n * c * (h * w + ph * ph)
    last = shapes[specs[-1]["dst"]]
    total += 4.0 * last[0] * last[1] * (last[2] * last[3] + 1)
    total += 2048.0 * num_classes + 8.0 * last[0] * num_classes + 9.0 * last[0] * 2048
    return total


def attach_traffic(roof, plan_sha, int8):
    """HBM bytes per launch of the dominant kernel from a SEPARATE PMC pass (FETCH_SIZE x2 + WRITE_SIZE, tools/gpu/traffic.sh: counters cannot be
    collected inside the timed run).  The figure belongs to the plan that pass ran under, named in `traffic_source` / `traffic_note`; it is null when
    no committed pass covers this kernel instantiation under the same launch plan."""
    import glob
    if not roof or "kernel" not in roof:
        return
    fname = "int8_hbm_traffic_per_kernel.json" if int8 else "hbm_traffic_per_kernel.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", fname)), reverse=True):
        prof = json.load(open(path))
        ks = prof.get("kernels", {})
        name = roof["kernel"].replace(" ", "")
        # the profiler prints every template argument (defaults included): match on the prefix the backend's own label gives
        cands = [k for k in ks if k == name or k.startswith(name[:-1] + ",")]
        if not cands:
            continue
        best = max(cands, key=lambda k: ks[k].get("launches", 0))
        t = ks[best]
        total = t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"]
        same_plan = plan_sha is not None and prof.get("plan_sha16") == plan_sha
        roof["traffic"] = total if same_plan else None  # a figure measured under another launch plan is not this run's traffic
        roof["traffic_kernel"] = best
        roof["traffic_source"] = os.path.relpath(path, ROOT)
        roof["traffic_plan_sha16"] = prof.get("plan_sha16")
        roof["traffic_note"] = ("HBM bytes per launch of the dominant kernel (FETCH_SIZE x 2 + WRITE_SIZE) from a separate rocprofv3 --pmc pass over the SAME launch plan "
                                "(plan_sha16 matches; counters cannot be collected inside the timed run)" if same_plan else
                                f"null: the committed PMC pass ran another launch plan (its figure for this kernel: {total} B per launch)")
        break


def per_shape_table(ctx, plan_1chain, reps=8):
    """`roofline.shapes`: every distinct convolution shape of ResNet-50 at batch 32, launched STAND-ALONE through
    rten_hip_conv2d_f32 under the committed one-chain plan (HIP-event timers around `reps` back-to-back launches, realistic operands), with the bound
    the shape itself allows: t_attainable = max(HBM time of its algorithmic bytes at 8 TB/s, MFMA time / tile-quantisation efficiency), where the
    quantisation efficiency of T workgroup tiles on 256 compute units is T / (256 * ceil(T / 256)).  north_star words its target per dominant shape."""
    from rten_amd.workloads import resnet50
    net = resnet50.ResNet50(ctx, BATCH_PER_GPU)
    net.upload_weights()
    net.x.upload(np.random.default_rng(0).random(net.shapes["x"], dtype=np.float32))
    net.forward()
    ctx.sync()
    tile_of = {0: (128, 128), 1: (128, 64), 2: (64, 128), 3: (64, 64)}
    groups = {}
    for l in net.specs:
        d = net.descs[l["name"]]
        key = (d.o, d.c, d.kh, d.stride_h, d.h, bool(l["res"]))
        groups.setdefault(key, []).append(l)
    total_fl = sum(2.0 * net.descs[l["name"]].o * net.descs[l["name"]].c * net.descs[l["name"]].kh ** 2 * net.descs[l["name"]].out_h * net.descs[l["name"]].out_w * BATCH_PER_GPU
                   for l in net.specs)
    rows = []
    for key, ls in groups.items():
        l = ls[0]
        d = net.descs[l["name"]]
        plan = tuple((plan_1chain or {}).get(l["name"], (3, 0, 1, 0)))
        net.variants[l["name"]] = plan
        net._conv(l)
        ctx.sync()
        best = 1e30
        for _ in range(2):
            ctx.timer_start(3)
            for _ in range(reps):
                net._conv(l)
            ctx.timer_stop(3)
            best = min(best, ctx.timer_ms(3) / reps)
        M, K, N = d.o, d.c * d.kh * d.kw, d.n * d.out_h * d.out_w
        fl = 2.0 * M * K * N
        alg_bytes = 4.0 * (d.n * d.c * d.h * d.w + M * N * (2 if l["res"] else 1) + M * K)
        v = plan[0]
        bm, bn = (32, 32) if v in (28, 29) else (64, 64) if v >= 24 else tile_of[v % 4]
        tiles = -(-M // bm) * -(-N // bn)
        q = tiles / (256.0 * -(-tiles // 256))
        t_mfma, t_hbm = fl / (F32_MATRIX_PEAK_TFLOPS * 1e12), alg_bytes / (HBM_PEAK_GBS * 1e9)
        t_att = max(t_hbm, t_mfma / q)
        rows.append({"shape": f"O{d.o} C{d.c} k{d.kh} s{d.stride_h} {d.h}x{d.w}->{d.out_h}x{d.out_w}" + (" +res" if l["res"] else ""), "layers": len(ls), "example": l["name"],
                     "M": M, "K": K, "N": N, "plan": list(plan), "us": round(best * 1e3, 2), "tflops": round(fl / (best * 1e-3) / 1e12, 2),
                     "frac_mfma": round(fl / (best * 1e-3) / 1e12 / F32_MATRIX_PEAK_TFLOPS, 4), "frac_hbm": round(alg_bytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "tiles": tiles, "tiles_per_cu": round(tiles / 256.0, 3), "quantisation_eff": round(q, 4),
                     "attainable": {"us": round(t_att * 1e6, 2), "bound": "hbm" if t_hbm >= t_mfma / q else "mfma x tile quantisation", "frac_mfma": round(t_mfma / t_att, 4)},
                     "frac_of_attainable": round(t_att / (best * 1e-3), 4), "share_of_conv_flops": round(fl * len(ls) / total_fl, 4)})
    rows.sort(key=lambda r: -r["share_of_conv_flops"])
    return rows


def run_via_executor(args):
    """The DEFAULT path: the workload through the product path a Rust host binds -- the C++ plan executor behind the C ABI (rten_hip_model_*: ONNX
    bytes in, values resident in HBM, the committed launch plan, sub-batch chains, hipGraph replay).  One process per GPU; every rank loads the model,
    rank 0 for real, the others with RTEN_HIP_MODEL_RECEIVE_WEIGHTS, and the weight arena (one allocation: every constant + prepacked weight) is
    broadcast once from rank 0 -- by the backend's own RCCL binding (rten_hip_comm_*) under the nccl backend."""
    import hashlib
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not under_launcher and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a number for GPUs that are not running",
              file=sys.stderr)
        return 2
    import torch
    dist, backend = None, "none"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if DRY else os.environ.get("RTEN_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        if not DRY:
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif not DRY:
        torch.cuda.set_device(local_rank)

    from rten_amd import lib, onnx_writer
    from rten_amd.tensor import DeviceTensor
    from rten_amd.workloads import resnet50
    int8 = args.config == "int8"
    if DRY:
        from rten_amd.recording import RecordingCtx, RecordingModel
        ctx, Model = RecordingCtx(local_rank), RecordingModel
        torch.cuda.synchronize = lambda *a, **k: None  # (nothing is ever enqueued on a device in this mode)
    else:
        ctx, Model = lib.Context(local_rank), lib.Model  # no CPU fallback: raises if the HIP extension / MI355X is missing
    weights = resnet50.make_weights()
    # f32 default: ONE chain per replica and two replicas; `--chains N` alone keeps round 4's schedule (one replica, N sub-batch chains)
    chains = 1 if (int8 or args.chains is None) else args.chains
    onnx_bytes = onnx_writer.resnet50_int8(weights) if int8 else onnx_writer.resnet50_f32(weights)
    lanes = args.lanes if args.lanes else (INT8_DEFAULT_LANES if int8 else (F32_DEFAULT_LANES if args.chains is None else 1))
    # int8: one replica alone may use quantized-output launches (int8.json); replicas running side by side may not (those launches need the device to
    # themselves): int8_lanes.json lists only the quantize-on-load layers
    # f32 replicas side by side: f32_lanes.json -- per-layer plans chosen UNDER CO-RUN (tools/tune_corun.py, round 6: three streams running the same layer; larger tiles
    # where the other replicas fill their tile-quantisation gaps: fewer bytes through LDS / L2 per FLOP, so the shader clock holds on real operand data,
    # tools/probes/kloop2.hip) plus the classifier on its 64x64 tiles -- the small-M streaming kernel the backend picks on its own is the faster launch alone
    # (8 vs 14 us) but spreads over every compute unit, which costs the other replicas more than it saves (same-box A/B, profiles/r08/classifier_under_lanes.txt)
    f32_plan = "f32_lanes.json" if (lanes > 1 and chains == 1) else f"f32_{chains}chain{'s' if chains > 1 else ''}.json"
    default_plan = os.path.join(ROOT, "profiles", "plans", ("int8.json" if lanes == 1 else "int8_lanes.json") if int8 else f32_plan)
    plan_path = args.load_plan or default_plan
    plan_text, plan_source = None, "backend defaults (no plan)"
    if not args.no_autotune and not args.autotune and os.path.exists(plan_path):
        plan_text, plan_source = open(plan_path).read(), os.path.relpath(os.path.abspath(plan_path), ROOT)
    if plan_text and int8 and (args.no_qout or lanes > 1 or (world > 1 and backend != "nccl")):
        # quantized-output launches need every workgroup of a launch resident at once and the device to themselves: not when several ranks share ONE
        # GPU (the gloo test mode), and not when a second replica's launches run beside them (lanes > 1)
        p = json.loads(plan_text)
        p.pop("qout", None)
        plan_text = json.dumps(p)

    def load(text):
        return Model(ctx, onnx_bytes, text, chains, receive_weights=(rank != 0))

    # ---- the launch plan every rank runs: the committed file, or (--autotune, f32) rank 0 tunes at load and the result is broadcast
    if args.autotune and not int8:
        tuned = [None]
        if rank == 0:
            m0 = Model(ctx, onnx_bytes, None, chains)
            m0.bind_input("x", (BATCH_PER_GPU, 3, 224, 224))
            m0.prepare(tune=True)
            tuned[0] = m0.plan_json()
            m0.close()
        if dist is not None:
            dist.broadcast_object_list(tuned, src=0)
        plan_text, plan_source = tuned[0], "tuned in this run by rank 0" + (" and broadcast" if world > 1 else "")
        if args.save_plan and rank == 0:
            open(args.save_plan, "w").write(plan_text)
    # lane 0 lives on `ctx`; every further lane is a REPLICA (rten_hip_model_clone) on a context (stream) of its own: own buffers and hipGraphs, lane 0's
    # constants and prepacked weights
    lane_ctx = [ctx] + [(ctx.__class__)(local_rank) for _ in range(lanes - 1)]
    model = load(plan_text)
    models = [model] + [model.clone(c) for c in lane_ctx[1:]]

    # ---- the one collective: the weight arena (shared by every lane), from the rank that loaded the model file for real
    comm_world = 1
    arena_ptr, arena_bytes = model.weight_arena()
    m_l = model
    if world > 1:
        ctx.sync()
        if backend == "nccl":
            uid = [lib.Comm.unique_id(ctx) if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = lib.Comm(ctx, uid[0], world, rank)
            comm.broadcast(arena_ptr, arena_bytes, root=0)  # RCCL over xGMI, once
            ctx.sync()
            comm_world = comm.world_size
            comm.close()
        else:  # several ranks on one GPU / no GPU at all: torch.distributed (gloo) carries the bytes through the host
            host = torch.empty(arena_bytes, dtype=torch.uint8)
            dev = DeviceTensor(ctx, (arena_bytes,), np.uint8, ptr=arena_ptr, keepalive=m_l)
            if rank == 0 and not DRY:
                host.copy_(torch.from_numpy(dev.numpy()))
            dist.broadcast(host, src=0)
            if rank != 0 and not DRY:
                dev.upload(host.numpy())
            comm_world = dist.get_world_size()
        if DRY and rank == 0:
            print(f"[recording] weight arena {arena_bytes} bytes broadcast to {comm_world} ranks", file=sys.stderr)
    # each rank gets its own (independent) synthetic batch, resident in HBM before timing starts (every lane holds a copy: a lane's batch is its own buffer)
    x = np.random.default_rng(1234 + rank).random((BATCH_PER_GPU, 3, 224, 224), dtype=np.float32)
    xts = []
    for m_l, c_l in zip(models, lane_ctx):
        xp_l = m_l.bind_input("x", (BATCH_PER_GPU, 3, 224, 224))
        m_l.prepare()
        xts.append(DeviceTensor(c_l, x.shape, np.float32, ptr=xp_l, keepalive=m_l))
        xts[-1].upload(x)
        c_l.sync()
    xt = xts[0]

    def sync_all():
        for m_l in models:
            m_l.sync()

    def barrier():
        if dist is not None:
            dist.barrier()

    # K steps back to back: the chains are joined ONCE, at the end of the region (the steps are independent batches; model.sync() covers every stream)
    for i in range(args.warmup):
        models[i % lanes].run(join=False)
    sync_all()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        models[i % lanes].run(join=False)  # consecutive batches go to the lanes round robin: step k + 1 overlaps step k
    sync_all()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    per_rank_ms = [round(elapsed / args.steps * 1e3, 4)]
    if dist is not None:
        devname = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed], dtype=torch.float64, device=devname)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 4) for t in allt]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    optr, oshape = model.output(0)
    logits = DeviceTensor(ctx, oshape, np.float32, ptr=optr, keepalive=model).numpy()
    logits_sha = hashlib.sha256(np.ascontiguousarray(logits).tobytes()).hexdigest()[:16]
    lanes_agree = True
    for l, (m_l, c_l) in enumerate(zip(models, lane_ctx)):  # every lane ran the same batch: the same bits, whatever ran beside it
        if l == 0 or l >= args.warmup + args.steps:  # (a lane that was never handed a batch has nothing to compare)
            continue
        op_l, os_l = m_l.output(0)
        lanes_agree = lanes_agree and np.array_equal(DeviceTensor(c_l, os_l, np.float32, ptr=op_l, keepalive=m_l).numpy().view(np.int32), logits.view(np.int32))
    if not lanes_agree:
        logits_sha = "lanes-disagree"  # reported through the collective below: every rank then leaves together
    plan_sha = hashlib.sha256(json.dumps(json.loads(plan_text), sort_keys=True).encode()).hexdigest()[:16] if plan_text else None
    shard_report = [(rank, logits_sha, plan_sha, model.planned_steps)]
    if dist is not None:
        box = [None] * world
        dist.all_gather_object(box, shard_report[0])
        shard_report = sorted(box)
    if any(r[1] == "lanes-disagree" for r in shard_report):
        if rank == 0:
            print("bench.py: the lanes of a rank computed different logits for the same batch: results are void", file=sys.stderr)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 3

    # ---- per-step join (a latency figure: every step waits for all chains) beside the free-running throughput figure above
    def one_batch_alone(m):
        lat = []
        for _ in range(min(args.steps, 20)):
            t1 = time.perf_counter()
            m.run()
            m.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        for _ in range(3):
            m.run()  # (the first back-to-back launches of a graph pay a one-time ~40 ms in the runtime: not part of the figure)
        m.sync()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(min(args.steps, 20)):
            m.run()  # ONE replica, joined on the caller's stream every step, host not blocked: no overlap between steps
        m.sync()
        return float(np.median(lat)), (time.perf_counter() - t1) / min(args.steps, 20) * 1e3
    p50, joined_ms = one_batch_alone(model)
    lanes_plan_alone = None
    one_plan = os.path.join(ROOT, "profiles", "plans", "f32_1chain.json")
    if not int8 and lanes > 1 and chains == 1 and world == 1 and not DRY and plan_path == default_plan and plan_text and os.path.exists(one_plan):
        # The lanes plan is chosen for company (round 6: per layer under co-run): ONE replica running it alone is slower than one replica under the plan a single
        # replica is given (3.4-3.6 vs 2.9 ms).  "One batch on one replica with nothing beside it" -- `ms_per_step_joined_every_step` / `p50_latency_ms`, the
        # figures of rounds 4-5 -- is therefore measured on a one-replica model under ITS plan (f32_1chain.json), loaded here beside the idle lanes; the lanes
        # plan's own figure is kept as `lanes_plan_alone`.
        lanes_plan_alone = {"p50_latency_ms": round(p50, 4)}
        solo = Model(ctx, onnx_bytes, open(one_plan).read(), 1)
        try:
            sp = solo.bind_input("x", (BATCH_PER_GPU, 3, 224, 224))
            solo.prepare()
            DeviceTensor(ctx, x.shape, np.float32, ptr=sp, keepalive=solo).upload(x)
            ctx.sync()
            for _ in range(3):
                solo.run()
            solo.sync()
            p50, joined_ms = one_batch_alone(solo)
            so_ptr, so_shape = solo.output(0)
            if not np.array_equal(DeviceTensor(ctx, so_shape, np.float32, ptr=so_ptr, keepalive=solo).numpy().view(np.int32), logits.view(np.int32)):
                lanes_plan_alone["one_replica_plan_logits"] = "DIFFER"  # (every plan is a choice among bit-identical launch forms)
        finally:
            solo.close()

    # ---- PCIe-inclusive rate (the reference's Model::run takes host tensors): batch uploaded and logits downloaded every step.  Never `value`.
    pcie_ms = None
    if rank == 0:
        t1 = time.perf_counter()
        for _ in range(min(args.steps, 20)):
            xt.upload(x)
            model.run(inputs_written_on_caller_stream=True)
            model.sync()
            DeviceTensor(ctx, oshape, np.float32, ptr=optr, keepalive=model).numpy()
        pcie_ms = (time.perf_counter() - t1) / min(args.steps, 20) * 1e3

    step_ms = elapsed / args.steps * 1e3
    gflop = (resnet50.conv_flops_per_image() + 2 * 2048 * 1000) / 1e9
    roof = None
    if rank == 0:
        rep = model.profile_pass(max(1, min(args.steps, 10)))  # serialised, eager, HIP events per launch: per-kernel detail, outside the timed region
        nrep = max(1, min(args.steps, 10))
        tot_ms = sum(r["ms"] for r in rep)
        if int8:
            conv = [r for r in rep if r["kernel"].startswith("igemm_i8")]
            qlist = (json.loads(plan_text).get("qout", []) + json.loads(plan_text).get("qout2", [])) if plan_text else []  # both forms never write / re-read the edge's f32 tensor
            alg, alg_l = int8_graph_floor_bytes(BATCH_PER_GPU), int8_graph_floor_bytes(BATCH_PER_GPU, qlist)
            step = {"algorithmic_bytes": alg, "achieved": round(alg / (step_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "whole forward pass (the TIMED step) against the HBM floor of the graph as the reference runs it (every conv output an f32 tensor; DESIGN.md section 8)",
                    "as_launched": {"algorithmic_bytes": alg_l, "achieved": round(alg_l / (step_ms * 1e-3) / 1e9, 1), "frac": round(alg_l / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "note": "floor of the launch sequence actually run: quantized-output launches never write / re-read the f32 tensor of a single-consumer edge"},
                    "mfma": {"achieved": round(gflop * BATCH_PER_GPU / step_ms, 2), "peak": I8_MATRIX_PEAK_TOPS, "unit": "TOP/s",
                             "frac": round(gflop * BATCH_PER_GPU / step_ms / I8_MATRIX_PEAK_TOPS, 4)}}
            if conv:
                dom = max(conv, key=lambda r: r["ms"])
                fam_ms, fam_ops = sum(r["ms"] for r in conv), sum(r["flops"] for r in conv)
                gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "traffic": None, "traffic_source": None,
                        "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2), "launches": dom["launches"], "bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
                        "kernel_share_of_step": round(dom["ms"] / max(tot_ms, 1e-9), 4),
                        "mfma": {"achieved": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 2), "peak": I8_MATRIX_PEAK_TOPS, "unit": "TOP/s",
                                 "frac": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / I8_MATRIX_PEAK_TOPS, 4)},
                        "step": step,
                        "igemm_i8_family": {"achieved": round(fam_ops / (fam_ms * 1e-3) / 1e12, 2), "unit": "TOP/s",
                                            "frac_of_i8_mfma_peak": round(fam_ops / (fam_ms * 1e-3) / 1e12 / I8_MATRIX_PEAK_TOPS, 4), "share_of_step": round(fam_ms / max(tot_ms, 1e-9), 4),
                                            "variants": {r["kernel"]: {"launches": r["launches"], "ms": round(r["ms"], 4), "tops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                                                                       "gbs": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in conv}},
                        "other_kernels": {r["kernel"]: {"launches": r["launches"], "ms": round(r["ms"], 4), "gbs": round(r["bytes"] / max(r["ms"] * 1e-3, 1e-12) / 1e9, 1)}
                                          for r in rep if not r["kernel"].startswith("igemm_i8")}}
                attach_traffic(roof, plan_sha, int8=True)
            else:
                roof = {"bound": "hbm", "achieved": step["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step["frac"], "traffic": None, "step": step}
        else:
            conv = [r for r in rep if r["kernel"].startswith("igemm_f32")]
            step_tf = gflop * BATCH_PER_GPU / step_ms
            roof = {"bound": "mfma", "achieved": round(step_tf, 3), "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(step_tf / F32_MATRIX_PEAK_TFLOPS, 4),
                    "what": "2*M*N*K of every convolution / classifier launch of one batch, divided by the TIMED step (all kernels and gaps included; "
                            "the chains overlap in the timed region, so this -- not a serialised per-kernel figure -- is the state the value was measured in)",
                    "traffic": None, "traffic_source": None}
            if conv:
                dom = max(conv, key=lambda r: r["ms"])
                dom_tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
                fam_ms, fam_fl = sum(r["ms"] for r in conv), sum(r["flops"] for r in conv)
                roof["kernel"] = dom["kernel"]
                roof["dominant_kernel"] = {"kernel": dom["kernel"], "achieved": round(dom_tf, 3), "frac": round(dom_tf / F32_MATRIX_PEAK_TFLOPS, 4),
                                           "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2), "launches": dom["launches"],
                                           "flops_per_launch": dom["flops"] / max(dom["launches"], 1), "share_of_serialised_pass": round(dom["ms"] / max(tot_ms, 1e-9), 4),
                                           "note": "stand-alone, serialised launches (HIP events per launch on the chain's stream)"
                                                   + (f" at the sub-batch shapes the {chains} chains launch: such a kernel under-fills the chip on its own, which is what "
                                                      "overlapping the chains is for" if chains > 1 else "")}
                roof["igemm_family"] = {"achieved": round(fam_fl / (fam_ms * 1e-3) / 1e12, 3), "frac": round(fam_fl / (fam_ms * 1e-3) / 1e12 / F32_MATRIX_PEAK_TFLOPS, 4),
                                        "share_of_serialised_pass": round(fam_ms / max(tot_ms, 1e-9), 4),
                                        "variants": {r["kernel"]: {"launches": r["launches"], "ms": round(r["ms"], 4), "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2)} for r in conv}}
            attach_traffic(roof, plan_sha, int8=False)
            if world == 1 and not DRY and not args.no_shapes and lanes > 1 and chains == 1 and plan_text and "dominant_kernel" in roof:
                # the lanes plan is chosen per layer UNDER CO-RUN (tools/tune_corun.py): its launch forms are slower when a launch has the device to itself (which is what
                # the serialised figures above show) and faster where other replicas fill their tile-quantisation gaps.  The per-kernel figure that belongs to the
                # timed schedule: the dominant layer family, the same layer on `lanes` streams at once, time per launch over all streams.
                for m_l in models:
                    m_l.sync()
                try:
                    roof["dominant_kernel"]["co_run"] = dominant_co_run(json.loads(plan_text), lanes, lane_ctx)
                except Exception as e:  # noqa: BLE001  (a measurement aid must not cost the line)
                    roof["dominant_kernel"]["co_run"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            if world == 1 and not DRY and not args.no_shapes:
                p1 = os.path.join(ROOT, "profiles", "plans", "f32_1chain.json")
                model.sync()
                roof["shapes"] = per_shape_table(ctx, json.load(open(p1)) if os.path.exists(p1) else None)
                roof["shapes_note"] = ("every distinct convolution shape at batch 32, stand-alone under profiles/plans/f32_1chain.json: us, TFLOP/s, fraction of the MFMA / HBM peak, "
                                       "tiles per compute unit, and the bound the shape allows (max of its HBM time and its MFMA time over the tile-quantisation efficiency)")
    if rank == 0:
        global_batch = BATCH_PER_GPU * world
        out = {"metric": f"inferences/sec, ResNet-50 {'int8 (dynamically quantized)' if int8 else 'f32'} batch 32 per GPU", "value": round(global_batch * args.steps / elapsed, 2),
               "unit": "inferences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 4),
               "ms_per_step_joined_every_step": round(joined_ms, 4), "p50_latency_ms": round(p50, 4),
               **({"lanes_plan_alone": lanes_plan_alone} if lanes_plan_alone else {}),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 x i8 -> i32 (f32 between layers)" if int8 else "f32",
               "data": "recording (no device: control-flow test)" if DRY else "synthetic",
               "config": {"workload": f"ResNet-50 v1.5 {'dynamically quantized int8 (DynamicQuantizeLinear -> ConvIntegerToFloat per conv, 7-bit per-tensor weights as tools/ort-quantize.py writes them)' if int8 else 'f32'} "
                                      f"inference, 224x224, batch 32 per GPU (BASELINE configs[{2 if int8 else 1}]" + ("; x8 GPUs = configs[4]" if int8 else "") + ") from ONNX bytes "
                                      "(rten_amd.onnx_writer), synthetic He-normal BN-folded weights (seed 1234), inputs U[0,1) resident in HBM",
                          "path": "executor",
                          "path_note": "C++ plan executor behind the C ABI (rten_hip_model_load_ex / _prepare / _run: include/rten_hip_graph.hpp + csrc/graph_abi.cpp) -- "
                                       "the path a Rust host binds (INTEGRATION.md 2.5); `--via-runner` times the hand-planned Python runner instead",
                          "global_batch": global_batch, "parallelism": f"batch-shard x{world} (weight arena RCCL-broadcast once)" if world > 1 else "single GPU",
                          "launch": "hipGraph replay",
                          "batch_lanes": {"lanes": lanes,
                                          "note": "independent replicas of the model (rten_hip_model_clone: own streams, buffers and hipGraphs, ONE shared weight arena); consecutive batches go to "
                                                  "them round robin, so step k + 1 overlaps step k -- a THROUGHPUT schedule for independent batches (what two request "
                                                  "threads on two HipSubgraph instances do); `ms_per_step_joined_every_step` / `p50_latency_ms` are ONE batch on one "
                                                  "replica with nothing beside it; quantized-output launches (which need the device to themselves) are off when lanes > 1"},
                          "batch_chains": {"chains": chains,
                                           "note": "independent sub-batch chains on their own streams, shared weights, logits bit-identical to one chain; in the warm-up and the timed "
                                                   "region the chains free-run across steps and are joined once at the end (`ms_per_step` is a THROUGHPUT figure; "
                                                   "`ms_per_step_joined_every_step` / `p50_latency_ms` join every step)"},
                          "launch_plan": {"source": plan_source, "sha16": plan_sha, "identical_on_all_ranks": len({r[2] for r in shard_report}) == 1,
                                          "steps_planned": model.planned_steps, "steps": model.num_steps, "warning": model.warning or None},
                          "weight_arena_bytes": arena_bytes, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                          ("gop_per_image" if int8 else "gflop_per_image"): round(gflop, 3), "device": ctx.device_info()},
               "ranks": {"world_size": world, "dist_backend": backend, "weight_broadcast_world": comm_world, "ms_per_step_per_rank": per_rank_ms,
                         "logits_sha16_per_rank": [r[1] for r in shard_report], "plan_sha16_per_rank": [r[2] for r in shard_report],
                         "planned_steps_per_rank": [r[3] for r in shard_report], "input_seed_per_rank": [1234 + r[0] for r in shard_report]},
               "pcie_inclusive": {"ms_per_step": round(pcie_ms, 4), "inferences_per_s": round(BATCH_PER_GPU / (pcie_ms * 1e-3), 1),
                                  "note": "rank 0: batch uploaded from pageable host memory and logits downloaded every step (19.3 MB in, 128 KB out); not `value`"} if pcie_ms else None,
               "roofline": roof}
        if int8:
            out["config"]["int8_pad_mode"] = ("RAW0_I8 -- ASSUMPTION: padded taps of an integer convolution hold raw 0 after the u8->i8 shift, the x86 reference's im2col behaviour "
                                              "(rten-gemm/src/im2col.rs:340-358, SURVEY App. C.1); unpinned by a reference-held vector; ZERO_POINT / RAW0_U8 are the other modes of the ABI")
            p = json.loads(plan_text) if plan_text else {}
            out["config"]["quantized_output_launches"] = sorted(p.get("qout", []))
            out["config"]["quantized_output_by_recomputation"] = sorted(p.get("qout2", []))
            out["config"]["quantize_on_load_layers"] = sorted(p.get("fused_dql", []))
        if world == 1 and not args.no_cpu_baseline and not DRY:
            out["cpu_baseline"] = cpu_baseline_int8(resnet50.conv_specs(), weights) if int8 else cpu_baseline(resnet50.conv_specs(), weights)
            if not int8:
                out["cpu_baseline"]["other_cpu_implementation"] = torch_cpu_reference(resnet50.conv_specs(), weights)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_secondary and not int8 and not DRY:
            out["secondary"] = secondary_configs()
            four = out["secondary"].get("resnet50_f32_b32_4chains_1lane") or {}
            if "p50_latency_ms" in four:
                out["p50_latency_ms_4chains"] = four["p50_latency_ms"]
        emit(out, args, key="int8" if int8 else None)
    if DRY:
        import collections
        c = collections.Counter(ctx.log)
        from rten_amd.sharding import shard_range
        print(f"[recording] rank {rank} seed {1234 + rank} shard {list(shard_range(BATCH_PER_GPU * world, rank, world))[:1]}..+{BATCH_PER_GPU} graph_launch {c['graph_launch']} "
              f"load {c['model_load']} load_receive {c['model_load_receive']} prepare {c['model_prepare']} h2d {c['rten_hip_memcpy_h2d']}", file=sys.stderr)
    for m_l in reversed(models):  # replicas before their origin
        m_l.close()
    if dist is not None:
        dist.barrier()  # rank 0's instrumented pass / JSON line happen before any rank tears the group down
        dist.destroy_process_group()
    return 0


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` started without a launcher: re-exec under torch.distributed.run, one rank per GPU (the
    form the driver uses itself), and hand back its exit code.  Rank 0's JSON line goes to our stdout unchanged."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=("f32", "int8"), default="f32",
                    help="f32: ResNet-50 f32 batch 32 per GPU (BASELINE configs[1], the headline); int8: the dynamically quantized graph "
                         "(configs[2]; with --gpus 8 = configs[4], 8 x 32 images, weights RCCL-broadcast)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-autotune", action="store_true", help="neither load a plan nor tune: the backend's built-in launch plans")
    ap.add_argument("--autotune", action="store_true", help="tune the per-layer launch plans in this run (rank 0 tunes, the plan is broadcast) even if a "
                                                              "committed plan for this configuration exists under profiles/plans/")
    ap.add_argument("--tune-placement", action="store_true", help="f32 chains: own 8 streams and search which of them to launch on (not needed with GPU_MAX_HW_QUEUES=8)")
    ap.add_argument("--no-qout", action="store_true", help="int8: do not quantize single-consumer conv outputs in the producing launch (rten_hip_conv2d_int8_qout)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layer-table", action="store_true", help="print the per-layer autotune table to stderr")
    ap.add_argument("--save-plan", default=None, help="write the autotuned per-layer plans (variant, split mode, groups) as JSON")
    ap.add_argument("--load-plan", default=None, help="use per-layer plans from a JSON file instead of autotuning (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of BASELINE configs[2] (int8 ResNet-50) and configs[3] (BERT-base) reported under \"secondary\"")
    ap.add_argument("--chains", type=int, default=None,
                    help="f32: run the batch as this many independent sub-batch chains on their own streams (round 4's schedule; default since round 5: one chain "
                         "per replica, two replicas -- see --lanes). "
                         "The int8 graph quantizes each activation over the whole batch, so it always runs as one chain")
    ap.add_argument("--concurrent", action="store_true", help="run the projection shortcuts on a second stream (parallel graph branches)")
    ap.add_argument("--via-executor", action="store_true", help="(the default since round 5; kept so that older command lines still parse)")
    ap.add_argument("--via-runner", action="store_true",
                    help="time the hand-planned Python runner (rten_amd/workloads/*.py over the per-operator C entry points) instead of the product path -- the C++ "
                         "plan executor behind the C ABI (rten_hip_model_*: ONNX bytes in, committed launch plan, chains, hipGraph replay), which is the default")
    ap.add_argument("--lanes", type=int, default=None,
                    help="executor: run this many independent replicas of the model (own streams, own buffers) and hand consecutive BATCHES to them round robin, so "
                         "that step k + 1 overlaps step k (default: 4 for int8, whose graph cannot be split into sub-batch chains; 3 for f32 with one chain each; "
                         "1 when --chains is given).  A throughput schedule: `p50_latency_ms` stays the latency of ONE batch on one replica")
    ap.add_argument("--no-shapes", action="store_true", help="f32: skip the stand-alone per-shape table (`roofline.shapes`)")
    ap.add_argument("--detail-file", default=None, help="where the full record goes (default gpurun_out/bench_detail[_<config>].json); the printed line stays compact")
    ap.add_argument("--recording-test", action="store_true",
                    help="control-flow test mode (tests/test_bench_world8.py): together with RTEN_BENCH_RECORDING=1, launches are recorded instead of issued")
    return ap.parse_args()


def main():
    args = parse_args()

    # ---- launch contract: N ranks, one per GPU.  Under a launcher (the driver's torch.distributed.run) WORLD_SIZE must
    # equal --gpus; without one, --gpus N > 1 spawns the ranks itself.  A single process never reports N GPUs.
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.config == "int8" and args.chains not in (None, 1):
        print("bench.py: --config int8 runs as one chain (DynamicQuantizeLinear takes min / max over the whole batch: sub-batches would change the codes)", file=sys.stderr)
        return 2
    if args.chains is not None and not 1 <= args.chains <= 8:
        print("bench.py: --chains must be 1..8", file=sys.stderr)
        return 2
    if DRY and not args.recording_test:
        print("bench.py: RTEN_BENCH_RECORDING=1 without --recording-test: the recording mode is a control-flow test (tests/test_bench_world8.py), "
              "never a measurement; refusing", file=sys.stderr)
        return 2
    # int8 --autotune: the per-edge / per-layer choices of the int8 plan (quantized-output edges, quantize-on-load layers) are measured by the Python
    # runner, which writes the plan file the executor reads
    if not (args.via_runner or (args.autotune and args.config == "int8")):
        return run_via_executor(args)
    return run_via_runner(args)


def run_via_runner(args):
    """`--via-runner`: the hand-planned Python runner over the per-operator C entry points (rounds 1-4's headline path; A/B against the executor)."""
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not under_launcher and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a number for GPUs that are not running",
              file=sys.stderr)
        return 2
    n_gpus = args.gpus
    dist = None
    import torch
    backend = "none"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RTEN_DIST_BACKEND=gloo lets the rank != 0 path (arena received by broadcast) be exercised with several ranks on ONE
        # GPU (RCCL refuses two ranks per device); the driver's multi-GPU runs use the default, nccl (= RCCL over xGMI).
        backend = "gloo" if DRY else os.environ.get("RTEN_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        if not DRY:
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif not DRY:
        torch.cuda.set_device(local_rank)

    from rten_amd import lib
    from rten_amd.workloads import resnet50, resnet50_int8
    from rten_amd.sharding import broadcast_weight_arena

    if DRY:
        from rten_amd.recording import RecordingCtx
        ctx = RecordingCtx(local_rank)
        torch.cuda.synchronize = lambda *a, **k: None  # (nothing is ever enqueued on a device in this mode)
    else:
        ctx = lib.Context(local_rank)  # no CPU fallback: raises if the HIP extension / MI355X is missing
    weights = resnet50.make_weights()
    int8 = args.config == "int8"

    chains = 1 if int8 else (4 if args.chains is None else args.chains)

    def build(**kw):
        if int8:
            return resnet50_int8.ResNet50Int8(ctx, BATCH_PER_GPU, weights, **kw)
        if chains > 1:
            return resnet50.ChainedResNet50(ctx, BATCH_PER_GPU, weights, chains=chains, pool=None if args.tune_placement else chains, **kw)
        return resnet50.ResNet50(ctx, BATCH_PER_GPU, weights, **kw)

    comm_world = 1
    if world > 1:
        # The weight arena (prepacked conv weights + biases + classifier; f32: 102 MB) is staged ONCE, by rank 0, and broadcast:
        # through the backend's own communicator (rten_hip_comm_* = RCCL behind the C ABI, what a Rust host would call) on
        # the context's stream.  Under RTEN_DIST_BACKEND=gloo (several ranks on one GPU) torch.distributed carries it.
        nbytes = resnet50_int8.i8_arena_layout(ctx.lib, BATCH_PER_GPU)[1] if int8 else resnet50.arena_bytes(ctx.lib, BATCH_PER_GPU)  # host arithmetic: nothing is built twice
        arena_t = torch.empty(nbytes, dtype=torch.uint8, device="cpu" if DRY else f"cuda:{local_rank}")
        net = build(i8_arena_ptr=arena_t.data_ptr(), i8_arena_keepalive=arena_t) if int8 else build(arena_ptr=arena_t.data_ptr(), arena_keepalive=arena_t)
        if rank == 0:
            net.upload_weights()
        ctx.sync()
        if backend == "nccl":
            uid = [lib.Comm.unique_id(ctx) if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = lib.Comm(ctx, uid[0], world, rank)
            comm.broadcast(arena_t.data_ptr(), nbytes, root=0)  # RCCL over xGMI, once
            ctx.sync()
            comm_world = comm.world_size
            comm.close()
        else:
            broadcast_weight_arena(arena_t, src=0)
            comm_world = dist.get_world_size()
        torch.cuda.synchronize()
        if DRY and rank == 0:  # what the test checks about the one collective: the size every rank allocated, from host arithmetic alone
            print(f"[recording] weight arena {nbytes} bytes broadcast to {comm_world} ranks", file=sys.stderr)
    else:
        net = build()
        net.upload_weights()

    # each rank gets its own (independent) synthetic batch, resident in HBM before timing starts
    x = np.random.default_rng(1234 + rank).random((BATCH_PER_GPU, 3, 224, 224), dtype=np.float32)
    net.x.upload(x)
    ctx.sync()

    table = None
    net.concurrent = args.concurrent

    # ---- the per-layer launch plan.  Default: the plan committed for this configuration (tuned once on an MI355X, profiles/plans/): every
    # rank of every run launches the same kernels, the committed rocprofv3 / PMC passes describe exactly the plan that is timed, and no
    # start-up time goes into tuning.  --autotune (or no committed plan): rank 0 tunes and the plan is broadcast -- ranks never tune on their own.
    def export_plan():
        if int8:
            return {"fused_dql": sorted(net.fused_layers or []), "qout": sorted(set(net.qout_next) - net._qout_off)}
        return net.plan_table() if chains > 1 else {k: list(v) for k, v in net.variants.items()}

    def apply_plan(plan):
        if int8:
            net.fused_dql, net.fused_layers = True, set(plan.get("fused_dql", []))
            if "qout" in plan:
                net._qout_off = set(net.qout_next) - set(plan["qout"])
        else:
            net.variants = plan if chains > 1 else {k: tuple(v) for k, v in plan.items()}

    plan, plan_source = None, "backend defaults (no plan)"
    default_plan = os.path.join(ROOT, "profiles", "plans", "int8.json" if int8 else f"f32_{chains}chain{'s' if chains > 1 else ''}.json")
    if int8:
        # quantized-output launches need every workgroup of a launch resident at once: not when several ranks share ONE GPU (the gloo test
        # mode) -- on a real multi-GPU node every rank has a device to itself
        net.fused_qout = not args.no_qout and (world == 1 or backend == "nccl")
    if args.load_plan:
        plan, plan_source = json.load(open(args.load_plan)), os.path.relpath(os.path.abspath(args.load_plan), ROOT)
    elif not args.autotune and not args.no_autotune and os.path.exists(default_plan):
        plan, plan_source = json.load(open(default_plan)), os.path.relpath(default_plan, ROOT)
    elif not args.no_autotune:
        qtab = None
        if rank == 0:
            table = net.autotune(reps=3)
            if int8 and net.fused_qout:
                qtab = net.autotune_qout()
            plan = export_plan()
        if dist is not None:
            box = [plan]
            dist.broadcast_object_list(box, src=0)
            plan = box[0]
        plan_source = "tuned in this run by rank 0" + (" and broadcast" if world > 1 else "")
        if args.save_plan and rank == 0:
            json.dump(plan, open(args.save_plan, "w"))
        if args.layer_table and rank == 0 and table and int8:
            for name, (sep, fus) in table.items():
                print(f"[layer] {name:8s} DynamicQuantizeLinear staged + conv {sep:6.1f} us | quantize-on-load conv {fus:6.1f} us -> {'fused' if name in net.fused_layers else 'staged'}",
                      file=sys.stderr)
            for name, (us2, us1) in (qtab or {}).items():
                print(f"[layer] {name:8s} conv + consumer's quantize {us2:6.1f} us | one quantized-output launch "
                      + (f"{us1:6.1f} us" if us1 is not None else "  (grid not resident at once)") + f" -> {'one' if name not in net._qout_off else 'two'}", file=sys.stderr)
        if args.layer_table and rank == 0 and table and not int8:
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
    if plan is not None:
        apply_plan(plan)
    import hashlib
    plan_sha = hashlib.sha256(json.dumps(plan, sort_keys=True).encode()).hexdigest()[:16] if plan is not None else None
    placement = None
    if not args.no_graph:
        net.capture()
        if chains > 1 and args.tune_placement:
            placement = net.tune_placement()  # which streams (hardware queues) the chain graphs are launched on

    def barrier():
        if dist is not None:
            dist.barrier()

    # K steps back to back: the sub-batch chains are joined ONCE, at the end of the region (a join per step would make chain 0, which runs on the
    # main context, wait for the slowest chain before its next step: a barrier between steps that the workload does not have -- the steps are
    # independent batches).  Both synchronisation points below cover every stream.
    free_run = hasattr(net, "join")

    def k_steps(k):
        for _ in range(k):
            net.run(join=False) if free_run else net.run()
        if free_run:
            net.join()

    k_steps(args.warmup)
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_steps(args.steps)
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    per_rank_ms = [round(elapsed / args.steps * 1e3, 4)]
    if dist is not None:
        dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 4) for t in allt]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- what every rank computed, and under which plan: rank r's logits are those of ITS 32 inputs (seed 1234 + r) -- the parity
    #      definition of a sharded run (each shard == an independent reference run on that shard) -- and every rank ran the
    #      same launch plan.  tests/test_gpu_multirank.py checks both against the oracle.
    logits_sha = hashlib.sha256(np.ascontiguousarray(net.logits.numpy()).tobytes()).hexdigest()[:16]
    # the plan that actually RAN on this rank (after warm-up an int8 edge whose quantized-output launch did not fit falls back to the two-launch form):
    # hashed per rank, compared below -- a rank that ran other kernels than rank 0 is reported, not assumed away
    effective = export_plan() if plan is not None else None
    effective_sha = hashlib.sha256(json.dumps(effective, sort_keys=True).encode()).hexdigest()[:16] if effective is not None else None
    shard_report = [(rank, logits_sha, effective_sha)]
    if dist is not None:
        box = [None] * world
        dist.all_gather_object(box, shard_report[0])
        shard_report = sorted(box)

    def check_qout(where):
        if int8 and net.qout_timeouts():
            print(f"bench.py: a quantized-output launch gave up waiting for its grid (workgroups not all resident) {where}: results are void", file=sys.stderr)
            sys.exit(3)
    check_qout("during the timed steps")

    # ---- p50 latency per batch (separate pass, host-timed per step)
    lat = []
    for _ in range(min(args.steps, 20)):
        t1 = time.perf_counter()
        net.run()
        ctx.sync()
        lat.append((time.perf_counter() - t1) * 1e3)
    p50 = float(np.median(lat))
    check_qout("during the latency pass")

    # ---- PCIe-inclusive rate (the reference's Model::run takes host tensors): the same K steps with the batch uploaded
    #      from host memory and the logits downloaded every step.  Reported beside `value`, never as `value`.
    pcie_ms = None
    if rank == 0:
        t1 = time.perf_counter()
        for _ in range(min(args.steps, 20)):
            net.x.upload(x)
            net.run()
            net.logits.numpy()
        pcie_ms = (time.perf_counter() - t1) / min(args.steps, 20) * 1e3

    check_qout("during the PCIe-inclusive pass")
    # ---- roofline of the dominant kernel: instrumented eager pass over the same K steps (HIP events per launch
    #      on the backend's stream).  Kept out of the timed region so `value` is not perturbed.
    roof = None
    if rank == 0:
        rep = net.profile_pass(args.steps)  # serialised launches (chain after chain): clean per-kernel durations
        tot_ms = sum(r["ms"] for r in rep)
        if int8:
            conv = [r for r in rep if r["kernel"].startswith("igemm_i8")]
            if conv:
                dom = max(conv, key=lambda r: r["ms"])
                fam_ms, fam_ops = sum(r["ms"] for r in conv), sum(r["flops"] for r in conv)
                alg, alg_l = int8_algorithmic_bytes(net), int8_algorithmic_bytes(net, as_launched=True)
                step_ms = elapsed / args.steps * 1e3
                gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                        "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2), "launches": dom["launches"],
                        "bytes_per_launch": dom["bytes"] / max(dom["launches"], 1),
                        "kernel_share_of_step": round(dom["ms"] / max(tot_ms, 1e-9), 4),
                        "mfma": {"achieved": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12, 2), "peak": I8_MATRIX_PEAK_TOPS, "unit": "TOP/s",
                                 "frac": round(dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / I8_MATRIX_PEAK_TOPS, 4)},
                        "step": {"algorithmic_bytes": alg, "achieved": round(alg / (step_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "whole forward pass against the HBM floor of the graph as the reference runs it (every conv output an f32 tensor; DESIGN.md section 7)",
                                 "as_launched": {"algorithmic_bytes": alg_l, "achieved": round(alg_l / (step_ms * 1e-3) / 1e9, 1),
                                                 "frac": round(alg_l / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "note": "floor of the launch sequence actually run: quantized-output launches never write / re-read the f32 tensor of a single-consumer edge"}},
                        "igemm_i8_family": {"achieved": round(fam_ops / (fam_ms * 1e-3) / 1e12, 2), "unit": "TOP/s",
                                            "frac_of_i8_mfma_peak": round(fam_ops / (fam_ms * 1e-3) / 1e12 / I8_MATRIX_PEAK_TOPS, 4),
                                            "share_of_step": round(fam_ms / max(tot_ms, 1e-9), 4),
                                            "variants": {r["kernel"]: {"launches": r["launches"], "ms": round(r["ms"], 4),
                                                                       "tops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                                                                       "gbs": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in conv}},
                        "other_kernels": {r["kernel"]: {"launches": r["launches"], "ms": round(r["ms"], 4), "gbs": round(r["bytes"] / max(r["ms"] * 1e-3, 1e-12) / 1e9, 1)}
                                          for r in rep if not r["kernel"].startswith("igemm_i8")}}
        else:
            conv = [r for r in rep if r["kernel"].startswith("igemm_f32")]
            if conv:
                dom = max(conv, key=lambda r: r["ms"])
                fam_ms = sum(r["ms"] for r in conv)
                fam_fl = sum(r["flops"] for r in conv)
                step_ms = elapsed / args.steps * 1e3
                step_tf = fam_fl / args.steps / (step_ms * 1e-3) / 1e12
                dom_tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
                # `achieved` / `frac` describe the TIMED state: the conv FLOPs of one batch over the timed step (hipGraph replay, all
                # chains overlapping, every other kernel and every gap included).  The per-kernel figures of the serialised
                # instrumented pass (HIP events per launch, one chain after the other) are detail: `dominant_kernel`, `igemm_family`.
                roof = {"bound": "mfma", "kernel": dom["kernel"],
                        "achieved": round(step_tf, 3), "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(step_tf / F32_MATRIX_PEAK_TFLOPS, 4),
                        "what": "2*M*N*K of every convolution / classifier launch of one batch, divided by the TIMED step (all kernels and gaps included; "
                                "the chains overlap in the timed region, so this -- not a serialised per-kernel figure -- is the state the value was measured in)",
                        "traffic": None, "traffic_source": None,
                        "dominant_kernel": {"kernel": dom["kernel"], "achieved": round(dom_tf, 3), "frac": round(dom_tf / F32_MATRIX_PEAK_TFLOPS, 4),
                                            "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2), "launches": dom["launches"],
                                            "flops_per_launch": dom["flops"] / max(dom["launches"], 1),
                                            "share_of_serialised_pass": round(dom["ms"] / max(tot_ms, 1e-9), 4),
                                            "note": "stand-alone, serialised launches (HIP events per launch on the backend's stream)"
                                                    + (f" at the sub-batch shapes the chains launch ({net.sizes} images): such a kernel under-fills the chip on its "
                                                       f"own, which is what overlapping {chains} chains is for" if chains > 1 else "")},
                        "igemm_family": {"achieved": round(fam_fl / (fam_ms * 1e-3) / 1e12, 3),
                                         "frac": round(fam_fl / (fam_ms * 1e-3) / 1e12 / F32_MATRIX_PEAK_TFLOPS, 4),
                                         "share_of_step": round(fam_ms / max(tot_ms, 1e-9), 4),
                                         "variants": {r["kernel"]: {"launches": r["launches"], "ms": round(r["ms"], 4),
                                                                    "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2)} for r in conv}}}

    if rank == 0 and roof:
        # HBM bytes per launch of the dominant kernel from a SEPARATE PMC pass (FETCH_SIZE x2 + WRITE_SIZE, tools/gpu/traffic.sh:
        # counters cannot be collected inside the timed run).  The figure belongs to the plan that pass ran under, named in
        # `traffic_source` / `traffic_note`; it is null when no committed pass covers this kernel instantiation.
        import glob
        fname = "int8_hbm_traffic_per_kernel.json" if int8 else "hbm_traffic_per_kernel.json"
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", fname)), reverse=True):
            prof = json.load(open(path))
            ks = prof.get("kernels", {})
            name = roof["kernel"].replace(" ", "")
            # the profiler prints every template argument (defaults included): match on the prefix the backend's own label gives
            cands = [k for k in ks if k == name or k.startswith(name[:-1] + ",")]
            if not cands:
                continue
            best = max(cands, key=lambda k: ks[k].get("launches", 0))
            t = ks[best]
            total = t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"]
            same_plan = plan_sha is not None and prof.get("plan_sha16") == plan_sha
            roof["traffic"] = total if same_plan else None  # a figure measured under another launch plan is not this run's traffic
            roof["traffic_kernel"] = best
            roof["traffic_source"] = os.path.relpath(path, ROOT)
            roof["traffic_plan_sha16"] = prof.get("plan_sha16")
            roof["traffic_note"] = ("HBM bytes per launch of the dominant kernel (FETCH_SIZE x 2 + WRITE_SIZE) from a separate rocprofv3 --pmc pass over the SAME launch plan "
                                    "(plan_sha16 matches; counters cannot be collected inside the timed run)" if same_plan else
                                    f"null: the committed PMC pass ran another launch plan (its figure for this kernel: {total} B per launch)")
            break
    if rank == 0:
        global_batch = BATCH_PER_GPU * n_gpus
        value = global_batch * args.steps / elapsed
        if int8:
            metric = "inferences/sec, ResNet-50 int8 (dynamically quantized) batch 32 per GPU"
            workload = ("ResNet-50 v1.5 dynamically quantized int8 inference (DynamicQuantizeLinear -> ConvIntegerToFloat per conv, 7-bit per-tensor weights as "
                        "tools/ort-quantize.py writes them), 224x224, batch 32 per GPU (BASELINE configs[2]; x8 GPUs = configs[4]); synthetic He-normal "
                        "BN-folded weights (seed 1234), inputs U[0,1) resident in HBM")
            flop = "gop_per_image"
        else:
            metric = "inferences/sec, ResNet-50 f32 batch 32 per GPU"
            workload = ("ResNet-50 v1.5 f32 inference, 224x224, batch 32 per GPU (BASELINE configs[1]); "
                        "synthetic He-normal BN-folded weights (seed 1234), inputs U[0,1) resident in HBM")
            flop = "gflop_per_image"
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "inferences/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "p50_latency_ms": round(p50, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 x i8 -> i32 (f32 between layers)" if int8 else "f32",
            "data": "recording (no device: control-flow test)" if DRY else "synthetic",
            "config": {"workload": workload, "path": "runner",
                       "path_note": "hand-planned Python runner over the per-operator C entry points (--via-runner); the default path is the C++ plan executor behind the C ABI",
                       "global_batch": global_batch, "parallelism": f"batch-shard x{n_gpus} (weights RCCL-broadcast once)" if n_gpus > 1 else "single GPU",
                       "launch": "eager" if args.no_graph else "hipGraph replay", "autotuned_tiles": bool(net.variants),
                       "launch_plan": {"source": plan_source, "sha16": plan_sha, "identical_on_all_ranks": len({r[2] for r in shard_report}) == 1},
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "shortcut_branch": "second stream" if net.concurrent else "main stream",
                       "batch_chains": {"chains": chains, "sub_batches": getattr(net, "sizes", [BATCH_PER_GPU]), "placement": getattr(net, "place", [0]),
                                        "placement_ms": [["".join(str(x) for x in pl), round(ms, 3)] for pl, ms in placement] if placement else None,
                                        "note": "independent sub-batch chains on their own streams, shared weights, logits bit-identical to one chain; inside the warm-up and the timed region the chains free-run across steps and are joined once at the end (the synchronisation points cover every stream)"},
                       flop: round((resnet50.conv_flops_per_image() + 2 * 2048 * 1000) / 1e9, 3),
                       "device": ctx.device_info()},
            "ranks": {"world_size": world, "dist_backend": backend, "weight_broadcast_world": comm_world, "ms_per_step_per_rank": per_rank_ms,
                      "logits_sha16_per_rank": [r[1] for r in shard_report], "plan_sha16_per_rank": [r[2] for r in shard_report],
                      "input_seed_per_rank": [1234 + r[0] for r in shard_report]},
            "pcie_inclusive": {"ms_per_step": round(pcie_ms, 4), "inferences_per_s": round(BATCH_PER_GPU / (pcie_ms * 1e-3), 1),
                               "note": "rank 0: batch uploaded from pageable host memory and logits downloaded every step (19.3 MB in, 128 KB out); not `value`"} if pcie_ms else None,
            "roofline": roof,
        }
        if int8:
            out["config"]["quantize_on_load_layers"] = sorted(net.fused_layers) if getattr(net, "fused_layers", None) else []
            out["config"]["quantized_output_launches"] = sorted(set(net.qout_next) - net._qout_off) if net.fused_qout else []
            out["config"]["int8_pad_mode"] = ("RAW0_I8 -- ASSUMPTION: padded taps of an integer convolution hold raw 0 after the u8->i8 shift, the x86 reference's im2col behaviour "
                                              "(rten-gemm/src/im2col.rs:340-358, SURVEY App. C.1); unpinned by a reference-held vector; ZERO_POINT / RAW0_U8 are the other modes of the ABI")
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_int8(net.specs, weights) if int8 else cpu_baseline(net.specs, weights)
            if not int8:
                out["cpu_baseline"]["other_cpu_implementation"] = torch_cpu_reference(net.specs, weights)
        else:
            out["cpu_baseline"] = None
        if n_gpus == 1 and world == 1 and not args.no_secondary and not int8:
            out["secondary"] = secondary_configs()
        emit(out, args, key=args.config + "_runner")
    if DRY:
        import collections
        c = collections.Counter(ctx.log)
        print(f"[recording] rank {rank} seed {1234 + rank} shard {list(__import__('rten_amd.sharding', fromlist=['shard_range']).shard_range(BATCH_PER_GPU * world, rank, world))[:1]}"
              f"..+{BATCH_PER_GPU} graph_launch {c['graph_launch']} qout {c['rten_hip_conv2d_int8_qout']} conv {c['rten_hip_conv2d_int8_stats']} "
              f"dql_loader {c['rten_hip_conv2d_int8_dql']} h2d {c['rten_hip_memcpy_h2d']}", file=sys.stderr)
    if dist is not None:
        dist.barrier()  # rank 0's instrumented pass / JSON line happen before any rank tears the group down
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
