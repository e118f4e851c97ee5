"""ResNet-50 v1.5 (timm/torchvision layout, 224x224) as the RTen executor sees it, on the HIP backend.

Graph: BatchNorm is pre-folded into conv weight/bias by the exporter, so the
reference graph is Conv(+bias) -> Relu ... Conv(+bias) -> Add(residual) -> Relu, MaxPool, GlobalAveragePool,
Flatten, Gemm(transB=1).  The device graph fuses Relu / residual Add into the conv epilogue
(rten_hip_conv2d_f32 flags) -- same operations, same order, same rounding.

There is no network in the build image, so weights are synthetic: He-normal, seed 1234, BN folded
(`make_weights`).  The same spec + weights drive the CPU oracle in tests/ for parity.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import lib as L
from ..tensor import DeviceTensor

STAGES = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]  # (bottleneck width, blocks, first stride)


def conv_specs():
    """Ordered list of conv layers: dict(name, cin, cout, k, stride, pad, relu, residual_from, input_from).
    Activations are named; 'x' is the network input."""
    layers = [dict(name="stem", cin=3, cout=64, k=7, stride=2, pad=3, relu=True, src="x", res=None, dst="stem")]
    cur, cin = "pool", 64
    for si, (width, blocks, stride) in enumerate(STAGES):
        for bi in range(blocks):
            s = stride if bi == 0 else 1
            pre = f"s{si}b{bi}"
            cout = width * 4
            # the projection shortcut comes first: it only depends on the block input, so the runner can put it on a
            # second stream next to c1 -> c2 (its output stays live until c3 either way)
            if bi == 0:
                layers.append(dict(name=pre + "ds", cin=cin, cout=cout, k=1, stride=s, pad=0, relu=False, src=cur, res=None, dst=pre + "id"))
                ident = pre + "id"
            else:
                ident = cur
            layers.append(dict(name=pre + "c1", cin=cin, cout=width, k=1, stride=1, pad=0, relu=True, src=cur, res=None, dst=pre + "t1"))
            layers.append(dict(name=pre + "c2", cin=width, cout=width, k=3, stride=s, pad=1, relu=True, src=pre + "t1", res=None, dst=pre + "t2"))
            layers.append(dict(name=pre + "c3", cin=width, cout=cout, k=1, stride=1, pad=0, relu=True, src=pre + "t2", res=ident, dst=pre + "out"))
            cur, cin = pre + "out", cout
    return layers


def make_weights(seed=1234, num_classes=1000):
    """Synthetic BN-folded weights.  Returns {name: (W [O,C,k,k] f32, bias [O] f32)} + 'fc': (W [1000,2048], b)."""
    rng = np.random.default_rng(seed)
    w = {}
    for l in conv_specs():
        fan_in = l["cin"] * l["k"] * l["k"]
        std = np.sqrt(2.0 / fan_in)
        if l["name"].endswith("c3"):
            std *= 0.5  # keep the residual stream from growing without BN statistics
        w[l["name"]] = (rng.normal(0.0, std, (l["cout"], l["cin"], l["k"], l["k"])).astype(np.float32),
                        rng.normal(0.0, 0.05, (l["cout"],)).astype(np.float32))
    w["fc"] = (rng.normal(0.0, np.sqrt(1.0 / 2048), (num_classes, 2048)).astype(np.float32),
               rng.normal(0.0, 0.05, (num_classes,)).astype(np.float32))
    return w


def conv_flops_per_image():
    total, hw = 0.0, 224
    sizes = {"x": 224}
    for l in conv_specs():
        h = sizes.get(l["src"], None)
        if h is None:
            h = sizes["pool"] if l["src"] == "pool" else hw
        oh = (h + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        sizes[l["dst"]] = oh
        if l["dst"] == "stem":
            sizes["pool"] = (oh + 2 - 3) // 2 + 1
        total += 2.0 * l["cout"] * l["cin"] * l["k"] * l["k"] * oh * oh
    return total


def layer_geometry(batch, image=224):
    """(shapes {tensor: (n, c, h, w)}, descs {layer: Conv2dDesc}) of the static plan for one batch size."""
    shapes = {"x": (batch, 3, image, image)}
    descs = {}
    for l in conv_specs():
        n, c, h, w = shapes["pool" if l["src"] == "pool" else l["src"]] if l["src"] != "x" else shapes["x"]
        oh = (h + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        ow = (w + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        descs[l["name"]] = L.Conv2dDesc(n, c, h, w, l["cout"], l["k"], l["k"], (C.c_int32 * 4)(l["pad"], l["pad"], l["pad"], l["pad"]),
                                        l["stride"], l["stride"], 1, 1, 1, oh, ow)
        shapes[l["dst"]] = (n, l["cout"], oh, ow)
        if l["dst"] == "stem":
            ph = (oh + 2 - 3) // 2 + 1
            shapes["pool"] = (n, l["cout"], ph, ph)
    return shapes, descs


def arena_layout(lib, descs, num_classes=1000):
    """({name: (weights offset, bias offset)}, total bytes) of the f32 weight arena: [packed conv weights | bias] per layer, then the
    classifier.  Host arithmetic only (rten_hip_conv2d_f32_packed_bytes touches no device): a rank that merely RECEIVES the arena
    sizes its buffer with this, without building a network first."""
    offs, total = {}, 0
    for l in conv_specs():
        nb = lib.rten_hip_conv2d_f32_packed_bytes(C.byref(descs[l["name"]]))
        offs[l["name"]] = (total, total + nb)
        total += nb + l["cout"] * 4
        total = (total + 255) & ~255
    fc_w = num_classes * 2048 * 4
    offs["fc"] = (total, total + fc_w)
    total += fc_w + num_classes * 4
    total = (total + 255) & ~255
    return offs, total


def arena_bytes(lib, batch=32, image=224, num_classes=1000):
    return arena_layout(lib, layer_geometry(batch, image)[1], num_classes)[1]


class ResNet50:
    """Device-resident ResNet-50 forward for a fixed batch size (static plan, optional hipGraph)."""

    def __init__(self, ctx, batch, weights=None, image=224, num_classes=1000, arena_ptr=None, arena_keepalive=None, x_view=None, logits_view=None,
                 total_batch=None):
        self.ctx, self.batch, self.image, self.num_classes = ctx, batch, image, num_classes
        self.total_batch = batch if total_batch is None else total_batch  # rows of the classifier product the reference would see (sub-batch chains)
        self._x_view, self._logits_view = x_view, logits_view  # (ptr, keepalive): this net works on a slice of a larger batch
        self.weights = weights if weights is not None else make_weights(num_classes=num_classes)
        self.specs = conv_specs()
        self.graph = None
        self.variants = {}
        self.side = None          # second context (stream) for the projection shortcuts
        self.concurrent = False   # True: run ds convs next to c1 -> c2 (measured: no gain at batch 32, kernels already fill the chip)
        self._plan(arena_ptr, arena_keepalive)

    # ---- static plan: shapes, weight arena, activation buffers, launch list
    def _plan(self, arena_ptr, arena_keepalive):
        ctx, N = self.ctx, self.batch
        shapes, descs = layer_geometry(N, self.image)
        self.shapes, self.descs = shapes, descs
        # weight arena: [packed conv weights | biases | fc W | fc b], one allocation so it can be broadcast
        offs, total = arena_layout(ctx.lib, descs, self.num_classes)
        self.arena_bytes = total
        self.arena = DeviceTensor(ctx, (total,), np.uint8, ptr=arena_ptr, keepalive=arena_keepalive)
        self.w_off = offs
        # activation buffers with liveness-based reuse: op i's output is taken from the free list, then the
        # inputs whose last consumer is op i are released (so an output never aliases a live input).
        ops = [(["x"], "stem"), (["stem"], "pool")]
        ds_src = {}
        for l in self.specs[1:]:
            ins = [l["src"]] + ([l["res"]] if l["res"] else [])
            if l["name"].endswith("ds"):
                ds_src[l["dst"]] = l["src"]
            if l["res"] in ds_src:
                ins.append(ds_src[l["res"]])  # the side stream may still be reading the block input until the join at c3
            ops.append((ins, l["dst"]))
        last_use = {}
        for i, (ins, _) in enumerate(ops):
            for nm in ins:
                last_use[nm] = i
        last_use[ops[-1][1]] = len(ops)  # final activation is read by GlobalAveragePool
        free, self.bufs = [], {}
        xv = self._x_view or (None, None)
        self.x = DeviceTensor(ctx, shapes["x"], np.float32, ptr=xv[0], keepalive=xv[1])
        for i, (ins, out) in enumerate(ops):
            need = int(np.prod(shapes[out])) * 4
            best = None
            for bfr in free:
                if bfr.nbytes >= need and (best is None or bfr.nbytes < best.nbytes):
                    best = bfr
            if best is not None:
                free.remove(best)
            else:
                best = DeviceTensor(ctx, (need // 4,), np.float32)
            self.bufs[out] = best.view(shapes[out])
            self._raw = getattr(self, "_raw", {})
            self._raw[out] = best
            for nm in ins:
                if nm != "x" and last_use[nm] == i:
                    free.append(self._raw[nm])
        self.gap = DeviceTensor(ctx, (N, 2048), np.float32)
        lv = self._logits_view or (None, None)
        self.logits = DeviceTensor(ctx, (N, self.num_classes), np.float32, ptr=lv[0], keepalive=lv[1])
        p = self.shapes["stem"]
        self.pool_desc = L.Pool2dDesc(p[0], p[1], p[2], p[3], 3, 3, 2, 2, (C.c_int32 * 4)(1, 1, 1, 1), self.shapes["pool"][2],
                                      self.shapes["pool"][3], 0)
        # Gemm(transB = 1, C = bias): the reference expands C into the output and runs gemm with beta = 1 (matmul.rs:63-82).  With
        # several rows that equals a per-column bias after the first depth block (what the fused form below does); a ONE-row product
        # takes the reference's gemv kernels, where the bias enters with the first depth block and not at the end, so batch 1 runs
        # the operator's own form: logits <- bias, then gemm(beta = 1).
        # (decided from the WHOLE batch: a one-image chain of a larger batch must not switch the classifier to the gemv order)
        self.fc_gemm_form = self.total_batch == 1
        self.fc_desc = (L.gemm_desc(N, self.num_classes, 2048, 2048, 1, 1, 2048, self.num_classes, beta=1.0) if self.fc_gemm_form else
                        L.gemm_desc(N, self.num_classes, 2048, 2048, 1, 1, 2048, self.num_classes, bias_kind=L.BIAS_PER_COL))

    def _wptr(self, name, which):
        a, b = self.w_off[name]
        return C.c_void_p(self.arena.ptr + (a if which == 0 else b))

    def upload_weights(self):
        """Stage all weights into the arena (rank 0 of a multi-GPU job; others receive the broadcast)."""
        ctx = self.ctx
        for l in self.specs:
            w, b = self.weights[l["name"]]
            tmp = DeviceTensor.from_numpy(ctx, w)
            ctx.call("rten_hip_conv2d_f32_prepack", C.byref(self.descs[l["name"]]), tmp.vp, self._wptr(l["name"], 0))
            ctx.call("rten_hip_memcpy_h2d", self._wptr(l["name"], 1), b.ctypes.data_as(C.c_void_p), C.c_size_t(b.nbytes))
            ctx.sync()
            tmp.free()
        fc_w, fc_b = self.weights["fc"]
        ctx.call("rten_hip_memcpy_h2d", self._wptr("fc", 0), fc_w.ctypes.data_as(C.c_void_p), C.c_size_t(fc_w.nbytes))
        ctx.call("rten_hip_memcpy_h2d", self._wptr("fc", 1), fc_b.ctypes.data_as(C.c_void_p), C.c_size_t(fc_b.nbytes))

    def _act(self, name):
        return self.x if name == "x" else self.bufs[name]

    def _conv(self, l, ctx=None):
        ctx = ctx or self.ctx
        plan = self.variants.get(l["name"])
        if plan is not None:
            v, mode, groups, order = (tuple(plan) + (0,))[:4] if isinstance(plan, (tuple, list)) else (plan, 0, 1, 0)
            ctx.set_gemm_variant(v)
            ctx.call("rten_hip_set_gemm_split", mode, groups)
            ctx.call("rten_hip_set_gemm_order", order)
        flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
        ctx.call("rten_hip_conv2d_f32", C.byref(self.descs[l["name"]]), self._act(l["src"]).vp, self._wptr(l["name"], 0), 1,
                 self._wptr(l["name"], 1), self._act(l["res"]).vp if l["res"] else None, flags, self._act(l["dst"]).vp)
        if plan is not None:
            ctx.set_gemm_variant(-1)
            ctx.call("rten_hip_set_gemm_split", 3, 1)  # back to the automatic plan
            ctx.call("rten_hip_set_gemm_order", 0)

    def forward(self, upto=None):
        """Enqueue one forward pass over self.x -> self.logits (asynchronous).  With `concurrent`, each projection
        shortcut runs on a second context (stream) next to c1 -> c2 and is joined before c3 reads it.
        `upto` = n: only the first n conv layers (a partial pass whose results the next full pass overwrites; used to put the
        chains of a ChainedResNet50 out of phase)."""
        ctx = self.ctx
        if self.concurrent and self.side is None:
            self.side = L.Context(ctx.device)
        self._conv(self.specs[0])
        ctx.call("rten_hip_max_pool2d_f32", C.byref(self.pool_desc), self.bufs["stem"].vp, self.bufs["pool"].vp)
        pending = set()  # shortcut outputs produced on the side stream and not yet joined
        for l in (self.specs[1:] if upto is None else self.specs[1:max(upto, 1)]):
            if self.concurrent and l["name"].endswith("ds"):
                self.side.wait(ctx)           # block input is ready
                self._conv(l, self.side)
                pending.add(l["dst"])
                continue
            if l["res"] in pending:
                ctx.wait(self.side)
                pending.discard(l["res"])
            self._conv(l)
        if upto is not None:
            if pending:
                ctx.wait(self.side)
            return
        last = self.specs[-1]["dst"]
        n, c, h, w = self.shapes[last]
        ctx.call("rten_hip_global_average_pool_f32", n * c, h * w, self.bufs[last].vp, self.gap.vp)
        if self.fc_gemm_form:
            ctx.call("rten_hip_memcpy_d2d", self.logits.vp, self._wptr("fc", 1), C.c_size_t(self.num_classes * 4))
            ctx.call("rten_hip_gemm_f32", C.byref(self.fc_desc), self.gap.vp, self._wptr("fc", 0), None, self.logits.vp)
        else:
            lone_row = self.batch == 1  # a one-image chain of a larger batch: the reference's product has several rows -> the blocked order
            if lone_row:
                ctx.call("rten_hip_set_gemv_order", 0, 0)
            ctx.call("rten_hip_gemm_f32", C.byref(self.fc_desc), self.gap.vp, self._wptr("fc", 0), self._wptr("fc", 1), self.logits.vp)
            if lone_row:
                ctx.call("rten_hip_set_gemv_order", 1, 0)

    def capture(self):
        """Capture the forward pass into a hipGraph (one host call per inference afterwards)."""
        self.forward()  # warm-up: scratch allocations, code objects
        self.ctx.sync()
        self.ctx.graph_begin()
        self.forward()
        self.graph = self.ctx.graph_end()
        return self.graph

    def run(self):
        if self.graph:
            self.ctx.graph_launch(self.graph)
        else:
            self.forward()

    def profile_pass(self, steps):
        """Instrumented eager pass (HIP events per launch, serialised launches): [{kernel, launches, ms, flops, bytes}]."""
        ctx = self.ctx
        ctx.profile_reset()
        ctx.profile(True)
        saved_graph, self.graph = self.graph, None
        saved_conc, self.concurrent = self.concurrent, False
        for _ in range(steps):
            self.forward()
        ctx.sync()
        ctx.profile(False)
        self.graph, self.concurrent = saved_graph, saved_conc
        return ctx.profile_report()

    def candidate_plans(self, l):
        """(variant, split mode, K groups, tile order) plans worth timing for one conv layer.  Split-K plans exist for
        the LDS-DMA variants (0..3) when K spans more than one depth block of 256."""
        nvar = self.ctx.lib.rten_hip_num_gemm_variants()
        plans = [(v, 0, 1, o) for v in range(nvar) for o in (0, 1)]
        d = self.descs[l["name"]]
        nblk = (d.c // d.groups * d.kh * d.kw + 255) // 256
        dma = [v for v in range(nvar) if not 4 <= v < 12]  # LDS-DMA pipelines (3 / 4 stages, fragments-first, 16x16x4 MFMAs, one wave per tile)
        # thin-tile tail (mode 4): whole rounds with this tile shape + the remaining columns as 16x64 tiles on 16x16x4 MFMAs
        plans += [(v, 4, 1, o) for v in dma for o in (0, 1)]
        # persistent plan (mode 5): num_cus x groups workgroups walk the tile list, tile DMA runs across tile boundaries
        plans += [(v, 5, r, o) for v in (0, 1, 2, 3, 20, 21, 22, 23) if v < nvar for r in (1, 2, 3, 4) for o in (0, 1)]
        # lean persistent kernel (mode 6; 64x64 tiles, K % 32 == 0): groups = workgroups per compute unit
        plans += [(3, 6, r, o) for r in (1, 2, 3) for o in (0, 1)]
        if nblk > 1:
            for v in [v for v in range(nvar) if not (8 <= v < 12 or 20 <= v < 24 or v >= 28)]:  # the wave-specialised and 16x16x4 kernels have no split form
                for groups in sorted({2, 3, 4, 6, nblk} & set(range(2, nblk + 1))):
                    plans.append((v, 1, groups, 0))
                    for o in (0, 2, 3):
                        plans.append((v, 2, groups, o))
        return plans

    def autotune(self, reps=3):
        """Pick the fastest (tile variant, split-K plan) per conv layer by measurement (load-time, like the
        reference picks kernels per ISA at start-up, rten-gemm/src/lib.rs:534-547).  Returns
        {layer: [(plan, ms), ...]}."""
        ctx = self.ctx
        table = {}
        for l in self.specs:
            best, best_ms, row = None, 1e30, []
            for plan in self.candidate_plans(l):
                self.variants[l["name"]] = plan
                self._conv(l)  # warm (also grows the split-K slab scratch before any graph capture)
                ms = 1e30
                for _ in range(2):  # best of two short runs: single runs differ by a few percent (clock / neighbours)
                    ctx.timer_start(1)
                    for _ in range(reps):
                        self._conv(l)
                    ctx.timer_stop(1)
                    ms = min(ms, ctx.timer_ms(1) / reps)
                row.append((plan, ms))
                if ms < best_ms:
                    best, best_ms = plan, ms
            self.variants[l["name"]] = best
            table[l["name"]] = row
        return table


def split_batch(batch, chains):
    """Sub-batch sizes and first-image indices of `chains` chains over a batch: sizes differ by at most one image, larger first."""
    if not 1 <= chains <= batch:
        raise ValueError(f"cannot run a batch of {batch} as {chains} chains")
    sub = batch // chains
    sizes = [sub + (1 if i < batch - sub * chains else 0) for i in range(chains)]
    return sizes, [sum(sizes[:i]) for i in range(chains)]


class ChainedResNet50:
    """The batch as `chains` independent sub-batch chains, each on its own stream with its own activations, plan and hipGraph;
    the weight arena is shared.

    Why: a conv kernel whose tile count is not a multiple of the CU count leaves most of the chip idle during its last partial
    round (tools/probe_quantization.py: 6.125 rounds cost 7), and inside ONE chain nothing can use those CUs because layer i+1
    depends on layer i.  Sub-batch chains are mutually independent (every output column of a convolution depends on its own
    image only, so the logits are bit-identical to the single-chain result), and the hardware schedules their kernels side by
    side: measured 3.07 -> 2.79 ms per batch of 32 with 4 chains (tools/probe_two_chains.py; 2 chains 2.90 ms, 8 chains slower --
    tiles of 4-image layers are mostly padding, and 8 co-resident kernels thrash the LDS / L2).

    Streams map onto a handful of hardware queues; two chains that land on the same queue serialise (measured 3.45 ms).  The
    runner therefore owns a pool of contexts and `tune_placement()` picks, by measurement, which of them the chain graphs are
    launched on (a hipGraph captured on one stream may be launched on another)."""

    POOL = 8

    def __init__(self, ctx, batch, weights=None, chains=4, image=224, num_classes=1000, arena_ptr=None, arena_keepalive=None, pool=None):
        """`pool` = number of contexts (streams) the runner owns: `chains` when every stream has a hardware queue of its own
        (GPU_MAX_HW_QUEUES >= chains, set by bench.py before the runtime starts: no placement search needed), POOL (the default) when
        tune_placement() is to search for a collision-free set."""
        self.POOL = self.POOL if pool is None else max(int(pool), chains)
        assert 1 <= chains <= min(batch, self.POOL)
        self.ctx, self.batch, self.chains = ctx, batch, chains
        self.weights = weights if weights is not None else make_weights(num_classes=num_classes)
        self.sizes, self.starts = split_batch(batch, chains)
        self.pool = [ctx] + [type(ctx)(ctx.device) for _ in range(self.POOL - 1)]  # (same kind as the caller's: a recording context spawns recording contexts)
        self.place = list(range(chains))  # chain i's graph is launched on pool[place[i]]
        self.x = DeviceTensor(ctx, (batch, 3, image, image), np.float32)
        self.logits = DeviceTensor(ctx, (batch, num_classes), np.float32)
        self.nets = []
        for i in range(chains):
            a = (arena_ptr, arena_keepalive) if i == 0 else (self.nets[0].arena.ptr, self.nets[0].arena)
            self.nets.append(ResNet50(self.pool[i], self.sizes[i], self.weights, image, num_classes, arena_ptr=a[0], arena_keepalive=a[1],
                                      x_view=(self.x.ptr + self.starts[i] * 3 * image * image * 4, self.x),
                                      logits_view=(self.logits.ptr + self.starts[i] * num_classes * 4, self.logits), total_batch=batch))
        n0 = self.nets[0]
        self.specs, self.descs, self.arena, self.arena_bytes = n0.specs, n0.descs, n0.arena, n0.arena_bytes
        self.graph = None      # list of per-chain graphs once captured
        self.concurrent = False
        self.cotune = {}

    @property
    def variants(self):
        return self.nets[0].variants

    @variants.setter
    def variants(self, v):
        """One plan table for every chain, or a table per sub-batch size ({"8": {layer: plan}, "7": {...}}) when the sizes differ."""
        keyed = bool(v) and all(isinstance(x, dict) for x in v.values())
        for net in self.nets:
            net.variants = {k: tuple(p) for k, p in (v[str(net.batch)] if keyed else v).items()}

    def upload_weights(self):
        self.nets[0].upload_weights()
        self.ctx.sync()

    def sync(self):
        for c in self.pool:
            c.sync()

    def forward(self):
        """Eager pass: every chain enqueues its layers on its own stream; the main context then waits for all of them."""
        for net in self.nets:
            net.forward()
        for net in self.nets:
            if net.ctx is not self.ctx:
                self.ctx.wait(net.ctx)

    def capture(self):
        for net in self.nets:
            net.capture()
        self.sync()
        self.graph = [net.graph for net in self.nets]
        return self.graph

    def run(self, join=True):
        """One step: every chain's graph on its own stream.  `join=True` orders the main context behind all chains (the logits are read there).
        `join=False` leaves that out: chain 0 runs ON the main context, so a per-step join makes it wait for the slowest chain before its next
        step -- a barrier between steps.  A caller that enqueues K steps back to back (bench.py's timed region) joins once, with join()."""
        if not self.graph:
            return self.forward()
        used = [self.pool[p] for p in self.place]
        # no wait on the main context BEFORE the launches: uploads through the C ABI are host-synchronous
        for g, c in zip(self.graph, used):
            c.graph_launch(g)
        if join:
            self.join()

    def join(self):
        """Order the main context behind everything enqueued on the chains so far."""
        for c in (self.pool[p] for p in self.place):
            if c is not self.ctx:
                self.ctx.wait(c)

    def _time_steps(self, steps):
        import time
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.run()
        self.sync()
        return (time.perf_counter() - t0) / steps

    def tune_placement(self, steps=8):
        """Choose the contexts (streams -> hardware queues) the chain graphs are launched on.  Every window of `chains`
        consecutive pool members is timed over a few steps; then one pass of local search swaps each member for each unused
        context and keeps a swap that is at least 1 % faster (the stream -> queue mapping is the runtime's business and differs
        from process to process: no window need be collision-free).  Returns [(placement, ms per step)] of everything tried."""
        assert self.graph
        rows = []
        if self.POOL == self.chains:  # nothing to choose from
            return rows

        def measure(place):
            self.place = list(place)
            self._time_steps(2)
            ms = min(self._time_steps(steps) for _ in range(2)) * 1e3
            rows.append((list(place), ms))
            return ms
        best, best_ms = None, 1e30
        for first in range(self.POOL - self.chains + 1):
            place = list(range(first, first + self.chains))
            ms = measure(place)
            if ms < best_ms:
                best, best_ms = place, ms
        for i in range(self.chains):
            for u in range(self.POOL):
                if u in best:
                    continue
                cand = list(best)
                cand[i] = u
                ms = measure(cand)
                if ms < best_ms * 0.99:
                    best, best_ms = cand, ms
        self.place = best
        return rows

    def autotune(self, reps=3, top=6, corun_reps=6):
        """Per layer: (1) every candidate plan timed alone at each distinct sub-batch size (ResNet50.autotune); (2) the `top`
        fastest of them timed again with ALL chains running that layer at the same time (one small hipGraph of `corun_reps`
        launches per chain) -- next to other kernels the cheapest plan in CU-time wins, which is not always the one with the
        lowest latency alone (split-K plans trade extra work for latency).  Returns the stand-alone table of chain 0."""
        import time
        tables = {}
        for net in self.nets:
            if net.batch not in tables:
                tables[net.batch] = net.autotune(reps)
                tables[(net.batch, "plan")] = dict(net.variants)
            net.variants = dict(tables[(net.batch, "plan")])
        if self.chains > 1 and top > 1:
            for l in self.specs:
                name = l["name"]
                cands = {}
                for b in {net.batch for net in self.nets}:
                    cands[b] = [p for p, _ in sorted(tables[b][name], key=lambda r: r[1])[:top]]
                best, rows = None, []
                for k in range(top):
                    graphs = []
                    for net in self.nets:
                        net.variants[name] = cands[net.batch][min(k, len(cands[net.batch]) - 1)]
                        net._conv(l)  # warm: scratch for this plan exists before the capture
                        net.ctx.sync()
                        net.ctx.graph_begin()
                        for _ in range(corun_reps):
                            net._conv(l)
                        graphs.append(net.ctx.graph_end())
                    ms = 1e30
                    for _ in range(3):
                        self.sync()
                        t0 = time.perf_counter()
                        for net, g in zip(self.nets, graphs):
                            net.ctx.graph_launch(g)
                        for net in self.nets:
                            net.ctx.sync()
                        ms = min(ms, (time.perf_counter() - t0) * 1e3 / corun_reps)
                    for net, g in zip(self.nets, graphs):
                        net.ctx.graph_destroy(g)
                    rows.append((k, ms))
                    if best is None or ms < best[1]:
                        best = (k, ms)
                for net in self.nets:
                    net.variants[name] = cands[net.batch][min(best[0], len(cands[net.batch]) - 1)]
                self.cotune[name] = rows
        return tables[self.nets[0].batch]

    def plan_table(self):
        """{sub-batch size: {layer: plan}} for --save-plan."""
        return {str(net.batch): {k: list(v) for k, v in net.variants.items()} for net in self.nets}

    def profile_pass(self, steps):
        """Chain after chain, serialised (clean per-kernel durations at the sub-batch shapes actually launched); merged."""
        merged = {}
        for net in self.nets:
            self.sync()
            for r in net.profile_pass(steps):
                m = merged.setdefault(r["kernel"], dict(r, launches=0, ms=0.0, flops=0.0, bytes=0.0))
                for k in ("launches", "ms", "flops", "bytes"):
                    m[k] += r[k]
        return list(merged.values())
