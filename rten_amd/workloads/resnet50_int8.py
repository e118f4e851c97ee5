"""ResNet-50 as a dynamically quantized int8 graph (BASELINE configs[2]: ort-quantized ResNet-50) on the HIP backend.

Every Conv node of the f32 graph becomes the chain the reference executes for ort-quantized models
(DynamicQuantizeLinear -> ConvInteger -> Cast -> Mul -> Add [-> Add] [-> Relu]; fused by the reference into
ConvIntegerToFloat, src/ops/conv.rs:495-587): here one quantize kernel pair, one scalar Mul and one int8 conv launch with
the cast_scale / bias / residual / Relu epilogue.  Weights are staged once (rten_hip_conv2d_int8_prepack).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .. import lib as L
from ..tensor import DeviceTensor
from .resnet50 import ResNet50


def quantize_weights(weights):
    """Symmetric per-tensor i8 weights: {name: (wq i8, w_scale f32, bias f32)} (same recipe as oracle.models)."""
    out = {}
    for name, (w, b) in weights.items():
        s = np.float32(np.abs(w).max() / 64.0)  # reduce_range=True: 7-bit weights (tools/ort-quantize.py:124-137)
        out[name] = (np.clip(np.rint(w / s), -64, 64).astype(np.int8), s, b)
    return out


def i8_arena_layout(lib, batch, pad_mode=L.PAD_RAW0_I8, image=224, num_classes=1000):
    """({name: (staged weights, w_scale, bias) offsets}, total bytes, fc packed bytes) of the int8 weight arena -- host arithmetic only
    (the *_packed_bytes functions touch no device), so a rank that receives the arena by broadcast sizes its buffer without building a net."""
    from .resnet50 import conv_specs, layer_geometry
    _, descs = layer_geometry(batch, image)
    offs, total = {}, 0
    for l in conv_specs():
        nb = lib.rten_hip_conv2d_int8_packed_bytes(C.byref(L.Conv2dInt8Desc(descs[l["name"]], 0, 1, 0, pad_mode, 1, 1)))
        if nb == 0:
            raise RuntimeError("int8 conv geometry not covered by the staged kernel: " + l["name"])
        offs[l["name"]] = (total, total + nb, total + nb + 256)
        total = (total + nb + 256 + l["cout"] * 4 + 255) & ~255
    fc_packed = lib.rten_hip_gemm_int8_packed_bytes(2048, num_classes)
    fc_nb = fc_packed or 2048 * num_classes
    offs["fc"] = (total, total + fc_nb, total + fc_nb + 256)
    total = (total + fc_nb + 256 + num_classes * 4 + 255) & ~255
    return offs, total, fc_packed


class ResNet50Int8(ResNet50):
    def __init__(self, ctx, batch, weights=None, pad_mode=L.PAD_RAW0_I8, i8_arena_ptr=None, i8_arena_keepalive=None, **kw):
        super().__init__(ctx, batch, weights, **kw)
        self.pad_mode = pad_mode
        self._staged_key = None
        self.producer_stats = True   # conv epilogues accumulate the min/max the next DynamicQuantizeLinear needs
        self.fused_dql = False       # pointwise stride-1 convs quantize their input in the GEMM's loader (rten_hip_conv2d_int8_dql) ...
        self.fused_layers = None     # ... all that qualify (None) or the set autotune() measured to be faster that way
        # Quantized-output launches (rten_hip_conv2d_int8_qout): a conv whose output goes through ONE DynamicQuantizeLinear -- it is read by one
        # conv, or by several that share the quantized tensor (a stage's projection shortcut and first 1x1, which ort-quantize feeds from one
        # DynamicQuantizeLinear) -- runs that quantizer in its own epilogue, behind a grid-wide min / max.  In a bottleneck block: the c1 -> c2
        # and c2 -> c3 edges, whose f32 tensor is then never written (32 of the 53 layers), and the c3 -> next block edges, where the f32
        # tensor is still written for the residual Add (`qout_keeps_f32`).
        self.fused_qout = False
        readers = {}
        for m in self.specs:
            readers.setdefault(m["src"], []).append(m)
        residuals = {m["res"] for m in self.specs if m["res"]}

        def one_quantizer(rs):
            g = {(self.descs[r["name"]].c, self.descs[r["name"]].h, self.descs[r["name"]].w, tuple(self.descs[r["name"]].pads)) for r in rs}
            return len(g) == 1
        self.qout_next = {m["name"]: readers[m["dst"]][0] for m in self.specs
                          if readers.get(m["dst"]) and one_quantizer(readers[m["dst"]]) and m["dst"] != "stem"}
        self.qout_keeps_f32 = {m["name"] for m in self.specs if m["name"] in self.qout_next and m["dst"] in residuals}
        self._qout_off = set()       # layers that run the two-launch sequence: grid not resident at once (found at the first attempt), or
                                     # measured slower by autotune_qout() (the grid-wide exchange costs ~6 us: it pays on the larger tensors only)
        gb = ctx.lib.rten_hip_grid_sync_bytes()
        self.sync_arena = DeviceTensor(ctx, (gb * len(self.specs),), np.uint8)  # initialised once; every launch leaves its block as it found it
        ctx.call("rten_hip_grid_sync_reset", self.sync_arena.vp, len(self.specs))
        self.syncs = {l["name"]: C.c_void_p(self.sync_arena.ptr + i * gb) for i, l in enumerate(self.specs)}
        sb = ctx.lib.rten_hip_minmax_stats_bytes()
        self.stats_arena = DeviceTensor(ctx, (sb * (len(self.specs) + 1),), np.uint8)  # one statistics block per conv output + the max-pool's
        self.stats = {l["dst"]: C.c_void_p(self.stats_arena.ptr + i * sb) for i, l in enumerate(self.specs)}
        no_fold = os.environ.get("RTEN_INT8_NO_FOLD") == "1"  # A/B switch: the launch sequence before these two folds
        if not no_fold:
            self.stats["pool"] = C.c_void_p(self.stats_arena.ptr + len(self.specs) * sb)  # (rten_hip_max_pool2d_f32_stats)
        # A quantized tensor read by two convolutions in a row (a stage's shortcut and first 1x1) is followed by one Mul(x_scale, w_scale)
        # per reader: both products come out of the quantizer's launch (rten_hip_dynamic_quantize_linear_staged_products)
        self.fold_products = not no_fold

        def geom_of(m):
            d = self.descs[m["name"]]
            return (m["src"], d.c, d.h, d.w, tuple(d.pads))
        self.shared_next = {a["name"]: b["name"] for a, b in zip(self.specs, self.specs[1:]) if geom_of(a) == geom_of(b)}
        self._sc_for = {}
        self.q = quantize_weights(self.weights)
        n_max = max(int(np.prod(s)) for s in self.shapes.values())
        self.xq = DeviceTensor(ctx, (n_max,), np.uint8)
        self.xs = DeviceTensor(ctx, (1,), np.float32)
        self.xz = DeviceTensor(ctx, (1,), np.uint8)
        self.sc = DeviceTensor(ctx, (1,), np.float32)
        self.idesc, self.wq, self.ws, self.bq = {}, {}, {}, {}
        staged_max = 0
        for l in self.specs:
            d = self.descs[l["name"]]
            self.idesc[l["name"]] = L.Conv2dInt8Desc(d, 0, 1, 0, pad_mode, 1, 1)
            staged_max = max(staged_max, ctx.lib.rten_hip_conv2d_int8_staged_bytes(C.byref(self.idesc[l["name"]])))
        self.staged = DeviceTensor(ctx, (max(staged_max, 256),), np.uint8)  # quantized activations in the int8 kernel's layout
        # second (codes, x_scale, x_zero_point) set + cast_scale slot: with `concurrent` a projection shortcut runs on a second stream
        # next to c1 -> c2 and keeps reading ITS quantized input while the main stream quantizes the next tensors
        self.qsets = [(self.staged, self.xs, self.xz),
                      (DeviceTensor(ctx, (max(staged_max, 256),), np.uint8), DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.uint8))]
        self.sc_side = DeviceTensor(ctx, (1,), np.float32)
        self.scs = [self.sc, DeviceTensor(ctx, (1,), np.float32)]  # Mul(x_scale, w_scale) per quantized-input set (quantized-output launches)
        self.scs2 = [DeviceTensor(ctx, (1,), np.float32), DeviceTensor(ctx, (1,), np.float32)]  # ... and the second reader's product
        self._cur, self._side_reads = 0, None
        self.fc_tmp = DeviceTensor(ctx, (batch, self.num_classes), np.float32)
        # classifier RHS [K = 2048, N = 1000] staged once (rten_hip_gemm_int8_prepack: PackedBMatrix, Graph::prepack_weights)
        self.fc_packed_bytes = ctx.lib.rten_hip_gemm_int8_packed_bytes(2048, self.num_classes)
        self.fc_idesc = L.GemmInt8Desc(batch, self.num_classes, 2048, 2048, 1, 1, 2048, self.num_classes, 0, 1, 1, 0, 1,
                                       1, 0, 0, 0, 1 if self.fc_packed_bytes else 0)
        # int8 weight arena: per layer [staged weights + row sums | w_scale f32 | bias f32], then the classifier; ONE allocation so
        # that rank 0 of a batch-sharded job stages it once and RCCL broadcasts it (bench.py --config int8)
        offs, total, _ = i8_arena_layout(ctx.lib, batch, pad_mode, self.image, self.num_classes)
        self.i8_off, self.i8_arena_bytes = offs, total
        self.i8_arena = DeviceTensor(ctx, (total,), np.uint8, ptr=i8_arena_ptr, keepalive=i8_arena_keepalive)
        for name, (a, b, c) in offs.items():
            o = self.num_classes if name == "fc" else next(l["cout"] for l in self.specs if l["name"] == name)
            self.wq[name] = DeviceTensor(ctx, (b - a,), np.uint8, ptr=self.i8_arena.ptr + a, keepalive=self.i8_arena)
            self.ws[name] = DeviceTensor(ctx, (1,), np.float32, ptr=self.i8_arena.ptr + b, keepalive=self.i8_arena)
            self.bq[name] = DeviceTensor(ctx, (o,), np.float32, ptr=self.i8_arena.ptr + c, keepalive=self.i8_arena)

    def upload_weights(self):
        """Stage every weight into the int8 arena (rank 0 of a multi-GPU job; the others receive the broadcast)."""
        ctx = self.ctx
        for l in self.specs:
            wq, ws, b = self.q[l["name"]]
            raw = DeviceTensor.from_numpy(ctx, wq)
            ctx.call("rten_hip_conv2d_int8_prepack", C.byref(self.idesc[l["name"]]), raw.vp, self.wq[l["name"]].vp)
            ctx.sync()
            raw.free()
            self.ws[l["name"]].upload(np.array([ws], np.float32))
            self.bq[l["name"]].upload(b)
        wq, ws, b = self.q["fc"]  # [1000, 2048]: B[k, n] = wq[n, k] via strides
        if self.fc_packed_bytes:
            raw = DeviceTensor.from_numpy(ctx, wq)
            ctx.call("rten_hip_gemm_int8_prepack", 2048, self.num_classes, raw.vp, 1, 2048, 1, self.wq["fc"].vp)
            ctx.sync()
            raw.free()
        else:
            self.wq["fc"].upload(wq.reshape(-1).view(np.uint8))
        self.ws["fc"].upload(np.array([ws], np.float32))
        self.bq["fc"].upload(b)

    def _quantize(self, src, n):
        ctx = self.ctx
        ctx.call("rten_hip_dynamic_quantize_linear", n, src.vp, self.xq.vp, self.xs.vp, self.xz.vp)

    def _conv_q(self, l):
        """One conv layer with quantized-output launches enabled (single stream).  The two (codes, x_scale, x_zero_point, scale
        product) sets alternate: a launch reads its quantized input from one set and, when its output has a single consumer, writes
        that consumer's quantized input into the other."""
        ctx, name = self.ctx, l["name"]
        d, cv = self.idesc[name], self.idesc[name].conv
        geom = (l["src"], cv.c, cv.h, cv.w, tuple(cv.pads))
        flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
        res = self._act(l["res"]).vp if l["res"] else None
        st_in = self.stats.get(l["src"]) if self.producer_stats else None
        runs_qout = name in self.qout_next and name not in self._qout_off  # (measured: where both forms apply, quantizing the OUTPUT in the epilogue is
        #                                                                     worth more to the replayed graph than quantizing the input in the loader)
        if (self.fused_dql and self.fused_layers and name in self.fused_layers and not runs_qout and st_in is not None and self._prestaged != name
                and self._staged_key != geom and cv.kh == 1 and cv.kw == 1 and cv.stride_h == 1 and cv.stride_w == 1 and not any(cv.pads) and cv.c % 64 == 0):
            # a pointwise layer autotune() found faster with DynamicQuantizeLinear inside its own operand loader (no staged tensor at all)
            ctx.call("rten_hip_conv2d_int8_dql", C.byref(d), self._act(l["src"]).vp, st_in, self.wq[name].vp, self.ws[name].vp, self.bq[name].vp, res, flags,
                     self._act(l["dst"]).vp, self.stats[l["dst"]], None, None)
            return
        sc_mine = self._sc_for.pop(name, None)  # this layer's Mul(x_scale, w_scale) already came out of the quantizer's launch
        if self._prestaged == name:      # the producing conv quantized this input in its epilogue, scale product included
            pass
        elif self._staged_key == geom:   # same tensor, same staged layout as the previous conv: only the Mul(x_scale, w_scale) differs
            if sc_mine is None:
                ctx.call("rten_hip_mul_f32", 1, self.qsets[self._cur][1].vp, self.ws[name].vp, 1, self.scs[self._cur].vp)
        else:
            sc_mine = None
            self._cur = 1 - self._cur
            staged, xs, xz = self.qsets[self._cur]
            st = self.stats.get(l["src"]) if self.producer_stats else None
            nxt2 = self.shared_next.get(name) if self.fold_products else None
            if nxt2 is not None:
                muls = (C.c_void_p * 2)(self.ws[name].ptr, self.ws[nxt2].ptr)
                outs = (C.c_void_p * 2)(self.scs[self._cur].ptr, self.scs2[self._cur].ptr)
                ctx.call("rten_hip_dynamic_quantize_linear_staged_products", C.byref(d), self._act(l["src"]).vp, st, staged.vp, xs.vp, xz.vp, 2, muls, outs)
                self._sc_for[nxt2] = self.scs2[self._cur]
            elif st is not None:
                ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), self._act(l["src"]).vp, st, staged.vp, xs.vp, xz.vp, self.ws[name].vp,
                         self.scs[self._cur].vp)
            else:
                ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), self._act(l["src"]).vp, staged.vp, xs.vp, xz.vp, self.ws[name].vp,
                         self.scs[self._cur].vp)
        self._staged_key, self._prestaged = geom, None
        (staged, xs, xz), sc = self.qsets[self._cur], (sc_mine if sc_mine is not None else self.scs[self._cur])
        nxt = self.qout_next.get(name)
        if nxt is not None and name not in self._qout_off:
            other = 1 - self._cur
            (ostaged, oxs, oxz), osc = self.qsets[other], self.scs[other]
            y = self._act(l["dst"]).vp if name in self.qout_keeps_f32 else None
            rc = ctx.lib.rten_hip_conv2d_int8_qout(ctx.h, C.byref(d), staged.vp, self.wq[name].vp, xz.vp, None, sc.vp, self.bq[name].vp, res, flags, y,
                                                   self.stats[l["dst"]], self.syncs[name], C.byref(self.idesc[nxt["name"]]), ostaged.vp, oxs.vp, oxz.vp,
                                                   self.ws[nxt["name"]].vp, osc.vp)
            if rc == L.OK:
                nc = self.idesc[nxt["name"]].conv  # a second reader of the same tensor finds the staged codes through `_staged_key`
                self._cur, self._prestaged, self._staged_key = other, nxt["name"], (nxt["src"], nc.c, nc.h, nc.w, tuple(nc.pads))
                return
            if rc != L.ERR_UNSUPPORTED:
                ctx.check(rc)
            self._qout_off.add(name)  # more workgroups than the device holds at once: the two-launch sequence, from now on
        args = (C.byref(d), staged.vp, self.wq[name].vp, xz.vp, None, sc.vp, self.bq[name].vp, res, flags, self._act(l["dst"]).vp)
        if self.producer_stats:
            ctx.call("rten_hip_conv2d_int8_stats", *args, self.stats[l["dst"]])
        else:
            ctx.call("rten_hip_conv2d_int8", *args)

    def autotune_qout(self, reps=5):
        """Per single-consumer edge: the quantized-output launch against conv + the consumer's staging launch, timed back to back; edges
        where one launch is not at least 3 % faster keep the two launches.  Returns {layer: (two launches us, one launch us or None)}."""
        ctx = self.ctx
        saved, self.fused_qout = self.fused_qout, False
        self.forward()  # every layer's quantized input / statistics exist
        ctx.sync()
        self.fused_qout = saved
        table = {}

        def timed(fn):
            fn()
            best = 1e30
            for _ in range(2):
                ctx.timer_start(1)
                for _ in range(reps):
                    fn()
                ctx.timer_stop(1)
                best = min(best, ctx.timer_ms(1) / reps * 1e3)
            return best
        for l in self.specs:
            name, nxt = l["name"], self.qout_next.get(l["name"])
            if nxt is None:
                continue
            d, nd = self.idesc[name], self.idesc[nxt["name"]]
            (staged, xs, xz), (ostaged, oxs, oxz) = self.qsets
            st = self.stats.get(l["src"])
            src = self._act(l["src"])
            if st is not None:
                ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), src.vp, st, staged.vp, xs.vp, xz.vp, self.ws[name].vp, self.scs[0].vp)
            else:
                ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), src.vp, staged.vp, xs.vp, xz.vp, self.ws[name].vp, self.scs[0].vp)
            flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
            res = self._act(l["res"]).vp if l["res"] else None
            dst, dstat = self._act(l["dst"]), self.stats[l["dst"]]

            def two():
                ctx.call("rten_hip_conv2d_int8_stats", C.byref(d), staged.vp, self.wq[name].vp, xz.vp, None, self.scs[0].vp, self.bq[name].vp, res, flags, dst.vp, dstat)
                ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(nd), dst.vp, dstat, ostaged.vp, oxs.vp, oxz.vp, self.ws[nxt["name"]].vp, self.scs[1].vp)

            def one():
                return ctx.lib.rten_hip_conv2d_int8_qout(ctx.h, C.byref(d), staged.vp, self.wq[name].vp, xz.vp, None, self.scs[0].vp, self.bq[name].vp, res, flags,
                                                         dst.vp if name in self.qout_keeps_f32 else None,
                                                         dstat, self.syncs[name], C.byref(nd), ostaged.vp, oxs.vp, oxz.vp, self.ws[nxt["name"]].vp, self.scs[1].vp)
            us2 = timed(two)
            us1 = timed(one) if one() == L.OK else None
            table[name] = (us2, us1)
            if us1 is None or us1 > us2 * 0.97:
                self._qout_off.add(name)
            else:
                self._qout_off.discard(name)
        return table

    def qout_timeouts(self):
        """Number of quantized-output launches that gave up waiting for their grid (0 unless the residency assumption broke)."""
        n = C.c_int32(0)
        self.ctx.call("rten_hip_grid_sync_timeouts", self.sync_arena.vp, len(self.specs), C.byref(n))
        return n.value

    def _conv(self, l, ctx=None):
        # a per-layer workgroup tile (rten_hip_set_int8_tile; what a plan file's "<layer>": [tile, 0, 1, 0] entry gives the executor's step): {layer: 0..3}
        t = getattr(self, "tiles", {}).get(l["name"], -1)
        if t < 0:
            return self._conv_impl(l, ctx)
        self.ctx.call("rten_hip_set_int8_tile", t, None)
        try:
            return self._conv_impl(l, ctx)
        finally:
            self.ctx.call("rten_hip_set_int8_tile", -1, None)

    def _conv_impl(self, l, ctx=None):
        if self.fused_qout and not self.concurrent:
            return self._conv_q(l)
        ctx = self.ctx
        name = l["name"]
        src = self._act(l["src"])
        d = self.idesc[name]
        on_side = self.concurrent and name.endswith("ds")
        sc = self.sc_side if on_side else self.sc
        # DynamicQuantizeLinear, writing the codes straight into the consumer's staged layout, and the Mul(x_scale, w_scale)
        # that feeds the conv's cast_scale.  When the producing conv left min/max statistics, the first sweep is skipped.
        st = self.stats.get(l["src"]) if self.producer_stats else None
        geom = (l["src"], d.conv.c, d.conv.h, d.conv.w, tuple(d.conv.pads))
        cv = d.conv
        if (self.fused_dql and (self.fused_layers is None or name in self.fused_layers) and st is not None and not on_side and cv.kh == 1 and cv.kw == 1 and cv.stride_h == 1 and cv.stride_w == 1
                and not any(cv.pads) and cv.c % 64 == 0):
            # DynamicQuantizeLinear + ConvIntegerToFloat of this layer in one launch: no staged tensor (a later conv that shares this
            # input quantizes it again for itself, with the same result)
            flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
            if l["res"] in self._pending:
                ctx.wait(self.side)
                self._pending.discard(l["res"])
                self._side_reads = None
            ctx.call("rten_hip_conv2d_int8_dql", C.byref(d), src.vp, st, self.wq[name].vp, self.ws[name].vp, self.bq[name].vp,
                     self._act(l["res"]).vp if l["res"] else None, flags, self._act(l["dst"]).vp, self.stats[l["dst"]], None, None)
            return
        if self._staged_key == geom:
            # same tensor, same staged layout as the previous conv (a stage's downsample and first 1x1 conv): the graph has ONE
            # DynamicQuantizeLinear for it (ort-quantize reuses a quantized input), only the Mul(x_scale, w_scale) differs
            staged, xs, xz = self.qsets[self._cur]
            ctx.call("rten_hip_mul_f32", 1, xs.vp, self.ws[name].vp, 1, sc.vp)
        else:
            nxt = (1 - self._cur) if self.concurrent else 0
            if self._side_reads == nxt:  # the shortcut conv on the side stream still reads this set: join before overwriting it
                ctx.wait(self.side)
                self._side_reads = None
            self._cur = nxt
            staged, xs, xz = self.qsets[nxt]
            if st is not None:
                ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), src.vp, st, staged.vp, xs.vp, xz.vp, self.ws[name].vp, sc.vp)
            else:
                ctx.call("rten_hip_dynamic_quantize_linear_staged", C.byref(d), src.vp, staged.vp, xs.vp, xz.vp, self.ws[name].vp, sc.vp)
        self._staged_key = geom
        flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
        args = (C.byref(d), staged.vp, self.wq[name].vp, xz.vp, None, sc.vp, self.bq[name].vp,
                self._act(l["res"]).vp if l["res"] else None, flags, self._act(l["dst"]).vp)
        run_on = ctx
        if on_side:
            self.side.wait(ctx)  # the quantized input and its scale are ready
            run_on = self.side
            self._side_reads = self._cur
            self._pending.add(l["dst"])
        elif l["res"] in self._pending:
            ctx.wait(self.side)  # the shortcut's output (this conv's residual) is ready
            self._pending.discard(l["res"])
            self._side_reads = None
        if self.producer_stats:
            run_on.call("rten_hip_conv2d_int8_stats", *args, self.stats[l["dst"]])
        else:
            run_on.call("rten_hip_conv2d_int8", *args)

    def forward(self):
        ctx = self.ctx
        self._staged_key = None
        self._cur, self._side_reads, self._pending, self._prestaged = 0, None, set(), None
        self._sc_for = {}
        if self.concurrent and self.side is None:
            self.side = L.Context(ctx.device)
        if self.producer_stats:
            ctx.call("rten_hip_minmax_stats_reset", self.stats_arena.vp, len(self.specs) + 1)  # one launch for every layer's block (+ the pool's)
        self._conv(self.specs[0])
        if self.producer_stats and "pool" in self.stats:  # the pooled tensor is quantized next: its min / max come out of the pooling launch
            ctx.call("rten_hip_max_pool2d_f32_stats", C.byref(self.pool_desc), self.bufs["stem"].vp, self.bufs["pool"].vp, self.stats["pool"])
        else:
            ctx.call("rten_hip_max_pool2d_f32", C.byref(self.pool_desc), self.bufs["stem"].vp, self.bufs["pool"].vp)
        for l in self.specs[1:]:
            self._conv(l)
        last = self.specs[-1]["dst"]
        n, c, h, w = self.shapes[last]
        ctx.call("rten_hip_global_average_pool_f32", n * c, h * w, self.bufs[last].vp, self.gap.vp)
        # classifier: DynamicQuantizeLinear -> MatMulIntegerToFloat -> Add(bias)
        self._quantize(self.gap, n * c)
        ctx.call("rten_hip_mul_f32", 1, self.xs.vp, self.ws["fc"].vp, 1, self.sc.vp)
        ctx.call("rten_hip_gemm_int8", C.byref(self.fc_idesc), self.xq.vp, self.wq["fc"].vp, self.xz.vp, None, self.sc.vp, self.fc_tmp.vp)
        ctx.call("rten_hip_add_f32", n * self.num_classes, self.fc_tmp.vp, self.bq["fc"].vp, self.num_classes, self.logits.vp)

    def autotune(self, reps=5):
        """The int8 kernels pick their tile by shape; what is measured at load is, per pointwise layer, whether DynamicQuantizeLinear
        runs as its own staging launch or inside the conv's loader (rten_hip_conv2d_int8_dql).  The fused form wins where few
        output-channel tiles share the input (the 1x1 reduce layers), the staged form where many do (the expand layers re-read and
        re-quantize the f32 input once per tile row).  Returns {layer: (separate us, fused us)}."""
        ctx = self.ctx
        self.fused_dql, self.fused_layers = False, None
        self.forward()  # activations and statistics of every layer exist
        ctx.sync()
        table, chosen = {}, set()

        def timed(fn):
            fn()
            best = 1e30
            for _ in range(2):
                ctx.timer_start(1)
                for _ in range(reps):
                    fn()
                ctx.timer_stop(1)
                best = min(best, ctx.timer_ms(1) / reps * 1e3)
            return best
        for l in self.specs:
            name, d = l["name"], self.idesc[l["name"]]
            cv, st = d.conv, self.stats.get(l["src"])
            if st is None or not (cv.kh == 1 and cv.kw == 1 and cv.stride_h == 1 and cv.stride_w == 1 and not any(cv.pads) and cv.c % 64 == 0):
                continue
            src, staged, xs, xz = self._act(l["src"]), *self.qsets[0]
            flags = (L.CONV_RELU if l["relu"] else 0) | (L.CONV_RESIDUAL if l["res"] else 0)
            res = self._act(l["res"]).vp if l["res"] else None

            def separate():
                ctx.call("rten_hip_dynamic_quantize_linear_staged_stats", C.byref(d), src.vp, st, staged.vp, xs.vp, xz.vp, self.ws[name].vp, self.sc.vp)
                ctx.call("rten_hip_conv2d_int8_stats", C.byref(d), staged.vp, self.wq[name].vp, xz.vp, None, self.sc.vp, self.bq[name].vp, res, flags,
                         self._act(l["dst"]).vp, self.stats[l["dst"]])

            def fused():
                ctx.call("rten_hip_conv2d_int8_dql", C.byref(d), src.vp, st, self.wq[name].vp, self.ws[name].vp, self.bq[name].vp, res, flags,
                         self._act(l["dst"]).vp, self.stats[l["dst"]], None, None)
            us = (timed(separate), timed(fused))
            table[name] = us
            # a layer that shares its quantized input with the previous conv (a stage's downsample + first 1x1) saves no staging launch
            if us[1] < us[0] * 0.97:
                chosen.add(name)
        self.fused_dql, self.fused_layers = True, chosen
        self.variants = {n: "dql" for n in chosen}
        return table
