"""One convolution layer of ResNet-50 timed UNDER SELF-CO-RUN: the same layer on N streams at once (round 6).

The default schedule of bench.py runs several replicas of the model side by side ("lanes"), so a layer's launch never has the chip to itself: whatever its last partial
round of tiles leaves idle another replica's launch fills, and on real operand data the shader clock follows the bytes a kernel moves per FLOP (DESIGN.md section 2.2,
tools/probes/kloop2.hip).  A launch plan for that schedule is therefore chosen -- and a kernel of that schedule is described -- by the time per launch over N
streams running the layer together, not by a launch that is alone on the device.  N runner networks (rten_amd.workloads.resnet50.ResNet50) share one weight
arena; each runs the layer on its REAL input activations (a forward pass stopped right before the layer; the operand data matters), from a captured hipGraph of
`reps` launches per stream.  Used by tools/tune_corun.py (every candidate plan of every layer family) and by bench.py (the dominant family under the committed plan:
`roofline.dominant_kernel.co_run`).  Measurement infrastructure: nothing on the product path imports this."""
import time

import numpy as np

from .. import lib as L
from . import resnet50


class CoRun:
    def __init__(self, streams, batch=32, plan=None, weights=None, device=0, ctxs=None):
        # `ctxs`: borrowed contexts (bench.py hands over its lanes' own: every stream of a process should keep a hardware queue of its own -- two streams on one
        # queue cost a co-run ~20 %, the placement effect of DESIGN.md / docs/KERNELS.md "streams map onto hardware queues")
        self._own = ctxs is None
        self.ctxs = [L.Context(device) for _ in range(streams)] if ctxs is None else list(ctxs)[:streams]
        self.nets = []
        weights = weights if weights is not None else resnet50.make_weights()
        for i, ctx in enumerate(self.ctxs):
            kw = {} if i == 0 else dict(arena_ptr=self.nets[0].arena.ptr, arena_keepalive=self.nets[0].arena)
            net = resnet50.ResNet50(ctx, batch, weights, **kw)
            if i == 0:
                net.upload_weights()
                ctx.sync()
            net.x.upload(np.random.default_rng(1234 + i).random((batch, 3, 224, 224), dtype=np.float32))
            net.variants = {k: tuple(v) for k, v in (plan or {}).items() if k != "fc"}
            self.nets.append(net)
        self.specs = self.nets[0].specs
        self.descs = self.nets[0].descs
        self._at = None

    def families(self):
        """{(O, C, k, stride, H, residual): [(index, layer spec), ...]}: layers of one family share a plan."""
        fams = {}
        for idx, l in enumerate(self.specs):
            d = self.descs[l["name"]]
            fams.setdefault((d.o, d.c, d.kh, d.stride_h, d.h, bool(l["res"])), []).append((idx, l))
        return fams

    def flops(self, name):
        d = self.descs[name]
        return 2.0 * d.o * (d.c // d.groups) * d.kh * d.kw * d.n * d.out_h * d.out_w

    def position(self, idx):
        """Every network runs its forward pass up to (not including) layer `idx`: the layer's real inputs are in place (the runner reuses activation buffers)."""
        if self._at != idx:
            for net in self.nets:
                net.forward(upto=idx)
            for c in self.ctxs:
                c.sync()
            self._at = idx

    def measure(self, idx, plan, reps=12, rounds=3):
        """Microseconds per launch of layer `idx` under `plan` = [variant, split mode, K groups, tile order], all streams running it at once."""
        l = self.specs[idx]
        self.position(idx)
        graphs = []
        try:
            for net in self.nets:
                net.variants[l["name"]] = tuple(plan)
                net._conv(l)  # warm: scratch growth outside the capture
            for c in self.ctxs:
                c.sync()
            for net in self.nets:
                net.ctx.graph_begin()
                for _ in range(reps):
                    net._conv(l)
                graphs.append((net.ctx, net.ctx.graph_end()))
            best = 1e30
            for _ in range(rounds):
                t0 = time.perf_counter()
                for c, g in graphs:
                    c.graph_launch(g)
                for c in self.ctxs:
                    c.sync()
                best = min(best, (time.perf_counter() - t0) / (reps * len(self.nets)) * 1e6)
            return best
        finally:
            for c, g in graphs:
                c.graph_destroy(g)

    def close(self):
        self.nets = []
        if self._own:
            for c in self.ctxs:
                c.close()
        self.ctxs = []
