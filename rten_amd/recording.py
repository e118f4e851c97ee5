"""A context that records instead of launching: host logic (launch sequences, shard bookkeeping, the bench's control flow) tested without a device.

Pure host functions of the library (sizes, layouts: `*_bytes`) go to the real librten_hip.so; everything that would touch a device is appended to
`log`.  A device-to-host copy fills its destination with zeros so that hashes of "downloaded" results are deterministic."""
import ctypes as C

from . import lib as L


class _LibProxy:
    def __init__(self, real, log):
        self._real, self._log = real, log

    def __getattr__(self, name):
        if name.endswith("_bytes") or name in ("rten_hip_abi_version", "rten_hip_num_gemm_variants"):
            return getattr(self._real, name)

        def recorded(*args):
            self._log.append(name)
            return L.OK
        return recorded


class RecordingCtx:
    def __init__(self, device=0):
        self.device = device
        self.log = []
        self.lib = _LibProxy(L.load(), self.log)
        self.h = C.c_void_p(0x1000)
        self._next = 1 << 32
        self._graphs = 0

    def alloc(self, nbytes):
        p = self._next
        self._next += (max(int(nbytes), 16) + 255) & ~255
        return p

    def release(self, ptr, nbytes):
        pass

    def call(self, name, *args):
        self.log.append(name)
        if name == "rten_hip_memcpy_d2h":  # (dst host pointer, src, size): zeros
            C.memset(args[0], 0, args[2].value if hasattr(args[2], "value") else int(args[2]))

    def check(self, rc):
        assert rc == L.OK

    def sync(self):
        self.log.append("sync")

    # graphs / timers / profiling: bookkeeping only
    def graph_begin(self):
        self.log.append("graph_begin")

    def graph_end(self):
        self.log.append("graph_end")
        self._graphs += 1
        return self._graphs

    def graph_launch(self, g):
        self.log.append("graph_launch")

    def graph_destroy(self, g):
        pass

    def timer_start(self, slot=0):
        pass

    def timer_stop(self, slot=0):
        pass

    def timer_ms(self, slot=0):
        return 1.0

    def profile(self, on):
        pass

    def profile_reset(self):
        pass

    def profile_report(self):
        return []

    def wait(self, other):
        self.log.append("wait")

    def set_gemm_variant(self, v):
        self.log.append("set_gemm_variant")

    def device_info(self):
        return {"name": "recording context (no device)", "compute_units": 256, "clock_mhz": 2400, "mem_bytes": 0}


class RecordingModel:
    """Stand-in for lib.Model (rten_hip_model_*) on a RecordingCtx: the executor path of bench.py -- load (with / without the receive-weights flag),
    weight arena, bind, prepare, run, sync, output -- with every device action logged instead of issued.  The arena size is host arithmetic on the
    model bytes (every rank computes the same number), planned_steps counts the plan file's entries."""

    def __init__(self, ctx, onnx_bytes, plan_json=None, chains=1, receive_weights=False):
        import json
        self.ctx, self.chains, self.receive_weights = ctx, chains, receive_weights
        ctx.log.append("model_load_receive" if receive_weights else "model_load")
        self.inputs, self.outputs, self.num_steps = ["x"], ["logits"], 57
        self.input_ptrs, self.planned_steps, self.warning = {}, 0, ""
        self._arena = (ctx.alloc(len(onnx_bytes)), (len(onnx_bytes) + 255) & ~255)
        self._plan = json.loads(plan_json) if plan_json else None
        self._batch = 0

    def weight_arena(self):
        return self._arena

    def clone(self, ctx):
        import json
        ctx.log.append("model_clone")
        m = RecordingModel.__new__(RecordingModel)
        m.__dict__.update(self.__dict__)
        m.ctx, m.input_ptrs = ctx, {}
        return m

    def bind_input(self, name, shape):
        self._batch = int(shape[0])
        n = 1
        for d in shape:
            n *= int(d)
        self.input_ptrs[name] = self.ctx.alloc(4 * n)
        return self.input_ptrs[name]

    def prepare(self, tune=False):
        self.ctx.log.append("model_prepare_tune" if tune else "model_prepare")
        p = self._plan or {}
        self.planned_steps = sum(len(v) for v in p.values())
        for _ in range(self.chains):
            self.ctx.graph_begin()
            self.ctx.graph_end()

    def plan_json(self):
        import json
        return json.dumps(self._plan or {"recorded": {}})

    def run(self, inputs_written_on_caller_stream=False, join=True):
        self.ctx.graph_launch(0)

    def sync(self):
        self.ctx.sync()

    def output(self, i=0):
        return self.ctx.alloc(4 * self._batch * 1000), (self._batch, 1000)

    def profile_pass(self, steps):
        return []

    def close(self):
        pass
