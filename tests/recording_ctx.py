"""A context that records instead of launching: host logic (launch sequences, shard bookkeeping, the bench's control flow) tested without a device.

Pure host functions of the library (sizes, layouts: `*_bytes`) go to the real librten_hip.so; everything that would touch a device is appended to
`log`.  A device-to-host copy fills its destination with zeros so that hashes of "downloaded" results are deterministic."""
import ctypes as C

from rten_amd import lib as L


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
