"""Device-resident tensors for the HIP backend (the device analogue of rten's `Value`, src/value.rs:487).

A `DeviceTensor` is a contiguous row-major buffer in HBM plus shape/dtype; views (`reshape`) share the
allocation.  Memory comes either from the library's allocator (`rten_hip_malloc`) or from a foreign
allocation (e.g. a torch CUDA tensor: pass `ptr=` and keep the owner alive via `keepalive`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .lib import Context


class DeviceTensor:
    def __init__(self, ctx: Context, shape, dtype=np.float32, ptr: int | None = None, keepalive=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._owned = ptr is None
        self._keepalive = keepalive
        if ptr is None:
            if hasattr(ctx, "alloc"):  # Context: pooled when enable_pool() is on
                self.ptr = ctx.alloc(self.nbytes)
            else:
                p = C.c_void_p()
                ctx.call("rten_hip_malloc", C.c_size_t(max(self.nbytes, 16)), C.byref(p))
                self.ptr = p.value
        else:
            self.ptr = int(ptr)

    # ---- construction helpers
    @classmethod
    def from_numpy(cls, ctx: Context, arr) -> "DeviceTensor":
        arr = np.asarray(arr)
        shape = arr.shape  # np.ascontiguousarray would promote 0-d to 1-d; scalars must stay rank 0
        t = cls(ctx, shape, arr.dtype)
        t.upload(arr)
        return t

    def upload(self, arr):
        arr = np.ascontiguousarray(np.asarray(arr, dtype=self.dtype))
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        if self.nbytes:
            self.ctx.call("rten_hip_memcpy_h2d", C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes))

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            self.ctx.call("rten_hip_memcpy_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(self.nbytes))
        return out

    def reshape(self, *shape) -> "DeviceTensor":
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        n = int(np.prod(self.shape, dtype=np.int64))
        shape = list(shape)
        if -1 in shape:
            known = int(np.prod([s for s in shape if s != -1], dtype=np.int64))
            shape[shape.index(-1)] = n // max(known, 1)
        assert int(np.prod(shape, dtype=np.int64)) == n
        return DeviceTensor(self.ctx, shape, self.dtype, ptr=self.ptr, keepalive=self)

    def view(self, shape) -> "DeviceTensor":
        """A tensor of `shape` aliasing the first prod(shape) elements of this buffer (static activation plans)."""
        n = int(np.prod(shape, dtype=np.int64))
        assert n * self.dtype.itemsize <= self.nbytes
        return DeviceTensor(self.ctx, shape, self.dtype, ptr=self.ptr, keepalive=self)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def vp(self):
        return C.c_void_p(self.ptr)

    def free(self):
        if self._owned and self.ptr and self.ctx.h:
            if hasattr(self.ctx, "release"):
                self.ctx.release(self.ptr, self.nbytes)
            else:
                self.ctx.call("rten_hip_free", C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __repr__(self):
        return f"DeviceTensor(shape={self.shape}, dtype={self.dtype}, ptr=0x{self.ptr:x})"
