"""ctypes binding of the C ABI in include/rten_hip.h (librten_hip.so, gfx950 only).

This is plumbing: it declares the C structs / prototypes and raises when the native library is
missing or reports an error.  There is deliberately NO fallback path: if the HIP extension cannot be
loaded or no MI355X is visible, every operator raises (`BackendUnavailable`).
"""
from __future__ import annotations

import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RTEN_HIP_LIBRARY") or os.path.join(_HERE, "librten_hip.so")  # (override: instrumented tuning builds)

OK = 0
ERR_INVALID_VALUE, ERR_INCOMPATIBLE_SHAPES, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE = 1, 2, 3, 4, 5

BIAS_NONE, BIAS_PER_ROW, BIAS_PER_COL = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
CONV_RELU, CONV_RESIDUAL = 1, 2
PAD_ZERO_POINT, PAD_RAW0_I8, PAD_RAW0_U8 = 0, 1, 2
MODEL_RECEIVE_WEIGHTS = 1  # rten_hip_model_load_ex flag


class BackendUnavailable(RuntimeError):
    """The HIP extension is missing or no gfx950 device is usable."""


class HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rten_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


class GemmDesc(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
                ("a_rs", C.c_int64), ("a_cs", C.c_int64), ("b_rs", C.c_int64), ("b_cs", C.c_int64),
                ("ldc", C.c_int64), ("batch", C.c_int32),
                ("a_bs", C.c_int64), ("b_bs", C.c_int64), ("c_bs", C.c_int64),
                ("batch_inner", C.c_int32), ("a_bsi", C.c_int64), ("b_bsi", C.c_int64), ("c_bsi", C.c_int64),
                ("alpha", C.c_float), ("beta", C.c_float), ("bias_kind", C.c_int32), ("act", C.c_int32)]


def gemm_desc(m, n, k, a_rs, a_cs, b_rs, b_cs, ldc, batch=1, a_bs=0, b_bs=0, c_bs=0, alpha=1.0, beta=0.0,
              bias_kind=BIAS_NONE, act=ACT_NONE, batch_inner=0, a_bsi=0, b_bsi=0, c_bsi=0) -> GemmDesc:
    return GemmDesc(m, n, k, a_rs, a_cs, b_rs, b_cs, ldc, batch, a_bs, b_bs, c_bs, batch_inner, a_bsi, b_bsi, c_bsi,
                    alpha, beta, bias_kind, act)


class SdpaDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("heads", C.c_int32), ("s", C.c_int32), ("t", C.c_int32), ("d", C.c_int32),
                ("dv", C.c_int32),
                ("q_bs", C.c_int64), ("q_hs", C.c_int64), ("q_rs", C.c_int64),
                ("k_bs", C.c_int64), ("k_hs", C.c_int64), ("k_rs", C.c_int64),
                ("v_bs", C.c_int64), ("v_hs", C.c_int64), ("v_rs", C.c_int64),
                ("o_bs", C.c_int64), ("o_hs", C.c_int64), ("o_rs", C.c_int64),
                ("mask_batch_stride", C.c_int64), ("mask_row_stride", C.c_int64), ("scale", C.c_float),
                ("flush_nan_to_zero", C.c_int32)]


class GemmInt8Desc(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
                ("a_rs", C.c_int64), ("a_cs", C.c_int64), ("b_rs", C.c_int64), ("b_cs", C.c_int64),
                ("ldc", C.c_int64), ("a_signed", C.c_int32), ("b_signed", C.c_int32),
                ("a_zp_len", C.c_int32), ("b_zp_len", C.c_int32), ("scale_len", C.c_int32),
                ("batch", C.c_int32), ("a_bs", C.c_int64), ("b_bs", C.c_int64), ("c_bs", C.c_int64),
                ("b_prepacked", C.c_int32)]


class Conv2dDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("o", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("pads", C.c_int32 * 4),
                ("stride_h", C.c_int32), ("stride_w", C.c_int32), ("dil_h", C.c_int32), ("dil_w", C.c_int32),
                ("groups", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32)]


class Conv2dInt8Desc(C.Structure):
    _fields_ = [("conv", Conv2dDesc), ("x_signed", C.c_int32), ("w_signed", C.c_int32),
                ("w_zp_len", C.c_int32), ("pad_mode", C.c_int32), ("weights_packed", C.c_int32), ("x_staged", C.c_int32), ("scale_len", C.c_int32)]


class Pool2dDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride_h", C.c_int32), ("stride_w", C.c_int32),
                ("pads", C.c_int32 * 4), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("count_include_pad", C.c_int32)]


# every symbol include/rten_hip.h declares (checked by tests/test_abi.py against the header)
_VP, _I32, _I64, _F32, _U32, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32, C.c_size_t
PROTOTYPES = {
    "rten_hip_init": (_I32, [_I32, _VP, C.POINTER(_VP)]),
    "rten_hip_destroy": (_I32, [_VP]),
    "rten_hip_last_error": (C.c_char_p, [_VP]),
    "rten_hip_abi_version": (_I32, []),
    "rten_hip_sync": (_I32, [_VP]),
    "rten_hip_device_info": (_I32, [_VP, C.c_char_p, _I32, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I64)]),
    "rten_hip_malloc": (_I32, [_VP, _SZ, C.POINTER(_VP)]),
    "rten_hip_free": (_I32, [_VP, _VP]),
    "rten_hip_memcpy_h2d": (_I32, [_VP, _VP, _VP, _SZ]),
    "rten_hip_memcpy_d2h": (_I32, [_VP, _VP, _VP, _SZ]),
    "rten_hip_memcpy_d2d": (_I32, [_VP, _VP, _VP, _SZ]),
    "rten_hip_memset": (_I32, [_VP, _VP, _I32, _SZ]),
    "rten_hip_timer_start": (_I32, [_VP, _I32]),
    "rten_hip_timer_stop": (_I32, [_VP, _I32]),
    "rten_hip_timer_elapsed_ms": (_I32, [_VP, _I32, C.POINTER(_F32)]),
    "rten_hip_graph_begin": (_I32, [_VP]),
    "rten_hip_graph_end": (_I32, [_VP, C.POINTER(C.c_uint64)]),
    "rten_hip_graph_abort": (_I32, [_VP]),
    "rten_hip_graph_launch": (_I32, [_VP, C.c_uint64]),
    "rten_hip_graph_destroy": (_I32, [_VP, C.c_uint64]),
    "rten_hip_profile_enable": (_I32, [_VP, _I32]),
    "rten_hip_profile_reset": (_I32, [_VP]),
    "rten_hip_profile_report": (_I32, [_VP, C.c_char_p, _I32]),
    "rten_hip_calc_output_size_and_padding": (_I32, [_I32] * 7 + [C.POINTER(_I32), _I32, _I32, _I32, C.POINTER(_I32),
                                                                 C.POINTER(_I32), C.POINTER(C.c_char_p)]),
    "rten_hip_gemm_f32": (_I32, [_VP, C.POINTER(GemmDesc), _VP, _VP, _VP, _VP]),
    "rten_hip_set_gemv_order": (_I32, [_VP, _I32, _I32]),
    "rten_hip_gemm_int8": (_I32, [_VP, C.POINTER(GemmInt8Desc), _VP, _VP, _VP, _VP, _VP, _VP]),
    "rten_hip_gemm_int8_packed_bytes": (_SZ, [_I32, _I32]),
    "rten_hip_gemm_int8_prepack": (_I32, [_VP, _I32, _I32, _VP, _I64, _I64, _I32, _VP]),
    "rten_hip_comm_get_unique_id": (_I32, [_VP, _VP]),
    "rten_hip_comm_init_rank": (_I32, [_VP, _VP, _I32, _I32, C.POINTER(_VP)]),
    "rten_hip_broadcast": (_I32, [_VP, _VP, _VP, _SZ, _I32]),
    "rten_hip_comm_world_size": (_I32, [_VP, C.POINTER(_I32), C.POINTER(_I32)]),
    "rten_hip_comm_destroy": (_I32, [_VP, _VP]),
    "rten_hip_conv2d_f32_packed_bytes": (_SZ, [C.POINTER(Conv2dDesc)]),
    "rten_hip_conv2d_f32_prepack": (_I32, [_VP, C.POINTER(Conv2dDesc), _VP, _VP]),
    "rten_hip_conv2d_f32": (_I32, [_VP, C.POINTER(Conv2dDesc), _VP, _VP, _I32, _VP, _VP, _U32, _VP]),
    "rten_hip_conv2d_f32_pair_supported": (_I32, [C.POINTER(Conv2dDesc), C.POINTER(Conv2dDesc)]),
    "rten_hip_conv2d_f32_pair": (_I32, [_VP, C.POINTER(Conv2dDesc), _VP, _VP, _VP, _VP, _U32, _VP, C.POINTER(Conv2dDesc), _VP, _VP, _U32, _VP]),
    "rten_hip_conv2d_f32_pair_shortcut_supported": (_I32, [C.POINTER(Conv2dDesc), C.POINTER(Conv2dDesc), C.POINTER(Conv2dDesc)]),
    "rten_hip_conv2d_f32_pair_shortcut": (_I32, [_VP, C.POINTER(Conv2dDesc), _VP, _VP, _VP, C.POINTER(Conv2dDesc), _VP, _VP, _VP, _U32, _VP, C.POINTER(Conv2dDesc), _VP, _VP, _U32, _VP]),
    "rten_hip_conv2d_int8": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _VP]),
    "rten_hip_conv2d_int8_packed_bytes": (_SZ, [C.POINTER(Conv2dInt8Desc)]),
    "rten_hip_conv2d_int8_prepack": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP]),
    "rten_hip_conv2d_int8_staged_bytes": (_SZ, [C.POINTER(Conv2dInt8Desc)]),
    "rten_hip_minmax_stats_bytes": (_SZ, []),
    "rten_hip_minmax_stats_reset": (_I32, [_VP, _VP, _I32]),
    "rten_hip_conv2d_int8_stats": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _VP, _VP]),
    "rten_hip_conv2d_int8_dql": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP]),
    "rten_hip_grid_sync_bytes": (_SZ, []),
    "rten_hip_conv2d_int8_qout": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, _VP, _VP, _VP,
                                          C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP]),
    "rten_hip_grid_sync_reset": (_I32, [_VP, _VP, _I32]),
    "rten_hip_grid_sync_timeouts": (_I32, [_VP, _VP, _I32, C.POINTER(_I32)]),
    "rten_hip_dynamic_quantize_linear_staged_stats": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rten_hip_dynamic_quantize_linear_staged_products": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP, _I32, C.POINTER(_VP), C.POINTER(_VP)]),
    "rten_hip_dynamic_quantize_linear_staged": (_I32, [_VP, C.POINTER(Conv2dInt8Desc), _VP, _VP, _VP, _VP, _VP, _VP]),
    "rten_hip_dynamic_quantize_linear": (_I32, [_VP, _I64, _VP, _VP, _VP, _VP]),
    "rten_hip_cast_scale": (_I32, [_VP, _I64, _VP, _VP, _I32, _VP]),
    "rten_hip_softmax_f32": (_I32, [_VP, _I64, _I32, _VP, _VP, _I64, _I64, _I32, _VP]),
    "rten_hip_layer_norm_f32": (_I32, [_VP, _I64, _I32, _VP, _VP, _VP, _F32, _F32, _F32, _VP]),
    "rten_hip_add_layer_norm_f32": (_I32, [_VP, _I64, _I32, _VP, _VP, _VP, _VP, _F32, _F32, _F32, _VP]),
    "rten_hip_batch_norm_f32": (_I32, [_VP, _I32, _I32, _I64, _VP, _VP, _VP, _VP, _VP, _F32, _VP]),
    "rten_hip_relu_f32": (_I32, [_VP, _I64, _VP, _VP]),
    "rten_hip_gelu_f32": (_I32, [_VP, _I64, _VP, _VP]),
    "rten_hip_erf_f32": (_I32, [_VP, _I64, _VP, _VP]),
    "rten_hip_tanh_f32": (_I32, [_VP, _I64, _VP, _VP]),
    "rten_hip_add_f32": (_I32, [_VP, _I64, _VP, _VP, _I64, _VP]),
    "rten_hip_mul_f32": (_I32, [_VP, _I64, _VP, _VP, _I64, _VP]),
    "rten_hip_sub_f32": (_I32, [_VP, _I64, _VP, _VP, _I64, _VP]),
    "rten_hip_div_f32": (_I32, [_VP, _I64, _VP, _VP, _I64, _VP]),
    "rten_hip_binary_broadcast_f32": (_I32, [_VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rten_hip_transpose_b32": (_I32, [_VP, _I32, _VP, _VP, _VP, _VP]),
    "rten_hip_copy_strided_b32": (_I32, [_VP, _I32, _VP, _VP, _VP, _VP]),
    "rten_hip_elementwise_nd": (_I32, [_VP, _I32, _I32, _VP, _VP, _I32, _VP, _VP, _I32, _VP, _VP, _VP, _VP, _I32]),
    "rten_hip_gather_axis_b32": (_I32, [_VP, _I64, _I64, _I64, _I64, _VP, _VP, _VP]),
    "rten_hip_copy_rows_b32": (_I32, [_VP, _I64, _I64, _VP, _I64, _VP, _I64]),
    "rten_hip_capture_active": (_I32, [_VP]),
    "rten_hip_reduce_sum_strided_f32": (_I32, [_VP, _I32, _VP, _VP, _I32, _VP, _VP, _VP, _VP]),
    "rten_hip_reduce_mean_strided_f32": (_I32, [_VP, _I32, _VP, _VP, _I32, _VP, _VP, _VP, _VP]),
    "rten_hip_conv_transpose_output_size": (_I32, [_I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP]),
    "rten_hip_conv_transpose2d_f32": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "rten_hip_matmul_nbits_f32": (_I32, [_VP, _I64, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP]),
    "rten_hip_add_channel_bias_f32": (_I32, [_VP, _I32, _I32, _I64, _VP, _VP, _VP]),
    "rten_hip_max_pool2d_f32": (_I32, [_VP, C.POINTER(Pool2dDesc), _VP, _VP]),
    "rten_hip_max_pool2d_f32_stats": (_I32, [_VP, C.POINTER(Pool2dDesc), _VP, _VP, _VP]),
    "rten_hip_average_pool2d_f32": (_I32, [_VP, C.POINTER(Pool2dDesc), _VP, _VP]),
    "rten_hip_global_average_pool_f32": (_I32, [_VP, _I64, _I32, _VP, _VP]),
    "rten_hip_sdpa_f32": (_I32, [_VP, C.POINTER(SdpaDesc), _VP, _VP, _VP, _VP, _VP]),
    "rten_hip_gather_rows_f32": (_I32, [_VP, _I64, _I32, _I32, _VP, _VP, _VP]),
    "rten_hip_set_gemm_variant_override": (_I32, [_VP, _I32]),
    "rten_hip_num_gemm_variants": (_I32, []),
    "rten_hip_set_gemm_split": (_I32, [_VP, _I32, _I32]),
    "rten_hip_stream_wait": (_I32, [_VP, _VP]),
    "rten_hip_set_gemm_order": (_I32, [_VP, _I32]),
    "rten_hip_set_int8_path": (_I32, [_VP, _I32]),
    "rten_hip_set_int8_tile": (_I32, [_VP, _I32, C.POINTER(C.c_int32)]),
    "rten_hip_set_sdpa_path": (_I32, [_VP, _I32]),
    # the plan executor behind the C ABI (csrc/graph_abi.cpp)
    "rten_hip_model_load": (_I32, [_VP, _VP, _SZ, C.c_char_p, _I32, _I32, C.POINTER(_VP)]),
    "rten_hip_model_load_ex": (_I32, [_VP, _VP, _SZ, C.c_char_p, _I32, _U32, C.POINTER(_VP)]),
    "rten_hip_model_load_error": (C.c_char_p, []),
    "rten_hip_model_clone": (_I32, [_VP, _VP, C.POINTER(_VP)]),
    "rten_hip_model_weight_arena": (_I32, [_VP, C.POINTER(_VP), C.POINTER(_SZ)]),
    "rten_hip_model_plan_json": (_I32, [_VP, C.c_char_p, _SZ, C.POINTER(_SZ)]),
    "rten_hip_model_set_plan": (_I32, [_VP, C.c_char_p]),
    "rten_hip_model_profile": (_I32, [_VP, _I32, C.c_char_p, _SZ, C.POINTER(_SZ)]),
    "rten_hip_device_id": (_I32, [_VP]),
    "rten_hip_tuning_save": (_I32, [_VP, C.POINTER(_I32)]),
    "rten_hip_tuning_restore": (_I32, [_VP, C.POINTER(_I32)]),
    "rten_hip_model_last_error": (C.c_char_p, [_VP]),
    "rten_hip_model_info": (_I32, [_VP, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "rten_hip_model_input_name": (C.c_char_p, [_VP, _I32]),
    "rten_hip_model_output_name": (C.c_char_p, [_VP, _I32]),
    "rten_hip_model_bind_input": (_I32, [_VP, _I32, C.POINTER(_I64), _I32, C.POINTER(_VP)]),
    "rten_hip_model_prepare": (_I32, [_VP, _I32]),
    "rten_hip_model_run": (_I32, [_VP, _U32]),
    "rten_hip_model_sync": (_I32, [_VP]),
    "rten_hip_model_output": (_I32, [_VP, _I32, C.POINTER(_VP), C.POINTER(_I64), C.POINTER(_I32)]),
    "rten_hip_model_input_dtype": (_I32, [_VP, _I32, C.POINTER(_I32)]),
    "rten_hip_model_output_dtype": (_I32, [_VP, _I32, C.POINTER(_I32)]),
    "rten_hip_model_destroy": (_I32, [_VP]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen librten_hip.so and bind every prototype.  Works without a GPU (symbols only)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise BackendUnavailable(f"{SO_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Context:
    """RAII wrapper of rten_hip_ctx.  `stream` may be a raw hipStream_t (e.g. torch's current stream)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.rten_hip_init(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc == ERR_NO_DEVICE:
            raise BackendUnavailable("no usable gfx950 (MI355X) device: the HIP backend has no CPU fallback")
        if rc != OK:
            raise HipError(rc, "rten_hip_init failed")
        self.h = h
        self.device = device
        self._pool_on = False
        self._pool = {}  # byte size -> free device pointers

    # Device buffer pool (src/buffer_pool.rs; the C++ layer's Context::enable_pool): off by default.  When on, freed tensors
    # go to a free list keyed by exact size and are handed out again without hipMalloc / hipFree (neither of which is
    # asynchronous); the context's single stream orders the reuse.
    def enable_pool(self, on: bool = True):
        self._pool_on = on
        if not on:
            self.trim_pool()

    def alloc(self, nbytes: int) -> int:
        nbytes = max(int(nbytes), 16)
        free = self._pool.get(nbytes) if self._pool_on else None
        if free:
            return free.pop()
        p = C.c_void_p()
        self.call("rten_hip_malloc", C.c_size_t(nbytes), C.byref(p))
        return p.value

    def release(self, ptr: int, nbytes: int):
        if self._pool_on:
            self._pool.setdefault(max(int(nbytes), 16), []).append(ptr)
        else:
            self.call("rten_hip_free", C.c_void_p(ptr))

    def trim_pool(self):
        for free in self._pool.values():
            for p in free:
                self.call("rten_hip_free", C.c_void_p(p))
        self._pool = {}

    def close(self):
        if getattr(self, "h", None):
            self.trim_pool()
            self.lib.rten_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != OK:
            raise HipError(rc, self.lib.rten_hip_last_error(self.h).decode(errors="replace"))

    def call(self, name: str, *args):
        self.check(getattr(self.lib, name)(self.h, *args))

    def sync(self):
        self.call("rten_hip_sync")

    def device_info(self):
        name = C.create_string_buffer(128)
        cus, mhz, mem = C.c_int32(), C.c_int32(), C.c_int64()
        self.call("rten_hip_device_info", name, 128, C.byref(cus), C.byref(mhz), C.byref(mem))
        return {"name": name.value.decode(), "compute_units": cus.value, "clock_mhz": mhz.value, "mem_bytes": mem.value}

    # timers / graphs / profiling
    def timer_start(self, slot=0):
        self.call("rten_hip_timer_start", slot)

    def timer_stop(self, slot=0):
        self.call("rten_hip_timer_stop", slot)

    def timer_ms(self, slot=0) -> float:
        ms = C.c_float()
        self.call("rten_hip_timer_elapsed_ms", slot, C.byref(ms))
        return ms.value

    def graph_begin(self):
        self.call("rten_hip_graph_begin")

    def graph_end(self) -> int:
        g = C.c_uint64()
        self.call("rten_hip_graph_end", C.byref(g))
        return g.value

    def graph_launch(self, g: int):
        self.call("rten_hip_graph_launch", C.c_uint64(g))

    def graph_destroy(self, g: int):
        self.call("rten_hip_graph_destroy", C.c_uint64(g))

    def profile(self, on: bool):
        self.call("rten_hip_profile_enable", 1 if on else 0)

    def profile_reset(self):
        self.call("rten_hip_profile_reset")

    def profile_report(self):
        buf = C.create_string_buffer(1 << 16)
        self.call("rten_hip_profile_report", buf, len(buf))
        return json.loads(buf.value.decode())

    def wait(self, other: "Context"):
        """Work enqueued on this context from now on runs after everything enqueued on `other` so far."""
        self.call("rten_hip_stream_wait", other.h)

    def set_gemm_variant(self, v: int):
        self.call("rten_hip_set_gemm_variant_override", v)


class Comm:
    """RAII wrapper of rten_hip_comm (RCCL behind the C ABI): the weight-arena broadcast of a batch-sharded deployment.
    `unique_id()` on rank 0, hand the 128 bytes to the other ranks by any host channel, then `Comm(ctx, id, world, rank)`."""

    ID_BYTES = 128

    @staticmethod
    def unique_id(ctx: Context) -> bytes:
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        ctx.call("rten_hip_comm_get_unique_id", buf)
        return bytes(buf)

    def __init__(self, ctx: Context, uid: bytes, world_size: int, rank: int):
        if len(uid) != Comm.ID_BYTES:
            raise ValueError("communicator id must be 128 bytes")
        self.ctx = ctx
        h = C.c_void_p()
        buf = (C.c_uint8 * Comm.ID_BYTES).from_buffer_copy(uid)
        ctx.call("rten_hip_comm_init_rank", buf, world_size, rank, C.byref(h))
        self.h = h
        self.world_size, self.rank = world_size, rank

    def broadcast(self, dptr: int, nbytes: int, root: int = 0):
        """In-place broadcast of a device buffer on the context's stream."""
        self.ctx.call("rten_hip_broadcast", self.h, C.c_void_p(dptr), C.c_size_t(nbytes), root)

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.call("rten_hip_comm_destroy", self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """RAII wrapper of rten_hip_model: the C++ plan executor (include/rten_hip_graph.hpp) behind the C ABI -- ONNX bytes in, every value resident in
    HBM, committed launch plan, `chains` independent sub-batch chains, hipGraph replay.  What a Rust `HipSubgraph` operator would own
    (INTEGRATION.md 2.5); `bench.py --via-executor` times it."""

    def __init__(self, ctx: Context, onnx_bytes: bytes, plan_json: str | None = None, chains: int = 1, receive_weights: bool = False):
        """`receive_weights`: this process gets the weight arena by broadcast (rank != 0 of a sharded job): large initializers are not uploaded;
        fill `weight_arena()` before `prepare()`."""
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        self._onnx = onnx_bytes
        rc = self.lib.rten_hip_model_load_ex(ctx.h, onnx_bytes, len(onnx_bytes), plan_json.encode() if plan_json else None, chains,
                                             MODEL_RECEIVE_WEIGHTS if receive_weights else 0, C.byref(h))
        if rc != OK:
            raise HipError(rc, self.lib.rten_hip_model_load_error().decode(errors="replace") or "rten_hip_model_load_ex failed")
        self._finish_init(h)

    @classmethod
    def _from_handle(cls, ctx, h, origin):
        m = cls.__new__(cls)
        m.ctx, m.lib, m._onnx, m._origin = ctx, ctx.lib, None, origin  # (the origin is kept alive: a replica shares its constants)
        m._finish_init(h)
        return m

    def clone(self, ctx: Context) -> "Model":
        """Another replica of this model on `ctx` (own stream, buffers and hipGraphs; THIS model's constants and prepacked weights): a "lane".
        Close replicas before their origin."""
        h = C.c_void_p()
        rc = self.lib.rten_hip_model_clone(self.h, ctx.h, C.byref(h))
        if rc != OK:
            raise HipError(rc, self.lib.rten_hip_model_load_error().decode(errors="replace") or "rten_hip_model_clone failed")
        return Model._from_handle(ctx, h, self)

    def _finish_init(self, h):
        self.h = h
        ni, no, ns, npl = _I32(), _I32(), _I32(), _I32()
        self._check(self.lib.rten_hip_model_info(h, C.byref(ni), C.byref(no), C.byref(ns), C.byref(npl)))
        self.inputs = [self.lib.rten_hip_model_input_name(h, i).decode() for i in range(ni.value)]
        self.outputs = [self.lib.rten_hip_model_output_name(h, i).decode() for i in range(no.value)]
        self.num_steps = ns.value
        self.input_ptrs = {}

    def _check(self, rc):
        if rc != OK:
            raise HipError(rc, self.lib.rten_hip_model_last_error(self.h).decode(errors="replace"))

    def bind_input(self, name: str, shape) -> int:
        """Declare an input's full-batch shape; returns the device pointer to write the input to."""
        p = C.c_void_p()
        sh = (C.c_int64 * len(shape))(*shape)
        self._check(self.lib.rten_hip_model_bind_input(self.h, self.inputs.index(name), sh, len(shape), C.byref(p)))
        self.input_ptrs[name] = p.value
        return p.value

    def weight_arena(self):
        """(device pointer, bytes) of the one allocation that holds every constant of the model: the unit of the one-time weight broadcast."""
        p, n = C.c_void_p(), _SZ()
        self._check(self.lib.rten_hip_model_weight_arena(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_plan(self, plan_json: str):
        """Replace the launch plan's step tables; call prepare() again to apply and re-capture."""
        self._check(self.lib.rten_hip_model_set_plan(self.h, plan_json.encode()))

    def plan_json(self) -> str:
        """The launch plan of the prepared model as plan-file text (keyed by sub-batch size)."""
        need = _SZ()
        self.lib.rten_hip_model_plan_json(self.h, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        self._check(self.lib.rten_hip_model_plan_json(self.h, buf, need.value, C.byref(need)))
        return buf.value.decode()

    def profile_pass(self, steps: int):
        """Instrumented eager pass over every chain (serialised launches, HIP events per launch), merged by kernel:
        [{kernel, launches, ms, flops, bytes}] -- the same rows as Context.profile_report()."""
        import json
        buf, need = C.create_string_buffer(1 << 20), _SZ()
        self._check(self.lib.rten_hip_model_profile(self.h, steps, buf, len(buf), C.byref(need)))
        merged = {}
        for chain in json.loads(buf.value.decode()):
            for r in chain:
                m = merged.setdefault(r["kernel"], dict(r, launches=0, ms=0.0, flops=0.0, bytes=0.0))
                for k in ("launches", "ms", "flops", "bytes"):
                    m[k] += r[k]
        return list(merged.values())

    @property
    def warning(self) -> str:
        """Text of a non-fatal condition of the last successful call (e.g. a plan file that matched no step); empty otherwise."""
        return self.lib.rten_hip_model_last_error(self.h).decode(errors="replace")

    def prepare(self, tune: bool = False):
        self._check(self.lib.rten_hip_model_prepare(self.h, 1 if tune else 0))
        ni, no, ns, npl = _I32(), _I32(), _I32(), _I32()
        self._check(self.lib.rten_hip_model_info(self.h, C.byref(ni), C.byref(no), C.byref(ns), C.byref(npl)))
        self.planned_steps = npl.value

    def run(self, inputs_written_on_caller_stream: bool = False, join: bool = True):
        """`join=False`: the caller's stream is not ordered behind the chains (flags bit 1) -- back-to-back runs then let the chains free-run
        across run boundaries; call sync() before reading the outputs."""
        self._check(self.lib.rten_hip_model_run(self.h, (1 if inputs_written_on_caller_stream else 0) | (0 if join else 2)))

    def sync(self):
        self._check(self.lib.rten_hip_model_sync(self.h))

    def output(self, i: int = 0):
        """(device pointer, shape) of output i after a run."""
        p, nd = C.c_void_p(), _I32()
        sh = (C.c_int64 * 8)()
        self._check(self.lib.rten_hip_model_output(self.h, i, C.byref(p), sh, C.byref(nd)))
        return p.value, tuple(sh[: nd.value])

    _DTYPES = ("float32", "int32", "uint8", "int8")  # RTEN_HIP_DTYPE_* as numpy dtype names

    def input_dtype(self, i: int = 0):
        """numpy dtype name of input i on the device (ONNX int64 inputs are int32 there, as in the reference)."""
        d = _I32()
        self._check(self.lib.rten_hip_model_input_dtype(self.h, i, C.byref(d)))
        return self._DTYPES[d.value]

    def output_dtype(self, i: int = 0):
        """numpy dtype name of output i (after prepare)."""
        d = _I32()
        self._check(self.lib.rten_hip_model_output_dtype(self.h, i, C.byref(d)))
        return self._DTYPES[d.value]

    def close(self):
        """Destroys the model.  Chain 0 runs on the context the model was loaded with: that context must still be alive (close models first)."""
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.lib.rten_hip_model_destroy(self.h)
            # (context already destroyed -- interpreter shutdown order: the model is leaked rather than torn down on a dead context)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
