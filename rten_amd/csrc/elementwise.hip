// Element-wise kernels (HBM-bound): Relu, Gelu, Erf, Add, Mul, per-channel bias Add, cast_scale,
// BatchNormalization.  Replaces src/ops/unary_elementwise.rs:399-420,611-613,
// src/ops/binary_elementwise.rs:476-495, src/ops/matmul.rs:734-773 (cast_scale),
// src/ops/norm.rs:194-224 (batch_norm_in_place) and their rten-vecmath inner loops.
//
// Layout: flat grid-stride loops, 16 B per lane per access when the tensor is 16-byte aligned
// (float4 == one global_load_dwordx4 per lane, 1 KiB per wave instruction), grid capped at
// 256 CUs x 8 workgroups.  Every function is an operation-for-operation restatement of the
// reference's scalar formula (vecmath.h), so results are bit-identical.
#include "internal.h"
#include "vecmath.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int EW_THREADS = 256;

inline int ew_blocks(int64_t work_items) {
    int64_t b = (work_items + EW_THREADS - 1) / EW_THREADS;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

enum UnaryOp { U_RELU, U_GELU, U_ERF, U_TANH };

template <int OP>
__device__ __forceinline__ float unary(float x) {
    if constexpr (OP == U_RELU) return vm::relu(x);
    else if constexpr (OP == U_GELU) return vm::gelu(x);
    else if constexpr (OP == U_TANH) return vm::tanh(x);
    else return vm::erf(x);
}

template <int OP>
__global__ __launch_bounds__(EW_THREADS) void unary_kernel(int64_t n, const float *__restrict__ x, float *__restrict__ y,
                                                           int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += stride) {
            f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
            f32x4 r;
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = unary<OP>(v[k]);
            reinterpret_cast<f32x4 *>(y)[i] = r;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) y[i] = unary<OP>(x[i]);
    } else {
        for (int64_t i = tid; i < n; i += stride) y[i] = unary<OP>(x[i]);
    }
}

enum BinaryOp { B_ADD, B_MUL, B_SUB, B_DIV };

template <int OP>
__device__ __forceinline__ float binary(float a, float b) {
    if constexpr (OP == B_ADD) return a + b;
    else if constexpr (OP == B_MUL) return a * b;
    else if constexpr (OP == B_SUB) return a - b;
    else return a / b;
}

template <int OP>
__global__ __launch_bounds__(EW_THREADS) void binary_kernel(int64_t n, const float *__restrict__ a,
                                                            const float *__restrict__ b, int64_t b_len,
                                                            float *__restrict__ y, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) { // same shape, aligned
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += stride) {
            f32x4 va = reinterpret_cast<const f32x4 *>(a)[i];
            f32x4 vb = reinterpret_cast<const f32x4 *>(b)[i];
            f32x4 r;
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = binary<OP>(va[k], vb[k]);
            reinterpret_cast<f32x4 *>(y)[i] = r;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) y[i] = binary<OP>(a[i], b[i]);
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            const float bv = b[b_len == n ? i : i % b_len];
            y[i] = binary<OP>(a[i], bv);
        }
    }
}

// y[n][c][inner] = x + bias[c]
__global__ __launch_bounds__(EW_THREADS) void channel_bias_kernel(int64_t total, int c, int64_t inner,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ bias, float *__restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)((i / inner) % c);
        y[i] = x[i] + bias[ch];
    }
}

__global__ __launch_bounds__(EW_THREADS) void cast_scale_kernel(int64_t n, const int32_t *__restrict__ x,
                                                                const float *__restrict__ scale, int scale_len,
                                                                float *__restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float s = scale[scale_len == 1 ? 0 : (int)(i % scale_len)];
        y[i] = (float)x[i] * s; // `*el as f32 * scale`, matmul.rs:751,761
    }
}

// y = fma(x - mean_c, scale_c / sqrt(var_c + eps), bias_c)  (norm.rs:146 + normalize.rs:112-127)
__global__ __launch_bounds__(EW_THREADS) void batch_norm_kernel(int64_t total, int c, int64_t inner,
                                                                const float *__restrict__ x,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ var, float eps,
                                                                float *__restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ch = (int)((i / inner) % c);
        const float ssr = scale[ch] / sqrtf(var[ch] + eps);
        y[i] = vm::fma(x[i] - mean[ch], ssr, bias[ch]);
    }
}

// out[i, :] = table[ids[i], :]  (Gather axis 0; out-of-range ids are clamped after host validation)
__global__ __launch_bounds__(EW_THREADS) void gather_rows_kernel(int64_t n_ids, int row_len, int table_rows,
                                                                 const float *__restrict__ table,
                                                                 const int32_t *__restrict__ ids, float *__restrict__ out) {
    const int64_t total = n_ids * row_len;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / row_len;
        const int c = (int)(i - r * row_len);
        int id = ids[r];
        if (id < 0) id += table_rows; // ONNX Gather: negative indices count from the end
        id = id < 0 ? 0 : (id >= table_rows ? table_rows - 1 : id);
        out[i] = table[(int64_t)id * row_len + c];
    }
}

// General numpy-style broadcasting (binary_elementwise.rs:58-170: operands expanded with stride 0 on broadcast axes) and
// Transpose / permute (src/ops/layout.rs:669+): one output element per thread-iteration, index decomposed over <= 6 axes.
struct NdArgs {
    int ndim;
    int64_t n;
    int32_t shape[6];
    int64_t a_stride[6], b_stride[6];
};

template <int OP>
__global__ __launch_bounds__(EW_THREADS) void binary_bcast_kernel(const NdArgs p, const float *__restrict__ a, const float *__restrict__ b,
                                                                  float *__restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
        int64_t r = i, ao = 0, bo = 0;
        for (int d = p.ndim - 1; d >= 0; d--) {
            const int64_t q = r / p.shape[d], c = r - q * p.shape[d];
            ao += c * p.a_stride[d];
            bo += c * p.b_stride[d];
            r = q;
        }
        y[i] = binary<OP>(a[ao], b[bo]);
    }
}

__global__ __launch_bounds__(EW_THREADS) void permute_kernel(const NdArgs p, const uint32_t *__restrict__ x, uint32_t *__restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
        int64_t r = i, xo = 0;
        for (int d = p.ndim - 1; d >= 0; d--) {
            const int64_t q = r / p.shape[d], c = r - q * p.shape[d];
            xo += c * p.a_stride[d]; // stride of output axis d in the input
            r = q;
        }
        y[i] = x[xo];
    }
}

inline bool al16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

template <int OP>
int32_t run_unary(rten_hip_ctx *ctx, int64_t n, const float *x, float *y, const char *name) {
    RTEN_CHECK_CTX(ctx);
    if (n < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (n == 0) return RTEN_HIP_OK;
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    const int vec = al16(x) && al16(y);
    ProfScope ps(ctx, name, 0.0, 8.0 * n);
    hipLaunchKernelGGL((unary_kernel<OP>), dim3(ew_blocks(vec ? n / 4 : n)), dim3(EW_THREADS), 0, ctx->stream, n, x, y, vec);
    RTEN_LAUNCH_CHECK(ctx, name);
    return RTEN_HIP_OK;
}

template <int OP>
int32_t run_binary(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len, float *y,
                   const char *name) {
    RTEN_CHECK_CTX(ctx);
    if (n < 0 || b_len < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (n == 0) return RTEN_HIP_OK;
    if (!a || !b || !y || b_len == 0 || n % b_len != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "Cannot broadcast inputs");
    const int vec = b_len == n && al16(a) && al16(b) && al16(y);
    ProfScope ps(ctx, name, 0.0, 12.0 * n);
    hipLaunchKernelGGL((binary_kernel<OP>), dim3(ew_blocks(vec ? n / 4 : n)), dim3(EW_THREADS), 0, ctx->stream, n, a, b,
                       b_len, y, vec);
    RTEN_LAUNCH_CHECK(ctx, name);
    return RTEN_HIP_OK;
}

} // namespace

RTEN_EXPORT int32_t rten_hip_relu_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y) {
    return run_unary<U_RELU>(ctx, n, x, y, "relu_f32");
}
RTEN_EXPORT int32_t rten_hip_gelu_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y) {
    return run_unary<U_GELU>(ctx, n, x, y, "gelu_f32");
}
RTEN_EXPORT int32_t rten_hip_erf_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y) {
    return run_unary<U_ERF>(ctx, n, x, y, "erf_f32");
}
RTEN_EXPORT int32_t rten_hip_tanh_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y) {
    return run_unary<U_TANH>(ctx, n, x, y, "tanh_f32");
}
RTEN_EXPORT int32_t rten_hip_add_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len,
                                     float *y) {
    return run_binary<B_ADD>(ctx, n, a, b, b_len, y, "add_f32");
}
RTEN_EXPORT int32_t rten_hip_mul_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len,
                                     float *y) {
    return run_binary<B_MUL>(ctx, n, a, b, b_len, y, "mul_f32");
}
RTEN_EXPORT int32_t rten_hip_sub_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len,
                                     float *y) {
    return run_binary<B_SUB>(ctx, n, a, b, b_len, y, "sub_f32");
}
RTEN_EXPORT int32_t rten_hip_div_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len,
                                     float *y) {
    return run_binary<B_DIV>(ctx, n, a, b, b_len, y, "div_f32");
}

// op: 0 add, 1 mul, 2 sub, 3 div.  out_shape[ndim]; a_strides / b_strides in elements, 0 on broadcast axes.
RTEN_EXPORT int32_t rten_hip_binary_broadcast_f32(rten_hip_ctx *ctx, int32_t op, int32_t ndim, const int64_t *out_shape, const int64_t *a_strides,
                                                  const int64_t *b_strides, const float *a, const float *b, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (op < 0 || op > 3 || ndim < 0 || ndim > 6 || (ndim && (!out_shape || !a_strides || !b_strides)))
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "binary_broadcast: op in 0..3, at most 6 dims");
    NdArgs p = {};
    p.ndim = ndim;
    p.n = 1;
    for (int d = 0; d < ndim; d++) {
        if (out_shape[d] < 0 || out_shape[d] > 0x7fffffff) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "binary_broadcast: bad dimension");
        p.shape[d] = (int32_t)out_shape[d]; p.a_stride[d] = a_strides[d]; p.b_stride[d] = b_strides[d];
        p.n *= out_shape[d];
    }
    if (p.n == 0) return RTEN_HIP_OK;
    if (!a || !b || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "binary_broadcast_f32", 0.0, 12.0 * p.n);
    const dim3 grid(ew_blocks(p.n)), block(EW_THREADS);
    if (op == 0) hipLaunchKernelGGL((binary_bcast_kernel<B_ADD>), grid, block, 0, ctx->stream, p, a, b, y);
    else if (op == 1) hipLaunchKernelGGL((binary_bcast_kernel<B_MUL>), grid, block, 0, ctx->stream, p, a, b, y);
    else if (op == 2) hipLaunchKernelGGL((binary_bcast_kernel<B_SUB>), grid, block, 0, ctx->stream, p, a, b, y);
    else hipLaunchKernelGGL((binary_bcast_kernel<B_DIV>), grid, block, 0, ctx->stream, p, a, b, y);
    RTEN_LAUNCH_CHECK(ctx, "binary_broadcast_f32");
    return RTEN_HIP_OK;
}

// y = transpose(x, perm) for 4-byte elements (f32 / i32): y.shape[d] = x_shape[perm[d]].
RTEN_EXPORT int32_t rten_hip_transpose_b32(rten_hip_ctx *ctx, int32_t ndim, const int64_t *x_shape, const int32_t *perm, const void *x, void *y) {
    RTEN_CHECK_CTX(ctx);
    if (ndim < 0 || ndim > 6 || (ndim && (!x_shape || !perm))) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "transpose: at most 6 dims");
    int64_t xstride[6], acc = 1;
    unsigned seen = 0;
    for (int d = ndim - 1; d >= 0; d--) { xstride[d] = acc; acc *= x_shape[d]; }
    NdArgs p = {};
    p.ndim = ndim;
    p.n = acc;
    for (int d = 0; d < ndim; d++) {
        if (perm[d] < 0 || perm[d] >= ndim || (seen >> perm[d]) & 1u) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Permutation is invalid");
        seen |= 1u << perm[d];
        if (x_shape[perm[d]] < 0 || x_shape[perm[d]] > 0x7fffffff) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "transpose: bad dimension");
        p.shape[d] = (int32_t)x_shape[perm[d]];
        p.a_stride[d] = xstride[perm[d]];
    }
    if (p.n == 0) return RTEN_HIP_OK;
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "transpose_b32", 0.0, 8.0 * p.n);
    hipLaunchKernelGGL(permute_kernel, dim3(ew_blocks(p.n)), dim3(EW_THREADS), 0, ctx->stream, p, (const uint32_t *)x, (uint32_t *)y);
    RTEN_LAUNCH_CHECK(ctx, "transpose_b32");
    return RTEN_HIP_OK;
}

// y (contiguous, `shape`) = x viewed through `x_strides` (elements; 0 = broadcast axis, a sum of strides = a diagonal).
// The device form of TensorView::to_tensor / to_contiguous / expand_to on the views Einsum builds
// (src/ops/einsum.rs:124-162,255-257,320-329,534-535; src/ops/layout.rs expand_to).
RTEN_EXPORT int32_t rten_hip_copy_strided_b32(rten_hip_ctx *ctx, int32_t ndim, const int64_t *shape, const int64_t *x_strides, const void *x, void *y) {
    RTEN_CHECK_CTX(ctx);
    if (ndim < 0 || ndim > 6 || (ndim && (!shape || !x_strides))) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "copy_strided: at most 6 dims");
    NdArgs p = {};
    p.ndim = ndim;
    p.n = 1;
    for (int d = 0; d < ndim; d++) {
        if (shape[d] < 0 || shape[d] > 0x7fffffff || x_strides[d] < 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "copy_strided: bad dimension");
        p.shape[d] = (int32_t)shape[d];
        p.a_stride[d] = x_strides[d];
        p.n *= shape[d];
    }
    if (p.n == 0) return RTEN_HIP_OK;
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "copy_strided_b32", 0.0, 8.0 * p.n);
    hipLaunchKernelGGL(permute_kernel, dim3(ew_blocks(p.n)), dim3(EW_THREADS), 0, ctx->stream, p, (const uint32_t *)x, (uint32_t *)y);
    RTEN_LAUNCH_CHECK(ctx, "copy_strided_b32");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_add_channel_bias_f32(rten_hip_ctx *ctx, int32_t n, int32_t c, int64_t inner,
                                                  const float *x, const float *bias, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (n < 0 || c <= 0 || inner < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    const int64_t total = (int64_t)n * c * inner;
    if (total == 0) return RTEN_HIP_OK;
    if (!x || !bias || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "add_channel_bias_f32", 0.0, 8.0 * total);
    hipLaunchKernelGGL(channel_bias_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ctx->stream, total, c, inner, x,
                       bias, y);
    RTEN_LAUNCH_CHECK(ctx, "channel_bias_kernel");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_cast_scale(rten_hip_ctx *ctx, int64_t n, const int32_t *x, const float *scale,
                                        int32_t scale_len, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (n < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (n == 0) return RTEN_HIP_OK;
    if (!x || !scale || !y || scale_len <= 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (scale_len != 1 && n % scale_len != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "Scale length does not match tensor columns");
    ProfScope ps(ctx, "cast_scale", 0.0, 8.0 * n);
    hipLaunchKernelGGL(cast_scale_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, ctx->stream, n, x, scale, scale_len, y);
    RTEN_LAUNCH_CHECK(ctx, "cast_scale_kernel");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_batch_norm_f32(rten_hip_ctx *ctx, int32_t n, int32_t c, int64_t inner, const float *x,
                                            const float *scale, const float *bias, const float *mean,
                                            const float *var, float epsilon, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (n < 0 || c <= 0 || inner < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    const int64_t total = (int64_t)n * c * inner;
    if (total == 0) return RTEN_HIP_OK;
    if (!x || !scale || !bias || !mean || !var || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "batch_norm_f32", 0.0, 8.0 * total);
    hipLaunchKernelGGL(batch_norm_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ctx->stream, total, c, inner, x,
                       scale, bias, mean, var, epsilon, y);
    RTEN_LAUNCH_CHECK(ctx, "batch_norm_kernel");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_gather_rows_f32(rten_hip_ctx *ctx, int64_t n_ids, int32_t row_len, int32_t table_rows,
                                             const float *table, const int32_t *ids, float *out) {
    RTEN_CHECK_CTX(ctx);
    if (n_ids < 0 || row_len < 0 || table_rows <= 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (n_ids == 0 || row_len == 0) return RTEN_HIP_OK;
    if (!table || !ids || !out) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "gather_rows_f32", 0.0, 8.0 * n_ids * row_len);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_blocks(n_ids * row_len)), dim3(EW_THREADS), 0, ctx->stream, n_ids,
                       row_len, table_rows, table, ids, out);
    RTEN_LAUNCH_CHECK(ctx, "gather_rows_kernel");
    return RTEN_HIP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Layout / logic operators of exporter-written graphs (the executor's "glue" either side of the hot-path operators): Cast, Not / And / Or / Xor,
// Equal / Less / Greater (...OrEqual), Where, integer Add / Sub / Mul / Div, over operands viewed through element strides (0 = broadcast axis), at most
// 6 dims, one output element per thread-iteration.  Replaces src/ops/convert.rs:18-60 (`as` casts: float -> int saturates, NaN -> 0; int -> narrower
// int wraps), src/ops/binary_elementwise.rs:546-598,733-786 (booleans are i32 0 / 1 in the reference: onnx_loader.rs:332-339), :1189-1247 (where_op:
// cond != 0), unary_elementwise.rs:563-565 (Not).  None of these is on a model's critical path (mask and index preparation): no vector forms.
namespace {
struct GenArgs {
    int32_t ndim, op, a_dt, b_dt, y_dt;
    int32_t shape[6];
    int64_t sa[6], sb[6], sc[6];
    int64_t n;
};
struct GenVal { float f; int i; };
__device__ __forceinline__ GenVal gen_load(const void *p, int dt, int64_t off) {
    GenVal v;
    v.f = 0.f; v.i = 0;
    if (dt == RTEN_HIP_DT_F32) v.f = ((const float *)p)[off];
    else if (dt == RTEN_HIP_DT_I32) v.i = ((const int32_t *)p)[off];
    else if (dt == RTEN_HIP_DT_U8) v.i = ((const uint8_t *)p)[off];
    else v.i = ((const int8_t *)p)[off];
    return v;
}
// Rust `f32 as iN / uN`: truncation toward zero, saturating at the type's bounds, NaN -> 0
__device__ __forceinline__ int gen_f2i(float f, float lo, float hi, int ilo, int ihi) {
    if (f != f) return 0;
    if (f <= lo) return ilo;
    if (f >= hi) return ihi;
    return (int)f;
}
__global__ __launch_bounds__(EW_THREADS) void generic_nd_kernel(const GenArgs p, const void *__restrict__ a, const void *__restrict__ b, const void *__restrict__ c,
                                                                void *__restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
        int64_t r = i, ao = 0, bo = 0, co = 0;
        for (int d = p.ndim - 1; d >= 0; d--) {
            const int64_t q = r / p.shape[d], k = r - q * p.shape[d];
            ao += k * p.sa[d]; bo += k * p.sb[d]; co += k * p.sc[d];
            r = q;
        }
        const GenVal va = gen_load(a, p.a_dt, ao);
        GenVal out;
        out.f = 0.f; out.i = 0;
        bool out_f = false;
        if (p.op == RTEN_HIP_EW_CAST) {
            const bool af = p.a_dt == RTEN_HIP_DT_F32;
            if (p.y_dt == RTEN_HIP_DT_F32) { out.f = af ? va.f : (float)va.i; out_f = true; }
            else if (p.y_dt == RTEN_HIP_DT_I32) out.i = af ? gen_f2i(va.f, -2147483648.f, 2147483648.f, (int)0x80000000, 0x7fffffff) : va.i;
            else if (p.y_dt == RTEN_HIP_DT_U8) out.i = af ? gen_f2i(va.f, 0.f, 255.f, 0, 255) : (va.i & 0xff);
            else out.i = af ? gen_f2i(va.f, -128.f, 127.f, -128, 127) : (int)(int8_t)(va.i & 0xff);
        } else if (p.op == RTEN_HIP_EW_NOT) {
            out.i = va.i == 0;
        } else if (p.op == RTEN_HIP_EW_WHERE) {
            out.i = va.i != 0 ? ((const int32_t *)b)[bo] : ((const int32_t *)c)[co]; // (4-byte payloads moved as raw words)
        } else {
            const GenVal vb = gen_load(b, p.b_dt, bo);
            const bool fl = p.a_dt == RTEN_HIP_DT_F32;
            switch (p.op) {
            case RTEN_HIP_EW_AND: out.i = (va.i != 0) && (vb.i != 0); break;
            case RTEN_HIP_EW_OR: out.i = (va.i != 0) || (vb.i != 0); break;
            case RTEN_HIP_EW_XOR: out.i = (va.i != 0) != (vb.i != 0); break;
            case RTEN_HIP_EW_EQUAL: out.i = fl ? va.f == vb.f : va.i == vb.i; break;
            case RTEN_HIP_EW_LESS: out.i = fl ? va.f < vb.f : va.i < vb.i; break;
            case RTEN_HIP_EW_LESS_EQ: out.i = fl ? va.f <= vb.f : va.i <= vb.i; break;
            case RTEN_HIP_EW_GREATER: out.i = fl ? va.f > vb.f : va.i > vb.i; break;
            case RTEN_HIP_EW_GREATER_EQ: out.i = fl ? va.f >= vb.f : va.i >= vb.i; break;
            case RTEN_HIP_EW_IADD: out.i = (int)((unsigned)va.i + (unsigned)vb.i); break;
            case RTEN_HIP_EW_ISUB: out.i = (int)((unsigned)va.i - (unsigned)vb.i); break;
            case RTEN_HIP_EW_IMUL: out.i = (int)((unsigned)va.i * (unsigned)vb.i); break;
            default: out.i = vb.i == 0 ? 0 : (va.i == (int)0x80000000 && vb.i == -1 ? va.i : va.i / vb.i); break; // IDIV (the host refuses a constant zero divisor)
            }
        }
        if (p.y_dt == RTEN_HIP_DT_F32) ((float *)y)[i] = out_f ? out.f : __int_as_float(out.i);
        else if (p.y_dt == RTEN_HIP_DT_I32) ((int32_t *)y)[i] = out.i;
        else ((uint8_t *)y)[i] = (uint8_t)out.i;
    }
}

// y[o][j][k] = data[o][ids[j]][k]: Gather along any axis of 4-byte elements (src/ops/gather.rs:21-110; negative indices count from the end)
__global__ __launch_bounds__(EW_THREADS) void gather_axis_kernel(int64_t outer, int64_t axis_len, int64_t inner, int64_t n_ids, const uint32_t *__restrict__ data,
                                                                 const int32_t *__restrict__ ids, uint32_t *__restrict__ y) {
    const int64_t total = outer * n_ids * inner, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t k = i % inner, j = (i / inner) % n_ids, o = i / (inner * n_ids);
        int64_t id = ids[j];
        if (id < 0) id += axis_len;
        id = id < 0 ? 0 : (id >= axis_len ? axis_len - 1 : id);
        y[i] = data[(o * axis_len + id) * inner + k];
    }
}

// dst[r][0 .. row) = src[r][0 .. row) with independent pitches (elements of 4 bytes): the pieces of a Concat (src/ops/concat.rs:108)
__global__ __launch_bounds__(EW_THREADS) void copy_rows_kernel(int64_t rows, int64_t row, const uint32_t *__restrict__ src, int64_t src_pitch, uint32_t *__restrict__ dst,
                                                               int64_t dst_pitch) {
    const int64_t total = rows * row, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / row, k = i - r * row;
        dst[r * dst_pitch + k] = src[r * src_pitch + k];
    }
}
} // namespace

RTEN_EXPORT int32_t rten_hip_elementwise_nd(rten_hip_ctx *ctx, int32_t op, int32_t ndim, const int64_t *shape, const void *a, int32_t a_dtype, const int64_t *a_strides,
                                            const void *b, int32_t b_dtype, const int64_t *b_strides, const void *c, const int64_t *c_strides, void *y, int32_t y_dtype) {
    RTEN_CHECK_CTX(ctx);
    if (op < RTEN_HIP_EW_CAST || op > RTEN_HIP_EW_IDIV || ndim < 0 || ndim > 6 || (ndim && (!shape || !a_strides)))
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "elementwise_nd: unknown op / more than 6 dims");
    const bool unary = op == RTEN_HIP_EW_CAST || op == RTEN_HIP_EW_NOT, where = op == RTEN_HIP_EW_WHERE;
    auto dt_ok = [](int32_t d) { return d >= RTEN_HIP_DT_F32 && d <= RTEN_HIP_DT_I8; };
    if (!dt_ok(a_dtype) || !dt_ok(y_dtype) || (!unary && !where && !dt_ok(b_dtype))) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "elementwise_nd: unknown element type");
    if (op == RTEN_HIP_EW_CAST) {
        // every pair of the reference's Cast except same-type (a copy: rten_hip_copy_strided_b32 / memcpy)
    } else if (where) {
        if (a_dtype != RTEN_HIP_DT_I32 || (y_dtype != RTEN_HIP_DT_F32 && y_dtype != RTEN_HIP_DT_I32)) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "elementwise_nd: Where takes an int32 condition and 4-byte operands");
    } else if (op >= RTEN_HIP_EW_EQUAL && op <= RTEN_HIP_EW_GREATER_EQ) {
        if ((a_dtype != RTEN_HIP_DT_F32 && a_dtype != RTEN_HIP_DT_I32) || b_dtype != a_dtype || y_dtype != RTEN_HIP_DT_I32) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "elementwise_nd: comparisons take two float32 or two int32 operands and give int32");
    } else if (a_dtype != RTEN_HIP_DT_I32 || y_dtype != RTEN_HIP_DT_I32 || (!unary && b_dtype != RTEN_HIP_DT_I32)) {
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "elementwise_nd: logical and integer operators take int32 operands");
    }
    GenArgs p = {};
    p.ndim = ndim; p.op = op; p.a_dt = a_dtype; p.b_dt = b_dtype; p.y_dt = y_dtype;
    p.n = 1;
    for (int d = 0; d < ndim; d++) {
        if (shape[d] < 0 || shape[d] > 0x7fffffff) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "elementwise_nd: bad dimension");
        p.shape[d] = (int32_t)shape[d];
        p.sa[d] = a_strides[d];
        p.sb[d] = (!unary && b_strides) ? b_strides[d] : 0;
        p.sc[d] = (where && c_strides) ? c_strides[d] : 0;
        p.n *= shape[d];
    }
    if (p.n == 0) return RTEN_HIP_OK;
    if (!a || !y || (!unary && !b) || (where && !c) || (!unary && ndim && !b_strides) || (where && ndim && !c_strides)) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "elementwise_nd", 0.0, 8.0 * p.n);
    hipLaunchKernelGGL(generic_nd_kernel, dim3(ew_blocks(p.n)), dim3(EW_THREADS), 0, ctx->stream, p, a, b, c, y);
    RTEN_LAUNCH_CHECK(ctx, "generic_nd_kernel");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_gather_axis_b32(rten_hip_ctx *ctx, int64_t outer, int64_t axis_len, int64_t inner, int64_t n_ids, const void *data, const int32_t *ids, void *y) {
    RTEN_CHECK_CTX(ctx);
    if (outer < 0 || axis_len <= 0 || inner < 0 || n_ids < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    const int64_t total = outer * n_ids * inner;
    if (total == 0) return RTEN_HIP_OK;
    if (!data || !ids || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "gather_axis_b32", 0.0, 8.0 * total);
    hipLaunchKernelGGL(gather_axis_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, ctx->stream, outer, axis_len, inner, n_ids, (const uint32_t *)data, ids, (uint32_t *)y);
    RTEN_LAUNCH_CHECK(ctx, "gather_axis_kernel");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_copy_rows_b32(rten_hip_ctx *ctx, int64_t rows, int64_t row_elems, const void *src, int64_t src_pitch, void *dst, int64_t dst_pitch) {
    RTEN_CHECK_CTX(ctx);
    if (rows < 0 || row_elems < 0 || src_pitch < 0 || dst_pitch < row_elems) return RTEN_HIP_ERR_INVALID_VALUE;
    if (rows * row_elems == 0) return RTEN_HIP_OK;
    if (!src || !dst) return RTEN_HIP_ERR_INVALID_VALUE;
    ProfScope ps(ctx, "copy_rows_b32", 0.0, 8.0 * rows * row_elems);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_blocks(rows * row_elems)), dim3(EW_THREADS), 0, ctx->stream, rows, row_elems, (const uint32_t *)src, src_pitch, (uint32_t *)dst, dst_pitch);
    RTEN_LAUNCH_CHECK(ctx, "copy_rows_kernel");
    return RTEN_HIP_OK;
}

// 1 while a graph capture is active on the context (host code that would upload from host memory must not do so then: the copy would be recorded with the
// host pointer and re-read at every replay)
RTEN_EXPORT int32_t rten_hip_capture_active(rten_hip_ctx *ctx) {
    if (!ctx) return 0;
    std::lock_guard<std::recursive_mutex> g(ctx->mu);
    return ctx->capturing ? 1 : 0;
}
