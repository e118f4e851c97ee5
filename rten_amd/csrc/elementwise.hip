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

enum UnaryOp { U_RELU, U_GELU, U_ERF };

template <int OP>
__device__ __forceinline__ float unary(float x) {
    if constexpr (OP == U_RELU) return vm::relu(x);
    else if constexpr (OP == U_GELU) return vm::gelu(x);
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
