// DynamicQuantizeLinear on device.  Replaces dynamic_quantize_linear (src/ops/quantize.rs:352-436),
// vecmath::MinMax (rten-vecmath/src/min_max.rs:20-44) and vecmath::Quantize (quantize.rs:39-79).
//
// Two HBM sweeps like the reference (min/max, then quantize); the scalar scale / zero-point algebra
// is evaluated on the device with the same individually rounded IEEE operations
// (range/255, min/scale, clamp, round-ties-even), so the u8 codes, scale and zero point are
// bit-identical.  The scale and zero point stay in device memory for the following
// ConvIntegerToFloat / MatMulIntegerToFloat epilogue -- no host round trip.
#include "quantize.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
using namespace dql;

namespace {

__global__ __launch_bounds__(256) void minmax_kernel(int64_t n, const float *__restrict__ x, float *ws, int vec) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // fminf/fmaxf drop NaNs like the reference's SIMD min(x, acc) / max(x, acc) (min_max.rs:27-30)
    if (vec) {
        const int64_t n4 = n >> 2;
        int64_t i = tid;
        for (; i + 3 * stride < n4; i += 4 * stride) { // four 16-byte loads in flight per lane
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = reinterpret_cast<const f32x4 *>(x)[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int k = 0; k < 4; k++) { mn = fminf(v[u][k], mn); mx = fmaxf(v[u][k], mx); }
        }
        for (; i < n4; i += stride) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
#pragma unroll
            for (int k = 0; k < 4; k++) { mn = fminf(v[k], mn); mx = fmaxf(v[k], mx); }
        }
        for (int64_t j = (n4 << 2) + tid; j < n; j += stride) { mn = fminf(x[j], mn); mx = fmaxf(x[j], mx); }
    } else {
        for (int64_t i = tid; i < n; i += stride) { mn = fminf(x[i], mn); mx = fmaxf(x[i], mx); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    __shared__ float smn[4], smx[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { smn[wave] = mn; smx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); }
        ws[2 * blockIdx.x] = mn; // per-workgroup partial; the consumer folds them (dql::block_minmax)
        ws[2 * blockIdx.x + 1] = mx;
    }
}

__global__ __launch_bounds__(256) void quantize_kernel(int64_t n, const float *__restrict__ x, const float *ws, int nparts,
                                                       uint8_t *__restrict__ y, float *scale_out, uint8_t *zp_out,
                                                       int vec) {
    float x_min, x_max;
    block_minmax(ws, nparts, x_min, x_max);
    const QParams q = dql_params(x_min, x_max);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid == 0) { *scale_out = q.scale; *zp_out = (uint8_t)q.zp; }
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += stride) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
            unsigned packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) packed |= quant_u8(v[k], q.inv_scale, q.zp) << (8 * k);
            reinterpret_cast<unsigned *>(y)[i] = packed;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) y[i] = (uint8_t)quant_u8(x[i], q.inv_scale, q.zp);
    } else {
        for (int64_t i = tid; i < n; i += stride) y[i] = (uint8_t)quant_u8(x[i], q.inv_scale, q.zp);
    }
}

__global__ void dql_empty_kernel(float *scale_out, uint8_t *zp_out) {
    *scale_out = 1.f; // quantize.rs:385-392
    *zp_out = 0;
}

} // namespace

constexpr int kMaxMinMaxParts = 480; // 480 pairs = 3840 B: inside the 4 KiB scratch header

const float *rten_dql_minmax(rten_hip_ctx *ctx, int64_t n, const float *x, int *nparts) {
    float *ws = (float *)rten_scratch(ctx, 4096);
    if (!ws) return nullptr;
    const int vec_in = (((uintptr_t)x & 15u) == 0);
    const int64_t items = vec_in ? n / 4 : n;
    const int64_t want = (items + 256 * 8 - 1) / (256 * 8); // about 8 items per lane
    int blocks = (int)(want > kMaxMinMaxParts ? kMaxMinMaxParts : want);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, x, ws, vec_in);
    *nparts = blocks;
    return ws;
}

RTEN_EXPORT int32_t rten_hip_dynamic_quantize_linear(rten_hip_ctx *ctx, int64_t n, const float *x, uint8_t *y,
                                                     float *scale, uint8_t *zero_point) {
    RTEN_CHECK_CTX(ctx);
    if (n < 0 || !scale || !zero_point) return RTEN_HIP_ERR_INVALID_VALUE;
    if (n == 0) {
        hipLaunchKernelGGL(dql_empty_kernel, dim3(1), dim3(1), 0, ctx->stream, scale, zero_point);
        RTEN_LAUNCH_CHECK(ctx, "dql_empty_kernel");
        return RTEN_HIP_OK;
    }
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    const int vec_in = (((uintptr_t)x & 15u) == 0);
    const int vec_q = vec_in && (((uintptr_t)y & 3u) == 0);
    int64_t items = vec_in ? n / 4 : n;
    int blocks = (int)((items + 255) / 256 > 2048 ? 2048 : (items + 255) / 256);
    if (blocks < 1) blocks = 1;
    ProfScope ps(ctx, "dynamic_quantize_linear", 0.0, 9.0 * n);
    int nparts = 0;
    const float *ws = rten_dql_minmax(ctx, n, x, &nparts);
    if (!ws) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "dql: scratch allocation failed");
    hipLaunchKernelGGL(quantize_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, x, ws, nparts, y, scale, zero_point, vec_q);
    RTEN_LAUNCH_CHECK(ctx, "dynamic_quantize_linear");
    return RTEN_HIP_OK;
}
