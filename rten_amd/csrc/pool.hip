// MaxPool / AveragePool over NCHW f32 (HBM-bound).  Replaces pool_impl / max_pool / average_pool,
// src/ops/pooling.rs:174-389,392-417,581-600: the window is folded ky-major then kx, padding cells are
// skipped (max starts from -inf, f32::max semantics; average divides by the number of in-image taps
// unless count_include_pad), so results are bit-identical to the reference.
//
// One thread per output element, adjacent lanes -> adjacent output columns, so a wavefront reads
// contiguous (strided by `stride_w`) runs of each input row.
#include "internal.h"

namespace {

template <bool IS_MAX>
__global__ __launch_bounds__(256) void pool2d_kernel(const rten_hip_pool2d_desc d, const float *__restrict__ x,
                                                     float *__restrict__ y) {
    const long long total = (long long)d.n * d.c * d.out_h * d.out_w;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ox = (int)(i % d.out_w);
        const long long r = i / d.out_w;
        const int oy = (int)(r % d.out_h);
        const long long nc = r / d.out_h;
        const float *in = x + nc * (long long)d.h * d.w;
        float acc = IS_MAX ? -__builtin_inff() : 0.f;
        int cnt = 0;
        for (int ky = 0; ky < d.kh; ky++) {
            const int iy = oy * d.stride_h + ky - d.pads[0];
            if ((unsigned)iy >= (unsigned)d.h) continue;
            for (int kx = 0; kx < d.kw; kx++) {
                const int ix = ox * d.stride_w + kx - d.pads[1];
                if ((unsigned)ix >= (unsigned)d.w) continue;
                const float v = in[(long long)iy * d.w + ix];
                acc = IS_MAX ? fmaxf(acc, v) : acc + v;
                cnt++;
            }
        }
        if (!IS_MAX) acc = d.count_include_pad ? acc / (float)(d.kh * d.kw) : acc / (float)cnt;
        y[i] = acc;
    }
}

template <bool IS_MAX>
int32_t run_pool(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *d, const float *x, float *y, const char *name) {
    RTEN_CHECK_CTX(ctx);
    if (!d) return RTEN_HIP_ERR_INVALID_VALUE;
    if (d->n < 0 || d->c < 0 || d->h <= 0 || d->w <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride_h <= 0 ||
        d->stride_w <= 0 || d->out_h < 0 || d->out_w < 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "pool: invalid geometry");
    const long long total = (long long)d->n * d->c * d->out_h * d->out_w;
    if (total == 0) return RTEN_HIP_OK;
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps(ctx, name, 0.0, 4.0 * ((double)d->n * d->c * d->h * d->w + (double)total));
    hipLaunchKernelGGL((pool2d_kernel<IS_MAX>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, *d, x, y);
    RTEN_LAUNCH_CHECK(ctx, name);
    return RTEN_HIP_OK;
}

} // namespace

RTEN_EXPORT int32_t rten_hip_max_pool2d_f32(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x,
                                            float *y) {
    return run_pool<true>(ctx, desc, x, y, "max_pool2d_f32");
}

RTEN_EXPORT int32_t rten_hip_average_pool2d_f32(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x,
                                                float *y) {
    return run_pool<false>(ctx, desc, x, y, "average_pool2d_f32");
}
