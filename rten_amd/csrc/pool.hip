// MaxPool / AveragePool over NCHW f32 (HBM-bound).  Replaces pool_impl / max_pool / average_pool,
// src/ops/pooling.rs:174-389,392-417,581-600: the window is folded ky-major then kx, padding cells are
// skipped (max starts from -inf, f32::max semantics; average divides by the number of in-image taps
// unless count_include_pad), so results are bit-identical to the reference.
//
// One thread per output element, adjacent lanes -> adjacent output columns, so a wavefront reads
// contiguous (strided by `stride_w`) runs of each input row.
#include "internal.h"
#include "quantize.h"

namespace {

// Optional producer-side statistics (rten_hip_max_pool2d_f32_stats): the min / max of the values this wave stored, folded into the
// 256 + 256 ordered-uint slots a consuming DynamicQuantizeLinear reads instead of sweeping the tensor (fminf / fmaxf drop NaNs like
// the reference's sweep, rten-vecmath/src/min_max.rs:27-30).  Every lane of the wave calls this; lanes without an output pass +-inf.
__device__ __forceinline__ void wave_stats(unsigned *stats, float mn, float mx) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0 && mn <= mx) {
        const unsigned slot = ((blockIdx.x * gridDim.y + blockIdx.y) * 4u + (threadIdx.x >> 6)) % (unsigned)dql::kStatSlots;
        atomicMin(&stats[slot], dql::f2ord(mn));
        atomicMax(&stats[dql::kStatSlots + slot], dql::f2ord(mx));
    }
}

// One thread per output element; grid.x = (image, channel) plane, grid.y = 256-output slabs of the plane, so the
// index arithmetic stays 32-bit.  With a compile-time window (KH, KW > 0) every tap's load is issued unconditionally
// from a clamped address before the first compare/add, i.e. KH*KW loads in flight per lane instead of one; taps are
// folded in (ky, kx) order either way (the order the header comment cites).
template <bool IS_MAX, int KH, int KW>
__global__ __launch_bounds__(256) void pool2d_kernel(const rten_hip_pool2d_desc d, const float *__restrict__ x,
                                                     float *__restrict__ y, unsigned *__restrict__ stats) {
    const int plane = d.out_h * d.out_w;
    const int o = blockIdx.y * 256 + threadIdx.x;
    float st_mn = __builtin_inff(), st_mx = -__builtin_inff();
    if (o < plane) {
    const int oy = o / d.out_w, ox = o - oy * d.out_w;
    const float *in = x + (long long)blockIdx.x * d.h * d.w;
    float acc = IS_MAX ? -__builtin_inff() : 0.f;
    int cnt = 0;
    const int y0 = oy * d.stride_h - d.pads[0], x0 = ox * d.stride_w - d.pads[1];
    if constexpr (KH > 0) {
        float raw[KH * KW];
        bool ok[KH * KW];
#pragma unroll
        for (int ky = 0; ky < KH; ky++)
#pragma unroll
            for (int kx = 0; kx < KW; kx++) {
                const int iy = y0 + ky, ix = x0 + kx;
                const bool in_range = (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
                ok[ky * KW + kx] = in_range;
                raw[ky * KW + kx] = in[in_range ? iy * d.w + ix : 0];
            }
#pragma unroll
        for (int t = 0; t < KH * KW; t++)
            if (ok[t]) {
                acc = IS_MAX ? fmaxf(acc, raw[t]) : acc + raw[t];
                cnt++;
            }
    } else {
        for (int ky = 0; ky < d.kh; ky++) {
            const int iy = y0 + ky;
            if ((unsigned)iy >= (unsigned)d.h) continue;
            for (int kx = 0; kx < d.kw; kx++) {
                const int ix = x0 + kx;
                if ((unsigned)ix >= (unsigned)d.w) continue;
                const float v = in[iy * d.w + ix];
                acc = IS_MAX ? fmaxf(acc, v) : acc + v;
                cnt++;
            }
        }
    }
    if (!IS_MAX) acc = d.count_include_pad ? acc / (float)(d.kh * d.kw) : acc / (float)cnt;
    y[(long long)blockIdx.x * plane + o] = acc;
    st_mn = fminf(acc, st_mn); st_mx = fmaxf(acc, st_mx);
    }
    if (stats) wave_stats(stats, st_mn, st_mx);
}

// 3 x 3 window, stride_h SH: four vertically adjacent outputs per thread (lanes walk x: coalesced).  The patch of (4-1)*SH + 3
// rows x 3 columns is loaded once; every output folds its own window in (ky, kx) order from registers.
template <bool IS_MAX, int SH>
__global__ __launch_bounds__(256) void pool3x3_y4_kernel(const rten_hip_pool2d_desc d, const float *__restrict__ x, float *__restrict__ y,
                                                         unsigned *__restrict__ stats) {
    constexpr int NROW = 3 * SH + 3;
    const int hq = (d.out_h + 3) >> 2, items = hq * d.out_w;
    const int q = blockIdx.y * 256 + threadIdx.x;
    float st_mn = __builtin_inff(), st_mx = -__builtin_inff();
    if (q < items) {
    const int yq = q / d.out_w, ox = q - yq * d.out_w, oy0 = yq * 4;
    const float *in = x + (long long)blockIdx.x * d.h * d.w;
    const int y0 = oy0 * SH - d.pads[0], x0 = ox * d.stride_w - d.pads[1];
    float xv[NROW][3];
    bool rok[NROW], cok[3];
#pragma unroll
    for (int kx = 0; kx < 3; kx++) cok[kx] = (unsigned)(x0 + kx) < (unsigned)d.w;
#pragma unroll
    for (int r = 0; r < NROW; r++) {
        const int iy = y0 + r;
        rok[r] = (unsigned)iy < (unsigned)d.h;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) xv[r][kx] = in[(rok[r] && cok[kx]) ? iy * d.w + x0 + kx : 0];
    }
    float *out = y + (long long)blockIdx.x * d.out_h * d.out_w + ox;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (oy0 + j >= d.out_h) break;
        float acc = IS_MAX ? -__builtin_inff() : 0.f;
        int cnt = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
                if (rok[j * SH + ky] && cok[kx]) {
                    acc = IS_MAX ? fmaxf(acc, xv[j * SH + ky][kx]) : acc + xv[j * SH + ky][kx];
                    cnt++;
                }
        if (!IS_MAX) acc = d.count_include_pad ? acc / 9.0f : acc / (float)cnt;
        out[(long long)(oy0 + j) * d.out_w] = acc;
        st_mn = fminf(acc, st_mn); st_mx = fmaxf(acc, st_mx);
    }
    }
    if (stats) wave_stats(stats, st_mn, st_mx);
}


// MaxPool 3 x 3 / stride 2 with one leading padding row and column (ResNet's stem pool, 112 x 112 -> 56 x 56: 25-39 us of every f32 / int8 step) as a
// streaming kernel: a thread owns TWO adjacent output columns x FOUR output rows.  Its window columns 4k - 1 .. 4k + 3 are one aligned 16-byte load per
// input row (every byte of the plane is requested exactly once per row group) plus the element before it, which is the last float of the LEFT neighbour's
// load and arrives by a lane shift instead of a second request (round 4's form: 27 dword requests per four outputs, each input element requested ~3 x).
// Nine input rows feed the four output rows.  Each output folds its window in the reference's (ky, kx) order with its tap skips (pooling.rs:174-389):
// same bits.  Statistics (the quantizer that follows in the int8 graph): one atomic pair per WORKGROUP, after a wave shuffle fold and an LDS fold
// (round 4: one pair per wave on 256 slots -- 25 k contended atomics, 38.8 us against 25.4 without them).
template <bool STATS>
__global__ __launch_bounds__(256) void maxpool3x3s2_stream_kernel(const rten_hip_pool2d_desc d, const float *__restrict__ x, float *__restrict__ y,
                                                                  unsigned *__restrict__ stats, long long total) {
    __shared__ float red[2][4];
    const int kq = d.w >> 2, rgs = (d.out_h + 3) >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const float NINF = -__builtin_inff();
    float st_mn = __builtin_inff(), st_mx = NINF;
    const bool live = gid < total;
    {
        const long long g2 = live ? gid : total - 1;
        const unsigned per_plane = (unsigned)(kq * rgs);
        const long long plane = g2 / per_plane;
        const unsigned rem = (unsigned)(g2 - plane * per_plane);
        const int g = (int)(rem / (unsigned)kq), k = (int)(rem - (unsigned)g * (unsigned)kq);
        const int oy0 = 4 * g, iy0 = 2 * oy0 - 1;
        const float *in = x + plane * (long long)d.h * d.w + 4 * k;
        float4 v[9];
        float left[9];
        bool rok[9];
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const int iy = iy0 + r;
            rok[r] = (unsigned)iy < (unsigned)d.h;
            v[r] = *reinterpret_cast<const float4 *>(in + (long long)(rok[r] ? iy : 0) * d.w);
        }
        // column 4k - 1: the left neighbour's .w (the lane below holds k - 1 of the SAME row group unless this lane starts a row: k == 0 is the padding
        // column, never read); lane 0 of a wave has no lane below it and fetches the element itself
#pragma unroll
        for (int r = 0; r < 9; r++) left[r] = __shfl_up(v[r].w, 1, 64);
        if (lane == 0 && k > 0) {
#pragma unroll
            for (int r = 0; r < 9; r++) left[r] = in[(long long)(rok[r] ? iy0 + r : 0) * d.w - 1];
        }
        const bool lok = k > 0;
        if (live) {
            float *out = y + plane * (long long)d.out_h * d.out_w + 2 * k;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (oy0 + j >= d.out_h) break;
                float a0 = NINF, a1 = NINF;
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    const int r = 2 * j + ky;
                    if (rok[r]) {
                        if (lok) a0 = fmaxf(a0, left[r]);
                        a0 = fmaxf(a0, v[r].x);
                        a0 = fmaxf(a0, v[r].y);
                        a1 = fmaxf(a1, v[r].y);
                        a1 = fmaxf(a1, v[r].z);
                        a1 = fmaxf(a1, v[r].w);
                    }
                }
                *reinterpret_cast<float2 *>(out + (long long)(oy0 + j) * d.out_w) = make_float2(a0, a1);
                if (STATS) { st_mn = fminf(fminf(a0, a1), st_mn); st_mx = fmaxf(fmaxf(a0, a1), st_mx); }
            }
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { st_mn = fminf(st_mn, __shfl_xor(st_mn, o, 64)); st_mx = fmaxf(st_mx, __shfl_xor(st_mx, o, 64)); }
        if (lane == 0) { red[0][threadIdx.x >> 6] = st_mn; red[1][threadIdx.x >> 6] = st_mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
            const float mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
            if (mn <= mx) {
                const unsigned slot = blockIdx.x % (unsigned)dql::kStatSlots;
                atomicMin(&stats[slot], dql::f2ord(mn));
                atomicMax(&stats[dql::kStatSlots + slot], dql::f2ord(mx));
            }
        }
    }
}

template <bool IS_MAX>
int32_t run_pool(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *d, const float *x, float *y, const char *name, unsigned *stats = nullptr) {
    RTEN_CHECK_CTX(ctx);
    if (!d) return RTEN_HIP_ERR_INVALID_VALUE;
    if (d->n < 0 || d->c < 0 || d->h <= 0 || d->w <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride_h <= 0 ||
        d->stride_w <= 0 || d->out_h < 0 || d->out_w < 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "pool: invalid geometry");
    const long long total = (long long)d->n * d->c * d->out_h * d->out_w;
    if (total == 0) return RTEN_HIP_OK;
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    const long long planes = (long long)d->n * d->c, plane_in = (long long)d->h * d->w, plane_out = (long long)d->out_h * d->out_w;
    if (planes > 0x7fffffffLL || plane_in > 0x7fffffffLL || plane_out > 65535LL * 256)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "pool: plane too large");
    const dim3 grid((unsigned)planes, (unsigned)((plane_out + 255) / 256));
    ProfScope ps(ctx, name, 0.0, 4.0 * ((double)d->n * d->c * d->h * d->w + (double)total));
    if constexpr (IS_MAX) {
        // the streaming form: 3 x 3 / 2, one leading padding row and column, rows of whole 16-byte groups whose last group is fully used
        // (rows outside the image are skipped per tap, so any bottom padding / out_h goes; the columns need out_w == w / 2: no window reaches past w - 1)
        const bool stream = d->kh == 3 && d->kw == 3 && d->stride_h == 2 && d->stride_w == 2 && d->pads[0] == 1 && d->pads[1] == 1 && d->w % 4 == 0 &&
                            d->out_w * 2 == d->w && d->out_h >= 1 && ((uintptr_t)x & 15u) == 0 && ((uintptr_t)y & 7u) == 0 &&
                            !(ctx->debug & 0x100000); // RTEN_HIP_DEBUG bit 0x100000: the round-4 kernel (A/B)
        if (stream) {
            const long long threads = planes * (long long)(d->w / 4) * ((d->out_h + 3) / 4);
            const dim3 gs((unsigned)((threads + 255) / 256));
            if (stats) hipLaunchKernelGGL((maxpool3x3s2_stream_kernel<true>), gs, dim3(256), 0, ctx->stream, *d, x, y, stats, threads);
            else hipLaunchKernelGGL((maxpool3x3s2_stream_kernel<false>), gs, dim3(256), 0, ctx->stream, *d, x, y, stats, threads);
            RTEN_LAUNCH_CHECK(ctx, name);
            return RTEN_HIP_OK;
        }
    }
    if (d->kh == 3 && d->kw == 3 && (d->stride_h == 1 || d->stride_h == 2) && d->out_h >= 4) {
        const long long items = (long long)((d->out_h + 3) / 4) * d->out_w;
        const dim3 grid4((unsigned)planes, (unsigned)((items + 255) / 256));
        if (d->stride_h == 1) hipLaunchKernelGGL((pool3x3_y4_kernel<IS_MAX, 1>), grid4, dim3(256), 0, ctx->stream, *d, x, y, stats);
        else hipLaunchKernelGGL((pool3x3_y4_kernel<IS_MAX, 2>), grid4, dim3(256), 0, ctx->stream, *d, x, y, stats);
    } else if (d->kh == 3 && d->kw == 3) hipLaunchKernelGGL((pool2d_kernel<IS_MAX, 3, 3>), grid, dim3(256), 0, ctx->stream, *d, x, y, stats);
    else if (d->kh == 2 && d->kw == 2) hipLaunchKernelGGL((pool2d_kernel<IS_MAX, 2, 2>), grid, dim3(256), 0, ctx->stream, *d, x, y, stats);
    else hipLaunchKernelGGL((pool2d_kernel<IS_MAX, 0, 0>), grid, dim3(256), 0, ctx->stream, *d, x, y, stats);
    RTEN_LAUNCH_CHECK(ctx, name);
    return RTEN_HIP_OK;
}

} // namespace

RTEN_EXPORT int32_t rten_hip_max_pool2d_f32(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x,
                                            float *y) {
    return run_pool<true>(ctx, desc, x, y, "max_pool2d_f32");
}

RTEN_EXPORT int32_t rten_hip_max_pool2d_f32_stats(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x, float *y, void *stats) {
    if (!stats) return RTEN_HIP_ERR_INVALID_VALUE;
    return run_pool<true>(ctx, desc, x, y, "max_pool2d_f32", (unsigned *)stats);
}

RTEN_EXPORT int32_t rten_hip_average_pool2d_f32(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x,
                                                float *y) {
    return run_pool<false>(ctx, desc, x, y, "average_pool2d_f32");
}
