// Depthwise convolution (groups == C == O), f32: the reference routes these to its own kernel (src/ops/conv.rs:269-284 ->
// src/ops/conv/depthwise.rs:95-146, 215-262), whose arithmetic is NOT the GEMM's: the accumulator starts at the bias, every
// in-bounds tap contributes one rounded multiply followed by one add (Rust does not contract `acc += x * w` into an FMA),
// taps are visited in (k_y, k_x) order and padding taps are not visited at all.  One thread per output element replays
// exactly that sequence, so the result is bit-identical to the reference's; the generic grouped-GEMM path would differ in the
// last bits (FMA chain from zero, bias afterwards).  HBM-bound: every input element is read kh*kw/stride^2 times through L1 / L2.
#include "internal.h"
#include "vecmath.h"

namespace {

template <int KH, int KW> // > 0: compile-time window with all taps' loads issued first; 0: runtime window
__global__ __launch_bounds__(256) void depthwise_conv2d_f32_kernel(const rten_hip_conv2d_desc d, const float *__restrict__ x, const float *__restrict__ w,
                                                                   int w_stride, const float *__restrict__ bias, const float *__restrict__ residual,
                                                                   int relu, float *__restrict__ y) {
    const int plane = d.out_h * d.out_w;
    const int o = blockIdx.y * 256 + threadIdx.x;
    if (o >= plane) return;
    const int nc = blockIdx.x, c = nc % d.c;
    const int oy = o / d.out_w, ox = o - oy * d.out_w;
    const float *xc = x + (long long)nc * d.h * d.w;
    const int kh = KH > 0 ? KH : d.kh, kw = KW > 0 ? KW : d.kw;
    const float *wc = w + (long long)c * kh * kw * w_stride;
    const int y0 = oy * d.stride_h - d.pads[0], x0 = ox * d.stride_w - d.pads[1];
    float acc = bias ? bias[c] : 0.0f;
    if constexpr (KH > 0) {
        float xv[KH * KW], wv[KH * KW];
        bool ok[KH * KW];
#pragma unroll
        for (int ky = 0; ky < KH; ky++)
#pragma unroll
            for (int kx = 0; kx < KW; kx++) {
                const int iy = y0 + ky * d.dil_h, ix = x0 + kx * d.dil_w;
                const bool in = (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
                ok[ky * KW + kx] = in;
                xv[ky * KW + kx] = xc[in ? iy * d.w + ix : 0];
                wv[ky * KW + kx] = wc[(ky * KW + kx) * w_stride];
            }
#pragma unroll
        for (int t = 0; t < KH * KW; t++)
            if (ok[t]) acc = __fadd_rn(acc, __fmul_rn(xv[t], wv[t]));
    } else {
        for (int ky = 0; ky < kh; ky++) {
            const int iy = y0 + ky * d.dil_h;
            if ((unsigned)iy >= (unsigned)d.h) continue;
            for (int kx = 0; kx < kw; kx++) {
                const int ix = x0 + kx * d.dil_w;
                if ((unsigned)ix >= (unsigned)d.w) continue;
                acc = __fadd_rn(acc, __fmul_rn(xc[iy * d.w + ix], wc[(ky * kw + kx) * w_stride]));
            }
        }
    }
    const long long oi = (long long)nc * plane + o;
    if (residual) acc = acc + residual[oi]; // the Add node that follows (binary_elementwise.rs:476-495)
    if (relu) acc = vm::relu(acc);
    y[oi] = acc;
}

// 3 x 3 window, dilation 1, stride_h SH: four VERTICALLY adjacent outputs per thread (lanes still walk x, so every load and store
// stays coalesced).  The ((4-1)*SH + 3) x 3 input patch is loaded once and each output replays its own (k_y, k_x)-ordered
// sequence from registers: the arithmetic per output element is unchanged, the kernel issues half the loads and a quarter of
// the workgroups.
template <int SH>
__global__ __launch_bounds__(256) void depthwise3x3_y4_kernel(const rten_hip_conv2d_desc d, const float *__restrict__ x, const float *__restrict__ w, int w_stride,
                                                             const float *__restrict__ bias, const float *__restrict__ residual, int relu, float *__restrict__ y) {
    constexpr int NROW = 3 * SH + 3;
    const int hq = (d.out_h + 3) >> 2, items = hq * d.out_w;
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= items) return;
    const int nc = blockIdx.x, c = nc % d.c;
    const int yq = q / d.out_w, ox = q - yq * d.out_w, oy0 = yq * 4;
    const float *xc = x + (long long)nc * d.h * d.w;
    const float *wc = w + (long long)c * 9 * w_stride;
    const int y0 = oy0 * SH - d.pads[0], x0 = ox * d.stride_w - d.pads[1];
    float xv[NROW][3], wv[9];
    bool rok[NROW], cok[3];
#pragma unroll
    for (int kx = 0; kx < 3; kx++) cok[kx] = (unsigned)(x0 + kx) < (unsigned)d.w;
#pragma unroll
    for (int r = 0; r < NROW; r++) {
        const int iy = y0 + r;
        rok[r] = (unsigned)iy < (unsigned)d.h;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) xv[r][kx] = xc[(rok[r] && cok[kx]) ? iy * d.w + x0 + kx : 0];
    }
#pragma unroll
    for (int t = 0; t < 9; t++) wv[t] = wc[t * w_stride];
    const float b = bias ? bias[c] : 0.0f;
    const long long o0 = (long long)nc * d.out_h * d.out_w + ox;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (oy0 + j >= d.out_h) break;
        float acc = b;
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
                if (rok[j * SH + ky] && cok[kx]) acc = __fadd_rn(acc, __fmul_rn(xv[j * SH + ky][kx], wv[ky * 3 + kx]));
        const long long oi = o0 + (long long)(oy0 + j) * d.out_w;
        if (residual) acc = acc + residual[oi];
        if (relu) acc = vm::relu(acc);
        y[oi] = acc;
    }
}

// 3 x 3 / stride 1 / dilation 1 with one padding column on either side (out_w == w, w a multiple of 4: MobileNet-style blocks) as a streaming kernel:
// a thread owns FOUR adjacent output columns x FOUR output rows.  Its window columns 4k - 1 .. 4k + 4 are one aligned 16-byte load per input row plus
// the last float of the LEFT neighbour's load and the first of the RIGHT neighbour's, which arrive by lane shifts (the four-outputs-per-thread form above:
// 18 dword requests per four outputs, each input element requested ~3 x; here every byte of the plane is requested once per row group, and the store is
// 16 bytes).  Each output replays the reference's sequence (accumulator = bias; per in-bounds tap in (k_y, k_x) order one rounded multiply and one add;
// padding taps not visited: conv/depthwise.rs:95-146): same bits.
__global__ __launch_bounds__(256) void depthwise3x3s1_stream_kernel(const rten_hip_conv2d_desc d, const float *__restrict__ x, const float *__restrict__ w, int w_stride,
                                                                   const float *__restrict__ bias, const float *__restrict__ residual, int relu, float *__restrict__ y,
                                                                   long long total) {
    const int kq = d.w >> 2, rgs = (d.out_h + 3) >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = gid < total;
    const long long g2 = live ? gid : total - 1;
    const unsigned per_plane = (unsigned)(kq * rgs);
    const long long plane = g2 / per_plane;
    const unsigned rem = (unsigned)(g2 - plane * per_plane);
    const int g = (int)(rem / (unsigned)kq), k = (int)(rem - (unsigned)g * (unsigned)kq);
    const int c = (int)(plane % d.c);
    const int oy0 = 4 * g, iy0 = oy0 - d.pads[0];
    const float *in = x + plane * (long long)d.h * d.w + 4 * k;
    const float *wc = w + (long long)c * 9 * w_stride;
    float4 v[6];
    float left[6], right[6], wv[9];
    bool rok[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const int iy = iy0 + r;
        rok[r] = (unsigned)iy < (unsigned)d.h;
        v[r] = *reinterpret_cast<const float4 *>(in + (long long)(rok[r] ? iy : 0) * d.w);
    }
#pragma unroll
    for (int t = 0; t < 9; t++) wv[t] = wc[t * w_stride];
    const float b = bias ? bias[c] : 0.0f;
    // columns 4k - 1 and 4k + 4: the neighbouring lanes hold k - 1 / k + 1 of the SAME row group unless this lane starts / ends a row (those are the padding
    // columns, never read); the first / last lane of a wave has no such neighbour and fetches the element itself
#pragma unroll
    for (int r = 0; r < 6; r++) { left[r] = __shfl_up(v[r].w, 1, 64); right[r] = __shfl_down(v[r].x, 1, 64); }
    const bool lok = k > 0, rgt = k < kq - 1;
    if (lane == 0 && lok) {
#pragma unroll
        for (int r = 0; r < 6; r++) left[r] = in[(long long)(rok[r] ? iy0 + r : 0) * d.w - 1];
    }
    if (lane == 63 && rgt) {
#pragma unroll
        for (int r = 0; r < 6; r++) right[r] = in[(long long)(rok[r] ? iy0 + r : 0) * d.w + 4];
    }
    if (!live) return;
    const long long o0 = plane * (long long)d.out_h * d.out_w + 4 * k;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (oy0 + j >= d.out_h) break;
        float acc[4] = {b, b, b, b};
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const int r = j + ky;
            if (!rok[r]) continue;
            const float col[6] = {left[r], v[r].x, v[r].y, v[r].z, v[r].w, right[r]};
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool in_w = (i + kx > 0 || lok) && (i + kx < 5 || rgt); // window column 4k + i - 1 + kx
                    if (in_w) acc[i] = __fadd_rn(acc[i], __fmul_rn(col[i + kx], wv[ky * 3 + kx]));
                }
        }
        const long long oi = o0 + (long long)(oy0 + j) * d.out_w;
        if (residual) {
            const float4 rr = *reinterpret_cast<const float4 *>(residual + oi);
            acc[0] = acc[0] + rr.x; acc[1] = acc[1] + rr.y; acc[2] = acc[2] + rr.z; acc[3] = acc[3] + rr.w;
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = vm::relu(acc[i]);
        }
        *reinterpret_cast<float4 *>(y + oi) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

} // namespace

// Called by rten_hip_conv2d_f32 for groups == C == O geometries (weights OIHW [C,1,kh,kw], or the prepacked form whose
// per-group K x 4 block holds the taps at stride 4).
int32_t rten_depthwise_conv2d_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const float *x, const float *w, int32_t weights_packed, const float *bias,
                                  const float *residual, uint32_t flags, float *y) {
    const long long planes = (long long)d->n * d->c, plane = (long long)d->out_h * d->out_w;
    if (planes > 0x7fffffffLL || plane > 65535LL * 256 || (long long)d->h * d->w > 0x7fffffffLL)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "depthwise conv: plane too large");
    const dim3 grid((unsigned)planes, (unsigned)((plane + 255) / 256));
    const int ws = weights_packed ? 4 : 1, relu = (flags & RTEN_HIP_CONV_RELU) ? 1 : 0;
    const float *res = (flags & RTEN_HIP_CONV_RESIDUAL) ? residual : nullptr;
    ProfScope ps(ctx, "depthwise_conv2d_f32", 2.0 * planes * plane * d->kh * d->kw, 4.0 * (planes * (double)d->h * d->w + planes * (double)plane));
    // the streaming form: 3 x 3, stride 1, dilation 1, one padding column on either side, rows of whole 16-byte groups (RTEN_HIP_DEBUG bit 0x100000: the round-4 kernel, A/B)
    const bool stream = d->kh == 3 && d->kw == 3 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 && d->pads[1] == 1 && d->out_w == d->w &&
                        d->w % 4 == 0 && d->out_h >= 1 && ((uintptr_t)x & 15u) == 0 && ((uintptr_t)y & 15u) == 0 && (!res || ((uintptr_t)res & 15u) == 0) &&
                        !(ctx->debug & 0x100000);
    if (stream) {
        const long long threads = planes * (long long)(d->w / 4) * ((d->out_h + 3) / 4);
        hipLaunchKernelGGL(depthwise3x3s1_stream_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, *d, x, w, ws, bias, res, relu, y, threads);
        RTEN_LAUNCH_CHECK(ctx, "depthwise3x3s1_stream_kernel launch");
        return RTEN_HIP_OK;
    }
    const bool y4 = d->kh == 3 && d->kw == 3 && d->dil_h == 1 && d->dil_w == 1 && (d->stride_h == 1 || d->stride_h == 2) && d->out_h >= 4;
    if (y4) {
        const long long items = (long long)((d->out_h + 3) / 4) * d->out_w;
        const dim3 grid4((unsigned)planes, (unsigned)((items + 255) / 256));
        if (d->stride_h == 1) hipLaunchKernelGGL((depthwise3x3_y4_kernel<1>), grid4, dim3(256), 0, ctx->stream, *d, x, w, ws, bias, res, relu, y);
        else hipLaunchKernelGGL((depthwise3x3_y4_kernel<2>), grid4, dim3(256), 0, ctx->stream, *d, x, w, ws, bias, res, relu, y);
    } else if (d->kh == 3 && d->kw == 3) hipLaunchKernelGGL((depthwise_conv2d_f32_kernel<3, 3>), grid, dim3(256), 0, ctx->stream, *d, x, w, ws, bias, res, relu, y);
    else if (d->kh == 5 && d->kw == 5) hipLaunchKernelGGL((depthwise_conv2d_f32_kernel<5, 5>), grid, dim3(256), 0, ctx->stream, *d, x, w, ws, bias, res, relu, y);
    else hipLaunchKernelGGL((depthwise_conv2d_f32_kernel<0, 0>), grid, dim3(256), 0, ctx->stream, *d, x, w, ws, bias, res, relu, y);
    RTEN_LAUNCH_CHECK(ctx, "depthwise_conv2d_f32_kernel launch");
    return RTEN_HIP_OK;
}
