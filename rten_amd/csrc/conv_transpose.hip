// ConvTranspose (src/ops/conv_transpose.rs:226-412): the inverse of the im2col convolution.  Per group:
//     columns[N][O_g*kh*kw, H*W] = kernel_mat^T [O_g*kh*kw, C_g] . input_mat [C_g, H*W]   one batched call of the f32 GEMM
//                                                                                         (same k-ordered FMA chain as the reference's gemm_uninit)
//     col2im (conv_transpose.rs:80-142): every output element starts at its bias and receives, for k_y, k_x in order, the one
//     column element that maps onto it -- gathered here by one thread per output element, so the add order is the reference's.
// The column matrix lives in the context's auxiliary scratch (the GEMM owns the main one for its split-K slabs).
#include "internal.h"

namespace {

__global__ __launch_bounds__(256) void col2im_kernel(const rten_hip_conv2d_desc d, int og, int o0, const float *__restrict__ cols, const float *__restrict__ bias,
                                                     float *__restrict__ y) {
    // grid: x = image * og + local out channel, y = 256-element slabs of the output plane
    const int plane = d.out_h * d.out_w;
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= plane) return;
    const int n = blockIdx.x / og, ol = blockIdx.x - n * og;
    const int oy = e / d.out_w, ox = e - oy * d.out_w;
    const int P = d.h * d.w, M = og * d.kh * d.kw;
    const float *cn = cols + ((long long)n * M + (long long)ol * d.kh * d.kw) * P;
    float acc = bias ? bias[o0 + ol] : 0.0f;
    if (d.dil_h == 1 && d.dil_w == 1) {
        // Without dilation the taps that land on this output element are ky = ry, ry + s_h, ... and kx = rx, rx + s_w, ... (ry = (oy + pad) mod s_h):
        // only those are visited -- in the same increasing (ky, kx) order, so the adds are the reference's -- instead of testing all kh * kw taps
        // (a 4x4 / stride 2 kernel: 4 visits instead of 16 divergent tests; the column reads of a wave are then two interleaved unit-stride runs).
        const int py = oy + d.pads[0], px = ox + d.pads[1];
        for (int ky = py % d.stride_h; ky < d.kh && ky <= py; ky += d.stride_h) {
            const int iy = (py - ky) / d.stride_h;
            if (iy >= d.h) continue;
            for (int kx = px % d.stride_w; kx < d.kw && kx <= px; kx += d.stride_w) {
                const int ix = (px - kx) / d.stride_w;
                if (ix >= d.w) continue;
                acc = acc + cn[(long long)(ky * d.kw + kx) * P + iy * d.w + ix];
            }
        }
        y[((long long)n * d.o + o0 + ol) * plane + e] = acc;
        return;
    }
    for (int ky = 0; ky < d.kh; ky++) {
        const int ty = oy + d.pads[0] - ky * d.dil_h;
        if (ty < 0 || ty % d.stride_h != 0) continue;
        const int iy = ty / d.stride_h;
        if (iy >= d.h) continue;
        for (int kx = 0; kx < d.kw; kx++) {
            const int tx = ox + d.pads[1] - kx * d.dil_w;
            if (tx < 0 || tx % d.stride_w != 0) continue;
            const int ix = tx / d.stride_w;
            if (ix >= d.w) continue;
            acc = acc + cn[(long long)(ky * d.kw + kx) * P + iy * d.w + ix];
        }
    }
    y[((long long)n * d.o + o0 + ol) * plane + e] = acc;
}

} // namespace

// conv_transpose_output_size_and_padding (conv_transpose.rs:144-224): same checks, same messages.
RTEN_EXPORT int32_t rten_hip_conv_transpose_output_size(int32_t in_h, int32_t in_w, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w, int32_t same,
                                                        const int32_t pads[4], int32_t dil_h, int32_t dil_w, int32_t out_pad_h, int32_t out_pad_w, int32_t out_hw[2],
                                                        int32_t out_pads[4], const char **msg) {
    auto fail = [&](const char *m) { if (msg) *msg = m; return RTEN_HIP_ERR_INVALID_VALUE; };
    if (!out_hw || !out_pads || (!same && !pads)) return fail("NULL argument");
    if (stride_h <= 0 || stride_w <= 0) return fail("Strides must be > 0");
    if (dil_h <= 0 || dil_w <= 0) return fail("Dilations must be > 0");
    if (kh <= 0 || kw <= 0) return fail("Kernel size must be > 0");
    if (in_h <= 0 || in_w <= 0) return fail("Input width and height must be > 0");
    const long long keh = (long long)(kh - 1) * dil_h + 1, kew = (long long)(kw - 1) * dil_w + 1;
    const long long full_h = (long long)(in_h - 1) * stride_h + keh + out_pad_h, full_w = (long long)(in_w - 1) * stride_w + kew + out_pad_w;
    if (same) {
        const long long oh = (long long)in_h * stride_h, ow = (long long)in_w * stride_w;
        const long long ph = full_h - oh, pw = full_w - ow;
        if (ph < 0 || pw < 0) return fail("Input is too small");
        out_hw[0] = (int32_t)oh; out_hw[1] = (int32_t)ow;
        out_pads[0] = (int32_t)(ph / 2); out_pads[1] = (int32_t)(pw / 2); out_pads[2] = (int32_t)((ph + 1) / 2); out_pads[3] = (int32_t)((pw + 1) / 2);
        return RTEN_HIP_OK;
    }
    const long long oh = full_h - pads[0] - pads[2], ow = full_w - pads[1] - pads[3];
    if (oh < 0 || ow < 0) return fail("Input is too small");
    out_hw[0] = (int32_t)oh; out_hw[1] = (int32_t)ow;
    for (int i = 0; i < 4; i++) out_pads[i] = pads[i];
    return RTEN_HIP_OK;
}

// desc: n, c, h, w = input; o = output channels (all groups); kernel [c, o / groups, kh, kw]; pads = resolved [top, left, bottom, right];
// out_h, out_w from rten_hip_conv_transpose_output_size.
RTEN_EXPORT int32_t rten_hip_conv_transpose2d_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const float *x, const float *w, const float *bias, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (!d) return RTEN_HIP_ERR_INVALID_VALUE;
    if (d->groups <= 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Group count must be > 0");
    if (d->n < 0 || d->c < 0 || d->o < 0 || d->h <= 0 || d->w <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride_h <= 0 || d->stride_w <= 0 || d->dil_h <= 0 ||
        d->dil_w <= 0 || d->out_h < 0 || d->out_w < 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv_transpose: invalid geometry");
    if (d->c % d->groups != 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Input channel count not divisible by groups");
    if (d->o % d->groups != 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Output channel count not divisible by groups");
    if (d->n == 0 || d->o == 0 || d->out_h == 0 || d->out_w == 0) return RTEN_HIP_OK;
    if (!x || !w || !y) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv_transpose: NULL operand");
    const int Cg = d->c / d->groups, Og = d->o / d->groups;
    const long long P = (long long)d->h * d->w, M = (long long)Og * d->kh * d->kw, plane = (long long)d->out_h * d->out_w;
    if (M > 0x7fffffffLL || P > 0x7fffffffLL || plane > 65535LL * 256 || (long long)d->n * Og > 0x7fffffffLL)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv_transpose: geometry too large");
    float *cols = (float *)rten_aux_scratch(ctx, (size_t)d->n * M * P * sizeof(float));
    if (!cols) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "conv_transpose: column buffer allocation failed (or attempted during graph capture)");
    for (int g = 0; g < d->groups; g++) {
        rten_hip_gemm_desc gd = {};
        gd.m = (int32_t)M; gd.n = (int32_t)P; gd.k = Cg;
        gd.a_rs = 1; gd.a_cs = M;        // A[m][k] = kernel[g*Cg + k][m]: the transposed kernel matrix
        gd.b_rs = P; gd.b_cs = 1; gd.ldc = P;
        gd.batch = d->n; gd.a_bs = 0; gd.b_bs = (long long)d->c * P; gd.c_bs = M * P;
        gd.alpha = 1.f; gd.beta = 0.f;
        const int32_t rc = rten_gemm_f32_blocked(ctx, &gd, w + (long long)g * Cg * M, x + (long long)g * Cg * P, nullptr, cols);
        if (rc) return rc;
        ProfScope ps(ctx, "col2im_f32", 0.0, 4.0 * ((double)d->n * M * P + (double)d->n * Og * plane));
        hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)(d->n * Og), (unsigned)((plane + 255) / 256)), dim3(256), 0, ctx->stream, *d, Og, g * Og, cols, bias, y);
        RTEN_LAUNCH_CHECK(ctx, "col2im_kernel launch");
    }
    return RTEN_HIP_OK;
}
