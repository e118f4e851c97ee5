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


// ---------------------------------------------------------------------------------------------------------------------------
// Fused form (round 5): the column matrix never exists.  Conditions: no dilation, one stride s <= 2 for both axes, C_g a multiple of 4 and <= 256 (the
// reference's GEMM is then ONE depth block per column element, rten-gemm/src/lib.rs:630-633), O_g <= 64, the row class's weights <= 64 KB.
// An output element is bias, then for (k_y, k_x) ascending the column element that lands on it (col2im, conv_transpose.rs:80-142), and that column
// element is a c-ordered fmaf chain from zero (gemm_uninit).  Here a WAVE owns one output row and 16 s consecutive columns of it, for all O_g channels:
// for k_y (the row's stride class), k_x ascending it runs the chain of that tap -- v_mfma_f32_16x16x4_f32, rows = output channels, columns = 16 input
// pixels i_x0 .. i_x0 + 15 of input row i_y, depth = c -- and adds it to the running outputs of the column class k_x belongs to (a separate add; pixels
// whose i_x falls outside the row keep their value: "not visited").  Same operations per element in the same order as GEMM + col2im: same bits.
// Workgroup = 4 waves of one (image, group, row class); the class's weights are staged once in LDS as [tap][c / 4][o][c % 4] (an MFMA A operand is then
// 64 consecutive floats).  Traffic: input + output + weights instead of writing and re-reading O_g * kh * kw * H * W floats per image.
// ---------------------------------------------------------------------------------------------------------------------------
typedef float ctf4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) float ct_smem[];

template <int JT, int S, bool FULL> // JT 16-channel blocks of the group's outputs; S = stride (1 or 2); FULL: C_g a multiple of 64 (whole units: no per-step tests)
__global__ __launch_bounds__(512) void conv_transpose_fused_kernel(const rten_hip_conv2d_desc d, const float *__restrict__ x, const float *__restrict__ w,
                                                                  const float *__restrict__ bias, float *__restrict__ y, int og, int cg, int chunks, int tasks_per_wg) {
    constexpr int OGP = 16 * JT, NW = 8; // waves per workgroup: two workgroups (2 x <= 64 KB of weights) keep 16 waves on a compute unit
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, quad = lane >> 4;
    // blockIdx.x = ((image * groups + group) * S + row class) * chunks + chunk
    int bi = blockIdx.x;
    const int chunk = bi % chunks; bi /= chunks;
    const int cls = bi % S; bi /= S;
    const int g = bi % d.groups, n = bi / d.groups;
    const int khw = d.kh * d.kw, nsteps = cg >> 2;
    // rows of this class: oy with (oy + pad_top) % S == cls; their taps: ky = cls, cls + S, ...
    const int nky = (d.kh - cls + S - 1) / S; // (cls < kh is checked by the launcher)
    // ---- stage the class's weights: kernel [C, O_g, kh, kw] -> Ws[(kyi * kw + kx)][c / 4][o][c % 4]; a thread moves one kernel row (kw taps) at a time
    {
        const float *wg = w + (long long)g * cg * og * khw;
        const int rows = cg * og * nky;
        for (int idx = t; idx < rows; idx += 64 * NW) {
            const int kyi = idx % nky, co = idx / nky, o = co % og, c = co / og;
            const float *src = wg + (long long)co * khw + (cls + kyi * S) * d.kw;
            float *dst = ct_smem + ((long long)(kyi * d.kw) * nsteps + (c >> 2)) * OGP * 4 + o * 4 + (c & 3);
            for (int kx = 0; kx < d.kw; kx++) dst[(long long)kx * nsteps * OGP * 4] = src[kx];
        }
    }
    __syncthreads();
    const int first = (cls - d.pads[0] % S + S) % S;          // rows oy = first, first + S, ...: (oy + pad_top) % S == cls
    const int rows_cls = (d.out_h + S - 1 - first) / S;
    const int nxb = (d.out_w + 16 * S - 1) / (16 * S);
    const int ntasks = rows_cls * nxb;
    float bo[JT][4];
#pragma unroll
    for (int j = 0; j < JT; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = 16 * j + 4 * quad + r;
            bo[j][r] = (bias && o < og) ? bias[g * og + o] : 0.0f;
        }
    const float *xg = x + ((long long)n * d.c + (long long)g * cg) * d.h * d.w;
    const long long plane_in = (long long)d.h * d.w;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)xg, 0, (int)((long long)cg * plane_in * 4), 0x00020000);
    const int nch = (nsteps + 15) >> 4, nunits = nky * d.kw * nch; // unit = 16 MFMA steps of one tap: its input values are fetched while the previous unit is multiplied
    for (int task = chunk * tasks_per_wg + wave; task < ntasks && task < (chunk + 1) * tasks_per_wg; task += NW) {
        const int ri = task / nxb, xb = task - ri * nxb;
        const int oy = first + ri * S, py = oy + d.pads[0];
        const int ox0 = xb * 16 * S;
        ctf4 tot[S][JT], acc[JT];
#pragma unroll
        for (int rx = 0; rx < S; rx++)
#pragma unroll
            for (int j = 0; j < JT; j++) tot[rx][j] = ctf4{bo[j][0], bo[j][1], bo[j][2], bo[j][3]};
        // unit u -> (tap, chunk of 16 steps); a tap whose input row does not exist is "not visited".  (Macros, not lambdas: with the value arrays passed
        // by reference the compiler kept them in scratch memory and drained the load counter in front of every multiply -- first version, 42 us.)
#define CT_DECODE(U)                                                                                                                     \
        const int ti_ = (U) / nch, ch_ = (U) - ti_ * nch, kyi_ = ti_ / d.kw, kx_ = ti_ - kyi_ * d.kw, ky_ = cls + kyi_ * S;            \
        const int iy_ = (py - ky_) / S;                                                                                                  \
        const int rx_ = ((kx_ - d.pads[1]) % S + S) % S;          /* the column class this tap feeds: (ox + pad_left - kx) % S == 0 */   \
        const int ix_ = l15 + (ox0 + rx_ + d.pads[1] - kx_) / S;  /* (an exact division) */                                              \
        const bool row_ = (U) < nunits && ky_ <= py && iy_ < d.h;                                                                        \
        const bool ok_ = row_ && (unsigned)ix_ < (unsigned)d.w;
#define CT_FETCH(U, XV)                                                                                                                  \
        {                                                                                                                                \
            CT_DECODE(U)                                                                                                                 \
            /* unconditional buffer loads: an offset beyond the slice returns 0 (a load under a per-lane condition becomes a branch, and the */ \
            /* load counter is then drained in front of every multiply) */                                                                \
            const unsigned off_ = ok_ ? (unsigned)(((quad + 64 * ch_) * (int)plane_in + iy_ * d.w + ix_) * 4) : 0x80000000u;             \
            _Pragma("unroll") for (int v = 0; v < 16; v++)                                                                               \
                XV[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, (int)((FULL || 16 * ch_ + v < nsteps) ? off_ : 0x80000000u), (int)((unsigned)(v * 4) * (unsigned)plane_in * 4u), 0)); \
        }
#define CT_MULTIPLY(U, XV)                                                                                                               \
        {                                                                                                                                \
            CT_DECODE(U)                                                                                                                 \
            if (row_) { /* (wave-uniform) */                                                                                             \
                if (ch_ == 0) {                                                                                                          \
                    _Pragma("unroll") for (int j = 0; j < JT; j++) acc[j] = ctf4{0.f, 0.f, 0.f, 0.f};                                    \
                }                                                                                                                        \
                const float *wp_ = ct_smem + ((long long)(kyi_ * d.kw + kx_) * nsteps + 16 * ch_) * OGP * 4 + l15 * 4 + quad;            \
                _Pragma("unroll") for (int v = 0; v < 16; v++) {                                                                         \
                    if (FULL || 16 * ch_ + v < nsteps) {                                                                                 \
                        _Pragma("unroll") for (int j = 0; j < JT; j++)                                                                   \
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp_[(v * OGP + 16 * j) * 4], XV[v], acc[j], 0, 0, 0);         \
                    }                                                                                                                    \
                }                                                                                                                        \
                if (ch_ == nch - 1) { /* the tap's chain is complete: one separate add per output it lands on */                         \
                    const bool hit0_ = ok_ && rx_ == 0, hit1_ = ok_ && rx_ == S - 1;                                                     \
                    _Pragma("unroll") for (int j = 0; j < JT; j++) {                                                                     \
                        const ctf4 s0_ = tot[0][j] + acc[j], s1_ = tot[S - 1][j] + acc[j];                                               \
                        if (S == 2 || true) {                                                                                            \
                            _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                              \
                                if (S == 1) tot[0][j][r] = hit0_ ? s0_[r] : tot[0][j][r];                                                \
                                else { tot[0][j][r] = hit0_ ? s0_[r] : tot[0][j][r]; tot[S - 1][j][r] = hit1_ ? s1_[r] : tot[S - 1][j][r]; } \
                            }                                                                                                            \
                        }                                                                                                                \
                    }                                                                                                                    \
                }                                                                                                                        \
            }                                                                                                                            \
        }
        float xa[16], xb2[16];
        CT_FETCH(0, xa)
        for (int u = 0; u < nunits; u += 2) {
            CT_FETCH(u + 1, xb2)
            CT_MULTIPLY(u, xa)
            CT_FETCH(u + 2, xa)
            if (u + 1 < nunits) CT_MULTIPLY(u + 1, xb2)
        }
#undef CT_DECODE
#undef CT_FETCH
#undef CT_MULTIPLY
        // ---- store: lane (pixel l15, quad) holds channels 16 j + 4 quad + r of columns ox0 + S l15 + {0 .. S-1}
        float *yo = y + ((long long)n * d.o + (long long)g * og) * d.out_h * d.out_w + (long long)oy * d.out_w;
        const bool pair = S == 2 && (d.out_w & 1) == 0 && (((uintptr_t)y) & 7u) == 0; // both columns of a lane as one 8-byte store
#pragma unroll
        for (int j = 0; j < JT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = 16 * j + 4 * quad + r;
                if (o >= og) continue;
                float *row = yo + (long long)o * d.out_h * d.out_w;
                const int ox = ox0 + S * l15;
                if (pair) {
                    if (ox < d.out_w) *reinterpret_cast<float2 *>(row + ox) = make_float2(tot[0][j][r], tot[S - 1][j][r]);
                } else {
#pragma unroll
                    for (int rx = 0; rx < S; rx++)
                        if (ox + rx < d.out_w) row[ox + rx] = tot[rx][j][r];
                }
            }
    }
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
    {   // the fused form (no column matrix): see conv_transpose_fused_kernel; RTEN_HIP_DEBUG bit 0x100000 = the GEMM + col2im sequence (A/B)
        const int S = d->stride_h;
        const int jt = Og <= 16 ? 1 : Og <= 32 ? 2 : 4;
        const size_t lds = (size_t)((d->kh + S - 1) / S) * d->kw * Cg * (16 * jt) * sizeof(float);
        if (!(ctx->debug & 0x100000) && !(M == 1 && ctx->gemv_order != 0) && d->dil_h == 1 && d->dil_w == 1 && d->stride_h == d->stride_w && S <= 2 && S <= d->kh && Cg % 4 == 0 && Cg <= 256 && Og <= 64 &&
            lds <= 64 * 1024 && plane < (1ll << 31) && (long long)Cg * P * 4 < (1ll << 31) && (long long)Cg * Og * d->kh * d->kw < (1ll << 31)) {
            const int rows_cls = (d->out_h + S - 1) / S, nxb = (d->out_w + 16 * S - 1) / (16 * S);
            const int tasks_per_wg = 8, chunks = (rows_cls * nxb + tasks_per_wg - 1) / tasks_per_wg; // one task per wave (8 waves)
            const long long wgs = (long long)d->n * d->groups * S * chunks;
            if (wgs <= 0x7fffffffLL) {
                ProfScope ps(ctx, "conv_transpose_fused_kernel", 2.0 * d->n * (double)d->c * P * Og * d->kh * d->kw, 4.0 * ((double)d->n * d->c * P + (double)d->n * d->o * plane));
                const dim3 grid((unsigned)wgs), block(512);
#define RTEN_CT_GO2(JTV, SV, FV) do { if (lds > 48 * 1024) hipFuncSetAttribute((const void *)conv_transpose_fused_kernel<JTV, SV, FV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                      hipLaunchKernelGGL((conv_transpose_fused_kernel<JTV, SV, FV>), grid, block, lds, ctx->stream, *d, x, w, bias, y, Og, Cg, chunks, tasks_per_wg); } while (0)
#define RTEN_CT_GO(JTV, SV) do { if (Cg % 64 == 0) RTEN_CT_GO2(JTV, SV, true); else RTEN_CT_GO2(JTV, SV, false); } while (0)
                if (S == 1) { if (jt == 1) RTEN_CT_GO(1, 1); else if (jt == 2) RTEN_CT_GO(2, 1); else RTEN_CT_GO(4, 1); }
                else { if (jt == 1) RTEN_CT_GO(1, 2); else if (jt == 2) RTEN_CT_GO(2, 2); else RTEN_CT_GO(4, 2); }
#undef RTEN_CT_GO
#undef RTEN_CT_GO2
                RTEN_LAUNCH_CHECK(ctx, "conv_transpose_fused_kernel launch");
                return RTEN_HIP_OK;
            }
        }
    }
    float *cols = (float *)rten_aux_scratch(ctx, (size_t)d->n * M * P * sizeof(float));
    if (!cols) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "conv_transpose: column buffer allocation failed (or attempted during graph capture)");
    for (int g = 0; g < d->groups; g++) {
        rten_hip_gemm_desc gd = {};
        gd.m = (int32_t)M; gd.n = (int32_t)P; gd.k = Cg;
        gd.a_rs = 1; gd.a_cs = M;        // A[m][k] = kernel[g*Cg + k][m]: the transposed kernel matrix
        gd.b_rs = P; gd.b_cs = 1; gd.ldc = P;
        gd.batch = d->n; gd.a_bs = 0; gd.b_bs = (long long)d->c * P; gd.c_bs = M * P;
        gd.alpha = 1.f; gd.beta = 0.f;
        // (a kernel matrix of ONE row -- O_g = kh = kw = 1 -- is a one-row product of unpacked operands: the reference's vector-matrix order, lib.rs:876-891)
        const int32_t rc = (M == 1 && ctx->gemv_order != 0) ? rten_hip_gemm_f32(ctx, &gd, w + (long long)g * Cg * M, x + (long long)g * Cg * P, nullptr, cols)
                                                            : rten_gemm_f32_blocked(ctx, &gd, w + (long long)g * Cg * M, x + (long long)g * Cg * P, nullptr, cols);
        if (rc) return rc;
        ProfScope ps(ctx, "col2im_f32", 0.0, 4.0 * ((double)d->n * M * P + (double)d->n * Og * plane));
        hipLaunchKernelGGL(col2im_kernel, dim3((unsigned)(d->n * Og), (unsigned)((plane + 255) / 256)), dim3(256), 0, ctx->stream, *d, Og, g * Og, cols, bias, y);
        RTEN_LAUNCH_CHECK(ctx, "col2im_kernel launch");
    }
    return RTEN_HIP_OK;
}
