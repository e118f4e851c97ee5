// Few-channel strided convolutions (ResNet's stem: 3 -> 64 channels, 7x7, stride 2; src/ops/conv.rs:124-365 through the im2col GEMM of
// rten-gemm/src/im2col.rs) as a DIRECT implicit GEMM: GEMM variant 32 (round 6).
//
// The generic path gathers the B operand element by element (K = C * KH * KW = 147 rows of an im2col matrix whose neighbouring columns are two
// pixels apart: 4-byte requests at half sector efficiency, 147 x 64 of them per 64-column k-slab), and the step pays 86 us for 7.6 GFLOP.  Here a
// workgroup owns a 16 x 16 tile of output pixels of one image and all (<= 64) output channels: the input patch it reads -- (16 - 1) * 2 + 7 = 37 rows and
// columns of each channel, zero where the image ends -- is loaded ONCE into LDS (21 KB, row pitch 48 so that the two output rows of an MFMA block fall on
// disjoint banks), the k-major prepacked weights arrive by LDS-DMA (148 x 64 floats: row 147 is past the buffer and reads as zeros), and the MFMA B
// fragment of depth index k = (c, ky, kx) and output pixel (oy, ox) is the patch element [c][2 oy + ky][2 ox + kx]: one ds_read_b32 whose address is a
// per-lane base plus a COMPILE-TIME offset.  The depth index runs in the reference's im2col order (channel, then kernel row, then kernel column),
// k-pair by k-pair on one accumulator per output, padded taps contribute a * 0 exactly as the generic kernel's out-of-range gathers do, and K <= 256 is
// one depth block: the result is bit-identical to variant 3.
#include "gemm_f32_common.h"

namespace {

struct StemArgs {
    const float *X, *W, *B;
    float *Y;
    int N, H, Wd, O, OH, OW, pt, pl;
    int w_cs;          // row length of the k-major packed weights
    unsigned x_bytes, w_bytes;
    int tiles_x, tiles_y, act;
    int pad_[12];      // (3 cache lines: kernarg_prefetch takes 3, 5 or 7)
};
static_assert(sizeof(StemArgs) > 128 && sizeof(StemArgs) <= 192, "StemArgs: three cache lines");

typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int C, int KH, int KW, int S>
__global__ __launch_bounds__(256, 2) void conv_small_c_f32_kernel(const StemArgs p) {
    kernarg_prefetch<(int)sizeof(StemArgs)>();
    constexpr int TH = 16, TW = 16;                         // output pixels per workgroup
    constexpr int PH = (TH - 1) * S + KH, PWU = (TW - 1) * S + KW; // patch rows / used columns
    constexpr int PW = 48;                                  // row pitch: two output rows (S * PW floats apart) land on disjoint bank halves
    static_assert(PWU <= PW && S == 2, "patch pitch / bank layout are laid out for stride 2");
    constexpr int K = C * KH * KW, NKK = (K + 1) / 2;       // depth, k-pairs
    constexpr int PLANE = PH * PW, PATCH = C * PLANE;       // floats
    constexpr int AROWS = 2 * NKK;                          // weight rows in LDS (an odd K: one row of zeros)
    constexpr int AS = (PATCH + 1 + 63) & ~63;              // weights start (after the patch and one zero cell)
    constexpr int NA = (AROWS * 64 + 1023) / 1024;          // dwordx4 DMA instructions per wave
    constexpr int ZERO = PATCH;                             // a cell that holds 0.0f (B operand of the padded depth row)
    __shared__ __attribute__((aligned(16))) float smem[AS + NA * 1024];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wq = t >> 6;

    int tile;
    {   // workgroups of one XCD take a contiguous range of tiles
        const int id = blockIdx.x, nt = (int)gridDim.x;
        const int xcd = id & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int per_img = p.tiles_x * p.tiles_y;
    const int img = tile / per_img, tr = tile - img * per_img;
    const int ty = tr / p.tiles_x, tx = tr - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)p.X, 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)p.W, 0, (int)p.w_bytes, 0x00020000);

    // ---- weights: [AROWS][64] by LDS-DMA (rows >= K and columns >= w_cs are out of range: zeros)
#pragma unroll
    for (int j = 0; j < NA; j++) {
        const int f = (wave * NA + j) * 256 + lane * 4;
        const int k = f >> 6, m = f & 63;
        const unsigned voff = (m < p.w_cs && k < K) ? (unsigned)((k * p.w_cs + m) * 4) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(smem + AS + (wave * NA + j) * 256), 16, (int)voff, 0, 0, 0);
    }
    // ---- the input patch: [C][PH][PW] (columns >= PWU unused), zero outside the image.  (Several tiles per workgroup with the next tile's patch requested
    // under this tile's MFMAs were measured: 73-78 us under co-run against 73.6 for this form, 88-106 alone against 82 -- two workgroups per compute unit
    // already overlap one's loads with the other's MFMAs.)
    {
        const int iy0 = oy0 * S - p.pt, ix0 = ox0 * S - p.pl;
        constexpr int NE = C * PH * PWU, NQ = (NE + 255) / 256;
        float v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int e = q * 256 + t;
            const int c = e / (PH * PWU), r = e - c * (PH * PWU);
            const int py = r / PWU, px = r - py * PWU;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = e < NE && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
            v[q] = buf_load1(rsX, ok ? (unsigned)((((long long)img * C + c) * p.H + iy) * p.Wd + ix) << 2 : OOB, 0);
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int e = q * 256 + t;
            const int c = e / (PH * PWU), r = e - c * (PH * PWU);
            const int py = r / PWU, px = r - py * PWU;
            if (e < NE) smem[c * PLANE + py * PW + px] = v[q];
        }
        if (t == 0) smem[ZERO] = 0.f;
    }

    // ---- this lane's pixels: wave w owns output rows [4 w, 4 w + 4) of the tile; block j = rows 4 w + 2 j, + 1; lane l31 = (row l31 >> 4, column l31 & 15)
    constexpr int D1 = 1, D2 = PW - (KW - 1), D3 = PLANE - (KH - 1) * PW - (KW - 1); // offset(k + 1) - offset(k): inside a kernel row / to the next row / to the next channel
    int b1[2], b2[2], b3[2], bz[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = 4 * wq + 2 * j + (l31 >> 4), c = l31 & 15;
        const int base = r * S * PW + c * S;
        b1[j] = base + half * D1;
        b2[j] = base + half * D2;
        b3[j] = base + half * D3;
        bz[j] = half ? ZERO : base; // the last pair of an odd K: depth row K is padding (A row of zeros; B reads the zero cell)
    }
    const float *As = smem + AS + half * 64 + l31;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    wait_vmcnt<0>();
    __syncthreads();

    // offset of depth index k in the patch, and which per-lane base its k-pair uses
    auto off_of = [](int k) { const int c = k / (KH * KW), r = k - c * (KH * KW); const int ky = r / KW, kx = r - ky * KW; return c * PLANE + ky * PW + kx; };
    float af[2][2], bf[2][2];
    auto load_frags = [&](int kk, int buf) {
        const int k0 = 2 * kk;
        const int o0 = off_of(k0);
#pragma unroll
        for (int i = 0; i < 2; i++) af[buf][i] = As[k0 * 64 + i * 32];
        const int r0 = k0 % (KH * KW), kx0 = r0 % KW, ky0 = r0 / KW;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (k0 + 1 >= K) bf[buf][j] = smem[bz[j] + (half ? 0 : o0)];
            else if (kx0 < KW - 1) bf[buf][j] = smem[b1[j] + o0];
            else if (ky0 < KH - 1) bf[buf][j] = smem[b2[j] + o0];
            else bf[buf][j] = smem[b3[j] + o0];
        }
    };
    load_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
        const int cur = kk & 1;
        if (kk + 1 < NKK) load_frags(kk + 1, cur ^ 1);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_iglp_opt(0); // (the next k-pair's ds_reads go between this pair's MFMAs instead of in front of their first use)

    // ---- epilogue: bias, activation, NCHW store (rows of 16 pixels: 64-byte runs)
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)p.Y, 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)(p.B ? p.B : p.W), 0, p.B ? p.O * 4 : 0, 0x00020000);
    const unsigned plane4 = (unsigned)(p.OH * p.OW) << 2;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; r++) bias[r] = buf_load1(rsB, (unsigned)(4 * half) << 2, (unsigned)(i * 32 + acc_row(r)) << 2);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int oy = oy0 + 4 * wq + 2 * j + (l31 >> 4), ox = ox0 + (l31 & 15);
            const bool ok = oy < p.OH && ox < p.OW;
            const unsigned pix = ok ? (unsigned)((((long long)img * p.O + 4 * half) * p.OH + oy) * p.OW + ox) << 2 : OOB;
            f32x16 v = acc[i][j];
            if (p.B) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = v[r] + bias[r];
            }
            if (p.act == RTEN_HIP_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = vm::relu(v[r]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = i * 32 + acc_row(r); // (+ 4 * half: in the lane offset)
                const unsigned voff = (m + 4 * half < p.O) ? pix : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v[r]), rsY, (int)voff, (int)((unsigned)m * plane4), 0);
            }
        }
    }
}

} // namespace

// gemm_f32.hip asks before its generic plan when GEMM variant 32 is selected
bool rten_small_c_conv_f32_supported(const rten_hip_conv2d_desc *d, int weights_packed, const float *residual) {
    return weights_packed && !residual && d->groups == 1 && d->c == 3 && d->kh == 7 && d->kw == 7 && d->stride_h == 2 && d->stride_w == 2 && d->dil_h == 1 && d->dil_w == 1 &&
           d->o >= 1 && d->o <= 64 && d->n >= 1 && d->out_h >= 1 && d->out_w >= 1 && (long long)d->n * d->c * d->h * d->w * 4 <= 0x7fffffffLL &&
           (long long)d->n * d->o * d->out_h * d->out_w * 4 <= 0x7fffffffLL;
}

int32_t rten_small_c_conv_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const float *x, const float *w_packed, const float *bias, uint32_t flags, float *y) {
    StemArgs a = {};
    a.X = x; a.W = w_packed; a.B = bias; a.Y = y;
    a.N = d->n; a.H = d->h; a.Wd = d->w; a.O = d->o; a.OH = d->out_h; a.OW = d->out_w; a.pt = d->pads[0]; a.pl = d->pads[1];
    a.w_cs = (d->o + 3) & ~3;
    a.x_bytes = (unsigned)((long long)d->n * d->c * d->h * d->w * 4);
    a.w_bytes = (unsigned)((long long)d->c * d->kh * d->kw * a.w_cs * 4);
    a.tiles_x = (d->out_w + 15) / 16; a.tiles_y = (d->out_h + 15) / 16;
    a.act = (flags & RTEN_HIP_CONV_RELU) ? RTEN_HIP_ACT_RELU : RTEN_HIP_ACT_NONE;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * d->n;
    if (tiles > 0x7fffffffLL) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv: too many output tiles");
    hipLaunchKernelGGL((conv_small_c_f32_kernel<3, 7, 7, 2>), dim3((unsigned)tiles), dim3(256), 0, ctx->stream, a);
    RTEN_LAUNCH_CHECK(ctx, "conv_small_c_f32_kernel launch");
    return RTEN_HIP_OK;
}
