// Fast int8 GEMM / convolution path for gfx950: v_mfma_i32_32x32x32_i8 fed by 16-byte LDS-DMA.
//
// Replaces (together with int8.hip, which stays as the generic fallback): GemmExecutor<u8,i8,i32> and its packing
// (rten-gemm/src/kernels/generic.rs:274-366, packing/int8.rs), the int8 im2col (im2col.rs:264-389) and the
// front-ends conv_integer / ConvIntegerToFloat / matmul_integer (src/ops/conv.rs:421-587, matmul.rs:582-810).
//
// Integer sums are exact in any order, so -- unlike the f32 path -- the K dimension may be re-ordered.  That is
// what makes the int8 path fast on this machine:
//   * Both MFMA operands want 16 CONSECUTIVE k bytes per lane.  Operands are therefore staged once per call as
//     CHUNK-MAJOR signed bytes [K/16][rows][16 B]: weights with k = (ky, kx, c) (channels padded to 16) plus their row
//     sums; conv activations as a zero-point-PADDED channel-blocked image [N][Cp/16][H+pads][W+pads][16 B], so that
//     the im2col gather of one (tap, 16-channel) chunk is one aligned 16-byte load per pixel and needs no bounds tests
//     at all (the spatial border holds the reference's padding value for the selected pad mode).
//     Chunk-major (rather than k-contiguous rows) because the 64 lanes of one tile-DMA instruction then read ONE
//     1 KiB run (GEMM) or a few row-long runs (conv) instead of 64 pieces a row pitch apart: with a power-of-two
//     pitch those 64 pieces fall on 2-4 of the 16 L2 channels (measured: +8 % end to end on ResNet-50 int8).
//     u8 operands move to the signed domain (x ^ 0x80) during staging; (x - zp) is invariant under that shift.
//   * Main loop: tiles go L2 -> LDS with `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction), chunk-major
//     LDS image [4 chunks][rows][16 B] (the DMA's lane-linear destination), so an MFMA operand fetch is one
//     conflict-free ds_read_b128.  Three LDS stages, counted vmcnt waits, one barrier per 64-byte k-tile.
//     Per-lane offsets are loop invariant; the k advance (conv: the (ky, kx, c) chunk walk) is scalar.
//     What bounds it (tools/probes/lds_fill_rate.hip, ablation in DESIGN.md section 7): the tile DMA of a workgroup,
//     not the MFMAs -- hence the tile / tile-order rules in dispatch_fast.
//   * Zero-point algebra of the reference (simd_generic.rs:676-746): C = dot - b_zp*rowsum(A) - a_zp*colsum(B)
//     + K*a_zp*b_zp with weight sums from the staging pass; the activation-side sums are only accumulated
//     (v_dot4 on the operand fragments) when the weight zero point can be non-zero.
//   * Epilogue fuses cast_scale, bias, residual Add and Relu; i32 or f32 output straight into NCHW / row-major.
//   * A launch of this path is usually ONE wave of workgroups (a few hundred tiles for 256 CUs), so its prologue and epilogue are on the critical path one
//     for one (tools/debug/i8_trace.py on a -DRTEN_TRACE build).  Hence: all kernel-argument lines requested at entry
//     (kernarg_prefetch), index arithmetic by multiply-shift (rten_div), no arithmetic on a loaded value before the main loop (it would drain the
//     vector-memory counter in front of the first operand request), the launch's single activation zero point through the scalar cache, and the
//     single-row-term epilogue of the convolution form as a template flag (RT).
#include <cstring>

#include "internal.h"
#include "quantize.h"
#include "vecmath.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr int KT = 64;          // k bytes per tile (4 chunks of 16)
constexpr unsigned OOB = 0x80000000u;

struct FastArgs {
    const uint8_t *A;      // [M][Kp] signed rows
    const uint8_t *B;      // GEMM: [N][Kp] signed rows; conv: padded NHWC image
    const int *rsum;       // [M] sums of A rows
    const int *csum;       // [N] sums of B rows (GEMM) or NULL (conv: accumulated in-kernel when needed)
    void *C;
    const uint8_t *a_zp, *b_zp;
    const float *scale, *bias, *res;
    int M, N, Kp, Kreal;
    unsigned a_bytes, b_bytes;
    long long c_rs, c_ns;
    int Pn;
    int a_signed, b_signed, a_zp_len, b_zp_len, scale_len, relu, need_csum;
    int scale_per_row; // conv: scale[m] per output channel instead of scale[0] / scale[n]
    unsigned *stats;   // optional: min/max of the f32 outputs, accumulated for the DynamicQuantizeLinear that consumes them
    int tiles_m, tiles_n, n_fastest;
    // conv geometry (padded image)
    int conv, OW, sy, sx, Hp, Wp, Cp, KH, KW, dy, dx;
    // BQ kernels (DynamicQuantizeLinear fused into the B loader of a pointwise conv): the f32 activations, the min/max block their
    // producer left, the plane size, and where the quantizer's own outputs go
    const float *xf;
    const unsigned *in_stats;
    int HW, Cin;
    float *xs_out;
    uint8_t *xz_out;
    // QO kernels (the DynamicQuantizeLinear that CONSUMES this convolution's output runs in its epilogue, behind a grid-wide min / max):
    // the barrier block, the consumer's staged image and its geometry, the quantizer's own outputs
    unsigned *sync;
    unsigned *fault; // the context's sticky fault word (host-mapped): written only when a grid-wide wait gives up
    uint8_t *q_out;
    float *q_scale_out;
    uint8_t *q_zp_out;
    const float *q_mul_by;
    float *q_product;
    int q_cb, q_Hp, q_Wp, q_pt, q_pl, q_H, q_W, q_pad_mode;
    unsigned q_bytes;
    int debug_flags; // RTEN_HIP_DEBUG tuning switches seen by the kernel (bit 0: general zero-point algebra everywhere)
    // KS kernels (cross-workgroup K split): `ks` workgroups share one output tile, each sums its slice of the k-tiles, parks the raw int32 partial in `slab`
    // ([tile][part][wave][register quad][lane] x 16 B) and arrives on the tile's counter; the last arrival adds the others' partials to its own registers
    // (integer addition: any order gives the same bits) and runs the epilogue
    int ks;
    int *slab;
    unsigned *ks_counters;
    int no_store; // statistics-only launch (pass 1 of the recompute form of a quantized output): the epilogue runs, nothing is stored
    // exact division by the conv geometry's run-time divisors as multiply + shift (filled by launch_fast): the prologue's per-lane
    // pixel decode and the chunk table otherwise spend ~40 VALU instructions per division, on every resident wave at once
    RtenDiv d_pn, d_ow, d_cpc, d_kw, d_hw, d_qw, d_tile;
};


__device__ __forceinline__ int zp_signed(const uint8_t *zp, int idx, int is_signed) {
    if (!zp) return is_signed ? 0 : -128;
    return is_signed ? (int)(int8_t)zp[idx] : (int)zp[idx] - 128;
}
// The same in two steps: the load (raw byte, -1 = no zero-point input) is issued in the kernel's prologue, the conversion waits until the epilogue --
// any arithmetic on the loaded byte makes the compiler drain every outstanding load right there, a full memory round trip before the first operand tile
// is even requested (tools/debug/i8_trace.py: ~1000 cycles per launch).
__device__ __forceinline__ int zp_raw(const uint8_t *zp, int idx) { return zp ? (int)zp[idx] : -1; }
__device__ __forceinline__ int zp_from_raw(int raw, int is_signed) {
    if (raw < 0) return is_signed ? 0 : -128;
    return is_signed ? (int)(int8_t)(uint8_t)raw : raw - 128;
}


template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// ---------------------------------------------------------------------------------------------------------
// staging kernels
// ---------------------------------------------------------------------------------------------------------

// rows of a strided u8/i8 matrix -> chunk-major signed operand [Kp/16][rows][16 B] (+ row sums).  With khw > 1 the
// source k index is (c, tap) (OIHW weights) and the destination k index is (tap, c) with channels padded to Cp.
// `pk_nch` > 0: the few-channel PACKED destination order (conv_geom, round 6): chunk (ky, j) holds kernel columns j * pk_cpc .. + pk_cpc - 1 of row ky, byte
// col * C + channel; columns past the kernel's width and the bytes past pk_cpc * C are zero weights.
__global__ __launch_bounds__(256) void i8_pack_rows_kernel(const uint8_t *__restrict__ src, long long row_stride, long long k_stride, int K,
                                                          int C, int khw, int Cp, int Kp, int rows, unsigned flip, uint8_t *__restrict__ dst,
                                                          int *__restrict__ sums, int pk_nch = 0, int pk_cpc = 0, int pk_kw = 0) {
    const int r = blockIdx.x;
    const uint8_t *s = src + (long long)r * row_stride;
    int sum = 0;
    for (int kq = threadIdx.x * 4; kq < Kp; kq += 1024) {
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int kd = kq + b;
            int tap = kd / Cp, c = kd - tap * Cp;
            bool ok = tap < khw && c < C;
            if (pk_nch > 0) { // (Cp == 16 here; `khw` = kernel rows x kernel columns of the REAL kernel)
                const int vt = kd >> 4, slot = kd & 15, ky = vt / pk_nch, j = vt - ky * pk_nch, col = slot / C, kx = j * pk_cpc + col;
                c = slot - col * C;
                ok = ky * pk_kw < khw && col < pk_cpc && kx < pk_kw;
                tap = ky * pk_kw + kx;
            }
            const int ks = c * khw + tap;
            const unsigned v = ok ? ((unsigned)s[(long long)(ks < K ? ks : 0) * k_stride] ^ flip) & 0xffu : 0u;
            w |= v << (8 * b);
        }
        *reinterpret_cast<unsigned *>(dst + ((long long)(kq >> 4) * rows + r) * 16 + (kq & 15)) = w;
        sum = __builtin_amdgcn_sdot4((int)w, 0x01010101, sum, false);
    }
    __shared__ int red[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) sums[r] = red[0] + red[1] + red[2] + red[3];
}

// Staging of conv activations: NCHW -> padded channel-blocked signed bytes [N][Cp/16][Hp][Wp][16 B]; border = pad
// value of the selected mode, padded channels 0.  A workgroup takes 256 consecutive SOURCE pixels of one image x one
// 16-channel block: a thread loads 4 consecutive pixels (one 16-byte load when the plane size allows) of 4 consecutive
// channels, packs the 4 channel bytes of each pixel into a dword and parks them in LDS ([channel quad][pixel], one
// conflict-free ds_write_b128); the workgroup then writes the run of PADDED positions that its pixels span -- border
// pieces included -- as consecutive 16-byte pieces (1 KiB per wave instruction).  `prep()` runs AFTER the loads are in
// flight (so e.g. the statistics fold of the fused quantizer hides behind them) and returns the mapping: `.fill` =
// signed-domain border byte, `.byte(v)` = source element -> signed-domain byte.
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<uint8_t> { typedef uchar4 type; };

template <typename T, bool VEC, typename Prep>
__device__ __forceinline__ void stage_blocked_tile(const T *__restrict__ x, uint8_t *__restrict__ xp, int C, int H, int W, int Hp, int Wp, int Cp, int pt, int pl,
                                                   Prep prep) {
    __shared__ __attribute__((aligned(16))) unsigned tile[4][256];
    const int t = threadIdx.x;
    const int HW = H * W, npix = Hp * Wp;
    const int p0 = blockIdx.x * 256, n = blockIdx.y, cb = blockIdx.z;
    const int px = (t & 63) * 4, cq = (t >> 6) * 4;
    T raw[4][4]; // [channel][pixel]
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = cb * 16 + cq + j;
        const long long base = ((long long)n * C + (c < C ? c : 0)) * HW + p0 + px;
        if constexpr (VEC) { // HW % 4 == 0: the 4 pixels are all inside or all outside, and the address is 4-element aligned
            const typename Vec4<T>::type v = *reinterpret_cast<const typename Vec4<T>::type *>(x + ((c < C && p0 + px < HW) ? base : 0));
            raw[j][0] = v.x; raw[j][1] = v.y; raw[j][2] = v.z; raw[j][3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) raw[j][i] = x[(c < C && p0 + px + i < HW) ? base + i : 0];
        }
    }
    const auto map = prep();
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w[i] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned v = (cb * 16 + cq + j < C) ? map.byte(raw[j][i]) & 0xffu : 0u;
            w[i] |= v << (8 * j);
        }
    }
    *reinterpret_cast<uint4 *>(&tile[t >> 6][px]) = make_uint4(w[0], w[1], w[2], w[3]);
    __syncthreads();
    // border piece of this channel block: pad byte on real channels, 0 on the padding channels
    unsigned fw[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        fw[q] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) fw[q] |= (cb * 16 + q * 4 + j < C ? map.fill : 0u) << (8 * j);
    }
    auto padded = [&](int sp) { const int y = sp / W; return (y + pt) * Wp + (sp - y * W) + pl; };
    const int s_end = p0 + 256 < HW ? p0 + 256 : HW;
    const int begin = p0 == 0 ? 0 : padded(p0);
    const int end = s_end == HW ? npix : padded(s_end);
    uint8_t *dst = xp + ((long long)n * (Cp / 16) + cb) * npix * 16;
    for (int pp = begin + t; pp < end; pp += 256) {
        const int yp = pp / Wp, xq = pp - yp * Wp;
        const int y = yp - pt, xs = xq - pl;
        uint4 v = make_uint4(fw[0], fw[1], fw[2], fw[3]);
        if ((unsigned)y < (unsigned)H && (unsigned)xs < (unsigned)W) {
            const int sl = y * W + xs - p0; // in [0, 256): the padded run [begin, end) covers exactly this workgroup's pixels
            v = make_uint4(tile[0][sl], tile[1][sl], tile[2][sl], tile[3][sl]);
        }
        *reinterpret_cast<uint4 *>(dst + (long long)pp * 16) = v;
    }
}

// Few-channel PACKED staging (round 6; conv_geom): a convolution with C <= 4 input channels (the 7x7 / stride 2 stem of ResNet-50: C = 3) staged as a 16-channel
// block moves 16 bytes per pixel for 3 of data and multiplies 13 zeros per tap (K = 49 x 16 = 784 for 147 real terms).  Here a 16-byte chunk holds `cpc` = 16 / C
// horizontally adjacent kernel COLUMNS x C channels of one kernel row, and every OUTPUT column gets its own `nch` = ceil(KW / cpc) chunks per padded input row
// (columns ox * stride - pad + j * cpc + col): image [N][Hp][OW * nch][16 B], which the unchanged kernel walks as a convolution with 16 channels, KW = nch and
// stride nch over "virtual columns" -- K = KH x nch x 16 (224 for the stem).  A thread writes one chunk: all its (clamped) loads first, then `prep()`.
template <typename T, int C, int NCH, typename Prep>
__device__ __forceinline__ void stage_packed_chunks(const T *__restrict__ x, uint8_t *__restrict__ xp, int H, int W, int Hp, int OW, int KW, int sx, int dx, int pt, int pl,
                                                    Prep prep) {
    // a thread owns one OUTPUT column of one padded input row: the NCH chunks of that column (32 consecutive bytes for the stem); grid = (OW / 128, Hp / 2, N).  The
    // channel count and the chunk count are compile-time, so the slot -> (kernel column, channel) map costs nothing; every load is issued (clamped) before prep().
    constexpr int CPC = 16 / C;
    // (256 threads = two padded rows x 128 output columns: prep()'s statistics fold is written for 256-thread workgroups)
    const int ox = blockIdx.x * 128 + (threadIdx.x & 127), ypr = blockIdx.y * 2 + (threadIdx.x >> 7), n = blockIdx.z;
    const int oxc = ox < OW ? ox : OW - 1, yp = ypr < Hp ? ypr : Hp - 1;
    const int y = yp - pt, x0 = oxc * sx - pl; // kernel column kx reads input column ox * stride - pad + kx * dilation
    const bool row_in = (unsigned)y < (unsigned)H;
    const long long plane = (long long)H * W;
    const T *row = x + (long long)n * C * plane + (long long)(row_in ? y : 0) * W;
    T raw[NCH * CPC][C];
#pragma unroll
    for (int kx = 0; kx < NCH * CPC; kx++) {
        const int xx = x0 + kx * dx;
        const bool in = kx < KW && row_in && (unsigned)xx < (unsigned)W;
#pragma unroll
        for (int ch = 0; ch < C; ch++) raw[kx][ch] = row[in ? ch * plane + xx : 0];
    }
    const auto map = prep();
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int col = 0; col < CPC; col++) {
            const int kx = j * CPC + col, xx = x0 + kx * dx;
            const bool real = kx < KW;                                        // a kernel column (else: a zero weight multiplies the byte, and the byte is 0)
            const bool in = real && row_in && (unsigned)xx < (unsigned)W;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                constexpr int dummy = 0; (void)dummy;
                const int slot = col * C + ch;
                const unsigned v = in ? map.byte(raw[kx][ch]) & 0xffu : (real ? map.fill : 0u);
                w[slot >> 2] |= v << (8 * (slot & 3));
            }
        }
        if (ox < OW && ypr < Hp) *reinterpret_cast<uint4 *>(xp + ((((long long)n * Hp + yp) * OW + ox) * NCH + j) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Small feature maps (fewer than 128 pixels per plane, e.g. 7x7): 64 consecutive PADDED positions x 64 channels per
// workgroup, one pixel per lane, so the lanes stay busy where the 256-pixel tile above would idle.
template <typename T, typename Prep>
__device__ __forceinline__ void stage_blocked_tile_small(const T *__restrict__ x, uint8_t *__restrict__ xp, int C, int H, int W, int Hp, int Wp, int Cp, int pt, int pl,
                                                Prep prep) {
    __shared__ uint8_t tile[64][64 + 16];
    const int t = threadIdx.x;
    const int pp0 = blockIdx.x * 64, n = blockIdx.y;
    const int npix = Hp * Wp;
    const int pl_ = t & 63, pp = pp0 + pl_;
    const int yp = pp / Wp, xq = pp - yp * Wp;
    const int y = yp - pt, xs = xq - pl;
    const bool in = pp < npix && (unsigned)y < (unsigned)H && (unsigned)xs < (unsigned)W;
    const long long src0 = ((long long)n * C * H + (in ? y : 0)) * W + (in ? xs : 0); // + c * H * W
    const int c0 = blockIdx.z * 64; // one 64-channel slab per workgroup: small feature maps still give thousands of workgroups
    // thread -> (pixel, 4 consecutive channels) per pass.  All 16 loads are issued first (clamped address, no branch
    // around a load), then converted and packed: one dword per pass into the LDS tile.
    T raw[16];
#pragma unroll
    for (int pass = 0; pass < 4; pass++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c = c0 + pass * 16 + (t >> 6) * 4 + b;
            raw[pass * 4 + b] = x[(in && c < C) ? src0 + (long long)c * H * W : 0];
        }
    const auto map = prep();
    const unsigned fill = map.fill;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int cl = pass * 16 + (t >> 6) * 4;
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c = c0 + cl + b;
            const unsigned v = (in && c < C) ? map.byte(raw[pass * 4 + b]) & 0xffu : (c < C ? fill : 0u);
            w |= v << (8 * b);
        }
        *reinterpret_cast<unsigned *>(&tile[pl_][cl]) = w;
    }
    __syncthreads();
    const int px = t & 63, ch = t >> 6;
    if (pp0 + px < npix && c0 + ch * 16 < Cp)
        *reinterpret_cast<uint4 *>(xp + (((long long)n * (Cp / 16) + (c0 / 16 + ch)) * npix + pp0 + px) * 16) = *reinterpret_cast<const uint4 *>(&tile[px][ch * 16]);
}

struct FlipMap {
    unsigned fill, flip;
    __device__ __forceinline__ unsigned byte(uint8_t b) const { return ((unsigned)b ^ flip) & 0xffu; }
};
struct QuantMap {
    unsigned fill;
    float inv_scale;
    int zp;
    __device__ __forceinline__ unsigned byte(float f) const { return dql::quant_u8(f, inv_scale, zp) ^ 0x80u; }
};

template <int KIND> // 0 = small maps, 1 = 256-pixel tiles with scalar loads, 2 = with 4-pixel vector loads
__global__ __launch_bounds__(256) void i8_nhwc_pad_kernel(const uint8_t *__restrict__ x, uint8_t *__restrict__ xp, int C, int H, int W, int Hp,
                                                         int Wp, int Cp, int pt, int pl, unsigned flip, const uint8_t *__restrict__ x_zp,
                                                         int x_signed, int pad_mode) {
    auto prep = [&]() {
        int pad_s = 0;
        if (pad_mode == RTEN_HIP_PAD_ZERO_POINT) pad_s = zp_signed(x_zp, 0, x_signed);
        else if (pad_mode == RTEN_HIP_PAD_RAW0_U8) pad_s = -128;
        return FlipMap{(unsigned)pad_s & 0xffu, flip};
    };
    if constexpr (KIND == 0) stage_blocked_tile_small(x, xp, C, H, W, Hp, Wp, Cp, pt, pl, prep);
    else stage_blocked_tile<uint8_t, KIND == 2>(x, xp, C, H, W, Hp, Wp, Cp, pt, pl, prep);
}

// DynamicQuantizeLinear's quantize sweep fused with the staging above: f32 NCHW -> u8 codes (bit-identical to
// quantize.hip: same scale / zero-point algebra, same to_int_round + saturate) written as padded channel-blocked
// signed bytes.  The min/max fold runs while the tile's loads are in flight.
// the scalar Mul(x_scale, w_scale) nodes that follow a DynamicQuantizeLinear in ort-quantized graphs (one per convolution that reads the
// quantized tensor: a stage's shortcut and first 1x1 convolution share one): product[i] = scale * mul_by[i][0]
constexpr int kMaxProducts = 4;
struct ScaleProducts {
    int count;
    const float *mul_by[kMaxProducts];
    float *product[kMaxProducts];
};

template <int C, int NCH>
__global__ __launch_bounds__(256) void i8_pad_packed_kernel(const uint8_t *__restrict__ x, uint8_t *__restrict__ xp, int H, int W, int Hp, int OW, int KW, int sx, int dx, int pt,
                                                           int pl, unsigned flip, const uint8_t *__restrict__ x_zp, int x_signed, int pad_mode) {
    auto prep = [&]() {
        int pad_s = 0;
        if (pad_mode == RTEN_HIP_PAD_ZERO_POINT) pad_s = zp_signed(x_zp, 0, x_signed);
        else if (pad_mode == RTEN_HIP_PAD_RAW0_U8) pad_s = -128;
        return FlipMap{(unsigned)pad_s & 0xffu, flip};
    };
    stage_packed_chunks<uint8_t, C, NCH>(x, xp, H, W, Hp, OW, KW, sx, dx, pt, pl, prep);
}

template <int KIND>
__global__ __launch_bounds__(256) void i8_quantize_stage_kernel(const float *__restrict__ x, const float *__restrict__ ws, int nparts, uint8_t *__restrict__ xp,
                                                               int C, int H, int W, int Hp, int Wp, int Cp, int pt, int pl, int pad_mode,
                                                               float *scale_out, uint8_t *zp_out, const ScaleProducts sp) {
    // explicit arguments: 0x58 bytes + ScaleProducts = 160 bytes, three lines (the tail of them is first touched AFTER the tile has arrived: on the critical path)
    kernarg_prefetch<0x58 + (int)sizeof(ScaleProducts)>();
    auto prep = [&]() {
        float x_min, x_max;
        if (nparts < 0) dql::block_minmax_slots(reinterpret_cast<const unsigned *>(ws), x_min, x_max); // producer-accumulated statistics
        else dql::block_minmax(ws, nparts, x_min, x_max);
        const dql::QParams q = dql::dql_params(x_min, x_max);
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
            *scale_out = q.scale;
            *zp_out = (uint8_t)q.zp;
#pragma unroll
            for (int i = 0; i < kMaxProducts; i++)
                if (i < sp.count) *sp.product[i] = q.scale * sp.mul_by[i][0];
        }
        int pad_s = 0; // signed-domain padding value
        if (pad_mode == RTEN_HIP_PAD_ZERO_POINT) pad_s = q.zp - 128;
        else if (pad_mode == RTEN_HIP_PAD_RAW0_U8) pad_s = -128;
        return QuantMap{(unsigned)pad_s & 0xffu, q.inv_scale, q.zp};
    };
    if constexpr (KIND == 0) stage_blocked_tile_small(x, xp, C, H, W, Hp, Wp, Cp, pt, pl, prep);
    else stage_blocked_tile<float, KIND == 2>(x, xp, C, H, W, Hp, Wp, Cp, pt, pl, prep);
}

// ... and DynamicQuantizeLinear's quantize sweep into the few-channel packed image (same prep as i8_quantize_stage_kernel: same codes)
template <int C, int NCH>
__global__ __launch_bounds__(256) void i8_quantize_packed_kernel(const float *__restrict__ x, const float *__restrict__ ws, int nparts, uint8_t *__restrict__ xp, int H, int W, int Hp,
                                                                int OW, int KW, int sx, int dx, int pt, int pl, int pad_mode, float *scale_out, uint8_t *zp_out,
                                                                const ScaleProducts sp) {
    auto prep = [&]() {
        float x_min, x_max;
        if (nparts < 0) dql::block_minmax_slots(reinterpret_cast<const unsigned *>(ws), x_min, x_max);
        else dql::block_minmax(ws, nparts, x_min, x_max);
        const dql::QParams q = dql::dql_params(x_min, x_max);
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
            *scale_out = q.scale;
            *zp_out = (uint8_t)q.zp;
#pragma unroll
            for (int i = 0; i < kMaxProducts; i++)
                if (i < sp.count) *sp.product[i] = q.scale * sp.mul_by[i][0];
        }
        int pad_s = 0;
        if (pad_mode == RTEN_HIP_PAD_ZERO_POINT) pad_s = q.zp - 128;
        else if (pad_mode == RTEN_HIP_PAD_RAW0_U8) pad_s = -128;
        return QuantMap{(unsigned)pad_s & 0xffu, q.inv_scale, q.zp};
    };
    stage_packed_chunks<float, C, NCH>(x, xp, H, W, Hp, OW, KW, sx, dx, pt, pl, prep);
}

// (channel count x chunks per output column: the instantiations conv_geom admits)
#define RTEN_PACKED_DISPATCH(KERNEL, c, nch, grid, ...)                                                                                  \
    do {                                                                                                                                 \
        switch ((c) * 4 + (nch)) {                                                                                                       \
        case 1 * 4 + 1: hipLaunchKernelGGL((KERNEL<1, 1>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                         \
        case 1 * 4 + 2: hipLaunchKernelGGL((KERNEL<1, 2>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                         \
        case 2 * 4 + 1: hipLaunchKernelGGL((KERNEL<2, 1>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                         \
        case 2 * 4 + 2: hipLaunchKernelGGL((KERNEL<2, 2>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                         \
        case 3 * 4 + 1: hipLaunchKernelGGL((KERNEL<3, 1>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                         \
        case 3 * 4 + 2: hipLaunchKernelGGL((KERNEL<3, 2>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                         \
        case 4 * 4 + 1: hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                         \
        default: hipLaunchKernelGGL((KERNEL<4, 2>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;                                \
        }                                                                                                                                \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------------------------
// RES: the epilogue adds a residual tile (prefetched into 16 * TM * TN registers); a separate instantiation so that the
// launches without one keep the smaller register footprint (one more workgroup per CU on the 128 x 128 tile).
// KTK = k bytes per tile (a multiple of 64): one barrier, one counted wait and one trip round the loop per KTK / 32 MFMA steps.
// The i8 matrix pipe retires a 32x32x32 step in 32 cycles -- sixteen times the f32 rate -- so with 64-byte k-tiles a wave's two
// MFMAs (64 cycles) sat behind ~600 cycles of barrier / wait / DMA issue / loop overhead (tools/probes/kloop.hip prices that
// overhead for the f32 loop); the long-K launches of stages 2-3 therefore take 256-byte k-tiles.  All LDS is ONE dynamic array.
extern __shared__ __attribute__((aligned(16))) uint8_t i8_smem[];

// KG = k-groups: the workgroup has KG x 4 waves; group g takes k-tiles g, g + KG, ... into accumulators of its own, and the
// groups' partial sums are added through LDS before the (group 0) epilogue -- exact, integer addition is order-free.  An
// under-filled launch (stage 3-4 of ResNet-50 at batch 32: ~200 tiles of 64 x 64 for 256 CUs) otherwise runs ONE wave per
// SIMD, and every latency of its k-loop (barrier, DMA issue, LDS fragment reads, dependent MFMAs: ~550 cycles per 64-byte
// k-tile, independent of where the operands come from -- profiles/r05/int8_kloop_ablation.txt) is exposed 72 times in a row.
// BQ = the B operand is QUANTIZED ON LOAD from the f32 activation tensor (pointwise stride-1 unpadded convolutions of a dynamically
// quantized graph): DynamicQuantizeLinear's parameters come from the min/max block the producing conv accumulated, a thread
// reads 16 channels of one pixel (lanes = consecutive pixels: 256-byte runs), converts with dql::quant_u8 and writes the
// 16-byte chunk piece the DMA would have delivered (conflict-free ds_write_b128).  The staging launch, its 1 B/element write
// and the staged image's read disappear; A still arrives by LDS-DMA.  Same codes, same sums: bit-identical.
// QO = the output is QUANTIZED IN THE EPILOGUE for the one convolution that consumes it (DynamicQuantizeLinear -> ConvInteger of the
// next layer, src/ops/quantize.rs:352-436): the finished f32 values stay in the accumulator registers, every workgroup publishes its
// min / max (the ordered-uint slots of the producer-statistics scheme) and arrives on a grid-wide counter barrier; behind it each
// workgroup folds the slots -- min / max are order free, so these are the statistics of the two-sweep operator -- derives the same
// scale / zero point, converts with the same dql::quant_u8 and writes the codes straight into the consumer's staged image
// (channel-blocked, padded; border pieces included).  The f32 tensor is written only if somebody else needs it (p.C != NULL).
// Every workgroup of the grid must be resident at once (the host checks the occupancy before it launches this form).
constexpr unsigned kSyncCtlWords = 16;    // exchange block: word 0 = departures, word 8 = time-out flag, then kSyncGranules 8-byte {min, max} granules
constexpr unsigned kSyncGranules = 2048;  // >= the workgroups the device holds at once (8 per compute unit x 256)
constexpr unsigned kSyncWords = kSyncCtlWords + 2 * kSyncGranules;
constexpr unsigned long long kGranuleReset = 0x00000000ffffffffull; // {min = 0xffffffff, max = 0}: what a reset leaves, never a real pair
constexpr unsigned kSyncSpinLimit = 1u << 17;
// KS = cross-workgroup K split (round 6): the launches of stages 2-3 at batch 32 are 52-392 tiles for 256 compute units with 16-72 k-tiles each -- a
// k-loop of 6-15 thousand cycles on a chip that is mostly empty.  With KS the grid is tiles x p.ks workgroups; part q of a tile walks k-tiles
// [q * nkt / ks, (q + 1) * nkt / ks), parks its raw int32 accumulators (write-through stores, as the f32 split-K slab) and arrives on the tile's counter;
// the last arrival reads the other parts back (L2-bypassing loads), adds them to its own registers and runs the unchanged epilogue.  Integer sums are
// exact in any order: same bits whichever workgroup arrives last (tests: 200 launches, every split count).
// QO == 2 = the RECOMPUTE form of a quantized output (round 6): the launch before this one ran the same convolution with `no_store` and left the output's
// min / max in the statistics block (p.in_stats); this launch folds the block in its prologue (as a BQ kernel does), computes the tile again and writes
// the codes straight into the consumer's staged image.  No grid-wide exchange, no residency requirement, no time-out: replicas side by side ("lanes") may
// use it, where the i8 pipe is 3-6 % busy and the step is bound by HBM bytes -- the f32 tensor of a single-consumer edge (4 B written + 4 B read back per
// element) never exists.  Same statistics (min / max are order-free), same dql_params, same quant_u8: the codes of the two-launch sequence.
template <int BM, int BN, int NSTAGE, bool RES, int KTK = 64, int KG = 1, bool BQ = false, int QO = 0, bool RT = false, bool KS = false>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void igemm_i8_fast_kernel(const FastArgs p) {
    static_assert(!KS || (KG == 1 && !BQ && !QO && !RES && RT && KTK == 64), "cross-workgroup K split: the plain single-row-term convolution form only");
    static_assert(!BQ || (KG == 1 && KTK == 64), "quantize-on-load: one k-group, 64-byte k-tiles");
    static_assert(!(BQ && QO), "quantize-on-load and quantized output are separate instantiations");
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RA = BM / 64, RB = BN / 64;      // DMA instructions per chunk (64 rows each)
    constexpr int CPW = KTK / 64;                  // chunks per wave per tile: wave w moves chunks w, w + 4, ... of the tile
    constexpr int PER_TILE = (RA + RB) * CPW;
    constexpr int SUB = (BM + BN) * KTK;           // bytes of one k-tile
    constexpr int STAGE = SUB * KG;                // a stage holds one k-tile per group
    uint8_t *const smem = i8_smem;
    kernarg_prefetch<(int)sizeof(FastArgs)>(); // (7 lines)

    const int t = threadIdx.x, lane = t & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(t >> 6); // 0 .. 4 KG - 1
    const int kg = wave_all >> 2;                                // k-group of this wave
    const int wave = wave_all & 3;                               // wave within the group: its chunk slot and its quadrant of the tile
    const int wq = wave;
    const int l31 = lane & 31, half = lane >> 5;
#ifdef RTEN_TRACE // build.sh -DRTEN_TRACE: cycle stamps at the phase boundaries, printed by one wave (tools/debug/i8_trace.py)
    unsigned long long tr_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = __builtin_readcyclecounter();
    const unsigned long long tr_c0 = tr_prev, tr_r0 = __builtin_amdgcn_s_memrealtime(); // (100 MHz: cycles per tick x 100 = the shader clock in MHz during this workgroup)
#define I8_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tr_seg[i] += now_ - tr_prev; tr_prev = now_; }
#else
#define I8_STAMP(i)
#endif
    int tile;
    [[maybe_unused]] int ks_part = 0;
    {
        const int nt = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, q = nt >> 3, r = nt & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        if constexpr (KS) { // an XCD's run of consecutive indices walks neighbouring tiles of ONE K slice: they share that slice of the operand panels
            const int ntiles = p.tiles_m * p.tiles_n;
            ks_part = tile / ntiles;
            tile -= ks_part * ntiles;
        }
    }
    const int tdiv = p.n_fastest ? p.tiles_n : p.tiles_m, tq = rten_div(tile, p.d_tile), tr = tile - tq * tdiv;
    const int bm = p.n_fastest ? tq : tr, bn = p.n_fastest ? tr : tq;
    const int m0 = bm * BM, n0 = bn * BN;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(BQ ? (void *)p.xf : (void *)p.B, 0, (int)p.b_bytes, 0x00020000);

    // loop-invariant per-lane row bases (bytes)
    unsigned a_voff[RA], b_voff[RB];
#pragma unroll
    for (int j = 0; j < RA; j++) {
        const int m = m0 + j * 64 + lane;
        a_voff[j] = m < p.M ? (unsigned)m * 16u : OOB;
    }
#pragma unroll
    for (int j = 0; j < RB; j++) {
        const int n = n0 + j * 64 + lane;
        if (n < p.N) {
            if (RT || p.conv) { // (RT kernels are convolutions by construction: the GEMM form is not compiled into them)
                const int nb = rten_div(n, p.d_pn), np = n - nb * p.Pn;
                const int oy = rten_div(np, p.d_ow), ox = np - oy * p.OW;
                b_voff[j] = (unsigned)(((nb * (p.Cp / 16) * p.Hp + oy * p.sy) * p.Wp + ox * p.sx) * 16);
            } else {
                b_voff[j] = (unsigned)n * 16u;
            }
        } else {
            b_voff[j] = OOB;
        }
    }

    I8_STAMP(0) // kernel arguments, tile decode, per-lane row / pixel bases
    // ---- this wave's chunk walk: chunk index wave, wave + 4, ... of the K axis (16-byte chunks).
    // The B-side scalar offset of a chunk -- conv: its (ky, kx, c16) position on the padded image -- comes from a table that the
    // workgroup builds in LDS before the loop (one division pair per chunk, once), and everything the loop derives from the
    // walk is kept in scalar registers: the previous form carried an odometer in vector registers and paid ~100 VALU + SALU
    // instructions per k-tile, waterfall loops around the DMA included, for two 32-cycle MFMAs (counters: profiles/r05).
    const int nchunks = p.Kp / 16;
    const int nkt = (p.Kp + KTK - 1) / KTK;
    [[maybe_unused]] const int kt_first = KS ? (int)((long long)ks_part * nkt / p.ks) : 0; // KS: this part's slice of the k-tiles
    const int nit = KS ? (int)((long long)(ks_part + 1) * nkt / p.ks) - kt_first : (nkt + KG - 1) / KG;   // loop trips: KG k-tiles per trip
    const int tchunks = ((nkt + KG - 1) / KG + NSTAGE) * KG * (KTK / 16); // chunks the walk can name (>= nchunks; the tail is dead)
    int *const btab = reinterpret_cast<int *>(smem + NSTAGE * STAGE);
    {
        const int cpc = (RT || p.conv) ? p.Cp / 16 : 1; // chunks per tap
        for (int c = t; c < tchunks; c += 256 * KG) {
            int off = -1; // dead chunk: K padding
            if (c < nchunks) {
                if (RT || p.conv) {
                    const int tap = rten_div(c, p.d_cpc), cc = c - tap * cpc, ky = rten_div(tap, p.d_kw), kx = tap - ky * p.KW;
                    if (ky < p.KH) off = ((cc * p.Hp + ky * p.dy) * p.Wp + kx * p.dx) * 16;
                } else {
                    off = c * 16 * p.N;
                }
            }
            btab[c] = off;
        }
    }
    __syncthreads();
    I8_STAMP(1) // chunk -> offset table in LDS + barrier
    // ---- BQ: DynamicQuantizeLinear parameters from the producer's statistics (every workgroup folds the 256 slots; min / max are
    // order-free), the quantizer's outputs, and this thread's pieces of the B tile
    [[maybe_unused]] float q_scale = 0.f, q_inv = 0.f;
    [[maybe_unused]] int q_zp = 0;
    constexpr int NP = BQ ? 4 * BN / 256 : 1; // 16-byte pieces (16 channels of one pixel) per thread per k-tile
    [[maybe_unused]] unsigned q_voff = OOB;   // byte offset of (image, channel 0, pixel) of this thread's pixel
    [[maybe_unused]] int q_slot0 = 0, q_px = 0;
    if constexpr (BQ || QO == 2) {
        float a = __builtin_inff(), b = -__builtin_inff();
        a = dql::ord2f(p.in_stats[t & 255]);
        b = dql::ord2f(p.in_stats[dql::kStatSlots + (t & 255)]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
        float *red = reinterpret_cast<float *>(smem + NSTAGE * STAGE + tchunks * 4); // 8 floats behind the chunk table (the launcher sizes the allocation for them)
        if (lane == 0 && wave_all < 4) { red[wave_all] = a; red[4 + wave_all] = b; }
        __syncthreads();
        const float mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3])), mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        const dql::QParams q = dql::dql_params(mn, mx);
        q_scale = q.scale; q_inv = q.inv_scale; q_zp = q.zp;
        if (BQ && t == 0 && blockIdx.x == 0) {
            if (p.xs_out) *p.xs_out = q.scale;
            if (p.xz_out) *p.xz_out = (uint8_t)q.zp;
        }
    }
    if constexpr (BQ) {
        q_px = t % BN;
        q_slot0 = __builtin_amdgcn_readfirstlane(t / BN); // wave-uniform: BN is a multiple of 64
        const int n = n0 + q_px;
        if (n < p.N) {
            const int img = rten_div(n, p.d_hw), pp = n - img * p.HW;
            q_voff = (unsigned)((img * p.Cin) * p.HW + pp) * 4u;
        }
    }
    int ch_idx = __builtin_amdgcn_readfirstlane(wave_all + (KS ? kt_first * (KTK / 16) : 0)); // chunk index of the next piece to issue: wave_all, wave_all + 4 KG, ...
    // the table entries of this wave's next 64 pieces ride in one vector register (lane i = i-th piece from here) and are picked
    // with v_readlane: no LDS round trip on the issue path.  Refilled every 64 pieces.
    auto bo_fill = [&](int first) {
        const int c = first + 4 * KG * lane;
        return c < tchunks ? btab[c] : -1;
    };
    int bo_vec = bo_fill(ch_idx);
    int bo_pos = 0;
    const unsigned a_step = 16u * (unsigned)p.M;
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    auto issue_tile = [&](int stage) {
#pragma unroll
        for (int q = 0; q < CPW; q++) { // LDS chunk slot wave + 4 q of the tile <- this wave's next chunk (its walk is every 4th chunk)
            uint8_t *As = smem + stage * STAGE + kg * SUB + (wave + 4 * q) * BM * 16;
            uint8_t *Bs = smem + stage * STAGE + kg * SUB + BM * KTK + (wave + 4 * q) * BN * 16;
            const int bo = __builtin_amdgcn_readlane(bo_vec, bo_pos);
            const bool a_live = ch_idx < nchunks, b_live = bo >= 0;
            const unsigned a_soff = a_live ? (unsigned)ch_idx * a_step : 0u;
            const unsigned b_soff = b_live ? (unsigned)bo : 0u;
#pragma unroll
            for (int j = 0; j < RA; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(As + j * 1024), 16, (int)(a_live ? a_voff[j] : OOB), (int)a_soff, 0, 0);
            if constexpr (!BQ) {
#pragma unroll
                for (int j = 0; j < RB; j++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(Bs + j * 1024), 16, (int)(b_live ? b_voff[j] : OOB), (int)b_soff, 0, 0);
            }
            ch_idx += 4 * (CPW == 1 ? KG : 1);
            if (++bo_pos == 64) {
                bo_vec = bo_fill(ch_idx);
                bo_pos = 0;
            }
        }
    };

    // BQ: the f32 values of k-tile kt (its 64 channels) for this thread's NP pieces -> registers; later -> codes -> LDS
    [[maybe_unused]] float qv[NP][16];
    [[maybe_unused]] auto q_load = [&](int kt) {
#pragma unroll
        for (int r = 0; r < NP; r++) {
            const int slot = q_slot0 + r * (256 / BN);
            const unsigned soff = (unsigned)((kt * 64 + slot * 16) * p.HW) * 4u; // channel kt*64 + slot*16 (+ j): beyond Cin -> out of range -> 0, never used
#pragma unroll
            for (int j = 0; j < 16; j++)
                qv[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, (int)q_voff, (int)(soff + (unsigned)(j * p.HW) * 4u), 0));
        }
    };
    [[maybe_unused]] auto q_store = [&](int stage) {
#pragma unroll
        for (int r = 0; r < NP; r++) {
            const int slot = q_slot0 + r * (256 / BN);
            unsigned w[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                w[d] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) w[d] |= ((dql::quant_u8(qv[r][4 * d + j], q_inv, q_zp) ^ 0x80u) & 0xffu) << (8 * j);
            }
            *reinterpret_cast<uint4 *>(smem + stage * STAGE + BM * KTK + (slot * BN + q_px) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };

    const int wm0 = (wq / WN) * (BM / WM), wn0 = (wq % WN) * (BN / WN);
    i32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
    int cs[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) cs[j] = 0;

    auto compute_tile = [&](int stage) {
        const uint8_t *As = smem + stage * STAGE + kg * SUB + (wm0 + l31) * 16;
        const uint8_t *Bs = smem + stage * STAGE + kg * SUB + BM * KTK + (wn0 + l31) * 16;
        constexpr int NS = KTK / 32; // MFMA k-steps per tile; all fragment reads of a 64-byte tile are issued before its first MFMA
        i32x4 af[NS < 2 ? NS : 2][TM], bf[NS < 2 ? NS : 2][TN];
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += 2) {
#pragma unroll
            for (int u = 0; u < 2 && s0 + u < NS; u++) {
                const int s = s0 + u;
#pragma unroll
                for (int i = 0; i < TM; i++) af[u][i] = *reinterpret_cast<const i32x4 *>(As + ((2 * s + half) * BM + i * 32) * 16);
#pragma unroll
                for (int j = 0; j < TN; j++) bf[u][j] = *reinterpret_cast<const i32x4 *>(Bs + ((2 * s + half) * BN + j * 32) * 16);
            }
#pragma unroll
            for (int u = 0; u < 2 && s0 + u < NS; u++) {
                if (!RT && p.need_csum) {
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++) cs[j] = __builtin_amdgcn_sdot4(bf[u][j][q], 0x01010101, cs[j], false);
                }
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[u][i], bf[u][j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- epilogue operands, requested BEFORE the main loop so that their latency hides behind it (these short-K
    // products are otherwise epilogue-latency bound): per-row constants by threads 0..BM-1 (parked in LDS once the
    // stage buffers are free), per-column constants and the residual tile straight into registers.  They are older
    // than every LDS-DMA load, so the counted vmcnt waits of the main loop cover them.
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)p.C, 0, 0x7ffffffc, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? (const void *)p.res : (const void *)p.C), 0, 0x7ffffffc, 0x00020000);
    const unsigned rs4 = (unsigned)p.c_rs << 2;
    const int mb = m0 + wm0 + 4 * half;
    int rc_rsum = 0, rc_az_raw = -1;
    float rc_bias = 0.f, rc_srow = 0.f;
    if (t < BM) { // (threads of k-group 0)
        const int m = m0 + t < p.M ? m0 + t : 0;
        rc_rsum = p.rsum[m];
        if constexpr (!RT) rc_az_raw = zp_raw(p.a_zp, p.a_zp_len > 1 ? m % p.a_zp_len : 0); // GEMM: period < M cycles the zero points (matmul.rs:266-280)
        rc_bias = p.bias ? p.bias[m] : 0.f;
        rc_srow = (p.scale && p.scale_per_row) ? p.scale[m] : 0.f;
        if constexpr (BQ) rc_srow = q_scale * rc_srow; // Mul(x_scale, w_scale[m])
    }
    unsigned basev[TN], bzv[TN], csv[TN];
    [[maybe_unused]] int bz_raw[TN]; // (converted after the main loop: zp_from_raw)
    // RT: ONE activation zero point for the whole launch.  A wave-uniform byte load comes back through a vector register and v_readfirstlane, i.e. with a
    // full drain of the vector-memory counter right here; the aligned word around the byte through the scalar cache instead is only waited for where it is used.
    [[maybe_unused]] unsigned bz_word = 0;
    [[maybe_unused]] int bz_shift = -1;
    if constexpr (RT && !BQ) {
        if (p.b_zp) {
            const unsigned long long addr = (unsigned long long)p.b_zp;
            bz_word = *(const __attribute__((address_space(4))) unsigned *)(addr & ~3ull);
            bz_shift = (int)(addr & 3ull) * 8;
        }
    }
    float scv[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn0 + j * 32 + l31;
        const bool cok = n < p.N;
        const int nn = cok ? n : 0;
        const int nb = rten_div(nn, p.d_pn), np = nn - nb * p.Pn;
        basev[j] = cok ? (unsigned)((long long)nb * p.c_ns + np) << 2 : OOB; // byte offset of (row 0, column n); rows ride the scalar offset
        if constexpr (!RT) bz_raw[j] = zp_raw(p.b_zp, p.b_zp_len == 1 ? 0 : nn);
        bzv[j] = 0u;
        csv[j] = (!RT && p.csum) ? (unsigned)p.csum[nn] : 0u;
        scv[j] = (p.scale && !p.scale_per_row) ? p.scale[p.scale_len == 1 ? 0 : nn] : 0.f;
        if constexpr (BQ) {
            bzv[j] = (unsigned)(q_zp - 128);   // the activation zero point in the signed domain
            scv[j] = q_scale * scv[j];         // Mul(x_scale, w_scale)
        }
    }
    // row r of 32-row block i sits at scalar byte offset (mb_u + i*32 + acc_row(r)) * rs4, with mb_u wave-uniform and the lane's
    // half (rows +4) folded into the vector offset
    const unsigned half_off = (unsigned)(4 * half) * rs4;
    const int mb_u = m0 + (wave / WN) * (BM / WM);
    I8_STAMP(2) // quantizer parameters (BQ), zero points / scales / row constants of this lane's columns and rows
    [[maybe_unused]] float rr[RES ? TM : 1][RES ? TN : 1][16];
    if constexpr (RES) {
        if (KG == 1 || kg == 0) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int m = mb + i * 32 + acc_row(r);
                        const unsigned vo = (m < p.M && basev[j] != OOB) ? basev[j] + half_off : OOB;
                        rr[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, (int)vo, (int)((unsigned)(mb_u + i * 32 + acc_row(r)) * rs4), 0));
                    }
        }
    }

    I8_STAMP(3) // residual tile requested (RES)
    int stage = 0;
    if constexpr (BQ) {
        // A by DMA two tiles ahead (as below); B: the values of tile it + 1 are in flight in registers while tile it is multiplied,
        // and are converted and written to their stage at the top of the next trip, before the barrier that publishes it.
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; i++) issue_tile(i);
        q_load(0);
        wait_vmcnt<0>();
        q_store(0);
        q_load(1);
        I8_STAMP(4) // first operand tiles requested (and the first B tile converted)
        for (int it = 0; it < nit; it++) {
            wait_vmcnt<0>(); // A of tile it (and it + 1), B values of tile it + 1
            const int stn = stage == NSTAGE - 1 ? 0 : stage + 1;
            q_store(stn);    // stage of tile it + 1: last read two trips ago
            __builtin_amdgcn_s_barrier();
            const int stp = stage == 0 ? NSTAGE - 1 : stage - 1;
            issue_tile(stp); // A of tile it + 2
            q_load(it + 2);
            compute_tile(stage);
            stage = stn;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; i++) issue_tile(i);
        // (A two-fragment-set software pipeline of the LDS reads under the MFMAs was measured and is slower.)
        I8_STAMP(4) // first operand tiles requested
        for (int it = 0; it < nit; it++) {
            wait_vmcnt<PER_TILE *(NSTAGE - 2)>();
            __builtin_amdgcn_s_barrier();
            const int stp = stage == 0 ? NSTAGE - 1 : stage - 1;
            issue_tile(stp);
            compute_tile(stage);
            stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        }
    }
    I8_STAMP(5) // k-loop
    wait_vmcnt<0>();
    __syncthreads(); // every wave is done with the stage buffers
    if constexpr (KS) {
        constexpr int NQ = TM * TN * 4; // 16-byte pieces per lane
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)(p.slab + ((long long)tile * p.ks) * (BM * BN)), 0, (int)((unsigned)p.ks * (BM * BN) * 4u), 0x00020000);
        const unsigned lane_off = (unsigned)(wave * (NQ * 64) + lane) * 16u;
        const unsigned part_bytes = (unsigned)(BM * BN) * 4u;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) // aux 17 = sc0 sc1: the store writes through to memory (the eight XCD L2s are not coherent with each other inside a kernel)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(unsigned)acc[i][j][4 * q], (unsigned)acc[i][j][4 * q + 1], (unsigned)acc[i][j][4 * q + 2], (unsigned)acc[i][j][4 * q + 3]},
                                                           rsS, (int)(lane_off + (unsigned)(((i * TN + j) * 4 + q) * 64) * 16u), (int)((unsigned)ks_part * part_bytes), 17);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through stores are acknowledged
        __syncthreads();
        int *const flag = reinterpret_cast<int *>(smem);
        if (t == 0) {
            const unsigned old = __hip_atomic_fetch_add(p.ks_counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = old == (unsigned)p.ks - 1u;
        }
        __syncthreads();
        const bool last = *flag != 0;
        __syncthreads(); // (the row constants are parked over the same LDS bytes next)
        if (!last) return;
        for (int q2 = 0; q2 < p.ks; q2++) {
            if (q2 == ks_part) continue;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int q = 0; q < 4; q++) { // L2-bypassing loads: the other parts were written by workgroups on any XCD
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsS, (int)(lane_off + (unsigned)(((i * TN + j) * 4 + q) * 64) * 16u), (int)((unsigned)q2 * part_bytes), 17);
                        acc[i][j][4 * q] += (int)v[0]; acc[i][j][4 * q + 1] += (int)v[1]; acc[i][j][4 * q + 2] += (int)v[2]; acc[i][j][4 * q + 3] += (int)v[3];
                    }
        }
        if (t == 0) p.ks_counters[tile] = 0u; // every launch leaves its counters zero
    }
    if constexpr (KG > 1) {
        // partial sums of k-groups 1 .. KG-1 -> LDS ([group][register][thread of the group]: 1 KiB per wave store), added by group 0
        int *red = reinterpret_cast<int *>(smem);
        constexpr int RW = TM * TN * 16 + TN; // words per thread: accumulators + column-sum partials
        const int tg = t & 255;
        if (kg > 0) {
            int *dst = red + (kg - 1) * RW * 256 + tg;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) dst[((i * TN + j) * 16 + r) * 256] = acc[i][j][r];
            if constexpr (!RT) {
#pragma unroll
                for (int j = 0; j < TN; j++) dst[(TM * TN * 16 + j) * 256] = cs[j];
            }
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int g = 1; g < KG; g++) {
                const int *src = red + (g - 1) * RW * 256 + tg;
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[i][j][r] += src[((i * TN + j) * 16 + r) * 256];
                if constexpr (!RT) {
#pragma unroll
                    for (int j = 0; j < TN; j++) cs[j] += src[(TM * TN * 16 + j) * 256];
                }
            }
        }
        __syncthreads(); // the row constants are parked over the same bytes next
    }
    int *rowc = reinterpret_cast<int *>(smem); // [4][BM]: row sum, a_zp, bias, per-row scale
    if (t < BM) {
        rowc[t] = rc_rsum;
        if constexpr (!RT) rowc[BM + t] = zp_from_raw(rc_az_raw, p.a_signed);
        rowc[2 * BM + t] = __builtin_bit_cast(int, rc_bias);
        rowc[3 * BM + t] = __builtin_bit_cast(int, rc_srow);
    }
    __syncthreads();
    if (!RT && !p.csum) {
#pragma unroll
        for (int j = 0; j < TN; j++) csv[j] = (unsigned)(cs[j] + __shfl_xor(cs[j], 32, 64)); // the two k halves of the column
    }
    I8_STAMP(6) // drain, k-group reduction, row constants through LDS
    if constexpr (!BQ) {
        [[maybe_unused]] const int raw_u = bz_shift < 0 ? -1 : (int)((bz_word >> bz_shift) & 0xffu);
#pragma unroll
        for (int j = 0; j < TN; j++) bzv[j] = (unsigned)zp_from_raw(RT ? raw_u : bz_raw[j], p.b_signed);
    }
    const bool epi = KG == 1 || kg == 0; // the epilogue belongs to k-group 0
    if (QO != 1 && !epi) return;         // (no barrier follows; the grid-wide exchange of QO == 1 needs every wave of the workgroup)

    // ---- epilogue: zero-point algebra, optional cast_scale / bias / residual / relu, store.  16 accumulator
    // registers (one 32 x 32 block) are finished at a time; all of a block's stores are issued back to back.
    float st_mn = __builtin_inff(), st_mx = -__builtin_inff(); // output statistics for the consuming DynamicQuantizeLinear
    const int ml0 = wm0 + 4 * half;
    const bool full_rows = m0 + BM <= p.M;                                                           // (workgroup-uniform)
    constexpr bool rowterm_only = RT; // weight zero point 0, one activation zero point (the launcher checks: a_zp == NULL, signed A, b_zp_len <= 1)
    if (epi) {
#pragma unroll
    for (int i = 0; i < TM; i++) {
        unsigned rsv[16], azv[16];
        float bv[16], srow[16];
        bool mok[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ml = ml0 + i * 32 + acc_row(r);
            mok[r] = m0 + ml < p.M;
            rsv[r] = (unsigned)rowc[ml];
            azv[r] = RT ? 0u : (unsigned)rowc[BM + ml];
            bv[r] = __builtin_bit_cast(float, rowc[2 * BM + ml]);
            srow[r] = __builtin_bit_cast(float, rowc[3 * BM + ml]);
        }
        // The convolution form of every ort-quantized graph -- weight zero point 0 (signed weights without a zero-point input), ONE activation
        // zero point -- needs a single correction term per ROW, bz * rowsum(A): formed once per 32-row block instead of three multiplies and
        // three adds per element (the epilogue of these short-K launches is VALU-issue bound: tools/debug/i8_trace.py measured 2.7-5.4 us of
        // epilogue next to a 2-7 us k-loop).  Same integers: the dropped terms are multiplied by a zero.
        [[maybe_unused]] unsigned trow[16];
        if constexpr (rowterm_only) {
#pragma unroll
            for (int r = 0; r < 16; r++) trow[r] = bzv[0] * rsv[r];
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const bool cok = basev[j] != OOB;
            unsigned v[16];
            if constexpr (rowterm_only) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = (unsigned)acc[i][j][r] - trow[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    v[r] = (unsigned)acc[i][j][r] - bzv[j] * rsv[r] - azv[r] * csv[j] + (unsigned)p.Kreal * azv[r] * bzv[j];
            }
            if (p.scale) {
                float f[16];
                if (p.scale_per_row) {
#pragma unroll
                    for (int r = 0; r < 16; r++) f[r] = (float)(int)v[r] * srow[r]; // cast_scale (matmul.rs:751,761)
                } else {
                    const float sj = scv[j];
#pragma unroll
                    for (int r = 0; r < 16; r++) f[r] = (float)(int)v[r] * sj;
                }
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 16; r++) f[r] = f[r] + bv[r];
                }
                if constexpr (RES) {
#pragma unroll
                    for (int r = 0; r < 16; r++) f[r] = f[r] + rr[i][j][r];
                }
                if (p.relu) {
#pragma unroll
                    for (int r = 0; r < 16; r++) f[r] = vm::relu(f[r]);
                }
                if (p.stats) { // fminf / fmaxf drop NaNs like the reference's min/max sweep (min_max.rs:27-30)
                    if (full_rows) { // every row of the tile exists: one column test for the block
                        if (cok) {
#pragma unroll
                            for (int r = 0; r < 16; r++) { st_mn = fminf(f[r], st_mn); st_mx = fmaxf(f[r], st_mx); }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            if (mok[r] && cok) { st_mn = fminf(f[r], st_mn); st_mx = fmaxf(f[r], st_mx); }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = __builtin_bit_cast(unsigned, f[r]);
            }
            if constexpr (QO) { // the finished values wait in the accumulator registers for the grid-wide statistics
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = (int)v[r];
            }
            if ((!QO || p.C) && !p.no_store) {
                const unsigned vo_col = cok ? basev[j] + half_off : OOB;
                if (full_rows) {
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        __builtin_amdgcn_raw_buffer_store_b32(v[r], rsC, (int)vo_col, (int)((unsigned)(mb_u + i * 32 + acc_row(r)) * rs4), 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        __builtin_amdgcn_raw_buffer_store_b32(v[r], rsC, (int)(mok[r] ? vo_col : OOB), (int)((unsigned)(mb_u + i * 32 + acc_row(r)) * rs4), 0);
                }
            }
        }
    }
    } // epi
    if constexpr (QO != 0) {
        dql::QParams q;
        const unsigned G = gridDim.x;
        if constexpr (QO == 2) { // the statistics came from the previous launch (folded in the prologue): straight to the codes
            q.scale = q_scale; q.inv_scale = q_inv; q.zp = q_zp;
            if (t == 0 && blockIdx.x == 0) {
                *p.q_scale_out = q.scale;
                *p.q_zp_out = (uint8_t)q.zp;
                if (p.q_mul_by) *p.q_product = q.scale * p.q_mul_by[0];
            }
        } else {
        // ---- (1) all-gather of the workgroups' min / max.  One hop: every workgroup publishes ONE 8-byte granule {min, max} (ordered-uint
        // images; the pair a reset leaves there, {0xffffffff, 0}, cannot be a real one, so the data is its own flag) with a write-through
        // store and then sweeps all G granules with relaxed agent-scope loads until none is the reset pair.  No counters, no fences, no
        // read-modify-write on the critical path (the first version -- eight arrival counters polled by every workgroup plus the slot
        // atomics -- cost 6 / 9 / 22 us at 200 / 392 / 784 workgroups: profiles/r06/int8_qout_per_layer.txt).
        unsigned long long *const gran = reinterpret_cast<unsigned long long *>(p.sync + kSyncCtlWords);
        float *const gred = reinterpret_cast<float *>(smem + 8192); // (the row constants at the start of the stage buffers are dead by now)
        if (epi) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { st_mn = fminf(st_mn, __shfl_xor(st_mn, o, 64)); st_mx = fmaxf(st_mx, __shfl_xor(st_mx, o, 64)); }
            if (lane == 0) {
                gred[wq] = st_mn; gred[4 + wq] = st_mx;
                // the statistics block as rten_hip_conv2d_int8_stats leaves it (for a second reader of the f32 output); nobody waits for these
                const unsigned slot = (blockIdx.x * 4u + (unsigned)wq) % (unsigned)dql::kStatSlots;
                atomicMin(&p.stats[slot], dql::f2ord(st_mn));
                atomicMax(&p.stats[dql::kStatSlots + slot], dql::f2ord(st_mx));
            }
        }
        __syncthreads();
        if (t == 0) {
            const float mn = fminf(fminf(gred[0], gred[1]), fminf(gred[2], gred[3])), mx = fmaxf(fmaxf(gred[4], gred[5]), fmaxf(gred[6], gred[7]));
            __hip_atomic_store(gran + blockIdx.x, ((unsigned long long)dql::f2ord(mx) << 32) | dql::f2ord(mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float ga = __builtin_inff(), gb = -__builtin_inff();
        for (unsigned spins = 0;;) {
            bool ok = true;
            ga = __builtin_inff(); gb = -__builtin_inff();
            for (unsigned g = (unsigned)t; g < G; g += 256u * KG) {
                const unsigned long long v = __hip_atomic_load(gran + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && v != kGranuleReset;
                ga = fminf(ga, dql::ord2f((unsigned)v)); gb = fmaxf(gb, dql::ord2f((unsigned)(v >> 32)));
            }
            if (__syncthreads_and(ok)) break;
            if (++spins >= kSyncSpinLimit) { // not every workgroup is resident (or a previous launch timed out): give up, LOUDLY --
                // the block's time-out flag, the context's sticky fault word (the next rten_hip_sync / graph_launch fails) and a NaN scale for the
                // consumer (below): statistics that miss a workgroup must never turn into plausible codes
                if (t == 0) {
                    __hip_atomic_store(p.sync + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p.fault) __hip_atomic_store(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    *p.q_scale_out = __builtin_nanf("");
                    if (p.q_mul_by) *p.q_product = __builtin_nanf("");
                }
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (t == 0) { // the last workgroup out resets the granules for the next launch (everybody has finished sweeping by then)
            const unsigned old = __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gred[8] = old == G - 1u ? 1.f : 0.f;
        }
        // ---- (2) the statistics of the whole tensor -> DynamicQuantizeLinear's parameters (quantize.rs:397-419)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { ga = fminf(ga, __shfl_xor(ga, o, 64)); gb = fmaxf(gb, __shfl_xor(gb, o, 64)); }
        if (lane == 0) { gred[16 + wave_all] = ga; gred[32 + wave_all] = gb; }
        __syncthreads();
        if (gred[8] != 0.f) {
            for (unsigned g = (unsigned)t; g < G; g += 256u * KG) __hip_atomic_store(gran + g, kGranuleReset, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == 0) __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!epi) return;
        float g_mn = gred[16], g_mx = gred[32];
#pragma unroll
        for (int wv = 1; wv < 4 * KG; wv++) { g_mn = fminf(g_mn, gred[16 + wv]); g_mx = fmaxf(g_mx, gred[32 + wv]); }
        q = dql::dql_params(g_mn, g_mx);
        if (t == 0 && blockIdx.x == 0) {
            // (a workgroup that timed out -- before or after this store -- leaves NaN here: whoever gives up first has set the flag)
            const bool void_run = __hip_atomic_load(p.sync + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            *p.q_scale_out = void_run ? __builtin_nanf("") : q.scale;
            *p.q_zp_out = (uint8_t)q.zp;
            if (p.q_mul_by) *p.q_product = void_run ? __builtin_nanf("") : q.scale * p.q_mul_by[0]; // the Mul(x_scale, w_scale) node of the consumer
        }
        } // QO == 1
        // ---- (3) codes -> the consumer's staged image [N][C/16][Hp][Wp][16 B] (signed domain).  A lane holds, per 32-row block, four
        // dwords of four consecutive channels each: rows 0-3, 8-11, 16-19, 24-27 (+4 in the upper half of the wave); two
        // v_permlane32_swap give the lower half the sixteen channels 0-15 of its pixel and the upper half channels 16-31.
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void *)p.q_out, 0, (int)p.q_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn0 + j * 32 + l31;
            const bool cok = n < p.N;
            const int nn = cok ? n : 0;
            const int nb = rten_div(nn, p.d_pn), np = nn - nb * p.Pn;
            const int oy = rten_div(np, p.d_qw), ox = np - oy * p.q_W;
            const unsigned pix = (unsigned)((oy + p.q_pt) * p.q_Wp + ox + p.q_pl);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    d[g] = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int bits = acc[i][j][4 * g + b]; // (copied to a scalar first: a bit_cast applied directly to an ext-vector element reads element 0 under this compiler)
                        d[g] |= ((dql::quant_u8(__int_as_float(bits), q.inv_scale, q.zp) ^ 0x80u) & 0xffu) << (8 * b);
                    }
                }
                const auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false); // {lower: own d0 | upper: partner's d2}, {partner's d0 | own d2}
                const auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                const unsigned w0 = s02[0], w1 = s02[1], w2 = s13[0], w3 = s13[1];
                const int chunk = (m0 + wm0 + i * 32) / 16 + half;
                const unsigned off = (chunk * 16 < p.M && cok) ? ((unsigned)((nb * p.q_cb + chunk) * (p.q_Hp * p.q_Wp)) + pix) * 16u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w0, w1, w2, w3}, rsQ, (int)off, 0, 0);
            }
        }
        // border pieces of the consumer's padding (value of its pad mode), dealt over the whole grid
        const int nbp = p.q_Hp * p.q_Wp - p.q_H * p.q_W;
        if (nbp > 0) {
            int pad_s = 0;
            if (p.q_pad_mode == RTEN_HIP_PAD_ZERO_POINT) pad_s = q.zp - 128;
            else if (p.q_pad_mode == RTEN_HIP_PAD_RAW0_U8) pad_s = -128;
            const unsigned fb = ((unsigned)pad_s & 0xffu) * 0x01010101u;
            const int planes = (p.N / p.Pn) * p.q_cb, total = planes * nbp;
            const int top = p.q_pt * p.q_Wp, side = p.q_Wp - p.q_W;
            for (int idx = (int)blockIdx.x * 256 + t; idx < total; idx += (int)G * 256) {
                const int plane = idx / nbp, b = idx - plane * nbp;
                int pos;
                if (b < top) pos = b;
                else {
                    const int b1 = b - top;
                    if (side > 0 && b1 < p.q_H * side) {
                        const int row = b1 / side, wi = b1 - row * side;
                        pos = (p.q_pt + row) * p.q_Wp + (wi < p.q_pl ? wi : p.q_W + wi);
                    } else pos = (p.q_pt + p.q_H) * p.q_Wp + (b1 - p.q_H * side);
                }
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{fb, fb, fb, fb}, rsQ, (int)(((unsigned)(plane * (p.q_Hp * p.q_Wp)) + (unsigned)pos) * 16u), 0, 0);
            }
        }
#ifdef RTEN_TRACE
        {
            I8_STAMP(7) // epilogue (zero-point algebra, scale, bias, residual, statistics, stores issued; QO: + the grid-wide exchange and the quantized stores)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t_end_ = __builtin_readcyclecounter();
            if (blockIdx.x == (gridDim.x > 8 ? 8 : 0) && t == 0)
                printf("[i8 trace] <%d,%d,res=%d,kg=%d,bq=%d,qo=%d,rt=%d> M %d N %d Kp %d grid %u: args+decode %llu, table %llu, operands %llu, residual req %llu, first tiles %llu, k-loop %llu (%d trips), drain+rowc %llu, epilogue %llu, store drain %llu cycles\n",
                       BM, BN, (int)RES, KG, (int)BQ, (int)QO, (int)RT, p.M, p.N, p.Kp, gridDim.x, tr_seg[0], tr_seg[1], tr_seg[2], tr_seg[3], tr_seg[4], tr_seg[5], nit, tr_seg[6], tr_seg[7], t_end_ - tr_prev);
        }
#endif
        return;
    }
    if (p.stats) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { st_mn = fminf(st_mn, __shfl_xor(st_mn, o, 64)); st_mx = fmaxf(st_mx, __shfl_xor(st_mx, o, 64)); }
        if (lane == 0) {
            const unsigned slot = (blockIdx.x * 4u + (unsigned)wq) % (unsigned)dql::kStatSlots;
            atomicMin(&p.stats[slot], dql::f2ord(st_mn));
            atomicMax(&p.stats[dql::kStatSlots + slot], dql::f2ord(st_mx));
        }
    }
#ifdef RTEN_TRACE
    {
        I8_STAMP(7) // epilogue (zero-point algebra, scale, bias, residual, statistics, stores issued; QO: + the grid-wide exchange and the quantized stores)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end_ = __builtin_readcyclecounter();
        if ((blockIdx.x == (gridDim.x > 8 ? 8 : 0) || (gridDim.x > 64 && blockIdx.x == gridDim.x - 9)) && t == 0) // (an early workgroup and a late one)
            printf("[i8 trace] <%d,%d,res=%d,kg=%d,bq=%d,qo=%d,rt=%d> M %d N %d Kp %d grid %u: args+decode %llu, table %llu, operands %llu, residual req %llu, first tiles %llu, k-loop %llu (%d trips), drain+rowc %llu, epilogue %llu, store drain %llu cycles; %llu cycles in %llu ticks of 10 ns\n",
                   BM, BN, (int)RES, KG, (int)BQ, (int)QO, (int)RT, p.M, p.N, p.Kp, gridDim.x, tr_seg[0], tr_seg[1], tr_seg[2], tr_seg[3], tr_seg[4], tr_seg[5], nit, tr_seg[6], tr_seg[7], t_end_ - tr_prev,
                   t_end_ - tr_c0, (unsigned long long)__builtin_amdgcn_s_memrealtime() - tr_r0);
    }
#endif
#undef I8_STAMP
}

// Transposing variant of the row packer for operands whose rows are NOT k-contiguous (e.g. MatMulInteger weights
// [K][N]: row n walks k with stride N).  64 x 64 byte tiles through LDS: coalesced reads along the source's
// contiguous axis, 4-byte writes along k.  Row sums via atomics into a zeroed array.
__global__ __launch_bounds__(256) void i8_pack_rows_t_kernel(const uint8_t *__restrict__ src, long long row_stride, long long k_stride, int rows,
                                                            int K, int Kp, unsigned flip, uint8_t *__restrict__ dst, int *__restrict__ sums) {
    __shared__ uint8_t tile[64][64 + 4];
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    // read: consecutive threads walk rows (the source's contiguous axis when row_stride == 1)
#pragma unroll
    for (int pass = 0; pass < 16; pass++) {
        const int rl = t & 63, kl = pass * 4 + (t >> 6);
        const int r = r0 + rl, k = k0 + kl;
        unsigned v = 0;
        if (r < rows && k < K) v = ((unsigned)src[(long long)r * row_stride + (long long)k * k_stride] ^ flip) & 0xffu;
        tile[rl][kl] = (uint8_t)v;
    }
    __syncthreads();
    // write: thread -> (row, 16-byte group), consecutive lanes on consecutive rows of one chunk plane; k0 + 64 <= Kp
    // always (Kp is a multiple of 64)
    const int rl = t & 63, g = t >> 6;
    const int r = r0 + rl;
    int sum = 0;
    if (r < rows) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            w[q] = (unsigned)tile[rl][g * 16 + q * 4] | ((unsigned)tile[rl][g * 16 + q * 4 + 1] << 8) | ((unsigned)tile[rl][g * 16 + q * 4 + 2] << 16) |
                   ((unsigned)tile[rl][g * 16 + q * 4 + 3] << 24);
            sum = __builtin_amdgcn_sdot4((int)w[q], 0x01010101, sum, false);
        }
        *reinterpret_cast<uint4 *>(dst + ((long long)(k0 / 16 + g) * rows + r) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (r < rows) atomicAdd(&sums[r], sum);
}

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// Workgroups of `kern` (threads, dynamic LDS) the whole device holds at once (the occupancy query is cached per kernel and LDS size).
int resident_capacity(rten_hip_ctx *ctx, const void *kern, int threads, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<const void *, size_t>, int> cache;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(kern, lds);
    auto it = cache.find(key);
    if (it == cache.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess) nb = 0;
        it = cache.emplace(key, nb < 8 ? nb : 8).first;
    }
    return it->second * ctx->num_cus;
}

// Returns false (nothing launched) when QO is requested and the grid cannot be resident all at once.
template <int BM, int BN, int NST, int KTK = 64, int KG = 1, bool BQ = false, int QO = 0, bool KS = false>
bool launch_fast(rten_hip_ctx *ctx, FastArgs &a, const char *name, double ops, double bytes) {
    static_assert(KTK == 64 || KG == 1, "k-groups walk 64-byte k-tiles");
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    constexpr size_t ring = (size_t)NST * KG * (BM + BN) * KTK;
    static_assert(ring <= 128 * 1024, "int8 tile ring exceeds the LDS of a compute unit");
    const int nkt = (a.Kp + KTK - 1) / KTK;
    const size_t lds = ring + (size_t)((nkt + KG - 1) / KG + NST) * KG * (KTK / 16) * 4 + 64; // + the chunk -> B offset table + the statistics fold's scratch
    a.d_pn = rten_make_div((long long)a.tiles_n * BN, a.Pn);
    a.d_tile = rten_make_div((long long)a.tiles_m * a.tiles_n, a.n_fastest ? a.tiles_n : a.tiles_m);
    if (BQ) a.d_hw = rten_make_div((long long)a.tiles_n * BN, a.HW);
    if (QO) a.d_qw = rten_make_div(a.Pn, a.q_W);
    if (a.conv) {
        const long long tchunks = (long long)((nkt + KG - 1) / KG + NST) * KG * (KTK / 16);
        a.d_ow = rten_make_div(a.Pn, a.OW);
        a.d_cpc = rten_make_div(tchunks, a.Cp / 16);
        a.d_kw = rten_make_div(tchunks, a.KW);
    }
    bool launched = true;
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (QO == 1 && ((long long)a.tiles_m * a.tiles_n > resident_capacity(ctx, (const void *)kern, 256 * KG, lds) || (long long)a.tiles_m * a.tiles_n > (long long)kSyncGranules)) {
            launched = false;
            return;
        }
        ProfScope ps(ctx, name, ops, bytes);
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.tiles_m * a.tiles_n * (KS ? a.ks : 1))), dim3(256 * KG), lds, ctx->stream, a);
    };
    // RT: the single-row-term zero-point algebra of the convolution form (weight zero point 0 in the signed domain, one activation zero point)
    const bool rt = a.conv && !(a.debug_flags & 1) && a.a_zp == nullptr && a.a_signed && a.b_zp_len <= 1 && !a.need_csum;
    if constexpr (QO == 2) { // the recompute form is built for the single-row-term convolution without a residual (the single-consumer edges of a bottleneck block)
        if (!rt || (a.res && a.scale)) { launched = false; return launched; }
        go(igemm_i8_fast_kernel<BM, BN, NST, false, KTK, KG, BQ, 2, true>);
        return launched;
    }
    if constexpr (KS) { // (the dispatcher checked: RT form, no residual)
        a.d_tile = rten_make_div((long long)a.tiles_m * a.tiles_n, a.n_fastest ? a.tiles_n : a.tiles_m);
        go(igemm_i8_fast_kernel<BM, BN, NST, false, KTK, KG, BQ, QO, true, true>);
        return launched;
    }
    if (rt) {
        if (a.res && a.scale) go(igemm_i8_fast_kernel<BM, BN, NST, true, KTK, KG, BQ, QO, true>);
        else go(igemm_i8_fast_kernel<BM, BN, NST, false, KTK, KG, BQ, QO, true>);
    } else {
        if (a.res && a.scale) go(igemm_i8_fast_kernel<BM, BN, NST, true, KTK, KG, BQ, QO, false>);
        else go(igemm_i8_fast_kernel<BM, BN, NST, false, KTK, KG, BQ, QO, false>);
    }
    return launched;
}

int32_t dispatch_fast(rten_hip_ctx *ctx, FastArgs &a, double ops, double bytes) {
    // tile choice: the largest tile that still gives every CU a workgroup (the tile DMA of a workgroup tops out far
    // below what the MFMAs could consume, so an idle CU costs more than the extra operand re-reads of a smaller tile)
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const long long t12864 = (long long)((a.M + 127) / 128) * ((a.N + 63) / 64);
    static const int env_tile = getenv("RTEN_I8_TILE") ? atoi(getenv("RTEN_I8_TILE")) : -1; // (tuning: 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64)
    const int tile = env_tile >= 0 ? env_tile : (ctx->int8_tile >= 0 ? ctx->int8_tile : (a.M <= 64 ? 2 : (t128 >= ctx->num_cus ? 0 : (t12864 >= ctx->num_cus ? 1 : 3))));
    // tile order: consecutive workgroup ids (one XCD's share) walk the axis of the SMALLER operand, so that the larger
    // one is fetched into as few of the eight L2s as possible
    a.n_fastest = (double)a.a_bytes > (double)a.b_bytes ? 1 : 0;
    // (256-byte k-tiles were measured and are slower: profiles/r05/int8_notes.md)
    if (a.q_out && a.in_stats) { // recompute form of a quantized output (QO == 2): the plain kernels' tile choice, nothing to fit
        const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
        const int nkt = (a.Kp + 63) / 64;
        const bool kg4 = !(ctx->debug & 0x800) && t64 * 4 <= 5 * ctx->num_cus && nkt >= 16;
        bool ok;
        if (tile == 0) ok = launch_fast<128, 128, 3, 64, 1, false, 2>(ctx, a, "igemm_i8_fast_kernel<128,128,qo2>", ops, bytes);
        else if (tile == 1) ok = launch_fast<128, 64, 3, 64, 1, false, 2>(ctx, a, "igemm_i8_fast_kernel<128,64,qo2>", ops, bytes);
        else if (tile == 2) ok = launch_fast<64, 128, 3, 64, 1, false, 2>(ctx, a, "igemm_i8_fast_kernel<64,128,qo2>", ops, bytes);
        else if (kg4) ok = launch_fast<64, 64, 3, 64, 4, false, 2>(ctx, a, "igemm_i8_fast_kernel<64,64,kg4,qo2>", ops, bytes);
        else ok = launch_fast<64, 64, 3, 64, 1, false, 2>(ctx, a, "igemm_i8_fast_kernel<64,64,qo2>", ops, bytes);
        if (!ok) return RTEN_HIP_ERR_UNSUPPORTED;
        RTEN_LAUNCH_CHECK(ctx, "igemm_i8_fast_kernel launch");
        return RTEN_HIP_OK;
    }
    if (a.q_out) { // quantized-output form (rten_hip_conv2d_int8_qout): the same tile choice, or the next larger one that fits the chip at once
        const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
        const int nkt = (a.Kp + 63) / 64;
        const bool kg4 = !(ctx->debug & 0x800) && t64 * 4 <= 5 * ctx->num_cus && nkt >= 16;
        bool ok = false;
        if (tile == 0) ok = launch_fast<128, 128, 3, 64, 1, false, true>(ctx, a, "igemm_i8_fast_kernel<128,128,qo>", ops, bytes);
        else if (tile == 1) ok = launch_fast<128, 64, 3, 64, 1, false, true>(ctx, a, "igemm_i8_fast_kernel<128,64,qo>", ops, bytes) ||
                                 launch_fast<128, 128, 3, 64, 1, false, true>(ctx, a, "igemm_i8_fast_kernel<128,128,qo>", ops, bytes);
        else if (tile == 2) ok = launch_fast<64, 128, 3, 64, 1, false, true>(ctx, a, "igemm_i8_fast_kernel<64,128,qo>", ops, bytes) ||
                                 launch_fast<64, 256, 3, 64, 1, false, true>(ctx, a, "igemm_i8_fast_kernel<64,256,qo>", ops, bytes);
        else if (kg4) ok = launch_fast<64, 64, 3, 64, 4, false, true>(ctx, a, "igemm_i8_fast_kernel<64,64,kg4,qo>", ops, bytes);
        else ok = launch_fast<64, 64, 3, 64, 1, false, true>(ctx, a, "igemm_i8_fast_kernel<64,64,qo>", ops, bytes) ||
                  launch_fast<128, 64, 3, 64, 1, false, true>(ctx, a, "igemm_i8_fast_kernel<128,64,qo>", ops, bytes);
        if (!ok) return RTEN_HIP_ERR_UNSUPPORTED; // more workgroups than the device holds at once: run the two-launch sequence
        RTEN_LAUNCH_CHECK(ctx, "igemm_i8_fast_kernel launch");
        return RTEN_HIP_OK;
    }
    if (a.xf) { // quantize-on-load form (rten_hip_conv2d_int8_dql)
        if (tile == 0) launch_fast<128, 128, 3, 64, 1, true>(ctx, a, "igemm_i8_fast_kernel<128,128,bq>", ops, bytes);
        else if (tile == 1) launch_fast<128, 64, 3, 64, 1, true>(ctx, a, "igemm_i8_fast_kernel<128,64,bq>", ops, bytes);
        else if (tile == 2) launch_fast<64, 128, 3, 64, 1, true>(ctx, a, "igemm_i8_fast_kernel<64,128,bq>", ops, bytes);
        else launch_fast<64, 64, 3, 64, 1, true>(ctx, a, "igemm_i8_fast_kernel<64,64,bq>", ops, bytes);
        RTEN_LAUNCH_CHECK(ctx, "igemm_i8_fast_kernel launch");
        return RTEN_HIP_OK;
    }
    // Cross-workgroup K split (KS kernels): under-filled long-K launches of the single-row-term convolution form without a residual.  `a.ks` = the largest
    // split the caller's slab holds (0: none); RTEN_I8_KS = "<parts>[,<tile>]" selects it (a measurement knob: tile 0 = 128x128, 1 = 128x64, 3 = 64x64).
    {
        const bool rt = a.conv && !(a.debug_flags & 1) && a.a_zp == nullptr && a.a_signed && a.b_zp_len <= 1 && !a.need_csum;
        static const char *env_ks = getenv("RTEN_I8_KS");
        static const int env_parts = env_ks ? atoi(env_ks) : -1, env_kst = (env_ks && strchr(env_ks, ',')) ? atoi(strchr(env_ks, ',') + 1) : -1;
        const int nkt = (a.Kp + 63) / 64;
        const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
        int parts = 0, kst = 3;
        if (rt && !a.res && a.ks >= 2 && a.slab && a.ks_counters && !(ctx->debug & 0x1000)) {
            // MEASURED AND NOT USED BY DEFAULT (profiles/r09/int8_cross_workgroup_k_split.txt): every split loses -- s2 3x3 13.7 -> 17.1 us (2 parts, 64x64),
            // 16.9 (2 parts, 128x64), 19.0 (128x128), 20.8 (4 parts); s3 3x3 12.6 -> 15.7; K = 1024 1x1 9.1 -> 13.6; the int8 step 1.48 -> 1.53 ms (one
            // replica), 0.89 -> 0.97 (four).  The hand-off (16-64 KB of write-through partials per workgroup, the counter, the L2-bypassing reload) costs 4-7 us
            // against a k-loop of ~6 us that it can shorten by half at best; the in-workgroup form (k-groups through LDS) stays the rule for stage 3.
            if (env_parts >= 0) { parts = env_parts; kst = env_kst >= 0 ? env_kst : 3; }
            if (parts > a.ks) parts = a.ks;
            if (parts > nkt) parts = nkt;
        }
        if (parts >= 2) {
            a.ks = parts;
            const long long tiles = kst == 0 ? t128 : kst == 1 ? t12864 : t64;
            if (tiles <= rten_hip_ctx::kSplitCounters) {
                if (kst == 0) launch_fast<128, 128, 3, 64, 1, false, false, true>(ctx, a, "igemm_i8_fast_kernel<128,128,ks>", ops, bytes);
                else if (kst == 1) launch_fast<128, 64, 3, 64, 1, false, false, true>(ctx, a, "igemm_i8_fast_kernel<128,64,ks>", ops, bytes);
                else launch_fast<64, 64, 3, 64, 1, false, false, true>(ctx, a, "igemm_i8_fast_kernel<64,64,ks>", ops, bytes);
                RTEN_LAUNCH_CHECK(ctx, "igemm_i8_fast_kernel launch");
                return RTEN_HIP_OK;
            }
        }
        a.ks = 0;
    }
    // (measured again in round 3 and dropped -- profiles/r06/int8_tile_experiments.txt: 128-byte k-tiles on the 128x64 / 64x128 / 64x64 tiles
    // +5 % on the whole conv time, the next larger tile for the under-filled launches +6 %, four LDS stages +2 %; no layer gains more than 5 %)
    if (tile == 0) {
        launch_fast<128, 128, 3>(ctx, a, "igemm_i8_fast_kernel<128,128>", ops, bytes);
    } else if (tile == 1) {
        launch_fast<128, 64, 3>(ctx, a, "igemm_i8_fast_kernel<128,64>", ops, bytes);
    } else if (tile == 2) {
        launch_fast<64, 128, 3>(ctx, a, "igemm_i8_fast_kernel<64,128>", ops, bytes);
    } else {
        // under-filled launches (fewer than ~2 workgroups per CU) split K over 2 or 4 k-groups inside the workgroup: more waves
        // per SIMD to hide the k-loop's latencies behind each other (RTEN_HIP_DEBUG bit 0x800 turns this off)
        const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
        const int nkt = (a.Kp + 63) / 64;
        // measured (profiles/r05/int8_per_layer.txt): 4 groups pay off below ~1.25 workgroups per CU with K >= 1024 bytes (stage-4 3x3:
        // 20.8 -> 14.8 us); 2 groups on the 1.5-workgroup launches of stage 3 do not (10.3 -> 11.5 us) and are not used
        const int kgs = (ctx->debug & 0x800) ? 1 : (t64 * 4 <= 5 * ctx->num_cus && nkt >= 16 ? 4 : 1);
        if (kgs == 4) launch_fast<64, 64, 3, 64, 4>(ctx, a, "igemm_i8_fast_kernel<64,64,kg4>", ops, bytes);
        else launch_fast<64, 64, 3>(ctx, a, "igemm_i8_fast_kernel<64,64>", ops, bytes);
    }
    RTEN_LAUNCH_CHECK(ctx, "igemm_i8_fast_kernel launch");
    return RTEN_HIP_OK;
}

} // namespace

namespace {
inline int gemm_kp(int k) { return (k + KT - 1) / KT * KT; }
inline bool gemm_b_covered(int k, int n) { return k > 0 && n > 0 && k <= (1 << 18) && (long long)n * gemm_kp(k) < (1ll << 31); } // K bound: the kernel's chunk table lives in LDS

// rows of a strided u8 / i8 matrix -> chunk-major signed operand + row sums (both packers; see the kernels)
void pack_rows(rten_hip_ctx *ctx, const void *src, long long row_stride, long long k_stride, int rows, int k, int Kp, unsigned flip, char *dst, char *sums) {
    if (k_stride == 1) {
        hipLaunchKernelGGL(i8_pack_rows_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->stream, (const uint8_t *)src, row_stride, k_stride, k, k, 1, Kp, Kp, rows, flip,
                           (uint8_t *)dst, (int *)sums);
    } else {
        hipMemsetAsync(sums, 0, (size_t)rows * 4, ctx->stream);
        hipLaunchKernelGGL(i8_pack_rows_t_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)(Kp / 64)), dim3(256), 0, ctx->stream, (const uint8_t *)src, row_stride, k_stride,
                           rows, k, Kp, flip, (uint8_t *)dst, (int *)sums);
    }
}
} // namespace

// Load-time staging of a constant MatMulInteger RHS (PackedBMatrix, rten-gemm/src/prepack.rs:19-120; packing/int8.rs:80-249
// keeps the column sums inside the packed image, as here).  Layout: [Kp/16][n][16] signed bytes, then int32[n] column sums.
RTEN_EXPORT size_t rten_hip_gemm_int8_packed_bytes(int32_t k, int32_t n) {
    if (!gemm_b_covered(k, n)) return 0;
    return up256((size_t)n * gemm_kp(k)) + up256((size_t)n * 4);
}

RTEN_EXPORT int32_t rten_hip_gemm_int8_prepack(rten_hip_ctx *ctx, int32_t k, int32_t n, const void *b, int64_t b_rs, int64_t b_cs, int32_t b_signed, void *packed) {
    RTEN_CHECK_CTX(ctx);
    if (!b || !packed) return RTEN_HIP_ERR_INVALID_VALUE;
    if (!gemm_b_covered(k, n)) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "gemm_int8 prepack: shape not covered by the staged kernel (packed_bytes == 0)");
    if (b_rs < 0 || b_cs < 0) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "gemm_int8 prepack: negative strides are not supported");
    const int Kp = gemm_kp(k);
    pack_rows(ctx, b, b_cs, b_rs, n, k, Kp, b_signed ? 0u : 0x80u, (char *)packed, (char *)packed + up256((size_t)n * Kp));
    RTEN_LAUNCH_CHECK(ctx, "i8_pack_rows_kernel launch");
    return RTEN_HIP_OK;
}

// Entry points used by int8.hip: return RTEN_HIP_ERR_UNSUPPORTED when the fast path does not cover the call
// (the caller then falls back to the generic kernel).
int32_t rten_i8_fast_gemm(rten_hip_ctx *ctx, const rten_hip_gemm_int8_desc *d, const void *a, const void *b, const void *a_zp,
                          const void *b_zp, const float *scale, void *c) {
    const int Kp = gemm_kp(d->k);
    if (d->k <= 0 || (long long)d->m * Kp >= (1ll << 31) || !gemm_b_covered(d->k, d->n) || (long long)d->m * d->ldc >= (1ll << 29))
        return RTEN_HIP_ERR_UNSUPPORTED;
    // per call: the activation side (A) is staged; B too unless the caller prepacked it at load (MatMulInteger weights)
    const size_t b_img = up256((size_t)d->n * Kp);
    const size_t offA = 4096, offRs = offA + up256((size_t)d->m * Kp), offB = offRs + up256((size_t)d->m * 4),
                 offCs = offB + (d->b_prepacked ? 0 : b_img), total = offCs + (d->b_prepacked ? 0 : up256((size_t)d->n * 4));
    char *sc = (char *)rten_scratch(ctx, total);
    if (!sc) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "int8 staging allocation failed (or attempted during graph capture)");
    pack_rows(ctx, a, d->a_rs, d->a_cs, d->m, d->k, Kp, d->a_signed ? 0u : 0x80u, sc + offA, sc + offRs);
    if (!d->b_prepacked) pack_rows(ctx, b, d->b_cs, d->b_rs, d->n, d->k, Kp, d->b_signed ? 0u : 0x80u, sc + offB, sc + offCs);
    RTEN_LAUNCH_CHECK(ctx, "i8_pack_rows_kernel launch");
    FastArgs g = {};
    g.A = (const uint8_t *)(sc + offA);
    g.B = d->b_prepacked ? (const uint8_t *)b : (const uint8_t *)(sc + offB);
    g.rsum = (const int *)(sc + offRs);
    g.csum = d->b_prepacked ? (const int *)((const char *)b + b_img) : (const int *)(sc + offCs);
    g.C = c;
    g.a_zp = d->a_zp_len ? (const uint8_t *)a_zp : nullptr;
    g.b_zp = d->b_zp_len ? (const uint8_t *)b_zp : nullptr;
    g.scale = d->scale_len ? scale : nullptr;
    g.M = d->m; g.N = d->n; g.Kp = Kp; g.Kreal = d->k;
    g.a_bytes = (unsigned)((size_t)d->m * Kp); g.b_bytes = (unsigned)((size_t)d->n * Kp);
    g.c_rs = d->ldc; g.c_ns = 0; g.Pn = d->n;
    g.a_signed = d->a_signed; g.b_signed = d->b_signed;
    g.a_zp_len = d->a_zp_len; g.b_zp_len = d->b_zp_len; g.scale_len = d->scale_len;
    return dispatch_fast(ctx, g, 2.0 * d->m * (double)d->n * d->k, (double)d->m * d->k + (double)d->k * d->n + 4.0 * d->m * d->n);
}

namespace {
// `packed` (round 6): the few-channel form of stage_packed_chunks -- C <= 4 input channels and a kernel at least two columns wide.  The choice depends on C, KW
// and the group count ONLY: weights are prepacked from a descriptor that knows nothing else (ConvInteger::prepack), and every zero-point form stays exact -- the
// unused bytes of a chunk are 0 in the signed domain on BOTH operands, like the padded channels of the 16-channel-block form, so they add nothing to the product,
// the row sums or the column sums.  Cp / Wp / taps / Kp / img then describe the VIRTUAL convolution the kernel walks (16 channels, KW = nch, stride nch over
// OW * nch virtual columns); sx / kw / dx say so to the launch.
struct ConvGeom { int Cp, Hp, Wp, taps, Kreal, Kp, P; size_t img; bool ok; bool packed; int nch, cpc, sx, kw, dx; };
ConvGeom conv_geom(const rten_hip_conv2d_int8_desc *di) {
    const rten_hip_conv2d_desc *d = &di->conv;
    ConvGeom g = {};
    g.Cp = (d->c + 15) / 16 * 16;
    g.Hp = d->h + d->pads[0] + d->pads[2];
    g.Wp = d->w + d->pads[1] + d->pads[3];
    g.taps = d->kh * d->kw;
    g.Kreal = d->c * g.taps;
    g.P = d->out_h * d->out_w;
    g.sx = d->stride_w; g.kw = d->kw; g.dx = d->dil_w;
    static const bool no_pack = getenv("RTEN_I8_NO_PACK") != nullptr; // (A/B switch: the 16-channel-block form for every geometry)
    if (!no_pack && d->groups == 1 && d->c >= 1 && d->c <= 4 && d->kw >= 2 && d->out_w > 0) {
        g.cpc = 16 / d->c;
        g.nch = (d->kw + g.cpc - 1) / g.cpc;
        if (g.nch < d->kw && g.nch <= 2) { // fewer chunks per kernel row than taps: the smaller product (one or two chunks per output column are instantiated)
            g.packed = true;
            g.Cp = 16;
            g.Wp = d->out_w * g.nch;
            g.taps = d->kh * g.nch;
            g.sx = g.nch; g.kw = g.nch; g.dx = 1;
        }
    }
    g.Kp = (g.taps * g.Cp + KT - 1) / KT * KT;
    g.img = (size_t)d->n * g.Hp * g.Wp * g.Cp;
    g.ok = d->groups == 1 && d->c > 0 && d->o > 0 && g.Kp <= (1 << 18) && // (the kernel's chunk -> offset table lives in LDS)
           // the chunk walk addresses taps on the padded image: the window must fit (true for valid conv geometry)
           (d->out_h - 1) * d->stride_h + (d->kh - 1) * d->dil_h < g.Hp && (d->out_w - 1) * g.sx + (g.kw - 1) * g.dx < g.Wp &&
           g.img < (1ull << 31) && (size_t)d->o * g.Kp < (1ull << 31) && (long long)d->n * d->o * g.P < (1ll << 29);
    return g;
}
} // namespace

namespace {
ScaleProducts one_product(const float *mul_by, float *product) {
    ScaleProducts sp = {};
    if (mul_by) { sp.count = 1; sp.mul_by[0] = mul_by; sp.product[0] = product; }
    return sp;
}

void launch_quantize_stage(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d, const ConvGeom &g, const float *x, const float *ws, int nparts, void *staged,
                           int pad_mode, float *scale, uint8_t *zero_point, const ScaleProducts &sp) {
    if (g.packed) {
        const dim3 pgrid((unsigned)((d->out_w + 127) / 128), (unsigned)((g.Hp + 1) / 2), (unsigned)d->n);
        RTEN_PACKED_DISPATCH(i8_quantize_packed_kernel, d->c, g.nch, pgrid, x, ws, nparts, (uint8_t *)staged, d->h, d->w, g.Hp, d->out_w, d->kw, d->stride_w, d->dil_w, d->pads[0],
                             d->pads[1], pad_mode, scale, zero_point, sp);
        return;
    }
    const dim3 grid((unsigned)((d->h * d->w + 255) / 256), (unsigned)d->n, (unsigned)(g.Cp / 16));
    const dim3 grid_small((unsigned)((g.Hp * g.Wp + 63) / 64), (unsigned)d->n, (unsigned)((g.Cp + 63) / 64));
    const bool vec = (d->h * d->w) % 4 == 0 && ((uintptr_t)x & 15) == 0;
#define QS_ARGS x, ws, nparts, (uint8_t *)staged, d->c, d->h, d->w, g.Hp, g.Wp, g.Cp, d->pads[0], d->pads[1], pad_mode, scale, zero_point, sp
    if (d->h * d->w < 128) hipLaunchKernelGGL(i8_quantize_stage_kernel<0>, grid_small, dim3(256), 0, ctx->stream, QS_ARGS);
    else if (vec) hipLaunchKernelGGL(i8_quantize_stage_kernel<2>, grid, dim3(256), 0, ctx->stream, QS_ARGS);
    else hipLaunchKernelGGL(i8_quantize_stage_kernel<1>, grid, dim3(256), 0, ctx->stream, QS_ARGS);
#undef QS_ARGS
}
} // namespace

RTEN_EXPORT size_t rten_hip_conv2d_int8_packed_bytes(const rten_hip_conv2d_int8_desc *di) {
    if (!di) return 0;
    const ConvGeom g = conv_geom(di);
    return g.ok ? up256((size_t)di->conv.o * g.Kp) + (size_t)di->conv.o * 4 : 0;
}

RTEN_EXPORT int32_t rten_hip_conv2d_int8_prepack(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *w, void *packed) {
    RTEN_CHECK_CTX(ctx);
    if (!di || !w || !packed) return RTEN_HIP_ERR_INVALID_VALUE;
    const ConvGeom g = conv_geom(di);
    if (!g.ok) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv_int8 prepack: geometry not covered by the staged kernel (packed_bytes == 0)");
    const rten_hip_conv2d_desc *d = &di->conv;
    hipLaunchKernelGGL(i8_pack_rows_kernel, dim3((unsigned)d->o), dim3(256), 0, ctx->stream, (const uint8_t *)w, (long long)g.Kreal, 1ll, g.Kreal, d->c, d->kh * d->kw,
                       g.Cp, g.Kp, d->o, di->w_signed ? 0u : 0x80u, (uint8_t *)packed, (int *)((char *)packed + up256((size_t)d->o * g.Kp)), g.packed ? g.nch : 0, g.cpc,
                       d->kw);
    RTEN_LAUNCH_CHECK(ctx, "i8_pack_rows_kernel launch");
    return RTEN_HIP_OK;
}

RTEN_EXPORT size_t rten_hip_conv2d_int8_staged_bytes(const rten_hip_conv2d_int8_desc *di) {
    if (!di || di->x_signed) return 0;
    const ConvGeom g = conv_geom(di);
    return g.ok ? up256(g.img) : 0;
}

RTEN_EXPORT int32_t rten_hip_dynamic_quantize_linear_staged(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const float *x, void *staged,
                                                            float *scale, uint8_t *zero_point, const float *mul_by, float *product) {
    RTEN_CHECK_CTX(ctx);
    if (!di || !x || !staged || !scale || !zero_point || (mul_by && !product)) return RTEN_HIP_ERR_INVALID_VALUE;
    const ConvGeom g = conv_geom(di);
    if (!g.ok || di->x_signed) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "quantize_staged: geometry not covered by the staged kernel (staged_bytes == 0)");
    const rten_hip_conv2d_desc *d = &di->conv;
    const int64_t n = (int64_t)d->n * d->c * d->h * d->w;
    ProfScope ps(ctx, "dynamic_quantize_linear_staged", 0.0, 8.0 * n + (double)g.img);
    int nparts = 0;
    const float *ws = rten_dql_minmax(ctx, n, x, &nparts);
    if (!ws) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "dql: scratch allocation failed");
    launch_quantize_stage(ctx, d, g, x, ws, nparts, staged, rten_effective_pad_mode(di), scale, zero_point, one_product(mul_by, product));
    RTEN_LAUNCH_CHECK(ctx, "i8_quantize_stage_kernel launch");
    return RTEN_HIP_OK;
}

RTEN_EXPORT size_t rten_hip_minmax_stats_bytes(void) { return 2 * dql::kStatSlots * sizeof(unsigned); }

namespace {
__global__ void stats_reset_kernel(unsigned *stats, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // one thread per slot word
    if (i < count * 2 * dql::kStatSlots) stats[i] = (i % (2 * dql::kStatSlots)) < dql::kStatSlots ? 0xffffffffu : 0u; // minima | maxima
}
} // namespace

RTEN_EXPORT int32_t rten_hip_minmax_stats_reset(rten_hip_ctx *ctx, void *stats, int32_t count) {
    RTEN_CHECK_CTX(ctx);
    if (!stats || count < 1) return RTEN_HIP_ERR_INVALID_VALUE;
    const int n = count * 2 * dql::kStatSlots;
    hipLaunchKernelGGL(stats_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned *)stats, count);
    RTEN_LAUNCH_CHECK(ctx, "stats_reset_kernel launch");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_dynamic_quantize_linear_staged_stats(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const float *x, const void *stats,
                                                                  void *staged, float *scale, uint8_t *zero_point, const float *mul_by, float *product) {
    RTEN_CHECK_CTX(ctx);
    if (!di || !x || !stats || !staged || !scale || !zero_point || (mul_by && !product)) return RTEN_HIP_ERR_INVALID_VALUE;
    const ConvGeom g = conv_geom(di);
    if (!g.ok || di->x_signed) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "quantize_staged: geometry not covered by the staged kernel (staged_bytes == 0)");
    const rten_hip_conv2d_desc *d = &di->conv;
    ProfScope ps(ctx, "dynamic_quantize_linear_staged_stats", 0.0, 4.0 * d->n * d->c * (double)d->h * d->w + (double)g.img);
    launch_quantize_stage(ctx, d, g, x, (const float *)stats, -1, staged, rten_effective_pad_mode(di), scale, zero_point, one_product(mul_by, product));
    RTEN_LAUNCH_CHECK(ctx, "i8_quantize_stage_kernel launch");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_dynamic_quantize_linear_staged_products(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const float *x, const void *stats,
                                                                     void *staged, float *scale, uint8_t *zero_point, int32_t count,
                                                                     const float *const *mul_by, float *const *product) {
    RTEN_CHECK_CTX(ctx);
    if (!di || !x || !staged || !scale || !zero_point || count < 0 || (count > 0 && (!mul_by || !product))) return RTEN_HIP_ERR_INVALID_VALUE;
    if (count > kMaxProducts) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "quantize_staged_products: at most 4 scale products per launch");
    ScaleProducts sp = {};
    sp.count = count;
    for (int i = 0; i < count; i++) {
        if (!mul_by[i] || !product[i]) return RTEN_HIP_ERR_INVALID_VALUE;
        sp.mul_by[i] = mul_by[i];
        sp.product[i] = product[i];
    }
    const ConvGeom g = conv_geom(di);
    if (!g.ok || di->x_signed) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "quantize_staged: geometry not covered by the staged kernel (staged_bytes == 0)");
    const rten_hip_conv2d_desc *d = &di->conv;
    const int64_t n = (int64_t)d->n * d->c * d->h * d->w;
    if (stats) {
        ProfScope ps(ctx, "dynamic_quantize_linear_staged_stats", 0.0, 4.0 * n + (double)g.img);
        launch_quantize_stage(ctx, d, g, x, (const float *)stats, -1, staged, rten_effective_pad_mode(di), scale, zero_point, sp);
    } else {
        ProfScope ps(ctx, "dynamic_quantize_linear_staged", 0.0, 8.0 * n + (double)g.img);
        int nparts = 0;
        const float *ws = rten_dql_minmax(ctx, n, x, &nparts);
        if (!ws) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "dql: scratch allocation failed");
        launch_quantize_stage(ctx, d, g, x, ws, nparts, staged, rten_effective_pad_mode(di), scale, zero_point, sp);
    }
    RTEN_LAUNCH_CHECK(ctx, "i8_quantize_stage_kernel launch");
    return RTEN_HIP_OK;
}

namespace {
struct QOutArgs { // the consumer side of rten_hip_conv2d_int8_qout
    const rten_hip_conv2d_int8_desc *next;
    void *sync, *staged;
    float *scale;
    uint8_t *zp;
    const float *mul_by;
    float *product;
};
int32_t i8_fast_conv_impl(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x, const void *w, const void *x_zp, const void *w_zp,
                          const float *scale, const float *bias, const float *residual, uint32_t flags, void *y, void *stats, const QOutArgs *qo);
} // namespace

int32_t rten_i8_fast_conv(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x, const void *w, const void *x_zp,
                          const void *w_zp, const float *scale, const float *bias, const float *residual, uint32_t flags, void *y, void *stats) {
    return i8_fast_conv_impl(ctx, di, x, w, x_zp, w_zp, scale, bias, residual, flags, y, stats, nullptr);
}

namespace {
int32_t i8_fast_conv_impl(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x, const void *w, const void *x_zp, const void *w_zp,
                          const float *scale, const float *bias, const float *residual, uint32_t flags, void *y, void *stats, const QOutArgs *qo) {
    const rten_hip_conv2d_desc *d = &di->conv;
    const ConvGeom cg = conv_geom(di);
    if (!cg.ok)
        return (di->weights_packed || di->x_staged) ? rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv_int8: staged operands for an unsupported geometry")
                                                    : RTEN_HIP_ERR_UNSUPPORTED;
    const size_t wbytes = di->weights_packed ? 0 : up256((size_t)d->o * cg.Kp) + up256((size_t)d->o * 4);
    // cross-workgroup K split (KS kernels): a slab for up to kKsMax int32 partials of the (tile-padded) output, for the launches the dispatcher may split
    constexpr int kKsMax = 4;
    const long long n_cols = (long long)d->n * cg.P;
    const long long t64 = (long long)((d->o + 63) / 64) * ((n_cols + 63) / 64);
    static const bool ks_knob = getenv("RTEN_I8_KS") != nullptr && atoi(getenv("RTEN_I8_KS")) >= 2; // (no slab, no scratch growth, unless the measurement knob is set)
    const bool ks_candidate = ks_knob && !qo && scale && !(flags & RTEN_HIP_CONV_RESIDUAL) && cg.Kp >= 16 * 64 && t64 <= 3 * (long long)ctx->num_cus && ctx->split_counters;
    const size_t slab_bytes = ks_candidate ? (size_t)kKsMax * (size_t)((d->o + 127) / 128 * 128) * (size_t)((n_cols + 127) / 128 * 128) * 4 : 0;
    const size_t offA = 4096, offB = offA + wbytes, offS = offB + (di->x_staged ? 0 : up256(cg.img)), total = offS + slab_bytes;
    char *sc = (char *)rten_scratch(ctx, total);
    if (!sc) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "int8 staging allocation failed (or attempted during graph capture)");
    const uint8_t *Ap;
    const int *rsum;
    if (di->weights_packed) {
        Ap = (const uint8_t *)w;
        rsum = (const int *)((const char *)w + up256((size_t)d->o * cg.Kp));
    } else {
        Ap = (const uint8_t *)(sc + offA);
        rsum = (const int *)(sc + offA + up256((size_t)d->o * cg.Kp));
        hipLaunchKernelGGL(i8_pack_rows_kernel, dim3((unsigned)d->o), dim3(256), 0, ctx->stream, (const uint8_t *)w, (long long)cg.Kreal, 1ll, cg.Kreal, d->c,
                           d->kh * d->kw, cg.Cp, cg.Kp, d->o, di->w_signed ? 0u : 0x80u, (uint8_t *)Ap, (int *)rsum, cg.packed ? cg.nch : 0, cg.cpc, d->kw);
    }
    if (!di->x_staged && cg.packed) {
        const dim3 pgrid((unsigned)((d->out_w + 127) / 128), (unsigned)((cg.Hp + 1) / 2), (unsigned)d->n);
        RTEN_PACKED_DISPATCH(i8_pad_packed_kernel, d->c, cg.nch, pgrid, (const uint8_t *)x, (uint8_t *)(sc + offB), d->h, d->w, cg.Hp, d->out_w, d->kw, d->stride_w, d->dil_w,
                             d->pads[0], d->pads[1], di->x_signed ? 0u : 0x80u, (const uint8_t *)x_zp, di->x_signed, rten_effective_pad_mode(di));
    } else if (!di->x_staged)
    {
        const dim3 grid((unsigned)((d->h * d->w + 255) / 256), (unsigned)d->n, (unsigned)(cg.Cp / 16));
        const dim3 grid_small((unsigned)((cg.Hp * cg.Wp + 63) / 64), (unsigned)d->n, (unsigned)((cg.Cp + 63) / 64));
        const bool vec = (d->h * d->w) % 4 == 0 && ((uintptr_t)x & 3) == 0;
#define PAD_ARGS (const uint8_t *)x, (uint8_t *)(sc + offB), d->c, d->h, d->w, cg.Hp, cg.Wp, cg.Cp, d->pads[0], d->pads[1], di->x_signed ? 0u : 0x80u, (const uint8_t *)x_zp, di->x_signed, rten_effective_pad_mode(di)
        if (d->h * d->w < 128) hipLaunchKernelGGL(i8_nhwc_pad_kernel<0>, grid_small, dim3(256), 0, ctx->stream, PAD_ARGS);
        else if (vec) hipLaunchKernelGGL(i8_nhwc_pad_kernel<2>, grid, dim3(256), 0, ctx->stream, PAD_ARGS);
        else hipLaunchKernelGGL(i8_nhwc_pad_kernel<1>, grid, dim3(256), 0, ctx->stream, PAD_ARGS);
#undef PAD_ARGS
    }
    RTEN_LAUNCH_CHECK(ctx, "int8 staging launch");
    FastArgs g = {};
    g.A = Ap; g.B = di->x_staged ? (const uint8_t *)x : (const uint8_t *)(sc + offB);
    g.rsum = rsum; g.csum = nullptr;
    g.C = y;
    g.a_zp = di->w_zp_len ? (const uint8_t *)w_zp : nullptr;
    g.b_zp = (const uint8_t *)x_zp;
    g.scale = scale; g.bias = bias;
    g.res = (flags & RTEN_HIP_CONV_RESIDUAL) ? residual : nullptr;
    g.relu = (flags & RTEN_HIP_CONV_RELU) ? 1 : 0;
    g.M = d->o; g.N = d->n * cg.P; g.Kp = cg.Kp; g.Kreal = cg.Kreal;
    g.a_bytes = (unsigned)((size_t)d->o * cg.Kp); g.b_bytes = (unsigned)cg.img;
    g.c_rs = cg.P; g.c_ns = (long long)d->o * cg.P; g.Pn = cg.P;
    g.a_signed = di->w_signed; g.b_signed = di->x_signed;
    g.a_zp_len = di->w_zp_len; g.b_zp_len = x_zp ? 1 : 0;
    g.scale_len = scale ? 1 : 0;
    g.scale_per_row = (scale && di->scale_len > 1) ? 1 : 0;
    g.stats = scale ? (unsigned *)stats : nullptr;
    g.need_csum = (di->w_zp_len != 0 || !di->w_signed) ? 1 : 0; // weight zero point may be non-zero in the signed domain
    g.debug_flags = (ctx->debug & 0x200000) ? 1 : 0;
    g.conv = 1; g.OW = d->out_w; g.sy = d->stride_h; g.sx = cg.sx; g.Hp = cg.Hp; g.Wp = cg.Wp; g.Cp = cg.Cp;
    g.KH = d->kh; g.KW = cg.kw; g.dy = d->dil_h; g.dx = cg.dx; // (cg.sx / kw / dx: the virtual geometry of a few-channel packed image, else the convolution's own)
    if (slab_bytes) { g.ks = kKsMax; g.slab = (int *)(sc + offS); g.ks_counters = ctx->split_counters; }
    double out_bytes = 4.0 * d->o * g.N;
    if (qo && !qo->sync) {
        // Recompute form (no exchange block): launch 1 = this convolution, statistics only (nothing stored); launch 2 = the same convolution again, which
        // folds those statistics, quantizes in its epilogue and writes the consumer's staged image (+ the f32 tensor when `y` is wanted)
        if (!stats) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "conv2d_int8_qout (recompute form): a statistics block is required");
        const bool rt = !(g.debug_flags & 1) && g.a_zp == nullptr && g.a_signed && !g.need_csum;
        if (!rt || g.res) return RTEN_HIP_ERR_UNSUPPORTED; // (built for the single-row-term form without a residual: the caller runs the two operators)
        FastArgs g1 = g;
        g1.no_store = 1;
        const int32_t rc1 = dispatch_fast(ctx, g1, 2.0 * d->o * (double)g.N * cg.Kreal, (double)d->o * cg.Kreal + (double)d->n * d->c * d->h * d->w);
        if (rc1 != RTEN_HIP_OK) return rc1;
        const ConvGeom ng = conv_geom(qo->next);
        g.in_stats = (const unsigned *)stats;
        g.stats = nullptr;
        g.q_out = (uint8_t *)qo->staged;
        g.q_scale_out = qo->scale; g.q_zp_out = qo->zp; g.q_mul_by = qo->mul_by; g.q_product = qo->product;
        g.q_cb = ng.Cp / 16; g.q_Hp = ng.Hp; g.q_Wp = ng.Wp; g.q_pt = qo->next->conv.pads[0]; g.q_pl = qo->next->conv.pads[1];
        g.q_H = d->out_h; g.q_W = d->out_w; g.q_pad_mode = rten_effective_pad_mode(qo->next);
        g.q_bytes = (unsigned)ng.img;
        g.ks = 0; g.slab = nullptr;
        return dispatch_fast(ctx, g, 2.0 * d->o * (double)g.N * cg.Kreal,
                             (double)d->o * cg.Kreal + (double)d->n * d->c * d->h * d->w + (y ? out_bytes : 0.0) + (double)ng.img);
    }
    if (qo) {
        const ConvGeom ng = conv_geom(qo->next);
        g.sync = (unsigned *)qo->sync;
        g.fault = ctx->fault_dev;
        g.q_out = (uint8_t *)qo->staged;
        g.q_scale_out = qo->scale; g.q_zp_out = qo->zp; g.q_mul_by = qo->mul_by; g.q_product = qo->product;
        g.q_cb = ng.Cp / 16; g.q_Hp = ng.Hp; g.q_Wp = ng.Wp; g.q_pt = qo->next->conv.pads[0]; g.q_pl = qo->next->conv.pads[1];
        g.q_H = d->out_h; g.q_W = d->out_w; g.q_pad_mode = rten_effective_pad_mode(qo->next);
        g.q_bytes = (unsigned)ng.img;
        out_bytes = (y ? out_bytes : 0.0) + (double)ng.img;
    }
    // algorithmic bytes: i8 weights + u8 activations (once) + f32 output (+ f32 residual when the epilogue adds one); quantized-output form:
    // the consumer's staged u8 image instead of (or, when y is wanted too, besides) the f32 output
    return dispatch_fast(ctx, g, 2.0 * d->o * (double)g.N * cg.Kreal,
                         (double)d->o * cg.Kreal + (double)d->n * d->c * d->h * d->w + out_bytes + (g.res ? 4.0 * d->o * g.N : 0.0));
}
} // namespace

// ConvIntegerToFloat [+ bias] [+ residual] [+ Relu] AND the DynamicQuantizeLinear of the one convolution that consumes its output, in ONE
// launch (igemm_i8_fast_kernel, QO): see rten_hip.h.  RTEN_HIP_ERR_UNSUPPORTED when the geometry is not covered or the launch's
// workgroups cannot all be resident at once -- run rten_hip_conv2d_int8_stats + rten_hip_dynamic_quantize_linear_staged_stats then.
RTEN_EXPORT size_t rten_hip_grid_sync_bytes(void) { return kSyncWords * sizeof(unsigned); }

RTEN_EXPORT int32_t rten_hip_conv2d_int8_qout(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const void *x, const void *w, const void *x_zp,
                                              const void *w_zp, const float *scale, const float *bias, const float *residual, uint32_t flags, float *y,
                                              void *stats, void *sync, const rten_hip_conv2d_int8_desc *next, void *next_staged, float *next_scale,
                                              uint8_t *next_zero_point, const float *mul_by, float *product) {
    RTEN_CHECK_CTX(ctx);
    if (!di || !x || !w || !scale || !stats || !next || !next_staged || !next_scale || !next_zero_point || (mul_by && !product))
        return RTEN_HIP_ERR_INVALID_VALUE; // (sync == NULL: the recompute form -- two launches, no grid-wide exchange, no residency requirement)
    const rten_hip_conv2d_desc *d = &di->conv, *nd = &next->conv;
    if (di->scale_len != 0 && di->scale_len != 1 && di->scale_len != d->o)
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "conv_int8: scale must be a scalar or have one value per output channel");
    if (di->w_zp_len != 0 && di->w_zp_len != 1 && di->w_zp_len != d->o) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Zero point has incorrect size");
    if (nd->n != d->n || nd->c != d->o || nd->h != d->out_h || nd->w != d->out_w)
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "conv2d_int8_qout: the consumer's input is not this convolution's output");
    const ConvGeom cg = conv_geom(di), ng = conv_geom(next);
    if (!cg.ok || !ng.ok || ng.packed || next->x_signed || !di->weights_packed || !di->x_staged || d->o % 16 != 0)
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv2d_int8_qout: staged operands, O % 16 == 0 and a consumer covered by the staged kernel only");
    if (d->n == 0 || d->out_h == 0 || d->out_w == 0) return RTEN_HIP_OK;
    const QOutArgs qo = {next, sync, next_staged, next_scale, next_zero_point, mul_by, product};
    return i8_fast_conv_impl(ctx, di, x, w, x_zp, w_zp, scale, bias, residual, flags, y, stats, &qo);
}

namespace {
__global__ void grid_sync_reset_kernel(unsigned *sync, int count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; // one thread per word
    if (i >= (long long)count * kSyncWords) return;
    const unsigned w = (unsigned)(i % kSyncWords);
    sync[i] = (w >= kSyncCtlWords && ((w - kSyncCtlWords) & 1u) == 0u) ? 0xffffffffu : 0u; // granule = {lo = 0xffffffff, hi = 0}
}
} // namespace

// Initialises `count` consecutive exchange blocks (once, when they are allocated; every launch leaves its block in this state).
RTEN_EXPORT int32_t rten_hip_grid_sync_reset(rten_hip_ctx *ctx, void *sync, int32_t count) {
    RTEN_CHECK_CTX(ctx);
    if (!sync || count < 1) return RTEN_HIP_ERR_INVALID_VALUE;
    // the reset ends with a stream synchronise (the sticky fault is cleared after what is in flight has drained): inside a capture that would
    // invalidate the capture instead of resetting anything -- refused, not attempted
    if (ctx->capturing) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "grid_sync_reset: not while a graph capture is active on this context");
    const long long n = (long long)count * kSyncWords;
    hipLaunchKernelGGL(grid_sync_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned *)sync, count);
    RTEN_LAUNCH_CHECK(ctx, "grid_sync_reset_kernel launch");
    if (ctx->fault_host) { // the caller starts over: the context's sticky fault goes with the blocks' flags (after what is in flight has drained)
        RTEN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        *ctx->fault_host = 0u;
    }
    return RTEN_HIP_OK;
}

// Time-out flags of `count` consecutive barrier blocks (non-zero: a quantized-output launch gave up waiting for its grid and its results are void).
RTEN_EXPORT int32_t rten_hip_grid_sync_timeouts(rten_hip_ctx *ctx, const void *sync, int32_t count, int32_t *timeouts) {
    RTEN_CHECK_CTX(ctx);
    if (!sync || count < 1 || !timeouts) return RTEN_HIP_ERR_INVALID_VALUE;
    std::vector<unsigned> host((size_t)count * kSyncWords);
    RTEN_HIP_TRY(ctx, hipMemcpyAsync(host.data(), sync, host.size() * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    RTEN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int n = 0;
    for (int i = 0; i < count; i++) n += host[(size_t)i * kSyncWords + 8] != 0;
    *timeouts = n;
    return RTEN_HIP_OK;
}


// DynamicQuantizeLinear -> ConvInteger -> Cast -> Mul(x_scale, w_scale) [-> Add(bias)] [-> Add(residual)] [-> Relu] of a dynamically
// quantized graph in ONE launch, for pointwise (1x1, stride 1, no padding, groups 1) convolutions whose input statistics were
// accumulated by the producing kernel (rten_hip_conv2d_int8_stats / this function): the quantizer runs inside the B loader of the
// integer GEMM (igemm_i8_fast_kernel, BQ).  `w` must be prepacked (rten_hip_conv2d_int8_prepack), `w_scale` is the scalar or
// per-output-channel weight scale (di->scale_len), `x_scale_out` / `x_zero_point_out` receive DynamicQuantizeLinear's own outputs
// (optional).  RTEN_HIP_ERR_UNSUPPORTED for any other geometry: run the staged sequence instead.
RTEN_EXPORT int32_t rten_hip_conv2d_int8_dql(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *di, const float *x, const void *in_stats, const void *w,
                                             const float *w_scale, const float *bias, const float *residual, uint32_t flags, float *y, void *out_stats,
                                             float *x_scale_out, uint8_t *x_zero_point_out) {
    RTEN_CHECK_CTX(ctx);
    if (!di || !x || !in_stats || !w || !w_scale || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    const rten_hip_conv2d_desc *d = &di->conv;
    const ConvGeom cg = conv_geom(di);
    const bool pointwise = d->kh == 1 && d->kw == 1 && d->stride_h == 1 && d->stride_w == 1 && d->groups == 1 && d->pads[0] == 0 && d->pads[1] == 0 &&
                           d->pads[2] == 0 && d->pads[3] == 0;
    if (!cg.ok || !pointwise || !di->weights_packed || di->x_signed || di->w_zp_len != 0 || d->c % 64 != 0 ||
        (long long)d->n * d->c * d->h * d->w * 4 >= (1ll << 31))
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "conv2d_int8_dql: pointwise stride-1 unpadded convolutions with prepacked weights and C % 64 == 0 only");
    if (di->scale_len != 0 && di->scale_len != 1 && di->scale_len != d->o)
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "conv_int8: scale must be a scalar or have one value per output channel");
    if (d->n == 0 || d->out_h == 0 || d->out_w == 0) return RTEN_HIP_OK;
    FastArgs g = {};
    g.A = (const uint8_t *)w;
    g.rsum = (const int *)((const char *)w + up256((size_t)d->o * cg.Kp));
    g.C = y;
    g.xf = x; g.in_stats = (const unsigned *)in_stats; g.HW = d->h * d->w; g.Cin = d->c;
    g.xs_out = x_scale_out; g.xz_out = x_zero_point_out;
    g.scale = w_scale; g.bias = bias;
    g.res = (flags & RTEN_HIP_CONV_RESIDUAL) ? residual : nullptr;
    g.relu = (flags & RTEN_HIP_CONV_RELU) ? 1 : 0;
    g.M = d->o; g.N = d->n * cg.P; g.Kp = cg.Kp; g.Kreal = cg.Kreal;
    g.a_bytes = (unsigned)((size_t)d->o * cg.Kp); g.b_bytes = (unsigned)((size_t)d->n * d->c * d->h * d->w * 4);
    g.c_rs = cg.P; g.c_ns = (long long)d->o * cg.P; g.Pn = cg.P;
    g.a_signed = di->w_signed; g.b_signed = 0;
    g.a_zp_len = 0; g.b_zp_len = 1;
    g.scale_len = 1;
    g.scale_per_row = di->scale_len > 1 ? 1 : 0;
    g.stats = (unsigned *)out_stats;
    g.need_csum = di->w_signed ? 0 : 1;
    g.conv = 1; g.OW = d->out_w; g.sy = 1; g.sx = 1; g.Hp = d->h; g.Wp = d->w; g.Cp = cg.Cp; g.KH = 1; g.KW = 1; g.dy = 1; g.dx = 1;
    // algorithmic bytes: i8 weights + f32 activations (read once) + f32 output (+ f32 residual)
    return dispatch_fast(ctx, g, 2.0 * d->o * (double)g.N * cg.Kreal,
                         (double)d->o * cg.Kreal + 4.0 * d->n * d->c * d->h * d->w + 4.0 * d->o * g.N + (g.res ? 4.0 * d->o * g.N : 0.0));
}
