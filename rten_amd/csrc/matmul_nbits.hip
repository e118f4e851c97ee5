// MatMulNBits: f32 LHS times a 4-bit block-quantised RHS (src/ops/matmul/contrib.rs:21-106, rten-gemm/src/block_quant.rs).
// quant [N][K/bs][bs/2] bytes (element 2j low nibble, 2j+1 high nibble of byte j), scales [N][K/bs], zero point 8;
// a weight is (float)(q - 8) * scale, one rounded multiply (packing.rs:300-312, block_quant.rs:243-262).
//
//  rows > 1 (contrib.rs:86-100): the reference dequantises while packing and runs its ordinary f32 GEMM.  Here one kernel
//      expands the weights to f32 [N][K] in the auxiliary scratch and the MFMA GEMM reads them through strides (B[k][n] at
//      n*K + k) -- the same k-ordered FMA chain in 256-deep blocks, so results are bit-identical for block sizes <= 256.
//  rows == 1 (block_quant.rs:166-389, the AVX-512 instantiation): 64 accumulator slots per column, slot s owning
//      k = s, s + 64, ... with one FMA per element; slots are then folded (s ^ 16, s ^ 32, then halves 8, 4, 2, 1) and a
//      scalar tail handles K % 128.  64 / SPL lanes share a column, each owning SPL adjacent slots (SPL / 2 bytes of every
//      64-element step); the LHS sits in LDS, shared by the four waves of a workgroup.
#include "internal.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ float nib(unsigned word, int e, float scale) { return (float)((int)((word >> (4 * e)) & 0xFu) - 8) * scale; }

// One thread expands one 32-bit word (8 weights).
__global__ __launch_bounds__(256) void dequant4_kernel(long long words, int K, int bs, const unsigned *__restrict__ quant, const float *__restrict__ scales,
                                                        float *__restrict__ out) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= words) return;
    const long long e0 = g * 8; // flat element index n*K + k; 8 | bs so the word lies inside one block
    const unsigned wv = quant[g];
    const float s = scales[e0 / bs];
    float4 lo, hi;
    lo.x = nib(wv, 0, s); lo.y = nib(wv, 1, s); lo.z = nib(wv, 2, s); lo.w = nib(wv, 3, s);
    hi.x = nib(wv, 4, s); hi.y = nib(wv, 5, s); hi.z = nib(wv, 6, s); hi.w = nib(wv, 7, s);
    float4 *o = reinterpret_cast<float4 *>(out + e0);
    o[0] = lo;
    o[1] = hi;
}

// (Round 5 measured the obvious alternative -- the 4 * SPL adjacent columns of a workgroup fetched with 16-byte coalesced loads into LDS, lanes reading their
//  1-4 bytes per step from there; same slots, chains and fold, bit-identical -- and it is SLOWER: 11.3 vs 6.6 us at N = K = 4096, 22.5 vs 12.2 us at
//  N = 14336 (profiles/r08/matmul_nbits_staging_experiment.txt): the launch is one workgroup per compute unit, so a load phase followed by a compute phase
//  serialises what the direct form below overlaps with its U-deep register prefetch.  The kernel is bound by VALU issue and the launch floor, not by its
//  small loads.  Commit "MatMulNBits decode: operands staged through LDS" has the code.)
constexpr int GEMV_CHUNK = 8192; // LHS elements staged in LDS at a time (32 KB, shared by the four waves of a workgroup)

// SPL = accumulator slots per lane (8, 4 or 2): 64 / SPL lanes share a column, a wave covers SPL columns, a workgroup 4 * SPL.
// Fewer slots per lane = more waves for the same matrix (the per-slot FMA chains are sequential in k, so a column cannot be split
// any other way); the launcher picks the largest SPL that still puts enough waves on the chip.
template <int SPL> struct QWord;
template <> struct QWord<8> { typedef unsigned T; };
template <> struct QWord<4> { typedef unsigned short T; };
template <> struct QWord<2> { typedef unsigned char T; };

typedef float v2f __attribute__((ext_vector_type(2)));

// GEMV_U = 64-element steps whose weight words are in flight together.  The kernel is bound by VALU issue (one wave instruction per
// four cycles per SIMD), so the inner loop is kept to: 3 mask ops per word, one v_cvt_f32_ubyteN per weight, and packed FMAs
// (two weights per v_pk_fma_f32) for both the dequantisation and the accumulation; lbs = log2(block size).
template <int SPL, int GEMV_U>
__global__ __launch_bounds__(256) void gemv4_kernel(int K, int N, int lbs, const float *__restrict__ a, const unsigned char *__restrict__ quant,
                                                     const float *__restrict__ scales, float *__restrict__ y) {
    typedef typename QWord<SPL>::T W;
    constexpr int LPC = 64 / SPL; // lanes per column
    __shared__ __attribute__((aligned(16))) float lhs[GEMV_CHUNK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPC;
    const int col = (blockIdx.x * 4 + wave) * SPL + lane / LPC;
    const bool live = col < N;
    const int c = live ? col : N - 1;
    const int kb = K >> lbs;
    const float *arow = a + (long long)blockIdx.y * K;
    const unsigned char *qcol = quant + (long long)c * (K / 2);
    const float *scol = scales + (long long)c * kb;
    const int kv = K - K % 128;

    v2f acc[SPL / 2];
#pragma unroll
    for (int e = 0; e < SPL / 2; e++) acc[e] = (v2f){0.f, 0.f};

    // one 64-element step: weight word wv, block scale s, LHS at lp
    auto step = [&](unsigned wv, float s, const float *lp) {
        v2f l[SPL / 2];
        if constexpr (SPL == 2) {
            l[0] = *reinterpret_cast<const v2f *>(lp);
        } else {
#pragma unroll
            for (int q = 0; q < SPL / 4; q++) {
                const float4 t = *reinterpret_cast<const float4 *>(lp + 4 * q);
                l[2 * q] = (v2f){t.x, t.y};
                l[2 * q + 1] = (v2f){t.z, t.w};
            }
        }
        // (q - 8) * s with one rounding: fma(q, s, -8 s) -- -8 s is exact, so the exact sum is (q - 8) s
        const unsigned lo = wv & 0x0F0F0F0Fu, hi = (wv >> 4) & 0x0F0F0F0Fu;
        const v2f s2 = (v2f){s, s}, m8 = (v2f){-8.0f * s, -8.0f * s};
#pragma unroll
        for (int t = 0; t < SPL / 2; t++) {
            const v2f q = (v2f){(float)((lo >> (8 * t)) & 0xFFu), (float)((hi >> (8 * t)) & 0xFFu)};
            const v2f w = __builtin_elementwise_fma(q, s2, m8);
            acc[t] = __builtin_elementwise_fma(l[t], w, acc[t]);
        }
    };

    W wv[GEMV_U];
    float sc[GEMV_U];
    for (int k0 = 0; k0 < kv; k0 += GEMV_CHUNK) {
        const int len = kv - k0 < GEMV_CHUNK ? kv - k0 : GEMV_CHUNK; // multiple of 128
        const int steps = len / 64, rounds = steps / GEMV_U;
        const unsigned char *qp = qcol + (k0 + sub * SPL) / 2; // + 32 bytes per step
        const int kl = k0 + sub * SPL;
        auto fetch = [&](int r) {
#pragma unroll
            for (int u = 0; u < GEMV_U; u++) {
                const int j = r * GEMV_U + u;
                wv[u] = *reinterpret_cast<const W *>(qp + j * 32);
                sc[u] = scol[(kl + j * 64) >> lbs];
            }
        };
        if (rounds) fetch(0); // in flight while the LHS chunk is staged
        __syncthreads();
        for (int i = threadIdx.x * 4; i < len; i += 1024) *reinterpret_cast<float4 *>(&lhs[i]) = *reinterpret_cast<const float4 *>(arow + k0 + i);
        __syncthreads();
        const float *lp = &lhs[sub * SPL];
        for (int r = 0; r < rounds; r++) {
            unsigned cw[GEMV_U];
            float cs[GEMV_U];
#pragma unroll
            for (int u = 0; u < GEMV_U; u++) { cw[u] = wv[u]; cs[u] = sc[u]; }
            if (r + 1 < rounds) fetch(r + 1);
#pragma unroll
            for (int u = 0; u < GEMV_U; u++) step(cw[u], cs[u], lp + (r * GEMV_U + u) * 64);
        }
        for (int j = rounds * GEMV_U; j < steps; j++) // fewer than U steps left in the chunk
            step(*reinterpret_cast<const W *>(qp + j * 32), scol[(kl + j * 64) >> lbs], lp + j * 64);
    }

    float accs[SPL];
#pragma unroll
    for (int t = 0; t < SPL / 2; t++) { accs[2 * t] = acc[t].x; accs[2 * t + 1] = acc[t].y; }

    // slot s = SPL * sub + e.  (acc0 + acc1) + (acc2 + acc3) over the four 16-lane accumulators = slots s ^ 16, then s ^ 32; then the
    // halving tree of the horizontal sum (8, 4, 2, 1).  A distance >= SPL is another lane of the column, a smaller one is inside the lane.
    constexpr int order[6] = {16, 32, 8, 4, 2, 1};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int w = order[i];
        if (w >= SPL) {
#pragma unroll
            for (int e = 0; e < SPL; e++) accs[e] = accs[e] + __shfl_xor(accs[e], w / SPL);
        } else {
#pragma unroll
            for (int e = 0; e < SPL; e++)
                if (e < w) accs[e] = accs[e] + accs[e + w];
        }
    }
    float out = accs[0];

    if (kv < K && sub == 0) { // scalar tail (block_quant.rs:351-377): separately rounded products
        float tail = 0.f;
        for (int k = kv; k < K; k += 2) {
            const unsigned byte = qcol[k / 2];
            const float s = scol[k >> lbs];
            const float lo = (float)((int)(byte & 0xFu) - 8) * s, hi = (float)((int)(byte >> 4) - 8) * s;
            const float p0 = __fmul_rn(arow[k], lo), p1 = __fmul_rn(arow[k + 1], hi);
            tail = tail + (p0 + p1);
        }
        out = out + tail;
    }
    if (live && sub == 0) y[(long long)blockIdx.y * N + col] = out;
}

template <int SPL>
void launch_gemv4(rten_hip_ctx *ctx, int u, int64_t batch, int k, int n, int bs, const float *a, const uint8_t *q, const float *sc, float *y) {
    const int lbs = __builtin_ctz((unsigned)bs);
    const dim3 grid((unsigned)((n + 4 * SPL - 1) / (4 * SPL)), (unsigned)batch);
    if (u == 32) hipLaunchKernelGGL((gemv4_kernel<SPL, 32>), grid, dim3(256), 0, ctx->stream, k, n, lbs, a, q, sc, y);
    else if (u == 16) hipLaunchKernelGGL((gemv4_kernel<SPL, 16>), grid, dim3(256), 0, ctx->stream, k, n, lbs, a, q, sc, y);
    else hipLaunchKernelGGL((gemv4_kernel<SPL, 8>), grid, dim3(256), 0, ctx->stream, k, n, lbs, a, q, sc, y);
}

} // namespace

// a [batch][rows][k]; b_quant [n][k / block_size][block_size / 2]; scales [n][k / block_size]; y [batch][rows][n].
RTEN_EXPORT int32_t rten_hip_matmul_nbits_f32(rten_hip_ctx *ctx, int64_t batch, int32_t rows, int32_t k, int32_t n, int32_t block_size, const float *a,
                                              const uint8_t *b_quant, const float *scales, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (batch < 0 || rows < 0 || k < 0 || n < 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "matmul_nbits: negative dimension");
    if (block_size < 16 || (block_size & (block_size - 1))) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "Unsupported K block size");
    if (k % block_size) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "Columns of first matrix does not match rows of second matrix");
    const long long out_len = (long long)batch * rows * n;
    if (out_len == 0) return RTEN_HIP_OK;
    if (!y) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "matmul_nbits: NULL output");
    if (k == 0) { // block_quant.rs:82-86
        RTEN_HIP_TRY(ctx, hipMemsetAsync(y, 0, (size_t)out_len * sizeof(float), ctx->stream));
        return RTEN_HIP_OK;
    }
    if (!a || !b_quant || !scales) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "matmul_nbits: NULL operand");
    if (rows == 1) {
        if (batch > 65535) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "matmul_nbits: more than 65535 vector rows");
        ProfScope ps(ctx, "gemv_f32_q4", 2.0 * batch * k * (double)n, (double)n * k / 2 + 4.0 * ((double)n * (k / block_size) + (double)batch * (k + n)));
        // waves = batch * n / SPL: the largest SPL (fewest instructions per weight) that still gives every SIMD a wave
        static const int forced = getenv("RTEN_HIP_GEMV_SPL") ? atoi(getenv("RTEN_HIP_GEMV_SPL")) : 0; // tuning only
        const long long cols = (long long)batch * n, want = 4LL * ctx->num_cus;
        const int spl = forced ? forced : (cols / 8 >= want ? 8 : cols / 4 >= want ? 4 : 2);
        static const int depth = getenv("RTEN_HIP_GEMV_U") ? atoi(getenv("RTEN_HIP_GEMV_U")) : 8;       // tuning only
        if (spl == 8) launch_gemv4<8>(ctx, depth, batch, k, n, block_size, a, b_quant, scales, y);
        else if (spl == 4) launch_gemv4<4>(ctx, depth, batch, k, n, block_size, a, b_quant, scales, y);
        else launch_gemv4<2>(ctx, depth, batch, k, n, block_size, a, b_quant, scales, y);
        RTEN_LAUNCH_CHECK(ctx, "gemv4_kernel launch");
        return RTEN_HIP_OK;
    }
    if (batch * rows > 0x7fffffffLL) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "matmul_nbits: too many rows");
    const long long words = (long long)n * k / 8;
    float *bm = (float *)rten_aux_scratch(ctx, (size_t)n * k * sizeof(float));
    if (!bm) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "matmul_nbits: weight expansion buffer allocation failed (or attempted during graph capture)");
    {
        ProfScope ps(ctx, "dequant_q4", 0.0, 4.5 * (double)n * k);
        hipLaunchKernelGGL(dequant4_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, words, k, block_size,
                           reinterpret_cast<const unsigned *>(b_quant), scales, bm);
        RTEN_LAUNCH_CHECK(ctx, "dequant4_kernel launch");
    }
    rten_hip_gemm_desc gd = {};
    gd.m = (int32_t)(batch * rows); gd.n = n; gd.k = k;
    gd.a_rs = k; gd.a_cs = 1;
    gd.b_rs = 1; gd.b_cs = k; // the expanded weights are [n][k]
    gd.ldc = n; gd.batch = 1;
    gd.alpha = 1.f; gd.beta = 0.f;
    return rten_gemm_f32_blocked(ctx, &gd, a, bm, nullptr, y);
}
