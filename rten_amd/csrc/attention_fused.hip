// Fused scaled-dot-product attention for gfx950: one kernel per (batch, head, 128-query tile) -- QK^T, mask add, row
// softmax and PV without the [B*H, S, T] score tensor ever leaving the CU.
//
// Replaces sdpa_head / sdpa_multi_head (src/ops/attention.rs:518-626) and the FusedMatMul(alpha) -> AddSoftmax -> MatMul
// chain of BERT-style graphs for head size 64 and key length <= 128; other shapes use the composed path (attention.hip).
//
// Bit-identical to the composed path and to the oracle, by construction:
//   * scores: v_mfma_f32_32x32x2_f32 over d = 0..63 in order (one depth block), then `acc * scale` (the GEMM's
//     beta == 0, alpha != 1 store form, simd_generic.rs:378-414), then `+ mask` as a separate add (attention.rs:59-61);
//   * softmax: max from f32::MIN, ReducedRangeExp, the 16-lane (AVX-512) ordered partial sums and their left-to-right
//     fold, `e * (1/sum)`, optional NaN flush -- the same operation sequence as rowwise.hip's softmax_kernel;
//   * PV: v_mfma over t = 0..T-1 in order (T <= 128 < kc: one depth block), plain store.
// Mapping: 256 threads; wave w owns query rows [32w, 32w+32) and ALL key columns, so the row softmax needs no
// cross-wave traffic: a row's T scores sit in 4 accumulator blocks x 32 lanes of one half-wave.  Q and K are staged
// in LDS as [d-quad][row][4] (16-byte global loads, k-contiguous operands like the GEMM's row-major-A image); the
// probabilities go back through LDS ([t][32 rows + 1] per wave) to become the A operand of PV; V is staged [t][64].
#include "internal.h"
#include "vecmath.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SQ = 128; // query rows per workgroup
constexpr int TT = 128; // key columns (upper bound; shorter T is zero-filled and masked)
constexpr int HD = 64;  // head size (q/k depth and v width)
constexpr int PLD = 33; // row pitch of the per-wave probability panel [t][32 rows]: odd -> conflict-free transposed stores

struct SdpaArgs {
    const float *q, *k, *v, *mask;
    float *out;
    int heads, s, t;
    long long q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
    long long mask_bs, mask_rs;
    float scale;
    int flush_nan;
    int s_tiles;
};

__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

__global__ __launch_bounds__(256, 1) void sdpa_fused_kernel(const SdpaArgs p) {
    // phase 1: Qs [16][SQ][4] + Ks [16][TT][4] = 64 KB; phase 3: Ps 4 x [TT][PLD] (66 KB, over Qs/Ks) + Vs [TT][HD] (32 KB)
    __shared__ __attribute__((aligned(16))) float smem[4 * TT * PLD + TT * HD];
    float *const Qs = smem, *const Ks = smem + 16 * SQ * 4;
    float *const Vs = smem + 4 * TT * PLD;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int st = blockIdx.x % p.s_tiles, bh = blockIdx.x / p.s_tiles;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int s0 = st * SQ;
    const float *qb = p.q + (long long)b * p.q_bs + (long long)h * p.q_hs;
    const float *kb = p.k + (long long)b * p.k_bs + (long long)h * p.k_hs;
    const float *vb = p.v + (long long)b * p.v_bs + (long long)h * p.v_hs;
    float *ob = p.out + (long long)b * p.o_bs + (long long)h * p.o_hs;

    // ---- stage Q, K ([d-quad][row][4]) and V ([t][64]); rows past S / T are zero
#pragma unroll
    for (int i = 0; i < SQ * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (s0 + row < p.s) v = *reinterpret_cast<const f32x4 *>(qb + (long long)(s0 + row) * p.q_rs + dq * 4);
        *reinterpret_cast<f32x4 *>(Qs + (dq * SQ + row) * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < p.t) v = *reinterpret_cast<const f32x4 *>(kb + (long long)row * p.k_rs + dq * 4);
        *reinterpret_cast<f32x4 *>(Ks + (dq * TT + row) * 4) = v;
    }
    f32x4 vreg[TT * 16 / 256]; // V goes to LDS after phase 1 (its region overlaps nothing, but keep the loads early)
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < p.t) vreg[i] = *reinterpret_cast<const f32x4 *>(vb + (long long)row * p.v_rs + dq * 4);
    }
    __syncthreads();

    // ---- phase 1: scores[32 rows of this wave][TT] = Q K^T, k = d in order
    f32x16 sc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) sc[j][r] = 0.f;
    {
        const float *Aq = Qs + (wave * 32 + l31) * 4 + half; // k = 2kk + half: same quad as 2kk, next element
        const float *Bk = Ks + l31 * 4 + half;
#pragma unroll
        for (int kk = 0; kk < HD / 2; kk++) {
            const float a = Aq[(kk >> 1) * SQ * 4 + ((2 * kk) & 3)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float bv = Bk[(kk >> 1) * TT * 4 + ((2 * kk) & 3) + j * 32 * 4];
                sc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, sc[j], 0, 0, 0);
            }
        }
    }
    __syncthreads(); // everyone is done with Qs / Ks: the region becomes the probability panels

    // ---- phase 2: row softmax in registers.  Register r of block j holds row acc_row(r) + 4*half, column j*32 + l31.
    float *const Ps = smem + wave * (TT * PLD);
    const int up = (lane & 32) | ((lane & 15) + 16); // the lane holding column + 16 of the same row
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = s0 + wave * 32 + acc_row(r) + 4 * half;
        const float *mrow = nullptr;
        if (p.mask) mrow = p.mask + (long long)b * p.mask_bs + (long long)(row < p.s ? row : 0) * p.mask_rs;
        float x[4], mx = -3.40282347e+38f; // f32::MIN (softmax.rs:181)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = j * 32 + l31;
            float v = sc[j][r] * p.scale;                 // gemm store form: t * alpha
            if (mrow && col < p.t) v = v + mrow[col];     // `*qk += m`
            x[j] = v;
            if (col < p.t) mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64)); // stays inside the 32-lane half
        // exp, then the 16-lane ordered partial sums: lane l < 16 adds e[l], e[l+16], e[l+32], ... in that order
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = j * 32 + l31;
            const float e = col < p.t ? vm::exp_reduced(x[j] - mx) : 0.f;
            x[j] = e;
            const float e_up = __shfl(e, up, 64);
            a = a + e;    // column j*32 + l      (a masked column adds +0: sums of exps are >= 0, so the bits do not change)
            a = a + e_up; // column j*32 + l + 16
        }
        float ssum = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) ssum = ssum + __shfl(a, (lane & 32) | k2, 64);
        const float inv = 1.0f / ssum;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float pr = x[j] * inv;
            if (p.flush_nan && !(pr == pr)) pr = 0.f;
            // transposed store: panel[t][row]; columns >= T hold 0 * inv (or NaN when the row is all masked): zero them
            Ps[(j * 32 + l31) * PLD + acc_row(r) + 4 * half] = (j * 32 + l31 < p.t) ? pr : 0.f;
        }
    }
    // V into LDS
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        *reinterpret_cast<f32x4 *>(Vs + row * HD + dq * 4) = vreg[i];
    }
    __syncthreads();

    // ---- phase 3: out[32 rows][64] = P V, k = t in order (columns >= T contribute p = 0 exactly as if absent? no: see below)
    // Only t < T may enter the chain: an fma with a zero product still leaves the accumulator unchanged (x + 0*v = x for
    // finite v; V rows >= T are zero-filled, so 0*0), hence looping to the padded TT is bit-neutral.
    f32x16 oc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) oc[j][r] = 0.f;
    {
        const float *Ap = Ps + half * PLD + l31;
        const float *Bv = Vs + half * HD + l31;
        const int kend = (p.t + 1) / 2;
        for (int kk = 0; kk < kend; kk++) {
            const float a = Ap[2 * kk * PLD];
#pragma unroll
            for (int j = 0; j < 2; j++) oc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bv[2 * kk * HD + j * 32], oc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = s0 + wave * 32 + acc_row(r) + 4 * half;
            if (row < p.s) ob[(long long)row * p.o_rs + j * 32 + l31] = oc[j][r];
        }
}

} // namespace

// Returns RTEN_HIP_ERR_UNSUPPORTED when the shape is not covered (the caller falls back to the composed path).
int32_t rten_sdpa_fused(rten_hip_ctx *ctx, const rten_hip_sdpa_desc *d, const float *q, const float *k, const float *v, const float *mask, float *out) {
    auto al16 = [](const void *p) { return ((uintptr_t)p & 15u) == 0; };
    if (d->d != HD || d->dv != HD || d->t > TT || d->t < 1) return RTEN_HIP_ERR_UNSUPPORTED;
    const int64_t strides[] = {d->q_bs, d->q_hs, d->q_rs, d->k_bs, d->k_hs, d->k_rs, d->v_bs, d->v_hs, d->v_rs};
    for (int64_t s : strides)
        if (s % 4 != 0) return RTEN_HIP_ERR_UNSUPPORTED;
    if (!al16(q) || !al16(k) || !al16(v)) return RTEN_HIP_ERR_UNSUPPORTED;
    SdpaArgs a = {};
    a.q = q; a.k = k; a.v = v; a.mask = mask; a.out = out;
    a.heads = d->heads; a.s = d->s; a.t = d->t;
    a.q_bs = d->q_bs; a.q_hs = d->q_hs; a.q_rs = d->q_rs; a.k_bs = d->k_bs; a.k_hs = d->k_hs; a.k_rs = d->k_rs;
    a.v_bs = d->v_bs; a.v_hs = d->v_hs; a.v_rs = d->v_rs; a.o_bs = d->o_bs; a.o_hs = d->o_hs; a.o_rs = d->o_rs;
    a.mask_bs = d->mask_batch_stride; a.mask_rs = d->mask_row_stride;
    a.scale = d->scale; a.flush_nan = d->flush_nan_to_zero;
    a.s_tiles = (d->s + SQ - 1) / SQ;
    const long long wgs = (long long)d->batch * d->heads * a.s_tiles;
    const double flops = 2.0 * d->batch * d->heads * (double)d->s * d->t * (d->d + d->dv);
    ProfScope ps(ctx, "sdpa_fused_kernel", flops, 4.0 * d->batch * d->heads * ((double)d->s * (d->d + d->dv) + (double)d->t * (d->d + d->dv)));
    hipLaunchKernelGGL(sdpa_fused_kernel, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, a);
    RTEN_LAUNCH_CHECK(ctx, "sdpa_fused_kernel launch");
    return RTEN_HIP_OK;
}
