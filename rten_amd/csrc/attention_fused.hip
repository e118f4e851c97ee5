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
// Mapping: 256 threads; the kernel computes the TRANSPOSED score tile S^T = K Q^T (products commute, so every score is
// the same d-ordered chain): wave w owns query columns [32w, 32w+32) and all key rows, so one lane-pair (l, l+32) holds
// a whole softmax row in registers -- the max and the 16 ordered partial sums are plain in-lane register arithmetic
// (key index t = 32j + acc_row(r) + 4*half, hence t mod 16 is fixed per register), and the left-to-right fold of the 16
// partials crosses between the two half-waves four times.  The probabilities never leave registers either: cross-half
// swaps (v_permlane32_swap, one per register pair) re-pair them into the MFMA A operand (k = t, even t in lanes 0-31, odd t in lanes
// 32-63) for PV.  Q and K are staged in LDS as [d-quad][row][4] (16-byte global loads); V replaces K after phase 1.
// 64 KB of LDS and < 128 VGPRs: two workgroups per CU.
#include "internal.h"
#include "vecmath.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SQ = 128; // query rows per workgroup
constexpr int TT = 128; // key columns (upper bound; shorter T is zero-filled and masked)
constexpr int HD = 64;  // head size (q/k depth and v width)
struct SdpaArgs {
    const float *q, *k, *v, *mask;
    float *out;
    int heads, s, t;
    long long q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
    long long mask_bs, mask_rs;
    float scale;
    int flush_nan;
    int s_tiles;
    int debug; // ablation bits of the ABL instantiation (RTEN_HIP_DEBUG >> 24, measurement only): 1 no mask, 2 no exp, 4 no PV MFMAs, 8 no QK^T MFMAs, 16 no global loads, 32 no stores
};

__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// ABL: the ablation instantiation (phases switched off by p.debug bits: WRONG results, timing only; tools/probe_sdpa.py); the product launch is ABL = false
// and carries none of the tests.  MLDS: the additive mask row of a [B, 1, 1, T] mask (row stride 0: every query row of the batch item adds the same
// T values) is staged in LDS once per workgroup instead of being fetched by 64 broadcast global loads per lane.
// FULL: s is a multiple of 128 and t == 128 (the shape of BERT-base): every `row < s` / `key < t` test is true, and being per-lane tests (the key index
// depends on the half-wave) each of them costs an exec-mask branch -- 185 of the 5000 instructions of the general form; FLUSH: the NaN -> 0 option.
template <bool ABL, bool MLDS, bool FULL, bool FLUSH>
__global__ __launch_bounds__(256, 2) void sdpa_fused_kernel(const SdpaArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[16 * SQ * 4 + 16 * TT * 4 + (MLDS ? TT : 0)]; // Qs | Ks (phase 1) -> Vs [TT][HD] (phase 3) | mask row
    float *const Qs = smem, *const Ks = smem + 16 * SQ * 4, *const Vs = Ks;
    [[maybe_unused]] float *const Ms = smem + 16 * SQ * 4 + 16 * TT * 4;
    const int dbg = ABL ? p.debug : 0;
    auto in_t = [&](int key) { return FULL || key < p.t; };   // key column exists
    auto in_s = [&](int row) { return FULL || row < p.s; };   // query row exists

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int st = blockIdx.x % p.s_tiles, bh = blockIdx.x / p.s_tiles;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int s0 = st * SQ;
    const float *qb = p.q + (long long)b * p.q_bs + (long long)h * p.q_hs;
    const float *kb = p.k + (long long)b * p.k_bs + (long long)h * p.k_hs;
    const float *vb = p.v + (long long)b * p.v_bs + (long long)h * p.v_hs;
    float *ob = p.out + (long long)b * p.o_bs + (long long)h * p.o_hs;

    // ---- stage Q and K ([d-quad][row][4]); rows past S / T are zero.  V is fetched now and parked in registers.
#pragma unroll
    for (int i = 0; i < SQ * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (in_s(s0 + row) && !(dbg & 16)) v = *reinterpret_cast<const f32x4 *>(qb + (long long)(s0 + row) * p.q_rs + dq * 4);
        *reinterpret_cast<f32x4 *>(Qs + (dq * SQ + row) * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (in_t(row) && !(dbg & 16)) v = *reinterpret_cast<const f32x4 *>(kb + (long long)row * p.k_rs + dq * 4);
        *reinterpret_cast<f32x4 *>(Ks + (dq * TT + row) * 4) = v;
    }
    f32x4 vreg[TT * 16 / 256];
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (in_t(row) && !(dbg & 16)) vreg[i] = *reinterpret_cast<const f32x4 *>(vb + (long long)row * p.v_rs + dq * 4);
    }
    if constexpr (MLDS) {
        if (t < TT) Ms[t] = in_t(t) ? p.mask[(long long)b * p.mask_bs + t] : 0.f;
    }
    __syncthreads();

    // ---- phase 1: S^T[t][s] for this wave's 32 query columns: A = K rows (m = t), B = Q rows (n = s), k = d in order
    f32x16 sc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) sc[j][r] = 0.f;
    {
        const float *Ak = Ks + l31 * 4 + half;               // k = 2kk + half: same quad as 2kk, next element
        const float *Bq = Qs + (wave * 32 + l31) * 4 + half;
        if (!(dbg & 8)) {
#pragma unroll
        for (int kk = 0; kk < HD / 2; kk++) {
            const float bq = Bq[(kk >> 1) * SQ * 4 + ((2 * kk) & 3)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float ak = Ak[(kk >> 1) * TT * 4 + ((2 * kk) & 3) + j * 32 * 4];
                sc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak, bq, sc[j], 0, 0, 0);
            }
        }
        }
    }
    __syncthreads(); // Ks is free: park V there ([t][64]) for phase 3
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        *reinterpret_cast<f32x4 *>(Vs + row * HD + dq * 4) = vreg[i];
    }

    // ---- phase 2: softmax of query row s = s0 + 32*wave + l31, spread over this lane and lane ^ 32.
    // sc[j][r] is key t = 32j + acc_row(r) + 4*half.
    const int srow = s0 + wave * 32 + l31;
    const float *mrow = nullptr;
    if (p.mask && !MLDS && !(dbg & 1)) mrow = p.mask + (long long)b * p.mask_bs + (long long)(in_s(srow) ? srow : 0) * p.mask_rs;
    float mx = -3.40282347e+38f; // f32::MIN (softmax.rs:181)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tk = 32 * j + acc_row(r) + 4 * half;
            float v = sc[j][r] * p.scale;               // the GEMM's `t * alpha` store form
            if constexpr (MLDS) { if (in_t(tk)) v = v + Ms[tk]; }      // `*qk += m`, the row from LDS
            else { if (mrow && in_t(tk)) v = v + mrow[tk]; }           // `*qk += m`
            sc[j][r] = v;
            if (in_t(tk)) mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // exp and the 16 ordered partial sums of the reference's 16-lane SIMD order: partial l adds keys l, l+16, l+32, ...
    // This lane holds l = c + 4*half (registers r = c, c+8 of each block) and l = 8 + c + 4*half (r = 4+c, 12+c).
    float lo[4] = {0.f, 0.f, 0.f, 0.f}, hi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tk = 32 * j + acc_row(r) + 4 * half;
            sc[j][r] = in_t(tk) ? ((dbg & 2) ? sc[j][r] - mx : vm::exp_reduced(sc[j][r] - mx)) : 0.f; // a masked key adds +0: the sums are >= 0, bits unchanged
        }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int g = 0; g < 2; g++) // keys 32j + 16g + ...: ascending key order within every partial
#pragma unroll
            for (int c = 0; c < 4; c++) {
                lo[c] = lo[c] + sc[j][8 * g + c];
                hi[c] = hi[c] + sc[j][8 * g + 4 + c];
            }
    // fold partials 0..15 left to right: 0-3 live in half 0 (lo), 4-7 in half 1 (lo), 8-11 in half 0 (hi), 12-15 in half 1 (hi)
    auto add4 = [](float x, const float (&a)[4]) { return (((x + a[0]) + a[1]) + a[2]) + a[3]; };
    float run = add4(0.f, lo);                      // valid in half 0
    run = add4(__shfl_xor(run, 32, 64), lo);        // valid in half 1
    run = add4(__shfl_xor(run, 32, 64), hi);        // valid in half 0
    run = add4(__shfl_xor(run, 32, 64), hi);        // valid in half 1: the row sum
    const float other = __shfl_xor(run, 32, 64);
    const float ssum = half ? run : other;
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tk = 32 * j + acc_row(r) + 4 * half;
            float pr = sc[j][r] * inv;
            if (FLUSH && !(pr == pr)) pr = 0.f;
            sc[j][r] = in_t(tk) ? pr : 0.f; // padded keys: exactly 0 (also when the row is all masked and inv is NaN)
        }
    // re-pair into MFMA A operands: after the swaps register 4g+c holds keys (8g+c | 8g+c+1) in (half 0 | half 1) and
    // register 4g+c+1 holds keys (8g+4+c | 8g+4+c+1), c in {0, 2}: k-pairs in ascending key order.
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                // v_permlane32_swap(a, b): a' = {a.lo, b.lo}, b' = {a.hi, b.hi} (tools/probes/permlane32_swap.hip) -- exactly
                // the re-pairing wanted, one VALU instruction per register pair (operands copied to scalars first: a bit_cast
                // applied directly to an ext-vector element reads element 0 under this compiler)
                const float r0 = sc[j][4 * g + c], r1 = sc[j][4 * g + c + 1];
                const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1);
                const auto sw = __builtin_amdgcn_permlane32_swap(u0, u1, false, false);
                const unsigned n0 = sw[0], n1 = sw[1];
                sc[j][4 * g + c] = __uint_as_float(n0);       // keys (8g+c | 8g+c+1)
                sc[j][4 * g + c + 1] = __uint_as_float(n1);   // keys (8g+4+c | 8g+4+c+1)
            }
    __syncthreads(); // V is in LDS

    // ---- phase 3: out[s][dv] = sum_t P[s][t] V[t][dv], t ascending (padded keys are 0 * 0: bit-neutral)
    f32x16 oc[2];
#pragma unroll
    for (int jn = 0; jn < 2; jn++)
#pragma unroll
        for (int r = 0; r < 16; r++) oc[jn][r] = 0.f;
    {
        const float *Bv = Vs + half * HD + l31;
        if (!(dbg & 4)) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int kl = 0; kl < 16; kl++) { // k-pair kl of block j: keys 32j + 2kl, 32j + 2kl + 1
                const int t0 = 2 * kl, g = t0 >> 3, off = t0 & 7;
                const float a = sc[j][off < 4 ? 4 * g + off : 4 * g + (off - 4) + 1];
#pragma unroll
                for (int jn = 0; jn < 2; jn++) oc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bv[(32 * j + t0) * HD + jn * 32], oc[jn], 0, 0, 0);
            }
        } else { // (keep the probabilities alive so that phase 2 is not optimised away)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) oc[j & 1][r] += sc[j][r];
        }
    }
    // C layout of out: lane column = dv (jn*32 + l31), register r = query row acc_row(r) + 4*half of this wave's 32
#pragma unroll
    for (int jn = 0; jn < 2; jn++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = s0 + wave * 32 + acc_row(r) + 4 * half;
            if (in_s(row) && !((dbg & 32) && oc[jn][r] != 12345.f)) ob[(long long)row * p.o_rs + jn * 32 + l31] = oc[jn][r];
        }
}


// ---------------------------------------------------------------------------------------------------------------------------
// 16-query form of the kernel above for its FULL shape (head 64, 128 keys, s a multiple of 64; no mask or a [B, 1, 1, T] mask): round 5.
// Why: the 32-query waves of the kernel above are 1.5 per SIMD at BERT-base's batch 32 (384 workgroups x 4 waves on 1024 SIMDs: half of the SIMDs
// run two, the other half one) -- a quarter of the matrix pipe's time is lost to that imbalance.  Here a wave owns 16 queries
// (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain like the 32x32x2 form, tools/probes/mfma_16x16x4_order.hip), a workgroup 64: 768 workgroups,
// three waves on every SIMD, 48.5 KB of LDS each (three per compute unit, so one's softmax runs under another's MFMAs).
// Same operation sequence per output element as above (scores = d-ordered chain, * scale, + mask; max from f32::MIN; ReducedRangeExp; the 16
// ordered partial sums and their left-to-right fold; e * (1 / sum); PV = key-ordered chain) -- same bits.
// Layouts.  Scores are computed transposed, S^T = K Q^T, block j = 16 keys x 16 queries, with the block's key ROWS PERMUTED: MFMA row m = 4 q + r
// carries key 16 j + 4 r + q.  The accumulator register r of lane (query = lane % 16, q = lane / 16) is MFMA row 4 q + r, i.e. key
// 16 j + 4 r + q = 4 (4 j + r) + q: exactly the B operand (k index = q) of PV step 4 j + r when PV is computed transposed too,
// out^T = V^T P^T.  The probabilities never move between lanes.  The reference's partial l = key % 16 = 4 r + q sits in register r of quad q
// (summed over j in-lane, ascending); the left-to-right fold of the 16 partials walks quad 0 -> 1 -> 2 -> 3 four times (15 lane hops).
// LDS: K and Q rows as 64 floats with depth d = 16 a + 4 i + g stored at 16 a + 4 g + i, 16-byte groups XOR-swizzled by the row (one conflict-free
// ds_read_b128 = the operands of four MFMA steps); V as [key][64] with the 16-float groups XOR-swizzled by key % 4.
// out^T's accumulator holds 4 consecutive dv of one query per lane: 16-byte stores.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SQ16 = 64;
__device__ __forceinline__ int sd_at(int row, int pos) { return row * HD + (pos ^ ((row & 15) << 2)); }

// ABL: the ablation instantiation (p.debug bits, WRONG results, timing only: 1 no mask add, 2 no exp, 4 no PV MFMAs, 8 no QK^T MFMAs, 16 no global loads, 32 no stores,
// 64 no softmax arithmetic at all, 128 no LDS staging writes)
template <bool MASK, bool FLUSH, int ABL = 0> // (ABL is a compile-time constant: run-time switches changed the register allocation of every variant -- first attempt, 290 us)
__global__ __launch_bounds__(256, 3) void sdpa_fused16_kernel(const SdpaArgs p) {
    constexpr int dbg = ABL;
    __shared__ __attribute__((aligned(16))) float smem[SQ16 * HD + TT * HD + TT]; // Qs | Ks (phase 1) -> Vs (phase 3) | mask row
    float *const Qs = smem, *const Ks = smem + SQ16 * HD, *const Vs = Ks, *const Ms = smem + SQ16 * HD + TT * HD;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, quad = lane >> 4;
    const int st = blockIdx.x % p.s_tiles, bh = blockIdx.x / p.s_tiles;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int s0 = st * SQ16;
    const float *qb = p.q + (long long)b * p.q_bs + (long long)h * p.q_hs;
    const float *kb = p.k + (long long)b * p.k_bs + (long long)h * p.k_hs;
    const float *vb = p.v + (long long)b * p.v_bs + (long long)h * p.v_hs;
    float *ob = p.out + (long long)b * p.o_bs + (long long)h * p.o_hs;

    // ---- stage Q and K (depth-permuted rows); V is fetched now and parked in registers until K is dead
    f32x4 qreg[SQ16 * 16 / 256], kreg[TT * 16 / 256], vreg[TT * 16 / 256];
#pragma unroll
    for (int i = 0; i < SQ16 * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        qreg[i] = (dbg & 16) ? f32x4{0.5f, 0.25f, 0.125f, 1.f} : *reinterpret_cast<const f32x4 *>(qb + (long long)(s0 + row) * p.q_rs + dq * 4);
    }
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        kreg[i] = (dbg & 16) ? f32x4{0.5f, 0.25f, 0.125f, 1.f} : *reinterpret_cast<const f32x4 *>(kb + (long long)row * p.k_rs + dq * 4);
    }
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        vreg[i] = (dbg & 16) ? f32x4{0.5f, 0.25f, 0.125f, 1.f} : *reinterpret_cast<const f32x4 *>(vb + (long long)row * p.v_rs + dq * 4);
    }
    if constexpr (MASK) {
        if (t < TT) Ms[t] = p.mask[(long long)b * p.mask_bs + t];
    }
    // depth d = 4 dq + c is MFMA step dq, k index c: position 16 (dq / 4) + 4 c + dq % 4
    if (!(dbg & 128)) {
#pragma unroll
    for (int i = 0; i < SQ16 * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15, pos = 16 * (dq >> 2) + (dq & 3);
        Qs[sd_at(row, pos)] = qreg[i][0]; Qs[sd_at(row, pos + 4)] = qreg[i][1]; Qs[sd_at(row, pos + 8)] = qreg[i][2]; Qs[sd_at(row, pos + 12)] = qreg[i][3];
    }
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15, pos = 16 * (dq >> 2) + (dq & 3);
        Ks[sd_at(row, pos)] = kreg[i][0]; Ks[sd_at(row, pos + 4)] = kreg[i][1]; Ks[sd_at(row, pos + 8)] = kreg[i][2]; Ks[sd_at(row, pos + 12)] = kreg[i][3];
    }
    }
    __syncthreads();

    // ---- phase 1: S^T blocks j = 0..7 (16 keys x this wave's 16 queries), d ascending: 16 MFMA steps per block, the 8 blocks interleaved
    f32x4 sc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const int krow = 4 * (l15 & 3) + (l15 >> 2); // MFMA row l15 = 4 q' + r' of block j carries key 16 j + 4 r' + q'
        const int qrow = wave * 16 + l15;
#pragma unroll
        for (int a4 = 0; a4 < 4; a4++) { // MFMA steps 4 a4 .. 4 a4 + 3
            const f32x4 qf = *reinterpret_cast<const f32x4 *>(Qs + sd_at(qrow, 16 * a4 + 4 * quad));
            f32x4 kf[8];
#pragma unroll
            for (int j = 0; j < 8; j++) kf[j] = *reinterpret_cast<const f32x4 *>(Ks + sd_at(16 * j + krow, 16 * a4 + 4 * quad));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (dbg & 8) sc[j][i] += kf[j][i] + qf[i];
                    else sc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][i], qf[i], sc[j], 0, 0, 0);
                }
        }
    }
    __syncthreads(); // Ks is free: park V there, [key][64] with the 16-float groups swizzled by key % 4
#pragma unroll
    for (int i = 0; i < TT * 16 / 256; i++) {
        const int f = i * 256 + t, row = f >> 4, dq = f & 15;
        if (!(dbg & 128)) *reinterpret_cast<f32x4 *>(Vs + row * HD + ((dq * 4) ^ ((row & 3) << 4))) = vreg[i];
    }

    // ---- phase 2: softmax of query s0 + 16 wave + l15, spread over the four lanes l15 + 16 q; sc[j][r] is key 16 j + 4 r + quad
    if (!(dbg & 64)) {
    float mx = -3.40282347e+38f; // f32::MIN (softmax.rs:181)
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = sc[j][r] * p.scale;                                  // the GEMM's `t * alpha` store form
            if constexpr (MASK) { if (!(dbg & 1)) v = v + Ms[16 * j + 4 * r + quad]; } // `*qk += m`
            sc[j][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float part[4] = {0.f, 0.f, 0.f, 0.f}; // partial 4 r + quad of the reference's 16-lane order: keys 4 r + quad, + 16, + 32, ... ascending
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            sc[j][r] = (dbg & 2) ? sc[j][r] - mx : vm::exp_reduced(sc[j][r] - mx);
            part[r] = part[r] + sc[j][r];
        }
    // fold partials 0..15 left to right (partial 4 r + q lives in quad q)
    float pq[4][4]; // pq[q][r] = partial 4 r + q: every lane fetches the twelve it does not hold (independent lane reads), then folds all sixteen itself
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) pq[q][r] = __shfl(part[r], 16 * q + l15, 64);
    float ssum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) ssum = ssum + pq[q][r];
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float pr = sc[j][r] * inv;
            if (FLUSH && !(pr == pr)) pr = 0.f;
            sc[j][r] = pr;
        }
    }
    __syncthreads(); // V is in LDS

    // ---- phase 3: out^T[dv][s] = sum over keys (ascending) of V^T[dv][key] P^T[key][s]: step = 4 j + r takes keys 4 step + {0..3}, P^T's operand is sc[j][r] as it stands
    f32x4 oc[4];
#pragma unroll
    for (int jn = 0; jn < 4; jn++) oc[jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int key = 4 * (4 * j + r) + quad; // (key % 4 == quad: the swizzle term is per lane, constant)
            const float *vr = Vs + key * HD;
#pragma unroll
            for (int jn = 0; jn < 4; jn++) {
                if (dbg & 4) oc[jn][r] += vr[(16 * jn + l15) ^ (quad << 4)] + sc[j][r];
                else oc[jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[(16 * jn + l15) ^ (quad << 4)], sc[j][r], oc[jn], 0, 0, 0);
            }
        }
    // accumulator register r' of block jn: dv = 16 jn + 4 quad + r', query l15 -> four consecutive floats of one output row
    float *orow = ob + (long long)(s0 + wave * 16 + l15) * p.o_rs + 4 * quad;
#pragma unroll
    for (int jn = 0; jn < 4; jn++)
        if (!((dbg & 32) && oc[jn][0] != 12345.f)) *reinterpret_cast<f32x4 *>(orow + 16 * jn) = oc[jn];
}

// ---------------------------------------------------------------------------------------------------------------------------
// General form: head size HD in {32, 64, 128}, up to 128 * NCH <= 512 keys.  Same mapping and the same operation sequence per output
// element as the kernel above (sdpa_head, src/ops/attention.rs:518-562: scores = gemm(alpha = scale), += mask, softmax with the
// reference's ordered partial sums, out = gemm), with the key axis walked in chunks of 128 through ONE LDS buffer: K chunk c feeds
// score blocks 4c .. 4c+3 (all NJ = 4 NCH blocks of a query row stay in the registers of one lane pair -- the reference's softmax
// is three passes over the whole row, not an online one, so nothing may be rescaled), V chunk c then takes K's place for PV.
// PV over more than 256 keys crosses the reference's depth-block boundary (kc = 256, rten-gemm/src/lib.rs:630-633): keys 0-255 and
// 256-511 are separate MFMA chains from zero, combined with one separate add (lib.rs:1008-1013), as everywhere else in this backend.
// 128 KB of LDS at HD = 128 and up to ~400 VGPRs: one workgroup per CU.
extern __shared__ __attribute__((aligned(16))) float sdpa_smem[];
template <int HD, int NCH>
__global__ __launch_bounds__(256, 1) void sdpa_fused_general_kernel(const SdpaArgs p) {
    constexpr int NJ = 4 * NCH, DQ = HD / 4, NO = HD / 32;
    float *const Qs = sdpa_smem, *const Ks = sdpa_smem + DQ * SQ * 4, *const Vs = Ks; // Qs [DQ][SQ][4] | Ks [DQ][128][4] -> Vs [128][HD]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int st = blockIdx.x % p.s_tiles, bh = blockIdx.x / p.s_tiles;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int s0 = st * SQ;
    const float *qb = p.q + (long long)b * p.q_bs + (long long)h * p.q_hs;
    const float *kb = p.k + (long long)b * p.k_bs + (long long)h * p.k_hs;
    const float *vb = p.v + (long long)b * p.v_bs + (long long)h * p.v_hs;
    float *ob = p.out + (long long)b * p.o_bs + (long long)h * p.o_hs;

#pragma unroll
    for (int i = 0; i < SQ * DQ / 256; i++) {
        const int f = i * 256 + t, row = f / DQ, dq = f % DQ;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (s0 + row < p.s) v = *reinterpret_cast<const f32x4 *>(qb + (long long)(s0 + row) * p.q_rs + dq * 4);
        *reinterpret_cast<f32x4 *>(Qs + (dq * SQ + row) * 4) = v;
    }

    // ---- phase 1: S^T[t][s] for this wave's 32 query columns, key chunk by key chunk (k = d ascending: one depth block)
    f32x16 sc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) sc[j][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (c > 0) __syncthreads(); // everybody is done with the previous K chunk
#pragma unroll
        for (int i = 0; i < TT * DQ / 256; i++) {
            const int f = i * 256 + t, row = f / DQ, dq = f % DQ;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c * TT + row < p.t) v = *reinterpret_cast<const f32x4 *>(kb + (long long)(c * TT + row) * p.k_rs + dq * 4);
            *reinterpret_cast<f32x4 *>(Ks + (dq * TT + row) * 4) = v;
        }
        __syncthreads();
        const float *Ak = Ks + l31 * 4 + half;
        const float *Bq = Qs + (wave * 32 + l31) * 4 + half;
#pragma unroll 4
        for (int kk = 0; kk < HD / 2; kk++) {
            const float bq = Bq[(kk >> 1) * SQ * 4 + ((2 * kk) & 3)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float ak = Ak[(kk >> 1) * TT * 4 + ((2 * kk) & 3) + j * 32 * 4];
                sc[4 * c + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak, bq, sc[4 * c + j], 0, 0, 0);
            }
        }
    }

    // ---- phase 2: softmax of query row s0 + 32 wave + l31 (this lane and lane ^ 32); sc[j][r] is key 32j + acc_row(r) + 4 half
    const int srow = s0 + wave * 32 + l31;
    const float *mrow = nullptr;
    if (p.mask) mrow = p.mask + (long long)b * p.mask_bs + (long long)(srow < p.s ? srow : 0) * p.mask_rs;
    float mx = -3.40282347e+38f; // f32::MIN (softmax.rs:181)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tk = 32 * j + acc_row(r) + 4 * half;
            float v = sc[j][r] * p.scale;
            if (mrow && tk < p.t) v = v + mrow[tk];
            sc[j][r] = v;
            if (tk < p.t) mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float lo[4] = {0.f, 0.f, 0.f, 0.f}, hi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tk = 32 * j + acc_row(r) + 4 * half;
            sc[j][r] = tk < p.t ? vm::exp_reduced(sc[j][r] - mx) : 0.f;
        }
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                lo[c] = lo[c] + sc[j][8 * g + c];
                hi[c] = hi[c] + sc[j][8 * g + 4 + c];
            }
    auto add4 = [](float x, const float (&a)[4]) { return (((x + a[0]) + a[1]) + a[2]) + a[3]; };
    float run = add4(0.f, lo);
    run = add4(__shfl_xor(run, 32, 64), lo);
    run = add4(__shfl_xor(run, 32, 64), hi);
    run = add4(__shfl_xor(run, 32, 64), hi);
    const float other = __shfl_xor(run, 32, 64);
    const float ssum = half ? run : other;
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int tk = 32 * j + acc_row(r) + 4 * half;
            float pr = sc[j][r] * inv;
            if (p.flush_nan && !(pr == pr)) pr = 0.f;
            sc[j][r] = tk < p.t ? pr : 0.f;
        }
    // re-pair into MFMA A operands (register 4g+c: keys (8g+c | 8g+c+1) in (half 0 | half 1), register 4g+c+1: keys (8g+4+c | 8g+4+c+1)):
    // one cross-half exchange per register pair.  (__shfl_xor rather than the v_permlane32_swap of the kernel above: with all 256 VGPRs
    // holding the score row, the builtin's in-place pair was mis-allocated by this compiler -- two keys per chunk were dropped at HD = 128.)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const float r0 = sc[j][4 * g + c], r1 = sc[j][4 * g + c + 1];
                const float recv = __shfl_xor(half ? r0 : r1, 32, 64); // lower half receives the partner's r0, upper half the partner's r1
                sc[j][4 * g + c] = half ? recv : r0;
                sc[j][4 * g + c + 1] = half ? r1 : recv;
            }

    // ---- phase 3: out[s][dv] = sum_t P[s][t] V[t][dv], t ascending, V chunk by V chunk; depth blocks of 256 keys folded with separate adds
    // (HD = 128 with 512 keys: two passes over the output width, 64 columns each -- the score row already fills the VGPR file and
    // 2 x 64 more accumulator registers do not fit beside it; V chunks are staged again for the second pass, from L2)
    constexpr int NPASS = (HD == 128 && NCH == 4) ? 2 : 1, NOP = NO / NPASS;
#pragma unroll
    for (int pass = 0; pass < NPASS; pass++) {
        f32x16 oc[NOP], tot[NCH > 2 ? NOP : 1];
#pragma unroll
        for (int jn = 0; jn < NOP; jn++)
#pragma unroll
            for (int r = 0; r < 16; r++) oc[jn][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            __syncthreads(); // K chunk / previous V chunk no longer read
#pragma unroll
            for (int i = 0; i < TT * DQ / 256; i++) {
                const int f = i * 256 + t, row = f / DQ, dq = f % DQ;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c * TT + row < p.t) v = *reinterpret_cast<const f32x4 *>(vb + (long long)(c * TT + row) * p.v_rs + dq * 4);
                *reinterpret_cast<f32x4 *>(Vs + row * HD + dq * 4) = v;
            }
            __syncthreads();
            if constexpr (NCH > 2) {
                if (c == 2) { // keys 0..255 were the first depth block: park it, start the second chain from zero
#pragma unroll
                    for (int jn = 0; jn < NOP; jn++) {
                        tot[jn] = oc[jn];
#pragma unroll
                        for (int r = 0; r < 16; r++) oc[jn][r] = 0.f;
                    }
                }
            }
            const float *Bv = Vs + half * HD + pass * NOP * 32 + l31;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int kl = 0; kl < 16; kl++) {
                    const int t0 = 2 * kl, g = t0 >> 3, off = t0 & 7;
                    const float a = sc[4 * c + j][off < 4 ? 4 * g + off : 4 * g + (off - 4) + 1];
#pragma unroll
                    for (int jn = 0; jn < NOP; jn++) oc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bv[(32 * j + t0) * HD + jn * 32], oc[jn], 0, 0, 0);
                }
        }
        if constexpr (NCH > 2) {
            if (p.t > 256) { // (with <= 256 real keys the second chain only saw zero-filled rows: the reference has one depth block then)
#pragma unroll
                for (int jn = 0; jn < NOP; jn++)
#pragma unroll
                    for (int r = 0; r < 16; r++) oc[jn][r] = tot[jn][r] + oc[jn][r];
            } else {
#pragma unroll
                for (int jn = 0; jn < NOP; jn++) oc[jn] = tot[jn];
            }
        }
#pragma unroll
        for (int jn = 0; jn < NOP; jn++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = s0 + wave * 32 + acc_row(r) + 4 * half;
                if (row < p.s) ob[(long long)row * p.o_rs + (pass * NOP + jn) * 32 + l31] = oc[jn][r];
            }
    }
}

template <int HD, int NCH>
void launch_general(rten_hip_ctx *ctx, const SdpaArgs &a, long long wgs) {
    constexpr size_t lds = (size_t)(SQ + TT) * HD * sizeof(float);
    auto kern = sdpa_fused_general_kernel<HD, NCH>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds, ctx->stream, a);
}

} // namespace

// Returns RTEN_HIP_ERR_UNSUPPORTED when the shape is not covered (the caller falls back to the composed path).
int32_t rten_sdpa_fused(rten_hip_ctx *ctx, const rten_hip_sdpa_desc *d, const float *q, const float *k, const float *v, const float *mask, float *out, bool force) {
    auto al16 = [](const void *p) { return ((uintptr_t)p & 15u) == 0; };
    const bool fast = d->d == HD && d->dv == HD && d->t <= TT;                                             // BERT-base: head 64, <= 128 keys
    const bool general = d->d == d->dv && (d->d == 32 || d->d == 64 || d->d == 128) && d->t <= 4 * TT;    // head 32 / 64 / 128, <= 512 keys
    // Beyond 128 keys the whole score row of a query (up to 256 registers per lane) leaves room for ONE workgroup per CU, and the composed
    // GEMM / softmax / GEMM path is faster (profiles/r06/ops_microbench.json: 134 vs 209 us at head 128 x 512 keys, 66 vs 102 us at head 64 x
    // 256 keys; the one-kernel form wins at <= 128 keys: 23 vs 42 us at head 32): the automatic choice keeps the one-kernel form to <= 128
    // keys, `force` (rten_hip_set_sdpa_path(ctx, 2)) takes it wherever it is covered.
    if (d->t < 1 || (!fast && !(general && (force || d->t <= TT)))) return RTEN_HIP_ERR_UNSUPPORTED;
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
    const double bytes = 4.0 * d->batch * d->heads * ((double)d->s * (d->d + d->dv) + (double)d->t * (d->d + d->dv));
    // 16-query waves (sdpa_fused16_kernel): the FULL shape with no mask or a [B, 1, 1, T] mask, 16-byte aligned output rows (RTEN_HIP_DEBUG bit 0x200000: the 32-query form, A/B)
    const bool k16 = fast && d->t == TT && d->s % SQ16 == 0 && (!mask || d->mask_row_stride == 0) && !(ctx->debug & 0xa00000) &&
                     d->o_bs % 4 == 0 && d->o_hs % 4 == 0 && d->o_rs % 4 == 0 && al16(out);
    if (k16) {
        ProfScope ps(ctx, "sdpa_fused16_kernel", flops, bytes);
        a.s_tiles = d->s / SQ16;
        const dim3 grid((unsigned)((long long)d->batch * d->heads * a.s_tiles)), block(256);
        const bool flush = d->flush_nan_to_zero != 0;
        a.debug = 0;
#ifdef RTEN_ABLATION // measurement builds only (build.sh -DRTEN_ABLATION): the ablation instantiations compute WRONG results; the product library has none
        a.debug = (int)((unsigned)ctx->debug >> 24);
#endif
        if (a.debug) { // the ablation ladder of tools/probe_sdpa.py (timing only)
#ifdef RTEN_ABLATION
#define RTEN_SDPA16_ABL(V) case V: if (mask) hipLaunchKernelGGL((sdpa_fused16_kernel<true, false, V>), grid, block, 0, ctx->stream, a); else hipLaunchKernelGGL((sdpa_fused16_kernel<false, false, V>), grid, block, 0, ctx->stream, a); break;
            switch (a.debug & 255) {
                RTEN_SDPA16_ABL(1) RTEN_SDPA16_ABL(2) RTEN_SDPA16_ABL(4) RTEN_SDPA16_ABL(8) RTEN_SDPA16_ABL(12) RTEN_SDPA16_ABL(16) RTEN_SDPA16_ABL(32) RTEN_SDPA16_ABL(48)
                RTEN_SDPA16_ABL(64) RTEN_SDPA16_ABL(128) RTEN_SDPA16_ABL(176) RTEN_SDPA16_ABL(240) RTEN_SDPA16_ABL(76) RTEN_SDPA16_ABL(255)
            default: if (mask) hipLaunchKernelGGL((sdpa_fused16_kernel<true, false, 0>), grid, block, 0, ctx->stream, a); else hipLaunchKernelGGL((sdpa_fused16_kernel<false, false, 0>), grid, block, 0, ctx->stream, a);
            }
#undef RTEN_SDPA16_ABL
#endif
        }
        else if (mask) { if (flush) hipLaunchKernelGGL((sdpa_fused16_kernel<true, true>), grid, block, 0, ctx->stream, a); else hipLaunchKernelGGL((sdpa_fused16_kernel<true, false>), grid, block, 0, ctx->stream, a); }
        else { if (flush) hipLaunchKernelGGL((sdpa_fused16_kernel<false, true>), grid, block, 0, ctx->stream, a); else hipLaunchKernelGGL((sdpa_fused16_kernel<false, false>), grid, block, 0, ctx->stream, a); }
    } else if (fast) {
        ProfScope ps(ctx, "sdpa_fused_kernel", flops, bytes);
        // the additive mask of a [B, 1, 1, T] attention mask (row stride 0, at least T values per batch item): one LDS copy per workgroup
        const bool mlds = mask && d->mask_row_stride == 0 && !(ctx->debug & 0x400000);
        a.debug = 0;
#ifdef RTEN_ABLATION
        a.debug = (ctx->debug >> 24) & 63;
#endif
        const bool full = d->t == TT && d->s % SQ == 0 && !(ctx->debug & 0x800000); // (bit 0x800000: the general form, A/B)
        const bool flush = d->flush_nan_to_zero != 0;
        const dim3 grid((unsigned)wgs), block(256);
#define RTEN_SDPA_GO(ABLV, MLV, FV, FLV) hipLaunchKernelGGL((sdpa_fused_kernel<ABLV, MLV, FV, FLV>), grid, block, 0, ctx->stream, a)
#define RTEN_SDPA_PICK(ABLV)                                                                                                             \
        do {                                                                                                                             \
            if (mlds) { if (full) { if (flush) RTEN_SDPA_GO(ABLV, true, true, true); else RTEN_SDPA_GO(ABLV, true, true, false); }       \
                        else { if (flush) RTEN_SDPA_GO(ABLV, true, false, true); else RTEN_SDPA_GO(ABLV, true, false, false); } }        \
            else { if (full) { if (flush) RTEN_SDPA_GO(ABLV, false, true, true); else RTEN_SDPA_GO(ABLV, false, true, false); }          \
                   else { if (flush) RTEN_SDPA_GO(ABLV, false, false, true); else RTEN_SDPA_GO(ABLV, false, false, false); } }           \
        } while (0)
#ifdef RTEN_ABLATION
        if (a.debug) RTEN_SDPA_PICK(true);
        else
#endif
        RTEN_SDPA_PICK(false);
#undef RTEN_SDPA_PICK
#undef RTEN_SDPA_GO
    } else {
        ProfScope ps(ctx, "sdpa_fused_general_kernel", flops, bytes);
        const int nch = (d->t + TT - 1) / TT; // 1 .. 4 chunks of 128 keys (3 runs as 4: the fourth chunk is zero-filled)
#define RTEN_SDPA_CASE(HDV)                                                        \
        if (nch <= 1) launch_general<HDV, 1>(ctx, a, wgs);                         \
        else if (nch == 2) launch_general<HDV, 2>(ctx, a, wgs);                    \
        else launch_general<HDV, 4>(ctx, a, wgs)
        if (d->d == 32) { RTEN_SDPA_CASE(32); }
        else if (d->d == 64) { RTEN_SDPA_CASE(64); }
        else { RTEN_SDPA_CASE(128); }
#undef RTEN_SDPA_CASE
    }
    RTEN_LAUNCH_CHECK(ctx, "sdpa_fused_kernel launch");
    return RTEN_HIP_OK;
}
