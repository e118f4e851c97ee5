/*
 * rten_oracle.c -- CPU restatement of the RTen hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X backend: a plain-C restatement of the
 * arithmetic the reference (robertknight/rten v0.25.0, CPU) performs on the path
 * rten-gemm / rten-vecmath behind src/ops/{matmul,conv,attention,norm,pooling,quantize}.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (librten_hip.so) never links or calls anything in here.
 *
 * Pinning: the reference is Rust and cannot be compiled in the build container (no
 * cargo/rustc), so the oracle is pinned against the literal golden vectors of the
 * reference's own tests (JSON files under tests/golden/, each citing file:line) and the pinned
 * RNG streams (rten-tensor/src/rng.rs:70-123).  See tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout).  Floating point is restated operation-for-operation: fused
 * multiply-adds where the reference uses `mul_add`, separate roundings elsewhere.
 * Compile with -ffp-contract=off so the compiler adds no contractions of its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define RTO_API __attribute__((visibility("default")))

static inline float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* ------------------------------------------------------------------------------------
 * RNG clones -- rten-tensor/src/rng.rs:6-66, rten-gemm/src/reduced_range_rng.rs:24-57
 * ---------------------------------------------------------------------------------- */
static inline uint64_t xorshift_next(uint64_t *state) {
    uint64_t t = *state;
    t ^= t << 13;
    t ^= t >> 7;
    t ^= t << 17;
    *state = t;
    return t;
}

/* rng.rs:26-32: top 40 bits scaled by 2^-40 */
RTO_API void rto_rng_f32(uint64_t *state, int64_t n, float *out) {
    const float scale = 1.0f / (float)(1ull << 40);
    for (int64_t i = 0; i < n; i++) {
        uint64_t v = xorshift_next(state) >> (64 - 40);
        out[i] = (float)v * scale;
    }
}
/* rng.rs:49-66: low bits of the 64-bit value */
RTO_API void rto_rng_u8(uint64_t *state, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)xorshift_next(state);
}
RTO_API void rto_rng_i8(uint64_t *state, int64_t n, int8_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (int8_t)xorshift_next(state);
}
RTO_API void rto_rng_i32(uint64_t *state, int64_t n, int32_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (int32_t)xorshift_next(state);
}
/* reduced_range_rng.rs:38-57 */
RTO_API void rto_rng_i8_reduced(uint64_t *state, int64_t n, int8_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (int8_t)((int16_t)(xorshift_next(state) % 128) - 64);
}
RTO_API void rto_rng_u8_reduced(uint64_t *state, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)(xorshift_next(state) % 128);
}

/* ------------------------------------------------------------------------------------
 * Output size / padding -- src/ops/pooling.rs:63-159
 * pad_mode: 0 = fixed pads[4] = top,left,bottom,right ; 1 = SAME (SAME_UPPER)
 * round_mode: 0 floor, 1 ceil.  Returns 0 on success, else an error code whose message is
 * the reference's OpError string (see rto_strerror).
 * ---------------------------------------------------------------------------------- */
enum {
    RTO_OK = 0,
    RTO_E_DILATION = 1,   /* "Dilations must be > 0" */
    RTO_E_KERNEL = 2,     /* "Kernel size must be > 0" */
    RTO_E_STRIDE = 3,     /* "Strides must be > 0" */
    RTO_E_TOO_SMALL = 4,  /* "Input too small for kernel size" */
};

RTO_API const char *rto_strerror(int code) {
    switch (code) {
    case RTO_OK: return "ok";
    case RTO_E_DILATION: return "Dilations must be > 0";
    case RTO_E_KERNEL: return "Kernel size must be > 0";
    case RTO_E_STRIDE: return "Strides must be > 0";
    case RTO_E_TOO_SMALL: return "Input too small for kernel size";
    default: return "unknown";
    }
}

static int axis_out_pad(int64_t in, int64_t k, int64_t stride, int same, int64_t ps, int64_t pe,
                        int64_t dil, int ceil_mode, int64_t *out, int64_t *pad_s, int64_t *pad_e) {
    if (dil <= 0) return RTO_E_DILATION;     /* pooling.rs:71 */
    if (k <= 0) return RTO_E_KERNEL;         /* pooling.rs:72 */
    if (stride <= 0) return RTO_E_STRIDE;    /* pooling.rs:73 */
    if (same) {                              /* pooling.rs:76-93 */
        int64_t o = (in + stride - 1) / stride;
        int64_t need = (o - 1) * stride + (k - 1) * dil + 1;
        int64_t total = need > in ? need - in : 0;
        *out = o;
        *pad_s = total / 2;
        *pad_e = (total + 1) / 2;
        return RTO_OK;
    }
    int64_t padded = in + ps + pe;           /* pooling.rs:94-122 */
    int64_t dk = k + (k - 1) * (dil - 1);
    if (padded < dk) return RTO_E_TOO_SMALL;
    int64_t win = padded - dil * (k - 1) - 1;
    int64_t o = ceil_mode ? (win + stride - 1) / stride + 1 : win / stride + 1;
    if (ceil_mode && (o - 1) * stride >= in + ps) o -= 1;
    *out = o;
    *pad_s = ps;
    *pad_e = pe;
    return RTO_OK;
}

RTO_API int rto_calc_output_size_and_padding(int64_t in_h, int64_t in_w, int64_t k_h, int64_t k_w,
                                             int64_t s_h, int64_t s_w, int same,
                                             const int64_t pads[4], int64_t d_h, int64_t d_w,
                                             int ceil_mode, int64_t out[2], int64_t out_pads[4]) {
    int64_t pt = same ? 0 : pads[0], pl = same ? 0 : pads[1];
    int64_t pb = same ? 0 : pads[2], pr = same ? 0 : pads[3];
    int rc = axis_out_pad(in_h, k_h, s_h, same, pt, pb, d_h, ceil_mode, &out[0], &out_pads[0], &out_pads[2]);
    if (rc) return rc;
    rc = axis_out_pad(in_w, k_w, s_w, same, pl, pr, d_w, ceil_mode, &out[1], &out_pads[1], &out_pads[3]);
    return rc;
}

/* ------------------------------------------------------------------------------------
 * f32 GEMM -- rten-gemm/src/lib.rs:794-1093 (gemm_impl), :1128-1259 (gemm_block),
 * kernels/simd_generic.rs:285-414 (micro-kernel incl. alpha/beta store cases).
 *
 * Numerics restated: depth is split into blocks of kc = min(256, K) (lib.rs:630-633,
 * 1024/size_of::<f32>()).  Per output element and depth block the micro-kernel runs a
 * k-ordered fused-multiply-add chain starting from 0.0 (simd_generic.rs:326-367) and then
 * stores with one of four (alpha,beta) forms (:378-414); the first block uses the caller's
 * beta, later blocks beta = 1 (lib.rs:1008-1013 `effective_beta`).  The bias is added after
 * the first depth block only (lib.rs:1221-1255).  Every output element is an independent
 * chain, so the result does not depend on MR/NR, thread count or ISA (all f32 kernels use
 * fused mul_add).  The M == 1 gemv fast path (lib.rs:876-891, simd_generic.rs:14-197) is restated
 * separately below (rto_gemm_f32 takes it like gemm_impl does; the convolution's GEMM never does).
 *
 * B is "virtual": element (k, n) comes from a callback so that dense, transposed and im2col
 * inputs share the code (the reference does the same via packing, lib.rs:958-1003).
 * ---------------------------------------------------------------------------------- */
#define RTO_KC_F32 256
#define RTO_NR 64
#define RTO_MR 4

typedef struct {
    /* dense */
    const float *b;
    int64_t rs, cs;
    /* im2col (rten-gemm/src/im2col.rs:56-88,145-208): row r -> (chan, ky, kx); col c -> (oy, ox) */
    const float *img;
    int64_t C, H, W, kh, kw, OH, OW, sy, sx, dy, dx, pt, pl;
    int is_im2col;
} bsrc_f32;

/* pack rows [k0,k1) x cols [n0,n0+nr) of virtual B into panel[(k-k0)*RTO_NR + j] (zero padded) */
static void pack_b_f32(const bsrc_f32 *s, int64_t k0, int64_t k1, int64_t n0, int64_t nr, float *panel) {
    if (!s->is_im2col) {
        for (int64_t k = k0; k < k1; k++) {
            float *row = panel + (k - k0) * RTO_NR;
            const float *src = s->b + k * s->rs + n0 * s->cs;
            if (s->cs == 1) {
                memcpy(row, src, (size_t)nr * sizeof(float));
            } else {
                for (int64_t j = 0; j < nr; j++) row[j] = src[j * s->cs];
            }
            for (int64_t j = nr; j < RTO_NR; j++) row[j] = 0.f;
        }
        return;
    }
    const int64_t khw = s->kh * s->kw;
    for (int64_t k = k0; k < k1; k++) {
        float *row = panel + (k - k0) * RTO_NR;
        int64_t c = k / khw, rem = k % khw, ky = rem / s->kw, kx = rem % s->kw;
        const float *chan = s->img + c * s->H * s->W;
        int64_t oy = n0 / s->OW, ox = n0 % s->OW;
        for (int64_t j = 0; j < nr; j++) {
            int64_t iy = oy * s->sy + ky * s->dy - s->pt;
            int64_t ix = ox * s->sx + kx * s->dx - s->pl;
            /* im2col.rs:188-203: out-of-image elements are written as 0 */
            row[j] = (iy >= 0 && iy < s->H && ix >= 0 && ix < s->W) ? chan[iy * s->W + ix] : 0.f;
            if (++ox == s->OW) { ox = 0; oy++; }
        }
        for (int64_t j = nr; j < RTO_NR; j++) row[j] = 0.f;
    }
}

/* bias_kind: 0 none, 1 per-row (BiasVector::Column, indexed by m), 2 per-column (BiasVector::Row, by n) */
static void gemm_f32_core(int64_t M, int64_t N, int64_t K, const float *A, int64_t a_rs, int64_t a_cs,
                          const bsrc_f32 *bs, float *C, int64_t ldc, float alpha, float beta,
                          const float *bias, int bias_kind) {
    if (M == 0 || N == 0) return;              /* lib.rs:835-839 */
    if (K == 0) {                              /* lib.rs:843-873 */
        for (int64_t m = 0; m < M; m++)
            for (int64_t n = 0; n < N; n++) {
                float v = (beta == 0.f) ? 0.f : C[m * ldc + n] * beta;
                if (bias_kind == 1) v = v + bias[m];
                else if (bias_kind == 2) v = v + bias[n];
                C[m * ldc + n] = v;
            }
        return;
    }
    const int64_t kc = K < RTO_KC_F32 ? K : RTO_KC_F32;
    float *panel = (float *)aligned_alloc(64, (size_t)kc * RTO_NR * sizeof(float));
    float *apack = (float *)aligned_alloc(64, (size_t)kc * RTO_MR * sizeof(float) + 64);
    for (int64_t n0 = 0; n0 < N; n0 += RTO_NR) {
        int64_t nr = N - n0 < RTO_NR ? N - n0 : RTO_NR;
        for (int64_t k0 = 0; k0 < K; k0 += kc) {
            int64_t k1 = k0 + kc < K ? k0 + kc : K;
            int64_t depth = k1 - k0;
            pack_b_f32(bs, k0, k1, n0, nr, panel);
            float eff_beta = (k0 == 0) ? beta : 1.f;
            for (int64_t m0 = 0; m0 < M; m0 += RTO_MR) {
                int64_t mr = M - m0 < RTO_MR ? M - m0 : RTO_MR;
                for (int64_t k = 0; k < depth; k++)
                    for (int64_t i = 0; i < RTO_MR; i++)
                        apack[k * RTO_MR + i] = i < mr ? A[(m0 + i) * a_rs + (k0 + k) * a_cs] : 0.f;
                float acc[RTO_MR][RTO_NR];
                for (int i = 0; i < RTO_MR; i++)
                    for (int j = 0; j < RTO_NR; j++) acc[i][j] = 0.f;
                for (int64_t k = 0; k < depth; k++) {
                    const float *brow = panel + k * RTO_NR;
                    for (int i = 0; i < RTO_MR; i++) {
                        float a = apack[k * RTO_MR + i];
                        for (int j = 0; j < RTO_NR; j++) acc[i][j] = fma32(a, brow[j], acc[i][j]);
                    }
                }
                for (int64_t i = 0; i < mr; i++) {
                    float *crow = C + (m0 + i) * ldc + n0;
                    for (int64_t j = 0; j < nr; j++) {
                        float t = acc[i][j], v;
                        /* simd_generic.rs:378-414 */
                        if (eff_beta == 0.f && alpha == 1.f) v = t;
                        else if (eff_beta == 1.f && alpha == 1.f) v = crow[j] + t;
                        else if (eff_beta == 0.f) v = t * alpha;
                        else v = fma32(t, alpha, crow[j] * eff_beta);
                        if (k0 == 0) {        /* lib.rs:1221-1255 */
                            if (bias_kind == 1) v = v + bias[m0 + i];
                            else if (bias_kind == 2) v = v + bias[n0 + j];
                        }
                        crow[j] = v;
                    }
                }
            }
        }
    }
    free(panel);
    free(apack);
}

/* ------------------------------------------------------------------------------------
 * M == 1: the reference's vector-matrix path -- rten-gemm/src/lib.rs:668-747 (gemv: column blocks, depth blocks, bias),
 * :876-891 (taken when A has one row and B is NOT prepacked; ModelOptions::prepack_weights defaults to false, src/model.rs:694),
 * kernels/simd_generic.rs:14-197 (the three kernels).  AVX-512 instantiation (x86_64.rs:452-470: 16 lanes, NR_REGS = 2), like
 * the reductions elsewhere in this file.  Unlike the blocked GEMM this is NOT one k-ordered chain per output:
 *  - B with unit column stride (row-major [K][N]): depth blocks of 8 (lib.rs:698); per block a fused-multiply-add chain from 0
 *    per column, scaled by alpha, folded into the output with the block's beta (caller's beta, then 1) by store / add / fma;
 *    columns are processed 32 at a time and the columns left over at the end of a column block run a scalar loop with SEPARATE
 *    multiply and add and `beta * out + acc * alpha` (simd_generic.rs:90-103);
 *  - B with unit row stride (transposed operands, e.g. Gemm transB): depth blocks of 512; 8 columns at a time, each with 16 lane
 *    accumulators over 16-element depth tiles, summed by _mm512_reduce_add_ps (lane l + lane l + 8, + 4, + 2, + 1), then a scalar
 *    fma tail over depth % 16, then `alpha * acc (+ beta * out)`; the < 8 columns left over in a column block run the scalar fma
 *    chain of the fallback kernel;
 *  - neither stride 1: depth blocks of 8, scalar fma chain per column, `acc *= alpha`, `acc + beta * out`.
 * The bias is added after the last depth block.  Column blocks have max(128, ceil(N / threads)) columns (lib.rs:697), so WHICH
 * columns are "left over" depends on the reference's thread count: `rto_set_gemv_threads(t)`; the default 0 stands for "at least
 * N / 128 threads" (every block 128 columns wide), the case on the many-core hosts this project measures on.
 * ---------------------------------------------------------------------------------- */
static int64_t g_gemv_threads = 0;
static int g_gemv_enabled = 1;
RTO_API void rto_set_gemv_threads(int64_t t) { g_gemv_threads = t; }
RTO_API void rto_set_gemv_enabled(int on) { g_gemv_enabled = on; } /* 0: B is "prepacked": M == 1 takes the blocked path too */

static void gemv_fallback(int64_t ncols, int64_t depth, const float *a, const float *b, int64_t rs, int64_t cs, float *out,
                          float alpha, float beta) { /* simd_generic.rs:179-197 */
    for (int64_t c = 0; c < ncols; c++) {
        float acc = 0.f;
        for (int64_t k = 0; k < depth; k++) acc = fma32(a[k], b[k * rs + c * cs], acc);
        acc = acc * alpha;
        if (beta == 0.f) out[c] = acc;
        else {
            volatile float t = beta * out[c]; /* separately rounded product */
            out[c] = acc + t;
        }
    }
}

static void gemv_transposed(int64_t ncols, int64_t depth, const float *a, const float *b, int64_t cs, float *out, float alpha,
                            float beta) { /* simd_generic.rs:107-174; b has unit row stride */
    const int64_t full = ncols / 8 * 8, dt = depth / 16 * 16;
    for (int64_t c = 0; c < full; c++) {
        const float *col = b + c * cs;
        float lanes[16];
        for (int l = 0; l < 16; l++) lanes[l] = 0.f;
        for (int64_t d = 0; d < dt; d += 16)
            for (int l = 0; l < 16; l++) lanes[l] = fma32(a[d + l], col[d + l], lanes[l]);
        for (int w = 8; w >= 1; w >>= 1)
            for (int l = 0; l < w; l++) lanes[l] = lanes[l] + lanes[l + w];
        float acc = lanes[0];
        for (int64_t k = dt; k < depth; k++) acc = fma32(a[k], col[k], acc);
        volatile float pa = alpha * acc;
        if (beta == 0.f) out[c] = pa;
        else {
            volatile float pb = beta * out[c];
            out[c] = pa + pb;
        }
    }
    if (full < ncols) gemv_fallback(ncols - full, depth, a, b + full * cs, 1, cs, out + full, alpha, beta);
}

static void gemv_rowmajor(int64_t ncols, int64_t depth, const float *a, const float *b, int64_t rs, float *out, float alpha,
                          float beta) { /* simd_generic.rs:28-104; b has unit column stride */
    const int64_t full = ncols / 32 * 32;
    for (int64_t c = 0; c < full; c++) {
        float acc = 0.f;
        for (int64_t k = 0; k < depth; k++) acc = fma32(a[k], b[k * rs + c], acc);
        if (alpha != 1.f) acc = acc * alpha;
        if (beta == 0.f) out[c] = acc;
        else if (beta == 1.f) out[c] = out[c] + acc;
        else out[c] = fma32(out[c], beta, acc);
    }
    for (int64_t c = full; c < ncols; c++) {
        float acc = 0.f;
        for (int64_t k = 0; k < depth; k++) {
            volatile float pr = a[k] * b[k * rs + c]; /* `acc += ax * b`: no fused multiply-add in Rust */
            acc = acc + pr;
        }
        const float tmp = beta == 0.f ? 0.f : out[c];
        volatile float p0 = beta * tmp, p1 = acc * alpha;
        out[c] = p0 + p1;
    }
}

static void gemv_f32(int64_t N, int64_t K, const float *A, int64_t a_cs, const float *B, int64_t b_rs, int64_t b_cs, float *out,
                     float alpha, float beta, const float *bias, int bias_kind) {
    float *a = (float *)malloc((size_t)(K > 0 ? K : 1) * sizeof(float)); /* a.to_contiguous() */
    for (int64_t k = 0; k < K; k++) a[k] = A[k * a_cs];
    int64_t cb = 128;
    if (g_gemv_threads > 0) {
        cb = (N + g_gemv_threads - 1) / g_gemv_threads;
        if (cb < 128) cb = 128;
    }
    const int64_t kb = b_rs == 1 ? 512 : 8;
    for (int64_t c0 = 0; c0 < N; c0 += cb) {
        const int64_t nc = N - c0 < cb ? N - c0 : cb;
        float eff_beta = beta;
        for (int64_t k0 = 0; k0 < K; k0 += kb) {
            const int64_t depth = K - k0 < kb ? K - k0 : kb;
            const float *bb = B + k0 * b_rs + c0 * b_cs;
            if (b_rs == 1) gemv_transposed(nc, depth, a + k0, bb, b_cs, out + c0, alpha, eff_beta);
            else if (b_cs != 1) gemv_fallback(nc, depth, a + k0, bb, b_rs, b_cs, out + c0, alpha, eff_beta);
            else gemv_rowmajor(nc, depth, a + k0, bb, b_rs, out + c0, alpha, eff_beta);
            eff_beta = 1.f;
        }
        for (int64_t c = 0; c < nc; c++) {
            if (bias_kind == 1) out[c0 + c] = out[c0 + c] + bias[0];       /* BiasVector::Column: one row */
            else if (bias_kind == 2) out[c0 + c] = out[c0 + c] + bias[c0 + c];
        }
    }
    free(a);
}

/* the blocked order whatever M is: the products INSIDE attention, ConvTranspose and MatMulNBits (rows > 1).  (With one query row /
 * one output row the reference's matmul would take its gemv kernels there as well; neither this oracle nor the backend follow it
 * into those operators -- DESIGN.md section 9.) */
static void gemm_f32_blocked(int64_t M, int64_t N, int64_t K, const float *A, int64_t a_rs, int64_t a_cs, const float *B, int64_t b_rs,
                             int64_t b_cs, float *C, int64_t ldc, float alpha, float beta, const float *bias, int bias_kind) {
    bsrc_f32 bs;
    memset(&bs, 0, sizeof bs);
    bs.b = B; bs.rs = b_rs; bs.cs = b_cs;
    gemm_f32_core(M, N, K, A, a_rs, a_cs, &bs, C, ldc, alpha, beta, bias, bias_kind);
}

RTO_API void rto_gemm_f32(int64_t M, int64_t N, int64_t K, const float *A, int64_t a_rs, int64_t a_cs,
                          const float *B, int64_t b_rs, int64_t b_cs, float *C, int64_t ldc,
                          float alpha, float beta, const float *bias, int bias_kind) {
    if (M == 1 && N > 0 && K > 0 && g_gemv_enabled) { /* lib.rs:876-891 (after the K == 0 case, :843-873) */
        gemv_f32(N, K, A, a_cs, B, b_rs, b_cs, C, alpha, beta, bias, bias_kind);
        return;
    }
    bsrc_f32 bs;
    memset(&bs, 0, sizeof bs);
    bs.b = B; bs.rs = b_rs; bs.cs = b_cs;
    gemm_f32_core(M, N, K, A, a_rs, a_cs, &bs, C, ldc, alpha, beta, bias, bias_kind);
}

/* batched: rten-gemm/src/lib.rs:329-372.  Strides in elements; a stride of 0 broadcasts. */
RTO_API void rto_gemm_f32_batched(int64_t batch, int64_t M, int64_t N, int64_t K, const float *A,
                                  int64_t a_rs, int64_t a_cs, int64_t a_bs, const float *B, int64_t b_rs,
                                  int64_t b_cs, int64_t b_bs, float *C, int64_t ldc, int64_t c_bs,
                                  float alpha, const float *bias, int bias_kind) {
#pragma omp parallel for schedule(dynamic)
    for (int64_t i = 0; i < batch; i++)
        rto_gemm_f32(M, N, K, A + i * a_bs, a_rs, a_cs, B + i * b_bs, b_rs, b_cs, C + i * c_bs, ldc,
                     alpha, 0.f, bias, bias_kind);
}

/* ------------------------------------------------------------------------------------
 * MatMulNBits (4-bit block-quantised RHS) -- src/ops/matmul/contrib.rs:21-106, rten-gemm/src/block_quant.rs.
 * quant is [N][K/bs][bs/2] bytes, element 2j of a block in the low nibble of byte j, element 2j+1 in the high
 * nibble; scales is [N][K/bs]; the zero point is fixed at 8 (block_quant.rs:802-805).  Dequantised value
 * = (float)(q - 8) * scale, one rounded multiply (packing.rs:300-312, block_quant.rs:243-262).
 *  - rows > 1 (contrib.rs:86-100): the dequantised matrix goes through the ordinary f32 GEMM
 *    (packing.rs:229-318 feeds the same micro-kernel), depth blocks kc = max(min(256, K), bs) (lib.rs:630-633,
 *    894-905).  The oracle GEMM blocks at min(256, K): identical for bs <= 256, the range ONNX Runtime emits.
 *  - rows == 1, ComputeMode::Float (block_quant.rs:166-389, AVX-512 instantiation): 128-element vblocks, eight
 *    16-lane groups per vblock, group i accumulating into acc[i % 4] with one FMA per element, so lane slot
 *    s = k % 64 owns k = s, s + 64, ...; then (acc0 + acc1) + (acc2 + acc3) lane-wise, a horizontal sum
 *    (_mm512_reduce_add_ps: halves folded 16 -> 8 -> 4 -> 2 -> 1, lane l + lane l + width/2), and a scalar tail over
 *    the K % 128 remainder: tail += a0 * lo + a1 * hi per byte with separately rounded products (:351-377).
 *    ComputeMode::Int8 (accuracy_level 4) is an opt-in approximation the reference itself may decline
 *    (contrib.rs:102-108); this backend always computes at Float accuracy.
 */
RTO_API void rto_dequantize_4bit(int64_t N, int64_t K, int64_t bs, const uint8_t *quant, const float *scales, float *b /* [K][N] */) {
    const int64_t kb = K / bs;
    for (int64_t n = 0; n < N; n++)
        for (int64_t k = 0; k < K; k++) {
            const uint8_t byte = quant[(n * kb + k / bs) * (bs / 2) + (k % bs) / 2];
            const int q = (k & 1) ? (byte >> 4) : (byte & 0x0F);
            b[k * N + n] = (float)(q - 8) * scales[n * kb + k / bs];
        }
}

static float vec_dot_4bit(int64_t K, int64_t bs, const float *a, const uint8_t *col, const float *col_scales) {
    float acc[64];
    for (int s = 0; s < 64; s++) acc[s] = 0.f;
    const int64_t kv = K - K % 128;
    for (int64_t k = 0; k < kv; k++) {
        const uint8_t byte = col[k / 2];
        const int q = (k & 1) ? (byte >> 4) : (byte & 0x0F);
        const float w = (float)(q - 8) * col_scales[k / bs];
        acc[k % 64] = fmaf(a[k], w, acc[k % 64]);
    }
    float v[16];
    for (int l = 0; l < 16; l++) v[l] = (acc[l] + acc[16 + l]) + (acc[32 + l] + acc[48 + l]);
    for (int w = 8; w >= 1; w >>= 1)
        for (int l = 0; l < w; l++) v[l] = v[l] + v[l + w];
    float out = v[0];
    if (kv < K) {
        float tail = 0.f;
        for (int64_t k = kv; k < K; k += 2) {
            const uint8_t byte = col[k / 2];
            const float s = col_scales[k / bs];
            const float lo = (float)((int)(byte & 0x0F) - 8) * s, hi = (float)((int)(byte >> 4) - 8) * s;
            volatile float p0 = a[k] * lo, p1 = a[k + 1] * hi; /* separately rounded products, no contraction */
            tail = tail + (p0 + p1);
        }
        out = out + tail;
    }
    return out;
}

/* lhs is [batch][rows][K]; out [batch][rows][N].  Returns 0 or an RTO_E_* code (contrib.rs:29-61). */
RTO_API int rto_matmul_nbits_f32(int64_t batch, int64_t rows, int64_t K, int64_t N, int64_t bs, const float *lhs,
                                 const uint8_t *quant, const float *scales, float *out) {
    if (bs < 16 || (bs & (bs - 1)) || K % bs) return 1;
    const int64_t kb = K / bs;
    if (K == 0) { memset(out, 0, (size_t)(batch * rows * N) * sizeof(float)); return 0; }
    if (rows == 1) {
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; n++)
            for (int64_t b = 0; b < batch; b++)
                out[b * N + n] = vec_dot_4bit(K, bs, lhs + b * K, quant + n * kb * (bs / 2), scales + n * kb);
        return 0;
    }
    float *bm = (float *)malloc((size_t)K * N * sizeof(float));
    rto_dequantize_4bit(N, K, bs, quant, scales, bm);
    gemm_f32_blocked(batch * rows, N, K, lhs, K, 1, bm, N, 1, out, N, 1.f, 0.f, NULL, 0);
    free(bm);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * f32 Conv -- src/ops/conv.rs:124-365 (conv_impl), :33-87 (pointwise), conv/im2col.rs:11-128.
 * NCHW input, OIHW kernel (I = C/groups), NCHW output.  Per image and group:
 * out[O_g, OH*OW] = W_g[O_g, C_g*kh*kw] . im2col(x)[C_g*kh*kw, OH*OW] with bias per out channel
 * (BiasVector::Column).  The pointwise fast path (conv.rs:250-267) is the same GEMM with the
 * image viewed as a [C, H*W] matrix, so a single restatement covers both.  Depthwise convolutions
 * (groups == C == O) take the reference's own path, rto_depthwise_conv2d_f32 below, as conv.rs:269-284
 * dispatches them: its arithmetic differs from the GEMM's (separate multiply and add, no FMA; the
 * accumulator starts at the bias; padded taps are skipped).
 * Fused extras (not in the reference Conv op; restated as the op sequence the reference graph
 * runs): residual Add (binary_elementwise.rs:476-495) then Relu (unary_elementwise.rs:611-613).
 * ---------------------------------------------------------------------------------- */
/* Depthwise convolution -- src/ops/conv/depthwise.rs:95-146 (GenericDepthwiseConvKernel<f32>::compute_row) and
 * :215-262 (channel loop).  One input channel per output channel.  Per output element:
 *     acc = bias[c] (or 0);  for k_y in 0..kh: skip rows outside the image;
 *                            for k_x in 0..kw: if the tap's input column is inside the image: acc += x * w
 * with `x * w` rounded to f32 before the add (Rust does not contract `a += b * c` into an FMA), taps visited in
 * (k_y, k_x) order, padding taps not visited at all.  (The reference walks whole output rows per tap; the per-element
 * sequence of operations is the one written here.) */
static void rto_depthwise_conv2d_f32(int64_t N, int64_t C, int64_t H, int64_t W, int64_t kh, int64_t kw, const int64_t pads[4],
                                     const int64_t strides[2], const int64_t dil[2], const float *X, const float *Wt, const float *bias,
                                     const float *residual, int relu, float *Y, int64_t OH, int64_t OW) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t n = 0; n < N; n++)
        for (int64_t c = 0; c < C; c++) {
            const float *xc = X + (n * C + c) * H * W;
            const float *wc = Wt + c * kh * kw;
            float *yc = Y + (n * C + c) * OH * OW;
            const float *rc = residual ? residual + (n * C + c) * OH * OW : NULL;
            for (int64_t oy = 0; oy < OH; oy++)
                for (int64_t ox = 0; ox < OW; ox++) {
                    float acc = bias ? bias[c] : 0.0f;
                    for (int64_t ky = 0; ky < kh; ky++) {
                        const int64_t iy = oy * strides[0] + ky * dil[0] - pads[0];
                        if (iy < 0 || iy >= H) continue;
                        for (int64_t kx = 0; kx < kw; kx++) {
                            const int64_t ix = ox * strides[1] + kx * dil[1] - pads[1];
                            if (ix < 0 || ix >= W) continue;
                            const float prod = xc[iy * W + ix] * wc[ky * kw + kx];
                            acc = acc + prod;
                        }
                    }
                    if (rc) acc = acc + rc[oy * OW + ox];
                    if (relu) acc = fmaxf(acc, 0.f); /* f32::max: NaN -> 0 */
                    yc[oy * OW + ox] = acc;
                }
        }
}

RTO_API int rto_conv2d_f32(int64_t N, int64_t C, int64_t H, int64_t W, int64_t O, int64_t kh, int64_t kw,
                           const int64_t pads[4], const int64_t strides[2], const int64_t dil[2],
                           int64_t groups, const float *X, const float *Wt, const float *bias,
                           const float *residual, int relu, float *Y, int64_t OH, int64_t OW) {
    const int pointwise = kh == 1 && kw == 1 && groups == 1 && strides[0] == 1 && strides[1] == 1 && dil[0] == 1 && dil[1] == 1 &&
                          pads[0] == 0 && pads[1] == 0 && pads[2] == 0 && pads[3] == 0;
    if (!pointwise && C == O && groups == C) { /* conv.rs:269-284 */
        rto_depthwise_conv2d_f32(N, C, H, W, kh, kw, pads, strides, dil, X, Wt, bias, residual, relu, Y, OH, OW);
        return 0;
    }
    const int64_t Cg = C / groups, Og = O / groups, Kg = Cg * kh * kw, P = OH * OW;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int64_t n = 0; n < N; n++) {
        for (int64_t g = 0; g < groups; g++) {
            bsrc_f32 bs;
            memset(&bs, 0, sizeof bs);
            bs.is_im2col = 1;
            bs.img = X + (n * C + g * Cg) * H * W;
            bs.C = Cg; bs.H = H; bs.W = W; bs.kh = kh; bs.kw = kw; bs.OH = OH; bs.OW = OW;
            bs.sy = strides[0]; bs.sx = strides[1]; bs.dy = dil[0]; bs.dx = dil[1];
            bs.pt = pads[0]; bs.pl = pads[1];
            float *out = Y + (n * O + g * Og) * P;
            gemm_f32_core(Og, P, Kg, Wt + g * Og * Kg, Kg, 1, &bs, out, P, 1.f, 0.f,
                          bias ? bias + g * Og : NULL, bias ? 1 : 0);
            if (residual || relu) {
                const float *res = residual ? residual + (n * O + g * Og) * P : NULL;
                for (int64_t i = 0; i < Og * P; i++) {
                    float v = out[i];
                    if (res) v = v + res[i];
                    if (relu) v = fmaxf(v, 0.f); /* f32::max: NaN -> 0 */
                    out[i] = v;
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Integer GEMM -- rten-gemm/src/kernels/generic.rs:274-366 and tests.rs:90-133:
 *   C[m,n] (+)= sum_k (A[m,k] - a_zp[m]) * (B[k,n] - b_zp[n])   in wrapping i32
 * The reference's kernels take u8 LHS x i8 RHS; the operator front-ends shift-cast other
 * signedness combos (src/shift_cast.rs:39-50), which leaves the mathematical value of every
 * (x - zp) unchanged, so the oracle works on widened integers directly.
 * a_signed / b_signed select how the raw bytes are interpreted.
 * ---------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------
 * ConvTranspose -- src/ops/conv_transpose.rs:226-412 (conv_transpose), :80-142 (col2im).
 * X [N, C, H, W], kernel [C, O/g, kh, kw] ("COHW"), Y [N, O, OH, OW].  Per group and image:
 *   columns[O_g*kh*kw, H*W] = kernel_mat^T [O_g*kh*kw, C_g] . input_mat [C_g, H*W]      (GemmExecutor::gemm_uninit: alpha 1, beta 0)
 *   out channel o: every element starts at bias[o] (or 0); then for k_y, k_x in order every column image
 *   columns[o, k_y, k_x] is accumulated at out[y*stride + k_y*dil - pad_top, x*stride + k_x*dil - pad_left].
 * An output element receives at most one addend per (k_y, k_x), so its value is bias, then the addends in (k_y, k_x) order.
 * ---------------------------------------------------------------------------------- */
RTO_API int rto_conv_transpose2d_f32(int64_t N, int64_t C, int64_t H, int64_t W, int64_t Og, int64_t kh, int64_t kw, const int64_t pads[4],
                                     const int64_t strides[2], const int64_t dil[2], int64_t groups, const float *X, const float *Wt,
                                     const float *bias, float *Y, int64_t OH, int64_t OW) {
    const int64_t Cg = C / groups, M = Og * kh * kw, P = H * W, O = Og * groups;
    float *cols = (float *)aligned_alloc(64, (size_t)((M * P * sizeof(float) + 63) / 64 * 64 + 64));
    if (!cols) return -1;
    for (int64_t g = 0; g < groups; g++)
        for (int64_t n = 0; n < N; n++) {
            /* A[m][k] = kernel[(g*Cg + k)][m]: row stride 1, column stride M (the transposed kernel matrix) */
            /* (gemm_impl on unpacked operands: a kernel matrix of ONE row -- O_g = kh = kw = 1 -- takes the vector-matrix path, lib.rs:876-891) */
            rto_gemm_f32(M, P, Cg, Wt + g * Cg * M, 1, M, X + (n * C + g * Cg) * P, P, 1, cols, P, 1.0f, 0.0f, NULL, 0);
            for (int64_t o = 0; o < Og; o++) {
                float *out = Y + (n * O + g * Og + o) * OH * OW;
                const float b = bias ? bias[g * Og + o] : 0.0f;
                for (int64_t i = 0; i < OH * OW; i++) out[i] = b;
                for (int64_t ky = 0; ky < kh; ky++)
                    for (int64_t kx = 0; kx < kw; kx++) {
                        const float *img = cols + ((o * kh + ky) * kw + kx) * P;
                        for (int64_t y = 0; y < H; y++) {
                            const int64_t oy = y * strides[0] + ky * dil[0] - pads[0];
                            if (oy < 0 || oy >= OH) continue;
                            for (int64_t x = 0; x < W; x++) {
                                const int64_t ox = x * strides[1] + kx * dil[1] - pads[1];
                                if (ox < 0 || ox >= OW) continue;
                                out[oy * OW + ox] = out[oy * OW + ox] + img[y * W + x];
                            }
                        }
                    }
            }
        }
    free(cols);
    return 0;
}

static inline int32_t ld8(const void *p, int64_t i, int is_signed) {
    return is_signed ? (int32_t)((const int8_t *)p)[i] : (int32_t)((const uint8_t *)p)[i];
}

RTO_API void rto_gemm_int8(int64_t M, int64_t N, int64_t K, const void *A, int a_signed, int64_t a_rs,
                           int64_t a_cs, const void *B, int b_signed, int64_t b_rs, int64_t b_cs,
                           int32_t *C, int64_t ldc, const void *a_zp, int64_t a_zp_len, const void *b_zp,
                           int64_t b_zp_len, int beta) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; m++) {
        int32_t az = a_zp ? ld8(a_zp, a_zp_len == 1 ? 0 : m, a_signed) : 0;
        for (int64_t n = 0; n < N; n++) {
            int32_t bz = b_zp ? ld8(b_zp, b_zp_len == 1 ? 0 : n, b_signed) : 0;
            uint32_t acc = 0;
            for (int64_t k = 0; k < K; k++) {
                int32_t a = ld8(A, m * a_rs + k * a_cs, a_signed) - az;
                int32_t b = ld8(B, k * b_rs + n * b_cs, b_signed) - bz;
                acc += (uint32_t)(a * b);
            }
            if (beta) acc += (uint32_t)C[m * ldc + n];
            C[m * ldc + n] = (int32_t)acc;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Integer Conv -- src/ops/conv.rs:421-476 (conv_integer) -> conv_impl::<i8,u8,i32>.
 * Y_i32[n,o,p] = sum_k (W[o,k] - w_zp[o]) * (col[k,p] - x_zp)
 * pad_mode selects what a padded (out-of-image) im2col element is worth, SURVEY App. C.1:
 *   0 ZERO_POINT : contributes 0 (ONNX / depthwise semantics, conv/depthwise.rs:160-182)
 *   1 RAW0_I8    : x86/generic reference: raw 0 written AFTER the u8->i8 shift cast
 *                  (rten-gemm/src/im2col.rs:194-198,351-357) == original-domain value of
 *                  (x_signed ? 0 : 128)
 *   2 RAW0_U8    : Arm/wasm reference (im2col.rs:349-353): raw 0 in the u8 domain == original-
 *                  domain value (x_signed ? -128 : 0)
 * ---------------------------------------------------------------------------------- */
RTO_API int rto_conv2d_int8(int64_t N, int64_t C, int64_t H, int64_t W, int64_t O, int64_t kh, int64_t kw,
                            const int64_t pads[4], const int64_t strides[2], const int64_t dil[2],
                            int64_t groups, const void *X, int x_signed, const void *Wt, int w_signed,
                            int32_t x_zp, const void *w_zp, int64_t w_zp_len, int pad_mode, int32_t *Y,
                            int64_t OH, int64_t OW) {
    const int64_t Cg = C / groups, Og = O / groups, P = OH * OW;
    {   /* depthwise geometries run conv/depthwise.rs:148-190 (conv.rs:269-284), which skips padded taps: they contribute 0 on
         * every platform, whatever the im2col quirk selected by pad_mode would do on the GEMM path */
        const int pointwise = kh == 1 && kw == 1 && groups == 1 && strides[0] == 1 && strides[1] == 1 && dil[0] == 1 && dil[1] == 1 &&
                              pads[0] == 0 && pads[1] == 0 && pads[2] == 0 && pads[3] == 0;
        if (!pointwise && C == O && groups == C) pad_mode = 0;
    }
    int32_t pad_val;
    if (pad_mode == 0) pad_val = x_zp;
    else if (pad_mode == 1) pad_val = x_signed ? 0 : 128;
    else pad_val = x_signed ? -128 : 0;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int64_t n = 0; n < N; n++) {
        for (int64_t o = 0; o < O; o++) {
            int64_t g = o / Og;
            int32_t wz = w_zp ? ld8(w_zp, w_zp_len == 1 ? 0 : o, w_signed) : 0;
            for (int64_t oy = 0; oy < OH; oy++)
                for (int64_t ox = 0; ox < OW; ox++) {
                    uint32_t acc = 0;
                    for (int64_t c = 0; c < Cg; c++)
                        for (int64_t ky = 0; ky < kh; ky++)
                            for (int64_t kx = 0; kx < kw; kx++) {
                                int64_t iy = oy * strides[0] + ky * dil[0] - pads[0];
                                int64_t ix = ox * strides[1] + kx * dil[1] - pads[1];
                                int32_t xv = (iy >= 0 && iy < H && ix >= 0 && ix < W)
                                                 ? ld8(X, ((n * C + g * Cg + c) * H + iy) * W + ix, x_signed)
                                                 : pad_val;
                                int32_t wv = ld8(Wt, ((o * Cg + c) * kh + ky) * kw + kx, w_signed);
                                acc += (uint32_t)((wv - wz) * (xv - x_zp));
                            }
                    Y[(n * O + o) * P + oy * OW + ox] = (int32_t)acc;
                }
        }
    }
    return 0;
}

/* cast_scale -- src/ops/matmul.rs:734-773: (i32 as f32) * scale, scalar or per last-axis column */
RTO_API void rto_cast_scale(int64_t n, const int32_t *x, const float *scale, int64_t scale_len, float *y) {
    for (int64_t i = 0; i < n; i++) y[i] = (float)x[i] * scale[scale_len == 1 ? 0 : i % scale_len];
}

/* ------------------------------------------------------------------------------------
 * DynamicQuantizeLinear -- src/ops/quantize.rs:352-436, :171-176,
 * rten-vecmath/src/quantize.rs:39-79, min_max.rs:20-44.
 * ---------------------------------------------------------------------------------- */
static inline float f32_min(float a, float b) { return fminf(a, b); } /* Rust f32::min: ignores NaN */
static inline float f32_max(float a, float b) { return fmaxf(a, b); }

static inline uint8_t sat_u8_from_f32(float v) { /* Rust `as u8`: saturating, NaN -> 0 */
    if (!(v == v)) return 0;
    if (v <= 0.f) return 0;
    if (v >= 255.f) return 255;
    return (uint8_t)v;
}

RTO_API void rto_dynamic_quantize_linear(int64_t n, const float *x, uint8_t *y, float *scale_out,
                                         uint8_t *zp_out) {
    if (n == 0) { *scale_out = 1.f; *zp_out = 0; return; } /* quantize.rs:385-392 */
    float x_min = INFINITY, x_max = -INFINITY;
    for (int64_t i = 0; i < n; i++) {
        /* SIMD min/max (min_max.rs:24-31) drop NaNs the way x86 min/max(x, acc) do: if x is NaN the
           accumulator is returned.  fminf/fmaxf have the same NaN-ignoring result. */
        x_min = f32_min(x[i], x_min);
        x_max = f32_max(x[i], x_max);
    }
    float x_min_adj = f32_min(x_min, 0.f);
    float x_max_adj = f32_max(x_max, 0.f);
    float range = x_max_adj - x_min_adj;
    float scale = range / 255.f;
    float min_scaled = x_min_adj / scale;
    float init_zp = 0.f - min_scaled;
    /* f32::clamp(0,255): NaN stays NaN */
    float clipped = init_zp < 0.f ? 0.f : (init_zp > 255.f ? 255.f : init_zp);
    float rounded = nearbyintf(clipped); /* round_ties_even under default rounding mode */
    uint8_t zp = sat_u8_from_f32(rounded < 0.f ? 0.f : (rounded > 255.f ? 255.f : rounded));
    if (!(rounded == rounded)) zp = 0;
    *scale_out = scale;
    *zp_out = zp;
    float inv_scale = 1.f / scale; /* quantize.rs:210 */
    for (int64_t i = 0; i < n; i++) {
        /* vecmath/quantize.rs:57-62 (SIMD body) and :70-74 (tail) agree for all finite products:
           round-to-nearest-even to i32, add zero point as integer, saturate to [0,255].
           Non-finite products: the SIMD path converts NaN/out-of-range to i32::MIN -> 0; the scalar
           path's `as i32` gives 0 for NaN (then + zp).  Both give 0 when zp == 0, which is the only
           case reachable from DQL (scale == 0 => zp == 0). */
        float p = x[i] * inv_scale;
        int32_t q;
        if (!(p == p)) q = INT32_MIN;
        else if (p >= 2147483648.f || p < -2147483648.f) q = INT32_MIN;
        else q = (int32_t)nearbyintf(p);
        int64_t t = (int64_t)q + (int64_t)zp;
        y[i] = (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t));
    }
}

/* ------------------------------------------------------------------------------------
 * exp / erf / gelu -- rten-vecmath/src/exp.rs:59-132 (Exp), :140-190 (ReducedRangeExp),
 * erf.rs:21-76.  All elementwise, restated op-for-op (mul_add -> fmaf) => bit-exact.
 * ---------------------------------------------------------------------------------- */
static const float INV_LOG2 = 1.44269504088896340736f; /* std::f32::consts::LOG2_E */
static const float ROUNDING_MAGIC = 12582912.f;
static const float LOG2_HI = -6.93145752e-1f;
static const float LOG2_LO = -1.42860677e-6f;
static const float EXP_P0 = 1.0f, EXP_P1 = 1.0f, EXP_P2 = 4.99999851e-1f, EXP_P3 = 1.66664720e-1f,
                   EXP_P4 = 4.16695364e-2f, EXP_P5 = 8.37312452e-3f, EXP_P6 = 1.37805939e-3f;

static inline float bits_f32(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t f32_bits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

static inline float exp_poly_reduce(float x, float *jout) {
    float j = fma32(x, INV_LOG2, ROUNDING_MAGIC);
    j = j - ROUNDING_MAGIC;
    float r = fma32(j, LOG2_HI, x);
    r = fma32(j, LOG2_LO, r);
    float t = EXP_P6;
    t = fma32(t, r, EXP_P5);
    t = fma32(t, r, EXP_P4);
    t = fma32(t, r, EXP_P3);
    t = fma32(t, r, EXP_P2);
    t = fma32(t, r, EXP_P1);
    r = fma32(t, r, EXP_P0);
    *jout = j;
    return r;
}

RTO_API float rto_exp_f32(float x) { /* exp.rs:59-132 */
    float j;
    float r = exp_poly_reduce(x, &j);
    int32_t k = (int32_t)j; /* to_int_trunc; j is integral */
    int32_t ia = (k > 0) ? 0 : (int32_t)0x83000000u;
    int32_t is = (int32_t)((uint32_t)ia + 0x7f000000u);
    int32_t it = (int32_t)(((uint32_t)k << 23) - (uint32_t)ia);
    r = r * bits_f32(is);
    r = r * bits_f32(it);
    if (x >= 104.0f) r = INFINITY;
    if (x <= -104.0f) r = 0.f;
    return r;
}

static const float EXP_LOWER_CUTOFF = -126.5f * 0.693147180559945309417f + 0.01f; /* exp.rs:131 */

RTO_API float rto_exp_reduced_f32(float x) { /* exp.rs:140-190; requires x <= 0 */
    float j;
    float r = exp_poly_reduce(x, &j);
    int32_t k = (int32_t)j;
    int32_t kp = (int32_t)((uint32_t)(k + 127) << 23);
    r = r * bits_f32(kp);
    if (x < EXP_LOWER_CUTOFF) r = 0.f;
    return r;
}

RTO_API float rto_erf_f32(float x0) { /* erf.rs:21-59 */
    int neg = x0 < 0.f;
    float x = neg ? (0.f - x0) : x0; /* abs via select(neg(x), x, x<0) (ops.rs:649-651) */
    const float p = 0.3275911f;
    const float a0 = 0.254829592f, a1 = -0.284496736f, a2 = 1.421413741f, a3 = -1.453152027f,
                a4 = 1.061405429f;
    float t = 1.0f / fma32(x, p, 1.0f);
    /* poly_eval (rten-simd/src/ops.rs:571-577): Horner from the last coeff, then * t */
    float y = a4;
    y = fma32(y, t, a3);
    y = fma32(y, t, a2);
    y = fma32(y, t, a1);
    y = fma32(y, t, a0);
    float at = y * t;
    float xm2 = 0.f - (x * x);
    float e = rto_exp_reduced_f32(xm2);
    float r = 1.0f - at * e;
    return neg ? (0.f - r) : r;
}

static const float SQRT_2_RCP = 0.70710678118654752440f; /* 1.0 / SQRT_2 in f32 (erf.rs:61) */

RTO_API float rto_gelu_f32(float x) { /* erf.rs:63-76 */
    float half_x = x * 0.5f;
    float y = x * SQRT_2_RCP;
    y = rto_erf_f32(y) + 1.0f;
    return half_x * y;
}

RTO_API void rto_gelu(int64_t n, const float *x, float *y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) y[i] = rto_gelu_f32(x[i]);
}
RTO_API void rto_erf(int64_t n, const float *x, float *y) {
    for (int64_t i = 0; i < n; i++) y[i] = rto_erf_f32(x[i]);
}
RTO_API float rto_tanh_f32(float x) { /* rten-vecmath/src/tanh.rs:12-72 */
    /* abs / neg are sign-bit operations on the x86 SIMD back ends (rten-simd/src/arch/x86_64/avx512.rs:314-321, avx2.rs likewise) */
    float ax = bits_f32((int32_t)((uint32_t)f32_bits(x) & 0x7fffffffu));
    const float p1 = 0.999999940395355224609375f, p3 = -0.33332359790802001953125f, p5 = 0.13310669362545013427734375f,
                p7 = -5.21197654306888580322265625e-2f, p9 = 1.5497927553951740264892578125e-2f;
    float x2 = x * x;
    float ys = fma32(p9, x2, p7);
    ys = fma32(ys, x2, p5);
    ys = fma32(ys, x2, p3);
    ys = fma32(ys, x2, p1);
    ys = ys * ax;                       /* |x| <= 0.55: odd polynomial */
    float e = rto_exp_f32(ax * 2.0f);
    float ym = (e - 1.0f) / (e + 1.0f); /* medium |x| */
    float y = ax >= 9.02f ? 1.0f : ym;  /* select(one, y_medium, x_cutoff) */
    y = ax <= 0.55f ? ys : y;           /* select(y_small, y, x_small) */
    y = ax <= 0.0004f ? ax : y;         /* select(abs_x, y, x_tiny) */
    return x <= 0.f ? bits_f32((int32_t)((uint32_t)f32_bits(y) ^ 0x80000000u)) : y; /* select(neg(y), y, le(x, 0)) */
}
RTO_API void rto_tanh(int64_t n, const float *x, float *y) {
    for (int64_t i = 0; i < n; i++) y[i] = rto_tanh_f32(x[i]);
}
RTO_API void rto_exp(int64_t n, const float *x, float *y) {
    for (int64_t i = 0; i < n; i++) y[i] = rto_exp_f32(x[i]);
}
/* Relu -- unary_elementwise.rs:611-613 `val.max(0.)` */
RTO_API void rto_relu(int64_t n, const float *x, float *y) {
    for (int64_t i = 0; i < n; i++) y[i] = fmaxf(x[i], 0.f);
}
/* Add (same shape, or b broadcast with period b_len along the flattened index) --
   binary_elementwise.rs:476-495 */
RTO_API void rto_add(int64_t n, const float *a, const float *b, int64_t b_len, float *y) {
    for (int64_t i = 0; i < n; i++) y[i] = a[i] + b[b_len == n ? i : i % b_len];
}

/* ------------------------------------------------------------------------------------
 * SIMD-ordered reductions.  The reference's vecmath reductions keep V-lane partial sums and
 * combine them at the end (rten-simd/src/iter.rs:97-120 fold_unroll<4>, sum.rs:27-33), so the
 * result depends on the ISA's lane count V (16 for AVX-512, 8 AVX2, 4 NEON/wasm/generic? see
 * `lanes` argument).  The oracle reproduces the order for a given V; GPU parity for reductions is
 * by tolerance, with V = 16 (the GPU box class host, AVX-512) as the default.
 *   fold_unroll<4>: 4 vector accumulators over chunks of 4V, acc0 += acc1, += acc2, += acc3,
 *   then remaining full vectors fold into acc0, then a masked tail (iter.rs fold: masked lanes keep
 *   their old value), then lanes are summed left to right starting from 0.
 * kind: 0 sum(x), 1 sum((x-off)^2) via mul_add, 2 sum(x*x) via mul_add
 * ---------------------------------------------------------------------------------- */
static float simd_reduce(const float *x, int64_t n, int V, int kind, float off) {
    float acc[4][64];
    for (int u = 0; u < 4; u++)
        for (int l = 0; l < V; l++) acc[u][l] = 0.f;
    int64_t i = 0;
    for (; i + 4 * V <= n; i += 4 * V)
        for (int u = 0; u < 4; u++)
            for (int l = 0; l < V; l++) {
                float v = x[i + u * V + l];
                if (kind == 0) acc[u][l] = acc[u][l] + v;
                else if (kind == 1) { float d = v - off; acc[u][l] = fma32(d, d, acc[u][l]); }
                else acc[u][l] = fma32(v, v, acc[u][l]);
            }
    for (int u = 1; u < 4; u++)
        for (int l = 0; l < V; l++) acc[0][l] = acc[0][l] + acc[u][l];
    for (; i + V <= n; i += V)
        for (int l = 0; l < V; l++) {
            float v = x[i + l];
            if (kind == 0) acc[0][l] = acc[0][l] + v;
            else if (kind == 1) { float d = v - off; acc[0][l] = fma32(d, d, acc[0][l]); }
            else acc[0][l] = fma32(v, v, acc[0][l]);
        }
    for (int l = 0; i + l < n; l++) {
        float v = x[i + l];
        if (kind == 0) acc[0][l] = acc[0][l] + v;
        else if (kind == 1) { float d = v - off; acc[0][l] = fma32(d, d, acc[0][l]); }
        else acc[0][l] = fma32(v, v, acc[0][l]);
    }
    float s = 0.f;
    for (int l = 0; l < V; l++) s = s + acc[0][l];
    return s;
}

RTO_API float rto_simd_sum(const float *x, int64_t n, int lanes) { return simd_reduce(x, n, lanes, 0, 0.f); }

/* Softmax -- rten-vecmath/src/softmax.rs:60-100,178-228.  max starts at f32::MIN; exp via
   ReducedRangeExp(x - max); sum in V-lane order (single accumulator, softmax.rs:204-224);
   normalise by multiplying with reciprocal(sum) = 1/sum (ops.rs:639-641). */
RTO_API void rto_softmax_row(int64_t n, const float *x, const float *addend, float *y, int flush_nan,
                             int lanes) {
    if (n == 0) return;
    float tmp_stack[1024];
    float *t = n <= 1024 ? tmp_stack : (float *)malloc((size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; i++) t[i] = addend ? x[i] + addend[i] : x[i]; /* attention.rs:59-61 */
    float mx = -FLT_MAX;
    for (int64_t i = 0; i < n; i++) mx = fmaxf(mx, t[i]); /* softmax.rs:180-192; order-independent for non-NaN data */
    float acc[64];
    for (int l = 0; l < lanes; l++) acc[l] = 0.f;
    for (int64_t i = 0; i < n; i++) {
        float e = rto_exp_reduced_f32(t[i] - mx);
        t[i] = e;
        acc[i % lanes] = acc[i % lanes] + e;
    }
    float s = 0.f;
    for (int l = 0; l < lanes; l++) s = s + acc[l];
    float inv = 1.0f / s;
    for (int64_t i = 0; i < n; i++) {
        float v = t[i] * inv;
        if (flush_nan && !(v == v)) v = 0.f;
        y[i] = v;
    }
    if (t != tmp_stack) free(t);
}

/* rows x cols softmax along the last axis.  addend (optional) is broadcast: row r uses
   addend + (r / add_div % add_mod) * cols -- covers the [B,1,1,S] mask against [B,H,S,S] scores
   (add_div = H*S, add_mod = B) and the same-shape case (add_div = 1, add_mod = rows). */
RTO_API void rto_softmax(int64_t rows, int64_t cols, const float *x, const float *addend, int64_t add_div,
                         int64_t add_mod, float *y, int flush_nan, int lanes) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        const float *a = addend ? addend + ((r / add_div) % add_mod) * cols : NULL;
        rto_softmax_row(cols, x + r * cols, a, y + r * cols, flush_nan, lanes);
    }
}

/* LayerNormalization -- src/ops/norm.rs:103-161 (normalize_slice), :456-529;
   rten-vecmath/src/normalize.rs:82-170.  gamma/beta: NULL => use the scalars. */
RTO_API void rto_layer_norm(int64_t rows, int64_t cols, const float *x, const float *gamma,
                            const float *beta, float gamma_scalar, float beta_scalar, float eps, float *y,
                            int lanes) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        const float *in = x + r * cols;
        float *out = y + r * cols;
        float mean = simd_reduce(in, cols, lanes, 0, 0.f) / (float)cols;          /* norm.rs:124 */
        float var = simd_reduce(in, cols, lanes, 1, mean) / (float)cols;         /* norm.rs:125 */
        float ssr = gamma_scalar / sqrtf(var + eps);                             /* norm.rs:146 */
        if (!gamma && !beta) {               /* normalize.rs:112-127 */
            for (int64_t i = 0; i < cols; i++) out[i] = fma32(in[i] - mean, ssr, beta_scalar);
        } else if (gamma && !beta && beta_scalar == 0.f) { /* normalize.rs:128-145 */
            for (int64_t i = 0; i < cols; i++) out[i] = (in[i] - mean) * (gamma[i] * ssr);
        } else {                             /* normalize.rs:146-166 */
            for (int64_t i = 0; i < cols; i++) {
                float sv = (gamma ? gamma[i] : 1.0f) * ssr;
                float bv = (beta ? beta[i] : 0.f) + beta_scalar;
                out[i] = fma32(in[i] - mean, sv, bv);
            }
        }
    }
}

/* BatchNormalization (inference) -- norm.rs:194-224 + normalize.rs:112-127:
   y = fma(x - mean_c, scale_c / sqrt(var_c + eps), bias_c) */
RTO_API void rto_batch_norm(int64_t N, int64_t C, int64_t inner, const float *x, const float *scale,
                            const float *bias, const float *mean, const float *var, float eps, float *y) {
    for (int64_t n = 0; n < N; n++)
        for (int64_t c = 0; c < C; c++) {
            float ssr = scale[c] / sqrtf(var[c] + eps);
            const float *in = x + (n * C + c) * inner;
            float *out = y + (n * C + c) * inner;
            for (int64_t i = 0; i < inner; i++) out[i] = fma32(in[i] - mean[c], ssr, bias[c]);
        }
}

/* ------------------------------------------------------------------------------------
 * Pooling -- src/ops/pooling.rs:174-389 (pool_impl), :392-417 (average), :581-600 (max),
 * :477-521 (global).  Padding cells are skipped; window is walked ky-major then kx.
 * ---------------------------------------------------------------------------------- */
RTO_API void rto_pool2d(int64_t N, int64_t C, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t sh,
                        int64_t sw, int64_t pt, int64_t pl, int64_t OH, int64_t OW, const float *x,
                        float *y, int is_max, int count_include_pad) {
#pragma omp parallel for schedule(static)
    for (int64_t nc = 0; nc < N * C; nc++) {
        const float *in = x + nc * H * W;
        float *out = y + nc * OH * OW;
        for (int64_t oy = 0; oy < OH; oy++)
            for (int64_t ox = 0; ox < OW; ox++) {
                float acc = is_max ? -INFINITY : 0.f;
                int64_t cnt = 0;
                for (int64_t ky = 0; ky < kh; ky++)
                    for (int64_t kx = 0; kx < kw; kx++) {
                        int64_t iy = oy * sh + ky, ix = ox * sw + kx;
                        if (iy >= pt && iy < H + pt && ix >= pl && ix < W + pl) {
                            float v = in[(iy - pt) * W + (ix - pl)];
                            acc = is_max ? fmaxf(acc, v) : acc + v;
                            cnt++;
                        }
                    }
                if (!is_max) acc = count_include_pad ? acc / (float)(kh * kw) : acc / (float)cnt;
                out[oy * OW + ox] = acc;
            }
    }
}

/* GlobalAveragePool -- pooling.rs:516-521: vecmath::Sum / len */
RTO_API void rto_global_avg_pool(int64_t NC, int64_t inner, const float *x, float *y, int lanes) {
    for (int64_t i = 0; i < NC; i++) y[i] = simd_reduce(x + i * inner, inner, lanes, 0, 0.f) / (float)inner;
}

/* ------------------------------------------------------------------------------------
 * Scaled dot-product attention for one (batch, head) -- src/ops/attention.rs:518-562:
 * scores = scale * Q K^T (gemm alpha), row softmax with NaN flush (+ optional additive mask row
 * applied by score_mod), out = scores . V
 * q:[S,D] k:[T,D] v:[T,Dv] mask:[S,T] or NULL (mask_rs: row stride, 0 = broadcast row)
 * ---------------------------------------------------------------------------------- */
RTO_API void rto_sdpa_head(int64_t S, int64_t T, int64_t D, int64_t Dv, const float *q, const float *k,
                           const float *v, const float *mask, int64_t mask_rs, float scale, float *out,
                           int lanes, int flush_nan) {
    float *scores = (float *)malloc((size_t)S * T * sizeof(float));
    /* both products go through gemm_impl with unpacked operands: ONE query row takes the vector-matrix kernels (lib.rs:876-891), like any other
     * one-row product; rto_gemm_f32 makes that choice */
    rto_gemm_f32(S, T, D, q, D, 1, k, 1, D, scores, T, scale, 0.f, NULL, 0);
    for (int64_t s = 0; s < S; s++)
        rto_softmax_row(T, scores + s * T, mask ? mask + s * mask_rs : NULL, scores + s * T, flush_nan, lanes);
    rto_gemm_f32(S, Dv, T, scores, T, 1, v, Dv, 1, out, Dv, 1.f, 0.f, NULL, 0);
    free(scores);
}

RTO_API void rto_sdpa(int64_t BH, int64_t S, int64_t T, int64_t D, int64_t Dv, const float *q,
                      const float *k, const float *v, const float *mask, int64_t mask_bh_div,
                      int64_t mask_rs, float scale, float *out, int lanes, int flush_nan) {
#pragma omp parallel for schedule(dynamic)
    for (int64_t i = 0; i < BH; i++) {
        const float *m = mask ? mask + (i / mask_bh_div) * (mask_rs ? S * T : T) : NULL;
        rto_sdpa_head(S, T, D, Dv, q + i * S * D, k + i * T * D, v + i * T * Dv, m, mask_rs, scale,
                      out + i * S * Dv, lanes, flush_nan);
    }
}

RTO_API int rto_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
