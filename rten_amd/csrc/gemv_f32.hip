// M == 1 products in the reference's own order: rten-gemm/src/lib.rs:668-747 (gemv: column blocks, depth blocks, bias),
// :876-891 (taken when A has one row and B is not prepacked -- ModelOptions::prepack_weights defaults to false), and
// kernels/simd_generic.rs:14-197, AVX-512 instantiation (16 lanes, 32-column tiles).  This is NOT the blocked GEMM's one chain
// per output element, so the MFMA kernels cannot produce its bits:
//   * B with unit column stride: depth blocks of 8; per block a k-ordered fmaf chain from 0, times alpha, folded into the output
//     with the block's beta (the caller's, then 1) by store / add / fma.  Columns left over after the 32-wide tiles of a column
//     block take the reference's scalar loop: separate multiply and add, `beta * out + acc * alpha`.
//   * B with unit row stride (transB, transposed views): depth blocks of 512; 16 lane accumulators per column over 16-element
//     depth tiles, _mm512_reduce_add_ps (lane l + l+8, +4, +2, +1), scalar fmaf tail, `alpha * acc (+ beta * out)`; the < 8 columns
//     left over in a column block take the scalar fmaf chain of the fallback kernel.
//   * neither stride 1: depth blocks of 8, scalar fmaf chain, `acc * alpha (+ beta * out)`.
// Column blocks are max(128, ceil(N / threads)) columns (lib.rs:697): WHICH columns are left over depends on the reference's thread
// count; rten_hip_set_gemv_order(ctx, on, threads) states the assumption (0 = at least N / 128 threads: blocks of 128).
// One product is a few microseconds of work; the kernels are written for the order, not for speed.
#include "internal.h"
#include "vecmath.h"

namespace {
struct GemvArgs {
    const float *A, *B, *bias;
    float *C;
    int N, K;
    long long a_cs, b_rs, b_cs;
    long long a_bs, b_bs, c_bs, a_bsi, b_bsi, c_bsi;
    int batch_inner;
    float alpha, beta;
    int bias_kind, act, col_block;
};

__device__ __forceinline__ float finish(const GemvArgs &p, float o, int col) {
    if (p.bias_kind == RTEN_HIP_BIAS_PER_ROW) o = o + p.bias[0];      // BiasVector::Column of a one-row product
    else if (p.bias_kind == RTEN_HIP_BIAS_PER_COL) o = o + p.bias[col];
    if (p.act == RTEN_HIP_ACT_RELU) o = vm::relu(o);
    else if (p.act == RTEN_HIP_ACT_GELU) o = vm::gelu(o);
    return o;
}

__device__ __forceinline__ void batch_bases(const GemvArgs &p, const float *&a, const float *&b, float *&c) {
    const int z = blockIdx.y;
    int zo = z, zi = 0;
    if (p.batch_inner > 1) { zo = z / p.batch_inner; zi = z - zo * p.batch_inner; }
    a = p.A + zo * p.a_bs + zi * p.a_bsi;
    b = p.B + zo * p.b_bs + zi * p.b_bsi;
    c = p.C + zo * p.c_bs + zi * p.c_bsi;
}

// one thread per column: row-major B (SCALAR_ALL = false: fmaf tiles + unfused left-over columns) or arbitrary strides (true)
template <bool SCALAR_ALL>
__global__ __launch_bounds__(256) void gemv_cols_kernel(const GemvArgs p) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= p.N) return;
    const float *a, *b;
    float *c;
    batch_bases(p, a, b, c);
    const int c0 = col / p.col_block * p.col_block, nc = p.N - c0 < p.col_block ? p.N - c0 : p.col_block;
    const bool tiled = !SCALAR_ALL && (col - c0) < nc / 32 * 32;
    float o = p.beta == 0.f ? 0.f : c[col];
    float eff_beta = p.beta;
    const float *bc = b + (long long)col * p.b_cs;
    for (int k0 = 0; k0 < p.K; k0 += 8) {
        const int depth = p.K - k0 < 8 ? p.K - k0 : 8;
        float acc = 0.f;
        if (SCALAR_ALL || tiled) {
            for (int k = 0; k < depth; k++) acc = fmaf(a[(long long)(k0 + k) * p.a_cs], bc[(long long)(k0 + k) * p.b_rs], acc);
            if (SCALAR_ALL) {
                acc = acc * p.alpha;
                o = eff_beta == 0.f ? acc : acc + eff_beta * o;
            } else {
                if (p.alpha != 1.f) acc = acc * p.alpha;
                o = eff_beta == 0.f ? acc : (eff_beta == 1.f ? o + acc : fmaf(o, eff_beta, acc));
            }
        } else {
            for (int k = 0; k < depth; k++) acc = acc + a[(long long)(k0 + k) * p.a_cs] * bc[(long long)(k0 + k) * p.b_rs]; // -ffp-contract=off: two roundings
            const float tmp = eff_beta == 0.f ? 0.f : o;
            o = eff_beta * tmp + acc * p.alpha;
        }
        eff_beta = 1.f;
    }
    c[col] = finish(p, o, col);
}

// B with unit row stride: 16 lanes per column (4 columns per wave, 16 per workgroup)
__global__ __launch_bounds__(256) void gemv_transposed_kernel(const GemvArgs p) {
    const int lane16 = threadIdx.x & 15;
    const int col = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = col < p.N;
    const float *a, *b;
    float *c;
    batch_bases(p, a, b, c);
    const int cc = live ? col : 0;
    const int c0 = cc / p.col_block * p.col_block, nc = p.N - c0 < p.col_block ? p.N - c0 : p.col_block;
    const bool vec = (cc - c0) < nc / 8 * 8; // else: a left-over column of its block -> the fallback kernel's scalar chain
    const float *bc = b + (long long)cc * p.b_cs;
    float o = (p.beta == 0.f || !live) ? 0.f : c[cc];
    float eff_beta = p.beta;
    for (int k0 = 0; k0 < p.K; k0 += 512) {
        const int depth = p.K - k0 < 512 ? p.K - k0 : 512;
        float acc;
        if (vec) {
            const int dt = depth / 16 * 16;
            float part = 0.f;
            for (int d = 0; d < dt; d += 16) part = fmaf(a[(long long)(k0 + d + lane16) * p.a_cs], bc[k0 + d + lane16], part);
            part = part + __shfl_down(part, 8, 16);
            part = part + __shfl_down(part, 4, 16);
            part = part + __shfl_down(part, 2, 16);
            part = part + __shfl_down(part, 1, 16);
            acc = part; // lane 0 of the group
            for (int k = dt; k < depth; k++) acc = fmaf(a[(long long)(k0 + k) * p.a_cs], bc[k0 + k], acc);
            const float pa = p.alpha * acc;
            o = eff_beta == 0.f ? pa : pa + eff_beta * o;
        } else {
            acc = 0.f;
            for (int k = 0; k < depth; k++) acc = fmaf(a[(long long)(k0 + k) * p.a_cs], bc[k0 + k], acc);
            acc = acc * p.alpha;
            o = eff_beta == 0.f ? acc : acc + eff_beta * o;
        }
        eff_beta = 1.f;
    }
    if (live && lane16 == 0) c[col] = finish(p, o, col);
}
} // namespace

// Called by rten_hip_gemm_f32 for m == 1, k > 0 when the context's gemv order is on.
int32_t rten_gemv_f32(rten_hip_ctx *ctx, const rten_hip_gemm_desc *d, const float *a, const float *b, const float *bias, float *c) {
    GemvArgs g = {};
    g.A = a; g.B = b; g.bias = bias; g.C = c;
    g.N = d->n; g.K = d->k;
    g.a_cs = d->a_cs; g.b_rs = d->b_rs; g.b_cs = d->b_cs;
    g.a_bs = d->a_bs; g.b_bs = d->b_bs; g.c_bs = d->c_bs; g.a_bsi = d->a_bsi; g.b_bsi = d->b_bsi; g.c_bsi = d->c_bsi;
    g.batch_inner = d->batch_inner;
    g.alpha = d->alpha; g.beta = d->beta;
    g.bias_kind = d->bias_kind; g.act = d->act;
    long long cb = 128;
    if (ctx->gemv_threads > 0) {
        cb = ((long long)d->n + ctx->gemv_threads - 1) / ctx->gemv_threads;
        if (cb < 128) cb = 128;
    }
    g.col_block = (int)(cb > 0x7fffffff ? 0x7fffffff : cb);
    const unsigned batch = (unsigned)(d->batch < 1 ? 1 : d->batch);
    ProfScope ps(ctx, d->b_rs == 1 ? "gemv_transposed_kernel" : "gemv_cols_kernel", 2.0 * d->n * (double)d->k * batch,
                 4.0 * ((double)d->n * d->k + d->k + d->n) * batch);
    if (d->b_rs == 1) hipLaunchKernelGGL(gemv_transposed_kernel, dim3((unsigned)((d->n + 15) / 16), batch), dim3(256), 0, ctx->stream, g);
    else if (d->b_cs != 1) hipLaunchKernelGGL(gemv_cols_kernel<true>, dim3((unsigned)((d->n + 255) / 256), batch), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL(gemv_cols_kernel<false>, dim3((unsigned)((d->n + 255) / 256), batch), dim3(256), 0, ctx->stream, g);
    RTEN_LAUNCH_CHECK(ctx, "gemv kernel launch");
    return RTEN_HIP_OK;
}
