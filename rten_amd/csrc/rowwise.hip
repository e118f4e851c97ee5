// Wavefront-per-row reduction kernels: Softmax / AddSoftmax, LayerNormalization, GlobalAveragePool.
// Replaces rten-vecmath/src/softmax.rs:60-100,178-228, sum.rs:12-34,100-130, normalize.rs:82-170 and
// their operator front-ends src/ops/norm.rs:103-161,456-529,825-840, src/ops/attention.rs:30-68,
// src/ops/pooling.rs:477-521.
//
// MI355X mapping: one 64-lane wavefront owns one row; the row lives in registers (up to 16 values
// per lane) so HBM sees exactly one read and one write per element; 4 rows per 256-thread workgroup.
//
// Reduction order: the reference keeps V-lane SIMD partial sums and adds the lanes left to right at
// the end (rten-simd/src/iter.rs:97-120 fold_unroll<4>, sum.rs:27-33).  A 64-lane wavefront IS four
// 16-lane AVX-512 vectors side by side, so the kernels below reproduce the V = 16 order exactly:
// lane t = 16u + l accumulates x[64c + t] over chunks c (the four unrolled accumulators), lanes 0..15
// then fold acc[l+16], acc[l+32], acc[l+48] in that order, absorb the remaining <4 full vectors and the
// masked tail, and the 16 lane totals are added sequentially.  Results are therefore bit-identical to
// the reference running on an AVX-512 host (and to oracle lanes=16); hosts with another SIMD width
// differ by reduction-order rounding only (tolerance in DESIGN.md).
#include <type_traits>

#include "internal.h"
#include "vecmath.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;
constexpr int MAX_CH = 16; // register-resident rows up to 1024 columns

__device__ __forceinline__ float lane_bcast(float v, int src_lane) { return __shfl(v, src_lane, 64); }

// fold_unroll<4> order (sum.rs:27-33,110-127).  kind 0: sum x ; kind 1: sum (x-off)^2 via mul_add.
// `get(i)` returns element i (i < n).  All lanes return the same total.
template <int KIND, typename Get>
__device__ __forceinline__ float simd16_reduce(Get get, int n, float off, int lane) {
    auto f = [&](float acc, float x) -> float {
        if constexpr (KIND == 0) return acc + x;
        else { const float d = x - off; return vm::fma(d, d, acc); }
    };
    float acc = 0.f;
    const int full4 = n / 64;
    for (int c = 0; c < full4; c++) acc = f(acc, get(c * 64 + lane));
    // acc0 += acc1; += acc2; += acc3  (lanes 0..15 hold the running vector)
    float a = acc;
    a = a + lane_bcast(acc, (lane & 15) + 16);
    a = a + lane_bcast(acc, (lane & 15) + 32);
    a = a + lane_bcast(acc, (lane & 15) + 48);
    int i0 = full4 * 64;
    const int l = lane & 15;
    for (; i0 + 16 <= n; i0 += 16) a = f(a, get(i0 + l));
    if (i0 + l < n) a = f(a, get(i0 + l)); // masked tail: other lanes keep their value
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), k));
    return s;
}

// ------------------------------------------------------------------------------------------------
// Softmax: max (f32::MIN start), e = ReducedRangeExp(x - max), sum in single-accumulator 16-lane order
// (softmax.rs:194-228), y = e * (1 / sum), optional NaN -> 0.
// ------------------------------------------------------------------------------------------------
// One wave per row, R consecutive rows per wave: short rows (128 columns = 2 elements per lane) are latency-bound on the
// load -> max -> exp -> sum -> scale chain, so a wave keeps R independent chains in flight (all loads first).
template <int CH, int R>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void softmax_kernel(int64_t rows, int cols, const float *__restrict__ x,
                                                                      const float *__restrict__ addend, int64_t add_div,
                                                                      int64_t add_mod, int flush_nan,
                                                                      float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    float v[R][CH];
    float mx[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1; // clamped: the tail rows of the last wave recompute the last row
        const float *xr = x + row * cols;
        const float *ar = addend ? addend + ((row / add_div) % add_mod) * cols : nullptr;
        mx[r] = -3.40282347e+38f; // f32::MIN (softmax.rs:181)
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int i = c * 64 + lane;
            float t = -3.40282347e+38f;
            if (i < cols) {
                t = xr[i];
                if (ar) t = t + ar[i]; // `*qk += m` (attention.rs:59-61)
                mx[r] = fmaxf(mx[r], t);
            }
            v[r][c] = t;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; r++) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
    // exp + 16-lane ordered sum: lane l < 16 adds e[64c + l], e[64c + l + 16], e[64c + l + 32], e[64c + l + 48]
    float a[R];
    const int l = lane & 15;
#pragma unroll
    for (int r = 0; r < R; r++) {
        a[r] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int i = c * 64 + lane;
            const float e = i < cols ? vm::exp_reduced(v[r][c] - mx[r]) : 0.f;
            v[r][c] = e;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float eq = lane_bcast(e, l + 16 * q);
                if (c * 64 + l + 16 * q < cols) a[r] = a[r] + eq;
            }
        }
    }
    float inv[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[r]), k));
        inv[r] = 1.0f / s;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (row0 + r >= rows) break;
        float *yr = y + (row0 + r) * cols;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int i = c * 64 + lane;
            if (i < cols) {
                float o = v[r][c] * inv[r];
                if (flush_nan && !(o == o)) o = 0.f;
                yr[i] = o;
            }
        }
    }
}

// Short rows (<= 256 columns): FOUR rows per wave, one per 16-lane DPP row.  Lane l of a row owns elements l, l + 16,
// l + 32, ... -- exactly the elements the reference's 16-lane accumulator lane l adds, in the order it adds them
// (softmax.rs:194-228), so the partial sums need no cross-lane traffic at all; the row maximum is 4 DPP steps, and the final
// in-order sum of the 16 partials is a 15-step DPP shift-and-add chain (lane 15 ends with ((a0 + a1) + a2) + ... + a15) shared
// by the four rows.  The wave-per-row kernel above spends ~30 LDS-crossbar / readlane operations per row on the same thing.
template <int EPL>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void softmax_rows16_kernel(int64_t rows, int cols, const float *__restrict__ x,
                                                                             const float *__restrict__ addend, int64_t add_div,
                                                                             int64_t add_mod, int flush_nan,
                                                                             float *__restrict__ y) {
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < rows;
    const int64_t rr = live ? row : rows - 1;
    const float *xr = x + rr * cols;
    const float *ar = addend ? addend + ((rr / add_div) % add_mod) * cols : nullptr;
    float v[EPL];
    float mx = -3.40282347e+38f; // f32::MIN (softmax.rs:181)
#pragma unroll
    for (int j = 0; j < EPL; j++) {
        const int i = l + 16 * j;
        v[j] = xr[i < cols ? i : 0];
    }
    if (ar) {
#pragma unroll
        for (int j = 0; j < EPL; j++) {
            const int i = l + 16 * j;
            v[j] = v[j] + ar[i < cols ? i : 0]; // `*qk += m` (attention.rs:59-61)
        }
    }
#pragma unroll
    for (int j = 0; j < EPL; j++) {
        if (l + 16 * j >= cols) v[j] = -3.40282347e+38f;
        mx = fmaxf(mx, v[j]);
    }
    // max over the 16 lanes of the row: quad xor 1, quad xor 2, row rotate 4, row rotate 8
    auto dpp = [](float f, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(f), decltype(ctrl)::value, 0xf, 0xf, true)); };
    mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0xB1>()));
    mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0x4E>()));
    mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0x124>()));
    mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0x128>()));
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; j++) {
        const float e = l + 16 * j < cols ? vm::exp_reduced(v[j] - mx) : 0.f;
        v[j] = e;
        if (l + 16 * j < cols) a = a + e;
    }
    // s = ((0 + a0) + a1) + ... + a15 in lane 15: shift right by one lane within the row (lane 0 receives 0), add own partial
    float acc = a;
#pragma unroll
    for (int k = 1; k < 16; k++) acc = dpp(acc, std::integral_constant<int, 0x111>()) + a;
    const float s = __int_as_float(__builtin_amdgcn_ds_bpermute((lane | 15) << 2, __float_as_int(acc)));
    const float inv = 1.0f / s;
    if (!live) return;
    float *yr = y + row * cols;
#pragma unroll
    for (int j = 0; j < EPL; j++) {
        const int i = l + 16 * j;
        if (i < cols) {
            float o = v[j] * inv;
            if (flush_nan && !(o == o)) o = 0.f;
            yr[i] = o;
        }
    }
}

// Generic fallback for long rows: same order, row re-read from global (L2 resident).
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void softmax_long_kernel(int64_t rows, int cols,
                                                                           const float *__restrict__ x,
                                                                           const float *__restrict__ addend,
                                                                           int64_t add_div, int64_t add_mod,
                                                                           int flush_nan, float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * cols;
    const float *ar = addend ? addend + ((row / add_div) % add_mod) * cols : nullptr;
    float *yr = y + row * cols;
    float mx = -3.40282347e+38f;
    for (int i = lane; i < cols; i += 64) {
        float t = xr[i];
        if (ar) t = t + ar[i];
        mx = fmaxf(mx, t);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float a = 0.f;
    const int l = lane & 15;
    const int nch = (cols + 63) / 64;
    for (int c = 0; c < nch; c++) {
        const int i = c * 64 + lane;
        float e = 0.f;
        if (i < cols) {
            float t = xr[i];
            if (ar) t = t + ar[i];
            e = vm::exp_reduced(t - mx);
            yr[i] = e;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float eq = lane_bcast(e, l + 16 * q);
            if (c * 64 + l + 16 * q < cols) a = a + eq;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), k));
    const float inv = 1.0f / s;
    for (int i = lane; i < cols; i += 64) {
        float r = yr[i] * inv;
        if (flush_nan && !(r == r)) r = 0.f;
        yr[i] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNormalization (norm.rs:103-161): mean = Sum/n, var = SumSquareSub(mean)/n (two-pass),
// ssr = gamma_scalar / sqrt(var + eps), then one of the three Normalize forms (normalize.rs:112-166).
// ------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void layer_norm_kernel(int64_t rows, int cols,
                                                                         const float *__restrict__ x,
                                                                         const float *__restrict__ gamma,
                                                                         const float *__restrict__ beta,
                                                                         float gamma_scalar, float beta_scalar,
                                                                         float eps, const float *addend, float *y) {
    // addend != nullptr: the Add that precedes the LayerNormalization in transformer blocks, fused (x + addend is formed
    // once, with the same single f32 add, and never written to memory).  y may alias x or addend: a wave reads its whole
    // row before it stores.
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * cols;
    const float *ar = addend ? addend + row * cols : nullptr;
    float *yr = y + row * cols;
    [[maybe_unused]] float v[CH > 0 ? CH : 1];
    // per-column scale / bias of the register-resident form: requested HERE, together with the row itself -- they depend on nothing, and
    // behind the two ordered reductions their L2 round trip (~1 us) would be the tail of every wave (round 4: 6.1 -> see ops_microbench.json)
    [[maybe_unused]] float gv[CH > 0 ? CH : 1], bv[CH > 0 ? CH : 1];
    if constexpr (CH > 0) {
        float xa[CH], aa[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; xa[c] = i < cols ? xr[i] : 0.f; }
        if (ar) {
#pragma unroll
            for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; aa[c] = i < cols ? ar[i] : 0.f; }
        }
#pragma unroll
        for (int c = 0; c < CH; c++) { gv[c] = 1.0f; bv[c] = 0.f; }
        if (gamma) {
#pragma unroll
            for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; gv[c] = gamma[i < cols ? i : 0]; }
        }
        if (beta) {
#pragma unroll
            for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; bv[c] = beta[i < cols ? i : 0]; }
        }
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = ar ? xa[c] + aa[c] : xa[c]; // (0 + 0 for the masked tail: never used)
    }
    // element fetch by index for the ordered reduction: register-resident rows are indexed through
    // their owning lane (i % 64 == this lane for the unrolled part; the <64-element remainder needs a
    // cross-lane read, served from global/L1 instead of a dynamic register index).
    auto get = [&](int i) -> float { return ar ? xr[i] + ar[i] : xr[i]; };
    float mean, var;
    if constexpr (CH > 0) {
        // Same order as simd16_reduce, but the full-chunk part reads registers.
        auto red = [&](auto f) -> float {
            float acc = 0.f;
            const int full4 = cols / 64;
#pragma unroll
            for (int c = 0; c < CH; c++)
                if (c < full4) acc = f(acc, v[c]);
            float a = acc;
            a = a + lane_bcast(acc, (lane & 15) + 16);
            a = a + lane_bcast(acc, (lane & 15) + 32);
            a = a + lane_bcast(acc, (lane & 15) + 48);
            int i0 = full4 * 64;
            const int l = lane & 15;
            for (; i0 + 16 <= cols; i0 += 16) a = f(a, get(i0 + l));
            if (i0 + l < cols) a = f(a, get(i0 + l));
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), k));
            return s;
        };
        mean = red([](float acc, float xv) { return acc + xv; }) / (float)cols;
        const float m = mean;
        var = red([m](float acc, float xv) { const float d = xv - m; return vm::fma(d, d, acc); }) / (float)cols;
    } else {
        mean = simd16_reduce<0>(get, cols, 0.f, lane) / (float)cols;
        var = simd16_reduce<1>(get, cols, mean, lane) / (float)cols;
    }
    const float ssr = gamma_scalar / sqrtf(var + eps);
    const int mode = (!gamma && !beta) ? 0 : ((gamma && !beta && beta_scalar == 0.f) ? 1 : 2);
    auto norm = [&](float xv, float gv, float bv) -> float { // the three Normalize forms, normalize.rs:112-166
        if (mode == 0) return vm::fma(xv - mean, ssr, beta_scalar);
        if (mode == 1) return (xv - mean) * (gv * ssr);
        return vm::fma(xv - mean, gv * ssr, bv + beta_scalar);
    };
    if constexpr (CH > 0) {
        // (uniform `mode` tests hoisted out of the element loop -- a test per element serialises every load behind a waitcnt)
        if (mode == 0) {
#pragma unroll
            for (int c = 0; c < CH; c++) v[c] = vm::fma(v[c] - mean, ssr, beta_scalar);
        } else if (mode == 1) {
#pragma unroll
            for (int c = 0; c < CH; c++) v[c] = (v[c] - mean) * (gv[c] * ssr);
        } else {
#pragma unroll
            for (int c = 0; c < CH; c++) v[c] = vm::fma(v[c] - mean, gv[c] * ssr, bv[c] + beta_scalar);
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int i = c * 64 + lane;
            if (i < cols) yr[i] = v[c];
        }
    } else {
        for (int i = lane; i < cols; i += 64) yr[i] = norm(get(i), gamma ? gamma[i] : 1.0f, beta ? beta[i] : 0.f);
    }
}

// Rows in sequence (round 6): the kernel above gives every row a wave of its own, so a [4096, 768] launch is ONE wave of 4096 wavefronts that all load, then all
// reduce, then all store -- the read stream and the write stream never overlap (6.1 us against 3.8 us for a copy of the same bytes).  Here a wave owns R
// CONSECUTIVE rows and walks them with the next row's loads in flight under the current row's reductions and stores (two register images, statically
// alternated), so HBM sees reads and writes together for most of the launch.  Same arithmetic, same order per row: bit-identical.  Register-resident rows
// (CH x 64 columns) only.
template <int CH>
struct LnRow { float xa[CH], aa[CH]; };
template <int CH, bool ADD>
__device__ __forceinline__ void ln_load_row(LnRow<CH> &r, const float *__restrict__ xr, const float *__restrict__ ar, int cols, int lane) {
#pragma unroll
    for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; r.xa[c] = i < cols ? xr[i] : 0.f; }
    if constexpr (ADD) {
#pragma unroll
        for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; r.aa[c] = i < cols ? ar[i] : 0.f; }
    }
}
template <int CH, bool ADD>
__device__ __forceinline__ void ln_finish_row(const LnRow<CH> &r, const float *__restrict__ xr, const float *__restrict__ ar, float *__restrict__ yr, int cols, int lane,
                                              const float (&gv)[CH], const float (&bv)[CH], int mode, float gamma_scalar, float beta_scalar, float eps) {
    float v[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) v[c] = ADD ? r.xa[c] + r.aa[c] : r.xa[c];
    auto get = [&](int i) -> float { return ADD ? xr[i] + ar[i] : xr[i]; }; // (the < 64-column remainder of a row: cross-lane, served from L1)
    auto red = [&](auto f) -> float {
        float acc = 0.f;
        const int full4 = cols / 64;
#pragma unroll
        for (int c = 0; c < CH; c++)
            if (c < full4) acc = f(acc, v[c]);
        float a = acc;
        a = a + lane_bcast(acc, (lane & 15) + 16);
        a = a + lane_bcast(acc, (lane & 15) + 32);
        a = a + lane_bcast(acc, (lane & 15) + 48);
        int i0 = full4 * 64;
        const int l = lane & 15;
        for (; i0 + 16 <= cols; i0 += 16) a = f(a, get(i0 + l));
        if (i0 + l < cols) a = f(a, get(i0 + l));
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), k));
        return s;
    };
    const float mean = red([](float acc, float xv) { return acc + xv; }) / (float)cols;
    const float var = red([mean](float acc, float xv) { const float d = xv - mean; return vm::fma(d, d, acc); }) / (float)cols;
    const float ssr = gamma_scalar / sqrtf(var + eps);
    if (mode == 0) {
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = vm::fma(v[c] - mean, ssr, beta_scalar);
    } else if (mode == 1) {
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = (v[c] - mean) * (gv[c] * ssr);
    } else {
#pragma unroll
        for (int c = 0; c < CH; c++) v[c] = vm::fma(v[c] - mean, gv[c] * ssr, bv[c] + beta_scalar);
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const int i = c * 64 + lane;
        if (i < cols) yr[i] = v[c];
    }
}
template <int CH, bool ADD>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void layer_norm_stream_kernel(int64_t rows, int cols, const float *__restrict__ x, const float *__restrict__ gamma,
                                                                                const float *__restrict__ beta, float gamma_scalar, float beta_scalar, float eps,
                                                                                const float *addend, float *y, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * rows_per_wave;
    if (row0 >= rows) return;
    const int nrows = (int)(rows - row0 < rows_per_wave ? rows - row0 : rows_per_wave);
    LnRow<CH> ra, rb;
    ln_load_row<CH, ADD>(ra, x + row0 * cols, ADD ? addend + row0 * cols : nullptr, cols, lane);
    float gv[CH], bv[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { gv[c] = 1.0f; bv[c] = 0.f; }
    if (gamma) {
#pragma unroll
        for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; gv[c] = gamma[i < cols ? i : 0]; }
    }
    if (beta) {
#pragma unroll
        for (int c = 0; c < CH; c++) { const int i = c * 64 + lane; bv[c] = beta[i < cols ? i : 0]; }
    }
    const int mode = (!gamma && !beta) ? 0 : ((gamma && !beta && beta_scalar == 0.f) ? 1 : 2);
    for (int k = 0; k < nrows; k += 2) {
        const int64_t r0 = row0 + k;
        if (k + 1 < nrows) ln_load_row<CH, ADD>(rb, x + (r0 + 1) * cols, ADD ? addend + (r0 + 1) * cols : nullptr, cols, lane);
        ln_finish_row<CH, ADD>(ra, x + r0 * cols, ADD ? addend + r0 * cols : nullptr, y + r0 * cols, cols, lane, gv, bv, mode, gamma_scalar, beta_scalar, eps);
        if (k + 1 >= nrows) break;
        if (k + 2 < nrows) ln_load_row<CH, ADD>(ra, x + (r0 + 2) * cols, ADD ? addend + (r0 + 2) * cols : nullptr, cols, lane);
        ln_finish_row<CH, ADD>(rb, x + (r0 + 1) * cols, ADD ? addend + (r0 + 1) * cols : nullptr, y + (r0 + 1) * cols, cols, lane, gv, bv, mode, gamma_scalar, beta_scalar, eps);
    }
}

// GlobalAveragePool (pooling.rs:516-521): Sum(chan) / len in the same SIMD order.
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void global_avg_pool_kernel(int64_t rows, int inner,
                                                                              const float *__restrict__ x,
                                                                              float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * inner;
    auto get = [&](int i) -> float { return xr[i]; };
    const float s = simd16_reduce<0>(get, inner, 0.f, lane);
    if (lane == 0) y[row] = s / (float)inner;
}

// Planes of at most 64 elements (ResNet's 7x7): four planes per wave, one per 16-lane DPP row.  Lane l owns elements l,
// l + 16, l + 32, l + 48 -- the ones the reference's accumulator lane l adds, in its order -- and the in-order sum of the
// 16 partials is the DPP shift-and-add chain of softmax_rows16_kernel.
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void global_avg_pool_rows16_kernel(int64_t rows, int inner,
                                                                                     const float *__restrict__ x,
                                                                                     float *__restrict__ y) {
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const int64_t rr = row < rows ? row : rows - 1;
    const float *xr = x + rr * inner;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = xr[l + 16 * q < inner ? l + 16 * q : 0];
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (l + 16 * q < inner) a = a + v[q];
    float acc = a;
#pragma unroll
    for (int k = 1; k < 16; k++) acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x111, 0xf, 0xf, true)) + a;
    if (l == 15 && row < rows) y[row] = acc / (float)inner;
}

// ReduceSum over a strided view (src/ops/reduce.rs:414-520,1101-1124): output element r = vecmath::Sum of the reduced
// slice, whose elements are the reduced axes walked in row-major order -- the order in which the reference packs a
// non-contiguous slice before calling the kernel (reduce.rs:470-505).  The view is read in place through its strides
// (stride 0 = broadcast axis, summed strides = a diagonal): nothing is packed.
struct ReduceArgs {
    int n_outer, n_inner;
    int64_t rows;
    int inner;
    float divisor; // 0: ReduceSum; slice length: ReduceMean = Sum / len (reduce.rs:532-537)
    int32_t oshape[6], ishape[6];
    int64_t ostride[6], istride[6];
};

// (The outermost kept axis needs no division -- what is left of the row index IS its coordinate -- and a 64-bit division is ~100 instructions on this
// machine: with one per row the last-axis kernels were division-bound, 10 us for 49152 rows of 128.  32-bit arithmetic whenever the row count allows.)
__device__ __forceinline__ int64_t reduce_row_base(const ReduceArgs &p, int64_t row) {
    if (p.n_outer <= 0) return 0;
    int64_t off = 0;
    if (p.rows <= 0x7fffffff) {
        unsigned r = (unsigned)row;
        for (int d = p.n_outer - 1; d > 0; d--) {
            const unsigned q = r / (unsigned)p.oshape[d];
            off += (int64_t)(r - q * (unsigned)p.oshape[d]) * p.ostride[d];
            r = q;
        }
        return off + (int64_t)r * p.ostride[0];
    }
    int64_t r = row;
    for (int d = p.n_outer - 1; d > 0; d--) {
        const int64_t q = r / p.oshape[d];
        off += (r - q * p.oshape[d]) * p.ostride[d];
        r = q;
    }
    return off + r * p.ostride[0];
}

__device__ __forceinline__ int64_t reduce_elem_off(const ReduceArgs &p, int i) {
    if (p.n_inner == 1) return (int64_t)i * p.istride[0];
    int r = i;
    int64_t off = 0;
    for (int d = p.n_inner - 1; d >= 0; d--) {
        const int q = r / p.ishape[d];
        off += (int64_t)(r - q * p.ishape[d]) * p.istride[d];
        r = q;
    }
    return off;
}

__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void reduce_sum_kernel(const ReduceArgs p, const float *__restrict__ x,
                                                                         float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float *xr = x + reduce_row_base(p, row);
    auto get = [&](int i) -> float { return xr[reduce_elem_off(p, i)]; };
    const float s = simd16_reduce<0>(get, p.inner, 0.f, lane);
    if (lane == 0) y[row] = p.divisor != 0.f ? s / p.divisor : s;
}

// Slices of at most 16 * EPL elements: four output elements per wave, one per 16-lane DPP row (global_avg_pool_rows16_kernel's
// scheme).  Lane l owns elements l + 16 q -- the ones the reference's accumulator lane l adds: the first 4 * (n / 64) of them
// go round-robin into the four unrolled accumulators, which fold left to right; the rest (whole vectors, masked tail) are
// added to the folded value one by one (rten-simd/src/iter.rs:97-120).
template <int EPL>
__global__ __launch_bounds__(64 * ROWS_PER_BLOCK) void reduce_sum_rows16_kernel(const ReduceArgs p, const float *__restrict__ x,
                                                                                float *__restrict__ y) {
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const int64_t rr = row < p.rows ? row : p.rows - 1;
    const float *xr = x + reduce_row_base(p, rr);
    float v[EPL];
#pragma unroll
    for (int q = 0; q < EPL; q++) v[q] = xr[reduce_elem_off(p, l + 16 * q < p.inner ? l + 16 * q : 0)];
    const int unrolled = (p.inner >> 6) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < EPL; q++)
        if (q < unrolled) acc[q & 3] = acc[q & 3] + v[q];
    float a = ((acc[0] + acc[1]) + acc[2]) + acc[3];
#pragma unroll
    for (int q = 0; q < EPL; q++)
        if (q >= unrolled && l + 16 * q < p.inner) a = a + v[q];
    float s = a;
#pragma unroll
    for (int k = 1; k < 16; k++) s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x111, 0xf, 0xf, true)) + a;
    if (l == 15 && row < p.rows) y[row] = p.divisor != 0.f ? s / p.divisor : s;
}

// Reduced axes strided, innermost kept axis contiguous (a column sum): a wave-per-row walk would touch 64 cache lines per
// load.  Here a 1024-thread workgroup owns 16 adjacent output elements j; wave w is the reference's accumulator lane l = w and
// its lanes are (u = unrolled accumulator, j): every load instruction reads four 64-byte runs.  Each thread's chain is the
// reference's acc[u][l]; the fold over u is three lane shuffles, the in-order sum over l goes through LDS.
__global__ __launch_bounds__(1024) void reduce_sum_cols_kernel(const ReduceArgs p, const float *__restrict__ x, float *__restrict__ y) {
    __shared__ float part[16][16];
    const int lane = threadIdx.x & 63, l = threadIdx.x >> 6, u = lane >> 4, j = lane & 15;
    const int last = p.oshape[p.n_outer - 1];
    const int groups = (last + 15) >> 4;
    // Two neighbouring column groups read the two 64-byte halves of the same 128-byte lines.  Workgroup ids go round-robin over the eight XCDs, so
    // neighbours in id order never share an L2 and every line is fetched twice; here each XCD gets a CONTIGUOUS run of groups (ids id, id + 8, ...
    // are dispatched to the same XCD one after the other), and the second half of a line is an L2 hit.
    unsigned bid = blockIdx.x;
    {
        const unsigned nt = gridDim.x, xcd = bid & 7, qn = nt >> 3, rn = nt & 7;
        bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    }
    const int64_t prefix = bid / groups;
    const int j0 = (int)(bid - prefix * groups) * 16;
    const int jj = j0 + j < last ? j0 + j : last - 1;
    const int64_t row = prefix * last + jj;
    const float *xr = x + reduce_row_base(p, row);
    const int n = p.inner, full4 = n >> 6;
    // The chain of adds is the reference's (one accumulator, elements in order); the LOADS are independent, so eight are requested before the
    // first add -- a load per add made this kernel one memory round trip per element (31.5 us for 4096 x 3072 -> 3072: round 3).
    float acc = 0.f;
    int c = 0;
    for (; c + 16 <= full4; c += 16) {
        float tv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) tv[k] = xr[reduce_elem_off(p, (c + k) * 64 + u * 16 + l)];
#pragma unroll
        for (int k = 0; k < 16; k++) acc = acc + tv[k];
    }
    for (; c + 8 <= full4; c += 8) {
        float tv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) tv[k] = xr[reduce_elem_off(p, (c + k) * 64 + u * 16 + l)];
#pragma unroll
        for (int k = 0; k < 8; k++) acc = acc + tv[k];
    }
    for (; c < full4; c++) acc = acc + xr[reduce_elem_off(p, c * 64 + u * 16 + l)];
    float a = lane_bcast(acc, j);
    a = a + lane_bcast(acc, j + 16);
    a = a + lane_bcast(acc, j + 32);
    a = a + lane_bcast(acc, j + 48);
    int i0 = full4 * 64;
    for (; i0 + 16 <= n; i0 += 16) a = a + xr[reduce_elem_off(p, i0 + l)];
    if (i0 + l < n) a = a + xr[reduce_elem_off(p, i0 + l)];
    if (u == 0) part[l][j] = a;
    __syncthreads();
    if (threadIdx.x < 16 && j0 + (int)threadIdx.x < last) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) s = s + part[k][threadIdx.x];
        y[prefix * last + j0 + threadIdx.x] = p.divisor != 0.f ? s / p.divisor : s;
    }
}

} // namespace

RTEN_EXPORT int32_t rten_hip_softmax_f32(rten_hip_ctx *ctx, int64_t rows, int32_t cols, const float *x,
                                         const float *addend, int64_t add_div, int64_t add_mod,
                                         int32_t flush_nan_to_zero, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (rows < 0 || cols < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (rows == 0 || cols == 0) return RTEN_HIP_OK; // norm.rs:711-713
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    if (addend && (add_div <= 0 || add_mod <= 0))
        return rten_set_error(ctx, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "Cannot broadcast inputs");
    if (!addend) { add_div = 1; add_mod = 1; }
    const dim3 block(64 * ROWS_PER_BLOCK);
    auto grid_for = [&](int r) { return dim3((unsigned)((rows + (int64_t)ROWS_PER_BLOCK * r - 1) / ((int64_t)ROWS_PER_BLOCK * r))); };
    const dim3 grid = grid_for(1);
    ProfScope ps(ctx, "softmax_f32", 0.0, 8.0 * rows * cols);
#define SM_LAUNCH(CH, R) hipLaunchKernelGGL((softmax_kernel<CH, R>), grid_for(R), block, 0, ctx->stream, rows, cols, x, addend, add_div, add_mod, flush_nan_to_zero, y)
#define SM16_LAUNCH(EPL) hipLaunchKernelGGL((softmax_rows16_kernel<EPL>), grid_for(4), block, 0, ctx->stream, rows, cols, x, addend, add_div, add_mod, flush_nan_to_zero, y)
    if (cols <= 64) SM16_LAUNCH(4);
    else if (cols <= 128) SM16_LAUNCH(8);
    else if (cols <= 256) SM16_LAUNCH(16);
    else if (cols <= 512) SM_LAUNCH(8, 1);
    else if (cols <= 1024) SM_LAUNCH(16, 1);
    else
        hipLaunchKernelGGL(softmax_long_kernel, grid, block, 0, ctx->stream, rows, cols, x, addend, add_div, add_mod,
                           flush_nan_to_zero, y);
#undef SM_LAUNCH
#undef SM16_LAUNCH
    RTEN_LAUNCH_CHECK(ctx, "softmax_kernel");
    return RTEN_HIP_OK;
}

static int32_t layer_norm_launch(rten_hip_ctx *ctx, int64_t rows, int32_t cols, const float *x, const float *addend, const float *gamma,
                                 const float *beta, float gamma_scalar, float beta_scalar, float epsilon, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (rows < 0 || cols < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (rows == 0 || cols == 0) return RTEN_HIP_OK;
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    const dim3 grid((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(64 * ROWS_PER_BLOCK);
    ProfScope ps(ctx, addend ? "add_layer_norm_f32" : "layer_norm_f32", 0.0, (addend ? 12.0 : 8.0) * rows * cols);
    // rows in sequence per wave once a launch has more than ~2 waves per SIMD to give (y may alias x / addend only in the one-row-per-wave form: a wave of the
    // streaming form requests row k + 1 before it stores row k, which is still safe -- rows are disjoint -- so aliasing is fine there too)
    static const int env_rows = getenv("RTEN_LN_ROWS") ? atoi(getenv("RTEN_LN_ROWS")) : -1; // (tuning: rows per wave; 0 = the one-row form)
    // measured (profiles/r09/layer_norm_rows.txt): [16384, 768] 18.1 -> 15.9 us at 4 rows per wave (0.70 -> 0.79 of 8 TB/s); [4096, 768] LOSES with any R
    // (6.3 -> 6.6 / 7.7 / 8.3 us at 2 / 3 / 4): too few waves left to keep HBM busy -- so only from 4 waves per SIMD of 4-row work upwards
    int rpw = env_rows >= 0 ? env_rows : (rows >= 4 * 4 * 4 * (int64_t)ctx->num_cus ? 4 : 0);
    if (rpw >= 2 && cols <= 1024 && cols > 128) {
        const int64_t waves = (rows + rpw - 1) / rpw;
        const dim3 sgrid((unsigned)((waves + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
#define LNS_LAUNCH(CH) do { if (addend) hipLaunchKernelGGL((layer_norm_stream_kernel<CH, true>), sgrid, block, 0, ctx->stream, rows, cols, x, gamma, beta, gamma_scalar, beta_scalar, epsilon, addend, y, rpw); \
                            else hipLaunchKernelGGL((layer_norm_stream_kernel<CH, false>), sgrid, block, 0, ctx->stream, rows, cols, x, gamma, beta, gamma_scalar, beta_scalar, epsilon, addend, y, rpw); } while (0)
        if (cols <= 256) LNS_LAUNCH(4);
        else if (cols <= 512) LNS_LAUNCH(8);
        else if (cols <= 768) LNS_LAUNCH(12);
        else LNS_LAUNCH(16);
#undef LNS_LAUNCH
        RTEN_LAUNCH_CHECK(ctx, "layer_norm_stream_kernel");
        return RTEN_HIP_OK;
    }
    // (tuning knob: a dynamic-LDS request that caps the workgroups per compute unit -- fewer resident waves let the stores of the first waves overlap the
    //  loads of the later ones instead of queueing behind all of them)
    static const int env_lds = getenv("RTEN_LN_LDS") ? atoi(getenv("RTEN_LN_LDS")) * 1024 : 0;
#define LN_LAUNCH(CH) do { if (env_lds > 64 * 1024) hipFuncSetAttribute((const void *)layer_norm_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, env_lds); \
                           hipLaunchKernelGGL((layer_norm_kernel<CH>), grid, block, env_lds, ctx->stream, rows, cols, x, gamma, beta, gamma_scalar, beta_scalar, epsilon, addend, y); } while (0)
    if (cols <= 128) LN_LAUNCH(2);
    else if (cols <= 256) LN_LAUNCH(4);
    else if (cols <= 512) LN_LAUNCH(8);
    else if (cols <= 768) LN_LAUNCH(12);
    else if (cols <= 1024) LN_LAUNCH(16);
    else LN_LAUNCH(0);
#undef LN_LAUNCH
    RTEN_LAUNCH_CHECK(ctx, "layer_norm_kernel");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_layer_norm_f32(rten_hip_ctx *ctx, int64_t rows, int32_t cols, const float *x,
                                            const float *gamma, const float *beta, float gamma_scalar,
                                            float beta_scalar, float epsilon, float *y) {
    return layer_norm_launch(ctx, rows, cols, x, nullptr, gamma, beta, gamma_scalar, beta_scalar, epsilon, y);
}

// LayerNormalization(x + addend): the residual Add of a transformer block fused into the normalisation that consumes it
// (same f32 add, same reductions: bit-identical to Add followed by LayerNormalization).  `addend` has x's shape.
RTEN_EXPORT int32_t rten_hip_add_layer_norm_f32(rten_hip_ctx *ctx, int64_t rows, int32_t cols, const float *x, const float *addend,
                                                const float *gamma, const float *beta, float gamma_scalar, float beta_scalar,
                                                float epsilon, float *y) {
    if (!addend) return RTEN_HIP_ERR_INVALID_VALUE;
    return layer_norm_launch(ctx, rows, cols, x, addend, gamma, beta, gamma_scalar, beta_scalar, epsilon, y);
}

RTEN_EXPORT int32_t rten_hip_global_average_pool_f32(rten_hip_ctx *ctx, int64_t nc, int32_t inner, const float *x,
                                                     float *y) {
    RTEN_CHECK_CTX(ctx);
    if (nc < 0 || inner <= 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (nc == 0) return RTEN_HIP_OK;
    if (!x || !y) return RTEN_HIP_ERR_INVALID_VALUE;
    const dim3 grid((unsigned)((nc + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block(64 * ROWS_PER_BLOCK);
    ProfScope ps(ctx, "global_average_pool_f32", 0.0, 4.0 * nc * (inner + 1));
    if (inner <= 64)
        hipLaunchKernelGGL(global_avg_pool_rows16_kernel, dim3((unsigned)((nc + 4 * ROWS_PER_BLOCK - 1) / (4 * ROWS_PER_BLOCK))), block, 0, ctx->stream, nc, inner, x, y);
    else
        hipLaunchKernelGGL(global_avg_pool_kernel, grid, block, 0, ctx->stream, nc, inner, x, y);
    RTEN_LAUNCH_CHECK(ctx, "global_avg_pool_kernel");
    return RTEN_HIP_OK;
}

static int32_t reduce_strided(rten_hip_ctx *ctx, bool mean, int32_t n_outer, const int64_t *outer_shape, const int64_t *outer_strides,
                              int32_t n_inner, const int64_t *inner_shape, const int64_t *inner_strides, const float *x, float *y) {
    RTEN_CHECK_CTX(ctx);
    if (n_outer < 0 || n_outer > 6 || n_inner < 0 || n_inner > 6 || (n_outer && (!outer_shape || !outer_strides)) ||
        (n_inner && (!inner_shape || !inner_strides)))
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "reduce_sum: at most 6 kept and 6 reduced dims");
    ReduceArgs p = {};
    p.n_outer = n_outer;
    p.n_inner = n_inner > 0 ? n_inner : 1;
    p.rows = 1;
    int64_t inner = 1;
    for (int d = 0; d < n_outer; d++) {
        if (outer_shape[d] < 0 || outer_shape[d] > 0x7fffffff || outer_strides[d] < 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "reduce_sum: bad dimension");
        p.oshape[d] = (int32_t)outer_shape[d];
        p.ostride[d] = outer_strides[d];
        p.rows *= outer_shape[d];
    }
    p.ishape[0] = 1;
    for (int d = 0; d < n_inner; d++) {
        if (inner_shape[d] < 0 || inner_strides[d] < 0) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "reduce_sum: bad dimension");
        inner *= inner_shape[d];
        if (inner > 0x7fffffff) return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "reduce_sum: reduced slice longer than 2^31 - 1");
        p.ishape[d] = (int32_t)inner_shape[d];
        p.istride[d] = inner_strides[d];
    }
    p.inner = (int)inner;
    p.divisor = mean ? (float)inner : 0.f;
    if (p.rows == 0) return RTEN_HIP_OK;
    if (!y) return RTEN_HIP_ERR_INVALID_VALUE;
    if (inner == 0) { // an empty slice gives the kernel's value for it (reduce.rs:446-452): Sum 0, Mean 0 / 0 = NaN
        RTEN_HIP_TRY(ctx, hipMemsetAsync(y, mean ? 0xff : 0, sizeof(float) * (size_t)p.rows, ctx->stream));
        return RTEN_HIP_OK;
    }
    if (!x) return RTEN_HIP_ERR_INVALID_VALUE;
    const dim3 block(64 * ROWS_PER_BLOCK);
    ProfScope ps(ctx, "reduce_sum_f32", 0.0, 4.0 * p.rows * (inner + 1));
    const dim3 grid16((unsigned)((p.rows + 4 * ROWS_PER_BLOCK - 1) / (4 * ROWS_PER_BLOCK)));
    const int64_t last = n_outer ? p.oshape[n_outer - 1] : 1;
    if (inner > 64 && n_outer && p.ostride[n_outer - 1] == 1 && last >= 16 && p.istride[p.n_inner - 1] > 1)
        hipLaunchKernelGGL(reduce_sum_cols_kernel, dim3((unsigned)(p.rows / last * ((last + 15) / 16))), dim3(1024), 0, ctx->stream, p, x, y);
    else if (inner <= 64) hipLaunchKernelGGL(reduce_sum_rows16_kernel<4>, grid16, block, 0, ctx->stream, p, x, y);
    else if (inner <= 128) hipLaunchKernelGGL(reduce_sum_rows16_kernel<8>, grid16, block, 0, ctx->stream, p, x, y);
    else if (inner <= 256) hipLaunchKernelGGL(reduce_sum_rows16_kernel<16>, grid16, block, 0, ctx->stream, p, x, y);
    else
        hipLaunchKernelGGL(reduce_sum_kernel, dim3((unsigned)((p.rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), block, 0, ctx->stream, p, x, y);
    RTEN_LAUNCH_CHECK(ctx, "reduce_sum_kernel");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_reduce_sum_strided_f32(rten_hip_ctx *ctx, int32_t n_outer, const int64_t *outer_shape, const int64_t *outer_strides,
                                                    int32_t n_inner, const int64_t *inner_shape, const int64_t *inner_strides,
                                                    const float *x, float *y) {
    return reduce_strided(ctx, false, n_outer, outer_shape, outer_strides, n_inner, inner_shape, inner_strides, x, y);
}

RTEN_EXPORT int32_t rten_hip_reduce_mean_strided_f32(rten_hip_ctx *ctx, int32_t n_outer, const int64_t *outer_shape, const int64_t *outer_strides,
                                                     int32_t n_inner, const int64_t *inner_shape, const int64_t *inner_strides,
                                                     const float *x, float *y) {
    return reduce_strided(ctx, true, n_outer, outer_shape, outer_strides, n_inner, inner_shape, inner_strides, x, y);
}
