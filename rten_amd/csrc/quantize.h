// Device pieces of DynamicQuantizeLinear shared by quantize.hip and the fused quantize-and-stage kernel of
// int8_fast.hip (src/ops/quantize.rs:352-436, rten-vecmath/src/min_max.rs:20-44, quantize.rs:39-79).
#pragma once
#include "internal.h"

typedef float dq_f32x4 __attribute__((ext_vector_type(4)));

namespace dql {

// order-preserving float <-> uint mapping for atomic min/max
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

struct QParams { float scale, inv_scale; int zp; };

// quantize.rs:411-419 (scalar algebra) -- identical operation sequence
__device__ __forceinline__ QParams dql_params(float x_min, float x_max) {
    const float x_min_adj = fminf(x_min, 0.f);
    const float x_max_adj = fmaxf(x_max, 0.f);
    const float range = x_max_adj - x_min_adj;
    const float scale = range / 255.f;
    const float min_scaled = x_min_adj / scale;
    const float init_zp = 0.f - min_scaled;
    const float clipped = init_zp < 0.f ? 0.f : (init_zp > 255.f ? 255.f : init_zp); // f32::clamp keeps NaN
    const float rounded = rintf(clipped);                                            // round_ties_even
    int zp = 0;
    if (rounded == rounded) zp = (int)(rounded < 0.f ? 0.f : (rounded > 255.f ? 255.f : rounded)); // saturating cast
    QParams q;
    q.scale = scale;
    q.inv_scale = 1.f / scale; // quantize.rs:210
    q.zp = zp;
    return q;
}

// vecmath/quantize.rs:57-62: to_int_round (cvtps2dq: NaN / out of range -> i32::MIN), + zp, saturate to u8
// Same function of (x, inv_scale, zp) for every input, in 8 VALU operations (the quantize-and-stage sweep is ~36 % of the int8
// ResNet-50 step and was VALU heavy): the conversion result only matters inside [-zp, 255 - zp], so p is clamped to +-1024
// first -- fmaxf drops a NaN onto the lower bound, whose code is 0 like cvtps2dq's INT_MIN -- and the one case where
// saturation and the x86 "integer indefinite" disagree, p >= 2^31 (INT_MIN -> code 0, not 255), is sent to the lower bound too.
__device__ __forceinline__ unsigned quant_u8(float x, float inv_scale, int zp) {
    const float p = x * inv_scale;
    float pc = fminf(fmaxf(p, -1024.f), 1024.f);
    pc = p >= 2147483648.f ? -1024.f : pc;
    const int t = (int)rintf(pc) + zp; // |pc| <= 1024: exact, no overflow
    return (unsigned)(t < 0 ? 0 : (t > 255 ? 255 : t));
}


// Second stage of the min/max reduction: every workgroup of the consuming kernel folds the per-workgroup partials
// the sweep left in `ws` (pairs {min, max}; min/max are order independent, so this equals the one-pass result).
// Must be called by all 256 threads of the workgroup.
__device__ __forceinline__ void block_minmax(const float *__restrict__ ws, int nparts, float &mn, float &mx) {
    __shared__ float s_mn[4], s_mx[4];
    float a = __builtin_inff(), b = -__builtin_inff();
    for (int i = threadIdx.x; i < nparts; i += 256) { a = fminf(ws[2 * i], a); b = fmaxf(ws[2 * i + 1], b); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = a; s_mx[threadIdx.x >> 6] = b; }
    __syncthreads();
    mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
}

// Min/max statistics accumulated by a producing kernel's epilogue: kStatSlots ordered-uint minima followed by kStatSlots
// ordered-uint maxima (reset to 0xffffffff / 0 by rten_hip_minmax_stats_reset; producers hit slot = id % kStatSlots).
constexpr int kStatSlots = 256;
__device__ __forceinline__ void block_minmax_slots(const unsigned *__restrict__ stats, float &mn, float &mx) {
    __shared__ float s_mn[4], s_mx[4];
    float a = __builtin_inff(), b = -__builtin_inff();
    if (threadIdx.x < kStatSlots) { a = ord2f(stats[threadIdx.x]); b = ord2f(stats[kStatSlots + threadIdx.x]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = a; s_mx[threadIdx.x >> 6] = b; }
    __syncthreads();
    mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
}

} // namespace dql

// Enqueues the min/max sweep of x[0..n): one {min, max} pair per workgroup at the start of the context scratch (no
// atomics, nothing to initialise); returns the device pointer and the number of pairs, or nullptr on allocation
// failure.  The consumer folds the pairs with dql::block_minmax.  (quantize.hip)
const float *rten_dql_minmax(rten_hip_ctx *ctx, int64_t n, const float *x, int *nparts);
