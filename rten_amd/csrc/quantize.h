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
__device__ __forceinline__ unsigned quant_u8(float x, float inv_scale, int zp) {
    const float p = x * inv_scale;
    int q;
    if (!(p == p) || p >= 2147483648.f || p < -2147483648.f) q = (int)0x80000000;
    else q = (int)rintf(p);
    long long t = (long long)q + zp;
    t = t < 0 ? 0 : (t > 255 ? 255 : t);
    return (unsigned)t;
}


} // namespace dql

// Enqueues the min/max sweep of x[0..n) into the first two words of the context scratch (ordered-uint encoding);
// returns the device pointer to the two words, or nullptr on allocation failure.  (quantize.hip)
unsigned *rten_dql_minmax(rten_hip_ctx *ctx, int64_t n, const float *x);
