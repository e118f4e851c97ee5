// Device-side restatement of the element-wise functions of rten-vecmath, operation for operation
// (mul_add -> fused v_fma_f32, everything else individually rounded), so that element-wise
// results are bit-identical to the reference CPU path.  The library is compiled with
// -ffp-contract=off: the only fused operations are the explicit __builtin_fmaf calls below.
//
//   ReducedRangeExp  rten-vecmath/src/exp.rs:140-190
//   Exp              rten-vecmath/src/exp.rs:59-132
//   Erf, Gelu        rten-vecmath/src/erf.rs:21-76
#pragma once
#include <hip/hip_runtime.h>

namespace vm {

__device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

constexpr float INV_LOG2 = 1.44269504088896340736f;
constexpr float ROUNDING_MAGIC = 12582912.f;
constexpr float LOG2_HI = -6.93145752e-1f;
constexpr float LOG2_LO = -1.42860677e-6f;
constexpr float P0 = 1.0f, P1 = 1.0f, P2 = 4.99999851e-1f, P3 = 1.66664720e-1f, P4 = 4.16695364e-2f,
                P5 = 8.37312452e-3f, P6 = 1.37805939e-3f;
// -126.5 * ln2 + 0.01 evaluated in f32 (exp.rs:131)
constexpr float EXP_LOWER_CUTOFF = -126.5f * 0.693147180559945309417f + 0.01f;

__device__ __forceinline__ float exp_core(float x, float &j_out) {
    float j = fma(x, INV_LOG2, ROUNDING_MAGIC);
    j = j - ROUNDING_MAGIC;
    float r = fma(j, LOG2_HI, x);
    r = fma(j, LOG2_LO, r);
    float t = P6;
    t = fma(t, r, P5);
    t = fma(t, r, P4);
    t = fma(t, r, P3);
    t = fma(t, r, P2);
    t = fma(t, r, P1);
    r = fma(t, r, P0);
    j_out = j;
    return r;
}

// exp(x) for x <= 0 (exp.rs:140-190)
__device__ __forceinline__ float exp_reduced(float x) {
    float j;
    float r = exp_core(x, j);
    int k = (int)j;
    float p2 = __int_as_float((int)((unsigned)(k + 127) << 23));
    r = r * p2;
    return x < EXP_LOWER_CUTOFF ? 0.f : r;
}

// exp(x), full range (exp.rs:59-132)
__device__ __forceinline__ float exp_full(float x) {
    float j;
    float r = exp_core(x, j);
    int k = (int)j;
    unsigned ia = k > 0 ? 0u : 0x83000000u;
    float s = __int_as_float((int)(ia + 0x7f000000u));
    float t = __int_as_float((int)(((unsigned)k << 23) - ia));
    r = r * s;
    r = r * t;
    if (x >= 104.0f) r = __builtin_inff();
    if (x <= -104.0f) r = 0.f;
    return r;
}

// erf(x) (erf.rs:21-59): Abramowitz-Stegun 7.1.26
__device__ __forceinline__ float erf(float x0) {
    const bool neg = x0 < 0.f;
    const float x = neg ? (0.f - x0) : x0;
    const float p = 0.3275911f;
    const float a0 = 0.254829592f, a1 = -0.284496736f, a2 = 1.421413741f, a3 = -1.453152027f, a4 = 1.061405429f;
    const float t = 1.0f / fma(x, p, 1.0f); // IEEE division (ops.reciprocal = div(1, x), rten-simd/src/ops.rs:639-641)
    float y = a4;
    y = fma(y, t, a3);
    y = fma(y, t, a2);
    y = fma(y, t, a1);
    y = fma(y, t, a0);
    const float at = y * t;
    const float xm2 = 0.f - (x * x);
    const float e = exp_reduced(xm2);
    const float r = 1.0f - at * e;
    return neg ? (0.f - r) : r;
}

constexpr float SQRT_2_RCP = 0.70710678118654752440f;

// gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) (erf.rs:61-76)
__device__ __forceinline__ float gelu(float x) {
    const float half_x = x * 0.5f;
    float y = x * SQRT_2_RCP;
    y = erf(y) + 1.0f;
    return half_x * y;
}

// tanh(x) (rten-vecmath/src/tanh.rs:12-72): odd polynomial for |x| <= 0.55, (exp(2|x|) - 1) / (exp(2|x|) + 1) above it, 1 from 9.02, |x| itself up to
// 0.0004; the sign is flipped where x <= 0 (so tanh(+0) = -0, as the reference's select on `le(x, 0)` gives)
__device__ __forceinline__ float tanh(float x) {
    const float ax = __builtin_fabsf(x);
    const float p1 = 0.999999940395355224609375f, p3 = -0.33332359790802001953125f, p5 = 0.13310669362545013427734375f,
                p7 = -5.21197654306888580322265625e-2f, p9 = 1.5497927553951740264892578125e-2f;
    const float x2 = x * x;
    float ys = fma(p9, x2, p7);
    ys = fma(ys, x2, p5);
    ys = fma(ys, x2, p3);
    ys = fma(ys, x2, p1);
    ys = ys * ax;
    const float e = exp_full(ax * 2.0f);
    const float ym = (e - 1.0f) / (e + 1.0f);
    float y = ax >= 9.02f ? 1.0f : ym;
    y = ax <= 0.55f ? ys : y;
    y = ax <= 0.0004f ? ax : y;
    return x <= 0.f ? __int_as_float(__float_as_int(y) ^ (int)0x80000000) : y; // neg = sign-bit xor on the x86 SIMD back ends (rten-simd/src/arch/x86_64/avx512.rs:319-321)
}

// Relu: f32::max(x, 0) -- NaN -> 0 (unary_elementwise.rs:611-613)
__device__ __forceinline__ float relu(float x) { return fmaxf(x, 0.f); }

} // namespace vm
