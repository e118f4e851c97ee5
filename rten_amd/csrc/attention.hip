// Scaled dot-product attention: sdpa_head / sdpa_multi_head, src/ops/attention.rs:518-626.
//
//     out[i] = softmax(scale * Q[i] K[i]^T + mask) V[i]          for every (batch, head) i
//
// Same three steps as the reference (gemm with alpha = scale, row softmax with NaN flush, gemm), but
// batched over all (batch, head) pairs: two batched MFMA GEMM launches + one wave-per-row softmax
// launch, with the [bh, s, t] score tensor kept in a device scratch buffer (L2/MALL resident at
// BERT sizes: 32*12*128*128*4 B = 25 MB).  Because each step reuses the bit-exact GEMM / softmax
// kernels, the result is bit-identical to the oracle's sdpa.
#include "internal.h"

RTEN_EXPORT int32_t rten_hip_sdpa_f32(rten_hip_ctx *ctx, int32_t bh, int32_t s, int32_t t, int32_t d, int32_t dv,
                                      const float *q, const float *k, const float *v, const float *mask,
                                      int32_t mask_bh_div, int64_t mask_batch_stride, int64_t mask_row_stride,
                                      float scale, float *out) {
    RTEN_CHECK_CTX(ctx);
    if (bh < 0 || s < 0 || t < 0 || d < 0 || dv < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    if (bh == 0 || s == 0 || dv == 0) return RTEN_HIP_OK;
    if (!q || !k || !v || !out) return RTEN_HIP_ERR_INVALID_VALUE;
    if (mask && (mask_bh_div <= 0 || (mask_row_stride != 0 && mask_row_stride != t) ||
                 (mask_batch_stride != (mask_row_stride ? (int64_t)s * t : (int64_t)t))))
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "sdpa: mask must be [B,1,1,T] or [B,1,S,T] contiguous");
    const size_t score_bytes = (size_t)bh * s * (size_t)t * sizeof(float);
    float *scores = (float *)rten_scratch(ctx, score_bytes + 256);
    if (!scores) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "sdpa: scratch allocation failed (warm up before capture)");
    scores += 64; // first 256 B of the scratch are reserved for the DQL workspace

    rten_hip_gemm_desc g = {};
    // scores = scale * Q K^T   (attention.rs:533-544)
    g.m = s; g.n = t; g.k = d;
    g.a_rs = d; g.a_cs = 1; g.b_rs = 1; g.b_cs = d; g.ldc = t;
    g.batch = bh; g.a_bs = (int64_t)s * d; g.b_bs = (int64_t)t * d; g.c_bs = (int64_t)s * t;
    g.alpha = scale; g.beta = 0.f;
    int32_t rc = rten_hip_gemm_f32(ctx, &g, q, k, nullptr, scores);
    if (rc) return rc;
    // softmax(score_mod(row)) with NaN flush (attention.rs:546-552).  Row r = (i*s + qi).
    if (t > 0) {
        int64_t add_div = 1, add_mod = 1;
        if (mask) {
            if (mask_row_stride) { // [B,1,S,T]: mask row = (i / div) * s + qi  -> not a pure div/mod of r unless div == 1
                if (mask_bh_div == 1) { add_div = 1; add_mod = (int64_t)bh * s; }
                else {
                    // expand per head: handled by launching one softmax per batch entry
                    const int heads = mask_bh_div, batches = bh / heads;
                    for (int b = 0; b < batches; b++) {
                        rc = rten_hip_softmax_f32(ctx, (int64_t)heads * s, t, scores + (int64_t)b * heads * s * t,
                                                  mask + (int64_t)b * mask_batch_stride, 1, s, 1,
                                                  scores + (int64_t)b * heads * s * t);
                        if (rc) return rc;
                    }
                    goto pv;
                }
            } else { // [B,1,1,T]: mask row = r / (heads*s)
                add_div = (int64_t)mask_bh_div * s;
                add_mod = bh / mask_bh_div;
            }
        }
        rc = rten_hip_softmax_f32(ctx, (int64_t)bh * s, t, scores, mask, add_div, add_mod, 1, scores);
        if (rc) return rc;
    }
pv:
    // out = P V   (attention.rs:554-561)
    g = {};
    g.m = s; g.n = dv; g.k = t;
    g.a_rs = t; g.a_cs = 1; g.b_rs = dv; g.b_cs = 1; g.ldc = dv;
    g.batch = bh; g.a_bs = (int64_t)s * t; g.b_bs = (int64_t)t * dv; g.c_bs = (int64_t)s * dv;
    g.alpha = 1.f; g.beta = 0.f;
    return rten_hip_gemm_f32(ctx, &g, scores, v, nullptr, out);
}
