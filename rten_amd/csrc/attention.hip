// Scaled dot-product attention: sdpa_head / sdpa_multi_head, src/ops/attention.rs:518-626.
//
//     out[b,h] = softmax(scale * Q[b,h] K[b,h]^T + mask[b]) V[b,h]
//
// Same three steps as the reference (gemm with alpha = scale, row softmax with NaN flush, gemm), but
// batched over all (batch, head) pairs: two batched MFMA GEMM launches + one wave-per-row softmax
// launch, with the [B*H, S, T] score tensor kept in a device scratch buffer (L2/MALL resident at BERT
// sizes: 32*12*128*128*4 B = 25 MB).  Q/K/V/out are addressed through (batch, head, row) strides, so
// BERT's [B*S, H*D] projection outputs are consumed and produced in place -- the Reshape/Transpose
// nodes of the ONNX graph become stride arithmetic, as TransposeFusion does on the CPU
// (src/optimize/fusions.rs:1066).  Each step reuses the bit-exact GEMM / softmax kernels, so the result
// is bit-identical to the oracle's sdpa.
#include "internal.h"

// attention_fused.hip: single-kernel path (head size 32 / 64 / 128, key length <= 512; without `force` only where it is the faster form);
// RTEN_HIP_ERR_UNSUPPORTED = not covered
int32_t rten_sdpa_fused(rten_hip_ctx *ctx, const rten_hip_sdpa_desc *d, const float *q, const float *k, const float *v, const float *mask, float *out, bool force);

// 0 = automatic (the one-kernel form where it covers the shape and is the faster one: <= 128 keys), 1 = composed path only (GEMM, softmax,
// GEMM), 2 = the one-kernel form wherever it covers the shape (head 32 / 64 / 128, <= 512 keys).
RTEN_EXPORT int32_t rten_hip_set_sdpa_path(rten_hip_ctx *ctx, int32_t mode) {
    RTEN_CHECK_CTX(ctx);
    if (mode < 0 || mode > 2) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "set_sdpa_path: mode must be 0, 1 or 2");
    ctx->sdpa_path = mode;
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_sdpa_f32(rten_hip_ctx *ctx, const rten_hip_sdpa_desc *d, const float *q, const float *k,
                                      const float *v, const float *mask, float *out) {
    RTEN_CHECK_CTX(ctx);
    if (!d) return RTEN_HIP_ERR_INVALID_VALUE;
    if (d->batch < 0 || d->heads < 0 || d->s < 0 || d->t < 0 || d->d < 0 || d->dv < 0) return RTEN_HIP_ERR_INVALID_VALUE;
    const long long bh = (long long)d->batch * d->heads;
    if (bh == 0 || d->s == 0 || d->dv == 0) return RTEN_HIP_OK;
    if (!q || !k || !v || !out) return RTEN_HIP_ERR_INVALID_VALUE;
    // [B,1,S,T] contiguous, [B,1,1,T] contiguous, or -- row stride 0 with batch stride S * T -- the FIRST row of each item of a [B,1,S,T] tensor whose S rows
    // are known to be equal (an exporter's expanded padding mask: rten_hip_ops.hpp, Tensor::uniform_dims)
    if (mask && ((d->mask_row_stride != 0 && d->mask_row_stride != d->t) ||
                 (d->mask_row_stride ? d->mask_batch_stride != (int64_t)d->s * d->t
                                     : (d->mask_batch_stride != (int64_t)d->t && d->mask_batch_stride != (int64_t)d->s * d->t))))
        return rten_set_error(ctx, RTEN_HIP_ERR_UNSUPPORTED, "sdpa: mask must be [B,1,1,T] or [B,1,S,T] contiguous");
    // ONE query row: both of sdpa_head's products are one-row products of unpacked operands, for which the reference's gemm_impl takes its vector-matrix
    // kernels (rten-gemm/src/lib.rs:876-891) -- not the blocked chain the one-kernel forms replay.  The composed path below then calls the GEMM entry
    // that makes the same choice (rten_hip_set_gemv_order(ctx, 0, ..) restores the blocked order everywhere, e.g. for prepacked weights).
    const bool one_row = d->s == 1 && ctx->gemv_order != 0;
    if (ctx->sdpa_path != 1 && !one_row) {
        const int32_t rc = rten_sdpa_fused(ctx, d, q, k, v, mask, out, ctx->sdpa_path == 2);
        if (rc != RTEN_HIP_ERR_UNSUPPORTED) return rc;
    }
    const size_t score_bytes = (size_t)bh * d->s * (size_t)d->t * sizeof(float);
    // (the auxiliary scratch: the GEMMs called below own `scratch` -- a PV product over more than 256 keys may park split-K slabs there,
    // which used to overwrite the scores it was reading: wrong results for 256 < T < 512 on the composed path until round 3)
    float *scores = (float *)rten_aux_scratch(ctx, score_bytes + 256);
    if (!scores) return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "sdpa: scratch allocation failed (warm up before capture)");
    scores += 64;

    rten_hip_gemm_desc g = {};
    // scores = scale * Q K^T   (attention.rs:533-544)
    g.m = d->s; g.n = d->t; g.k = d->d;
    g.a_rs = d->q_rs; g.a_cs = 1; g.b_rs = 1; g.b_cs = d->k_rs; g.ldc = d->t;
    g.batch = (int32_t)bh; g.batch_inner = d->heads;
    g.a_bs = d->q_bs; g.a_bsi = d->q_hs; g.b_bs = d->k_bs; g.b_bsi = d->k_hs;
    g.c_bs = (int64_t)d->heads * d->s * d->t; g.c_bsi = (int64_t)d->s * d->t;
    g.alpha = d->scale; g.beta = 0.f;
    int32_t rc = one_row ? rten_hip_gemm_f32(ctx, &g, q, k, nullptr, scores) : rten_gemm_f32_blocked(ctx, &g, q, k, nullptr, scores);
    if (rc) return rc;
    // row softmax with NaN flush (attention.rs:546-552); score row r = ((b*H + h)*S + qi)
    if (d->t > 0) {
        if (mask && d->mask_row_stride) { // [B,1,S,T]: addend row = b*S + qi -> one launch per image
            for (int b = 0; b < d->batch; b++) {
                float *sb = scores + (long long)b * d->heads * d->s * d->t;
                rc = rten_hip_softmax_f32(ctx, (int64_t)d->heads * d->s, d->t, sb, mask + (long long)b * d->mask_batch_stride, 1,
                                          d->s, d->flush_nan_to_zero, sb);
                if (rc) return rc;
            }
        } else if (mask && d->mask_batch_stride != d->t) { // one shared row per item, S * T apart: one launch per image
            for (int b = 0; b < d->batch; b++) {
                float *sb = scores + (long long)b * d->heads * d->s * d->t;
                rc = rten_hip_softmax_f32(ctx, (int64_t)d->heads * d->s, d->t, sb, mask + (long long)b * d->mask_batch_stride, (int64_t)d->heads * d->s, 1,
                                          d->flush_nan_to_zero, sb);
                if (rc) return rc;
            }
        } else { // no mask, or [B,1,1,T]: addend row = r / (H*S)
            rc = rten_hip_softmax_f32(ctx, bh * d->s, d->t, scores, mask, (int64_t)d->heads * d->s, d->batch, d->flush_nan_to_zero, scores);
            if (rc) return rc;
        }
    }
    // out = P V   (attention.rs:554-561)
    g = {};
    g.m = d->s; g.n = d->dv; g.k = d->t;
    g.a_rs = d->t; g.a_cs = 1; g.b_rs = d->v_rs; g.b_cs = 1; g.ldc = d->o_rs;
    g.batch = (int32_t)bh; g.batch_inner = d->heads;
    g.a_bs = (int64_t)d->heads * d->s * d->t; g.a_bsi = (int64_t)d->s * d->t;
    g.b_bs = d->v_bs; g.b_bsi = d->v_hs; g.c_bs = d->o_bs; g.c_bsi = d->o_hs;
    g.alpha = 1.f; g.beta = 0.f;
    return one_row ? rten_hip_gemm_f32(ctx, &g, scores, v, nullptr, out) : rten_gemm_f32_blocked(ctx, &g, scores, v, nullptr, out);
}
