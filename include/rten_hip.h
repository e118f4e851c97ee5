/*
 * rten_hip.h -- C ABI of the MI355X (gfx950) operator backend for RTen.
 *
 * This is the drop-in boundary: the entry points a Rust `Operator` implementation in RTen binds
 * through `extern "C"` to replace the rten-gemm / rten-vecmath CPU micro-kernels behind
 * src/ops/{matmul,conv,attention,norm,pooling,quantize}.  Each declaration cites the reference
 * interface it replaces (paths relative to the robertknight/rten v0.25.0 checkout); the Rust-side
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - Plain C types only.  Every tensor pointer is a DEVICE pointer unless the parameter name ends
 *    in `_host`.  Tensors are contiguous row-major unless strides are given explicitly (strides
 *    are in ELEMENTS, as in rten-tensor layouts).
 *  - Every function returns an int32 status (0 = ok).  Validation that the reference performs in
 *    Rust before touching data (shape checks, error strings) stays on the host side of the ABI;
 *    the ABI reports RTEN_HIP_ERR_INVALID_VALUE for arguments it cannot execute and
 *    RTEN_HIP_ERR_HIP for runtime failures, with text in rten_hip_last_error().
 *  - All work is enqueued on the context's HIP stream and is asynchronous w.r.t. the host;
 *    rten_hip_sync() (or a D2H copy) makes results visible.
 *  - Thread safety (Model::run(&self) may be entered by several host threads at once, src/model.rs:308-550,
 *    call site src/graph.rs:782): a context MAY be shared.  Every entry point that takes a context locks the
 *    context's internal (recursive) mutex for the duration of the call, binds the context's device to the
 *    calling thread, and enqueues all launches of one operator back to back, so concurrent callers
 *    interleave at operator granularity on the one stream and the shared scratch / staging buffers are
 *    reused in stream order.  rten_hip_last_error() is per calling thread.  A hipGraph capture
 *    (rten_hip_graph_begin .. rten_hip_graph_end) holds the lock: other threads' calls on that context
 *    block until the capture ends.  Tuning knobs (set_gemm_variant_override, set_gemm_split, ...) are
 *    per context, not per thread: callers that tune concurrently use one context per thread.
 *  - Numerics: integer paths are bit-exact w.r.t. the reference.  f32 GEMM/conv reproduce the
 *    reference's accumulation order exactly (k-ordered FMA chains in depth blocks of 256,
 *    rten-gemm/src/lib.rs:630-633 + kernels/simd_generic.rs:326-414), element-wise kernels
 *    (Gelu/Erf/Relu/Add/cast_scale/DQL) are operation-for-operation restatements; reductions
 *    (softmax sum, LayerNorm mean/variance, GlobalAveragePool) reproduce the reference's 16-lane
 *    (AVX-512) partial-sum order bit for bit and differ from its other ISA widths only by
 *    summation-order rounding.  One-row products (M == 1) of rten_hip_gemm_f32 follow the reference's
 *    vector-matrix kernels (rten-gemm/src/lib.rs:668-747,876-891; kernels/simd_generic.rs:14-197) bit for bit
 *    under the thread-count assumption stated by rten_hip_set_gemv_order (the reference's own result depends on
 *    its thread count there).  So do the one-row products inside composite operators whose reference code calls gemm on unpacked
 *    operands: sdpa with ONE query row (src/ops/attention.rs:518-562) and ConvTranspose with a one-row kernel matrix
 *    (O_g = kh = kw = 1, src/ops/conv_transpose.rs:376-383); MatMulNBits' right-hand side is block-quantized, not `Unpacked`, and keeps
 *    the blocked order for one row too (rten-gemm/src/lib.rs:876).
 */
#ifndef RTEN_HIP_H
#define RTEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v8 (round 6, third part): + rten_hip_conv2d_f32_pair / _pair_supported (an expand layer and the next block's reduce layer in one launch; a launch plan
 * lists the pairs: "pairs"), rten_hip_conv2d_f32_pair_shortcut / _supported (the same with the block's shortcut convolution computed in the launch: "pair_shortcuts");
 * GEMM variant 32 (the direct stem kernel); rten_hip_model_plan_json also names the load-time lists.
 * v7 (round 6, second part): + rten_hip_set_int8_tile (the int8 kernels' workgroup tile as a context knob: a launch plan may carry a per-layer entry for an int8
 * convolution step); rten_hip_sdpa_desc: mask_row_stride = 0 with mask_batch_stride = S * T reads one shared row per batch item out of an expanded mask; plan files
 * may key MatMul-family entries by product shape ("shapes"); convolutions with C <= 4 stage their input in the few-channel packed form (opaque layouts only).
 * v6 (round 6): + rten_hip_elementwise_nd (Cast / Not / And / Or / Xor / Equal / Less.. / Where / integer arithmetic over strided operands),
 * rten_hip_gather_axis_b32, rten_hip_copy_rows_b32, rten_hip_tanh_f32, rten_hip_capture_active -- the layout / logic operators an exporter-written
 * transformer graph carries around its hot-path operators (src/ops/convert.rs, binary_elementwise.rs, gather.rs, concat.rs); rten_hip_model_clone
 * refuses a model whose plan lists quantized-output edges; rten_hip_conv2d_int8_qout with sync == NULL = the recompute form (plan key "qout2");
 * measurement-only paths (rten_hip_set_gemm_order bit 3, RTEN_HIP_DEBUG bits 24-31) exist in -DRTEN_ABLATION builds only; rten_hip_tuning_restore puts
 * every knob back even when one setter objects.
 * v5 (round 5b): + rten_hip_model_input_dtype / rten_hip_model_output_dtype (a host that moves the inputs / outputs of a resident subgraph must know
 * their element types: BERT-class graphs take integer inputs).
 * v4 (round 5): + rten_hip_device_id, rten_hip_tuning_save / _restore (a library-level caller on a borrowed context puts the owner's knobs back),
 * rten_hip_model_load_ex (device taken from the context; RTEN_HIP_MODEL_RECEIVE_WEIGHTS), rten_hip_model_load_error, rten_hip_model_clone (replicas that
 * share one weight set: lanes), rten_hip_model_plan_json, rten_hip_model_profile, rten_hip_model_weight_arena (one
 * allocation for every constant of a model: the unit of the one-time RCCL broadcast), plan files may carry {"fused_dql": [...]}, "qout" edges may have
 * several scale products, rten_hip_model_load validates device_id, rten_hip_grid_sync_reset refuses to run inside a capture, scratch buffers a live
 * hipGraph replays from are retired instead of freed, rten_hip_set_gemm_order bit 3 (relaxed split-K: one partial per K group -- NOT bit-exact, tuning /
 * measurement only).
 * v3 (round 4): + rten_hip_model_* (the plan executor behind the C ABI), rten_hip_set_gemm_order bits 4-6 (occupancy cap),
 * GEMM variants 24-30 (one wave per tile, two-stage ring, image patches), sticky device fault reported by rten_hip_sync / rten_hip_graph_launch.  v2 (round 3) had added
 * rten_hip_graph_abort, rten_hip_conv2d_int8_qout, rten_hip_grid_sync_*, rten_hip_dynamic_quantize_linear_staged_products, rten_hip_max_pool2d_f32_stats and
 * rten_hip_set_sdpa_path mode 2 WITHOUT a bump: a binding built against this header must refuse a library whose rten_hip_abi_version() differs. */
#define RTEN_HIP_ABI_VERSION 8

/* ---- status codes (map onto OpError variants, src/operator.rs:116-144) ---- */
#define RTEN_HIP_OK 0
#define RTEN_HIP_ERR_INVALID_VALUE 1 /* -> OpError::InvalidValue */
#define RTEN_HIP_ERR_INCOMPATIBLE_SHAPES 2 /* -> OpError::IncompatibleInputShapes */
#define RTEN_HIP_ERR_UNSUPPORTED 3 /* -> OpError::UnsupportedValue */
#define RTEN_HIP_ERR_HIP 4 /* HIP runtime error; see rten_hip_last_error */
#define RTEN_HIP_ERR_NO_DEVICE 5 /* no gfx950 device / extension unusable: callers must fail loudly */

typedef struct rten_hip_ctx rten_hip_ctx;

/* ---- context, memory, stream plumbing (new: the reference has no device boundary) ---- */

/* Create a context on `device_id`.  `external_stream` may be NULL (the context creates its own
 * stream) or a hipStream_t owned by the caller (e.g. torch's current stream). */
int32_t rten_hip_init(int32_t device_id, void *external_stream, rten_hip_ctx **out_ctx);
/* Destroys the context.  Same-thread rule: a context with an ACTIVE graph capture (rten_hip_graph_begin without _end / _abort) can only be destroyed
 * by the thread that began the capture (the capture is aborted first); from any other thread the call returns RTEN_HIP_ERR_INVALID_VALUE and destroys
 * nothing -- end or abort the capture on its own thread, then destroy (a Rust `Drop` that may run anywhere must do the same). */
int32_t rten_hip_destroy(rten_hip_ctx *ctx);
const char *rten_hip_last_error(rten_hip_ctx *ctx);
int32_t rten_hip_abi_version(void);
/* Waits for the context's stream.  Also reports the context's sticky device fault (a kernel that had to give up -- see rten_hip_conv2d_int8_qout):
 * RTEN_HIP_ERR_HIP on every call until rten_hip_grid_sync_reset. */
int32_t rten_hip_sync(rten_hip_ctx *ctx);
/* Device properties used by the measurement harness. */
/* the device the context was created on (-1: NULL context) */
int32_t rten_hip_device_id(const rten_hip_ctx *ctx);
int32_t rten_hip_device_info(rten_hip_ctx *ctx, char *name_buf, int32_t name_len, int32_t *compute_units,
                             int32_t *clock_mhz, int64_t *total_mem_bytes);

int32_t rten_hip_malloc(rten_hip_ctx *ctx, size_t bytes, void **out_dptr);
int32_t rten_hip_free(rten_hip_ctx *ctx, void *dptr);
int32_t rten_hip_memcpy_h2d(rten_hip_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int32_t rten_hip_memcpy_d2h(rten_hip_ctx *ctx, void *dst_host, const void *src, size_t bytes); /* syncs */
int32_t rten_hip_memcpy_d2d(rten_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int32_t rten_hip_memset(rten_hip_ctx *ctx, void *dst, int32_t byte_value, size_t bytes);

/* Event timers on the context's stream (slot in [0, 64)). */
int32_t rten_hip_timer_start(rten_hip_ctx *ctx, int32_t slot);
int32_t rten_hip_timer_stop(rten_hip_ctx *ctx, int32_t slot);
int32_t rten_hip_timer_elapsed_ms(rten_hip_ctx *ctx, int32_t slot, float *out_ms); /* syncs on stop event */

/* hipGraph capture of a launch sequence (the executor's per-run plan, src/graph.rs:880-1286). */
int32_t rten_hip_graph_begin(rten_hip_ctx *ctx);
int32_t rten_hip_graph_end(rten_hip_ctx *ctx, uint64_t *out_graph);
/* Abandons an active capture (an operator failed while capturing): ends it, drops what was recorded, releases the capture lock.
 * graph_begin / graph_end / graph_abort must be called by the same host thread; graph_end ends the capture on every path. */
int32_t rten_hip_graph_abort(rten_hip_ctx *ctx);
int32_t rten_hip_graph_launch(rten_hip_ctx *ctx, uint64_t graph);
int32_t rten_hip_graph_destroy(rten_hip_ctx *ctx, uint64_t graph);

/* Order two contexts (streams) of one device: work enqueued on `waiter` after the call runs after everything
 * enqueued on `signaler` so far.  Independent operators (e.g. a residual block's downsample branch) can run on a
 * second context concurrently; inside a capture the second context becomes a parallel branch of the same graph and
 * must be joined back (rten_hip_stream_wait(capturing_ctx, side_ctx)) before rten_hip_graph_end. */
int32_t rten_hip_stream_wait(rten_hip_ctx *waiter, rten_hip_ctx *signaler);

/* Per-kernel-class profiling: when enabled, every conv/gemm launch is bracketed by events and its
 * time and algorithmic work are accumulated per kernel name (Profiler, src/timing.rs).  Not usable
 * during graph capture. */
int32_t rten_hip_profile_enable(rten_hip_ctx *ctx, int32_t on);
int32_t rten_hip_profile_reset(rten_hip_ctx *ctx);
/* Writes a JSON array [{"kernel":..,"launches":..,"ms":..,"flops":..,"bytes":..},...] (NUL-terminated). */
int32_t rten_hip_profile_report(rten_hip_ctx *ctx, char *buf, int32_t buf_len);

/* ---- shapes: calc_output_size_and_padding, src/ops/pooling.rs:139-159 (shared by conv + pool) ----
 * same_padding != 0 selects Padding::Same (SAME_UPPER); otherwise pads = top,left,bottom,right.
 * Returns RTEN_HIP_ERR_INVALID_VALUE with the reference's message in *err_msg (static string). */
int32_t rten_hip_calc_output_size_and_padding(int32_t in_h, int32_t in_w, int32_t k_h, int32_t k_w,
                                              int32_t stride_h, int32_t stride_w, int32_t same_padding,
                                              const int32_t pads[4], int32_t dil_h, int32_t dil_w,
                                              int32_t ceil_mode, int32_t out_hw[2], int32_t out_pads[4],
                                              const char **err_msg);

/* ---- f32 GEMM: GemmExecutor::gemm / gemm_uninit / batched_gemm_uninit,
 *      rten-gemm/src/lib.rs:255-372 (called from src/ops/matmul.rs:32-104,208-385) ---- */
#define RTEN_HIP_BIAS_NONE 0
#define RTEN_HIP_BIAS_PER_ROW 1 /* BiasVector::Column: bias[m] */
#define RTEN_HIP_BIAS_PER_COL 2 /* BiasVector::Row:    bias[n] */
#define RTEN_HIP_ACT_NONE 0
#define RTEN_HIP_ACT_RELU 1 /* Relu, src/ops/unary_elementwise.rs:611-613 */
#define RTEN_HIP_ACT_GELU 2 /* Gelu, rten-vecmath/src/erf.rs:61-76 */

typedef struct {
    int32_t m, n, k;
    int64_t a_rs, a_cs; /* A[m,k] element strides (transposes are strides, matmul.rs:47-48) */
    int64_t b_rs, b_cs; /* B[k,n] element strides */
    int64_t ldc;        /* C row stride; C columns are contiguous */
    int32_t batch;      /* >= 1; batched_gemm_uninit */
    int64_t a_bs, b_bs, c_bs; /* batch strides in elements; 0 broadcasts that operand */
    /* optional second (inner) batch level: batch index z -> (z / batch_inner, z % batch_inner), e.g.
     * (image, head) for attention on [B,S,H,D] tensors without materialising transposes
     * (TransformInputs / TransposeFusion, src/optimize/fusions.rs:1066).  batch_inner <= 1 disables it. */
    int32_t batch_inner;
    int64_t a_bsi, b_bsi, c_bsi;
    float alpha, beta;  /* C = alpha*A.B + beta*C ; beta == 0 => C is never read */
    int32_t bias_kind;  /* RTEN_HIP_BIAS_* */
    int32_t act;        /* RTEN_HIP_ACT_* applied after bias (fused follow-on op) */
} rten_hip_gemm_desc;

int32_t rten_hip_gemm_f32(rten_hip_ctx *ctx, const rten_hip_gemm_desc *desc, const float *a, const float *b,
                          const float *bias, float *c);
/* One-row products (m == 1, per batch element): the reference takes its vector-matrix kernels when A has one row and B is not
 * prepacked (rten-gemm/src/lib.rs:668-747, 876-891; kernels/simd_generic.rs:14-197), and their accumulation order is not the
 * blocked GEMM's (depth blocks of 8 or 512, 16-lane partial sums for transposed B, unfused scalar loops for the columns left over
 * in a column block).  `on` (default 1): rten_hip_gemm_f32 reproduces that order bit for bit; 0: m == 1 takes the blocked order, i.e.
 * the reference with prepacked weights (ModelOptions::prepack_weights).  The reference's column blocks are
 * max(128, ceil(n / threads)) wide, so WHICH columns are "left over" depends on its thread count: `reference_threads` states it
 * (0 = at least n / 128 threads, every block 128 columns). */
int32_t rten_hip_set_gemv_order(rten_hip_ctx *ctx, int32_t on, int32_t reference_threads);

/* ---- int8 GEMM: GemmExecutor<u8,i8,i32>, kernels/generic.rs:274-366; front-ends
 *      matmul_integer / MatMulIntegerToFloat, src/ops/matmul.rs:582-647,789-794 ----
 * C_i32[m,n] = sum_k (A[m,k]-a_zp[m]) * (B[k,n]-b_zp[n]).  A and B may each be u8 or i8 (the four
 * signedness combos of matmul.rs:684-690); zero points have the operand's type, length 1 or M / N.
 * If `scale` is non-NULL the output is f32: (acc as f32) * scale[0 or n]  (cast_scale, :734-773). */
typedef struct {
    int32_t m, n, k;
    int64_t a_rs, a_cs, b_rs, b_cs, ldc;
    int32_t a_signed, b_signed;
    /* a_zp_len: 0 (none), 1 (scalar) or a period p that divides m: row r uses a_zp[r % p].  p == m is the plain
     * per-row form; p < m is the zero-point cycling of a batched LHS collapsed to [A*M, K] (matmul.rs:266-280).
     * b_zp_len: 0, 1 or n. */
    int32_t a_zp_len, b_zp_len;
    int32_t scale_len;          /* 0 (i32 output), 1 or n */
    /* batched_gemm_uninit over a broadcast prefix (matmul.rs:302-372): `batch` products, operand z at
     * base + z * {a,b,c}_bs elements; a stride of 0 broadcasts that operand.  batch <= 1: one product.  Every
     * product uses the same zero points / scale. */
    int32_t batch;
    int64_t a_bs, b_bs, c_bs;
    /* != 0: `b` is a buffer written by rten_hip_gemm_int8_prepack for this (k, n, b_signed); b_rs / b_cs / b_bs
     * are ignored (a prepacked RHS is a single matrix, as in the reference: packed_b is only used when
     * num_b_matrices == 1, matmul.rs:318-327). */
    int32_t b_prepacked;
} rten_hip_gemm_int8_desc;

int32_t rten_hip_gemm_int8(rten_hip_ctx *ctx, const rten_hip_gemm_int8_desc *desc, const void *a, const void *b,
                           const void *a_zp, const void *b_zp, const float *scale, void *c);

/* Load-time staging of a constant MatMulInteger / MatMulIntegerToFloat RHS: PackedBMatrix via
 * Operator::prepack -> matmul_prepack_b (src/ops/matmul.rs:696-705,812-840), Graph::prepack_weights
 * (src/graph.rs:488-562), rten-gemm/src/prepack.rs:19-120, packing/int8.rs:80-249 (the packed image carries the
 * column sums the zero-point epilogue needs).  B [k, n] is read through element strides and written as
 * chunk-major signed bytes [k/16][n][16] followed by int32 column sums; consumed by rten_hip_gemm_int8 with
 * desc.b_prepacked != 0.  packed_bytes returns 0 for shapes the staged kernel does not cover. */
size_t rten_hip_gemm_int8_packed_bytes(int32_t k, int32_t n);
int32_t rten_hip_gemm_int8_prepack(rten_hip_ctx *ctx, int32_t k, int32_t n, const void *b, int64_t b_rs, int64_t b_cs,
                                   int32_t b_signed, void *packed);

/* ---- Conv (f32): Conv::run -> conv_impl, src/ops/conv.rs:124-365,384-400 ----
 * X [n,c,h,w], W [o, c/groups, kh, kw], bias [o] or NULL, Y [n,o,oh,ow]; pads are the FIXED pads
 * (top,left,bottom,right) produced by rten_hip_calc_output_size_and_padding. */
typedef struct {
    int32_t n, c, h, w;
    int32_t o, kh, kw;
    int32_t pads[4];
    int32_t stride_h, stride_w, dil_h, dil_w;
    int32_t groups;
    int32_t out_h, out_w;
} rten_hip_conv2d_desc;

#define RTEN_HIP_CONV_RELU 1u     /* fuse the following Relu */
#define RTEN_HIP_CONV_RESIDUAL 2u /* fuse the following Add(residual) (before Relu) */

/* Load-time weight staging (the GPU analogue of PrepackedInput / Graph::prepack_weights,
 * src/operator.rs:25-66, src/graph.rs:488-562): re-lays W out as [groups][K][O/groups] for
 * coalesced MFMA tile loads.  `packed` must hold rten_hip_conv2d_f32_packed_bytes() bytes. */
size_t rten_hip_conv2d_f32_packed_bytes(const rten_hip_conv2d_desc *desc);
int32_t rten_hip_conv2d_f32_prepack(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *desc, const float *w,
                                    float *packed);
/* weights_packed != 0: `w` is a prepacked buffer; else `w` is the plain OIHW tensor. */
int32_t rten_hip_conv2d_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *desc, const float *x, const float *w,
                            int32_t weights_packed, const float *bias, const float *residual, uint32_t flags,
                            float *y);

/* Two pointwise convolutions in ONE launch (v8): y1 = act1(conv1x1(x, w1) + bias1 [+ residual]), y2 = act2(conv1x1(y1, w2) + bias2) -- an expand
 * layer of a bottleneck block and the reduce layer of the next block (src/ops/conv.rs:248-284 twice).  y1 is written as always; the second
 * convolution reads it from on-chip memory instead of from HBM.  Both weights PREPACKED (rten_hip_conv2d_f32_prepack); both results are bit-identical
 * to two rten_hip_conv2d_f32 calls.  Forms: unit-stride unpadded 1x1 convolutions, groups 1, d1->c == 64, d1->o a multiple of 64 up to 256 (= d2->c),
 * d2->o 64 or 128, out_h * out_w a multiple of 4; anything else: RTEN_HIP_ERR_UNSUPPORTED (ask rten_hip_conv2d_f32_pair_supported first, 1 / 0).
 * flags2 must not carry RTEN_HIP_CONV_RESIDUAL. */
int32_t rten_hip_conv2d_f32_pair_supported(const rten_hip_conv2d_desc *d1, const rten_hip_conv2d_desc *d2);
int32_t rten_hip_conv2d_f32_pair(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d1, const float *x, const float *w1_packed, const float *bias1,
                                 const float *residual, uint32_t flags1, float *y1, const rten_hip_conv2d_desc *d2, const float *w2_packed,
                                 const float *bias2, uint32_t flags2, float *y2);

/* ... with the first convolution's residual COMPUTED in the launch (v8): residual = conv1x1(xd, wd) + biasd, a 64-channel unit-stride pointwise convolution over the same
 * pixels with d1's output channels -- the shortcut layer of a stage's first bottleneck block, whose output no other operator reads: it is neither written nor read back.
 * ds->c == 64, ds->o == d1->o, d2->o == 64; flags1 / flags2 must not carry RTEN_HIP_CONV_RESIDUAL.  Same bits as the three launches (the shortcut's value is rounded to
 * f32 with its bias before it is added, as the tensor would have been). */
int32_t rten_hip_conv2d_f32_pair_shortcut_supported(const rten_hip_conv2d_desc *d1, const rten_hip_conv2d_desc *ds, const rten_hip_conv2d_desc *d2);
int32_t rten_hip_conv2d_f32_pair_shortcut(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *d1, const float *x, const float *w1_packed, const float *bias1,
                                          const rten_hip_conv2d_desc *ds, const float *xd, const float *wd_packed, const float *biasd, uint32_t flags1, float *y1,
                                          const rten_hip_conv2d_desc *d2, const float *w2_packed, const float *bias2, uint32_t flags2, float *y2);

/* ---- Conv (int8): ConvInteger / ConvIntegerToFloat, src/ops/conv.rs:421-476,495-526,571-578 ----
 * x u8|i8 NCHW, w i8|u8 OIHW, x_zp: device scalar of x's type, w_zp: device, length 1 or o.
 * pad_mode: value of out-of-image taps: */
#define RTEN_HIP_PAD_ZERO_POINT 0 /* contributes 0 (ONNX semantics) */
#define RTEN_HIP_PAD_RAW0_I8 1    /* x86 reference: raw 0 after the u8->i8 shift cast */
#define RTEN_HIP_PAD_RAW0_U8 2    /* Arm/wasm reference */
typedef struct {
    rten_hip_conv2d_desc conv;
    int32_t x_signed, w_signed;
    int32_t w_zp_len; /* 0, 1 or o */
    int32_t pad_mode;
    int32_t weights_packed; /* != 0: `w` is a buffer written by rten_hip_conv2d_int8_prepack */
    int32_t x_staged;       /* != 0: `x` is a staged image written by rten_hip_dynamic_quantize_linear_staged */
    int32_t scale_len;      /* 0 or 1: scalar scale (ConvIntegerToFloat, fusions.rs:1046-1049); o: per-output-channel
                             * scale (the unfused ConvInteger -> Cast -> Mul([1,O,1,1]) form of per-channel weights) */
} rten_hip_conv2d_int8_desc;
/* scale == NULL: y is i32 (ConvInteger).  scale != NULL (device scalar, or [o] with desc.scale_len == o): y is f32 =
 * (acc as f32) * scale[0 or o], then optional bias[o] add (the Add node that follows in ort-quantized
 * graphs), optional residual add and Relu per `flags`. */
/* Load-time staging of constant ConvInteger weights (PrepackedInput / Graph::prepack_weights analogue,
 * src/operator.rs:25-66): (ky, kx, c)-ordered signed rows with channels padded to 16, followed by the row sums the
 * zero-point epilogue needs (packing/int8.rs).  packed_bytes returns 0 when the staged kernel does not cover the
 * geometry (grouped convolution): pass the plain OIHW tensor then. */
size_t rten_hip_conv2d_int8_packed_bytes(const rten_hip_conv2d_int8_desc *desc);
int32_t rten_hip_conv2d_int8_prepack(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const void *w, void *packed);
/* DynamicQuantizeLinear fused with the activation staging of the ConvInteger that consumes it (the reference runs
 * DynamicQuantizeLinear -> ConvInteger back to back in ort-quantized graphs, src/ops/quantize.rs:352-436 ->
 * src/ops/conv.rs:421-476): same scale / zero point / u8 codes bit for bit, but the codes are written once, directly
 * in the layout the int8 kernel gathers from (zero-point-padded, channel-blocked [N][C/16][H+pads][W+pads][16], signed domain; opaque to the caller) instead of as an NCHW u8 tensor.
 * `desc` is the consumer's descriptor (x_signed must be 0: DynamicQuantizeLinear produces u8); staged_bytes returns 0
 * when the staged kernel does not cover the geometry.  `mul_by` / `product` fold the Mul(x_scale, w_scale) node that
 * follows in ort-quantized graphs (one f32 multiply, same bits). */
size_t rten_hip_conv2d_int8_staged_bytes(const rten_hip_conv2d_int8_desc *desc);
int32_t rten_hip_dynamic_quantize_linear_staged(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const float *x,
                                                void *staged, float *scale, uint8_t *zero_point,
                                                const float *mul_by /* optional */, float *product /* = scale * mul_by[0] */);
int32_t rten_hip_conv2d_int8(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const void *x, const void *w,
                             const void *x_zp, const void *w_zp, const float *scale, const float *bias,
                             const float *residual, uint32_t flags, void *y);

/* Producer-side statistics: a ConvIntegerToFloat whose output is quantized next (DynamicQuantizeLinear -> ConvInteger, the
 * shape of every layer of an ort-quantized CNN) accumulates the output's min / max in its epilogue, and the quantize step
 * reads them instead of sweeping the tensor a first time.  Same statistics (min / max are order independent), same bits.
 * `stats` is a device buffer of rten_hip_minmax_stats_bytes(), reset before its producer runs; `count` consecutive
 * buffers (e.g. one per layer of a model, in one allocation) are reset by a single launch. */
size_t rten_hip_minmax_stats_bytes(void);
int32_t rten_hip_minmax_stats_reset(rten_hip_ctx *ctx, void *stats, int32_t count);
int32_t rten_hip_conv2d_int8_stats(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const void *x, const void *w,
                                   const void *x_zp, const void *w_zp, const float *scale, const float *bias,
                                   const float *residual, uint32_t flags, void *y, void *stats);
int32_t rten_hip_dynamic_quantize_linear_staged_stats(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const float *x,
                                                      const void *stats, void *staged, float *scale, uint8_t *zero_point,
                                                      const float *mul_by, float *product);
/* One DynamicQuantizeLinear read by SEVERAL ConvInteger nodes (ort-quantize reuses a quantized input: a stage's shortcut convolution and
 * its first 1x1 convolution read the same tensor) is followed by one scalar Mul(x_scale, w_scale_i) per consumer.  This form folds all of
 * them into the quantizer's launch: product[i][0] = scale * mul_by[i][0] for i < count (count <= 4; `mul_by` / `product` are HOST arrays of
 * device pointers), each a single f32 multiply -- the bits of the separate Mul nodes.  `stats` optional: non-NULL = the producer's
 * statistics block (as rten_hip_dynamic_quantize_linear_staged_stats), NULL = the quantizer sweeps the tensor itself (as
 * rten_hip_dynamic_quantize_linear_staged). */
int32_t rten_hip_dynamic_quantize_linear_staged_products(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const float *x,
                                                         const void *stats /* optional */, void *staged, float *scale,
                                                         uint8_t *zero_point, int32_t count, const float *const *mul_by,
                                                         float *const *product);
/* The whole chain DynamicQuantizeLinear -> ConvInteger -> Cast -> Mul(x_scale, w_scale) [-> Add bias] [-> Add residual] [-> Relu]
 * (the reference's DynamicQuantizeLinear + ConvIntegerToFloat, src/ops/quantize.rs:352-436 + src/ops/conv.rs:495-587) in ONE launch
 * for pointwise convolutions (1x1, stride 1, no padding, groups 1, C % 64 == 0) whose input statistics `in_stats` were accumulated
 * by the producing launch: the quantizer runs in the integer GEMM's operand loader, no quantized tensor is written.  `w` must be
 * prepacked (desc->weights_packed), `w_scale` is the scalar or per-output-channel weight scale (desc->scale_len values), the
 * quantizer's own outputs go to `x_scale_out` / `x_zero_point_out` (optional), the float outputs' statistics to `out_stats`
 * (optional).  Bit-identical to the separate operators.  RTEN_HIP_ERR_UNSUPPORTED for other geometries. */
int32_t rten_hip_conv2d_int8_dql(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const float *x, const void *in_stats,
                                 const void *w, const float *w_scale, const float *bias, const float *residual, uint32_t flags,
                                 float *y, void *out_stats, float *x_scale_out, uint8_t *x_zero_point_out);

/* ConvIntegerToFloat [-> Add bias] [-> Add residual] [-> Relu] AND the DynamicQuantizeLinear of the ONE convolution that consumes its
 * output (the c1 -> c2 -> c3 edges of a bottleneck block in an ort-quantized CNN; the reference runs ConvIntegerToFloat,
 * src/ops/conv.rs:495-587, then DynamicQuantizeLinear, src/ops/quantize.rs:352-436, as two operators with an f32 tensor between them)
 * in ONE launch: the finished f32 values stay in registers, every workgroup publishes its min / max as one 8-byte granule of the
 * exchange block `sync` and sweeps everybody else's (a grid-wide all-gather: the only cross-workgroup step), then quantizes its own
 * values with the same scale / zero-point algebra and writes the u8 codes
 * directly as the staged image of `next` (the consumer's descriptor; what rten_hip_dynamic_quantize_linear_staged would have written,
 * byte for byte, border pieces included).  `y` may be NULL: the f32 tensor is then never materialised (4 B written + 4 B read per
 * element less).  `next_scale` / `next_zero_point` receive DynamicQuantizeLinear's outputs, `product` = scale * mul_by[0] (optional).
 * `x` must be staged and `w` prepacked (desc->x_staged, desc->weights_packed); `stats` as for rten_hip_conv2d_int8_stats (reset
 * before the call; it receives the output's statistics as there); `sync` is a device buffer of rten_hip_grid_sync_bytes(), initialised
 * ONCE with rten_hip_grid_sync_reset when allocated -- every launch leaves it in that state.  The launch needs all its workgroups resident at once and nothing else running on the device's compute units:
 * RTEN_HIP_ERR_UNSUPPORTED when the grid does not fit (or the geometry is not covered) -> run the two-launch sequence.  A launch that
 * nevertheless waits longer than ~0.5 s for its grid (the device was shared with other work) gives up LOUDLY: it sets the block's time-out flag
 * (rten_hip_grid_sync_timeouts), stores NaN as `next_scale` / `product` -- statistics that miss a workgroup never become plausible codes: every value
 * downstream is NaN -- and raises the context's sticky fault: rten_hip_sync and rten_hip_graph_launch fail with RTEN_HIP_ERR_HIP from then on, until
 * rten_hip_grid_sync_reset (which also clears the fault).  A library-level executor keeps this launch form opt-in (launch plan), never a default.
 * `sync` == NULL (v6) selects the RECOMPUTE form instead: two launches -- the convolution with its epilogue but no stores (statistics into `stats`), then
 * the convolution again, which folds `stats`, quantizes in its epilogue and writes the same staged image / scale / zero point / product.  No grid-wide
 * exchange, no residency requirement, no time-out: safe beside other work (replicas, "lanes").  Covered: the single-row-term form (signed weights without
 * a zero point, one activation zero point) without a residual; RTEN_HIP_ERR_UNSUPPORTED otherwise -> the two-operator sequence. */
size_t rten_hip_grid_sync_bytes(void);
int32_t rten_hip_grid_sync_reset(rten_hip_ctx *ctx, void *sync, int32_t count /* consecutive blocks */);
int32_t rten_hip_conv2d_int8_qout(rten_hip_ctx *ctx, const rten_hip_conv2d_int8_desc *desc, const void *x, const void *w,
                                  const void *x_zp, const void *w_zp, const float *scale, const float *bias, const float *residual,
                                  uint32_t flags, float *y /* optional */, void *stats, void *sync,
                                  const rten_hip_conv2d_int8_desc *next, void *next_staged, float *next_scale,
                                  uint8_t *next_zero_point, const float *mul_by /* optional */, float *product);
int32_t rten_hip_grid_sync_timeouts(rten_hip_ctx *ctx, const void *sync, int32_t count, int32_t *timeouts);

/* ---- DynamicQuantizeLinear, src/ops/quantize.rs:352-436 + rten-vecmath/src/quantize.rs:39-79 ---- */
int32_t rten_hip_dynamic_quantize_linear(rten_hip_ctx *ctx, int64_t n, const float *x, uint8_t *y, float *scale,
                                         uint8_t *zero_point);
/* cast_scale, src/ops/matmul.rs:734-773: y = (x as f32) * scale[scale_len == 1 ? 0 : i % scale_len] */
int32_t rten_hip_cast_scale(rten_hip_ctx *ctx, int64_t n, const int32_t *x, const float *scale, int32_t scale_len,
                            float *y);

/* ---- Softmax / AddSoftmax: src/ops/norm.rs:825-840, src/ops/attention.rs:30-68,
 *      rten-vecmath/src/softmax.rs:60-100 ----
 * rows x cols, softmax along cols.  addend (optional) is added first; row r uses addend row
 * (r / add_div) % add_mod  (covers [B,1,1,S] masks against [B,H,S,S] scores and same-shape). */
int32_t rten_hip_softmax_f32(rten_hip_ctx *ctx, int64_t rows, int32_t cols, const float *x, const float *addend,
                             int64_t add_div, int64_t add_mod, int32_t flush_nan_to_zero, float *y);

/* ---- LayerNormalization: src/ops/norm.rs:456-529 + rten-vecmath/src/normalize.rs:82-170 ----
 * gamma/beta NULL => gamma_scalar/beta_scalar (scalar-broadcast scale / absent bias). */
int32_t rten_hip_layer_norm_f32(rten_hip_ctx *ctx, int64_t rows, int32_t cols, const float *x, const float *gamma,
                                const float *beta, float gamma_scalar, float beta_scalar, float epsilon,
                                float *y);
/* LayerNormalization(x + addend): the Add -> LayerNormalization pair of every transformer block as one kernel (the sum is
 * formed with the same f32 add and never written to memory; bit-identical to the two operators).  y may alias x / addend. */
int32_t rten_hip_add_layer_norm_f32(rten_hip_ctx *ctx, int64_t rows, int32_t cols, const float *x, const float *addend,
                                    const float *gamma, const float *beta, float gamma_scalar, float beta_scalar,
                                    float epsilon, float *y);
/* BatchNormalization (inference), src/ops/norm.rs:194-224 */
int32_t rten_hip_batch_norm_f32(rten_hip_ctx *ctx, int32_t n, int32_t c, int64_t inner, const float *x,
                                const float *scale, const float *bias, const float *mean, const float *var,
                                float epsilon, float *y);

/* ---- element-wise: src/ops/unary_elementwise.rs:399-420,611-613; binary_elementwise.rs:476-495 ---- */
int32_t rten_hip_relu_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y);
int32_t rten_hip_gelu_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y);
int32_t rten_hip_erf_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y);
/* Tanh (rten-vecmath/src/tanh.rs:12-72: odd polynomial below 0.55, (exp(2|x|) - 1) / (exp(2|x|) + 1) above, 1 from 9.02), bit-identical to the reference's
 * AVX-512 / AVX2 form (sign handled as bit operations: tanh(+0) = -0 there). */
int32_t rten_hip_tanh_f32(rten_hip_ctx *ctx, int64_t n, const float *x, float *y);
/* y[i] = a[i] + b[i % b_len] (b_len == n: same shape; b_len < n: trailing-dims broadcast) */
int32_t rten_hip_add_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len, float *y);
int32_t rten_hip_mul_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len, float *y);
int32_t rten_hip_sub_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len, float *y);
int32_t rten_hip_div_f32(rten_hip_ctx *ctx, int64_t n, const float *a, const float *b, int64_t b_len, float *y);
/* General numpy broadcasting (binary_elementwise.rs:58-170): op 0 add, 1 mul, 2 sub, 3 div; out_shape[ndim] (ndim <= 6);
 * a_strides / b_strides in elements of the operands as expanded to out_shape, 0 on broadcast axes. */
int32_t rten_hip_binary_broadcast_f32(rten_hip_ctx *ctx, int32_t op, int32_t ndim, const int64_t *out_shape, const int64_t *a_strides,
                                      const int64_t *b_strides, const float *a, const float *b, float *y);
/* Transpose (src/ops/layout.rs:669+) of 4-byte elements: y.shape[d] = x_shape[perm[d]], ndim <= 6; an invalid perm is
 * "Permutation is invalid" like the reference. */
int32_t rten_hip_transpose_b32(rten_hip_ctx *ctx, int32_t ndim, const int64_t *x_shape, const int32_t *perm, const void *x, void *y);
/* y (contiguous, shape[ndim], ndim <= 6) = x read through x_strides (elements; 0 = broadcast axis, a sum of strides = a
 * diagonal): TensorView::to_tensor / to_contiguous / expand_to of the permuted, diagonal and expanded views that Einsum
 * builds (src/ops/einsum.rs:124-162,255-257,320-329,534-535). */
int32_t rten_hip_copy_strided_b32(rten_hip_ctx *ctx, int32_t ndim, const int64_t *shape, const int64_t *x_strides, const void *x, void *y);
/* Layout / logic operators of exporter-written graphs over operands viewed through element strides (0 = broadcast axis; shape[ndim], ndim <= 6; y is
 * contiguous).  Booleans are int32 0 / 1, as in the reference (onnx_loader.rs:332-339 narrows BOOL to Int32).
 *   RTEN_HIP_EW_CAST                  y = a as y_dtype (src/ops/convert.rs:18-60; Rust `as`: float -> int truncates toward zero, saturates, NaN -> 0;
 *                                     int -> narrower int wraps).  Every pair of float32 / int32 / uint8 / int8 except same-type (a plain copy).
 *   RTEN_HIP_EW_NOT / AND / OR / XOR  int32 operands, int32 0 / 1 result (unary_elementwise.rs:563-565, binary_elementwise.rs:546-598)
 *   RTEN_HIP_EW_EQUAL .. GREATER_EQ   two float32 or two int32 operands, int32 0 / 1 result (binary_elementwise.rs:733-786)
 *   RTEN_HIP_EW_WHERE                 y = a != 0 ? b : c; a int32, b / c / y 4-byte elements of one type (binary_elementwise.rs:1189-1247)
 *   RTEN_HIP_EW_IADD / ISUB / IMUL / IDIV   int32 arithmetic (wrapping; division truncates toward zero; x / 0 = 0 on the device -- the reference
 *                                     panics -- so hosts reject a constant zero divisor)
 * Operands an operator does not take are NULL.  Mask and index preparation, never a model's critical path: one scalar kernel, no vector forms. */
#define RTEN_HIP_DT_F32 0
#define RTEN_HIP_DT_I32 1
#define RTEN_HIP_DT_U8 2
#define RTEN_HIP_DT_I8 3
#define RTEN_HIP_EW_CAST 0
#define RTEN_HIP_EW_NOT 1
#define RTEN_HIP_EW_AND 2
#define RTEN_HIP_EW_OR 3
#define RTEN_HIP_EW_XOR 4
#define RTEN_HIP_EW_EQUAL 5
#define RTEN_HIP_EW_LESS 6
#define RTEN_HIP_EW_LESS_EQ 7
#define RTEN_HIP_EW_GREATER 8
#define RTEN_HIP_EW_GREATER_EQ 9
#define RTEN_HIP_EW_WHERE 10
#define RTEN_HIP_EW_IADD 11
#define RTEN_HIP_EW_ISUB 12
#define RTEN_HIP_EW_IMUL 13
#define RTEN_HIP_EW_IDIV 14
int32_t rten_hip_elementwise_nd(rten_hip_ctx *ctx, int32_t op, int32_t ndim, const int64_t *shape, const void *a, int32_t a_dtype, const int64_t *a_strides,
                                const void *b, int32_t b_dtype, const int64_t *b_strides, const void *c, const int64_t *c_strides, void *y, int32_t y_dtype);
/* Gather along any axis of 4-byte elements (src/ops/gather.rs:21-110): y[o][j][k] = data[o][ids[j]][k] with data viewed as [outer][axis_len][inner];
 * negative indices count from the end; out-of-range indices are clamped (the reference reports "Entry in indices is out of range"). */
int32_t rten_hip_gather_axis_b32(rten_hip_ctx *ctx, int64_t outer, int64_t axis_len, int64_t inner, int64_t n_ids, const void *data, const int32_t *ids, void *y);
/* rows x row_elems 4-byte elements from src (row pitch src_pitch elements) to dst (row pitch dst_pitch >= row_elems): one piece of a Concat
 * (src/ops/concat.rs:108) written into its slot of the output. */
int32_t rten_hip_copy_rows_b32(rten_hip_ctx *ctx, int64_t rows, int64_t row_elems, const void *src, int64_t src_pitch, void *dst, int64_t dst_pitch);
/* 1 while a graph capture is active on `ctx` (rten_hip_graph_begin .. _end): host code must not upload from host memory then -- the copy would be
 * recorded with the host pointer and re-read at every replay. */
int32_t rten_hip_capture_active(rten_hip_ctx *ctx);
/* ReduceSum of a strided view (reduce_sum, src/ops/reduce.rs:414-520,1101-1124; vecmath::Sum, rten-vecmath/src/sum.rs:12-34):
 * y[r] (r = row-major index over the kept dims outer_shape[n_outer]) = sum of the slice spanned by the reduced dims
 * inner_shape[n_inner] walked in row-major order -- the order the reference packs a non-contiguous slice in -- added in the
 * AVX-512 16-lane order (bit-identical to the reference on the GPU box's host).  An empty slice sums to 0.  <= 6 + 6 dims. */
int32_t rten_hip_reduce_sum_strided_f32(rten_hip_ctx *ctx, int32_t n_outer, const int64_t *outer_shape, const int64_t *outer_strides,
                                        int32_t n_inner, const int64_t *inner_shape, const int64_t *inner_strides,
                                        const float *x, float *y);
/* ReduceMean (reduce_mean, src/ops/reduce.rs:523-541): the same sum divided by the slice length as f32; an empty slice gives NaN. */
int32_t rten_hip_reduce_mean_strided_f32(rten_hip_ctx *ctx, int32_t n_outer, const int64_t *outer_shape, const int64_t *outer_strides,
                                         int32_t n_inner, const int64_t *inner_shape, const int64_t *inner_strides,
                                         const float *x, float *y);
/* y[(i / inner) ...] += bias[c]: per-channel bias add for NCHW tensors ([1,O,1,1] constant Add) */
int32_t rten_hip_add_channel_bias_f32(rten_hip_ctx *ctx, int32_t n, int32_t c, int64_t inner, const float *x,
                                      const float *bias, float *y);

/* ---- ConvTranspose: src/ops/conv_transpose.rs:226-412 (GEMM into a column matrix + col2im), :144-224 (output size) ----
 * desc: n, c, h, w = input; o = output channels over all groups; kernel [c, o / groups, kh, kw]; pads / out_h / out_w as
 * resolved by rten_hip_conv_transpose_output_size (same = Padding::Same; the same error strings through *msg). */
int32_t rten_hip_conv_transpose_output_size(int32_t in_h, int32_t in_w, int32_t kh, int32_t kw, int32_t stride_h, int32_t stride_w, int32_t same,
                                            const int32_t pads[4], int32_t dil_h, int32_t dil_w, int32_t out_pad_h, int32_t out_pad_w, int32_t out_hw[2],
                                            int32_t out_pads[4], const char **msg);
int32_t rten_hip_conv_transpose2d_f32(rten_hip_ctx *ctx, const rten_hip_conv2d_desc *desc, const float *x, const float *w, const float *bias, float *y);

/* ---- MatMulNBits (com.microsoft contrib op): src/ops/matmul/contrib.rs:21-106, rten-gemm/src/block_quant.rs:61-141 (vector x
 * matrix), rten-gemm/src/packing.rs:229-318 (dequantise-while-packing for the ordinary GEMM) ----
 * a [batch][rows][k] f32; b_quant [n][k / block_size][block_size / 2] packed 4-bit (even element low nibble), zero point 8;
 * scales [n][k / block_size]; y [batch][rows][n].  block_size: power of two >= 16 ("Unsupported K block size" otherwise).
 * rows == 1 reproduces the reference's 64-slot accumulation order, rows > 1 its f32 GEMM on the dequantised matrix. */
int32_t rten_hip_matmul_nbits_f32(rten_hip_ctx *ctx, int64_t batch, int32_t rows, int32_t k, int32_t n, int32_t block_size, const float *a,
                                  const uint8_t *b_quant, const float *scales, float *y);

/* ---- pooling: src/ops/pooling.rs:174-389,392-417,516-521,581-600 ---- */
typedef struct {
    int32_t n, c, h, w;
    int32_t kh, kw, stride_h, stride_w;
    int32_t pads[4];
    int32_t out_h, out_w;
    int32_t count_include_pad; /* AveragePool only */
} rten_hip_pool2d_desc;
int32_t rten_hip_max_pool2d_f32(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x, float *y);
/* MaxPool whose output is quantized next (stem -> MaxPool -> DynamicQuantizeLinear in an ort-quantized CNN): the same values, plus their
 * min / max accumulated into `stats` (rten_hip_minmax_stats_bytes(), reset before the call) for
 * rten_hip_dynamic_quantize_linear_staged_stats / _products / rten_hip_conv2d_int8_dql, which then skip their own sweep. */
int32_t rten_hip_max_pool2d_f32_stats(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x, float *y, void *stats);
int32_t rten_hip_average_pool2d_f32(rten_hip_ctx *ctx, const rten_hip_pool2d_desc *desc, const float *x, float *y);
int32_t rten_hip_global_average_pool_f32(rten_hip_ctx *ctx, int64_t nc, int32_t inner, const float *x, float *y);

/* ---- fused attention: sdpa_head / sdpa_multi_head, src/ops/attention.rs:518-626 ----
 * out[b,h] = softmax(scale * Q[b,h] K[b,h]^T + mask[b]) V[b,h].  Q/K/V/out are addressed by element
 * strides (batch, head, row; the last dim is contiguous) so both [B,H,S,D] tensors and the
 * [B,S,H*D] projection outputs of BERT (head = column block) are read in place.
 * mask: NULL, or additive f32 [B,1,1,T] (mask_row_stride = 0, mask_batch_stride = T) / [B,1,S,T] (mask_row_stride = T, mask_batch_stride = S * T) /
 * one shared row per batch item read out of a [B,1,S,T] tensor whose S rows are equal (mask_row_stride = 0, mask_batch_stride = S * T). */
typedef struct {
    int32_t batch, heads, s, t, d, dv;
    int64_t q_bs, q_hs, q_rs;
    int64_t k_bs, k_hs, k_rs;
    int64_t v_bs, v_hs, v_rs;
    int64_t o_bs, o_hs, o_rs;
    int64_t mask_batch_stride, mask_row_stride;
    float scale;
    int32_t flush_nan_to_zero; /* 1: sdpa_head semantics (attention.rs:551); 0: the FusedMatMul -> AddSoftmax -> MatMul graph */
} rten_hip_sdpa_desc;
int32_t rten_hip_sdpa_f32(rten_hip_ctx *ctx, const rten_hip_sdpa_desc *desc, const float *q, const float *k,
                          const float *v, const float *mask, float *out);

/* ---- Gather of rows (embedding lookup; src/ops/gather.rs Gather axis 0): out[i,:] = table[ids[i],:] ---- */
int32_t rten_hip_gather_rows_f32(rten_hip_ctx *ctx, int64_t n_ids, int32_t row_len, int32_t table_rows,
                                 const float *table, const int32_t *ids, float *out);

/* ---- multi-GPU: one process per GPU, batches sharded with no data-path collective; the only
 *      collective is the one-time broadcast of the prepacked weight arena over xGMI.  The reference has no
 *      analogue (single-process CPU executor); a Rust host calls these next to Model::load.  librccl is
 *      loaded on first use (dlopen), so single-GPU users never map it.
 *      Rank 0 obtains a 128-byte id and hands it to the other ranks by any host-side channel (file, env, socket);
 *      every rank then calls comm_init_rank with its own context (device).  broadcast runs on the context's
 *      stream: stream-ordered with the kernels that consume the weights. */
#define RTEN_HIP_COMM_ID_BYTES 128
typedef struct rten_hip_comm rten_hip_comm;
int32_t rten_hip_comm_get_unique_id(rten_hip_ctx *ctx, uint8_t id[RTEN_HIP_COMM_ID_BYTES]);
int32_t rten_hip_comm_init_rank(rten_hip_ctx *ctx, const uint8_t id[RTEN_HIP_COMM_ID_BYTES], int32_t world_size, int32_t rank,
                                rten_hip_comm **out_comm);
/* In-place broadcast of `bytes` bytes at device pointer `buf` from rank `root` to every rank. */
int32_t rten_hip_broadcast(rten_hip_ctx *ctx, rten_hip_comm *comm, void *buf, size_t bytes, int32_t root);
int32_t rten_hip_comm_world_size(rten_hip_comm *comm, int32_t *world_size, int32_t *rank);
int32_t rten_hip_comm_destroy(rten_hip_ctx *ctx, rten_hip_comm *comm);

/* ---- the plan executor behind the C ABI (new; the host-side analogue is Model::load + Model::run, src/model.rs:235-550, for ONE resident subgraph) ----
 * An ONNX graph (bytes of a ModelProto) compiled into a static plan whose every value stays in HBM: constants uploaded and conv weights prepacked once,
 * the fusions of optimize/fusions.rs:1012-1058 applied, per-layer launch plan from a plan file (profiles/plans/[*].json: {step: [variant, split mode,
 * K groups, order]}, optionally keyed by sub-batch size) or tuned at prepare time, the batch run as `chains` independent dim-0 slices on their own
 * streams, each captured into a hipGraph.  This is what lets a Rust host keep activations on the device WITHOUT a new `Value` variant: a maximal run
 * of accelerated nodes becomes one `Operator` that owns an rten_hip_model (precedent: SubgraphOperator, src/operator.rs:630-646).
 * ("model", not "graph": rten_hip_graph_* above is the hipGraph capture of a context's stream.)
 *   load -> bind_input (each input's full-batch shape; returns the device pointer the caller writes the input to) -> prepare -> { run, sync, output }*
 * Chain 0 runs on `ctx` itself (a model with N chains owns N - 1 streams of its own: one stream more than chains costs real time on this runtime), so
 * `ctx` must outlive the model: destroy models before their context.
 * A plan file may also carry {"qout": [ConvInteger node names]} (profiles/plans/int8.json): with chains == 1 those fused ConvIntegerToFloat steps
 * run the DynamicQuantizeLinear of the one convolution reading their output in the same launch (rten_hip_conv2d_int8_qout above, with its opt-in and
 * time-out contract; an edge whose launch is refused at run time runs the two operators).  rten_hip_model_info's n_planned_steps counts them.
 * {"qout2": [names]} (v6) lists edges for the recompute form of the same entry point (sync == NULL): no exchange, so they are allowed with replicas.
 * Errors: the usual status codes; rten_hip_model_last_error has the text.  Not thread-safe per model object (one caller at a time). */
typedef struct rten_hip_model rten_hip_model;
int32_t rten_hip_model_load(rten_hip_ctx *ctx, const void *onnx_bytes, size_t onnx_len, const char *plan_json /* optional */, int32_t chains,
                            int32_t device_id, rten_hip_model **out_model);
const char *rten_hip_model_last_error(const rten_hip_model *model);
int32_t rten_hip_model_info(const rten_hip_model *model, int32_t *n_inputs, int32_t *n_outputs, int32_t *n_steps, int32_t *n_planned_steps);
const char *rten_hip_model_input_name(const rten_hip_model *model, int32_t i);
const char *rten_hip_model_output_name(const rten_hip_model *model, int32_t i);
int32_t rten_hip_model_bind_input(rten_hip_model *model, int32_t i, const int64_t *shape, int32_t ndim, void **dev_ptr);
/* Element type of input i (as bind_input allocates it) / of output i (after prepare), RTEN_HIP_DTYPE_*.  ONNX int64 graph inputs are I32 on the device, like
 * every 64-bit index tensor in the reference (rten-onnx narrows them at load: a host hands 32-bit integers over). */
#define RTEN_HIP_DTYPE_F32 0
#define RTEN_HIP_DTYPE_I32 1
#define RTEN_HIP_DTYPE_U8 2
#define RTEN_HIP_DTYPE_I8 3
int32_t rten_hip_model_input_dtype(const rten_hip_model *model, int32_t i, int32_t *dtype);
int32_t rten_hip_model_output_dtype(const rten_hip_model *model, int32_t i, int32_t *dtype);
/* tune != 0 and no plan file: every f32 convolution step times its candidate launch plans once per distinct sub-batch size. */
int32_t rten_hip_model_prepare(rten_hip_model *model, int32_t tune);
/* flags bit 0: the inputs were written on the caller context's stream since the last run (the chains wait for it first); bit 1: do not order the
 * caller's stream after the chains (the caller calls rten_hip_model_sync before reading the outputs). */
int32_t rten_hip_model_run(rten_hip_model *model, uint32_t flags);
/* rten_hip_model_load with the device taken from `ctx` (every chain is created on the context's device) and load flags.
 * RTEN_HIP_MODEL_RECEIVE_WEIGHTS: this process RECEIVES the weight arena -- rank != 0 of a batch-sharded job: initializers of 64 KB
 * and more are allocated but not uploaded; the caller then fills rten_hip_model_weight_arena() by rten_hip_broadcast from the rank that loaded the
 * model without the flag, BEFORE rten_hip_model_prepare.  The arena is one device allocation that holds every constant of the model (initializers,
 * constants derived at load time, prepacked conv / MatMul weights); its layout depends only on the model bytes, the plan file and the chain count,
 * so every rank computes the same one.
 * rten_hip_model_load_error: why the calling thread's last rten_hip_model_load / _load_ex failed (parse error, operator outside the registry, bad plan
 * file, a batch-coupled graph with chains > 1, ...): a failed load has no model object to ask. */
#define RTEN_HIP_MODEL_RECEIVE_WEIGHTS 1u
int32_t rten_hip_model_load_ex(rten_hip_ctx *ctx, const void *onnx_bytes, size_t onnx_len, const char *plan_json /* optional */, int32_t chains,
                               uint32_t flags, rten_hip_model **out_model);
const char *rten_hip_model_load_error(void);
int32_t rten_hip_model_weight_arena(rten_hip_model *model, void **dev_ptr, size_t *bytes);
/* Another replica of `model` on `ctx` (same device; `ctx` outlives the replica): the same graph / plan / options, its own buffers, streams and hipGraphs,
 * and the ORIGIN's constants and prepacked weights -- no second copy of the weight arena.  Independent batches handed to different replicas overlap on
 * the device ("lanes": bench.py --lanes; the batch-level analogue of sub-batch chains, and the only one a batch-coupled graph -- the dynamically
 * quantized one -- can use).  Bind inputs and prepare a replica like any model; destroy replicas before their origin (destroy(origin) refuses).
 * RTEN_HIP_ERR_INVALID_VALUE (reason in rten_hip_model_load_error) when the origin's plan lists quantized-output edges ("qout"): those launches are
 * grid-wide exchanges that need the device to themselves, and replicas run side by side -- the rule rten_hip_model_load_ex applies to chains != 1. */
int32_t rten_hip_model_clone(rten_hip_model *model, rten_hip_ctx *ctx, rten_hip_model **out_model);
/* The launch plan a prepared model runs under, as plan-file text keyed by sub-batch size (what prepare(tune = 1) chose / the plan file gave): f32
 * convolution steps AND MatMul / FusedMatMul / Gemm steps (v4: the latter take plan entries and are tuned too).  `*needed` = bytes incl. terminator. */
int32_t rten_hip_model_plan_json(rten_hip_model *model, char *buf, size_t buf_len, size_t *needed);
/* Replaces the step tables of the model's launch plan (plan-file text) and marks it unprepared; the next rten_hip_model_prepare applies the new plan and
 * re-captures the chains (the bound inputs stay).  For tuners that measure the whole model under its real schedule (tools/tune_lanes.py). */
int32_t rten_hip_model_set_plan(rten_hip_model *model, const char *plan_json);
/* Measurement aid: every chain runs its plan eagerly `steps` times, one chain after the other, under the per-launch profiler (rten_hip_profile_*);
 * result = a JSON array of one rten_hip_profile_report array per chain.  Call it with `buf` NULL / too small to learn `*needed` is NOT supported
 * (the pass would run twice): pass a buffer of 1 MiB. */
int32_t rten_hip_model_profile(rten_hip_model *model, int32_t steps, char *buf, size_t buf_len, size_t *needed);
int32_t rten_hip_model_sync(rten_hip_model *model);
int32_t rten_hip_model_output(rten_hip_model *model, int32_t i, const void **dev_ptr, int64_t *shape /* 8 entries */, int32_t *ndim);
int32_t rten_hip_model_destroy(rten_hip_model *model);

/* ---- tuning knobs are sticky per context.  save / restore snapshot all of them (GEMM variant override, split-K plan, tile order, gemv order and its
 * thread assumption, int8 path, attention path): code that changes knobs around its own launches on a context it does not own restores the owner's
 * settings, not the defaults. */
int32_t rten_hip_tuning_save(rten_hip_ctx *ctx, int32_t state[8]);
int32_t rten_hip_tuning_restore(rten_hip_ctx *ctx, const int32_t state[8]);

/* ---- tuning: per-shape kernel-variant selection by measurement at load time ----
 * variant < 0 restores the built-in heuristic.  Used by the harness's autotuner.  Variant 31 (round 5): rten_hip_gemm_f32 calls with one batch and
 * at most 64 rows stream B through every compute unit, one wave per 16x16 output block per depth block (also what the heuristic picks there). */
int32_t rten_hip_set_gemm_variant_override(rten_hip_ctx *ctx, int32_t variant);
int32_t rten_hip_num_gemm_variants(void);
/* Exact split-K for GEMM / conv (results stay bit-identical: K is only cut at the reference's depth-block
 * boundaries, rten-gemm/src/lib.rs:630-633, and the per-block partial sums are added in block order by a fixup
 * kernel).  mode 0 = off, 1 = split only the tiles beyond the last full round of compute units, 2 = split every
 * tile, 3 = automatic (default: split every tile when the launch would have fewer workgroups than half the compute
 * units); groups = K groups per split tile (modes 1, 2).  Mode 4 (no K split): a convolution whose tile count is not a
 * multiple of the compute units runs its whole rounds with the selected tile shape and the remaining columns as thin
 * 16 x 64 tiles on v_mfma_f32_16x16x4_f32 (a quarter of the per-SIMD work, so the partial last round costs a quarter of
 * a round); other calls ignore it.  Mode 5: persistent launch of (compute units x groups) workgroups, each walking a list of
 * tiles with its tile DMA running across tile boundaries (LDS-DMA variants 0..3 and 20..23; others ignore it).  Mode 6: the
 * lean form of mode 5 for convolutions with 64 x 64 tiles and K a multiple of 32 (k-tiles of 32, no selects or table waits in
 * the loop); other calls ignore it.  A tuning knob like the variant override: sticky. */
int32_t rten_hip_set_gemm_split(rten_hip_ctx *ctx, int32_t mode, int32_t groups);
/* Workgroup -> tile order (tuning knob, sticky, default 0): bit 0 = tiles walk n fastest instead of m fastest;
 * bit 1 = split-K workgroups walk tiles fastest and K groups slowest, so that each XCD's private L2 holds one K
 * slice of both operands; bit 3 = RELAXED split-K (LDS-DMA pipelines, split modes 1-2): a K group's depth blocks accumulate in one register block
 * and one partial per group is folded -- NOT the reference's order (results differ in the last bits): exists to MEASURE what an order-free split would
 * gain (profiles/r08/), never part of a committed plan, and since v6 ACCEPTED ONLY BY MEASUREMENT BUILDS (build.sh -DRTEN_ABLATION): the product library
 * rejects it with RTEN_HIP_ERR_INVALID_VALUE, so that no plan file can switch the bit-exactness contract off; bits 4-6 = occupancy cap of the LDS-DMA kernels (workgroups per compute unit, 2..7; 0 = whatever fits:
 * the launch is padded with dynamic LDS it never touches). */
int32_t rten_hip_set_gemm_order(rten_hip_ctx *ctx, int32_t order);
/* int8 kernels: 0 = automatic (operands staged chunk-major / padded channel-blocked + 16-byte LDS-DMA MFMA kernel whenever it
 * covers the call), 1 = generic byte-gather kernel only.  Both produce the reference's bits. */
int32_t rten_hip_set_int8_path(rten_hip_ctx *ctx, int32_t mode);
/* int8 convolution / GEMM workgroup tile (tuning knob, sticky, default -1; v7): -1 = the backend's per-shape rule (the largest tile that still gives every compute
 * unit a workgroup), 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64.  `previous` (optional) receives the value it replaces, so that a scope can put it back.  Every
 * tile computes the same integer sums: a launch plan's per-layer entry for an int8 convolution step (profiles/plans/int8*.json: "<step>": [tile, 0, 1, 0]) changes time only. */
int32_t rten_hip_set_int8_tile(rten_hip_ctx *ctx, int32_t tile, int32_t *previous);
/* attention: 0 = automatic (one fused kernel -- QK^T, mask, softmax and PV without a score tensor in memory -- for head size 32 / 64 / 128
 * and key length <= 128, where it is the faster form), 1 = composed path only (batched GEMM, row softmax, batched GEMM), 2 = the fused
 * kernel wherever it covers the shape (head size 32 / 64 / 128, key length <= 512).  Same bits on every path. */
int32_t rten_hip_set_sdpa_path(rten_hip_ctx *ctx, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* RTEN_HIP_H */
