// C++ host-layer tests: the operators of include/rten_hip_ops.hpp against the CPU oracle (oracle/rten_oracle.c), bit for bit,
// plus the reference's error messages.  `--host-only` runs only the validation tests (no GPU needed).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "rten_hip_ops.hpp"

extern "C" {
void rto_rng_f32(uint64_t *state, int64_t n, float *out);
void rto_rng_u8(uint64_t *state, int64_t n, uint8_t *out);
void rto_rng_i8_reduced(uint64_t *state, int64_t n, int8_t *out);
int rto_conv2d_f32(int64_t N, int64_t C, int64_t H, int64_t W, int64_t O, int64_t kh, int64_t kw, const int64_t pads[4], const int64_t strides[2],
                   const int64_t dil[2], int64_t groups, const float *X, const float *Wt, const float *bias, const float *residual, int relu, float *Y,
                   int64_t OH, int64_t OW);
void rto_gemm_f32(int64_t M, int64_t N, int64_t K, const float *A, int64_t a_rs, int64_t a_cs, const float *B, int64_t b_rs, int64_t b_cs, float *C, int64_t ldc,
                  float alpha, float beta, const float *bias, int bias_kind);
void rto_gemm_int8(int64_t M, int64_t N, int64_t K, const void *A, int a_signed, int64_t a_rs, int64_t a_cs, const void *B, int b_signed, int64_t b_rs,
                   int64_t b_cs, int32_t *C, int64_t ldc, const void *a_zp, int64_t a_zp_len, const void *b_zp, int64_t b_zp_len, int beta);
int rto_conv2d_int8(int64_t N, int64_t C, int64_t H, int64_t W, int64_t O, int64_t kh, int64_t kw, const int64_t pads[4], const int64_t strides[2],
                    const int64_t dil[2], int64_t groups, const void *X, int x_signed, const void *Wt, int w_signed, int32_t x_zp, const void *w_zp,
                    int64_t w_zp_len, int pad_mode, int32_t *Y, int64_t OH, int64_t OW);
void rto_cast_scale(int64_t n, const int32_t *x, const float *scale, int64_t scale_len, float *y);
void rto_dynamic_quantize_linear(int64_t n, const float *x, uint8_t *y, float *scale_out, uint8_t *zp_out);
void rto_gelu(int64_t n, const float *x, float *y);
void rto_softmax(int64_t rows, int64_t cols, const float *x, const float *addend, int64_t add_div, int64_t add_mod, float *y, int flush_nan, int lanes);
void rto_layer_norm(int64_t rows, int64_t cols, const float *x, const float *gamma, const float *beta, float gamma_scalar, float beta_scalar, float eps, float *y,
                    int lanes);
void rto_pool2d(int64_t N, int64_t C, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t pt, int64_t pl, int64_t OH, int64_t OW,
                const float *x, float *y, int is_max, int count_include_pad);
void rto_global_avg_pool(int64_t NC, int64_t inner, const float *x, float *y, int lanes);
float rto_simd_sum(const float *x, int64_t n, int lanes);
}

using namespace rten_hip;
static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, what); failures++; } } while (0)

static std::vector<float> randf(uint64_t seed, int64_t n, float shift = 0.5f) {
    std::vector<float> v((size_t)n);
    uint64_t st = seed;
    rto_rng_f32(&st, n, v.data());
    for (auto &x : v) x -= shift;
    return v;
}
static bool same_bits(const std::vector<float> &a, const std::vector<float> &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (!(a[i] == b[i] || (std::isnan(a[i]) && std::isnan(b[i])))) { std::printf("  first difference at %zu: %.9g vs %.9g\n", i, a[i], b[i]); return false; }
    return true;
}
template <typename F> static void expect_error(OpError::Kind kind, const char *msg, F f, const char *what) {
    try { f(); CHECK(false, what); }
    catch (const OpError &e) { CHECK(e.kind == kind && e.msg == msg, what); if (!(e.kind == kind && e.msg == msg)) std::printf("  got %s\n", e.what()); }
}

static void host_only_tests() {
    // messages asserted by the reference's own tests: src/ops/conv.rs:1182-1268, src/ops/pooling.rs
    Conv conv;
    expect_error(OpError::InvalidValue, "input must have 4 dims (NCHW)", [&] { conv.geometry({1, 2}, {1, 1, 3, 3}); }, "conv input dims");
    expect_error(OpError::InvalidValue, "kernel must have 4 dims (OCHW)", [&] { conv.geometry({1, 4, 8, 8}, {4, 3, 3}); }, "conv kernel dims");
    expect_error(OpError::IncompatibleInputShapes, "Input channels (per group) does not match kernel input channels", [&] { conv.geometry({1, 5, 8, 8}, {4, 4, 3, 3}); }, "conv channels");
    conv.groups = 0;
    expect_error(OpError::InvalidValue, "Group count must be > 0", [&] { conv.geometry({1, 4, 8, 8}, {4, 4, 3, 3}); }, "conv groups 0");
    conv.groups = 3;
    expect_error(OpError::InvalidValue, "Input channel count not divisible by groups", [&] { conv.geometry({1, 4, 8, 8}, {4, 4, 3, 3}); }, "conv groups");
    conv.groups = 1;
    conv.padding = Padding::Fixed({1, 1});
    expect_error(OpError::InvalidValue, "Expected 4 padding values", [&] { conv.geometry({1, 4, 8, 8}, {4, 4, 3, 3}); }, "conv pads");
    conv.padding = Padding::Fixed({0, 0, 0, 0});
    conv.strides = {0, 0};
    expect_error(OpError::InvalidValue, "Strides must be > 0", [&] { conv.geometry({1, 4, 8, 8}, {4, 4, 3, 3}); }, "conv strides 0");
    conv.strides = {1, 1};
    const rten_hip_conv2d_desc d = conv.geometry({2, 4, 9, 11}, {6, 4, 3, 3});
    CHECK(d.out_h == 7 && d.out_w == 9, "conv output size");
    Conv same;
    same.padding = Padding::Same();
    same.strides = {2, 2};
    const rten_hip_conv2d_desc ds = same.geometry({1, 4, 9, 9}, {4, 4, 3, 3});
    CHECK(ds.out_h == 5 && ds.out_w == 5, "conv same padding output size");
    const OpRegistry reg = OpRegistry::with_all_ops();
    CHECK(reg.contains("Conv") && reg.contains("MatMulInteger") && reg.contains("Einsum") && reg.contains("Where") && reg.contains("Tanh") && !reg.contains("Loop"), "registry contents");
    CHECK(std::string(reg.create("LayerNormalization")->name()) == "LayerNormalization", "registry create");
    expect_error(OpError::UnsupportedValue, "operator not registered: Loop", [&] { reg.create("Loop"); }, "registry missing op");
}


// Einsum: parser, validation and path tables of the reference (einsum_parser.rs:277-557, einsum.rs:971-1293,1313-1498)
static void einsum_host_tests() {
    namespace E = rten_hip::einsum_detail;
    auto parsed = [](const char *eq) { std::vector<std::string> t; std::string o; E::parse_equation(eq, t, o); std::string j; for (auto &x : t) j += x + ","; return j + "->" + o; };
    CHECK(parsed(" i j , j k -> i k ") == "ij,jk,->ik", "einsum parse whitespace");
    CHECK(parsed("ij,jk") == "ij,jk,->ik", "einsum implicit output");
    CHECK(parsed("aBc") == "aBc,->Bac", "einsum implicit output ASCII order");
    CHECK(parsed("...ij") == "...ij,->...ij", "einsum implicit ellipsis");
    CHECK(parsed("") == ",->" && parsed("->") == ",->", "einsum empty equation");
    expect_error(OpError::InvalidValue, "Input term is invalid", [&] { parsed("i1j"); }, "einsum digit label");
    expect_error(OpError::InvalidValue, "Input term is invalid", [&] { parsed("i...j..."); }, "einsum two ellipses");
    expect_error(OpError::InvalidValue, "Output term is invalid", [&] { parsed("ij,jk->i.k"); }, "einsum bad output");
    expect_error(OpError::InvalidValue, "Einsum output term contains repeated labels", [&] { parsed("ij->ii"); }, "einsum repeated output");
    expect_error(OpError::InvalidValue, "Einsum output term contains a label not present in any input term", [&] { parsed("ij,jk->IK"); }, "einsum unknown label");
    auto bnd = [](const char *eq, std::vector<int> nd) { std::vector<std::string> t; std::string o; E::parse_equation(eq, t, o); return E::broadcast_ndim(t, nd); };
    CHECK(bnd("i...j->j...i", {5}) == 3 && bnd("ij,jk", {2, 2}) == 0, "einsum broadcast dims");
    expect_error(OpError::InvalidValue, "Number of terms in Einsum equation does not match input tensor count", [&] { bnd("ij,jk->ik", {2}); }, "einsum input count");
    expect_error(OpError::InvalidValue, "Einsum term dimension count does not match input tensor", [&] { bnd("ij", {1}); }, "einsum rank");
    expect_error(OpError::UnsupportedValue, "Einsum input or term has too many dimensions", [&] { bnd("...", {11}); }, "einsum too many dims");
    expect_error(OpError::InvalidValue, "Number of broadcast dims does not match across inputs", [&] { bnd("...,...->...", {1, 2}); }, "einsum broadcast mismatch");
    auto path = [](const char *eq, int b) {
        std::vector<std::string> t; std::string o; E::parse_equation(eq, t, o);
        std::string j;
        for (auto &s : E::plan_path(t, o, b)) j += s.lhs + "@" + std::to_string(s.lhs_src) + (s.binary ? "," + s.rhs + "@" + std::to_string(s.rhs_src) : "") + "->" + s.out + ";";
        return j;
    };
    CHECK(path("ab,bc,cd,de->ea", 0) == "ab@0,bc@1->ac;ac@-1,cd@2->ad;ad@-1,de@3->ea;", "einsum path chain");
    CHECK(path("ab,cd,ef", 0) == "ab@0,cd@1->abcd;abcd@-1,ef@2->abcdef;", "einsum path outer");
    CHECK(path("ii,j,i->", 0) == "ii@0,j@1->i;i@-1,i@2->;", "einsum path repeated label kept");
    CHECK(path("ii,i,j->", 0) == "ii@0,i@1->;@-1,j@2->;", "einsum path repeated label dropped");
    CHECK(path("i...j->j...i", 3) == "i012j@0->j012i;", "einsum path ellipsis");
    CHECK(path("...i,...j,...k->...ijk", 2) == "01i@0,01j@1->01ij;01ij@-1,01k@2->01ijk;", "einsum path ellipsis chain");
}

static void device_tests() {
    Context ctx(0);
    const int64_t one2[2] = {1, 1};
    { // Conv + bias + residual + relu, plain and prepacked
        const int64_t N = 2, C = 8, H = 9, W = 11, O = 6, k = 3, OH = 9, OW = 11, pads[4] = {1, 1, 1, 1};
        auto x = randf(1, N * C * H * W), w = randf(2, O * C * k * k), b = randf(3, O), r = randf(4, N * O * OH * OW);
        std::vector<float> want((size_t)(N * O * OH * OW));
        rto_conv2d_f32(N, C, H, W, O, k, k, pads, one2, one2, 1, x.data(), w.data(), b.data(), r.data(), 1, want.data(), OH, OW);
        Tensor tx = Tensor::from_host(ctx, {N, C, H, W}, x.data()), tw = Tensor::from_host(ctx, {O, C, k, k}, w.data()), tb = Tensor::from_host(ctx, {O}, b.data()),
               tr = Tensor::from_host(ctx, {N, O, OH, OW}, r.data());
        Conv conv;
        conv.padding = Padding::Fixed({1, 1, 1, 1});
        conv.fuse_relu = true;
        CHECK(same_bits(conv.run(ctx, {&tx, &tw, &tb, &tr})[0].to_host<float>(), want), "Conv bits");
        Tensor packed = conv.prepack(ctx, tw);
        CHECK(same_bits(conv.run_packed(ctx, {&tx, &tw, &tb, &tr}, &packed)[0].to_host<float>(), want), "Conv prepacked bits");
        Tensor bad = Tensor::from_host(ctx, {O + 1}, r.data());
        expect_error(OpError::IncompatibleInputShapes, "bias.size(0) != out_channels", [&] { conv.run(ctx, {&tx, &tw, &bad}); }, "conv bias size");
        expect_error(OpError::MissingInputs, "", [&] { conv.run(ctx, {&tx}); }, "conv missing input");
    }
    { // MatMul (K > 256: two depth blocks), FusedMatMul, Gemm(transB) with a row-vector C
        const int64_t M = 70, K = 300, Nn = 130;
        auto a = randf(5, M * K), b = randf(6, K * Nn), bias = randf(7, Nn);
        std::vector<float> want((size_t)(M * Nn)), want2(want.size()), want3(want.size());
        rto_gemm_f32(M, Nn, K, a.data(), K, 1, b.data(), Nn, 1, want.data(), Nn, 1.f, 0.f, nullptr, 0);
        rto_gemm_f32(M, Nn, K, a.data(), K, 1, b.data(), Nn, 1, want2.data(), Nn, 0.5f, 0.f, bias.data(), 2);
        Tensor ta = Tensor::from_host(ctx, {M, K}, a.data()), tb = Tensor::from_host(ctx, {K, Nn}, b.data()), tbias = Tensor::from_host(ctx, {Nn}, bias.data());
        CHECK(same_bits(MatMul().run(ctx, {&ta, &tb})[0].to_host<float>(), want), "MatMul bits");
        FusedMatMul fm;
        fm.alpha = 0.5f;
        CHECK(same_bits(fm.run(ctx, {&ta, &tb, &tbias})[0].to_host<float>(), want2), "FusedMatMul bits");
        std::vector<float> bt((size_t)(K * Nn)); // B^T [N, K]
        for (int64_t i = 0; i < K; i++) for (int64_t j = 0; j < Nn; j++) bt[(size_t)(j * K + i)] = b[(size_t)(i * Nn + j)];
        for (int64_t i = 0; i < M; i++) for (int64_t j = 0; j < Nn; j++) want3[(size_t)(i * Nn + j)] = bias[(size_t)j];
        rto_gemm_f32(M, Nn, K, a.data(), K, 1, bt.data(), 1, K, want3.data(), Nn, 1.f, 1.f, nullptr, 0);
        Tensor tbt = Tensor::from_host(ctx, {Nn, K}, bt.data());
        Gemm g;
        g.transpose_b = true;
        CHECK(same_bits(g.run(ctx, {&ta, &tbt, &tbias})[0].to_host<float>(), want3), "Gemm bits");
        Tensor t3 = Tensor::from_host(ctx, {K + 1, Nn}, b.data());
        expect_error(OpError::IncompatibleInputShapes, "Columns of first matrix does not match rows of second matrix", [&] { MatMul().run(ctx, {&ta, &t3}); }, "matmul shapes");
    }
    { // Gemm with ONE row and a row-vector C: the reference's gemv kernels, where C enters with the first depth block (matmul.rs:63-82,
      // rten-gemm/src/lib.rs:668-747) -- the batch-1 classifier of ResNet-50 -- with and without transB
        const int64_t K = 1100, Nn = 1000;
        auto a = randf(31, K), b = randf(32, K * Nn), bias = randf(33, Nn);
        std::vector<float> bt((size_t)(K * Nn));
        for (int64_t i = 0; i < K; i++) for (int64_t j = 0; j < Nn; j++) bt[(size_t)(j * K + i)] = b[(size_t)(i * Nn + j)];
        Tensor ta = Tensor::from_host(ctx, {1, K}, a.data()), tb = Tensor::from_host(ctx, {K, Nn}, b.data()), tbt = Tensor::from_host(ctx, {Nn, K}, bt.data()),
               tbias = Tensor::from_host(ctx, {Nn}, bias.data());
        std::vector<float> want(bias), want_t(bias);
        rto_gemm_f32(1, Nn, K, a.data(), K, 1, b.data(), Nn, 1, want.data(), Nn, 1.f, 1.f, nullptr, 0);
        rto_gemm_f32(1, Nn, K, a.data(), K, 1, bt.data(), 1, K, want_t.data(), Nn, 1.f, 1.f, nullptr, 0);
        Gemm g;
        CHECK(same_bits(g.run(ctx, {&ta, &tb, &tbias})[0].to_host<float>(), want), "Gemm one row bits");
        g.transpose_b = true;
        CHECK(same_bits(g.run(ctx, {&ta, &tbt, &tbias})[0].to_host<float>(), want_t), "Gemm one row transB bits");
    }
    { // Softmax, LayerNormalization, Gelu
        const int64_t R = 37, Cc = 100;
        auto x = randf(8, R * Cc), g = randf(9, Cc), b = randf(10, Cc);
        std::vector<float> want((size_t)(R * Cc));
        Tensor tx = Tensor::from_host(ctx, {R, Cc}, x.data()), tg = Tensor::from_host(ctx, {Cc}, g.data()), tb = Tensor::from_host(ctx, {Cc}, b.data());
        rto_softmax(R, Cc, x.data(), nullptr, 1, 1, want.data(), 0, 16);
        CHECK(same_bits(Softmax().run(ctx, {&tx})[0].to_host<float>(), want), "Softmax bits (AVX-512 order)");
        rto_layer_norm(R, Cc, x.data(), g.data(), b.data(), 1.f, 0.f, 1e-5f, want.data(), 16);
        CHECK(same_bits(LayerNormalization().run(ctx, {&tx, &tg, &tb})[0].to_host<float>(), want), "LayerNormalization bits (AVX-512 order)");
        rto_gelu(R * Cc, x.data(), want.data());
        CHECK(same_bits(Gelu().run(ctx, {&tx})[0].to_host<float>(), want), "Gelu bits");
        Softmax sm;
        sm.axis = 2;
        expect_error(OpError::InvalidValue, "Axis is invalid", [&] { sm.run(ctx, {&tx}); }, "softmax axis");
    }
    { // DynamicQuantizeLinear -> ConvIntegerToFloat(+bias, relu); MatMulInteger
        const int64_t N = 2, C = 16, H = 14, W = 14, O = 24, k = 3, pads[4] = {1, 1, 1, 1};
        auto x = randf(11, N * C * H * W, 0.3f), bias = randf(12, O);
        std::vector<int8_t> w((size_t)(O * C * k * k));
        uint64_t st = 13;
        rto_rng_i8_reduced(&st, (int64_t)w.size(), w.data());
        std::vector<uint8_t> q(x.size());
        float s;
        uint8_t z;
        rto_dynamic_quantize_linear((int64_t)x.size(), x.data(), q.data(), &s, &z);
        std::vector<int32_t> acc((size_t)(N * O * H * W));
        rto_conv2d_int8(N, C, H, W, O, k, k, pads, one2, one2, 1, q.data(), 0, w.data(), 1, z, nullptr, 0, RTEN_HIP_PAD_RAW0_I8, acc.data(), H, W);
        const float sc = s * 0.004f;
        std::vector<float> want(acc.size());
        rto_cast_scale((int64_t)acc.size(), acc.data(), &sc, 1, want.data());
        for (int64_t i = 0; i < (int64_t)want.size(); i++) { float v = want[(size_t)i] + bias[(size_t)((i / (H * W)) % O)]; want[(size_t)i] = v > 0.f ? v : 0.f; }
        Tensor tx = Tensor::from_host(ctx, {N, C, H, W}, x.data()), tw = Tensor::from_host(ctx, {O, C, k, k}, w.data()), tbias = Tensor::from_host(ctx, {O}, bias.data()),
               tsc = Tensor::from_host(ctx, {}, &sc);
        OutputList dq = DynamicQuantizeLinear().run(ctx, {&tx});
        CHECK(dq[1].to_host<float>()[0] == s && dq[2].to_host<uint8_t>()[0] == z, "DynamicQuantizeLinear scale / zero point");
        CHECK(dq[0].to_host<uint8_t>() == q, "DynamicQuantizeLinear codes");
        ConvIntegerToFloat cf;
        cf.conv.conv.padding = Padding::Fixed({1, 1, 1, 1});
        cf.fuse_relu = true;
        CHECK(same_bits(cf.run(ctx, {&dq[0], &tw, &dq[2], nullptr, &tsc, &tbias})[0].to_host<float>(), want), "ConvIntegerToFloat bits");
        std::vector<int32_t> wacc(acc.size());
        ConvInteger ci;
        ci.conv.padding = Padding::Fixed({1, 1, 1, 1});
        CHECK(ci.run(ctx, {&dq[0], &tw, &dq[2]})[0].to_host<int32_t>() == acc, "ConvInteger bits");
        const int64_t M = 33, K = 70, Nn = 20;
        std::vector<uint8_t> a((size_t)(M * K)), azp(1, 7);
        std::vector<int8_t> bm((size_t)(K * Nn));
        rto_rng_u8(&st, M * K, a.data());
        rto_rng_i8_reduced(&st, K * Nn, bm.data());
        std::vector<int32_t> mwant((size_t)(M * Nn));
        rto_gemm_int8(M, Nn, K, a.data(), 0, K, 1, bm.data(), 1, Nn, 1, mwant.data(), Nn, azp.data(), 1, nullptr, 0, 0);
        Tensor ta = Tensor::from_host(ctx, {M, K}, a.data()), tbm = Tensor::from_host(ctx, {K, Nn}, bm.data()), tz = Tensor::from_host(ctx, {}, azp.data());
        CHECK(MatMulInteger().run(ctx, {&ta, &tbm, &tz})[0].to_host<int32_t>() == mwant, "MatMulInteger bits");
        // Operator::prepack (matmul.rs:696-705): the staged RHS gives the same bits
        const MatMulInteger mmi;
        const Tensor packed = mmi.prepack(ctx, tbm);
        CHECK(packed.len() == (int64_t)rten_hip_gemm_int8_packed_bytes((int32_t)K, (int32_t)Nn) && packed.len() > 0, "MatMulInteger prepack size");
        CHECK(mmi.run_scaled(ctx, {&ta, &tbm, &tz}, nullptr, &packed)[0].to_host<int32_t>() == mwant, "MatMulInteger prepacked bits");
        // [A, M, K] x [K, N] collapses to one product and the row zero points cycle with period M (matmul.rs:259-296);
        // a batched RHS runs as batched_gemm_uninit (:302-372)
        const int64_t A3 = 3, M3 = 11;
        std::vector<uint8_t> a3((size_t)(A3 * M3 * K)), azv((size_t)M3);
        rto_rng_u8(&st, A3 * M3 * K, a3.data());
        rto_rng_u8(&st, M3, azv.data());
        std::vector<int8_t> b3((size_t)(A3 * K * Nn)), bzv((size_t)Nn);
        rto_rng_i8_reduced(&st, A3 * K * Nn, b3.data());
        rto_rng_i8_reduced(&st, Nn, bzv.data());
        std::vector<int32_t> want1((size_t)(A3 * M3 * Nn)), want2(want1.size());
        for (int64_t z = 0; z < A3; z++) {
            rto_gemm_int8(M3, Nn, K, a3.data() + z * M3 * K, 0, K, 1, bm.data(), 1, Nn, 1, want1.data() + z * M3 * Nn, Nn, azv.data(), M3, bzv.data(), Nn, 0);
            rto_gemm_int8(M3, Nn, K, a3.data() + z * M3 * K, 0, K, 1, b3.data() + z * K * Nn, 1, Nn, 1, want2.data() + z * M3 * Nn, Nn, azv.data(), M3, bzv.data(), Nn, 0);
        }
        Tensor ta3 = Tensor::from_host(ctx, {A3, M3, K}, a3.data()), tb3 = Tensor::from_host(ctx, {A3, K, Nn}, b3.data()), taz = Tensor::from_host(ctx, {M3}, azv.data()),
               tbz = Tensor::from_host(ctx, {Nn}, bzv.data());
        OutputList y1 = mmi.run(ctx, {&ta3, &tbm, &taz, &tbz});
        CHECK((y1[0].shape() == std::vector<int64_t>{A3, M3, Nn}) && y1[0].to_host<int32_t>() == want1, "MatMulInteger [A,M,K]x[K,N] with cycled zero points");
        CHECK(mmi.run_scaled(ctx, {&ta3, &tbm, &taz, &tbz}, nullptr, &packed)[0].to_host<int32_t>() == want1, "MatMulInteger collapsed + prepacked");
        CHECK(mmi.run(ctx, {&ta3, &tb3, &taz, &tbz})[0].to_host<int32_t>() == want2, "MatMulInteger batched RHS");
        Tensor tbad = Tensor::from_host(ctx, {2, K, Nn}, b3.data());
        expect_error(OpError::IncompatibleInputShapes, "Cannot broadcast shapes", [&] { mmi.run(ctx, {&ta3, &tbad}); }, "MatMulInteger broadcast error");
        Tensor tzbad = Tensor::from_host(ctx, {A3}, azv.data());
        expect_error(OpError::InvalidValue, "Zero point has incorrect size", [&] { mmi.run(ctx, {&ta3, &tbm, &tzbad}); }, "MatMulInteger zero point size");
    }
    { // MaxPool, GlobalAveragePool
        const int64_t N = 2, C = 5, H = 12, W = 10;
        auto x = randf(14, N * C * H * W);
        Tensor tx = Tensor::from_host(ctx, {N, C, H, W}, x.data());
        MaxPool mp;
        mp.kernel_size = {3, 3};
        mp.strides = {2, 2};
        mp.padding = Padding::Fixed({1, 1, 1, 1});
        OutputList y = mp.run(ctx, {&tx});
        const int64_t OH = y[0].size(2), OW = y[0].size(3);
        std::vector<float> want((size_t)(N * C * OH * OW));
        rto_pool2d(N, C, H, W, 3, 3, 2, 2, 1, 1, OH, OW, x.data(), want.data(), 1, 0);
        CHECK(same_bits(y[0].to_host<float>(), want), "MaxPool bits");
        std::vector<float> gw((size_t)(N * C));
        rto_global_avg_pool(N * C, H * W, x.data(), gw.data(), 16);
        CHECK(same_bits(GlobalAveragePool().run(ctx, {&tx})[0].to_host<float>(), gw), "GlobalAveragePool bits");
    }
    { // Einsum / ReduceSum through strides, against compositions of the oracle's GEMM and ordered sum
        const int64_t B = 2, S = 17, T = 19, H = 3, D = 8;
        auto q = randf(31, B * S * H * D), kk = randf(32, B * T * H * D);
        Tensor tq = Tensor::from_host(ctx, {B, S, H, D}, q.data()), tk = Tensor::from_host(ctx, {B, T, H, D}, kk.data());
        Einsum scores;
        scores.equation = "bqhd,bkhd->bhqk"; // un-transposed [B, S, H, D] heads: one strided two-level batched GEMM
        OutputList y = scores.run(ctx, {&tq, &tk});
        CHECK(y[0].shape() == (std::vector<int64_t>{B, H, S, T}), "Einsum attention-scores shape");
        std::vector<float> want((size_t)(B * H * S * T));
        for (int64_t b = 0; b < B; b++)
            for (int64_t h = 0; h < H; h++)
                rto_gemm_f32(S, T, D, q.data() + b * S * H * D + h * D, H * D, 1, kk.data() + b * T * H * D + h * D, 1, H * D, want.data() + (b * H + h) * S * T, T, 1.0f,
                             0.0f, nullptr, 0);
        CHECK(same_bits(y[0].to_host<float>(), want), "Einsum bqhd,bkhd->bhqk bits");
        { // second attention product with the permuted output [B, S, H, D] written through the GEMM's C strides (no copy)
            auto vv = randf(35, B * T * H * D);
            Tensor tv = Tensor::from_host(ctx, {B, T, H, D}, vv.data());
            Einsum pv;
            pv.equation = "bhqk,bkhd->bqhd";
            OutputList o = pv.run(ctx, {&y[0], &tv});
            CHECK(o[0].shape() == (std::vector<int64_t>{B, S, H, D}), "Einsum attention-context shape");
            std::vector<float> wo((size_t)(B * S * H * D));
            for (int64_t b = 0; b < B; b++)
                for (int64_t h = 0; h < H; h++)
                    rto_gemm_f32(S, D, T, want.data() + (b * H + h) * S * T, T, 1, vv.data() + b * T * H * D + h * D, H * D, 1, wo.data() + b * S * H * D + h * D, H * D,
                                 1.0f, 0.0f, nullptr, 0);
            CHECK(same_bits(o[0].to_host<float>(), wo), "Einsum bhqk,bkhd->bqhd bits");
        }
        const int64_t M = 70, K = 513, N = 45;
        auto a = randf(33, M * K), bb = randf(34, N * K);
        Tensor ta = Tensor::from_host(ctx, {M, K}, a.data()), tb = Tensor::from_host(ctx, {N, K}, bb.data());
        Einsum abt;
        abt.equation = "ik,jk->ji"; // transposed RHS as strides, transposed output as the one copy
        std::vector<float> c((size_t)(M * N)), ct((size_t)(M * N));
        rto_gemm_f32(M, N, K, a.data(), K, 1, bb.data(), 1, K, c.data(), N, 1.0f, 0.0f, nullptr, 0);
        for (int64_t i = 0; i < M; i++) for (int64_t j = 0; j < N; j++) ct[(size_t)(j * M + i)] = c[(size_t)(i * N + j)];
        CHECK(same_bits(abt.run(ctx, {&ta, &tb})[0].to_host<float>(), ct), "Einsum ik,jk->ji bits");
        Einsum colsum;
        colsum.equation = "ij->j"; // reduction over the strided axis, in place
        std::vector<float> cs((size_t)K), col((size_t)M);
        for (int64_t j = 0; j < K; j++) { for (int64_t i = 0; i < M; i++) col[(size_t)i] = a[(size_t)(i * K + j)]; cs[(size_t)j] = rto_simd_sum(col.data(), M, 16); }
        CHECK(same_bits(colsum.run(ctx, {&ta})[0].to_host<float>(), cs), "Einsum ij->j bits");
        ReduceSum rs;
        rs.axes = {-1};
        rs.keep_dims = false;
        std::vector<float> rsum((size_t)M);
        for (int64_t i = 0; i < M; i++) rsum[(size_t)i] = rto_simd_sum(a.data() + i * K, K, 16);
        CHECK(same_bits(rs.run(ctx, {&ta})[0].to_host<float>(), rsum), "ReduceSum bits");
        Einsum diag;
        diag.equation = "ii->i";
        expect_error(OpError::InvalidValue, "Dimension sizes for repeated labels in term do not match", [&] { diag.run(ctx, {&ta}); }, "einsum diagonal sizes");
        Einsum mism;
        mism.equation = "ij,jk->ik";
        expect_error(OpError::IncompatibleInputShapes, "Einsum label has different sizes in different terms", [&] { mism.run(ctx, {&ta, &tb}); }, "einsum label sizes");
    }
    ctx.sync();
}

int main(int argc, char **argv) {
    const bool host_only = argc > 1 && std::strcmp(argv[1], "--host-only") == 0;
    try {
        host_only_tests();
        einsum_host_tests();
        if (!host_only) device_tests();
    } catch (const OpError &e) {
        std::printf("FAIL unexpected OpError: %s\n", e.what());
        failures++;
    }
    std::printf(failures ? "%d FAILURES\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
