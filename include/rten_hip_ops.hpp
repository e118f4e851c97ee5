// rten_hip_ops.hpp -- C++ host side above the C ABI (include/rten_hip.h): the reference's operator interface for the hot
// path, mirrored in C++ because the reference is compiled code and its own toolchain (Rust) is not in this image.
//
// What is mirrored (same names, attribute meaning, input order, validation order and OpError messages, which the
// reference's tests assert verbatim -- src/ops/conv.rs:1182-1268, src/ops/matmul.rs:1284-1333):
//   * `Operator` / `OpRunContext` / `OpRegistry`   src/operator.rs:486-613, src/op_registry.rs:25-72
//   * `OpError`                                    src/operator.rs:116-144
//   * Conv, ConvInteger, ConvIntegerToFloat        src/ops/conv.rs:367-403, 478-587
//   * MatMul, FusedMatMul, Gemm, MatMulInteger(ToFloat)   src/ops/matmul.rs:106-156, 387-510, 582-810
//   * Softmax, LayerNormalization                  src/ops/norm.rs:456-529, 825-840
//   * Gelu, Erf, Relu, Add, Mul                    src/ops/unary_elementwise.rs, binary_elementwise.rs:476-495
//   * MaxPool, AveragePool, GlobalAveragePool      src/ops/pooling.rs:174-521
//   * DynamicQuantizeLinear                        src/ops/quantize.rs:352-436
//   * Einsum, ReduceSum                            src/ops/einsum.rs:21-692, src/ops/reduce.rs:414-520,1126-1165
// Validation runs on the host before any launch; arithmetic is done by librten_hip.so on device-resident tensors.  There
// is no CPU fallback: without a gfx950 device `Context` throws.  Header only; C++17.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "rten_hip.h"

namespace rten_hip {

// ---- OpError (src/operator.rs:116-144)
struct OpError : std::runtime_error {
    enum Kind { InvalidValue, IncompatibleInputShapes, UnsupportedValue, UnsupportedType, MissingInputs, InputCastFailed, BackendUnavailable, Hip };
    Kind kind;
    std::string msg;
    OpError(Kind k, std::string m) : std::runtime_error(kind_name(k) + (m.empty() ? "" : "(\"" + m + "\")")), kind(k), msg(std::move(m)) {}
    static std::string kind_name(Kind k) {
        static const char *n[] = {"InvalidValue", "IncompatibleInputShapes", "UnsupportedValue", "UnsupportedType", "MissingInputs", "InputCastFailed",
                                  "BackendUnavailable", "Hip"};
        return n[k];
    }
};

// ---- Context: RAII over rten_hip_ctx.  May be shared by host threads (rten_hip.h, "Thread safety"): the C ABI locks the
// context per call and the buffer pool below has its own mutex.
class Context {
  public:
    explicit Context(int device = 0, void *stream = nullptr) {
        const int32_t rc = rten_hip_init(device, stream, &h_);
        if (rc == RTEN_HIP_ERR_NO_DEVICE) throw OpError(OpError::BackendUnavailable, "no usable gfx950 (MI355X) device: the HIP backend has no CPU fallback");
        if (rc != RTEN_HIP_OK) throw OpError(OpError::Hip, "rten_hip_init failed");
    }
    // Borrows a context created elsewhere (a C-ABI caller's): same stream, same lock; never destroyed here.
    struct Borrow {};
    Context(rten_hip_ctx *borrowed, Borrow) : h_(borrowed), owned_(false) {
        if (!borrowed) throw OpError(OpError::InvalidValue, "null context");
    }
    ~Context() { if (h_) { trim_pool(); if (one_) rten_hip_free(h_, one_); if (zero_) rten_hip_free(h_, zero_); if (owned_) rten_hip_destroy(h_); } }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    rten_hip_ctx *raw() const { return h_; }
    void sync() { check(rten_hip_sync(h_)); }

    // Device buffer pool (src/buffer_pool.rs: operators take output buffers from the pool, the executor returns dead values
    // to it).  Off by default; a graph executor turns it on so that steady-state runs allocate nothing.  Buffers are handed
    // out by exact byte size; the stream orders reuse (one stream per context).
    void enable_pool(bool on = true) { pool_on_ = on; if (!on) trim_pool(); }
    void *alloc(size_t bytes) {
        if (pool_on_) {
            std::lock_guard<std::mutex> lk(pool_mu_);
            auto it = pool_.find(bytes);
            if (it != pool_.end() && !it->second.empty()) { void *p = it->second.back(); it->second.pop_back(); return p; }
        }
        void *p = nullptr;
        check(rten_hip_malloc(h_, bytes, &p));
        return p;
    }
    void release(void *p, size_t bytes) {
        if (!p) return;
        if (pool_on_) { std::lock_guard<std::mutex> lk(pool_mu_); pool_[bytes].push_back(p); }
        else rten_hip_free(h_, p);
    }
    void trim_pool() {
        std::lock_guard<std::mutex> lk(pool_mu_);
        for (auto &kv : pool_) for (void *p : kv.second) rten_hip_free(h_, p);
        pool_.clear();
    }
    const int32_t *zero_i32() { // device-resident int32 0 (x + 0: a raw-word move through the generic element-wise kernel)
        if (!zero_) {
            const int32_t v = 0;
            check(rten_hip_malloc(h_, 4, &zero_));
            check(rten_hip_memcpy_h2d(h_, zero_, &v, 4));
        }
        return (const int32_t *)zero_;
    }
    const float *one() { // device-resident 1.0f (x * 1.0f is exact: Cast as cast_scale)
        if (!one_) {
            const float v = 1.0f;
            check(rten_hip_malloc(h_, 4, &one_));
            check(rten_hip_memcpy_h2d(h_, one_, &v, 4));
        }
        return (const float *)one_;
    }
    // maps ABI status codes onto OpError variants (include/rten_hip.h, "status codes")
    void check(int32_t rc) const {
        if (rc == RTEN_HIP_OK) return;
        const std::string m = rten_hip_last_error(h_);
        switch (rc) {
        case RTEN_HIP_ERR_INVALID_VALUE: throw OpError(OpError::InvalidValue, m);
        case RTEN_HIP_ERR_INCOMPATIBLE_SHAPES: throw OpError(OpError::IncompatibleInputShapes, m);
        case RTEN_HIP_ERR_UNSUPPORTED: throw OpError(OpError::UnsupportedValue, m);
        default: throw OpError(OpError::Hip, m);
        }
    }

  private:
    rten_hip_ctx *h_ = nullptr;
    bool owned_ = true;
    bool pool_on_ = false;
    void *one_ = nullptr, *zero_ = nullptr;
    std::map<size_t, std::vector<void *>> pool_;
    std::mutex pool_mu_; // the pool is shared by concurrent Model::run callers (the C ABI context locks itself)
};

// ---- device tensor (the backend's `Value`: contiguous, row-major, device resident)
enum class DType { F32, I32, U8, I8 };
inline size_t dtype_size(DType t) { return (t == DType::F32 || t == DType::I32) ? 4 : 1; }

// A small tensor whose VALUES the host knows (shape arithmetic: the output of Shape, constants, and what Gather / Concat / Slice / Unsqueeze / arithmetic /
// comparisons make of them).  The reference folds such subgraphs at load (`propagate_constants`, src/optimize.rs:705, over its symbolic shapes); here
// the executor evaluates them on the host at run time (rten_hip_graph.hpp: no device work, nothing of it in a captured hipGraph) and attaches the
// result to the device tensor that carries the value, so that Reshape / Expand / Slice / ConstantOfShape read their shape operands without a
// device round trip.  Integers and booleans are int32-valued (onnx_loader.rs:332-339), kept in int64 for the arithmetic.
struct HostVal {
    std::vector<int64_t> shape;
    bool is_float = false;
    std::vector<int64_t> i;
    std::vector<float> f;
    int64_t len() const { int64_t n = 1; for (int64_t d : shape) n *= d; return n; }
    bool operator==(const HostVal &o) const {
        if (shape != o.shape || is_float != o.is_float || i != o.i || f.size() != o.f.size()) return false;
        return f.empty() || std::memcmp(f.data(), o.f.data(), f.size() * sizeof(float)) == 0; // bitwise (NaN payloads, -0)
    }
};

class Tensor {
  public:
    Tensor() = default;
    Tensor(Context &ctx, std::vector<int64_t> shape, DType dt) : ctx_(&ctx), shape_(std::move(shape)), dtype_(dt) {
        cap_ = bytes() ? bytes() : 4;
        ptr_ = ctx.alloc(cap_);
    }
    // Shape plus an explicit byte capacity (>= bytes()): opaque device layouts that are larger than the logical tensor,
    // e.g. the int8 kernel's zero-point-padded staged image of an [N, C, H, W] activation.
    Tensor(Context &ctx, std::vector<int64_t> shape, DType dt, size_t capacity) : ctx_(&ctx), shape_(std::move(shape)), dtype_(dt) {
        cap_ = std::max<size_t>(std::max(capacity, bytes()), 4);
        ptr_ = ctx.alloc(cap_);
    }
    // Non-owning alias of `base`'s storage with another shape (Reshape / Flatten / Squeeze as views, src/ops/layout.rs):
    // the caller keeps `base` alive for as long as the view is used.
    static Tensor view_of(const Tensor &base, std::vector<int64_t> shape) {
        Tensor t;
        t.ctx_ = base.ctx_; t.ptr_ = base.ptr_; t.shape_ = std::move(shape); t.dtype_ = base.dtype_; t.owns_ = false;
        return t;
    }
    // Non-owning alias of `bytes_off` bytes into `base` (a sub-batch slice of a resident full-batch buffer).
    static Tensor view_at(const Tensor &base, size_t bytes_off, std::vector<int64_t> shape) {
        Tensor t = view_of(base, std::move(shape));
        t.ptr_ = (char *)base.ptr_ + bytes_off;
        return t;
    }
    static Tensor view_at(const Tensor &base, size_t bytes_off, std::vector<int64_t> shape, DType dt) { // ... typed: a slot of a byte arena
        Tensor t = view_at(base, bytes_off, std::move(shape));
        t.dtype_ = dt;
        return t;
    }
    template <typename T>
    static Tensor from_host(Context &ctx, std::vector<int64_t> shape, const T *data) {
        Tensor t(ctx, std::move(shape), dtype_of<T>());
        if (t.bytes()) ctx.check(rten_hip_memcpy_h2d(ctx.raw(), t.ptr_, data, t.bytes()));
        return t;
    }
    ~Tensor() { release(); }
    Tensor(Tensor &&o) noexcept { *this = std::move(o); }
    Tensor &operator=(Tensor &&o) noexcept {
        if (this != &o) { release(); ctx_ = o.ctx_; ptr_ = o.ptr_; shape_ = std::move(o.shape_); dtype_ = o.dtype_; cap_ = o.cap_; owns_ = o.owns_; host_ = std::move(o.host_); uniform_ = o.uniform_; o.ptr_ = nullptr; }
        return *this;
    }
    Tensor(const Tensor &) = delete;
    Tensor &operator=(const Tensor &) = delete;

    const std::vector<int64_t> &shape() const { return shape_; }
    int64_t size(int i) const { return shape_[(size_t)i]; }
    int ndim() const { return (int)shape_.size(); }
    int64_t len() const { return std::accumulate(shape_.begin(), shape_.end(), (int64_t)1, std::multiplies<int64_t>()); }
    size_t bytes() const { return (size_t)len() * dtype_size(dtype_); }
    DType dtype() const { return dtype_; }
    void *ptr() const { return ptr_; }
    template <typename T> std::vector<T> to_host() const {
        std::vector<T> out((size_t)len());
        if (bytes()) ctx_->check(rten_hip_memcpy_d2h(ctx_->raw(), out.data(), ptr_, bytes())); // synchronises
        return out;
    }
    void reshape(std::vector<int64_t> s) { shape_ = std::move(s); uniform_ = 0; }
    // Bit d set = every slice along dim d holds the same values (the tensor is a broadcast along d that was materialised): what an element-wise step knows
    // about its result when each operand had size 1 there, was itself uniform there, or is a host value that is.  A consumer that broadcasts along d anyway may
    // then read ONE slice -- the exporter-written attention mask [B, 1, S, T] whose S rows are copies of a [B, 1, 1, T] row takes the fused attention kernel's
    // shared-mask form (round 6).  Fresh tensors and views start at 0 (nothing known).
    uint32_t uniform_dims() const { return uniform_; }
    void set_uniform_dims(uint32_t m) { uniform_ = m; }
    // the host's copy of the values, when it has one (see HostVal); views made with view_of() start without one
    const HostVal *host() const { return host_.get(); }
    const std::shared_ptr<const HostVal> &host_ptr() const { return host_; }
    void set_host(std::shared_ptr<const HostVal> h) { host_ = std::move(h); }

    template <typename T> static DType dtype_of() {
        if (std::is_same<T, float>::value) return DType::F32;
        if (std::is_same<T, int32_t>::value) return DType::I32;
        if (std::is_same<T, uint8_t>::value) return DType::U8;
        return DType::I8;
    }

  private:
    void release() { if (ptr_ && ctx_ && owns_) ctx_->release(ptr_, cap_); ptr_ = nullptr; }
    Context *ctx_ = nullptr;
    void *ptr_ = nullptr;
    size_t cap_ = 0;
    bool owns_ = true;
    std::vector<int64_t> shape_;
    DType dtype_ = DType::F32;
    std::shared_ptr<const HostVal> host_;
    uint32_t uniform_ = 0;
};

// ---- Operator interface (src/operator.rs:486-613).  Optional inputs are null pointers (InputList::get).
using InputList = std::vector<const Tensor *>;
using OutputList = std::vector<Tensor>;

inline const Tensor &require(const InputList &in, size_t i) {
    if (i >= in.size() || !in[i]) throw OpError(OpError::MissingInputs, "");
    return *in[i];
}
inline const Tensor *get(const InputList &in, size_t i) { return i < in.size() ? in[i] : nullptr; }
inline const Tensor &want(const Tensor &t, DType dt, const char *what) {
    if (t.dtype() != dt) throw OpError(OpError::InputCastFailed, std::string("expected ") + what + " tensor");
    return t;
}
inline void *vp(const Tensor *t) { return t ? t->ptr() : nullptr; }

class Operator {
  public:
    virtual ~Operator() = default;
    virtual const char *name() const = 0;
    virtual int max_inputs() const { return -1; } // < 0: unbounded (Operator::max_inputs -> None)
    virtual OutputList run(Context &ctx, const InputList &inputs) const = 0;
};

// Padding::Same / Padding::Fixed (src/ops/mod.rs)
struct Padding {
    bool same = false;
    std::vector<int> fixed{0, 0, 0, 0};
    static Padding Same() { Padding p; p.same = true; return p; }
    static Padding Fixed(std::vector<int> v) { Padding p; p.fixed = std::move(v); return p; }
};

struct OutputSize { int oh, ow; int pads[4]; };
// calc_output_size_and_padding (src/ops/pooling.rs:139-159): same error strings through the ABI
inline OutputSize calc_output_size_and_padding(int h, int w, int kh, int kw, const std::vector<int> &strides, const Padding &padding,
                                               const std::vector<int> &dilations = {1, 1}, bool ceil_mode = false) {
    if (!padding.same && padding.fixed.size() != 4) throw OpError(OpError::InvalidValue, "Expected 4 padding values");
    int32_t pads[4] = {0, 0, 0, 0}, out[2], opads[4];
    if (!padding.same) for (int i = 0; i < 4; i++) pads[i] = padding.fixed[(size_t)i];
    const char *msg = nullptr;
    const int32_t rc = rten_hip_calc_output_size_and_padding(h, w, kh, kw, strides[0], strides[1], padding.same ? 1 : 0, pads, dilations[0], dilations[1],
                                                             ceil_mode ? 1 : 0, out, opads, &msg);
    if (rc) throw OpError(OpError::InvalidValue, msg ? msg : "");
    return OutputSize{out[0], out[1], {opads[0], opads[1], opads[2], opads[3]}};
}

// Launch plan of the f32 GEMM / implicit-GEMM conv kernels: tile variant, exact split-K mode and group count, tile
// order (rten_hip_set_gemm_variant_override / _split / _order).  Unset = the backend's automatic choice.  An executor
// picks one per layer by measurement at load time, as the reference picks its kernel per ISA (rten-gemm/src/lib.rs:534-547).
struct GemmPlan {
    bool set = false;
    int variant = -1, mode = 3, groups = 1, order = 0;
};
class PlanScope { // applies a plan around one call and restores the automatic plan afterwards
  public:
    PlanScope(Context &ctx, const GemmPlan &p) : ctx_(ctx), on_(p.set) {
        if (!on_) return;
        ctx.check(rten_hip_tuning_save(ctx.raw(), saved_)); // (the context may be a caller's: its knobs come back as they were, not as defaults)
        ctx.check(rten_hip_set_gemm_variant_override(ctx.raw(), p.variant));
        ctx.check(rten_hip_set_gemm_split(ctx.raw(), p.mode, p.groups));
        ctx.check(rten_hip_set_gemm_order(ctx.raw(), p.order));
    }
    ~PlanScope() {
        if (!on_) return;
        rten_hip_tuning_restore(ctx_.raw(), saved_);
    }

  private:
    Context &ctx_;
    bool on_;
    int32_t saved_[8] = {};
};

// ------------------------------------------------------------------------------------------------ Conv
struct Conv : Operator {
    GemmPlan plan;
    int groups = 1;
    std::vector<int> dilations{1, 1};
    Padding padding;
    std::vector<int> strides{1, 1};
    bool fuse_relu = false; // backend fusion of the following Relu; a 4th input is the residual Add operand

    const char *name() const override { return "Conv"; }
    int max_inputs() const override { return 4; }

    // shape checks in the reference's order with its messages (src/ops/conv.rs:136-214)
    rten_hip_conv2d_desc geometry(const std::vector<int64_t> &xs, const std::vector<int64_t> &ws) const {
        if (xs.size() != 4) throw OpError(OpError::InvalidValue, "input must have 4 dims (NCHW)");
        if (ws.size() != 4) throw OpError(OpError::InvalidValue, "kernel must have 4 dims (OCHW)");
        if (strides.size() != 2) throw OpError(OpError::InvalidValue, "expected 2 stride values");
        if (dilations.size() != 2) throw OpError(OpError::InvalidValue, "expected 2 dilation values");
        const int n = (int)xs[0], c = (int)xs[1], h = (int)xs[2], w = (int)xs[3];
        const int o = (int)ws[0], kc = (int)ws[1], kh = (int)ws[2], kw = (int)ws[3];
        const OutputSize os = calc_output_size_and_padding(h, w, kh, kw, strides, padding, dilations);
        if (groups == 0) throw OpError(OpError::InvalidValue, "Group count must be > 0");
        if (c % groups != 0) throw OpError(OpError::InvalidValue, "Input channel count not divisible by groups");
        if (c / groups != kc) throw OpError(OpError::IncompatibleInputShapes, "Input channels (per group) does not match kernel input channels");
        if (o % groups != 0) throw OpError(OpError::InvalidValue, "Output channel count not divisible by groups");
        rten_hip_conv2d_desc d{};
        d.n = n; d.c = c; d.h = h; d.w = w; d.o = o; d.kh = kh; d.kw = kw;
        for (int i = 0; i < 4; i++) d.pads[i] = os.pads[i];
        d.stride_h = strides[0]; d.stride_w = strides[1]; d.dil_h = dilations[0]; d.dil_w = dilations[1];
        d.groups = groups; d.out_h = os.oh; d.out_w = os.ow;
        return d;
    }

    // PrepackedInput analogue (src/operator.rs:25-66): stage the constant weight once
    Tensor prepack(Context &ctx, const Tensor &weight) const {
        rten_hip_conv2d_desc d{};
        d.n = 1; d.c = (int)weight.size(1) * groups; d.h = d.w = 1; d.o = (int)weight.size(0); d.kh = (int)weight.size(2); d.kw = (int)weight.size(3);
        d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1; d.groups = groups; d.out_h = d.out_w = 1;
        Tensor packed(ctx, {(int64_t)(rten_hip_conv2d_f32_packed_bytes(&d) / 4)}, DType::F32);
        ctx.check(rten_hip_conv2d_f32_prepack(ctx.raw(), &d, (const float *)weight.ptr(), (float *)packed.ptr()));
        return packed;
    }

    OutputList run(Context &ctx, const InputList &in) const override { return run_packed(ctx, in, nullptr); }
    OutputList run_packed(Context &ctx, const InputList &in, const Tensor *packed_weight) const {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        const Tensor &w = want(require(in, 1), DType::F32, "float32");
        const Tensor *bias = get(in, 2), *residual = get(in, 3);
        if (x.ndim() == 3) { // 1-D convolution: expand to 2-D, remove the extra axis from the result (conv.rs:142-182; views only)
            if (w.ndim() != 3) throw OpError(OpError::InvalidValue, "kernel must have 3 dims (OCW)");
            Conv op2 = *this;
            if (!padding.same) {
                if (padding.fixed.size() != 2) throw OpError(OpError::InvalidValue, "expected 2 pad values");
                op2.padding = Padding::Fixed({0, padding.fixed[0], 0, padding.fixed[1]});
            }
            if (strides.size() != 1) throw OpError(OpError::InvalidValue, "expected 1 stride value");
            if (dilations.size() != 1) throw OpError(OpError::InvalidValue, "expected 1 dilation value");
            op2.strides = {1, strides[0]};
            op2.dilations = {1, dilations[0]};
            const Tensor x2 = Tensor::view_of(x, {x.size(0), x.size(1), 1, x.size(2)}), w2 = Tensor::view_of(w, {w.size(0), w.size(1), 1, w.size(2)});
            Tensor r2;
            if (residual) r2 = Tensor::view_of(*residual, {residual->size(0), residual->size(1), 1, residual->size(2)});
            OutputList out = op2.run_packed(ctx, {&x2, &w2, bias, residual ? &r2 : nullptr}, packed_weight);
            out[0].reshape({out[0].size(0), out[0].size(1), out[0].size(3)});
            return out;
        }
        const rten_hip_conv2d_desc d = geometry(x.shape(), w.shape());
        if (bias && bias->size(0) != d.o) throw OpError(OpError::IncompatibleInputShapes, "bias.size(0) != out_channels");
        Tensor y(ctx, {d.n, d.o, d.out_h, d.out_w}, DType::F32);
        const uint32_t flags = (fuse_relu ? RTEN_HIP_CONV_RELU : 0u) | (residual ? RTEN_HIP_CONV_RESIDUAL : 0u);
        PlanScope scope(ctx, plan);
        ctx.check(rten_hip_conv2d_f32(ctx.raw(), &d, (const float *)x.ptr(), (const float *)(packed_weight ? packed_weight->ptr() : w.ptr()),
                                      packed_weight ? 1 : 0, (const float *)vp(bias), (const float *)vp(residual), flags, (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// ------------------------------------------------------------------------------------------------ ConvTranspose
struct ConvTranspose : Operator { // src/ops/conv_transpose.rs:414-458; kernel layout [C, O/g, kh, kw]
    Padding padding;
    int groups = 1;
    std::vector<int> strides{1, 1}, dilations{1, 1}, output_padding; // output_padding empty = zeros
    const char *name() const override { return "ConvTranspose"; }
    int max_inputs() const override { return 3; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32"), &w = want(require(in, 1), DType::F32, "float32");
        const Tensor *bias = get(in, 2);
        if (x.ndim() == 3) { // 1-D: expand to 2-D, remove the extra axis from the result (conv_transpose.rs:237-291)
            if (w.ndim() != 3) throw OpError(OpError::InvalidValue, "kernel must have 3 dims (OCW)");
            ConvTranspose op2 = *this;
            if (!padding.same) {
                if (padding.fixed.size() != 2) throw OpError(OpError::InvalidValue, "expected 2 pad values");
                op2.padding = Padding::Fixed({0, padding.fixed[0], 0, padding.fixed[1]});
            }
            if (strides.size() != 1) throw OpError(OpError::InvalidValue, "expected 1 stride value");
            if (dilations.size() != 1) throw OpError(OpError::InvalidValue, "expected 1 dilation value");
            if (!output_padding.empty() && output_padding.size() != 1) throw OpError(OpError::InvalidValue, "expected 1 output_padding value");
            op2.strides = {1, strides[0]};
            op2.dilations = {1, dilations[0]};
            if (!output_padding.empty()) op2.output_padding = {0, output_padding[0]};
            const Tensor x2 = Tensor::view_of(x, {x.size(0), x.size(1), 1, x.size(2)}), w2 = Tensor::view_of(w, {w.size(0), w.size(1), 1, w.size(2)});
            OutputList out = op2.run(ctx, {&x2, &w2, bias});
            out[0].reshape({out[0].size(0), out[0].size(1), out[0].size(3)});
            return out;
        }
        if (groups == 0) throw OpError(OpError::InvalidValue, "Group count must be > 0");
        if (x.ndim() != 4) throw OpError(OpError::InvalidValue, "input must have 4 dims (NCHW)");
        if (w.ndim() != 4) throw OpError(OpError::InvalidValue, "kernel must have 4 dims (COHW)");
        const int64_t o = w.size(1) * groups;
        if (bias && bias->size(0) != o) throw OpError(OpError::IncompatibleInputShapes, "bias.size(0) != out_channels");
        if (x.size(1) != w.size(0)) throw OpError(OpError::IncompatibleInputShapes, "Input channels does not match kernel input channels");
        if (w.size(0) % groups != 0) throw OpError(OpError::InvalidValue, "Input channel count not divisible by groups");
        if (strides.size() != 2) throw OpError(OpError::InvalidValue, "expected 2 stride values");
        if (dilations.size() != 2) throw OpError(OpError::InvalidValue, "expected 2 dilation values");
        if (!output_padding.empty() && output_padding.size() != 2) throw OpError(OpError::InvalidValue, "expected 2 output_padding values");
        if (!padding.same && padding.fixed.size() != 4) throw OpError(OpError::InvalidValue, "Wrong number of pad values");
        int32_t pads[4] = {0, 0, 0, 0}, out_hw[2], out_pads[4];
        if (!padding.same) for (int i = 0; i < 4; i++) pads[i] = padding.fixed[(size_t)i];
        const char *msg = nullptr;
        if (rten_hip_conv_transpose_output_size((int)x.size(2), (int)x.size(3), (int)w.size(2), (int)w.size(3), strides[0], strides[1], padding.same ? 1 : 0, pads, dilations[0],
                                                dilations[1], output_padding.empty() ? 0 : output_padding[0], output_padding.empty() ? 0 : output_padding[1], out_hw, out_pads, &msg))
            throw OpError(OpError::InvalidValue, msg ? msg : "");
        rten_hip_conv2d_desc d{};
        d.n = (int)x.size(0); d.c = (int)x.size(1); d.h = (int)x.size(2); d.w = (int)x.size(3); d.o = (int)o; d.kh = (int)w.size(2); d.kw = (int)w.size(3);
        for (int i = 0; i < 4; i++) d.pads[i] = out_pads[i];
        d.stride_h = strides[0]; d.stride_w = strides[1]; d.dil_h = dilations[0]; d.dil_w = dilations[1]; d.groups = groups; d.out_h = out_hw[0]; d.out_w = out_hw[1];
        Tensor y(ctx, {d.n, d.o, d.out_h, d.out_w}, DType::F32);
        ctx.check(rten_hip_conv_transpose2d_f32(ctx.raw(), &d, (const float *)x.ptr(), (const float *)w.ptr(), (const float *)vp(bias), (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// zero_point_to_vec (src/ops/matmul.rs:513-531)
inline int zero_point_len(const Tensor *zp, int64_t expected) {
    // zero_point_to_vec, src/ops/matmul.rs:513-531
    if (!zp) return 0;
    if (zp->ndim() == 0) return 1;
    if (zp->ndim() == 1) {
        if (zp->len() != expected) throw OpError(OpError::InvalidValue, "Zero point has incorrect size");
        return (int)expected;
    }
    throw OpError(OpError::UnsupportedValue, "Only scalar or vector zero points are supported");
}

struct ConvInteger : Operator {
    Conv conv; // geometry attributes (groups, dilations, padding, strides)
    int pad_mode = RTEN_HIP_PAD_RAW0_I8; // value of padded taps; x86 reference default
    const char *name() const override { return "ConvInteger"; }
    int max_inputs() const override { return 4; }

    rten_hip_conv2d_int8_desc desc(const Tensor &x, const Tensor &w, const Tensor *x_zp, const Tensor *w_zp) const {
        auto is8 = [](DType t) { return t == DType::U8 || t == DType::I8; };
        if (!is8(x.dtype()) || !is8(w.dtype())) throw OpError(OpError::UnsupportedType, "");
        if (x_zp && x_zp->len() != 1) throw OpError(OpError::InvalidValue, "input zero point must be a scalar");
        const int wz = zero_point_len(w_zp, w.ndim() ? w.size(0) : 0);
        rten_hip_conv2d_int8_desc di{};
        di.conv = conv.geometry(x.shape(), w.shape());
        di.x_signed = x.dtype() == DType::I8; di.w_signed = w.dtype() == DType::I8; di.w_zp_len = wz; di.pad_mode = pad_mode;
        return di;
    }
    OutputList run(Context &ctx, const InputList &in) const override { return run_fused(ctx, in, nullptr, nullptr, nullptr, false); }

    // PrepackedInput analogue for constant weights (rten_hip_conv2d_int8_prepack); an empty tensor (len 0) when the staged
    // kernel does not cover the geometry -- pass the plain weights then.
    Tensor prepack(Context &ctx, const Tensor &w) const {
        rten_hip_conv2d_int8_desc di{};
        di.conv.n = 1; di.conv.c = (int)w.size(1) * conv.groups; di.conv.o = (int)w.size(0); di.conv.kh = (int)w.size(2); di.conv.kw = (int)w.size(3);
        di.conv.h = di.conv.kh; di.conv.w = di.conv.kw; di.conv.out_h = di.conv.out_w = 1;
        di.conv.stride_h = di.conv.stride_w = di.conv.dil_h = di.conv.dil_w = 1; di.conv.groups = conv.groups;
        di.w_signed = w.dtype() == DType::I8;
        const size_t nbytes = rten_hip_conv2d_int8_packed_bytes(&di);
        if (!nbytes) return Tensor(ctx, {0}, DType::U8);
        Tensor packed(ctx, {(int64_t)nbytes}, DType::U8);
        ctx.check(rten_hip_conv2d_int8_prepack(ctx.raw(), &di, w.ptr(), packed.ptr()));
        return packed;
    }

    // Options of the staged pipeline: prepacked weights, `x` already in the kernel's staged layout
    // (written by DynamicQuantizeLinearStaged below; its logical shape is still [N, C, H, W]), and a statistics block in
    // which the epilogue accumulates min/max of the f32 outputs for the DynamicQuantizeLinear that consumes them.
    struct Staging {
        const Tensor *packed_weight = nullptr;
        bool x_staged = false;
        void *stats_out = nullptr;
    };

    // scale != null: ConvIntegerToFloat epilogue (cast_scale, then the following Add(bias) / Add(residual) / Relu)
    OutputList run_fused(Context &ctx, const InputList &in, const Tensor *scale, const Tensor *bias, const Tensor *residual, bool relu) const {
        return run_staged(ctx, in, scale, bias, residual, relu, Staging());
    }
    // per_channel_scale: `scale` holds one value per output channel ([1,O,1,1] / [O,1,1], the Cast -> Mul form of per-channel weights)
    OutputList run_staged(Context &ctx, const InputList &in, const Tensor *scale, const Tensor *bias, const Tensor *residual, bool relu,
                          const Staging &sg, bool per_channel_scale = false) const {
        const Tensor &x = require(in, 0), &w = require(in, 1);
        const Tensor *x_zp = get(in, 2), *w_zp = get(in, 3);
        rten_hip_conv2d_int8_desc di = desc(x, w, x_zp, w_zp);
        const bool packed = sg.packed_weight && sg.packed_weight->len() > 0;
        di.weights_packed = packed ? 1 : 0;
        di.x_staged = sg.x_staged ? 1 : 0;
        if (scale && per_channel_scale) {
            if (scale->len() != di.conv.o) throw OpError(OpError::IncompatibleInputShapes, "per-channel scale length does not match output channels");
            di.scale_len = di.conv.o;
        }
        if (bias && bias->len() != di.conv.o) throw OpError(OpError::IncompatibleInputShapes, "bias length does not match output channels");
        Tensor y(ctx, {di.conv.n, di.conv.o, di.conv.out_h, di.conv.out_w}, scale ? DType::F32 : DType::I32);
        const uint32_t flags = (relu ? RTEN_HIP_CONV_RELU : 0u) | (residual ? RTEN_HIP_CONV_RESIDUAL : 0u);
        const void *wp = packed ? sg.packed_weight->ptr() : w.ptr();
        if (sg.stats_out && scale)
            ctx.check(rten_hip_conv2d_int8_stats(ctx.raw(), &di, x.ptr(), wp, vp(x_zp), vp(w_zp), (const float *)vp(scale), (const float *)vp(bias),
                                                 (const float *)vp(residual), flags, y.ptr(), sg.stats_out));
        else
        ctx.check(rten_hip_conv2d_int8(ctx.raw(), &di, x.ptr(), wp, vp(x_zp), vp(w_zp), (const float *)vp(scale), (const float *)vp(bias),
                                       (const float *)vp(residual), flags, y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

struct ConvIntegerToFloat : Operator { // src/ops/conv.rs:552-587: inputs X, W, x_zp, w_zp, scale (+ bias, residual: backend fusion)
    ConvInteger conv;
    bool fuse_relu = false;
    const char *name() const override { return "ConvIntegerToFloat"; }
    int max_inputs() const override { return 7; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &scale = want(require(in, 4), DType::F32, "float32");
        if (scale.len() != 1) throw OpError(OpError::InvalidValue, "scale should be a scalar");
        return conv.run_fused(ctx, InputList(in.begin(), in.begin() + 4), &scale, get(in, 5), get(in, 6), fuse_relu);
    }
};

// ------------------------------------------------------------------------------------------------ MatMul family
namespace detail {
inline int64_t prod(const std::vector<int64_t> &v, size_t b, size_t e) {
    int64_t p = 1;
    for (size_t i = b; i < e; i++) p *= v[i];
    return p;
}
// numpy.matmul shape rules (src/ops/matmul.rs:208-385); contiguous inputs; b_transposed folds transB into strides
inline Tensor matmul(Context &ctx, const Tensor &a, const Tensor &b, const Tensor *bias, float alpha, int act, bool b_transposed) {
    std::vector<int64_t> as = a.shape(), bs = b.shape();
    if (as.empty() || bs.empty()) throw OpError(OpError::InvalidValue, "Inputs must have >= 1 dimensions");
    const bool a_vec = as.size() == 1, b_vec = bs.size() == 1;
    if (a_vec) as.insert(as.begin(), 1);
    if (b_vec) { if (b_transposed) bs.insert(bs.begin(), 1); else bs.push_back(1); }
    if (b_transposed) std::swap(bs[bs.size() - 1], bs[bs.size() - 2]);
    const int64_t m = as[as.size() - 2], k = as.back(), kb = bs[bs.size() - 2], n = bs.back();
    if (k != kb) throw OpError(OpError::IncompatibleInputShapes, "Columns of first matrix does not match rows of second matrix");
    const int64_t na = prod(as, 0, as.size() - 2), nb = prod(bs, 0, bs.size() - 2);
    // broadcast the batch prefixes
    std::vector<int64_t> pa(as.begin(), as.end() - 2), pb(bs.begin(), bs.end() - 2), pre;
    const size_t r = std::max(pa.size(), pb.size());
    pa.insert(pa.begin(), r - pa.size(), 1);
    pb.insert(pb.begin(), r - pb.size(), 1);
    for (size_t i = 0; i < r; i++) {
        if (pa[i] != pb[i] && pa[i] != 1 && pb[i] != 1) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast shapes");
        pre.push_back(std::max(pa[i], pb[i]));
    }
    std::vector<int64_t> oshape = pre;
    oshape.push_back(m);
    oshape.push_back(n);
    Tensor y(ctx, oshape, DType::F32);
    if (y.len() > 0) {
        rten_hip_gemm_desc d{};
        d.k = (int)k; d.n = (int)n; d.a_rs = k; d.a_cs = 1; d.b_rs = b_transposed ? 1 : n; d.b_cs = b_transposed ? k : 1; d.ldc = n;
        d.alpha = alpha; d.beta = 0.f; d.bias_kind = bias ? RTEN_HIP_BIAS_PER_COL : RTEN_HIP_BIAS_NONE; d.act = act;
        if (na > 1 && nb == 1) { // matmul.rs:266-297: one [A*M, K] x [K, N] product
            d.m = (int)(na * m); d.batch = 1;
        } else {
            const int64_t batch = prod(pre, 0, pre.size());
            if ((na != 1 && na != batch) || (nb != 1 && nb != batch)) throw OpError(OpError::UnsupportedValue, "partial batch broadcasting is not supported by the device path");
            d.m = (int)m; d.batch = (int)batch; d.a_bs = na > 1 ? m * k : 0; d.b_bs = nb > 1 ? k * n : 0; d.c_bs = m * n;
        }
        ctx.check(rten_hip_gemm_f32(ctx.raw(), &d, (const float *)a.ptr(), (const float *)b.ptr(), (const float *)vp(bias), (float *)y.ptr()));
    }
    if (a_vec) oshape.erase(oshape.end() - 2);
    if (b_vec) oshape.pop_back();
    y.reshape(oshape);
    return y;
}
} // namespace detail

struct MatMul : Operator { // src/ops/matmul.rs:387-428
    const char *name() const override { return "MatMul"; }
    int max_inputs() const override { return 2; }
    OutputList run(Context &ctx, const InputList &in) const override {
        OutputList out;
        out.push_back(detail::matmul(ctx, want(require(in, 0), DType::F32, "float32"), want(require(in, 1), DType::F32, "float32"), nullptr, 1.f, RTEN_HIP_ACT_NONE, false));
        return out;
    }
};

struct FusedMatMul : Operator { // src/ops/matmul.rs:455-510: MatMul + per-column bias + alpha (act: backend fusion of the following Gelu / Relu)
    float alpha = 1.f;
    int act = RTEN_HIP_ACT_NONE;
    bool transpose_b = false;
    const char *name() const override { return "FusedMatMul"; }
    int max_inputs() const override { return 3; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor *bias = get(in, 2);
        if (bias && bias->ndim() != 1) throw OpError(OpError::InputCastFailed, "expected tensor with 1 dims");
        OutputList out;
        out.push_back(detail::matmul(ctx, want(require(in, 0), DType::F32, "float32"), want(require(in, 1), DType::F32, "float32"), bias, alpha, act, transpose_b));
        return out;
    }
};

struct Gemm : Operator { // src/ops/matmul.rs:106-156: c = alpha * (a b) + beta * c with transA / transB
    float alpha = 1.f, beta = 1.f;
    bool transpose_a = false, transpose_b = false;
    const char *name() const override { return "Gemm"; }
    int max_inputs() const override { return 3; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &a = want(require(in, 0), DType::F32, "float32"), &b = want(require(in, 1), DType::F32, "float32");
        const Tensor *c = get(in, 2);
        if (a.ndim() != 2 || b.ndim() != 2) throw OpError(OpError::InputCastFailed, "expected tensor with 2 dims");
        const int64_t m = transpose_a ? a.size(1) : a.size(0), k = transpose_a ? a.size(0) : a.size(1);
        const int64_t kb = transpose_b ? b.size(1) : b.size(0), n = transpose_b ? b.size(0) : b.size(1);
        if (k != kb) throw OpError(OpError::IncompatibleInputShapes, "Columns of first matrix does not match rows of second matrix");
        // `c` must broadcast to [m, n] (matmul.rs:63-70); the backend takes a row vector [n] or a full matrix
        int bias_kind = RTEN_HIP_BIAS_NONE;
        bool full_c = false;
        if (c) {
            if (c->len() == n && (c->ndim() == 1 || (c->ndim() == 2 && c->size(0) == 1))) bias_kind = RTEN_HIP_BIAS_PER_COL;
            else if (c->ndim() == 2 && c->size(0) == m && c->size(1) == n) full_c = true;
            else throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast c to output shape");
        }
        Tensor y(ctx, {m, n}, DType::F32);
        rten_hip_gemm_desc d{};
        d.m = (int)m; d.n = (int)n; d.k = (int)k; d.batch = 1; d.ldc = n; d.alpha = alpha;
        d.a_rs = transpose_a ? 1 : k; d.a_cs = transpose_a ? m : 1; d.b_rs = transpose_b ? 1 : n; d.b_cs = transpose_b ? k : 1;
        if (full_c) { // output = c; gemm(beta) (matmul.rs:72-82)
            ctx.check(rten_hip_memcpy_d2d(ctx.raw(), y.ptr(), c->ptr(), y.bytes()));
            d.beta = beta;
            ctx.check(rten_hip_gemm_f32(ctx.raw(), &d, (const float *)a.ptr(), (const float *)b.ptr(), nullptr, (float *)y.ptr()));
        } else if (c && beta == 1.f && alpha == 1.f && m != 1) {
            // row vector, c + ab, several rows: the per-column bias epilogue adds in the same place (after the first depth block).  A ONE-row
            // product takes the reference's gemv kernels (rten-gemm/src/lib.rs:668-747,876-891), where the expanded C enters with the
            // FIRST depth block -- ((c + acc0) + acc1) ... -- not after the last: that case runs the general form below.
            d.beta = 0.f; d.bias_kind = bias_kind;
            ctx.check(rten_hip_gemm_f32(ctx.raw(), &d, (const float *)a.ptr(), (const float *)b.ptr(), (const float *)c->ptr(), (float *)y.ptr()));
        } else if (c) { // general case: output = expand(c), then gemm(alpha, beta) (matmul.rs:63-82)
            ctx.check(rten_hip_memset(ctx.raw(), y.ptr(), 0, y.bytes()));
            ctx.check(rten_hip_add_f32(ctx.raw(), y.len(), (const float *)y.ptr(), (const float *)c->ptr(), n, (float *)y.ptr()));
            d.beta = beta;
            ctx.check(rten_hip_gemm_f32(ctx.raw(), &d, (const float *)a.ptr(), (const float *)b.ptr(), nullptr, (float *)y.ptr()));
        } else {
            d.beta = 0.f;
            ctx.check(rten_hip_gemm_f32(ctx.raw(), &d, (const float *)a.ptr(), (const float *)b.ptr(), nullptr, (float *)y.ptr()));
        }
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

struct MatMulInteger : Operator { // src/ops/matmul.rs:582-700 (matmul_integer -> matmul_impl :208-385); scale != null: MatMulIntegerToFloat (:789-794)
    const char *name() const override { return "MatMulInteger"; }
    int max_inputs() const override { return 4; }
    OutputList run(Context &ctx, const InputList &in) const override { return run_scaled(ctx, in, nullptr); }

    // Operator::prepack for input 1 (matmul.rs:696-705 -> matmul_prepack_b; Graph::prepack_weights, src/graph.rs:488-562): a
    // constant 2-D RHS staged once as the int8 kernel's chunk-major image + column sums.  Returns an empty tensor when the
    // staged kernel does not cover the shape (the op then runs unpacked).
    Tensor prepack(Context &ctx, const Tensor &b) const {
        if ((b.dtype() != DType::U8 && b.dtype() != DType::I8) || b.ndim() != 2) return Tensor();
        const size_t bytes = rten_hip_gemm_int8_packed_bytes((int32_t)b.size(0), (int32_t)b.size(1));
        if (!bytes) return Tensor();
        Tensor packed(ctx, {(int64_t)bytes}, DType::U8);
        ctx.check(rten_hip_gemm_int8_prepack(ctx.raw(), (int32_t)b.size(0), (int32_t)b.size(1), b.ptr(), b.size(1), 1, b.dtype() == DType::I8, packed.ptr()));
        return packed;
    }

    // All of matmul_impl's forms: vector operands (numpy.matmul rules), `[A.., M, K] x [K, N]` collapsed to one product with
    // the row zero points cycled (:259-296), batched / broadcast prefixes (batched_gemm_uninit, :302-372).
    OutputList run_scaled(Context &ctx, const InputList &in, const Tensor *scale, const Tensor *packed_b = nullptr) const {
        const Tensor &a = require(in, 0), &b = require(in, 1);
        auto is8 = [](DType t) { return t == DType::U8 || t == DType::I8; };
        if (!is8(a.dtype()) || !is8(b.dtype())) throw OpError(OpError::UnsupportedType, "");
        const Tensor *a_zp = get(in, 2), *b_zp = get(in, 3);
        const int64_t a_rows = a.ndim() > 1 ? a.size(a.ndim() - 2) : 1, b_cols = b.ndim() > 1 ? b.size(b.ndim() - 1) : 1;
        const int azl = zero_point_len(a_zp, a_rows), bzl = zero_point_len(b_zp, b_cols);
        if (a.ndim() < 1 || b.ndim() < 1) throw OpError(OpError::InvalidValue, "Inputs must have >= 1 dimensions");
        std::vector<int64_t> ash = a.shape(), bsh = b.shape();
        const bool a_vec = ash.size() == 1, b_vec = bsh.size() == 1;
        if (a_vec) ash.insert(ash.begin(), 1);
        if (b_vec) bsh.push_back(1);
        const int64_t m = ash[ash.size() - 2], k = ash.back(), kb = bsh[bsh.size() - 2], n = bsh.back();
        if (k != kb) throw OpError(OpError::IncompatibleInputShapes, "Columns of first matrix does not match rows of second matrix");
        std::vector<int64_t> ap(ash.begin(), ash.end() - 2), bp(bsh.begin(), bsh.end() - 2);
        const size_t np = std::max(ap.size(), bp.size());
        ap.insert(ap.begin(), np - ap.size(), 1);
        bp.insert(bp.begin(), np - bp.size(), 1);
        std::vector<int64_t> op(np);
        for (size_t i = 0; i < np; i++) {
            if (ap[i] != bp[i] && ap[i] != 1 && bp[i] != 1) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast shapes");
            op[i] = (ap[i] == 0 || bp[i] == 0) ? 0 : std::max(ap[i], bp[i]);
        }
        int sl = 0;
        if (scale) {
            if (scale->len() != 1 && scale->len() != n) throw OpError(OpError::IncompatibleInputShapes, "Scale length does not match tensor columns");
            sl = (int)scale->len();
        }
        std::vector<int64_t> out_shape = op;
        if (!a_vec) out_shape.push_back(m);
        if (!b_vec) out_shape.push_back(n);
        Tensor y(ctx, out_shape, scale ? DType::F32 : DType::I32);
        OutputList out;
        if (y.len() == 0) { out.push_back(std::move(y)); return out; }
        auto prod = [](const std::vector<int64_t> &v) { int64_t p = 1; for (int64_t x : v) p *= x; return p; };
        const int64_t num_a = prod(ap), num_b = prod(bp), nb = prod(op);
        const bool packed = packed_b && packed_b->len() && num_b == 1;
        auto call = [&](int64_t mm, int batch, int64_t a_bs, int64_t b_bs, int64_t c_bs, int64_t a_off, int64_t b_off, int64_t c_off) {
            rten_hip_gemm_int8_desc d{};
            d.m = (int)mm; d.n = (int)n; d.k = (int)k; d.a_rs = k; d.a_cs = 1; d.b_rs = n; d.b_cs = 1; d.ldc = n;
            d.a_signed = a.dtype() == DType::I8; d.b_signed = b.dtype() == DType::I8;
            d.a_zp_len = azl; d.b_zp_len = bzl; d.scale_len = sl;
            d.batch = batch; d.a_bs = a_bs; d.b_bs = b_bs; d.c_bs = c_bs; d.b_prepacked = packed ? 1 : 0;
            ctx.check(rten_hip_gemm_int8(ctx.raw(), &d, (const uint8_t *)a.ptr() + a_off, packed ? packed_b->ptr() : (const void *)((const uint8_t *)b.ptr() + b_off), vp(a_zp), vp(b_zp),
                                         (const float *)vp(scale), (uint8_t *)y.ptr() + c_off * 4));
        };
        if (num_b == 1) { // one [A*M, K] x [K, N] product; row r uses a_zp[r % M]
            call(num_a * m, 1, 0, 0, 0, 0, 0, 0);
        } else if ((num_a == 1 || ap == op) && bp == op) {
            call(m, (int)nb, num_a == 1 ? 0 : m * k, k * n, m * n, 0, 0, 0);
        } else { // general broadcast: one product per output matrix
            std::vector<int64_t> as(np), bs(np), idx(np, 0);
            for (size_t i = 0; i < np; i++) {
                int64_t sa = 1, sb = 1;
                for (size_t j = i + 1; j < np; j++) { sa *= ap[j]; sb *= bp[j]; }
                as[i] = ap[i] == 1 ? 0 : sa; bs[i] = bp[i] == 1 ? 0 : sb;
            }
            for (int64_t z = 0; z < nb; z++) {
                int64_t ia = 0, ib = 0;
                for (size_t i = 0; i < np; i++) { ia += idx[i] * as[i]; ib += idx[i] * bs[i]; }
                call(m, 1, 0, 0, 0, ia * m * k, ib * k * n, z * m * n);
                for (size_t i = np; i-- > 0;) { if (++idx[i] < op[i]) break; idx[i] = 0; }
            }
        }
        out.push_back(std::move(y));
        return out;
    }
};

struct MatMulNBits : Operator { // src/ops/matmul/contrib.rs:119-196; the reference op has no K / N fields (they come from the shapes)
    int bits = 4;
    int64_t block_size = 32;
    int accuracy_level = 0; // AccuracyLevel::Int8 (4) is an opt-in approximation the reference may decline (contrib.rs:102-108): always Float here
    const char *name() const override { return "MatMulNBits"; }
    int max_inputs() const override { return 3; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &lhs = want(require(in, 0), DType::F32, "float32");
        const Tensor &rhs = require(in, 1);
        if (rhs.dtype() != DType::U8) throw OpError(OpError::InputCastFailed, "input 1: expected uint8");
        if (rhs.ndim() != 3) throw OpError(OpError::InputCastFailed, "input 1: expected 3 dims");
        const Tensor &scales = want(require(in, 2), DType::F32, "float32");
        const int64_t n = rhs.size(0);
        int64_t k_blocks;
        if (scales.ndim() == 2) {
            k_blocks = scales.size(1);
        } else if (scales.ndim() == 1) { // earlier versions of the spec used 1-D scales (contrib.rs:152-164)
            const int64_t k = lhs.ndim() >= 1 ? lhs.size(lhs.ndim() - 1) : 1;
            k_blocks = block_size > 0 ? k / block_size : 0;
            if (scales.len() != n * k_blocks) throw OpError(OpError::InvalidValue, "Expected 1D `scales` size to match columns * block_size");
        } else {
            throw OpError(OpError::InvalidValue, "Expected `scales` to have one or two dims");
        }
        if (in.size() > 3) throw OpError(OpError::UnsupportedValue, "zero_points, g_idx and bias inputs are unsupported");
        if (lhs.ndim() < 2) throw OpError(OpError::InvalidValue, "A input must have at least 2 dims");
        if (bits != 4 && bits != 8) throw OpError(OpError::UnsupportedValue, "Unsupported bits-per-element"); // BlockQuantizedMatrix::new (block_quant.rs:690-708)
        const int64_t elems_per_block = rhs.size(2) * (8 / bits);
        if (elems_per_block < 16 || (elems_per_block & (elems_per_block - 1))) throw OpError(OpError::UnsupportedValue, "Unsupported K block size");
        const int64_t rows = lhs.size(lhs.ndim() - 2), k = lhs.size(lhs.ndim() - 1);
        if (k != rhs.size(1) * elems_per_block) throw OpError(OpError::IncompatibleInputShapes, "Columns of first matrix does not match rows of second matrix");
        if (bits != 4) throw OpError(OpError::UnsupportedValue, "MatMulNBits: only 4-bit elements (GemmError::QuantBitsNotSupported, block_quant.rs:77-79)");
        if (scales.ndim() == 2 && (scales.size(0) != n || k_blocks != rhs.size(1))) throw OpError(OpError::IncompatibleInputShapes, "scales shape does not match the quantised matrix");
        std::vector<int64_t> out_shape(lhs.shape().begin(), lhs.shape().end() - 1);
        out_shape.push_back(n);
        int64_t batch = 1;
        for (int i = 0; i + 2 < lhs.ndim(); i++) batch *= lhs.size(i);
        Tensor y(ctx, out_shape, DType::F32);
        ctx.check(rten_hip_matmul_nbits_f32(ctx.raw(), batch, (int)rows, (int)k, (int)n, (int)elems_per_block, (const float *)lhs.ptr(), (const uint8_t *)rhs.ptr(),
                                            (const float *)scales.ptr(), (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// ------------------------------------------------------------------------------------------------ row-wise / element-wise
inline int resolve_axis(int axis, int ndim) { // resolve_axis (src/ops/mod.rs): same message
    const int a = axis < 0 ? axis + ndim : axis;
    if (a < 0 || a >= ndim) throw OpError(OpError::InvalidValue, "Axis is invalid");
    return a;
}

struct Softmax : Operator { // src/ops/norm.rs:825-840 (last-axis lanes contiguous: other axes need a transpose first)
    int axis = -1;
    const char *name() const override { return "Softmax"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        const int a = resolve_axis(axis, x.ndim());
        if (a != x.ndim() - 1) {
            // normalize_lanes (src/ops/norm.rs:705-754): move the axis last, make contiguous, apply, move it back
            const int nd = x.ndim();
            if (nd > 6) throw OpError(OpError::UnsupportedValue, "Softmax over a non-last axis of more than 6 dims is not supported by the device path");
            std::vector<int32_t> fwd, back((size_t)nd);
            for (int i = 0; i < nd; i++) if (i != a) fwd.push_back(i);
            fwd.push_back(a);
            for (int i = 0; i < nd; i++) back[(size_t)fwd[(size_t)i]] = i;
            std::vector<int64_t> tshape;
            for (int i = 0; i < nd; i++) tshape.push_back(x.size(fwd[(size_t)i]));
            Tensor t(ctx, tshape, DType::F32), u(ctx, tshape, DType::F32), y(ctx, x.shape(), DType::F32);
            if (x.len()) {
                const int64_t cols = x.size(a), rows = x.len() / cols;
                ctx.check(rten_hip_transpose_b32(ctx.raw(), nd, x.shape().data(), fwd.data(), x.ptr(), t.ptr()));
                ctx.check(rten_hip_softmax_f32(ctx.raw(), rows, (int)cols, (const float *)t.ptr(), nullptr, 1, 1, 0, (float *)u.ptr()));
                ctx.check(rten_hip_transpose_b32(ctx.raw(), nd, tshape.data(), back.data(), u.ptr(), y.ptr()));
            }
            OutputList out;
            out.push_back(std::move(y));
            return out;
        }
        const int64_t cols = x.size(a), rows = cols ? x.len() / cols : 0;
        Tensor y(ctx, x.shape(), DType::F32);
        if (x.len()) ctx.check(rten_hip_softmax_f32(ctx.raw(), rows, (int)cols, (const float *)x.ptr(), nullptr, 1, 1, 0, (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

struct LayerNormalization : Operator { // src/ops/norm.rs:456-529
    int axis = -1;
    float epsilon = 1e-5f;
    const char *name() const override { return "LayerNormalization"; }
    int max_inputs() const override { return 3; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        const Tensor &scale = want(require(in, 1), DType::F32, "float32");
        const Tensor *bias = get(in, 2);
        const int a = resolve_axis(axis, x.ndim());
        const int64_t cols = detail::prod(x.shape(), (size_t)a, x.shape().size()), rows = cols ? x.len() / cols : 0;
        if (scale.len() != cols) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast scale to input shape");
        if (bias && bias->len() != cols) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast bias to input shape");
        Tensor y(ctx, x.shape(), DType::F32);
        if (x.len())
            ctx.check(rten_hip_layer_norm_f32(ctx.raw(), rows, (int)cols, (const float *)x.ptr(), (const float *)scale.ptr(), (const float *)vp(bias), 1.f, 0.f, epsilon,
                                              (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

template <int32_t (*FN)(rten_hip_ctx *, int64_t, const float *, float *)>
struct UnaryOp : Operator {
    const char *nm;
    explicit UnaryOp(const char *n) : nm(n) {}
    const char *name() const override { return nm; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        Tensor y(ctx, x.shape(), DType::F32);
        if (x.len()) ctx.check(FN(ctx.raw(), x.len(), (const float *)x.ptr(), (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};
struct Relu : UnaryOp<rten_hip_relu_f32> { Relu() : UnaryOp("Relu") {} };
struct Gelu : UnaryOp<rten_hip_gelu_f32> { Gelu() : UnaryOp("Gelu") {} };
struct Erf : UnaryOp<rten_hip_erf_f32> { Erf() : UnaryOp("Erf") {} };

template <int32_t (*FN)(rten_hip_ctx *, int64_t, const float *, const float *, int64_t, float *), int OPCODE>
struct BinaryOp : Operator { // binary_elementwise.rs:58-170,476-495: numpy broadcasting
    const char *nm;
    explicit BinaryOp(const char *n) : nm(n) {}
    const char *name() const override { return nm; }
    int max_inputs() const override { return 2; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &a = want(require(in, 0), DType::F32, "float32"), &b = want(require(in, 1), DType::F32, "float32");
        const int64_t n = a.len(), bl = b.len();
        // fast path: equal shapes, or `b` broadcast over the leading dims of `a` (flat kernels, 16 B per lane)
        bool fast = bl > 0 && n % bl == 0 && b.ndim() <= a.ndim();
        for (int i = 0; fast && i < b.ndim(); i++) {
            const int64_t bd = b.size(b.ndim() - 1 - i), ad = a.size(a.ndim() - 1 - i);
            if (bd != ad && !(bd == 1 && detail::prod(b.shape(), 0, (size_t)(b.ndim() - 1 - i)) == 1)) fast = false;
        }
        OutputList out;
        if (fast) {
            Tensor y(ctx, a.shape(), DType::F32);
            if (n) ctx.check(FN(ctx.raw(), n, (const float *)a.ptr(), (const float *)b.ptr(), bl, (float *)y.ptr()));
            out.push_back(std::move(y));
            return out;
        }
        // general case: align the shapes on the right, stride 0 on every axis of extent 1
        const int nd = std::max(a.ndim(), b.ndim());
        if (nd > 6) throw OpError(OpError::UnsupportedValue, "broadcasting over more than 6 dims is not supported by the device path");
        std::vector<int64_t> oshape((size_t)nd), as((size_t)nd, 0), bs((size_t)nd, 0);
        int64_t sa = 1, sb = 1;
        for (int i = nd - 1; i >= 0; i--) {
            const int ia = i - (nd - a.ndim()), ib = i - (nd - b.ndim());
            const int64_t da = ia >= 0 ? a.size(ia) : 1, db = ib >= 0 ? b.size(ib) : 1;
            if (da != db && da != 1 && db != 1) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast inputs to same shape");
            oshape[(size_t)i] = std::max(da, db) == 1 ? 1 : (da == 1 ? db : da);
            if (da == 0 || db == 0) oshape[(size_t)i] = 0;
            as[(size_t)i] = da == 1 ? 0 : sa;
            bs[(size_t)i] = db == 1 ? 0 : sb;
            sa *= da; sb *= db;
        }
        Tensor y(ctx, oshape, DType::F32);
        if (y.len()) ctx.check(rten_hip_binary_broadcast_f32(ctx.raw(), OPCODE, nd, oshape.data(), as.data(), bs.data(), (const float *)a.ptr(), (const float *)b.ptr(), (float *)y.ptr()));
        out.push_back(std::move(y));
        return out;
    }
};
struct Add : BinaryOp<rten_hip_add_f32, 0> { Add() : BinaryOp("Add") {} };
struct Mul : BinaryOp<rten_hip_mul_f32, 1> { Mul() : BinaryOp("Mul") {} };
struct Sub : BinaryOp<rten_hip_sub_f32, 2> { Sub() : BinaryOp("Sub") {} };
struct Div : BinaryOp<rten_hip_div_f32, 3> { Div() : BinaryOp("Div") {} };

// ------------------------------------------------------------------------------------------------ layout (src/ops/layout.rs, gather.rs)
struct Transpose : Operator { // layout.rs:669+: perm absent = reverse the axes
    std::vector<int> perm;
    const char *name() const override { return "Transpose"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = require(in, 0);
        if (dtype_size(x.dtype()) != 4) throw OpError(OpError::UnsupportedType, "");
        const int nd = x.ndim();
        std::vector<int32_t> p(perm.begin(), perm.end());
        if (p.empty()) for (int i = nd - 1; i >= 0; i--) p.push_back(i);
        if ((int)p.size() != nd) throw OpError(OpError::InvalidValue, "Permutation is invalid");
        std::vector<int64_t> oshape;
        for (int i = 0; i < nd; i++) {
            if (p[(size_t)i] < 0) p[(size_t)i] += nd;
            if (p[(size_t)i] < 0 || p[(size_t)i] >= nd) throw OpError(OpError::InvalidValue, "Permutation is invalid");
            oshape.push_back(x.size(p[(size_t)i]));
        }
        Tensor y(ctx, oshape, x.dtype());
        ctx.check(rten_hip_transpose_b32(ctx.raw(), nd, x.shape().data(), p.data(), x.ptr(), y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// Gather (src/ops/gather.rs:21-110): 4-byte data, int32 indices, any axis; the embedding-lookup form (axis 0 of a float table) keeps its own kernel.
// Out-of-range indices are clamped on the device (the reference reports "Entry in indices is out of range").
struct Gather : Operator {
    int axis = 0;
    const char *name() const override { return "Gather"; }
    int max_inputs() const override { return 2; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &table = require(in, 0);
        const Tensor &ids = want(require(in, 1), DType::I32, "int32");
        if (dtype_size(table.dtype()) != 4) throw OpError(OpError::UnsupportedType, "");
        if (table.ndim() < 1) throw OpError(OpError::InvalidValue, "Input must have >= 1 dims");
        const int ax = resolve_axis(axis, table.ndim());
        std::vector<int64_t> oshape(table.shape().begin(), table.shape().begin() + ax);
        oshape.insert(oshape.end(), ids.shape().begin(), ids.shape().end());
        oshape.insert(oshape.end(), table.shape().begin() + ax + 1, table.shape().end());
        const int64_t outer = detail::prod(table.shape(), 0, (size_t)ax), inner = detail::prod(table.shape(), (size_t)ax + 1, table.shape().size());
        Tensor y(ctx, oshape, table.dtype());
        if (y.len()) {
            if (ax == 0 && table.dtype() == DType::F32)
                ctx.check(rten_hip_gather_rows_f32(ctx.raw(), ids.len(), (int32_t)inner, (int32_t)table.size(0), (const float *)table.ptr(), (const int32_t *)ids.ptr(), (float *)y.ptr()));
            else
                ctx.check(rten_hip_gather_axis_b32(ctx.raw(), outer, table.size(ax), inner, ids.len(), table.ptr(), (const int32_t *)ids.ptr(), y.ptr()));
        }
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// AddSoftmax (src/ops/attention.rs:94-156): softmax(x + m) over the last axis in one pass; m has x's shape, or is an
// attention mask [B, 1, 1, T] against x = [B, H, S, T], or a single row [T].
struct AddSoftmax : Operator {
    bool flush_nans_to_zero = false;
    const char *name() const override { return "AddSoftmax"; }
    int max_inputs() const override { return 2; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32"), &m = want(require(in, 1), DType::F32, "float32");
        if (x.ndim() < 1) throw OpError(OpError::InvalidValue, "Input must have >= 1 dims");
        const int64_t cols = x.size(x.ndim() - 1), rows = cols ? x.len() / cols : 0;
        int64_t add_div, add_mod;
        if (m.shape() == x.shape()) { add_div = 1; add_mod = rows ? rows : 1; }
        else if (m.len() == cols && m.size(m.ndim() - 1) == cols) { add_div = rows ? rows : 1; add_mod = 1; }
        else if (x.ndim() == 4 && m.ndim() == 4 && m.size(0) == x.size(0) && m.size(1) == 1 && m.size(2) == 1 && m.size(3) == cols) {
            add_div = x.size(1) * x.size(2); add_mod = x.size(0);
        } else throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast inputs to same shape");
        Tensor y(ctx, x.shape(), DType::F32);
        if (x.len()) ctx.check(rten_hip_softmax_f32(ctx.raw(), rows, (int)cols, (const float *)x.ptr(), (const float *)m.ptr(), add_div, add_mod, flush_nans_to_zero ? 1 : 0, (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// Multi-head scaled dot-product attention over projection outputs laid out [B, S, heads * d] (head = column block): the
// Reshape / Transpose nodes around QK^T -> softmax -> PV become stride arithmetic (TransposeFusion's analogue,
// src/optimize/fusions.rs:1066; the core is sdpa_multi_head, src/ops/attention.rs:589-626).  Inputs q, k, v (+ additive mask
// [B,1,1,T] or [B,1,S,T]); q / k / v may be column blocks of one wider buffer: pass `row_stride` / `col_offset` views via
// the *_rs, *_off fields (elements).
struct MultiHeadSdpa : Operator {
    int heads = 1;
    int head_dim = 0; // heads <= 0: the graph's Reshape spells the head COUNT as -1 ([B, S, -1, d], transformers' exporter idiom): heads = hidden / head_dim
    float scale = 1.f;
    bool flush_nans_to_zero = false; // the FusedMatMul -> AddSoftmax -> MatMul graph does not flush; sdpa_head does (attention.rs:551)
    int64_t q_rs = 0, k_rs = 0, v_rs = 0;    // row strides (0: the tensor's own last dim)
    int64_t q_off = 0, k_off = 0, v_off = 0; // column offsets into the rows
    int64_t width = 0;                       // heads * d when q / k / v are column blocks of wider rows (0: last dim)
    const char *name() const override { return "MultiHeadSdpa"; }
    int max_inputs() const override { return 4; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &q = want(require(in, 0), DType::F32, "float32"), &k = want(require(in, 1), DType::F32, "float32"), &v = want(require(in, 2), DType::F32, "float32");
        const Tensor *mask = get(in, 3);
        if (q.ndim() != 3 || k.ndim() != 3 || v.ndim() != 3) throw OpError(OpError::InvalidValue, "expected [batch, seq, hidden] projections");
        const int64_t B = q.size(0), S = q.size(1), T = k.size(1);
        const int64_t Hd = width ? width : q.size(2);
        const int heads = this->heads > 0 ? this->heads : (head_dim > 0 && Hd % head_dim == 0 ? (int)(Hd / head_dim) : 0);
        if (heads <= 0 || Hd % heads != 0) throw OpError(OpError::InvalidValue, "hidden size is not divisible by the number of heads");
        if (k.size(0) != B || v.size(0) != B || v.size(1) != T) throw OpError(OpError::IncompatibleInputShapes, "q / k / v batch or sequence sizes do not match");
        if (!width && (k.size(2) != Hd || v.size(2) != Hd)) throw OpError(OpError::IncompatibleInputShapes, "q / k / v hidden sizes do not match");
        const int64_t d = Hd / heads;
        rten_hip_sdpa_desc sd{};
        sd.batch = (int)B; sd.heads = heads; sd.s = (int)S; sd.t = (int)T; sd.d = (int)d; sd.dv = (int)d;
        sd.q_rs = q_rs ? q_rs : q.size(2); sd.k_rs = k_rs ? k_rs : k.size(2); sd.v_rs = v_rs ? v_rs : v.size(2);
        sd.q_bs = S * sd.q_rs; sd.k_bs = T * sd.k_rs; sd.v_bs = T * sd.v_rs;
        sd.q_hs = sd.k_hs = sd.v_hs = d;
        sd.o_rs = Hd; sd.o_bs = S * Hd; sd.o_hs = d;
        sd.scale = scale; sd.flush_nan_to_zero = flush_nans_to_zero ? 1 : 0;
        Tensor expanded; // a mask that broadcasts to [B, 1, S, T] in another form (e.g. a causal [1, 1, S, T] or [S, T]) is expanded first
        if (mask) {
            want(*mask, DType::F32, "float32");
            if (mask->ndim() == 4 && mask->size(0) == B && mask->size(1) == 1 && mask->size(3) == T && (mask->size(2) == 1 || mask->size(2) == S)) {
                // (an [B, 1, S, T] mask known to hold S copies of one row -- Tensor::uniform_dims, the exporter's expanded padding mask -- is read as that row)
                const bool one_row = mask->size(2) == 1 || (mask->uniform_dims() >> 2 & 1u);
                sd.mask_row_stride = one_row ? 0 : T;
                sd.mask_batch_stride = mask->size(2) == 1 ? T : S * T;
            } else {
                // numpy broadcasting of the Add(scores [B, H, S, T], mask): every form without a head axis is the same addend for
                // all heads, so the expanded copy gives the unfused graph's bits
                const int nd = mask->ndim();
                if (nd > 4) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast inputs to same shape");
                int64_t ms[4] = {1, 1, 1, 1};
                for (int i = 0; i < nd; i++) ms[4 - nd + i] = mask->size(i);
                const int64_t want_shape[4] = {B, 1, S, T};
                if (ms[1] != 1) throw OpError(OpError::UnsupportedValue, "fused attention: a mask with a head axis is not supported (run the graph unfused)");
                int64_t strides[4], acc = 1;
                for (int i = 3; i >= 0; i--) {
                    if (ms[i] != 1 && ms[i] != want_shape[i]) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast inputs to same shape");
                    strides[i] = ms[i] == 1 ? 0 : acc;
                    acc *= ms[i];
                }
                expanded = Tensor(ctx, {B, 1, S, T}, DType::F32);
                if (expanded.len()) ctx.check(rten_hip_copy_strided_b32(ctx.raw(), 4, want_shape, strides, mask->ptr(), expanded.ptr()));
                mask = &expanded;
                sd.mask_row_stride = T;
                sd.mask_batch_stride = S * T;
            }
        }
        Tensor y(ctx, {B, S, Hd}, DType::F32);
        if (y.len())
            ctx.check(rten_hip_sdpa_f32(ctx.raw(), &sd, (const float *)q.ptr() + q_off, (const float *)k.ptr() + k_off, (const float *)v.ptr() + v_off,
                                        (const float *)vp(mask), (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// Add(residual) -> LayerNormalization as one kernel (the sum is formed in registers in the reference's order: x + r).
struct AddLayerNormalization : Operator {
    float epsilon = 1e-5f;
    const char *name() const override { return "AddLayerNormalization"; }
    int max_inputs() const override { return 4; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32"), &r = want(require(in, 1), DType::F32, "float32");
        const Tensor &scale = want(require(in, 2), DType::F32, "float32");
        const Tensor *bias = get(in, 3);
        if (x.shape() != r.shape()) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast inputs to same shape");
        const int64_t cols = x.ndim() ? x.size(x.ndim() - 1) : 1, rows = cols ? x.len() / cols : 0;
        if (scale.len() != cols) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast scale to input shape");
        if (bias && bias->len() != cols) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast bias to input shape");
        Tensor y(ctx, x.shape(), DType::F32);
        if (x.len())
            ctx.check(rten_hip_add_layer_norm_f32(ctx.raw(), rows, (int)cols, (const float *)x.ptr(), (const float *)r.ptr(), (const float *)scale.ptr(), (const float *)vp(bias), 1.f,
                                                  0.f, epsilon, (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// ------------------------------------------------------------------------------------------------ pooling
struct PoolBase : Operator {
    std::vector<int> kernel_size{1, 1}, strides{1, 1};
    Padding padding;
    bool ceil_mode = false;
    rten_hip_pool2d_desc desc(const Tensor &x, bool count_include_pad) const {
        if (x.ndim() != 4) throw OpError(OpError::InvalidValue, "input must have 4 dims (NCHW)");
        const OutputSize os = calc_output_size_and_padding((int)x.size(2), (int)x.size(3), kernel_size[0], kernel_size[1], strides, padding, {1, 1}, ceil_mode);
        rten_hip_pool2d_desc d{};
        d.n = (int)x.size(0); d.c = (int)x.size(1); d.h = (int)x.size(2); d.w = (int)x.size(3);
        d.kh = kernel_size[0]; d.kw = kernel_size[1]; d.stride_h = strides[0]; d.stride_w = strides[1];
        for (int i = 0; i < 4; i++) d.pads[i] = os.pads[i];
        d.out_h = os.oh; d.out_w = os.ow; d.count_include_pad = count_include_pad;
        return d;
    }
};
struct MaxPool : PoolBase { // src/ops/pooling.rs:477-521
    void *stats_out = nullptr; // executor-set: a DynamicQuantizeLinear reads the output next -> its min / max come out of this launch
    const char *name() const override { return "MaxPool"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        const rten_hip_pool2d_desc d = desc(x, false);
        Tensor y(ctx, {d.n, d.c, d.out_h, d.out_w}, DType::F32);
        if (stats_out) ctx.check(rten_hip_max_pool2d_f32_stats(ctx.raw(), &d, (const float *)x.ptr(), (float *)y.ptr(), stats_out));
        else ctx.check(rten_hip_max_pool2d_f32(ctx.raw(), &d, (const float *)x.ptr(), (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};
struct AveragePool : PoolBase { // src/ops/pooling.rs:174-389
    bool count_include_pad = false;
    const char *name() const override { return "AveragePool"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        const rten_hip_pool2d_desc d = desc(x, count_include_pad);
        Tensor y(ctx, {d.n, d.c, d.out_h, d.out_w}, DType::F32);
        ctx.check(rten_hip_average_pool2d_f32(ctx.raw(), &d, (const float *)x.ptr(), (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};
struct GlobalAveragePool : Operator { // src/ops/pooling.rs:392-417
    const char *name() const override { return "GlobalAveragePool"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        if (x.ndim() != 4) throw OpError(OpError::InvalidValue, "input must have 4 dims (NCHW)");
        Tensor y(ctx, {x.size(0), x.size(1), 1, 1}, DType::F32);
        if (x.len()) ctx.check(rten_hip_global_average_pool_f32(ctx.raw(), x.size(0) * x.size(1), (int)(x.size(2) * x.size(3)), (const float *)x.ptr(), (float *)y.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        return out;
    }
};

// ------------------------------------------------------------------------------------------------ quantization
struct DynamicQuantizeLinear : Operator { // src/ops/quantize.rs:352-436: outputs y (u8), y_scale (f32 scalar), y_zero_point (u8 scalar)
    const char *name() const override { return "DynamicQuantizeLinear"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        Tensor y(ctx, x.shape(), DType::U8), s(ctx, {}, DType::F32), z(ctx, {}, DType::U8);
        ctx.check(rten_hip_dynamic_quantize_linear(ctx.raw(), x.len(), (const float *)x.ptr(), (uint8_t *)y.ptr(), (float *)s.ptr(), (uint8_t *)z.ptr()));
        OutputList out;
        out.push_back(std::move(y));
        out.push_back(std::move(s));
        out.push_back(std::move(z));
        return out;
    }
};

// DynamicQuantizeLinear whose u8 output is only consumed by int8 convolutions of one staging geometry: the codes are
// written once, directly in the kernel's staged layout (bit-identical values), and -- when the producer of `x` left
// min/max statistics -- without the extra sweep over x.  Outputs: staged image (logical shape of x), y_scale, y_zero_point.
struct DynamicQuantizeLinearStaged : Operator {
    ConvInteger consumer;        // geometry attributes of (one of) the consuming convolutions
    std::vector<int64_t> kernel; // [O, C/g, kh, kw] of that consumer
    const void *stats_in = nullptr;
    // optional: the scalar Mul(y_scale, w_scale) nodes that follow in ort-quantized graphs, one per convolution reading the codes (a stage's
    // shortcut and first 1x1 share one quantizer) -> outputs 4, 5, ... in this order (rten_hip_dynamic_quantize_linear_staged_products)
    std::vector<const Tensor *> mul_by;
    static constexpr size_t kMaxProducts = 4;
    const char *name() const override { return "DynamicQuantizeLinear"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        rten_hip_conv2d_int8_desc di{};
        di.conv = consumer.conv.geometry(x.shape(), kernel);
        di.x_signed = 0; di.w_signed = 1; di.pad_mode = consumer.pad_mode;
        const size_t nbytes = rten_hip_conv2d_int8_staged_bytes(&di);
        if (!nbytes) throw OpError(OpError::UnsupportedValue, "quantize_staged: geometry not covered by the staged kernel");
        if (mul_by.size() > kMaxProducts) throw OpError(OpError::UnsupportedValue, "quantize_staged: at most 4 scale products per launch");
        Tensor y(ctx, x.shape(), DType::U8, nbytes), s(ctx, {}, DType::F32), z(ctx, {}, DType::U8);
        std::vector<Tensor> prods;
        prods.reserve(kMaxProducts);
        const float *mb[kMaxProducts] = {};
        float *pr[kMaxProducts] = {};
        for (size_t i = 0; i < mul_by.size(); i++) {
            if (!mul_by[i] || mul_by[i]->len() != 1) throw OpError(OpError::InvalidValue, "scale should be a scalar");
            prods.emplace_back(ctx, mul_by[i]->shape(), DType::F32);
            mb[i] = (const float *)mul_by[i]->ptr();
            pr[i] = (float *)prods.back().ptr();
        }
        ctx.check(rten_hip_dynamic_quantize_linear_staged_products(ctx.raw(), &di, (const float *)x.ptr(), stats_in, y.ptr(), (float *)s.ptr(), (uint8_t *)z.ptr(),
                                                                   (int32_t)mul_by.size(), mul_by.empty() ? nullptr : mb, mul_by.empty() ? nullptr : pr));
        OutputList out;
        out.push_back(std::move(y));
        out.push_back(std::move(s));
        out.push_back(std::move(z));
        for (Tensor &t : prods) out.push_back(std::move(t));
        return out;
    }

    // This quantizer INSIDE the launch of the ConvIntegerToFloat step `prod` that produces its input (rten_hip_conv2d_int8_qout: the f32 values never
    // leave the registers unless `keep_f32` -- a residual Add also reads them).  `in` = prod's X (staged), W, x_zp, w_zp; `sync` = the edge's exchange
    // block.  out = {prod's f32 output (empty unless keep_f32), codes, scale, zero point[, product]}.  Returns false when the launch form does not
    // apply (geometry not covered, or more workgroups than the device holds at once: RTEN_HIP_ERR_UNSUPPORTED): run the two operators then.  Opt-in
    // (launch plan) only: see the time-out contract in rten_hip.h.
    // `sync` == nullptr with `recompute`: the two-launch recompute form (statistics-only pass, then the convolution again with the codes as its output) -- no
    // grid-wide exchange, so replicas running side by side may use it.
    bool run_in_producer(Context &ctx, const ConvInteger &prod, const InputList &in, const Tensor &scale, const Tensor *bias, const Tensor *residual, bool relu,
                         const ConvInteger::Staging &sg, bool per_channel_scale, void *sync, bool keep_f32, OutputList &out, bool recompute = false) const {
        if (mul_by.size() > kMaxProducts || (!sync && !recompute) || !sg.stats_out || !sg.x_staged || !sg.packed_weight || !sg.packed_weight->len()) return false;
        if (recompute) sync = nullptr;
        const Tensor &x = require(in, 0), &w = require(in, 1);
        const Tensor *x_zp = get(in, 2), *w_zp = get(in, 3);
        rten_hip_conv2d_int8_desc di = prod.desc(x, w, x_zp, w_zp);
        di.weights_packed = 1;
        di.x_staged = 1;
        if (per_channel_scale) di.scale_len = di.conv.o;
        const std::vector<int64_t> yshape{di.conv.n, di.conv.o, di.conv.out_h, di.conv.out_w};
        rten_hip_conv2d_int8_desc dn{};
        dn.conv = consumer.conv.geometry(yshape, kernel);
        dn.x_signed = 0; dn.w_signed = 1; dn.pad_mode = consumer.pad_mode;
        const size_t nbytes = rten_hip_conv2d_int8_staged_bytes(&dn);
        if (!nbytes) return false;
        for (const Tensor *mb : mul_by) if (!mb || mb->len() != 1) throw OpError(OpError::InvalidValue, "scale should be a scalar");
        Tensor y(ctx, keep_f32 ? yshape : std::vector<int64_t>{0}, DType::F32);
        Tensor q(ctx, yshape, DType::U8, nbytes), s(ctx, {}, DType::F32), z(ctx, {}, DType::U8);
        Tensor pr(ctx, mul_by.empty() ? std::vector<int64_t>{0} : mul_by[0]->shape(), DType::F32);
        const uint32_t flags = (relu ? RTEN_HIP_CONV_RELU : 0u) | (residual ? RTEN_HIP_CONV_RESIDUAL : 0u);
        const int32_t rc = rten_hip_conv2d_int8_qout(ctx.raw(), &di, x.ptr(), sg.packed_weight->ptr(), vp(x_zp), vp(w_zp), (const float *)scale.ptr(),
                                                     (const float *)vp(bias), (const float *)vp(residual), flags, keep_f32 ? (float *)y.ptr() : nullptr, sg.stats_out, sync,
                                                     &dn, q.ptr(), (float *)s.ptr(), (uint8_t *)z.ptr(), mul_by.empty() ? nullptr : (const float *)mul_by[0]->ptr(),
                                                     mul_by.empty() ? nullptr : (float *)pr.ptr());
        if (rc == RTEN_HIP_ERR_UNSUPPORTED) return false;
        ctx.check(rc);
        // a quantizer read by several convolutions (a stage's shortcut and first 1x1): the launch folds the first Mul(y_scale, w_scale); the others
        // are the graph's own scalar Mul nodes, one tiny launch each (same f32 multiply)
        std::vector<Tensor> more;
        for (size_t i = 1; i < mul_by.size(); i++) {
            more.emplace_back(ctx, mul_by[i]->shape(), DType::F32);
            ctx.check(rten_hip_mul_f32(ctx.raw(), 1, (const float *)s.ptr(), (const float *)mul_by[i]->ptr(), 1, (float *)more.back().ptr()));
        }
        out.clear();
        out.push_back(std::move(y));
        out.push_back(std::move(q));
        out.push_back(std::move(s));
        out.push_back(std::move(z));
        if (!mul_by.empty()) out.push_back(std::move(pr));
        for (Tensor &t : more) out.push_back(std::move(t));
        return true;
    }
};

// Cast (src/ops/convert.rs): the device path covers what the quantized graphs need, int32 -> float32 (exact conversion with
// round-to-nearest-even above 2^24, the same instruction cast_scale uses) and identity casts.
inline int32_t abi_dtype(DType t) { return t == DType::F32 ? RTEN_HIP_DT_F32 : t == DType::I32 ? RTEN_HIP_DT_I32 : t == DType::U8 ? RTEN_HIP_DT_U8 : RTEN_HIP_DT_I8; }

// numpy broadcasting of up to three operands: the common shape and each operand's element strides on it (0 on broadcast axes)
struct Broadcast {
    std::vector<int64_t> shape;
    std::vector<int64_t> strides[3];
};
inline Broadcast broadcast_shapes(const std::vector<const std::vector<int64_t> *> &shapes) {
    Broadcast b;
    size_t nd = 0;
    for (auto *sh : shapes) nd = std::max(nd, sh->size());
    if (nd > 6) throw OpError(OpError::UnsupportedValue, "broadcasting over more than 6 dims is not supported by the device path");
    b.shape.assign(nd, 1);
    for (auto *sh : shapes)
        for (size_t i = 0; i < sh->size(); i++) {
            const size_t o = nd - sh->size() + i;
            const int64_t d = (*sh)[i];
            if (d != 1) {
                if (b.shape[o] != 1 && b.shape[o] != d) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast inputs");
                b.shape[o] = d;
            }
        }
    for (size_t k = 0; k < shapes.size() && k < 3; k++) {
        const auto &sh = *shapes[k];
        b.strides[k].assign(nd, 0);
        int64_t acc = 1;
        for (size_t i = sh.size(); i-- > 0;) {
            b.strides[k][nd - sh.size() + i] = sh[i] == 1 ? 0 : acc;
            acc *= sh[i];
        }
    }
    return b;
}

// Which dims of an element-wise result are uniform (Tensor::uniform_dims): dim d of the broadcast shape is, when EVERY operand is constant along it -- its own
// size there is 1 (or the dim is absent), it carries the flag, or it is a host value whose slices along that dim are equal (checked here, the values are small).
inline bool host_uniform_along(const HostVal &h, size_t dim) {
    int64_t inner = 1, n = h.shape[dim];
    for (size_t i = dim + 1; i < h.shape.size(); i++) inner *= h.shape[i];
    const int64_t total = h.len();
    if (n <= 1 || total == 0) return true;
    for (int64_t base = 0; base < total; base += n * inner)
        for (int64_t k = 1; k < n; k++)
            for (int64_t j = 0; j < inner; j++) {
                const size_t x = (size_t)(base + j), y = (size_t)(base + k * inner + j);
                if (h.is_float ? std::memcmp(&h.f[x], &h.f[y], 4) != 0 : h.i[x] != h.i[y]) return false;
            }
    return true;
}
inline uint32_t uniform_after_broadcast(const std::vector<const Tensor *> &ins, const std::vector<int64_t> &out_shape) {
    uint32_t m = 0;
    const size_t nd = out_shape.size();
    for (size_t d = 0; d < nd && d < 32; d++) {
        if (out_shape[d] <= 1) continue; // (a size-1 dim carries no information; consumers test sizes first)
        bool all = true;
        for (const Tensor *t : ins) {
            if (!t) continue;
            const size_t tn = (size_t)t->ndim();
            if (d + tn < nd) continue;               // the operand has no such dim: broadcast along it
            const size_t td = d - (nd - tn);
            if (t->size((int)td) == 1) continue;
            if (t->uniform_dims() >> td & 1u) continue;
            if (t->host() && t->host()->len() <= (int64_t)1 << 20 && host_uniform_along(*t->host(), td)) continue;
            all = false;
            break;
        }
        if (all) m |= 1u << d;
    }
    return m;
}

struct Cast : Operator { // src/ops/convert.rs:18-60: every pair of float32 / int32 / uint8 / int8 (`as` casts)
    DType to = DType::F32;
    const char *name() const override { return "Cast"; }
    int max_inputs() const override { return 1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &x = require(in, 0);
        OutputList out;
        Tensor y(ctx, x.shape(), to);
        if (x.dtype() == to) { // identity cast: a copy
            if (x.bytes()) ctx.check(rten_hip_memcpy_d2d(ctx.raw(), y.ptr(), x.ptr(), x.bytes()));
        } else if (x.dtype() == DType::I32 && to == DType::F32) {
            if (x.len()) ctx.check(rten_hip_cast_scale(ctx.raw(), x.len(), (const int32_t *)x.ptr(), ctx.one(), 1, (float *)y.ptr()));
        } else if (x.len()) {
            const int64_t n = x.len(), one = 1;
            ctx.check(rten_hip_elementwise_nd(ctx.raw(), RTEN_HIP_EW_CAST, 1, &n, x.ptr(), abi_dtype(x.dtype()), &one, nullptr, 0, nullptr, nullptr, nullptr, y.ptr(), abi_dtype(to)));
        }
        y.set_uniform_dims(x.uniform_dims());
        out.push_back(std::move(y));
        return out;
    }
};

struct Tanh : UnaryOp<rten_hip_tanh_f32> { Tanh() : UnaryOp("Tanh") {} }; // rten-vecmath/src/tanh.rs (BERT's pooler)

// Not / And / Or / Xor, Equal / Less / LessOrEqual / Greater / GreaterOrEqual, integer Add / Sub / Mul / Div: numpy broadcasting, int32 results
// (unary_elementwise.rs:563-565, binary_elementwise.rs:546-598,733-786).  `code` = the RTEN_HIP_EW_* operation.
struct ElementwiseNd : Operator {
    int code = RTEN_HIP_EW_AND;
    const char *nm = "And";
    ElementwiseNd() = default;
    ElementwiseNd(int c, const char *n) : code(c), nm(n) {}
    const char *name() const override { return nm; }
    int max_inputs() const override { return code == RTEN_HIP_EW_NOT ? 1 : 2; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &a = require(in, 0);
        OutputList out;
        if (code == RTEN_HIP_EW_NOT) {
            want(a, DType::I32, "int32");
            Tensor y(ctx, a.shape(), DType::I32);
            const int64_t n = a.len(), one = 1;
            if (n) ctx.check(rten_hip_elementwise_nd(ctx.raw(), code, 1, &n, a.ptr(), RTEN_HIP_DT_I32, &one, nullptr, 0, nullptr, nullptr, nullptr, y.ptr(), RTEN_HIP_DT_I32));
            y.set_uniform_dims(a.uniform_dims());
            out.push_back(std::move(y));
            return out;
        }
        const Tensor &b = require(in, 1);
        const bool cmp = code >= RTEN_HIP_EW_EQUAL && code <= RTEN_HIP_EW_GREATER_EQ;
        if (cmp) { if (a.dtype() != b.dtype() || (a.dtype() != DType::F32 && a.dtype() != DType::I32)) throw OpError(OpError::UnsupportedType, ""); }
        else { want(a, DType::I32, "int32"); want(b, DType::I32, "int32"); }
        const Broadcast bc = broadcast_shapes({&a.shape(), &b.shape()});
        Tensor y(ctx, bc.shape, DType::I32);
        if (y.len())
            ctx.check(rten_hip_elementwise_nd(ctx.raw(), code, (int32_t)bc.shape.size(), bc.shape.data(), a.ptr(), abi_dtype(a.dtype()), bc.strides[0].data(), b.ptr(), abi_dtype(b.dtype()),
                                              bc.strides[1].data(), nullptr, nullptr, y.ptr(), RTEN_HIP_DT_I32));
        y.set_uniform_dims(uniform_after_broadcast({&a, &b}, bc.shape));
        out.push_back(std::move(y));
        return out;
    }
};

struct Where : Operator { // binary_elementwise.rs:1189-1280: cond != 0 ? x : y, the three operands broadcast together
    const char *name() const override { return "Where"; }
    int max_inputs() const override { return 3; }
    OutputList run(Context &ctx, const InputList &in) const override {
        const Tensor &c = want(require(in, 0), DType::I32, "int32"), &x = require(in, 1), &y = require(in, 2);
        if (x.dtype() != y.dtype() || dtype_size(x.dtype()) != 4) throw OpError(OpError::UnsupportedType, "");
        const Broadcast bc = broadcast_shapes({&c.shape(), &x.shape(), &y.shape()});
        Tensor o(ctx, bc.shape, x.dtype());
        if (o.len())
            ctx.check(rten_hip_elementwise_nd(ctx.raw(), RTEN_HIP_EW_WHERE, (int32_t)bc.shape.size(), bc.shape.data(), c.ptr(), RTEN_HIP_DT_I32, bc.strides[0].data(), x.ptr(),
                                              abi_dtype(x.dtype()), bc.strides[1].data(), y.ptr(), bc.strides[2].data(), o.ptr(), abi_dtype(x.dtype())));
        o.set_uniform_dims(uniform_after_broadcast({&c, &x, &y}, bc.shape));
        OutputList out;
        out.push_back(std::move(o));
        return out;
    }
};

// Expand (src/ops/layout.rs:177-262): numpy broadcast of x against `shape` (the values of the second input, which the host must know)
inline Tensor expand_to(Context &ctx, const Tensor &x, const std::vector<int64_t> &target) {
    if (dtype_size(x.dtype()) != 4) throw OpError(OpError::UnsupportedType, "");
    const Broadcast bc = broadcast_shapes({&x.shape(), &target});
    Tensor y(ctx, bc.shape, x.dtype());
    if (y.len()) ctx.check(rten_hip_copy_strided_b32(ctx.raw(), (int32_t)bc.shape.size(), bc.shape.data(), bc.strides[0].data(), x.ptr(), y.ptr()));
    y.set_uniform_dims(uniform_after_broadcast({&x}, bc.shape));
    return y;
}

// Slice (src/ops/slice.rs:21-118): per-axis [start, end) with a positive or negative step; clamping rules of the ONNX operator.  Returns the
// resolved (start, count, step) per axis of `shape`.
struct SliceRange { int64_t start, count, step; };
inline std::vector<SliceRange> resolve_slice(const std::vector<int64_t> &shape, const std::vector<int64_t> &starts, const std::vector<int64_t> &ends,
                                             const std::vector<int64_t> &axes, const std::vector<int64_t> &steps) {
    const int nd = (int)shape.size();
    if (starts.size() != ends.size() || (!axes.empty() && axes.size() != starts.size()) || (!steps.empty() && steps.size() != starts.size()))
        throw OpError(OpError::InvalidValue, "Slice: starts, ends, axes and steps must have the same length");
    std::vector<SliceRange> r((size_t)nd);
    for (int d = 0; d < nd; d++) r[(size_t)d] = {0, shape[(size_t)d], 1};
    for (size_t k = 0; k < starts.size(); k++) {
        const int ax = resolve_axis((int)(axes.empty() ? (int64_t)k : axes[k]), nd);
        const int64_t dim = shape[(size_t)ax], step = steps.empty() ? 1 : steps[k];
        if (step == 0) throw OpError(OpError::InvalidValue, "Slice: steps must be non-zero");
        int64_t st = starts[k], en = ends[k];
        if (st < 0) st += dim;
        if (en < 0) en += dim;
        if (step > 0) {
            st = std::min(std::max<int64_t>(st, 0), dim);
            en = std::min(std::max<int64_t>(en, 0), dim);
            r[(size_t)ax] = {st, en > st ? (en - st + step - 1) / step : 0, step};
        } else {
            st = std::min(std::max<int64_t>(st, -1), dim - 1);
            en = std::min(std::max<int64_t>(en, -1), dim - 1);
            r[(size_t)ax] = {st, st > en ? (st - en + (-step) - 1) / (-step) : 0, step};
        }
    }
    return r;
}
inline Tensor slice_tensor(Context &ctx, const Tensor &x, const std::vector<SliceRange> &r) {
    if (dtype_size(x.dtype()) != 4) throw OpError(OpError::UnsupportedType, "");
    const int nd = x.ndim();
    if (nd > 6) throw OpError(OpError::UnsupportedValue, "Slice: more than 6 dims on the device path");
    std::vector<int64_t> oshape((size_t)nd), st((size_t)nd);
    int64_t acc = 1, base = 0;
    bool forward = true;
    for (int d = nd - 1; d >= 0; d--) {
        oshape[(size_t)d] = r[(size_t)d].count;
        st[(size_t)d] = acc * r[(size_t)d].step;
        base += acc * r[(size_t)d].start;
        if (r[(size_t)d].step < 0) forward = false;
        acc *= x.size(d);
    }
    Tensor y(ctx, oshape, x.dtype());
    if (!y.len()) return y;
    if (forward) {
        ctx.check(rten_hip_copy_strided_b32(ctx.raw(), nd, oshape.data(), st.data(), (const char *)x.ptr() + base * 4, y.ptr()));
    } else { // negative steps: the strided-copy entry point takes non-negative strides; the generic kernel takes signed ones (Cast int32 -> int32 moves raw words)
        const int32_t dt = RTEN_HIP_DT_I32;
        ctx.check(rten_hip_elementwise_nd(ctx.raw(), RTEN_HIP_EW_IADD, nd, oshape.data(), (const char *)x.ptr() + base * 4, dt, st.data(), ctx.zero_i32(), dt,
                                          std::vector<int64_t>((size_t)nd, 0).data(), nullptr, nullptr, y.ptr(), dt));
    }
    return y;
}

// Concat (src/ops/concat.rs:21-108) of 4-byte tensors along `axis`
inline Tensor concat_tensors(Context &ctx, const InputList &in, int axis) {
    const Tensor &first = require(in, 0);
    const int nd = first.ndim();
    const int ax = resolve_axis(axis, nd);
    std::vector<int64_t> oshape = first.shape();
    oshape[(size_t)ax] = 0;
    for (size_t k = 0; k < in.size(); k++) {
        const Tensor &t = require(in, k);
        if (t.dtype() != first.dtype() || dtype_size(t.dtype()) != 4) throw OpError(OpError::UnsupportedType, "");
        if (t.ndim() != nd) throw OpError(OpError::IncompatibleInputShapes, "Tensors must have the same number of dimensions");
        for (int d = 0; d < nd; d++) if (d != ax && t.size(d) != first.size(d)) throw OpError(OpError::IncompatibleInputShapes, "Dimensions must be the same except for concat axis");
        oshape[(size_t)ax] += t.size(ax);
    }
    Tensor y(ctx, oshape, first.dtype());
    const int64_t outer = detail::prod(oshape, 0, (size_t)ax), inner = detail::prod(oshape, (size_t)ax + 1, oshape.size());
    const int64_t dst_pitch = oshape[(size_t)ax] * inner;
    int64_t off = 0;
    for (size_t k = 0; k < in.size(); k++) {
        const int64_t row = in[k]->size(ax) * inner;
        if (row && outer) ctx.check(rten_hip_copy_rows_b32(ctx.raw(), outer, row, in[k]->ptr(), row, (char *)y.ptr() + off * 4, dst_pitch));
        off += row;
    }
    return y;
}

// ------------------------------------------------------------------------------------------------ Einsum / ReduceSum
// src/ops/einsum.rs:21-692, rten-shape-inference/src/einsum_parser.rs:68-275, src/ops/reduce.rs:414-520,1101-1165.
// The reference walks a path of two-term steps over permuted TensorViews and copies where a kernel wants contiguous data.
// Here a view is (device pointer, shape, element strides) and the kernels read THROUGH the strides: permutes, inserted axes,
// diagonals and 1 -> n expansion are stride arithmetic; ReduceSum runs in place on the strided view; a product is ONE
// rten_hip_gemm_f32 launch (M / K / N are strides, batch labels become the descriptor's two batch levels once neighbouring
// axes with compatible strides are merged); the final permutation into the output order goes into the GEMM's C strides when it
// keeps N innermost and is a copy otherwise.  The order of the
// arithmetic is the reference's (same path, same M / N / batch labels, lone labels summed first, [A, M, K] x [K, N] folded
// into one GEMM), so results are bit-identical except for a GEMM with one row (ISA-dependent gemv path).
namespace einsum_detail {
using Shape = std::vector<int64_t>;
constexpr char INS_M = '<', INS_N = '>', MERGED_K = '*'; // einsum.rs:366-380
constexpr int MAX_DIMS = 10;                             // einsum_parser.rs:241-245

inline bool has(const std::string &s, char c) { return s.find(c) != std::string::npos; }
inline std::string dedup(const std::string &s) {
    std::string o;
    for (char c : s) if (!has(o, c)) o.push_back(c);
    return o;
}
inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f'; }
inline std::string strip_ws(const std::string &s) {
    std::string o;
    for (char c : s) if (!is_ws(c)) o.push_back(c);
    return o;
}
inline bool valid_term(const std::string &t) { // is_valid_term, einsum_parser.rs:231-237
    const size_t e = t.find("...");
    std::string letters = t;
    if (e != std::string::npos) {
        if (t.find("...", e + 3) != std::string::npos) return false;
        letters = t.substr(0, e) + t.substr(e + 3);
    }
    for (char c : letters) if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) return false;
    return true;
}

// EinsumExpr::parse, einsum_parser.rs:68-103
inline void parse_equation(const std::string &equation, std::vector<std::string> &terms, std::string &out) {
    const size_t arrow = equation.find("->");
    const std::string lhs = equation.substr(0, arrow);
    terms.clear();
    size_t b = 0;
    for (;;) {
        const size_t c = lhs.find(',', b);
        terms.push_back(strip_ws(lhs.substr(b, c == std::string::npos ? c : c - b)));
        if (c == std::string::npos) break;
        b = c + 1;
    }
    for (auto &t : terms) if (!valid_term(t)) throw OpError(OpError::InvalidValue, "Input term is invalid");
    if (arrow != std::string::npos) out = strip_ws(equation.substr(arrow + 2));
    else { // default_output, :191-228: "..." then the letters used exactly once, in ASCII order
        int count[128] = {0};
        bool dots = false;
        for (auto &t : terms) { dots = dots || t.find("...") != std::string::npos; for (char c : t) if (c != '.') count[(int)c]++; }
        out = dots ? "..." : "";
        for (int c = 0; c < 128; c++) if (count[c] == 1) out.push_back((char)c);
    }
    if (!valid_term(out)) throw OpError(OpError::InvalidValue, "Output term is invalid");
    for (size_t i = 0; i < out.size(); i++)
        if (out[i] != '.' && out.find(out[i], i + 1) != std::string::npos) throw OpError(OpError::InvalidValue, "Einsum output term contains repeated labels");
    for (char c : out) {
        if (c == '.') continue;
        bool found = false;
        for (auto &t : terms) found = found || has(t, c);
        if (!found) throw OpError(OpError::InvalidValue, "Einsum output term contains a label not present in any input term");
    }
}

// EinsumExpr::validate_inputs (einsum_parser.rs:109-165) with the error mapping of einsum.rs:69-86
inline int broadcast_ndim(const std::vector<std::string> &terms, const std::vector<int> &ndims) {
    if (ndims.size() != terms.size()) throw OpError(OpError::InvalidValue, "Number of terms in Einsum equation does not match input tensor count");
    int b = -1;
    for (size_t i = 0; i < terms.size(); i++) {
        const bool dots = terms[i].find("...") != std::string::npos;
        const int named = (int)terms[i].size() - (dots ? 3 : 0), nd = ndims[i];
        if (dots ? nd < named : nd != named) throw OpError(OpError::InvalidValue, "Einsum term dimension count does not match input tensor");
        if (nd > MAX_DIMS) throw OpError(OpError::UnsupportedValue, "Einsum input or term has too many dimensions");
        if (dots) {
            if (b >= 0 && b != nd - named) throw OpError(OpError::InvalidValue, "Number of broadcast dims does not match across inputs");
            b = nd - named;
        }
    }
    return b < 0 ? 0 : b;
}

inline std::string expand_ellipsis(const std::string &t, int n) { // einsum_parser.rs:254-265
    const size_t e = t.find("...");
    if (e == std::string::npos) return t;
    std::string digits;
    for (int i = 0; i < n; i++) digits.push_back((char)('0' + i));
    return t.substr(0, e) + digits + t.substr(e + 3);
}

struct Step { // EinsumStep, einsum.rs:558-564; a source is an input index or -1 for the previous step's result
    std::string lhs;
    int lhs_src = 0;
    bool binary = false;
    std::string rhs;
    int rhs_src = 0;
    std::string out;
};

// einsum_path, einsum.rs:605-692
inline std::vector<Step> plan_path(std::vector<std::string> terms, std::string out, int bdims) {
    out = expand_ellipsis(out, bdims);
    for (auto &t : terms) t = expand_ellipsis(t, bdims);
    std::vector<Step> steps;
    if (terms.size() <= 2) {
        Step s;
        s.lhs = terms[0]; s.binary = terms.size() == 2; s.out = out;
        if (s.binary) { s.rhs = terms[1]; s.rhs_src = 1; }
        steps.push_back(s);
        return steps;
    }
    std::map<char, int> pending; // reduced label -> terms that still have to consume it
    for (auto &t : terms) for (char c : dedup(t)) if (!has(out, c)) pending[c]++;
    auto consume = [&](const std::string &t) { for (char c : dedup(t)) if (pending.count(c)) pending[c]--; };
    auto keep = [&](const std::string &a, const std::string &b) {
        std::string o;
        for (char c : dedup(a + b)) if (has(out, c) || (pending.count(c) && pending[c] > 0)) o.push_back(c);
        return o;
    };
    consume(terms[0]);
    consume(terms[1]);
    std::string cur = keep(terms[0], terms[1]);
    Step first;
    first.lhs = terms[0]; first.binary = true; first.rhs = terms[1]; first.rhs_src = 1; first.out = cur;
    steps.push_back(first);
    for (size_t i = 2; i < terms.size(); i++) {
        consume(terms[i]);
        const std::string nxt = i + 1 == terms.size() ? out : keep(cur, terms[i]);
        Step s;
        s.lhs = cur; s.lhs_src = -1; s.binary = true; s.rhs = terms[i]; s.rhs_src = (int)i; s.out = nxt;
        steps.push_back(s);
        cur = nxt;
    }
    return steps;
}

struct View {
    const float *p = nullptr;
    Shape shape, strides;
    static View of(const Tensor &t) {
        View v;
        v.p = (const float *)t.ptr();
        v.shape = t.shape();
        v.strides.assign(v.shape.size(), 0);
        int64_t acc = 1;
        for (int d = (int)v.shape.size() - 1; d >= 0; d--) { v.strides[(size_t)d] = acc; acc *= v.shape[(size_t)d]; }
        return v;
    }
    int nd() const { return (int)shape.size(); }
    int64_t len() const { return detail::prod(shape, 0, shape.size()); }
    View relabel(const std::string &have, const std::string &want) const { // permute_and_insert_axes, einsum.rs:414-442
        View v;
        v.p = p;
        for (char c : want) {
            const size_t i = have.find(c);
            v.shape.push_back(i == std::string::npos ? 1 : shape[i]);
            v.strides.push_back(i == std::string::npos ? 0 : strides[i]);
        }
        return v;
    }
    View expanded(const Shape &to) const {
        View v = *this;
        for (size_t i = 0; i < to.size(); i++) if (shape[i] == 1 && to[i] != 1) v.strides[i] = 0;
        v.shape = to;
        return v;
    }
};

// drops 1-sized axes and merges neighbours (outer, inner) with outer stride == inner stride * inner size in every operand
inline void merge_axes(const Shape &shape, const std::vector<Shape> &strides, Shape &mshape, std::vector<Shape> &mstrides) {
    mshape.clear();
    mstrides.assign(strides.size(), Shape());
    for (size_t i = 0; i < shape.size(); i++) {
        if (shape[i] == 1) continue;
        bool merge = !mshape.empty();
        for (size_t k = 0; merge && k < strides.size(); k++) merge = mstrides[k].back() == strides[k][i] * shape[i];
        if (merge) { mshape.back() *= shape[i]; for (size_t k = 0; k < strides.size(); k++) mstrides[k].back() = strides[k][i]; }
        else { mshape.push_back(shape[i]); for (size_t k = 0; k < strides.size(); k++) mstrides[k].push_back(strides[k][i]); }
    }
}
inline Shape broadcast(const Shape &a, const Shape &b, const char *msg) {
    Shape o(a.size());
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i] != b[i] && a[i] != 1 && b[i] != 1) throw OpError(OpError::IncompatibleInputShapes, msg);
        o[i] = a[i] == 1 ? b[i] : a[i];
    }
    return o;
}

inline Tensor materialize(Context &ctx, const View &v) { // to_tensor / to_contiguous / expand_to
    Tensor y(ctx, v.shape, DType::F32);
    if (y.len()) {
        Shape ms; std::vector<Shape> mst;
        merge_axes(v.shape, {v.strides}, ms, mst);
        if (ms.size() > 6) throw OpError(OpError::UnsupportedValue, "Einsum view with more than 6 non-mergeable dims is not supported by the device path");
        ctx.check(rten_hip_copy_strided_b32(ctx.raw(), (int)ms.size(), ms.data(), mst[0].data(), v.p, y.ptr()));
    }
    return y;
}

// reduce_sum(view, axes, keep_dims = false): axes sorted and unique
inline Tensor reduce_sum(Context &ctx, const View &v, const std::vector<int> &axes, bool mean = false) {
    Shape ks, kst, rs, rst;
    for (int d = 0; d < v.nd(); d++) {
        const bool red = std::find(axes.begin(), axes.end(), d) != axes.end();
        (red ? rs : ks).push_back(v.shape[(size_t)d]);
        (red ? rst : kst).push_back(v.strides[(size_t)d]);
    }
    Tensor y(ctx, ks, DType::F32);
    if (y.len()) {
        Shape mo, mi; std::vector<Shape> mos, mis;
        merge_axes(ks, {kst}, mo, mos);
        merge_axes(rs, {rst}, mi, mis);
        if (mo.size() > 6 || mi.size() > 6) throw OpError(OpError::UnsupportedValue, "Einsum reduction over more than 6 non-mergeable dims is not supported by the device path");
        ctx.check((mean ? rten_hip_reduce_mean_strided_f32 : rten_hip_reduce_sum_strided_f32)(ctx.raw(), (int)mo.size(), mo.data(), mos[0].data(), (int)mi.size(), mi.data(), mis[0].data(), v.p, (float *)y.ptr()));
    }
    return y;
}

inline Tensor mul(Context &ctx, const View &a, const View &b) { // mul() with numpy broadcasting on views of equal rank
    const Shape shape = broadcast(a.shape, b.shape, "Cannot broadcast inputs");
    Tensor y(ctx, shape, DType::F32);
    if (y.len()) {
        Shape ms; std::vector<Shape> mst;
        merge_axes(shape, {a.expanded(shape).strides, b.expanded(shape).strides}, ms, mst);
        if (ms.size() > 6) throw OpError(OpError::UnsupportedValue, "broadcasting over more than 6 dims is not supported by the device path");
        ctx.check(rten_hip_binary_broadcast_f32(ctx.raw(), 1, (int)ms.size(), ms.data(), mst[0].data(), mst[1].data(), a.p, b.p, (float *)y.ptr()));
    }
    return y;
}

// matmul() of [batch.., M, K] x [batch.., K, N] views of equal rank, src/ops/matmul.rs:208-385
// `into`: [batch.., M, N] view (N contiguous) of the caller's output tensor; the GEMM then writes the permuted output directly
// through ldc and the C batch strides.  Returns false when those strides do not fit the descriptor's two batch levels.
inline bool matmul_into(Context &ctx, View a, View b, const View &cv) {
    const int nd = a.nd();
    const int64_t m = a.shape[(size_t)nd - 2], k = a.shape[(size_t)nd - 1], n = b.shape[(size_t)nd - 1];
    if (k != b.shape[(size_t)nd - 2]) throw OpError(OpError::IncompatibleInputShapes, "Columns of first matrix does not match rows of second matrix");
    const Shape apre(a.shape.begin(), a.shape.end() - 2), bpre(b.shape.begin(), b.shape.end() - 2);
    const Shape pre = broadcast(apre, bpre, "Cannot broadcast shapes");
    float *const c = const_cast<float *>(cv.p);
    if (cv.len() == 0) return true;
    if (k == 0) { ctx.check(rten_hip_memset(ctx.raw(), c, 0, (size_t)cv.len() * 4)); return true; } // the output tensor is exactly this view's elements
    std::vector<Tensor> keep; // re-laid operands stay alive until the launch is enqueued
    auto relay = [&](View &v) { keep.push_back(materialize(ctx, v)); v = View::of(keep.back()); };
    auto gemm_axis_is_broadcast = [&](const View &v) {
        for (int d = nd - 2; d < nd; d++) if (v.strides[(size_t)d] == 0 && v.shape[(size_t)d] > 1) return true;
        return false;
    };
    if (gemm_axis_is_broadcast(a)) relay(a); // expand_dim, einsum.rs:210-223
    if (gemm_axis_is_broadcast(b)) relay(b);
    rten_hip_gemm_desc d;
    std::memset(&d, 0, sizeof d);
    d.n = (int32_t)n; d.k = (int32_t)k; d.ldc = std::max(cv.strides[(size_t)nd - 2], n); d.alpha = 1.0f; d.batch = 1;
    const int64_t na = detail::prod(apre, 0, apre.size()), nb = detail::prod(bpre, 0, bpre.size());
    Shape ms; std::vector<Shape> mst;
    Shape crows_shape = pre, crows, crows_strides(cv.strides.begin(), cv.strides.end() - 1);
    crows_shape.push_back(m);
    std::vector<Shape> crows_st;
    merge_axes(crows_shape, {crows_strides}, crows, crows_st);
    if (na > 1 && nb == 1 && crows.size() <= 1) { // [A, M, K] x [K, N] as one GEMM of A * M rows (matmul.rs:266-297)
        auto rows = [&] { merge_axes(Shape(a.shape.begin(), a.shape.end() - 1), {Shape(a.strides.begin(), a.strides.end() - 1)}, ms, mst); };
        rows();
        if (ms.size() > 1) { relay(a); rows(); }
        d.m = (int32_t)(na * m);
        d.a_rs = ms.empty() ? 0 : mst[0][0]; d.a_cs = a.strides[(size_t)nd - 1];
        d.b_rs = b.strides[(size_t)nd - 2]; d.b_cs = b.strides[(size_t)nd - 1];
        d.ldc = crows.empty() ? n : std::max(crows_st[0][0], n);
        ctx.check(rten_hip_gemm_f32(ctx.raw(), &d, a.p, b.p, nullptr, c));
        return true;
    }
    Shape ea_shape = pre, eb_shape = pre;
    ea_shape.push_back(m); ea_shape.push_back(k);
    eb_shape.push_back(k); eb_shape.push_back(n);
    View ea = a.expanded(ea_shape), eb = b.expanded(eb_shape);
    auto batch_strides = [&](const View &v) { return Shape(v.strides.begin(), v.strides.end() - 2); };
    merge_axes(pre, {batch_strides(ea), batch_strides(eb), batch_strides(cv)}, ms, mst);
    if (ms.size() > 2) { // more batch levels than the descriptor has: re-lay the operand(s) that do not merge by themselves
        auto needs_relay = [&](const View &v) {
            Shape s1; std::vector<Shape> st1;
            merge_axes(pre, {batch_strides(v)}, s1, st1);
            return s1.size() > 1 || (!st1[0].empty() && st1[0][0] == 0);
        };
        if (needs_relay(ea)) relay(ea);
        if (needs_relay(eb)) relay(eb);
        merge_axes(pre, {batch_strides(ea), batch_strides(eb), batch_strides(cv)}, ms, mst);
        if (ms.size() > 2) return false; // only a permuted C can still refuse to merge: the caller multiplies, then copies
    }
    d.m = (int32_t)m;
    d.a_rs = ea.strides[(size_t)nd - 2]; d.a_cs = ea.strides[(size_t)nd - 1];
    d.b_rs = eb.strides[(size_t)nd - 2]; d.b_cs = eb.strides[(size_t)nd - 1];
    d.batch = (int32_t)detail::prod(ms, 0, ms.size());
    if (ms.size() == 2) {
        d.a_bs = mst[0][0]; d.b_bs = mst[1][0]; d.c_bs = mst[2][0];
        d.batch_inner = (int32_t)ms[1]; d.a_bsi = mst[0][1]; d.b_bsi = mst[1][1]; d.c_bsi = mst[2][1];
    } else if (ms.size() == 1) {
        d.a_bs = mst[0][0]; d.b_bs = mst[1][0]; d.c_bs = mst[2][0];
    }
    ctx.check(rten_hip_gemm_f32(ctx.raw(), &d, ea.p, eb.p, nullptr, c));
    return true;
}
inline Tensor matmul(Context &ctx, const View &a, const View &b) {
    const int nd = a.nd();
    Shape oshape = broadcast(Shape(a.shape.begin(), a.shape.end() - 2), Shape(b.shape.begin(), b.shape.end() - 2), "Cannot broadcast shapes");
    oshape.push_back(a.shape[(size_t)nd - 2]);
    oshape.push_back(b.shape[(size_t)nd - 1]);
    Tensor y(ctx, oshape, DType::F32);
    matmul_into(ctx, a, b, View::of(y)); // a contiguous C always fits
    return y;
}

inline View diagonals(std::string &term, const View &v) { // take_diagonals, einsum.rs:124-162
    const std::string labels = dedup(term);
    View o;
    o.p = v.p;
    for (char c : labels) {
        int64_t size = -1, stride = 0;
        for (size_t i = 0; i < term.size(); i++) {
            if (term[i] != c) continue;
            if (size >= 0 && v.shape[i] != size) throw OpError(OpError::InvalidValue, "Dimension sizes for repeated labels in term do not match");
            size = v.shape[i];
            stride += v.strides[i];
        }
        o.shape.push_back(size);
        o.strides.push_back(stride);
    }
    term = labels;
    return o;
}

inline int64_t bsize(int64_t a, int64_t b) { // broadcast_size, einsum.rs:197-205
    if (a == b || b == 1) return a;
    if (a == 1) return b;
    throw OpError(OpError::IncompatibleInputShapes, "Einsum label has different sizes in different terms");
}

// einsum_matmul, einsum.rs:449-537
inline Tensor contract(Context &ctx, const View &x, const View &y, const std::string &tx, const std::string &ty, const std::string &out, char kl) {
    char nl = INS_N, ml = INS_M;
    for (size_t i = ty.size(); i-- > 0;) if (!has(tx, ty[i])) { nl = ty[i]; break; }
    for (size_t i = tx.size(); i-- > 0;) if (!has(ty, tx[i])) { ml = tx[i]; break; }
    std::string batch;
    for (char c : dedup(tx + ty)) if (c != kl && c != ml && c != nl) batch.push_back(c);
    View xv = x.relabel(tx, batch + ml + kl), yv = y.relabel(ty, batch + kl + nl);
    const int64_t ks = bsize(xv.shape.back(), yv.shape[yv.shape.size() - 2]);
    Shape xs = xv.shape, ys = yv.shape;
    xs.back() = ks;
    ys[ys.size() - 2] = ks;
    const std::string full = batch + ml + nl;
    std::string order;
    for (char c : full) if (c != INS_M && c != INS_N) order.push_back(c);
    if (order != out && (nl == INS_N || out.back() == nl)) {
        // the output permutation keeps N innermost: the GEMM writes the permuted tensor directly (C strides), no copy
        Shape fshape;
        for (char c : out) {
            const size_t i = full.find(c);
            fshape.push_back(i + 2 == full.size() ? xs[xs.size() - 2] : i + 1 == full.size() ? ys.back() : (xs[i] == 1 ? ys[i] : xs[i]));
        }
        Tensor final(ctx, fshape, DType::F32);
        if (matmul_into(ctx, xv.expanded(xs), yv.expanded(ys), View::of(final).relabel(out, full))) return final;
    }
    Tensor r = matmul(ctx, xv.expanded(xs), yv.expanded(ys));
    Shape shape;
    for (size_t i = 0; i < full.size(); i++) if (full[i] != INS_M && full[i] != INS_N) shape.push_back(r.size((int)i));
    r.reshape(shape);
    if (order == out) return r;
    return materialize(ctx, View::of(r).relabel(order, out));
}

// einsum_step, einsum.rs:238-364
inline Tensor run_step(Context &ctx, const Step &s, View x, const View *yin) {
    std::string tx = s.lhs, ty = s.rhs;
    const std::string &out = s.out;
    x = diagonals(tx, x);
    auto reduced = [&](const std::string &labels) { std::string r; for (char c : labels) if (!has(out, c)) r.push_back(c); return r; };
    auto trailing = [&](size_t from, size_t count) { std::vector<int> a; for (size_t i = 0; i < count; i++) a.push_back((int)(from + i)); return a; };
    if (!s.binary) {
        const std::string red = reduced(tx);
        const View xv = x.relabel(tx, out + red);
        return red.empty() ? materialize(ctx, xv) : reduce_sum(ctx, xv, trailing(out.size(), red.size()));
    }
    View y = diagonals(ty, *yin);
    std::vector<Tensor> keep;
    auto drop_lone = [&](View &v, std::string &term, const std::string &other) { // sum_lone_dims, einsum.rs:168-190
        std::vector<int> lone;
        std::string kept;
        for (size_t i = 0; i < term.size(); i++) {
            if (has(other, term[i]) || has(out, term[i])) kept.push_back(term[i]);
            else lone.push_back((int)i);
        }
        if (!lone.empty()) { keep.push_back(reduce_sum(ctx, v, lone)); v = View::of(keep.back()); }
        term = kept;
    };
    drop_lone(x, tx, ty);
    drop_lone(y, ty, tx);
    const std::string red = reduced(dedup(tx + ty));
    if (red.size() == 1) return contract(ctx, x, y, tx, ty, out, red[0]);
    const View xv = x.relabel(tx, out + red), yv = y.relabel(ty, out + red);
    if (red.empty()) return mul(ctx, xv, yv);
    Shape xs = xv.shape, ys = yv.shape; // several reduced labels: re-laid side by side and merged into one K (einsum.rs:310-344)
    int64_t ksz = 1;
    for (size_t i = out.size(); i < xs.size(); i++) { xs[i] = ys[i] = bsize(xs[i], ys[i]); ksz *= xs[i]; }
    Tensor xc = materialize(ctx, xv.expanded(xs)), yc = materialize(ctx, yv.expanded(ys));
    xs.resize(out.size()); xs.push_back(ksz);
    ys.resize(out.size()); ys.push_back(ksz);
    xc.reshape(xs);
    yc.reshape(ys);
    return contract(ctx, View::of(xc), View::of(yc), out + MERGED_K, out + MERGED_K, out, MERGED_K);
}
} // namespace einsum_detail

struct ReduceSum : Operator { // src/ops/reduce.rs:1126-1165 (f32); axes as the attribute (or the resolved second input)
    std::vector<int> axes;
    bool keep_dims = true, noop_with_empty_axes = false, mean = false;
    const char *name() const override { return mean ? "ReduceMean" : "ReduceSum"; }
    int max_inputs() const override { return 2; }
    OutputList run(Context &ctx, const InputList &in) const override {
        namespace E = einsum_detail;
        const Tensor &x = want(require(in, 0), DType::F32, "float32");
        const int nd = x.ndim();
        OutputList out;
        if ((axes.empty() && noop_with_empty_axes) || nd == 0) { out.push_back(E::materialize(ctx, E::View::of(x))); return out; }
        std::vector<int> ax;
        if (axes.empty()) for (int d = 0; d < nd; d++) ax.push_back(d);
        for (int a : axes) ax.push_back(resolve_axis(a, nd));
        std::sort(ax.begin(), ax.end());
        ax.erase(std::unique(ax.begin(), ax.end()), ax.end()); // resolve_axes, src/ops/mod.rs:259-271
        Tensor y = E::reduce_sum(ctx, E::View::of(x), ax, mean);
        if (keep_dims) {
            std::vector<int64_t> s = x.shape();
            for (int a : ax) s[(size_t)a] = 1;
            y.reshape(s);
        }
        out.push_back(std::move(y));
        return out;
    }
};

struct ReduceMean : ReduceSum { // src/ops/reduce.rs:523-580: Sum / len per slice
    ReduceMean() { mean = true; }
};

struct Einsum : Operator { // src/ops/einsum.rs:21-108
    std::string equation;
    const char *name() const override { return "Einsum"; }
    int max_inputs() const override { return -1; }
    OutputList run(Context &ctx, const InputList &in) const override {
        namespace E = einsum_detail;
        std::vector<int> ndims;
        for (size_t i = 0; i < in.size(); i++) ndims.push_back(want(require(in, i), DType::F32, "float32").ndim());
        std::vector<std::string> terms;
        std::string out;
        E::parse_equation(equation, terms, out);
        const int bdims = E::broadcast_ndim(terms, ndims);
        Tensor result;
        for (const E::Step &s : E::plan_path(terms, out, bdims)) {
            const E::View x = s.lhs_src < 0 ? E::View::of(result) : E::View::of(*in[(size_t)s.lhs_src]);
            E::View y;
            if (s.binary) y = s.rhs_src < 0 ? E::View::of(result) : E::View::of(*in[(size_t)s.rhs_src]);
            Tensor next = E::run_step(ctx, s, x, s.binary ? &y : nullptr);
            result = std::move(next); // the previous result is released only after the step that read it was enqueued
        }
        OutputList o;
        o.push_back(std::move(result));
        return o;
    }
};

// ------------------------------------------------------------------------------------------------ registry (src/op_registry.rs:25-72)
class OpRegistry {
  public:
    using Factory = std::function<std::unique_ptr<Operator>()>;
    template <typename Op> void register_op(const std::string &name) { factories_[name] = [] { return std::unique_ptr<Operator>(new Op()); }; }
    std::unique_ptr<Operator> create(const std::string &name) const {
        auto it = factories_.find(name);
        if (it == factories_.end()) throw OpError(OpError::UnsupportedValue, "operator not registered: " + name); // ReadOpError::OperatorUnavailable
        return it->second();
    }
    bool contains(const std::string &name) const { return factories_.count(name) != 0; }
    std::vector<std::string> names() const {
        std::vector<std::string> v;
        for (auto &kv : factories_) v.push_back(kv.first);
        return v;
    }
    static OpRegistry with_all_ops() { // every hot-path operator this backend overrides
        OpRegistry r;
        r.register_op<Conv>("Conv");
        r.register_op<ConvTranspose>("ConvTranspose");
        r.register_op<ConvInteger>("ConvInteger");
        r.register_op<ConvIntegerToFloat>("ConvIntegerToFloat");
        r.register_op<MatMul>("MatMul");
        r.register_op<FusedMatMul>("FusedMatMul");
        r.register_op<Gemm>("Gemm");
        r.register_op<MatMulInteger>("MatMulInteger");
        r.register_op<MatMulNBits>("MatMulNBits");
        r.register_op<Softmax>("Softmax");
        r.register_op<LayerNormalization>("LayerNormalization");
        r.register_op<Relu>("Relu");
        r.register_op<Gelu>("Gelu");
        r.register_op<Erf>("Erf");
        r.register_op<Add>("Add");
        r.register_op<Mul>("Mul");
        r.register_op<Sub>("Sub");
        r.register_op<Div>("Div");
        r.register_op<Transpose>("Transpose");
        r.register_op<Gather>("Gather");
        r.register_op<AddSoftmax>("AddSoftmax");
        r.register_op<MaxPool>("MaxPool");
        r.register_op<AveragePool>("AveragePool");
        r.register_op<GlobalAveragePool>("GlobalAveragePool");
        r.register_op<DynamicQuantizeLinear>("DynamicQuantizeLinear");
        r.register_op<Cast>("Cast");
        r.register_op<Tanh>("Tanh");
        r.register_op<Where>("Where");
        r.register_op<ReduceSum>("ReduceSum");
        r.register_op<ReduceMean>("ReduceMean");
        r.register_op<Einsum>("Einsum");
        return r;
    }

  private:
    std::map<std::string, Factory> factories_;
};

} // namespace rten_hip
