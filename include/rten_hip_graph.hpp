// rten_hip_graph.hpp -- ONNX loader + device-resident graph executor above rten_hip_ops.hpp (the
// "residency-aware executor glue" either side of the hot-path operators).
//
// What it mirrors in the reference, and where it stops:
//   * onnx::parse        rten-onnx/src/onnx.rs (hand-rolled protobuf reader, same subset: ModelProto / GraphProto /
//                        NodeProto / AttributeProto / TensorProto / ValueInfoProto), src/model/onnx_loader.rs
//                        (int64 initializers are narrowed to int32 at load, onnx_loader.rs:332-339).
//   * Graph::compile     Graph::prepack_weights (src/graph.rs:488-562: constant conv weights are staged once) and the
//                        fusion passes that touch this backend's operators (src/optimize/fusions.rs):
//                        Conv (+ Add residual) (+ Relu); ConvInteger -> Cast -> Mul(scale) (= ConvIntegerToFloat,
//                        fusions.rs:1012-1058) (+ Add bias [1,O,1,1]) (+ Add residual) (+ Relu); MatMulInteger -> Cast -> Mul
//                        (= MatMulIntegerToFloat); Reshape / Flatten / Squeeze / Unsqueeze / Identity as views.
//   * Graph::run         Graph::run_plan (src/graph.rs:1139-1231): operators run sequentially in plan order, values are
//                        reference counted and their buffers go back to the pool when the last consumer has run
//                        (buffer_pool.rs); here every value stays in HBM between operators -- only graph inputs and
//                        outputs cross PCIe.
// Scope of this first version: the CNN-classifier graphs of BASELINE configs 0-2 (ResNet-50 f32 and its
// dynamically-quantized form).  An operator outside the registry is a load-time error naming the node, never a CPU
// fallback.  Control flow, sequences, symbolic shape inference and the transformer-graph layout ops are not built.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <set>

#include "rten_hip_ops.hpp"

namespace rten_hip {

// ======================================================================================================== ONNX reader
namespace onnx {

enum DataType { FLOAT = 1, UINT8 = 2, INT8 = 3, INT32 = 6, INT64 = 7, BOOL = 9 };

struct TensorProto {
    std::string name;
    std::vector<int64_t> dims;
    int data_type = 0;
    std::string raw;                 // raw_data, or the typed fields re-encoded little-endian
    int64_t len() const { int64_t n = 1; for (int64_t d : dims) n *= d; return n; }
};
struct Attr {
    std::string name;
    int type = 0; // AttributeProto.AttributeType
    float f = 0.f;
    int64_t i = 0;
    std::string s;
    std::vector<float> floats;
    std::vector<int64_t> ints;
    TensorProto t;
};
struct Node {
    std::string op_type, name, domain;
    std::vector<std::string> inputs, outputs;
    std::vector<Attr> attrs;
    const Attr *attr(const std::string &n) const {
        for (auto &a : attrs) if (a.name == n) return &a;
        return nullptr;
    }
    int64_t get_int(const std::string &n, int64_t dflt) const { const Attr *a = attr(n); return a ? a->i : dflt; }
    float get_float(const std::string &n, float dflt) const { const Attr *a = attr(n); return a ? a->f : dflt; }
    std::vector<int> get_ints(const std::string &n, std::vector<int> dflt) const {
        const Attr *a = attr(n);
        if (!a) return dflt;
        return std::vector<int>(a->ints.begin(), a->ints.end());
    }
};
struct ValueInfo {
    std::string name;
    int elem_type = 0;
    std::vector<int64_t> dims; // -1: symbolic
    std::vector<std::string> dim_params;
};
struct Model {
    int64_t ir_version = 0, opset = 0;
    std::string producer, graph_name;
    std::vector<Node> nodes;
    std::vector<TensorProto> initializers;
    std::vector<ValueInfo> inputs, outputs;
};

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

// protobuf wire format: (field << 3 | wire type) varint keys; wire 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32
class Reader {
  public:
    Reader(const uint8_t *p, size_t n) : p_(p), end_(p + n) {}
    bool done() const { return p_ >= end_; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 70; shift += 7) {
            if (p_ >= end_) throw ParseError("onnx: truncated varint");
            const uint8_t b = *p_++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        throw ParseError("onnx: varint too long");
    }
    void key(int &field, int &wire) { const uint64_t k = varint(); field = (int)(k >> 3); wire = (int)(k & 7); }
    Reader sub() {
        const uint64_t n = varint();
        if (n > (uint64_t)(end_ - p_)) throw ParseError("onnx: length-delimited field overruns its message");
        Reader r(p_, (size_t)n);
        p_ += n;
        return r;
    }
    std::string str() { Reader r = sub(); return std::string((const char *)r.p_, (size_t)(r.end_ - r.p_)); }
    uint32_t fixed32() { if (end_ - p_ < 4) throw ParseError("onnx: truncated fixed32"); uint32_t v; std::memcpy(&v, p_, 4); p_ += 4; return v; }
    uint64_t fixed64() { if (end_ - p_ < 8) throw ParseError("onnx: truncated fixed64"); uint64_t v; std::memcpy(&v, p_, 8); p_ += 8; return v; }
    void skip(int wire) {
        if (wire == 0) varint();
        else if (wire == 1) fixed64();
        else if (wire == 2) sub();
        else if (wire == 5) fixed32();
        else throw ParseError("onnx: unsupported wire type");
    }
    // repeated scalar: packed (wire 2) or one element (wire 0 / 5)
    template <typename F> void repeated(int wire, int scalar_wire, F f) {
        if (wire == 2) { Reader r = sub(); while (!r.done()) f(r); }
        else if (wire == scalar_wire) f(*this);
        else throw ParseError("onnx: unexpected wire type in a repeated scalar");
    }

  private:
    const uint8_t *p_, *end_;
};

inline TensorProto parse_tensor(Reader r) {
    TensorProto t;
    std::vector<float> fdata;
    std::vector<int64_t> i32data, i64data;
    bool has_raw = false;
    while (!r.done()) {
        int f, w;
        r.key(f, w);
        if (f == 1) r.repeated(w, 0, [&](Reader &q) { t.dims.push_back((int64_t)q.varint()); });
        else if (f == 2 && w == 0) t.data_type = (int)r.varint();
        else if (f == 4) r.repeated(w, 5, [&](Reader &q) { const uint32_t b = q.fixed32(); float v; std::memcpy(&v, &b, 4); fdata.push_back(v); });
        else if (f == 5) r.repeated(w, 0, [&](Reader &q) { i32data.push_back((int64_t)q.varint()); });
        else if (f == 7) r.repeated(w, 0, [&](Reader &q) { i64data.push_back((int64_t)q.varint()); });
        else if (f == 8 && w == 2) t.name = r.str();
        else if (f == 9 && w == 2) { t.raw = r.str(); has_raw = true; }
        else if (f == 13 || f == 14) throw ParseError("onnx: external tensor data is not supported (" + t.name + ")");
        else r.skip(w);
    }
    if (!has_raw) { // typed fields -> little-endian bytes of the declared type
        auto put = [&](const void *p, size_t n) { t.raw.append((const char *)p, n); };
        if (t.data_type == FLOAT) for (float v : fdata) put(&v, 4);
        else if (t.data_type == INT64) for (int64_t v : i64data) put(&v, 8);
        else if (t.data_type == INT32) for (int64_t v : i32data) { const int32_t x = (int32_t)v; put(&x, 4); }
        else if (t.data_type == UINT8 || t.data_type == INT8 || t.data_type == BOOL) for (int64_t v : i32data) { const uint8_t x = (uint8_t)v; put(&x, 1); }
    }
    return t;
}

inline Attr parse_attr(Reader r) {
    Attr a;
    while (!r.done()) {
        int f, w;
        r.key(f, w);
        if (f == 1 && w == 2) a.name = r.str();
        else if (f == 2 && w == 5) { const uint32_t b = r.fixed32(); std::memcpy(&a.f, &b, 4); }
        else if (f == 3 && w == 0) a.i = (int64_t)r.varint();
        else if (f == 4 && w == 2) a.s = r.str();
        else if (f == 5 && w == 2) a.t = parse_tensor(r.sub());
        else if (f == 7) r.repeated(w, 5, [&](Reader &q) { const uint32_t b = q.fixed32(); float v; std::memcpy(&v, &b, 4); a.floats.push_back(v); });
        else if (f == 8) r.repeated(w, 0, [&](Reader &q) { a.ints.push_back((int64_t)q.varint()); });
        else if (f == 20 && w == 0) a.type = (int)r.varint();
        else r.skip(w);
    }
    return a;
}

inline Node parse_node(Reader r) {
    Node n;
    while (!r.done()) {
        int f, w;
        r.key(f, w);
        if (f == 1 && w == 2) n.inputs.push_back(r.str());
        else if (f == 2 && w == 2) n.outputs.push_back(r.str());
        else if (f == 3 && w == 2) n.name = r.str();
        else if (f == 4 && w == 2) n.op_type = r.str();
        else if (f == 5 && w == 2) n.attrs.push_back(parse_attr(r.sub()));
        else if (f == 7 && w == 2) n.domain = r.str();
        else r.skip(w);
    }
    return n;
}

inline ValueInfo parse_value_info(Reader r) {
    ValueInfo v;
    while (!r.done()) {
        int f, w;
        r.key(f, w);
        if (f == 1 && w == 2) v.name = r.str();
        else if (f == 2 && w == 2) { // TypeProto
            Reader tp = r.sub();
            while (!tp.done()) {
                int f2, w2;
                tp.key(f2, w2);
                if (f2 != 1 || w2 != 2) { tp.skip(w2); continue; } // tensor_type only
                Reader tt = tp.sub();
                while (!tt.done()) {
                    int f3, w3;
                    tt.key(f3, w3);
                    if (f3 == 1 && w3 == 0) v.elem_type = (int)tt.varint();
                    else if (f3 == 2 && w3 == 2) { // TensorShapeProto
                        Reader sh = tt.sub();
                        while (!sh.done()) {
                            int f4, w4;
                            sh.key(f4, w4);
                            if (f4 != 1 || w4 != 2) { sh.skip(w4); continue; }
                            Reader dm = sh.sub();
                            int64_t val = -1;
                            std::string param;
                            while (!dm.done()) {
                                int f5, w5;
                                dm.key(f5, w5);
                                if (f5 == 1 && w5 == 0) val = (int64_t)dm.varint();
                                else if (f5 == 2 && w5 == 2) param = dm.str();
                                else dm.skip(w5);
                            }
                            v.dims.push_back(val);
                            v.dim_params.push_back(param);
                        }
                    } else tt.skip(w3);
                }
            }
        } else r.skip(w);
    }
    return v;
}

inline Model parse(const uint8_t *data, size_t n) {
    Model m;
    Reader r(data, n);
    bool have_graph = false;
    while (!r.done()) {
        int f, w;
        r.key(f, w);
        if (f == 1 && w == 0) m.ir_version = (int64_t)r.varint();
        else if (f == 2 && w == 2) m.producer = r.str();
        else if (f == 8 && w == 2) { // OperatorSetIdProto
            Reader o = r.sub();
            std::string domain;
            int64_t version = 0;
            while (!o.done()) {
                int f2, w2;
                o.key(f2, w2);
                if (f2 == 1 && w2 == 2) domain = o.str();
                else if (f2 == 2 && w2 == 0) version = (int64_t)o.varint();
                else o.skip(w2);
            }
            if (domain.empty() || domain == "ai.onnx") m.opset = version;
        } else if (f == 7 && w == 2) {
            have_graph = true;
            Reader g = r.sub();
            while (!g.done()) {
                int f2, w2;
                g.key(f2, w2);
                if (f2 == 1 && w2 == 2) m.nodes.push_back(parse_node(g.sub()));
                else if (f2 == 2 && w2 == 2) m.graph_name = g.str();
                else if (f2 == 5 && w2 == 2) m.initializers.push_back(parse_tensor(g.sub()));
                else if (f2 == 11 && w2 == 2) m.inputs.push_back(parse_value_info(g.sub()));
                else if (f2 == 12 && w2 == 2) m.outputs.push_back(parse_value_info(g.sub()));
                else g.skip(w2);
            }
        } else r.skip(w);
    }
    if (!have_graph) throw ParseError("onnx: model has no graph");
    // graph inputs that are initializers are constants, not runtime inputs (IR < 4 models list both)
    std::set<std::string> init_names;
    for (auto &t : m.initializers) init_names.insert(t.name);
    m.inputs.erase(std::remove_if(m.inputs.begin(), m.inputs.end(), [&](const ValueInfo &v) { return init_names.count(v.name) != 0; }), m.inputs.end());
    return m;
}

inline Model load(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw ParseError("onnx: cannot open " + path);
    std::string buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse((const uint8_t *)buf.data(), buf.size());
}

} // namespace onnx


// ======================================================================================================== shape arithmetic on the host
// Exporter-written graphs (PyTorch / transformers: dynamic-axis `view`s, position-id slices, attention-mask expansion) carry subgraphs whose VALUES depend only
// on input SHAPES and constants: Shape -> Gather -> Unsqueeze -> Concat -> Reshape, ConstantOfShape -> NonZero (an arange), index arithmetic, comparisons.
// The reference folds them at load against its symbolic shapes (`propagate_constants`, src/optimize.rs:705; shape inference in src/infer_shapes.rs).  This
// executor has no shape inference: it evaluates them ON THE HOST while a run walks the plan -- a step whose operands all carry a HostVal computes its result
// on the host and attaches it to a (cached) device copy; Reshape / Expand / Slice / ConstantOfShape read their shape operands from the HostVal.  Shapes are
// fixed between `prepare` and the next `bind_input`, so the cached device copies are made once, in the eager warm-up run, and a captured hipGraph contains
// none of this.  Semantics follow the reference's operators (int32 everywhere: onnx_loader.rs:332-339; wrapping integer arithmetic; booleans 0 / 1).
namespace hostops {
constexpr int64_t kMaxHostElems = (int64_t)1 << 22; // beyond this a value is not worth mirroring on the host

inline int64_t wrap32(int64_t v) { return (int64_t)(int32_t)(uint32_t)(uint64_t)v; }
inline std::vector<int64_t> strides_for(const std::vector<int64_t> &shape, const std::vector<int64_t> &out) { // broadcast strides of `shape` on `out`
    std::vector<int64_t> st(out.size(), 0);
    int64_t acc = 1;
    for (size_t i = shape.size(); i-- > 0;) { st[out.size() - shape.size() + i] = shape[i] == 1 ? 0 : acc; acc *= shape[i]; }
    return st;
}
inline std::vector<int64_t> bshape(const std::vector<const HostVal *> &v) {
    std::vector<const std::vector<int64_t> *> sh;
    for (auto *x : v) sh.push_back(&x->shape);
    return broadcast_shapes(sh).shape;
}
// calls f(out_index, offsets[k]) for every element of the broadcast shape
template <typename F> void for_each_broadcast(const std::vector<int64_t> &out, const std::vector<std::vector<int64_t>> &st, F f) {
    int64_t n = 1;
    for (int64_t d : out) n *= d;
    std::vector<int64_t> idx(out.size(), 0), off(st.size(), 0);
    for (int64_t i = 0; i < n; i++) {
        f(i, off);
        for (size_t d = out.size(); d-- > 0;) {
            idx[d]++;
            for (size_t k = 0; k < st.size(); k++) off[k] += st[k][d];
            if (idx[d] < out[d]) break;
            for (size_t k = 0; k < st.size(); k++) off[k] -= st[k][d] * out[d];
            idx[d] = 0;
        }
    }
}
inline HostVal make_ints(std::vector<int64_t> shape, std::vector<int64_t> v) { HostVal h; h.shape = std::move(shape); h.i = std::move(v); return h; }

// ---- element-wise: Add Sub Mul Div (both types), And Or Xor, Equal Less LessOrEqual Greater GreaterOrEqual
inline bool binary(const std::string &op, const HostVal &a, const HostVal &b, HostVal &out) {
    if (a.is_float != b.is_float) return false;
    out.shape = bshape({&a, &b});
    if (out.len() > kMaxHostElems) return false;
    const std::vector<std::vector<int64_t>> st{strides_for(a.shape, out.shape), strides_for(b.shape, out.shape)};
    const bool arith = op == "Add" || op == "Sub" || op == "Mul" || op == "Div";
    const bool logic = op == "And" || op == "Or" || op == "Xor";
    if (logic && a.is_float) return false;
    out.is_float = arith && a.is_float;
    const int64_t n = out.len();
    if (out.is_float) out.f.resize((size_t)n); else out.i.resize((size_t)n);
    bool ok = true;
    for_each_broadcast(out.shape, st, [&](int64_t i, const std::vector<int64_t> &o) {
        if (a.is_float) {
            const float x = a.f[(size_t)o[0]], y = b.f[(size_t)o[1]];
            if (op == "Add") out.f[(size_t)i] = x + y; else if (op == "Sub") out.f[(size_t)i] = x - y; else if (op == "Mul") out.f[(size_t)i] = x * y;
            else if (op == "Div") out.f[(size_t)i] = x / y;
            else if (op == "Equal") out.i[(size_t)i] = x == y; else if (op == "Less") out.i[(size_t)i] = x < y; else if (op == "LessOrEqual") out.i[(size_t)i] = x <= y;
            else if (op == "Greater") out.i[(size_t)i] = x > y; else if (op == "GreaterOrEqual") out.i[(size_t)i] = x >= y; else ok = false;
        } else {
            const int64_t x = a.i[(size_t)o[0]], y = b.i[(size_t)o[1]];
            int64_t r = 0;
            if (op == "Add") r = wrap32(x + y); else if (op == "Sub") r = wrap32(x - y); else if (op == "Mul") r = wrap32(x * y);
            else if (op == "Div") { if (y == 0) throw OpError(OpError::InvalidValue, "integer division by zero"); r = wrap32(x / y); }
            else if (op == "And") r = (x != 0) && (y != 0); else if (op == "Or") r = (x != 0) || (y != 0); else if (op == "Xor") r = (x != 0) != (y != 0);
            else if (op == "Equal") r = x == y; else if (op == "Less") r = x < y; else if (op == "LessOrEqual") r = x <= y;
            else if (op == "Greater") r = x > y; else if (op == "GreaterOrEqual") r = x >= y; else ok = false;
            out.i[(size_t)i] = r;
        }
    });
    return ok;
}
inline bool where(const HostVal &c, const HostVal &x, const HostVal &y, HostVal &out) {
    if (c.is_float || x.is_float != y.is_float) return false;
    out.shape = bshape({&c, &x, &y});
    if (out.len() > kMaxHostElems) return false;
    out.is_float = x.is_float;
    const std::vector<std::vector<int64_t>> st{strides_for(c.shape, out.shape), strides_for(x.shape, out.shape), strides_for(y.shape, out.shape)};
    if (out.is_float) out.f.resize((size_t)out.len()); else out.i.resize((size_t)out.len());
    for_each_broadcast(out.shape, st, [&](int64_t i, const std::vector<int64_t> &o) {
        const bool t = c.i[(size_t)o[0]] != 0;
        if (out.is_float) out.f[(size_t)i] = t ? x.f[(size_t)o[1]] : y.f[(size_t)o[2]];
        else out.i[(size_t)i] = t ? x.i[(size_t)o[1]] : y.i[(size_t)o[2]];
    });
    return true;
}
inline HostVal expand(const HostVal &x, const std::vector<int64_t> &target) {
    HostVal out;
    out.is_float = x.is_float;
    HostVal t; t.shape = target;
    out.shape = bshape({&x, &t});
    const std::vector<std::vector<int64_t>> st{strides_for(x.shape, out.shape)};
    if (out.is_float) out.f.resize((size_t)out.len()); else out.i.resize((size_t)out.len());
    for_each_broadcast(out.shape, st, [&](int64_t i, const std::vector<int64_t> &o) { if (out.is_float) out.f[(size_t)i] = x.f[(size_t)o[0]]; else out.i[(size_t)i] = x.i[(size_t)o[0]]; });
    return out;
}
inline HostVal cast(const HostVal &x, DType to) { // src/ops/convert.rs: Rust `as`
    HostVal out;
    out.shape = x.shape;
    out.is_float = to == DType::F32;
    const size_t n = (size_t)x.len();
    if (out.is_float) { out.f.resize(n); for (size_t k = 0; k < n; k++) out.f[k] = x.is_float ? x.f[k] : (float)(int32_t)x.i[k]; return out; }
    out.i.resize(n);
    for (size_t k = 0; k < n; k++) {
        if (x.is_float) {
            const float f = x.f[k];
            const double lo = to == DType::I32 ? -2147483648.0 : to == DType::U8 ? 0.0 : -128.0, hi = to == DType::I32 ? 2147483647.0 : to == DType::U8 ? 255.0 : 127.0;
            out.i[k] = f != f ? 0 : (int64_t)std::trunc(std::min(std::max((double)f, lo), hi));
        } else {
            const int64_t v = x.i[k];
            out.i[k] = to == DType::I32 ? wrap32(v) : to == DType::U8 ? (v & 0xff) : (int64_t)(int8_t)(v & 0xff);
        }
    }
    return out;
}
inline HostVal reshaped(const HostVal &x, std::vector<int64_t> shape) { HostVal o = x; o.shape = std::move(shape); return o; }
inline HostVal transpose(const HostVal &x, const std::vector<int> &perm_in) {
    const int nd = (int)x.shape.size();
    std::vector<int> perm = perm_in;
    if (perm.empty()) for (int k = nd - 1; k >= 0; k--) perm.push_back(k);
    if ((int)perm.size() != nd) throw OpError(OpError::InvalidValue, "Permutation is invalid");
    std::vector<int64_t> xs((size_t)nd, 1);
    for (int d = nd - 2; d >= 0; d--) xs[(size_t)d] = xs[(size_t)d + 1] * x.shape[(size_t)d + 1];
    HostVal out;
    out.is_float = x.is_float;
    std::vector<int64_t> st((size_t)nd);
    for (int d = 0; d < nd; d++) {
        const int pd = perm[(size_t)d] < 0 ? perm[(size_t)d] + nd : perm[(size_t)d];
        if (pd < 0 || pd >= nd) throw OpError(OpError::InvalidValue, "Permutation is invalid");
        out.shape.push_back(x.shape[(size_t)pd]);
        st[(size_t)d] = xs[(size_t)pd];
    }
    if (out.is_float) out.f.resize((size_t)out.len()); else out.i.resize((size_t)out.len());
    for_each_broadcast(out.shape, {st}, [&](int64_t i, const std::vector<int64_t> &o) { if (out.is_float) out.f[(size_t)i] = x.f[(size_t)o[0]]; else out.i[(size_t)i] = x.i[(size_t)o[0]]; });
    return out;
}
inline HostVal gather(const HostVal &x, const HostVal &ids, int axis) { // src/ops/gather.rs:21-110
    const int nd = (int)x.shape.size();
    if (nd < 1) throw OpError(OpError::InvalidValue, "Input must have >= 1 dims");
    const int ax = resolve_axis(axis, nd);
    const int64_t outer = detail::prod(x.shape, 0, (size_t)ax), alen = x.shape[(size_t)ax], inner = detail::prod(x.shape, (size_t)ax + 1, x.shape.size());
    HostVal out;
    out.is_float = x.is_float;
    out.shape.assign(x.shape.begin(), x.shape.begin() + ax);
    out.shape.insert(out.shape.end(), ids.shape.begin(), ids.shape.end());
    out.shape.insert(out.shape.end(), x.shape.begin() + ax + 1, x.shape.end());
    const int64_t nid = ids.len();
    for (int64_t o = 0; o < outer; o++)
        for (int64_t j = 0; j < nid; j++) {
            int64_t id = ids.i[(size_t)j];
            if (id < 0) id += alen;
            if (id < 0 || id >= alen) throw OpError(OpError::InvalidValue, "Entry in indices is out of range");
            for (int64_t k = 0; k < inner; k++) {
                const size_t src = (size_t)((o * alen + id) * inner + k);
                if (out.is_float) out.f.push_back(x.f[src]); else out.i.push_back(x.i[src]);
            }
        }
    return out;
}
inline HostVal concat(const std::vector<const HostVal *> &v, int axis) { // src/ops/concat.rs
    const int nd = (int)v[0]->shape.size();
    const int ax = resolve_axis(axis, nd);
    HostVal out;
    out.is_float = v[0]->is_float;
    out.shape = v[0]->shape;
    out.shape[(size_t)ax] = 0;
    for (auto *x : v) {
        if ((int)x->shape.size() != nd) throw OpError(OpError::IncompatibleInputShapes, "Tensors must have the same number of dimensions");
        if (x->is_float != out.is_float) throw OpError(OpError::UnsupportedType, "");
        out.shape[(size_t)ax] += x->shape[(size_t)ax];
    }
    const int64_t outer = detail::prod(out.shape, 0, (size_t)ax), inner = detail::prod(out.shape, (size_t)ax + 1, out.shape.size());
    for (int64_t o = 0; o < outer; o++)
        for (auto *x : v) {
            const int64_t row = x->shape[(size_t)ax] * inner;
            for (int64_t k = 0; k < row; k++) { if (out.is_float) out.f.push_back(x->f[(size_t)(o * row + k)]); else out.i.push_back(x->i[(size_t)(o * row + k)]); }
        }
    return out;
}
inline HostVal slice(const HostVal &x, const std::vector<SliceRange> &r) {
    const int nd = (int)x.shape.size();
    HostVal out;
    out.is_float = x.is_float;
    std::vector<int64_t> st((size_t)nd);
    int64_t acc = 1, base = 0;
    for (int d = nd - 1; d >= 0; d--) { out.shape.insert(out.shape.begin(), r[(size_t)d].count); st[(size_t)d] = acc * r[(size_t)d].step; base += acc * r[(size_t)d].start; acc *= x.shape[(size_t)d]; }
    if (out.is_float) out.f.resize((size_t)out.len()); else out.i.resize((size_t)out.len());
    for_each_broadcast(out.shape, {st}, [&](int64_t i, const std::vector<int64_t> &o) { if (out.is_float) out.f[(size_t)i] = x.f[(size_t)(base + o[0])]; else out.i[(size_t)i] = x.i[(size_t)(base + o[0])]; });
    return out;
}
inline HostVal nonzero(const HostVal &x) { // src/ops/non_zero.rs: [rank, count] int32 indices of the non-zero elements, row-major order
    const int nd = (int)x.shape.size();
    std::vector<std::vector<int64_t>> cols;
    const int64_t n = x.len();
    std::vector<int64_t> idx((size_t)nd, 0);
    for (int64_t i = 0; i < n; i++) {
        const bool nz = x.is_float ? x.f[(size_t)i] != 0.f : x.i[(size_t)i] != 0;
        if (nz) cols.push_back(idx);
        for (int d = nd - 1; d >= 0; d--) { if (++idx[(size_t)d] < x.shape[(size_t)d]) break; idx[(size_t)d] = 0; }
    }
    HostVal out;
    out.shape = {(int64_t)std::max(nd, 1), (int64_t)cols.size()}; // (a scalar input gives [1, 0 or 1] in ONNX; the reference rejects scalars)
    out.i.resize((size_t)(out.shape[0] * out.shape[1]));
    for (int d = 0; d < nd; d++) for (size_t c = 0; c < cols.size(); c++) out.i[(size_t)d * cols.size() + c] = cols[c][(size_t)d];
    return out;
}
} // namespace hostops

// ======================================================================================================== executor
struct GraphError : std::runtime_error { using std::runtime_error::runtime_error; };

class Graph {
  public:
    struct Options {
        bool fuse = true;    // the fusion passes listed at the top of this file
        bool prepack = true; // stage constant conv weights once (Graph::prepack_weights)
        // Opt-in, per edge (a launch plan: profiles/plans/int8.json "qout"): ConvInteger nodes, by name, whose fused ConvIntegerToFloat step also
        // runs the DynamicQuantizeLinear of the one convolution reading its output (rten_hip_conv2d_int8_qout).  That launch needs the device to
        // itself (all its workgroups resident at once; rten_hip.h states the time-out contract), so it is never a default.
        std::set<std::string> qout;
        // The same edges in the RECOMPUTE form (plan key "qout2"; round 6): the producing convolution runs twice -- statistics only, then again with the
        // consumer's codes as its output -- so there is no grid-wide exchange and no residency requirement: replicas side by side ("lanes") may use it.  The
        // f32 tensor of the edge is never written or read back (8 B per element of HBM traffic saved for one more pass over small int8 operands).
        std::set<std::string> qout_recompute;
        // Opt-in, per layer (profiles/plans/int8.json "fused_dql"): pointwise ConvInteger nodes, by name, whose fused ConvIntegerToFloat step runs its
        // DynamicQuantizeLinear inside its own operand loader (rten_hip_conv2d_int8_dql) instead of reading the staged codes.
        std::set<std::string> fused_dql;
        // Opt-in, per layer (a launch plan: profiles/plans/f32_lanes.json "pairs"): f32 Conv steps, by name, that also run the ONE pointwise convolution reading
        // their output -- a bottleneck block's expand layer and the next block's reduce layer in one launch (rten_hip_conv2d_f32_pair; round 6): the
        // second layer takes its operand from LDS instead of reading the tensor back from HBM.  Same bits; a pair the kernel has no form for runs as two launches.
        std::set<std::string> pairs;
        // ... and, of those, the pairs that ALSO compute their first convolution's residual in the launch (plan key "pair_shortcuts"): the residual is the output of a
        // 64-channel pointwise Conv step nothing else reads -- a stage's shortcut layer -- which then has no launch and no tensor of its own
        // (rten_hip_conv2d_f32_pair_shortcut).
        std::set<std::string> pair_shortcuts;
        // A rank that RECEIVES the weight arena (coalesce_constants() + one broadcast from the rank that loaded the file for real): initializers of
        // 64 KB and more are allocated but not uploaded.  Everything derived from them on the device (prepacked weights) is computed on whatever the
        // buffers hold and overwritten by the broadcast, which covers every constant buffer of the graph.
        bool skip_large_uploads = false;
    };
    struct Timing { std::string node, op; double ms; };

    Graph(Context &ctx, const onnx::Model &m, Options opt) : ctx_(ctx), opt_(opt) { compile(m); }
    Graph(Context &ctx, const onnx::Model &m) : Graph(ctx, m, Options()) {}
    // A second plan over the SAME model that shares `donor`'s device constants and prepacked weights instead of uploading its own (the sub-batch
    // chains of one model: four copies of a 100 MB weight set would also compete for the same L2 / MALL lines).  `donor` must outlive this graph
    // and live on the same device; both must have been built with the same options.
    Graph(Context &ctx, const onnx::Model &m, Options opt, const Graph &donor) : ctx_(ctx), opt_(opt), donor_(&donor) { compile(m); }
    // the node list after the load-time canonicalisation of exporter idioms (Constant nodes, GeluFusion, LayerNormalizationFusion): needs no device
    static onnx::Model canonical_form(const onnx::Model &m) { return canonicalize(m); }

    const std::vector<onnx::ValueInfo> &inputs() const { return inputs_; }
    const std::vector<onnx::ValueInfo> &outputs() const { return outputs_; }
    size_t num_steps() const { return steps_.size(); }
    size_t num_fused_away() const { return fused_away_; }
    size_t num_staged_quantizers() const { return staged_dql_; }
    size_t num_qout_edges() const { return qout_edges_; } // quantizers merged into their producer's launch (Options::qout)
    size_t num_stats_blocks() const { return stats_blocks_; }
    size_t num_conv_pairs() const { return conv_pairs_; } // convolution pairs that run as one launch (Options::pairs)
    size_t num_dql_loader_steps() const { return dql_loader_steps_; } // convolutions that quantize in their own loader (Options::fused_dql)

    // Moves every device constant of this graph -- uploaded initializers, constants derived at load (merged QKV weights), prepacked conv / MatMul
    // weights -- into ONE allocation, in a deterministic order (value id, then prepack order): the weight arena a batch-sharded deployment
    // broadcasts once from the rank that read the model file (RCCL over xGMI).  Two processes that load the same model with the same
    // options get the same layout.  Must run before a sharing graph (the `donor` constructor) is built on this one and before any capture.
    std::pair<void *, size_t> coalesce_constants() {
        if (donor_) throw GraphError("coalesce_constants: a graph that shares a donor's constants has none of its own");
        if (arena_) return {arena_->ptr(), (size_t)arena_->len()};
        auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
        size_t total = 0;
        for (auto &kv : consts_) total += pad(kv.second.bytes());
        for (auto &pk : packed_) total += pad(pk->bytes());
        arena_.reset(new Tensor(ctx_, {(int64_t)std::max<size_t>(total, 256)}, DType::U8));
        size_t off = 0;
        auto move_in = [&](Tensor &t) {
            const size_t b = t.bytes();
            if (b) ctx_.check(rten_hip_memcpy_d2d(ctx_.raw(), (char *)arena_->ptr() + off, t.ptr(), b));
            Tensor v = Tensor::view_at(*arena_, off, t.shape(), t.dtype());
            v.set_host(t.host_ptr()); // a small constant's host mirror (shape arithmetic reads it) moves with it
            ctx_.sync();       // the copy has read the old buffer before it goes back to the allocator
            t = std::move(v);  // same Tensor object (steps hold pointers to it), new storage
            off += pad(b);
        };
        for (auto &kv : consts_) move_in(kv.second);
        for (auto &pk : packed_) move_in(*pk);
        ctx_.trim_pool();
        return {arena_->ptr(), (size_t)arena_->len()};
    }

    // Empty, or the first step whose result for one row of dim 0 depends on the other rows: such a graph cannot be split into sub-batch chains
    // (DynamicQuantizeLinear takes its min / max over the whole tensor; a reduction / normalisation over axis 0 mixes rows).
    std::string batch_coupled_step() const {
        for (auto &st : steps_) if (st.batch_coupled) return st.kind_name + " \"" + st.name + "\"";
        return "";
    }
    // ... and what only a RUN can tell (ranks are run-time facts here): a Softmax / LayerNormalization / ReduceSum / ReduceMean whose NEGATIVE axis resolved to
    // dim 0, or a device Transpose that moved dim 0.  Empty, or the first such step of the last run (rten_hip_model_prepare checks it after its probe run).
    const std::string &runtime_batch_coupled_step() const { return runtime_coupled_; }
    std::vector<std::string> step_names() const {
        std::vector<std::string> v;
        for (auto &s : steps_) v.push_back(s.kind_name + ":" + s.name);
        return v;
    }

    // Runs the plan.  `feeds` are device tensors named like the graph inputs; the returned tensors are the graph outputs
    // in declaration order, still on the device.  With `timings`, every step is followed by a sync and timed on the host
    // (RTEN_TIMING's per-operator table, src/timing.rs) -- a profiling aid, it serialises the stream.
    using Feeds = std::vector<std::pair<std::string, const Tensor *>>;

    // Load-time plan selection: one pass over the graph in which every f32 convolution step times its candidate launch
    // plans on its real operands (tile variant x exact split-K plan x tile order, the candidate set of
    // rten_amd/workloads/resnet50.py::candidate_plans) and keeps the fastest.  Returns the number of tuned steps.
    size_t autotune(const Feeds &feeds, int reps = 3) {
        tune_reps_ = reps;
        tuned_ = 0;
        run(feeds);
        ctx_.sync();
        tune_reps_ = 0;
        return tuned_;
    }

    // A committed launch plan instead of timing at load ({conv step name: GemmPlan}, profiles/plans/*.json: tuned once on an MI355X so that
    // every process -- and every rank of a sharded deployment -- launches the same kernels).  Steps the table does not name keep the
    // backend's automatic plan.  Returns the number of steps that took an entry.
    size_t apply_plan(const std::map<std::string, GemmPlan> &table) {
        size_t n = 0;
        for (auto &st : steps_) {
            if (!st.conv && !st.gemm_plan && !st.i8) continue;
            auto it = table.find(st.name);
            if (it == table.end()) { if (st.i8) st.i8->tile = -1; continue; }
            if (st.i8) { // an int8 convolution step: the entry's first number is its workgroup tile (round 6: "<step>": [tile, 0, 1, 0])
                if (it->second.variant < 0 || it->second.variant > 3) continue;
                st.i8->tile = it->second.variant;
            }
            else if (st.conv) st.conv->plan = it->second;
            else *st.gemm_plan = it->second;
            n++;
        }
        return n;
    }
    // Plan entries keyed by the PRODUCT'S SHAPE -- "gemm:MxKxN", M = the rows of A over all its leading dims -- for MatMul-family steps no name entry covers (round
    // 6): a plan chosen on one writer's graph (profiles/plans/bert_base_b32_s128_lanes.json, its "shapes" table) then applies to another exporter's file of
    // the same model, whose steps carry other names.  Shapes are run-time facts here, so the lookup happens in run(), once per step (the entry stays).
    void set_shape_plans(std::map<std::string, GemmPlan> t) {
        for (size_t i : shape_planned_steps_) *steps_[i].gemm_plan = GemmPlan{}; // (a re-plan: entries a previous table left on steps go with that table)
        shape_planned_steps_.clear();
        shape_plans_ = std::move(t);
        shape_planned_ = 0;
    }
    size_t num_shape_planned() const { return shape_planned_; }
    std::map<std::string, GemmPlan> plans() const { // what autotune() / apply_plan() left on the convolution steps
        std::map<std::string, GemmPlan> t;
        for (auto &st : steps_) {
            if (st.conv && st.conv->plan.set) t[st.name] = st.conv->plan;
            if (st.gemm_plan && st.gemm_plan->set) t[st.name] = *st.gemm_plan;
            if (st.i8 && st.i8->tile >= 0) t[st.name] = GemmPlan{true, st.i8->tile, 0, 1, 0};
        }
        return t;
    }
    bool captured() const { return graph_ != 0; }
    const std::vector<Tensor> &captured_outputs() const { return captured_outputs_; }

    // hipGraph capture of one run (launch-bound at small batch: ~60 operators of 5-50 us).  The feeds and the returned
    // outputs keep their device addresses: write new inputs into the same feed tensors, call replay(), read the outputs.
    // While a capture is alive the context's buffer pool must not serve anyone else (the graph's intermediates live there).
    // `tail` (optional) runs at the end of the captured region, on the capturing stream, with the run's outputs: work recorded there -- e.g. the
    // copy of a sub-batch's rows into a resident full-batch tensor -- replays with the graph.
    const std::vector<Tensor> &capture(const Feeds &feeds, const std::function<void(const std::vector<Tensor> &)> &tail = nullptr) {
        run(feeds); // warm: every buffer size is in the pool, scratch is grown, code objects are loaded
        ctx_.sync();
        if (graph_) { // re-capture: the previous executable graph is released first
            ctx_.check(rten_hip_graph_destroy(ctx_.raw(), graph_));
            graph_ = 0;
        }
        ctx_.check(rten_hip_graph_begin(ctx_.raw()));
        try {
            captured_outputs_ = run(feeds);
            if (tail) tail(captured_outputs_);
        } catch (...) { // leave the stream (and the context's lock) out of capture mode before the error propagates
            uint64_t dead = 0;
            if (rten_hip_graph_end(ctx_.raw(), &dead) == RTEN_HIP_OK && dead) rten_hip_graph_destroy(ctx_.raw(), dead);
            throw;
        }
        ctx_.check(rten_hip_graph_end(ctx_.raw(), &graph_));
        return captured_outputs_;
    }
    void replay() {
        if (!graph_) throw GraphError("replay: no captured graph");
        ctx_.check(rten_hip_graph_launch(ctx_.raw(), graph_));
    }
    ~Graph() { if (graph_) rten_hip_graph_destroy(ctx_.raw(), graph_); }
    Graph(const Graph &) = delete;
    Graph &operator=(const Graph &) = delete;

    std::vector<Tensor> run(const Feeds &feeds, std::vector<Timing> *timings = nullptr) {
        std::vector<const Tensor *> val(names_.size(), nullptr);
        std::vector<std::unique_ptr<Tensor>> owned(names_.size());
        std::vector<int> pending(uses_);
        runtime_coupled_.clear();
        for (auto &kv : consts_) val[(size_t)kv.first] = &kv.second;
        for (auto &fd : feeds) {
            auto it = ids_.find(fd.first);
            if (it == ids_.end()) throw GraphError("run: no graph input named " + fd.first);
            val[(size_t)it->second] = fd.second;
        }
        for (auto &in : inputs_) if (!val[(size_t)ids_.at(in.name)]) throw GraphError("run: missing input " + in.name);
        if (stats_blocks_) ctx_.check(rten_hip_minmax_stats_reset(ctx_.raw(), stats_arena_->ptr(), (int32_t)stats_blocks_)); // one launch for every block
        for (auto &st : steps_) {
            InputList in;
            for (int id : st.in) {
                if (id >= 0 && !val[(size_t)id]) throw GraphError("run: value " + names_[(size_t)id] + " needed by " + st.name + " was never produced");
                in.push_back(id < 0 ? nullptr : val[(size_t)id]);
            }
            const auto t0 = std::chrono::steady_clock::now();
            OutputList out;
            if (st.gemm_plan && !st.gemm_plan->set && !shape_plans_.empty() && tune_reps_ == 0 && in.size() > 1 && in[0] && in[1] && in[0]->ndim() >= 2 && in[1]->ndim() == 2) {
                const Tensor &a = *in[0], &b = *in[1];
                const int64_t k = a.size(a.ndim() - 1), mrows = a.len() / std::max<int64_t>(k, 1);
                const int64_t ncols = b.size(0) == k ? b.size(1) : (b.size(1) == k ? b.size(0) : -1); // [K, N], or [N, K] (Gemm with transB)
                auto it = ncols < 0 ? shape_plans_.end() : shape_plans_.find("gemm:" + std::to_string(mrows) + "x" + std::to_string(k) + "x" + std::to_string(ncols));
                if (it != shape_plans_.end()) { *st.gemm_plan = it->second; shape_planned_++; shape_planned_steps_.push_back((size_t)(&st - steps_.data())); }
            }
            if (tune_reps_ > 0 && (st.conv || st.gemm_plan)) tune_step(st, in);
            try {
                if (st.i8 && st.i8->tile >= 0) { // the plan's workgroup tile around this step's launches (the caller's own setting comes back)
                    struct TileScope {
                        Context &c; int32_t prev = -1; bool on = false;
                        TileScope(Context &c_, int tile) : c(c_) { on = rten_hip_set_int8_tile(c.raw(), tile, &prev) == RTEN_HIP_OK; }
                        ~TileScope() { if (on) rten_hip_set_int8_tile(c.raw(), prev, nullptr); }
                    } scope(ctx_, st.i8->tile);
                    out = st.run(ctx_, in);
                } else
                out = st.run(ctx_, in);
            } catch (const OpError &e) { // name the node, like the reference's RunError::OperatorError { name, error }
                throw OpError(e.kind, "operator " + st.kind_name + " \"" + st.name + "\": " + e.msg);
            }
            if (timings) {
                ctx_.sync();
                timings->push_back({st.name, st.kind_name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()});
            }
            if (out.size() < st.out.size()) throw GraphError("run: " + st.name + " produced fewer outputs than the graph names");
            for (size_t i = 0; i < st.out.size(); i++) {
                if (st.out[i] < 0) continue;
                owned[(size_t)st.out[i]].reset(new Tensor(std::move(out[i])));
                val[(size_t)st.out[i]] = owned[(size_t)st.out[i]].get();
            }
            // a view keeps its base alive: the base's count was raised by one at compile time for exactly this reason
            for (int id : st.release_after) {
                if (--pending[(size_t)id] == 0) { owned[(size_t)id].reset(); }
            }
        }
        std::vector<Tensor> result;
        std::set<int> moved;
        for (auto &o : outputs_) {
            const int id = ids_.at(o.name);
            if (!val[(size_t)id]) throw GraphError("run: output " + o.name + " is not produced by an operator");
            // a graph output that is a graph input or an initializer (not owned by this run), one that is listed twice, and a
            // view (its storage belongs to a value that dies with this run) are handed out as copies
            if (!owned[(size_t)id] || moved.count(id) || view_values_.count(id)) {
                const Tensor &v = moved.count(id) ? result[(size_t)std::distance(outputs_.begin(), std::find_if(outputs_.begin(), outputs_.end(), [&](const onnx::ValueInfo &x) { return x.name == o.name; }))]
                                                  : *val[(size_t)id];
                Tensor copy(ctx_, v.shape(), v.dtype());
                if (v.bytes()) ctx_.check(rten_hip_memcpy_d2d(ctx_.raw(), copy.ptr(), v.ptr(), v.bytes()));
                result.push_back(std::move(copy));
            } else {
                result.push_back(std::move(*owned[(size_t)id]));
                moved.insert(id);
            }
        }
        return result;
    }

  private:
    struct I8Conv {
        std::shared_ptr<ConvInteger> op;
        ConvInteger::Staging sg;
        bool to_float = false;
        bool relu = false;
        int tile = -1;         // a launch plan's per-layer workgroup tile for this step's int8 kernel (rten_hip_set_int8_tile; -1 = the backend's rule)
        bool qout_off = false; // the one-launch form was refused once (grid not resident at once): the two-launch sequence from then on
        bool qout_producer = false; // this step also runs its consumer's quantizer (Options::qout): it keeps reading staged codes itself
    };
    struct Step {
        std::string name, kind_name;
        std::vector<int> in, out, release_after;
        std::function<OutputList(Context &, const InputList &)> run;
        bool view = false;
        std::shared_ptr<Conv> conv; // f32 convolution steps: the launch plan is tunable
        std::shared_ptr<GemmPlan> gemm_plan; // MatMul / FusedMatMul / Gemm steps: the same launch plan, applied around the step's GEMM
        std::shared_ptr<I8Conv> i8; // int8 convolution steps: staged-pipeline options are decided after all steps exist
        std::shared_ptr<DynamicQuantizeLinearStaged> dql_staged;
        std::shared_ptr<MaxPool> maxpool; // a max-pool whose output is quantized next can accumulate the quantizer's statistics
        bool removed = false;
        bool batch_coupled = false; // the result for one dim-0 row depends on other rows (no sub-batch chains for such a graph)
        size_t pos = 0; // index of the LAST graph node folded into this step: the step runs where that node stood
    };

    Context &ctx_;
    Options opt_;
    std::vector<std::string> names_;
    std::map<std::string, int> ids_;
    std::map<int, Tensor> consts_;
    std::vector<std::unique_ptr<Tensor>> packed_; // prepacked conv weights
    std::map<std::string, const Tensor *> packed_of_; // step name -> its prepacked operand (what a sharing graph looks up)
    const Graph *donor_ = nullptr;
    // a constant of this graph: uploaded here, or a non-owning view of the donor's
    void add_const(int id, const std::function<Tensor()> &make) {
        if (donor_) { const Tensor &d = donor_->consts_.at(id); Tensor v = Tensor::view_of(d, d.shape()); v.set_host(d.host_ptr()); consts_.emplace(id, std::move(v)); }
        else consts_.emplace(id, make());
    }
    // the prepacked operand of step `name`: packed here, or the donor's
    const Tensor *packed_for(const std::string &name, const std::function<Tensor()> &pack) {
        if (donor_) { auto it = donor_->packed_of_.find(name); return it == donor_->packed_of_.end() ? nullptr : it->second; }
        Tensor pk = pack();
        if (!pk.len()) return nullptr;
        packed_.emplace_back(new Tensor(std::move(pk)));
        packed_of_[name] = packed_.back().get();
        return packed_.back().get();
    }
    std::vector<Step> steps_;
    std::string runtime_coupled_;
    void note_axis(const std::string &kind, const std::string &name, int axis, const Tensor &x) {
        if (axis < 0 && x.ndim() > 0 && axis + x.ndim() == 0 && runtime_coupled_.empty()) runtime_coupled_ = kind + " \"" + name + "\" (axis " + std::to_string(axis) + " of a rank-" + std::to_string(x.ndim()) + " input is dim 0)";
    }
    std::vector<int> uses_;
    std::vector<onnx::ValueInfo> inputs_, outputs_;
    size_t fused_away_ = 0;
    size_t conv_pairs_ = 0;
    std::set<int> view_values_;
    int tune_reps_ = 0;
    size_t tuned_ = 0;
    std::map<std::string, GemmPlan> shape_plans_;
    size_t shape_planned_ = 0;
    std::vector<size_t> shape_planned_steps_;
    size_t stats_blocks_ = 0, staged_dql_ = 0, qout_edges_ = 0, dql_loader_steps_ = 0;
    std::unique_ptr<Tensor> stats_arena_, sync_arena_, arena_;

    // what the ConvIntegerToFloat step's own lambda decides per run: does (scale, bias, residual) take the fused epilogue?
    static bool i8_fused_form(const I8Conv &state, const InputList &in, bool &per_channel) {
        const Tensor *scale = in[4], *bias = in[5], *residual = in[6];
        per_channel = false;
        if (!scale) return false;
        const Tensor &x = require(in, 0), &w = require(in, 1);
        bool fused = scale->dtype() == DType::F32 && x.ndim() == 4 && w.ndim() == 4;
        const int64_t o = fused ? w.size(0) : 0;
        if (fused && scale->len() != 1) {
            const auto &sh = scale->shape();
            per_channel = scale->len() == o && ((sh.size() == 4 && sh[0] == 1 && sh[1] == o) || (sh.size() == 3 && sh[0] == o));
            fused = per_channel;
        }
        if (fused && bias) fused = bias->dtype() == DType::F32 && bias->len() == o && bias->ndim() == 4 && bias->size(1) == o;
        if (fused && residual) {
            const rten_hip_conv2d_desc d = state.op->conv.geometry(x.shape(), w.shape());
            fused = residual->dtype() == DType::F32 && residual->shape() == std::vector<int64_t>{d.n, d.o, d.out_h, d.out_w};
        }
        return fused;
    }

    // The staged int8 pipeline at graph level.  A DynamicQuantizeLinear whose codes feed only int8
    // convolutions of one padding geometry writes them straight into the kernel's staged layout; if its input is the f32
    // output of a fused ConvIntegerToFloat step or of a MaxPool, that launch accumulates the min/max the quantizer needs
    // (one statistics block per such tensor, all reset by one launch at the start of a run).
    void plan_int8_staging() {
        std::map<int, size_t> producer;
        std::map<int, std::vector<size_t>> consumers;
        for (size_t i = 0; i < steps_.size(); i++) {
            for (int id : steps_[i].out) if (id >= 0) producer[id] = i;
            for (int id : steps_[i].in) if (id >= 0) consumers[id].push_back(i);
        }
        struct Pending { size_t dql; size_t producer; };
        std::vector<Pending> want_stats;
        for (size_t i = 0; i < steps_.size(); i++) {
            Step &dq = steps_[i];
            if (dq.kind_name != "DynamicQuantizeLinear" || dq.out.size() < 3 || dq.out[0] < 0 || dq.i8) continue;
            bool graph_out = false;
            for (auto &o : outputs_) if (ids_.at(o.name) == dq.out[0]) graph_out = true;
            const auto &users = consumers[dq.out[0]];
            if (graph_out || users.empty()) continue;
            bool ok = true;
            const Step *first = nullptr;
            for (size_t u : users) {
                const Step &cs = steps_[u];
                if (!cs.i8 || cs.in[0] != dq.out[0] || cs.in[2] != dq.out[2] || !consts_.count(cs.in[1]) || cs.i8->op->conv.padding.same ||
                    cs.i8->op->conv.groups != 1) { ok = false; break; }
                if (std::count(cs.in.begin(), cs.in.end(), dq.out[0]) != 1) { ok = false; break; }
                if (!first) first = &cs;
                else if (cs.i8->op->conv.padding.fixed != first->i8->op->conv.padding.fixed || cs.i8->op->pad_mode != first->i8->op->pad_mode) { ok = false; break; }
            }
            if (!ok || !first) continue;
            auto op = std::make_shared<DynamicQuantizeLinearStaged>();
            op->consumer = *first->i8->op;
            op->kernel = consts_.at(first->in[1]).shape();
            for (size_t u : users) steps_[u].i8->sg.x_staged = true;
            dq.kind_name = "DynamicQuantizeLinear(staged)";
            dq.run = [op](Context &c, const InputList &in) { return op->run(c, in); };
            dq.dql_staged = op;
            staged_dql_++;
            // fold the following Mul(y_scale, constant scalar) nodes -- one per convolution that reads the codes -- into the quantizer (same f32
            // multiply each): their products become outputs 4, 5, ...
            for (size_t u : consumers[dq.out[1]]) {
                if (op->mul_by.size() >= DynamicQuantizeLinearStaged::kMaxProducts) break;
                Step &mu = steps_[u];
                if (mu.kind_name != "Mul" || mu.in.size() != 2 || mu.removed) continue;
                const int other = mu.in[0] == dq.out[1] ? mu.in[1] : mu.in[0];
                if (other == dq.out[1] || !consts_.count(other) || consts_.at(other).len() != 1 || consts_.at(other).dtype() != DType::F32) continue;
                bool is_out = false;
                for (auto &o : outputs_) if (ids_.at(o.name) == mu.out[0]) is_out = true;
                if (is_out) continue;
                op->mul_by.push_back(&consts_.at(other));
                dq.out.push_back(mu.out[0]);
                mu.removed = true;
                fused_away_++;
            }
            auto p = producer.find(dq.in[0]);
            if (p != producer.end() && steps_[p->second].out[0] == dq.in[0] &&
                ((steps_[p->second].i8 && steps_[p->second].i8->to_float) || steps_[p->second].maxpool))
                want_stats.push_back({i, p->second});
        }
        // (indices into steps_ stay valid until here; drop the absorbed Mul steps last)
        struct Eraser { std::vector<Step> &v; ~Eraser() { v.erase(std::remove_if(v.begin(), v.end(), [](const Step &s) { return s.removed; }), v.end()); } } eraser{steps_};
        if (want_stats.empty()) return;
        const size_t sb = rten_hip_minmax_stats_bytes();
        std::map<size_t, size_t> block_of; // producer step -> block
        for (auto &w : want_stats) if (!block_of.count(w.producer)) { const size_t b = block_of.size(); block_of[w.producer] = b; }
        stats_blocks_ = block_of.size();
        stats_arena_.reset(new Tensor(ctx_, {(int64_t)(sb * stats_blocks_)}, DType::U8));
        for (auto &w : want_stats) {
            void *blk = (char *)stats_arena_->ptr() + sb * block_of[w.producer];
            if (steps_[w.producer].maxpool) steps_[w.producer].maxpool->stats_out = blk;
            else steps_[w.producer].i8->sg.stats_out = blk;
            steps_[w.dql].dql_staged->stats_in = blk;
            steps_[w.dql].kind_name = "DynamicQuantizeLinear(staged, producer statistics)";
        }
        if (opt_.qout.empty() && opt_.qout_recompute.empty()) return;
        // Opt-in edges: the quantizer moves INTO its producer's launch.  The merged step produces the conv's f32 tensor (only if somebody else reads
        // it: a residual Add) and the quantizer's outputs; the quantizer's step disappears.  A launch that is refused at run time (its grid is not
        // resident at once) runs the two operators instead, from then on.
        std::vector<Pending> edges;
        for (auto &w : want_stats) {
            Step &P = steps_[w.producer], &D = steps_[w.dql];
            if (!P.i8 || !P.i8->to_float || !(opt_.qout.count(P.name) || opt_.qout_recompute.count(P.name)) || !P.i8->sg.x_staged || !P.i8->sg.packed_weight) continue;
            size_t quantizers = 0;
            for (size_t u : consumers[D.in[0]]) if (steps_[u].dql_staged) quantizers++;
            if (quantizers != 1) continue;
            edges.push_back(w);
        }
        if (edges.empty()) return;
        const size_t gb = rten_hip_grid_sync_bytes();
        sync_arena_.reset(new Tensor(ctx_, {(int64_t)(gb * edges.size())}, DType::U8));
        ctx_.check(rten_hip_grid_sync_reset(ctx_.raw(), sync_arena_->ptr(), (int32_t)edges.size()));
        for (size_t e = 0; e < edges.size(); e++) {
            Step &P = steps_[edges[e].producer], &D = steps_[edges[e].dql];
            bool keep = consumers[D.in[0]].size() > 1;
            for (auto &o : outputs_) if (ids_.at(o.name) == D.in[0]) keep = true;
            const bool recompute = !opt_.qout.count(P.name); // (listed under "qout2" only: no exchange block is used)
            void *sync = recompute ? nullptr : (char *)sync_arena_->ptr() + gb * e;
            auto state = P.i8;
            state->qout_producer = true;
            auto dql = D.dql_staged;
            auto two_launches = P.run;
            P.run = [state, dql, two_launches, sync, keep, recompute](Context &c, const InputList &in) {
                OutputList out;
                bool per_channel = false;
                if (!state->qout_off && i8_fused_form(*state, in, per_channel)) {
                    if (dql->run_in_producer(c, *state->op, InputList(in.begin(), in.begin() + 4), *in[4], in[5], in[6], state->relu, state->sg, per_channel, sync, keep, out, recompute))
                        return out;
                    state->qout_off = true;
                }
                out = two_launches(c, in);
                OutputList q = dql->run(c, {&out[0]});
                for (Tensor &t : q) out.push_back(std::move(t));
                return out;
            };
            P.out.insert(P.out.end(), D.out.begin(), D.out.end());
            P.kind_name += recompute ? " + DynamicQuantizeLinear(staged) by recomputation (statistics pass + code pass)" : " + DynamicQuantizeLinear(staged) in one launch";
            D.removed = true;
            fused_away_++;
            qout_edges_++;
        }
    }
    // Opt-in per layer (Options::pairs): the f32 Conv step `A` (with whatever Add / Relu it absorbed) and the one f32 Conv step `B` whose input is A's output
    // become ONE step with two outputs, at A's place.  B must take A's output as its data input, constant weights, no residual; the kernel's forms (unit-stride
    // pointwise, K1 = 64, M1 <= 256, M2 in {64, 128}: rten_hip_conv2d_f32_pair_supported) are run-time facts, checked per run -- a pair outside them runs A's and
    // B's own steps one after the other, as the graph spells them.  A's output stays a value of its own: other steps (the next block's residual Add, a downsample
    // layer) still read it.
    void plan_conv_pairs() {
        if (opt_.pairs.empty()) return;
        auto packed_lookup = [&](const std::string &name) -> const Tensor * {
            const auto &tbl = donor_ ? donor_->packed_of_ : packed_of_;
            auto it = tbl.find(name);
            return it == tbl.end() ? nullptr : it->second;
        };
        for (size_t i = 0; i < steps_.size(); i++) {
            Step &A = steps_[i];
            if (!A.conv || A.removed || !opt_.pairs.count(A.name) || A.out.empty() || A.out[0] < 0 || A.in.size() < 4) continue;
            // B: a Conv step reading A's output (constant weights, no residual).  A stage's last block has two such readers -- the next stage's reduce layer and its
            // strided shortcut layer: the one whose attributes fit the kernel (1x1, unit stride, no padding, one group) is taken, whichever comes first in the graph
            auto unit_pointwise = [&](const Step &c) {
                const Tensor &w = consts_.at(c.in[1]);
                const Conv &op = *c.conv;
                return w.ndim() == 4 && w.size(2) == 1 && w.size(3) == 1 && op.groups == 1 && op.strides == std::vector<int>{1, 1} &&
                       (op.padding.same || op.padding.fixed == std::vector<int>{0, 0, 0, 0});
            };
            Step *Bp = nullptr;
            for (size_t k = i + 1; k < steps_.size(); k++) {
                Step &c = steps_[k];
                if (c.conv && !c.removed && c.in.size() >= 4 && c.in[0] == A.out[0] && c.in[3] < 0 && c.in[1] >= 0 && consts_.count(c.in[1]) && (c.in[2] < 0 || consts_.count(c.in[2]))) {
                    if (!Bp) Bp = &c;
                    if (unit_pointwise(c)) { Bp = &c; break; }
                }
            }
            if (!Bp) continue;
            Step &B = *Bp;
            const Tensor *pa = packed_lookup(A.name), *pb = packed_lookup(B.name);
            if (!pa || !pb) continue;
            const std::shared_ptr<Conv> opa = A.conv, opb = B.conv;
            const auto run_a = A.run, run_b = B.run;
            // the shortcut form: A's residual is the output of a Conv step D (constant weights, no residual, no Relu) that only A reads
            Step *Dp = nullptr;
            const Tensor *pd = nullptr;
            if (opt_.pair_shortcuts.count(A.name) && A.in[3] >= 0) {
                size_t readers = 0;
                for (auto &o : outputs_) if (ids_.at(o.name) == A.in[3]) readers += 2;
                for (auto &st : steps_) if (!st.removed) for (int id : st.in) if (id == A.in[3]) readers++;
                for (size_t k = 0; k < i && readers == 1; k++) {
                    Step &c = steps_[k];
                    if (c.conv && !c.removed && !c.out.empty() && c.out[0] == A.in[3] && c.in.size() >= 4 && c.in[3] < 0 && !c.conv->fuse_relu && c.in[1] >= 0 && consts_.count(c.in[1]) &&
                        (c.in[2] < 0 || consts_.count(c.in[2])) && (pd = packed_lookup(c.name)) != nullptr) { Dp = &c; break; }
                }
            }
            const std::shared_ptr<Conv> opd = Dp ? Dp->conv : nullptr;
            const auto run_d = Dp ? Dp->run : run_a;
            if (Dp) A.in[3] = Dp->in[0]; // the step now reads the shortcut's INPUT where it read the shortcut's output
            A.in.push_back(B.in[1]);
            A.in.push_back(B.in[2]);
            if (Dp) { A.in.push_back(Dp->in[1]); A.in.push_back(Dp->in[2]); }
            A.out.push_back(B.out[0]);
            A.kind_name += ">" + B.kind_name;
            A.conv.reset(); // (not a tunable step any more: the one-launch form has no launch plan)
            A.run = [opa, opb, opd, pa, pb, pd, run_a, run_b, run_d](Context &c, const InputList &in) -> OutputList {
                const Tensor &x = require(in, 0), &w1 = require(in, 1), &w2 = require(in, 4);
                const Tensor *b1 = get(in, 2), *res = get(in, 3), *b2 = get(in, 5);
                if (opd) { // shortcut form: in[3] is the shortcut's input, in[6] / in[7] its weights / bias
                    const Tensor &xd = require(in, 3), &wd = require(in, 6);
                    const Tensor *bd = get(in, 7);
                    bool ok = x.dtype() == DType::F32 && xd.dtype() == DType::F32 && x.ndim() == 4 && xd.ndim() == 4 && w1.ndim() == 4 && w2.ndim() == 4 && wd.ndim() == 4 &&
                              opa->groups == 1 && opb->groups == 1 && opd->groups == 1;
                    rten_hip_conv2d_desc d1{}, d2{}, dd{};
                    if (ok) {
                        d1 = opa->geometry(x.shape(), w1.shape());
                        d2 = opb->geometry({d1.n, d1.o, d1.out_h, d1.out_w}, w2.shape());
                        dd = opd->geometry(xd.shape(), wd.shape());
                        ok = rten_hip_conv2d_f32_pair_shortcut_supported(&d1, &dd, &d2) == 1 && (!b1 || (b1->dtype() == DType::F32 && b1->len() == d1.o)) &&
                             (!b2 || (b2->dtype() == DType::F32 && b2->len() == d2.o)) && (!bd || (bd->dtype() == DType::F32 && bd->len() == dd.o));
                    }
                    if (!ok) { // the three steps as they were
                        OutputList r = run_d(c, {in[3], in[6], in[7], nullptr});
                        OutputList y1 = run_a(c, {in[0], in[1], in[2], &r[0]});
                        OutputList y2 = run_b(c, {&y1[0], in[4], in[5], nullptr});
                        y1.push_back(std::move(y2[0]));
                        return y1;
                    }
                    OutputList out;
                    out.emplace_back(c, std::vector<int64_t>{d1.n, d1.o, d1.out_h, d1.out_w}, DType::F32);
                    out.emplace_back(c, std::vector<int64_t>{d2.n, d2.o, d2.out_h, d2.out_w}, DType::F32);
                    c.check(rten_hip_conv2d_f32_pair_shortcut(c.raw(), &d1, (const float *)x.ptr(), (const float *)pa->ptr(), (const float *)vp(b1), &dd, (const float *)xd.ptr(),
                                                              (const float *)pd->ptr(), (const float *)vp(bd), opa->fuse_relu ? RTEN_HIP_CONV_RELU : 0u, (float *)out[0].ptr(), &d2,
                                                              (const float *)pb->ptr(), (const float *)vp(b2), opb->fuse_relu ? RTEN_HIP_CONV_RELU : 0u, (float *)out[1].ptr()));
                    return out;
                }
                bool ok = x.dtype() == DType::F32 && x.ndim() == 4 && w1.ndim() == 4 && w2.ndim() == 4 && w1.dtype() == DType::F32 && w2.dtype() == DType::F32 && opa->groups == 1 && opb->groups == 1;
                rten_hip_conv2d_desc d1{}, d2{};
                if (ok) {
                    d1 = opa->geometry(x.shape(), w1.shape());
                    d2 = opb->geometry({d1.n, d1.o, d1.out_h, d1.out_w}, w2.shape());
                    ok = rten_hip_conv2d_f32_pair_supported(&d1, &d2) == 1 && (!res || (res->dtype() == DType::F32 && res->shape() == std::vector<int64_t>{d1.n, d1.o, d1.out_h, d1.out_w})) &&
                         (!b1 || (b1->dtype() == DType::F32 && b1->len() == d1.o)) && (!b2 || (b2->dtype() == DType::F32 && b2->len() == d2.o));
                }
                if (!ok) { // the two steps as they were
                    OutputList y1 = run_a(c, {in[0], in[1], in[2], in[3]});
                    OutputList y2 = run_b(c, {&y1[0], in[4], in[5], nullptr});
                    y1.push_back(std::move(y2[0]));
                    return y1;
                }
                OutputList out;
                out.emplace_back(c, std::vector<int64_t>{d1.n, d1.o, d1.out_h, d1.out_w}, DType::F32);
                out.emplace_back(c, std::vector<int64_t>{d2.n, d2.o, d2.out_h, d2.out_w}, DType::F32);
                const uint32_t f1 = (opa->fuse_relu ? RTEN_HIP_CONV_RELU : 0u) | (res ? RTEN_HIP_CONV_RESIDUAL : 0u), f2 = opb->fuse_relu ? RTEN_HIP_CONV_RELU : 0u;
                c.check(rten_hip_conv2d_f32_pair(c.raw(), &d1, (const float *)x.ptr(), (const float *)pa->ptr(), (const float *)vp(b1), (const float *)vp(res), f1, (float *)out[0].ptr(),
                                                 &d2, (const float *)pb->ptr(), (const float *)vp(b2), f2, (float *)out[1].ptr()));
                return out;
            };
            B.removed = true;
            fused_away_++;
            conv_pairs_++;
            if (Dp) { Dp->removed = true; fused_away_++; conv_pairs_++; A.kind_name = Dp->kind_name + "+" + A.kind_name; }
        }
        steps_.erase(std::remove_if(steps_.begin(), steps_.end(), [](const Step &st) { return st.removed; }), steps_.end());
    }
    // Opt-in per layer (Options::fused_dql, the runner's `fused_layers`): a pointwise ConvIntegerToFloat step whose input comes from a staged
    // quantizer with producer statistics reads the quantizer's f32 INPUT and quantizes in its own operand loader (rten_hip_conv2d_int8_dql:
    // DynamicQuantizeLinear + ConvIntegerToFloat of the reference, src/ops/quantize.rs:352-436 + src/ops/conv.rs:495-587, in one launch, same bits).
    // The quantizer's step stays when another convolution still reads its codes (a stage's shortcut) and goes when none does.
    void plan_dql_loaders() {
        if (opt_.fused_dql.empty()) return;
        std::map<int, size_t> producer;
        for (size_t i = 0; i < steps_.size(); i++) for (int id : steps_[i].out) if (id >= 0) producer[id] = i;
        for (size_t i = 0; i < steps_.size(); i++) {
            Step &cs = steps_[i];
            if (!cs.i8 || !cs.i8->to_float || cs.i8->qout_producer || !opt_.fused_dql.count(cs.name) || !cs.i8->sg.x_staged || !cs.i8->sg.packed_weight || !cs.i8->sg.packed_weight->len()) continue;
            auto pit = producer.find(cs.in[0]);
            if (pit == producer.end()) continue;
            Step &D = steps_[pit->second];
            if (!D.dql_staged || D.removed || !D.dql_staged->stats_in || D.out.empty() || D.out[0] != cs.in[0]) continue; // (a quantizer merged into its producer has no step)
            // The loader form pays only when the quantizer's own launch DISAPPEARS: every reader of its codes must be a listed, convertible layer.
            // A stage's first 1x1 shares its quantizer with the shortcut convolution: converting it alone keeps the staging launch and swaps a
            // 15-22 us convolution for a 44 us one (measured, session r5c) -- the runner's rule too (`_staged_key == geom`: no loader form).
            {
                bool all_listed = true;
                for (auto &other : steps_) {
                    if (&other == &cs || other.removed) continue;
                    for (size_t k = 0; k < other.in.size() && k < 1; k++)
                        if (other.in[k] == D.out[0] && !(other.i8 && opt_.fused_dql.count(other.name))) all_listed = false;
                    for (size_t k = 1; k < other.in.size(); k++)
                        if (other.in[k] == D.out[0]) all_listed = false; // the codes read as anything but a convolution's X
                }
                if (!all_listed) continue;
            }
            // geometry the loader form covers: 1x1, stride 1, no padding, groups 1, C % 64 == 0 -- all load-time facts
            const Tensor &w = consts_.at(cs.in[1]);
            const Conv &cv = cs.i8->op->conv;
            const bool unit = cv.strides == std::vector<int>{1, 1} && cv.dilations == std::vector<int>{1, 1} && !cv.padding.same && cv.padding.fixed == std::vector<int>{0, 0, 0, 0};
            if (w.ndim() != 4 || w.size(2) != 1 || w.size(3) != 1 || !unit || cv.groups != 1 || w.size(1) % 64 != 0 || w.dtype() != DType::I8) continue;
            if (cs.in[3] >= 0) { // a weight zero point: only the constant scalar 0 of symmetric weights
                auto zi = consts_.find(cs.in[3]);
                if (zi == consts_.end() || zi->second.len() != 1 || zi->second.dtype() != DType::I8 || zi->second.to_host<int8_t>()[0] != 0) continue;
            }
            // the conv's scale must be one of the products Mul(x_scale, w_scale) folded into the quantizer: w_scale is that Mul's constant
            const Tensor *w_scale = nullptr;
            for (size_t k = 0; k < D.dql_staged->mul_by.size(); k++) if (D.out.size() > 3 + k && D.out[3 + k] == cs.in[4]) w_scale = D.dql_staged->mul_by[k];
            if (!w_scale) continue;
            const void *stats_in = D.dql_staged->stats_in;
            auto state = cs.i8;
            const int f32_in = D.in[0];
            cs.in[0] = f32_in; cs.in[2] = -1; cs.in[4] = -1;
            cs.kind_name = "DynamicQuantizeLinear+" + cs.kind_name + " (quantize on load)";
            cs.run = [state, w_scale, stats_in](Context &c, const InputList &in) {
                const Tensor &x = want(require(in, 0), DType::F32, "float32"), &w = require(in, 1);
                const Tensor *bias = in[5], *residual = in[6];
                rten_hip_conv2d_int8_desc di{};
                di.conv = state->op->conv.geometry(x.shape(), w.shape());
                di.x_signed = 0; di.w_signed = 1; di.pad_mode = state->op->pad_mode; di.weights_packed = 1;
                if (bias && bias->len() != di.conv.o) throw OpError(OpError::IncompatibleInputShapes, "bias length does not match output channels");
                if (residual && residual->shape() != std::vector<int64_t>{di.conv.n, di.conv.o, di.conv.out_h, di.conv.out_w})
                    throw OpError(OpError::IncompatibleInputShapes, "quantize-on-load convolution: the residual must have the output's shape");
                Tensor y(c, {di.conv.n, di.conv.o, di.conv.out_h, di.conv.out_w}, DType::F32);
                const uint32_t flags = (state->relu ? RTEN_HIP_CONV_RELU : 0u) | (residual ? RTEN_HIP_CONV_RESIDUAL : 0u);
                c.check(rten_hip_conv2d_int8_dql(c.raw(), &di, (const float *)x.ptr(), stats_in, state->sg.packed_weight->ptr(), (const float *)w_scale->ptr(),
                                                 (const float *)vp(bias), (const float *)vp(residual), flags, (float *)y.ptr(), state->sg.stats_out, nullptr, nullptr));
                OutputList out;
                out.push_back(std::move(y));
                return out;
            };
            dql_loader_steps_++;
            // does anybody still read the quantizer's outputs?
            bool read = false;
            for (auto &st : steps_) {
                if (&st == &D || st.removed) continue;
                for (int id : st.in) for (int o : D.out) if (id >= 0 && id == o) read = true;
            }
            for (auto &o : outputs_) for (int oid : D.out) if (ids_.at(o.name) == oid) read = true;
            if (!read) { D.removed = true; fused_away_++; }
        }
        steps_.erase(std::remove_if(steps_.begin(), steps_.end(), [](const Step &st) { return st.removed; }), steps_.end());
    }
    uint64_t graph_ = 0;
    std::vector<Tensor> captured_outputs_;

    void tune_step(Step &st, const InputList &in) {
        std::vector<GemmPlan> plans;
        const int nvar = rten_hip_num_gemm_variants();
        auto set_plan = [&](const GemmPlan &p) { if (st.conv) st.conv->plan = p; else *st.gemm_plan = p; };
        if (st.gemm_plan) {
            // MatMul-family steps (row-major A): the LDS-DMA pipelines with three / four stages x tile order (m fastest / n fastest within an XCD's
            // share: which operand stays L2-resident) -- the candidate set of rten_amd/workloads/bert.py::autotune
            for (int v : {0, 1, 2, 3, 12, 13, 14, 15}) if (v < nvar) for (int o = 0; o < 2; o++) plans.push_back(GemmPlan{true, v, 3, 1, o});
            // + the small-M weight-streaming kernel (variant 31: one batch, M <= 64; any other call runs it as variant 3, i.e. a duplicate candidate)
            if (31 < nvar) plans.push_back(GemmPlan{true, 31, 3, 1, 0});
        } else {
        const Tensor &w = require(in, 1);
        const int64_t k = w.len() / std::max<int64_t>(w.size(0), 1); // per-group depth C/g * kh * kw
        const int nblk = (int)((k + 255) / 256);
        for (int v = 0; v < nvar; v++) for (int o = 0; o < 2; o++) plans.push_back(GemmPlan{true, v, 0, 1, o});
        // thin-tile tail (split mode 4) on the LDS-DMA pipelines: whole rounds with the variant's tile + 16x64 tiles on 16x16x4 MFMAs
        for (int v = 0; v < nvar; v++) if (v < 4 || v >= 12) for (int o = 0; o < 2; o++) plans.push_back(GemmPlan{true, v, 4, 1, o});
        // persistent plan (split mode 5, groups = workgroups per compute unit): tile list per workgroup, DMA across tile boundaries
        for (int v : {0, 1, 2, 3, 20, 21, 22, 23}) if (v < nvar) for (int r = 1; r <= 4; r++) for (int o = 0; o < 2; o++) plans.push_back(GemmPlan{true, v, 5, r, o});
        // lean persistent kernel (split mode 6: 64x64 tiles, K a multiple of 32)
        for (int r = 1; r <= 3; r++) for (int o = 0; o < 2; o++) plans.push_back(GemmPlan{true, 3, 6, r, o});
        if (nblk > 1) {
            static const int split_variants[] = {0, 1, 2, 3, 4, 5, 6, 7, 12, 13, 14, 15, 16, 17, 18, 19};
            for (int v : split_variants) {
                if (v >= nvar) continue;
                std::set<int> group_counts;
                for (int g : {2, 3, 4, 6, nblk}) if (g >= 2 && g <= nblk) group_counts.insert(g);
                for (int g : group_counts) {
                    plans.push_back(GemmPlan{true, v, 1, g, 0});
                    for (int o : {0, 2, 3}) plans.push_back(GemmPlan{true, v, 2, g, o});
                }
            }
        }
        }
        GemmPlan best;
        float best_ms = 1e30f;
        for (const GemmPlan &p : plans) {
            set_plan(p);
            try {
                st.run(ctx_, in); // warm (also grows the split-K slab scratch before any capture)
                float ms = 1e30f;
                for (int round = 0; round < 2; round++) { // best of two short runs
                    ctx_.check(rten_hip_timer_start(ctx_.raw(), 2));
                    for (int r = 0; r < tune_reps_; r++) st.run(ctx_, in);
                    ctx_.check(rten_hip_timer_stop(ctx_.raw(), 2));
                    float t = 0;
                    ctx_.check(rten_hip_timer_elapsed_ms(ctx_.raw(), 2, &t));
                    ms = std::min(ms, t / (float)tune_reps_);
                }
                if (ms < best_ms) { best_ms = ms; best = p; }
            } catch (const OpError &) { // a plan the kernel family does not offer for this shape
            }
        }
        set_plan(best);
        tuned_++;
    }

    int id_of(const std::string &n) {
        if (n.empty()) return -1; // omitted optional input
        auto it = ids_.find(n);
        if (it != ids_.end()) return it->second;
        const int id = (int)names_.size();
        names_.push_back(n);
        ids_[n] = id;
        return id;
    }

    static DType dtype_of(int onnx_type, const std::string &what) {
        switch (onnx_type) {
        case onnx::FLOAT: return DType::F32;
        case onnx::INT32: case onnx::INT64: case onnx::BOOL: return DType::I32; // int64 and bool are int32 from load on (onnx_loader.rs:332-339,464-492)
        case onnx::UINT8: return DType::U8;
        case onnx::INT8: return DType::I8;
        default: throw GraphError("unsupported tensor element type " + std::to_string(onnx_type) + " (" + what + ")");
        }
    }

    Tensor upload(const onnx::TensorProto &t) {
        const DType dt = dtype_of(t.data_type, t.name);
        const int64_t n = t.len();
        const size_t esz = t.data_type == onnx::INT64 ? 8 : t.data_type == onnx::BOOL ? 1 : dtype_size(dt);
        if ((size_t)n * esz != t.raw.size()) throw GraphError("initializer " + t.name + ": data size does not match its dims");
        // small integer / float constants keep a host copy: the operands of shape arithmetic (hostops above)
        constexpr int64_t kHostConst = 4096;
        if (t.data_type == onnx::INT64 || t.data_type == onnx::BOOL) {
            std::vector<int32_t> narrow((size_t)n);
            for (int64_t i = 0; i < n; i++) {
                if (t.data_type == onnx::BOOL) { narrow[(size_t)i] = t.raw[(size_t)i] != 0; continue; }
                int64_t v;
                std::memcpy(&v, t.raw.data() + 8 * i, 8);
                narrow[(size_t)i] = (int32_t)std::max<int64_t>(INT32_MIN, std::min<int64_t>(INT32_MAX, v)); // saturating, like the loader
            }
            Tensor d = Tensor::from_host<int32_t>(ctx_, t.dims, narrow.data());
            if (n <= kHostConst) { auto h = std::make_shared<HostVal>(); h->shape = t.dims; h->i.assign(narrow.begin(), narrow.end()); d.set_host(h); }
            return d;
        }
        Tensor d(ctx_, t.dims, dt);
        if (d.bytes() && !(opt_.skip_large_uploads && d.bytes() >= ((size_t)64 << 10)))
            ctx_.check(rten_hip_memcpy_h2d(ctx_.raw(), d.ptr(), t.raw.data(), d.bytes()));
        if (n <= kHostConst && (dt == DType::I32 || dt == DType::F32)) {
            auto h = std::make_shared<HostVal>();
            h->shape = t.dims;
            h->is_float = dt == DType::F32;
            if (h->is_float) { h->f.resize((size_t)n); std::memcpy(h->f.data(), t.raw.data(), (size_t)n * 4); }
            else { h->i.resize((size_t)n); for (int64_t i = 0; i < n; i++) { int32_t v; std::memcpy(&v, t.raw.data() + 4 * i, 4); h->i[(size_t)i] = v; } }
            d.set_host(h);
        }
        return d;
    }

    // ---- host-evaluated steps (hostops above).  The device copy of a host-computed value is made once per distinct value and cached with the step: shapes
    // are fixed between prepare and the next bind_input, so the eager warm-up run that precedes every capture fills the cache and the captured run only
    // finds hits (an upload inside a capture would be recorded with the host pointer: refused).
    struct HostCache { std::vector<std::pair<std::shared_ptr<const HostVal>, std::unique_ptr<Tensor>>> entries; };
    static Tensor materialize(Context &c, HostVal &&hv, HostCache &cache) {
        for (auto &e : cache.entries)
            if (*e.first == hv) { Tensor v = Tensor::view_of(*e.second, e.second->shape()); v.set_host(e.first); return v; }
        if (rten_hip_capture_active(c.raw()))
            throw GraphError("a shape-dependent constant changed between the warm-up run and the capture (input shapes must stay fixed while a graph is captured)");
        auto h = std::make_shared<const HostVal>(std::move(hv));
        std::unique_ptr<Tensor> t;
        if (h->is_float) t.reset(new Tensor(Tensor::from_host<float>(c, h->shape, h->f.data())));
        else { std::vector<int32_t> w(h->i.begin(), h->i.end()); t.reset(new Tensor(Tensor::from_host<int32_t>(c, h->shape, w.data()))); }
        if (cache.entries.size() >= 8) cache.entries.erase(cache.entries.begin()); // (shapes changed a few times: keep the recent ones)
        cache.entries.emplace_back(h, std::move(t));
        Tensor v = Tensor::view_of(*cache.entries.back().second, h->shape);
        v.set_host(h);
        return v;
    }
    static std::vector<int64_t> host_ints(const Tensor *t, const std::string &what) {
        if (!t) return {};
        if (!t->host() || t->host()->is_float) throw OpError(OpError::UnsupportedValue, what + " must be a constant or computable from the input shapes (it depends on device data)");
        return t->host()->i;
    }
    // A step that runs on the host when every operand carries a host value (`hf` returns false to decline) and on the device otherwise.
    using HostFn = std::function<bool(const std::vector<const HostVal *> &, HostVal &)>;
    using DevFn = std::function<OutputList(Context &, const InputList &)>;
    static void make_hostable(Step &st, HostFn hf, DevFn df) {
        auto cache = std::make_shared<HostCache>();
        st.run = [hf, df, cache](Context &c, const InputList &in) {
            bool all = true;
            std::vector<const HostVal *> hv;
            for (const Tensor *t : in) {
                if (!t) { hv.push_back(nullptr); continue; }
                if (!t->host()) { all = false; break; }
                hv.push_back(t->host());
            }
            if (all) {
                HostVal out;
                if (hf(hv, out)) {
                    OutputList o;
                    o.push_back(materialize(c, std::move(out), *cache));
                    return o;
                }
            }
            return df(c, in);
        };
    }

    static Padding padding_of(const onnx::Node &n, const char *op) {
        const onnx::Attr *ap = n.attr("auto_pad");
        if (ap && !ap->s.empty() && ap->s != "NOTSET") {
            if (ap->s == "SAME_UPPER") return Padding::Same();
            if (ap->s == "VALID") return Padding::Fixed({0, 0, 0, 0});
            throw GraphError(std::string(op) + " " + n.name + ": auto_pad " + ap->s + " is not supported");
        }
        std::vector<int> p = n.get_ints("pads", {0, 0, 0, 0});
        if (p.size() != 4) throw GraphError(std::string(op) + " " + n.name + ": only 2-D spatial operators are supported");
        return Padding::Fixed({p[0], p[1], p[2], p[3]}); // ONNX [top, left, bottom, right] == the reference's order
    }
    static Conv conv_attrs(const onnx::Node &n) {
        Conv c;
        c.groups = (int)n.get_int("group", 1);
        c.dilations = n.get_ints("dilations", {1, 1});
        c.strides = n.get_ints("strides", {1, 1});
        c.padding = padding_of(n, "Conv");
        return c;
    }

    // ---- canonicalisation of exporter idioms, on the ONNX node list (the reference does these in its graph optimiser, so its
    //      numerics are the FUSED operators' numerics -- they are applied whether or not the backend's own epilogue fusions are on):
    //   * Constant nodes become initializers (`Constant` -> constant node, rten-onnx loader);
    //   * x * (Erf(x / sqrt(2) | x * (1 / sqrt(2))) + 1) * 0.5                              -> Gelu               (GeluFusion, optimize/fusions.rs:407-430)
    //   * (x - ReduceMean(x)) / Sqrt(eps + ReduceMean(Pow(x - ReduceMean(x), 2))) * scale [+ bias], means over the last axis
    //                                                                                      -> LayerNormalization (fusions.rs:674-747)
    //     (PyTorch's exporter writes nn.LayerNorm and nn.GELU this way).
    static onnx::Model canonicalize(const onnx::Model &src) {
        onnx::Model m = src;
        for (auto &n : m.nodes)
            if (n.op_type == "Constant" && n.outputs.size() == 1) {
                const onnx::Attr *v = n.attr("value");
                if (!v) throw GraphError("Constant " + n.name + ": only the tensor `value` form is supported");
                onnx::TensorProto t = v->t;
                t.name = n.outputs[0];
                m.initializers.push_back(std::move(t));
                n.op_type.clear(); // removed below
            }
        std::map<std::string, size_t> producer, uses;
        std::map<std::string, const onnx::TensorProto *> inits;
        auto index = [&] {
            producer.clear(); uses.clear(); inits.clear();
            for (size_t i = 0; i < m.nodes.size(); i++) {
                if (m.nodes[i].op_type.empty()) continue;
                for (auto &o : m.nodes[i].outputs) if (!o.empty()) producer[o] = i;
                for (auto &in : m.nodes[i].inputs) if (!in.empty()) uses[in]++;
            }
            for (auto &o : m.outputs) uses[o.name] += 2; // a graph output is never an interior value of a pattern
            for (auto &t : m.initializers) inits[t.name] = &t;
        };
        index();
        auto scalar = [&](const std::string &v, float &out) {
            auto it = inits.find(v);
            if (it == inits.end() || it->second->data_type != onnx::FLOAT || it->second->len() != 1 || it->second->raw.size() != 4) return false;
            std::memcpy(&out, it->second->raw.data(), 4);
            return true;
        };
        auto is_init = [&](const std::string &v) { return inits.count(v) != 0; };
        auto node_of = [&](const std::string &v, const char *op, bool interior = true) -> onnx::Node * { // producer of v if it is `op` (and v has one use)
            auto it = producer.find(v);
            if (it == producer.end() || m.nodes[it->second].op_type != op) return nullptr;
            if (interior && uses[v] != 1) return nullptr;
            return &m.nodes[it->second];
        };
        // binary node with one operand equal to a scalar constant c (|c - want| tiny); returns the other operand
        auto with_scalar = [&](onnx::Node *n, float want, bool commutative, std::string &other) {
            if (!n || n->inputs.size() != 2) return false;
            for (int k = 0; k < (commutative ? 2 : 1); k++) {
                float c;
                const std::string &cs = n->inputs[(size_t)(commutative ? k : 1)], &os = n->inputs[(size_t)(commutative ? 1 - k : 0)];
                if (scalar(cs, c) && std::fabs(c - want) <= 1e-6f * std::fabs(want)) { other = os; return true; }
            }
            return false;
        };
        auto last_axis_mean = [&](onnx::Node *n) {
            if (!n || n->inputs.empty() || n->get_int("keepdims", 1) != 1) return false;
            std::vector<int> axes = n->get_ints("axes", {});
            if (axes.empty() && n->inputs.size() > 1 && is_init(n->inputs[1])) { // opset >= 18: axes as a constant input (ReduceMeanAxesFusion)
                const onnx::TensorProto *t = inits[n->inputs[1]];
                if (t->data_type == onnx::INT64 && t->len() == 1 && t->raw.size() == 8) { int64_t a; std::memcpy(&a, t->raw.data(), 8); axes = {(int)a}; }
            }
            return axes.size() == 1 && axes[0] == -1;
        };
        const float sqrt2 = std::sqrt(2.0f);
        for (size_t i = 0; i < m.nodes.size(); i++) {
            onnx::Node &last = m.nodes[i];
            if (last.op_type == "Mul") { // ---- Gelu
                std::string a, xin, e_plus_1, erf_out, xs, x2;
                if (!with_scalar(&last, 0.5f, true, a)) continue;
                onnx::Node *mul = node_of(a, "Mul");
                if (!mul || mul->inputs.size() != 2) continue;
                onnx::Node *add = nullptr;
                for (int k = 0; k < 2 && !add; k++) { add = node_of(mul->inputs[(size_t)k], "Add"); if (add) { e_plus_1 = mul->inputs[(size_t)k]; xin = mul->inputs[(size_t)(1 - k)]; } }
                if (!add || !with_scalar(add, 1.0f, true, erf_out)) continue;
                onnx::Node *erf = node_of(erf_out, "Erf");
                if (!erf || erf->inputs.size() != 1) continue;
                onnx::Node *scale = node_of(erf->inputs[0], "Div");
                bool ok = scale && with_scalar(scale, sqrt2, false, x2);
                if (!ok) { scale = node_of(erf->inputs[0], "Mul"); ok = scale && with_scalar(scale, 1.0f / sqrt2, true, x2); }
                if (!ok || x2 != xin) continue;
                onnx::Node g;
                g.op_type = "Gelu"; g.name = last.name.empty() ? "gelu" : last.name; g.inputs = {xin}; g.outputs = last.outputs;
                scale->op_type.clear(); erf->op_type.clear(); add->op_type.clear(); mul->op_type.clear();
                last = g;
                index();
            } else if (last.op_type == "Add" || last.op_type == "Mul") {
                continue;
            }
        }
        for (size_t i = 0; i < m.nodes.size(); i++) { // ---- LayerNormalization: anchored at the scaling Mul
            onnx::Node &mulnode = m.nodes[i];
            if (mulnode.op_type != "Mul" || mulnode.inputs.size() != 2) continue;
            onnx::Node *div = nullptr;
            std::string scale_name;
            for (int k = 0; k < 2 && !div; k++)
                if (is_init(mulnode.inputs[(size_t)(1 - k)])) { div = node_of(mulnode.inputs[(size_t)k], "Div"); scale_name = mulnode.inputs[(size_t)(1 - k)]; }
            if (!div || div->inputs.size() != 2) continue;
            onnx::Node *sqrt_n = node_of(div->inputs[1], "Sqrt");
            auto cit = producer.find(div->inputs[0]);
            if (!sqrt_n || cit == producer.end() || m.nodes[cit->second].op_type != "Sub" || uses[div->inputs[0]] != 2) continue;
            onnx::Node *center = &m.nodes[cit->second];
            onnx::Node *addeps = node_of(sqrt_n->inputs[0], "Add");
            if (!addeps || addeps->inputs.size() != 2) continue;
            float eps = 0.f;
            onnx::Node *mean2 = nullptr;
            for (int k = 0; k < 2 && !mean2; k++) if (scalar(addeps->inputs[(size_t)k], eps)) mean2 = node_of(addeps->inputs[(size_t)(1 - k)], "ReduceMean");
            if (!mean2 || !last_axis_mean(mean2)) continue;
            onnx::Node *pow = node_of(mean2->inputs[0], "Pow");
            std::string powed;
            if (!pow || !with_scalar(pow, 2.0f, false, powed) || powed != div->inputs[0]) continue;
            onnx::Node *mean1 = node_of(center->inputs[1], "ReduceMean");
            if (!mean1 || !last_axis_mean(mean1) || mean1->inputs[0] != center->inputs[0]) continue;
            const std::string x = center->inputs[0];
            onnx::Node ln;
            ln.op_type = "LayerNormalization";
            ln.inputs = {x, scale_name};
            onnx::Node *tail = &mulnode; // optional "+ bias" (a constant): the fused node then takes the Add's place
            if (uses[mulnode.outputs[0]] == 1)
                for (size_t j = i + 1; j < m.nodes.size(); j++) {
                    onnx::Node &c = m.nodes[j];
                    if (c.op_type != "Add" || c.inputs.size() != 2) continue;
                    const int k = c.inputs[0] == mulnode.outputs[0] ? 0 : (c.inputs[1] == mulnode.outputs[0] ? 1 : -1);
                    if (k < 0) continue;
                    if (is_init(c.inputs[(size_t)(1 - k)])) { ln.inputs.push_back(c.inputs[(size_t)(1 - k)]); tail = &c; }
                    break;
                }
            ln.name = tail->name.empty() ? "layer_norm" : tail->name;
            ln.outputs = tail->outputs;
            onnx::Attr axis, epsilon;
            axis.name = "axis"; axis.type = 2; axis.i = -1;
            epsilon.name = "epsilon"; epsilon.type = 1; epsilon.f = eps;
            ln.attrs = {axis, epsilon};
            mean1->op_type.clear(); center->op_type.clear(); pow->op_type.clear(); mean2->op_type.clear(); addeps->op_type.clear(); sqrt_n->op_type.clear();
            div->op_type.clear();
            if (tail != &mulnode) mulnode.op_type.clear();
            *tail = ln;
            index();
        }
        std::vector<onnx::Node> kept;
        for (auto &n : m.nodes) if (!n.op_type.empty()) kept.push_back(std::move(n));
        m.nodes = std::move(kept);
        return m;
    }

    // ---- compile: constants, fusion, steps, liveness
    void compile(const onnx::Model &m_in) {
        const onnx::Model m_canonical = canonicalize(m_in);
        const onnx::Model &m = m_canonical;
        inputs_ = m.inputs;
        outputs_ = m.outputs;
        for (auto &t : m.initializers) add_const(id_of(t.name), [&] { return upload(t); });
        for (auto &in : m.inputs) id_of(in.name);

        const size_t N = m.nodes.size();
        // consumers of every value (graph outputs count as a use: they must not be fused away)
        std::map<std::string, std::vector<size_t>> users;
        std::map<std::string, size_t> producer;
        for (size_t i = 0; i < N; i++) {
            for (auto &s : m.nodes[i].inputs) if (!s.empty()) users[s].push_back(i);
            for (auto &s : m.nodes[i].outputs) if (!s.empty()) producer[s] = i;
        }
        std::set<std::string> graph_outs;
        for (auto &o : m.outputs) graph_outs.insert(o.name);
        std::vector<bool> dead(N, false);
        const std::vector<bool> *dead_ptr = &dead;
        auto sole_user = [&](const std::string &v, const char *op) -> long {
            if (graph_outs.count(v)) return -1;
            auto it = users.find(v);
            if (it == users.end() || it->second.size() != 1 || dead_ptr->at(it->second[0])) return -1;
            return m.nodes[it->second[0]].op_type == op ? (long)it->second[0] : -1;
        };
        auto is_const = [&](const std::string &v) { auto it = ids_.find(v); return it != ids_.end() && consts_.count(it->second) != 0; };
        // A fused step runs at the position of the LAST node it absorbs (st.pos), where -- the node list being
        // topologically sorted -- every operand of every absorbed node is already available.  An Add of two operator
        // outputs is claimed by the LATER producer only (its other operand exists before that producer runs), so two
        // convolutions feeding one Add (projection shortcut + main branch) do not both absorb it.
        auto ready_before = [&](const std::string &v, size_t at) {
            auto it = producer.find(v);
            return it == producer.end() || it->second < at;
        };
        auto other_input = [&](const onnx::Node &n, const std::string &v) { return n.inputs[0] == v ? n.inputs[1] : n.inputs[0]; };

        // ---- attention pre-pass (TransposeFusion + MatMulScale + AddSoftmax of the reference, taken one step further):
        //   q_lin -> Reshape[0,0,h,d] -> Transpose(0,2,1,3) -+
        //   k_lin -> Reshape[0,0,h,d] -> Transpose(0,2,3,1) -+> MatMul -> (Div | Mul scalar) -> Add(mask) -> Softmax(-1) -+
        //   v_lin -> Reshape[0,0,h,d] -> Transpose(0,2,1,3) ------------------------------------------------------------+> MatMul -> Transpose(0,2,1,3) -> Reshape[0,0,H]
        // becomes one MultiHeadSdpa step over the [B, S, H] projections; if the three projections are MatMul(x, W) + Add(b)
        // of one input with constant weights, they become one GEMM against [Wq | Wk | Wv] whose column blocks the
        // attention kernel reads in place (bit-identical: every output element is the same k-ordered dot product).
        // lead0 / lead1: leading dims when the Reshapes spell them out ([B, S, h, d], as PyTorch's exporter writes a static-shape
        // `view`) instead of copying them ([0, 0, h, d]); checked against the projections at run time (0 = copied)
        struct Attn { std::string q, k, v, mask, out, x; int heads = 0; float scale = 1.f; bool merged = false; int wqkv = -1, bqkv = -1; int64_t hidden = 0; int lead0 = 0, lead1 = 0; int head_dim = 0;
                      std::string lead_check; }; // a run-time Reshape target (dynamic-axes export): its leading dims are checked against the projection's when the step runs
        std::map<size_t, Attn> attn_at;
        if (opt_.fuse) {
            auto single_use = [&](const std::string &v) { auto it = users.find(v); return !graph_outs.count(v) && it != users.end() && it->second.size() == 1; };
            auto made_by = [&](const std::string &v, const char *op) -> long {
                auto it = producer.find(v);
                return it != producer.end() && !dead[it->second] && m.nodes[it->second].op_type == op && single_use(v) ? (long)it->second : -1;
            };
            auto const_i32 = [&](const std::string &v, std::vector<int32_t> &out) {
                if (!is_const(v)) return false;
                const Tensor &t = consts_.at(ids_.at(v));
                if (t.dtype() != DType::I32) return false;
                out = t.to_host<int32_t>();
                return true;
            };
            // value -> (Reshape[0,0,h,d] node, Transpose node) feeding it with the given perm; returns the projection's name
            auto leading_ok = [](const std::vector<int32_t> &shp, Attn &at, bool first) { // [0, 0, ..] or one explicit [B, S, ..] throughout
                if ((shp[0] == 0) != (shp[1] == 0) || shp[0] < 0 || shp[1] < 0) return false;
                if (first) { at.lead0 = shp[0]; at.lead1 = shp[1]; return true; }
                return at.lead0 == shp[0] && at.lead1 == shp[1];
            };
            // The target of a head-split / head-merge Reshape: a constant, or -- exports with dynamic axes -- Concat(dim 0, dim 1, [h | -1], [d]) whose leading
            // entries are computed from a Shape at run time and whose trailing entries are constants.  The run-time form reads as [0, 0, ..] ("copy the
            // projection's leading dims"); `dyn` names the Concat's output so that the fused step can check that claim against its host value.
            auto reshape_target = [&](const onnx::Node &rn, std::vector<int32_t> &shp, std::string &dyn) {
                dyn.clear();
                if (rn.inputs.size() < 2) return false;
                if (const_i32(rn.inputs[1], shp)) return true;
                const long c = made_by(rn.inputs[1], "Concat");
                if (c < 0 || m.nodes[(size_t)c].get_int("axis", 0) != 0 || m.nodes[(size_t)c].inputs.size() < 3) return false;
                const onnx::Node &cn = m.nodes[(size_t)c];
                shp.assign(cn.inputs.size(), 0);
                for (size_t k = 0; k < cn.inputs.size(); k++) {
                    std::vector<int32_t> one;
                    if (k < 2) { if (is_const(cn.inputs[k])) return false; continue; } // (a half-constant leading pair is not this idiom)
                    if (!const_i32(cn.inputs[k], one) || one.size() != 1) return false;
                    shp[k] = one[0];
                }
                dyn = rn.inputs[1];
                return true;
            };
            auto split_heads = [&](const std::string &v, std::vector<int> perm, Attn &at, bool first, std::vector<size_t> &nodes) -> std::string {
                int &heads = at.heads;
                const long t = made_by(v, "Transpose");
                if (t < 0 || m.nodes[(size_t)t].get_ints("perm", {}) != perm) return "";
                const long r = made_by(m.nodes[(size_t)t].inputs[0], "Reshape");
                std::vector<int32_t> shp;
                std::string dyn;
                // [.., .., h, d] or [.., .., -1, d]: transformers' exporter spells the head COUNT as -1 (hidden / d, resolved when the step runs)
                if (r < 0 || !reshape_target(m.nodes[(size_t)r], shp, dyn) || shp.size() != 4 || (shp[2] <= 0 && !(shp[2] == -1 && shp[3] > 0)) || !leading_ok(shp, at, first)) return "";
                if (first) at.lead_check = dyn;
                if (heads && heads != shp[2]) return "";
                // one head size for q, k and v (the kernel's d == dv); -1 ("the rest") is accepted only if all three say so
                if (heads && at.head_dim != shp[3]) return "";
                heads = shp[2];
                at.head_dim = shp[3];
                nodes.push_back((size_t)t); nodes.push_back((size_t)r);
                return m.nodes[(size_t)r].inputs[0];
            };
            for (size_t i = 0; i < N; i++) {
                const onnx::Node &mm = m.nodes[i];
                if (dead[i] || mm.op_type != "MatMul" || mm.inputs.size() != 2) continue;
                Attn at;
                std::vector<size_t> nodes{i};
                at.q = split_heads(mm.inputs[0], {0, 2, 1, 3}, at, true, nodes);
                at.k = at.q.empty() ? "" : split_heads(mm.inputs[1], {0, 2, 3, 1}, at, false, nodes);
                if (at.k.empty()) continue;
                std::string cur = mm.outputs[0];
                for (const char *sop : {"Div", "Mul"}) {
                    const long d = sole_user(cur, sop);
                    if (d < 0 || m.nodes[(size_t)d].inputs[0] != cur || !is_const(m.nodes[(size_t)d].inputs[1])) continue;
                    const Tensor &c = consts_.at(ids_.at(m.nodes[(size_t)d].inputs[1]));
                    if (c.len() != 1 || c.dtype() != DType::F32 || at.scale != 1.f) continue;
                    const float cv = c.to_host<float>()[0];
                    at.scale = std::string(sop) == "Div" ? 1.0f / cv : cv;
                    nodes.push_back((size_t)d);
                    cur = m.nodes[(size_t)d].outputs[0];
                }
                const long add = sole_user(cur, "Add");
                if (add >= 0) { at.mask = other_input(m.nodes[(size_t)add], cur); nodes.push_back((size_t)add); cur = m.nodes[(size_t)add].outputs[0]; }
                const long sm = sole_user(cur, "Softmax");
                const int64_t sm_axis = sm < 0 ? 0 : m.nodes[(size_t)sm].get_int("axis", -1);
                if (sm < 0 || (sm_axis != -1 && sm_axis != 3)) continue; // the scores are 4-D here: PyTorch's exporter writes the last axis as 3
                nodes.push_back((size_t)sm);
                cur = m.nodes[(size_t)sm].outputs[0];
                const long pv = sole_user(cur, "MatMul");
                if (pv < 0 || m.nodes[(size_t)pv].inputs[0] != cur) continue;
                nodes.push_back((size_t)pv);
                at.v = split_heads(m.nodes[(size_t)pv].inputs[1], {0, 2, 1, 3}, at, false, nodes);
                if (at.v.empty()) continue;
                const long tr = sole_user(m.nodes[(size_t)pv].outputs[0], "Transpose");
                if (tr < 0 || m.nodes[(size_t)tr].get_ints("perm", {}) != std::vector<int>{0, 2, 1, 3}) continue;
                const long rs = sole_user(m.nodes[(size_t)tr].outputs[0], "Reshape");
                std::vector<int32_t> shp;
                std::string dyn_out;
                if (rs < 0 || !reshape_target(m.nodes[(size_t)rs], shp, dyn_out) || shp.size() != 3 || !leading_ok(shp, at, false)) continue;
                nodes.push_back((size_t)tr); nodes.push_back((size_t)rs);
                at.out = m.nodes[(size_t)rs].outputs[0];
                if (!at.mask.empty() && producer.count(at.mask) && producer[at.mask] > i) continue; // mask must exist before the scores
                if (!at.mask.empty() && is_const(at.mask)) { // a constant mask with a head axis is outside the kernel's forms: leave the graph unfused
                    const Tensor &mk = consts_.at(ids_.at(at.mask));
                    if (mk.dtype() != DType::F32 || mk.ndim() > 4 || (mk.ndim() == 4 && mk.size(1) != 1) || (mk.ndim() == 3 && mk.size(0) != 1)) continue;
                }
                // optional: one GEMM for the three projections
                struct Lin { long mm = -1, add = -1; std::string x, w, b; };
                auto linear = [&](const std::string &v) {
                    Lin l;
                    const long a = made_by(v, "Add");
                    if (a < 0) return l;
                    for (int side = 0; side < 2; side++) {
                        const std::string &mmv = m.nodes[(size_t)a].inputs[(size_t)side], &bv = m.nodes[(size_t)a].inputs[(size_t)(1 - side)];
                        const long mmi = made_by(mmv, "MatMul");
                        if (mmi < 0 || !is_const(bv) || !is_const(m.nodes[(size_t)mmi].inputs[1])) continue;
                        const Tensor &w = consts_.at(ids_.at(m.nodes[(size_t)mmi].inputs[1])), &b = consts_.at(ids_.at(bv));
                        if (w.ndim() != 2 || b.ndim() != 1 || b.len() != w.size(1) || w.dtype() != DType::F32 || b.dtype() != DType::F32) continue;
                        l.mm = mmi; l.add = a; l.x = m.nodes[(size_t)mmi].inputs[0]; l.w = m.nodes[(size_t)mmi].inputs[1]; l.b = bv;
                        return l;
                    }
                    return l;
                };
                const Lin lq = linear(at.q), lk = linear(at.k), lv = linear(at.v);
                if (lq.mm >= 0 && lk.mm >= 0 && lv.mm >= 0 && lq.x == lk.x && lq.x == lv.x) {
                    const Tensor &wq = consts_.at(ids_.at(lq.w)), &wk = consts_.at(ids_.at(lk.w)), &wv = consts_.at(ids_.at(lv.w));
                    if (wq.shape() == wk.shape() && wq.shape() == wv.shape()) {
                        const int64_t K = wq.size(0), Nn = wq.size(1);
                        const std::vector<float> hq = wq.to_host<float>(), hk = wk.to_host<float>(), hv = wv.to_host<float>();
                        std::vector<float> wcat((size_t)(K * 3 * Nn)), bcat;
                        for (int64_t r = 0; r < K; r++) {
                            std::copy(hq.begin() + r * Nn, hq.begin() + (r + 1) * Nn, wcat.begin() + r * 3 * Nn);
                            std::copy(hk.begin() + r * Nn, hk.begin() + (r + 1) * Nn, wcat.begin() + r * 3 * Nn + Nn);
                            std::copy(hv.begin() + r * Nn, hv.begin() + (r + 1) * Nn, wcat.begin() + r * 3 * Nn + 2 * Nn);
                        }
                        for (const std::string *bn : {&lq.b, &lk.b, &lv.b}) { const std::vector<float> hb = consts_.at(ids_.at(*bn)).to_host<float>(); bcat.insert(bcat.end(), hb.begin(), hb.end()); }
                        at.wqkv = id_of("__qkv_w." + std::to_string(i));
                        at.bqkv = id_of("__qkv_b." + std::to_string(i));
                        add_const(at.wqkv, [&] { return Tensor::from_host<float>(ctx_, {K, 3 * Nn}, wcat.data()); });
                        add_const(at.bqkv, [&] { return Tensor::from_host<float>(ctx_, {3 * Nn}, bcat.data()); });
                        at.merged = true; at.x = lq.x; at.hidden = Nn;
                        for (long d : {lq.mm, lq.add, lk.mm, lk.add, lv.mm, lv.add}) nodes.push_back((size_t)d);
                    }
                }
                for (size_t d : nodes) dead[d] = true;
                fused_away_ += nodes.size() - (at.merged ? 2 : 1);
                attn_at[(size_t)rs] = at;
            }
        }

        for (size_t i = 0; i < N; i++) {
            auto af = attn_at.find(i);
            if (af != attn_at.end()) {
                const Attn &at = af->second;
                auto op = std::make_shared<MultiHeadSdpa>();
                op->heads = at.heads; op->head_dim = at.head_dim; op->scale = at.scale; op->flush_nans_to_zero = false;
                Step st;
                st.pos = i;
                st.name = at.out;
                st.kind_name = at.merged ? "MultiHeadSdpa(QKV column blocks)" : "MultiHeadSdpa";
                if (at.merged) {
                    auto lin = std::make_shared<FusedMatMul>();
                    Step pj;
                    pj.pos = i;
                    pj.name = at.out + ".qkv";
                    pj.kind_name = "FusedMatMul(QKV)";
                    pj.in = {id_of(at.x), at.wqkv, at.bqkv};
                    pj.out = {id_of("__qkv." + at.out)};
                    auto plan = std::make_shared<GemmPlan>();
                    pj.gemm_plan = plan;
                    pj.run = [lin, plan](Context &c, const InputList &in) { PlanScope scope(c, *plan); return lin->run(c, in); };
                    steps_.push_back(std::move(pj));
                    const int qkv = id_of("__qkv." + at.out);
                    op->q_rs = op->k_rs = op->v_rs = 3 * at.hidden; op->width = at.hidden;
                    op->q_off = 0; op->k_off = at.hidden; op->v_off = 2 * at.hidden;
                    st.in = {qkv, qkv, qkv, at.mask.empty() ? -1 : id_of(at.mask)};
                } else {
                    st.in = {id_of(at.q), id_of(at.k), id_of(at.v), at.mask.empty() ? -1 : id_of(at.mask)};
                }
                st.out = {id_of(at.out)};
                if (!at.lead_check.empty()) st.in.push_back(id_of(at.lead_check)); // (5th operand: the run-time Reshape target, a host value)
                const int lead0 = at.lead0, lead1 = at.lead1;
                st.run = [op, lead0, lead1](Context &c, const InputList &in) {
                    if (lead0 && (require(in, 0).ndim() != 3 || require(in, 0).size(0) != lead0 || require(in, 0).size(1) != lead1))
                        throw OpError(OpError::InvalidValue, "fused attention: the graph's Reshape spells out leading dims that differ from the projection's");
                    if (in.size() > 4 && in[4]) { // the Reshape target the graph computes at run time must say what the fusion assumed: [B, S, ..] of the projection
                        const std::vector<int64_t> tgt = host_ints(in[4], "fused attention: the head-split Reshape's target");
                        if (require(in, 0).ndim() != 3 || tgt.size() != 4 || tgt[0] != require(in, 0).size(0) || tgt[1] != require(in, 0).size(1))
                            throw OpError(OpError::InvalidValue, "fused attention: the graph's Reshape target differs from the projection's leading dims");
                        return op->run(c, InputList(in.begin(), in.begin() + 4));
                    }
                    return op->run(c, in);
                };
                steps_.push_back(std::move(st));
                continue;
            }
            if (dead[i]) continue;
            const onnx::Node &n = m.nodes[i];
            // contrib operators this backend carries live in com.microsoft (onnx_registry.rs registers them under that domain)
            const bool contrib = n.domain == "com.microsoft" && n.op_type == "MatMulNBits";
            if (!n.domain.empty() && n.domain != "ai.onnx" && !contrib) throw GraphError("node " + n.name + ": operator domain " + n.domain + " is not supported");
            Step st;
            st.name = n.name.empty() ? n.outputs.at(0) : n.name;
            st.kind_name = n.op_type;
            st.pos = i;
            for (auto &s : n.inputs) st.in.push_back(id_of(s));
            std::string out_name = n.outputs.at(0);

            if (n.op_type == "Conv") {
                auto op = std::make_shared<Conv>(conv_attrs(n));
                std::string residual;
                if (opt_.fuse) {
                    long a = sole_user(out_name, "Add");
                    if (a >= 0 && m.nodes[(size_t)a].inputs.size() == 2) {
                        const std::string other = other_input(m.nodes[(size_t)a], out_name);
                        if (other != out_name && ready_before(other, i)) { residual = other; dead[(size_t)a] = true; out_name = m.nodes[(size_t)a].outputs[0]; fused_away_++; st.pos = (size_t)a; }
                    }
                    long r = sole_user(out_name, "Relu");
                    if (r >= 0) { op->fuse_relu = true; dead[(size_t)r] = true; out_name = m.nodes[(size_t)r].outputs[0]; fused_away_++; st.pos = (size_t)r; }
                }
                while (st.in.size() < 3) st.in.push_back(-1);
                st.in.push_back(residual.empty() ? -1 : id_of(residual));
                const Tensor *packed = nullptr;
                if (opt_.prepack && is_const(n.inputs.at(1)) && op->groups == 1) {
                    const Tensor &w = consts_.at(ids_.at(n.inputs[1]));
                    if (w.ndim() == 4) packed = packed_for(n.name.empty() ? n.outputs.at(0) : n.name, [&] { return op->prepack(ctx_, w); });
                }
                st.kind_name = std::string("Conv") + (residual.empty() ? "" : "+Add") + (op->fuse_relu ? "+Relu" : "");
                st.conv = op;
                // The Add's other operand is used as the kernel's residual only when it has exactly the conv's output shape
                // (the kernel indexes it with the output's strides).  Its shape is a run-time fact (no shape inference here), and
                // it may be a constant or any broadcasting operand -- e.g. the exporter's explicit bias Add([1,O,1,1]) -- so a
                // mismatch runs the operators as the graph spells them: Conv, broadcasting Add, Relu.  (The constant is NOT
                // routed to the conv's bias input: the GEMM adds a bias after the first depth block, the graph adds it last.)
                st.run = [op, packed](Context &c, const InputList &in) {
                    const Tensor *res = in.size() > 3 ? in[3] : nullptr;
                    if (res) {
                        const Tensor &x = require(in, 0), &w = require(in, 1);
                        bool same = res->dtype() == DType::F32 && x.ndim() == 4 && w.ndim() == 4; // 1-D convs with an Add: always unfused
                        if (same) {
                            const rten_hip_conv2d_desc d = op->geometry(x.shape(), w.shape());
                            same = res->shape() == std::vector<int64_t>{d.n, d.o, d.out_h, d.out_w};
                        }
                        if (!same) {
                            Conv plain = *op; plain.fuse_relu = false;
                            OutputList y = plain.run_packed(c, {in[0], in[1], in.size() > 2 ? in[2] : nullptr}, packed);
                            OutputList sum = Add().run(c, {&y[0], res});
                            return op->fuse_relu ? Relu().run(c, {&sum[0]}) : std::move(sum);
                        }
                    }
                    return op->run_packed(c, in, packed);
                };
            } else if (n.op_type == "ConvTranspose") {
                auto op = std::make_shared<ConvTranspose>();
                op->groups = (int)n.get_int("group", 1);
                op->strides = n.get_ints("strides", {1, 1});
                op->dilations = n.get_ints("dilations", {1, 1});
                op->output_padding = n.get_ints("output_padding", {});
                op->padding = padding_of(n, "ConvTranspose");
                if (n.attr("output_shape")) throw GraphError("ConvTranspose " + st.name + ": the output_shape attribute is not supported");
                st.run = [op](Context &c, const InputList &in) { return op->run(c, in); };
            } else if (n.op_type == "MatMulNBits") { // com.microsoft contrib op (onnx_registry reads bits / block_size / accuracy_level)
                auto op = std::make_shared<MatMulNBits>();
                op->bits = (int)n.get_int("bits", 4);
                op->block_size = n.get_int("block_size", 32);
                op->accuracy_level = (int)n.get_int("accuracy_level", 0);
                st.run = [op](Context &c, const InputList &in) { return op->run(c, in); };
            } else if (n.op_type == "ConvInteger") {
                auto op = std::make_shared<ConvInteger>();
                op->conv = conv_attrs(n);
                std::string scale, bias, residual;
                bool relu = false;
                long cast = opt_.fuse ? sole_user(out_name, "Cast") : -1;
                long mul = cast >= 0 && m.nodes[(size_t)cast].get_int("to", 0) == onnx::FLOAT ? sole_user(m.nodes[(size_t)cast].outputs[0], "Mul") : -1;
                if (mul >= 0) {
                    // ConvIntegerToFloat: the Mul's other operand must be a single scale available before the conv runs
                    const std::string sc = other_input(m.nodes[(size_t)mul], m.nodes[(size_t)cast].outputs[0]);
                    {
                        scale = sc;
                        dead[(size_t)cast] = dead[(size_t)mul] = true;
                        fused_away_ += 2;
                        out_name = m.nodes[(size_t)mul].outputs[0];
                        st.pos = (size_t)mul;
                        long add = sole_user(out_name, "Add");
                        if (add >= 0) { // Add(bias constant [1, O, 1, 1])
                            const std::string b = other_input(m.nodes[(size_t)add], out_name);
                            if (is_const(b)) {
                                const Tensor &bt = consts_.at(ids_.at(b));
                                if (bt.ndim() == 4 && bt.size(0) == 1 && bt.size(2) == 1 && bt.size(3) == 1) {
                                    bias = b; dead[(size_t)add] = true; fused_away_++; out_name = m.nodes[(size_t)add].outputs[0]; st.pos = (size_t)add;
                                }
                            }
                        }
                        long add2 = bias.empty() ? -1 : sole_user(out_name, "Add");
                        if (add2 >= 0) {
                            const std::string other = other_input(m.nodes[(size_t)add2], out_name);
                            if (other != out_name && ready_before(other, i)) { residual = other; dead[(size_t)add2] = true; fused_away_++; out_name = m.nodes[(size_t)add2].outputs[0]; st.pos = (size_t)add2; }
                        }
                        long r = sole_user(out_name, "Relu");
                        if (r >= 0) { relu = true; dead[(size_t)r] = true; fused_away_++; out_name = m.nodes[(size_t)r].outputs[0]; st.pos = (size_t)r; }
                    }
                }
                while (st.in.size() < 4) st.in.push_back(-1);
                // A constant weight zero point that is all zeros (what ort-quantize writes for symmetric weights) is no zero point: dropped at load, the
                // kernel then takes its single-row-term epilogue (integer algebra: the dropped terms are exact zeros).  Measured: 28.1 -> 24.7 us on the
                // dominant launch of the int8 graph -- the whole gap between this executor and the hand-planned runner, which never passed one.
                if (opt_.fuse && st.in[3] >= 0 && consts_.count(st.in[3])) {
                    const Tensor &z = consts_.at(st.in[3]);
                    bool all_zero = z.len() > 0 && (z.dtype() == DType::I8 || z.dtype() == DType::U8);
                    if (all_zero) for (uint8_t b : z.to_host<uint8_t>()) if (b) { all_zero = false; break; }
                    if (all_zero) st.in[3] = -1;
                }
                st.in.push_back(scale.empty() ? -1 : id_of(scale));
                st.in.push_back(bias.empty() ? -1 : id_of(bias));
                st.in.push_back(residual.empty() ? -1 : id_of(residual));
                if (!scale.empty()) st.kind_name = std::string("ConvIntegerToFloat") + (bias.empty() ? "" : "+bias") + (residual.empty() ? "" : "+Add") + (relu ? "+Relu" : "");
                auto state = std::make_shared<I8Conv>();
                state->op = op;
                state->to_float = !scale.empty();
                if (opt_.prepack && is_const(n.inputs.at(1)) && op->conv.groups == 1) {
                    const Tensor &w = consts_.at(ids_.at(n.inputs[1]));
                    if (w.ndim() == 4) state->sg.packed_weight = packed_for(n.name.empty() ? n.outputs.at(0) : n.name, [&] { return op->prepack(ctx_, w); });
                }
                st.i8 = state;
                // ConvIntegerToFloatFusion (fusions.rs:1012-1058) fuses only a scale of shape [] or [1] and leaves every other graph
                // as ConvInteger -> Cast -> Mul.  Shapes are run-time facts here, so the step decides per run: a scalar f32 scale
                // (or a per-output-channel [1,O,1,1] / [O,1,1] one, which the kernel applies exactly as Cast -> Mul would) with a
                // bias of O channels and a residual of the output's shape takes the fused kernel; anything else runs the
                // operators as the graph spells them.
                state->relu = relu;
                st.run = [state, relu](Context &c, const InputList &in) {
                    const Tensor *scale = in[4], *bias = in[5], *residual = in[6];
                    if (!scale) return state->op->run_staged(c, InputList(in.begin(), in.begin() + 4), nullptr, nullptr, nullptr, false, state->sg);
                    bool per_channel = false;
                    const bool fused = i8_fused_form(*state, in, per_channel);
                    if (fused) return state->op->run_staged(c, InputList(in.begin(), in.begin() + 4), scale, bias, residual, relu, state->sg, per_channel);
                    ConvInteger::Staging sg = state->sg;
                    sg.stats_out = nullptr; // statistics are accumulated by the float epilogue only
                    OutputList acc = state->op->run_staged(c, InputList(in.begin(), in.begin() + 4), nullptr, nullptr, nullptr, false, sg);
                    Cast cast; cast.to = DType::F32;
                    OutputList f = cast.run(c, {&acc[0]});
                    OutputList y = Mul().run(c, {&f[0], scale});
                    if (bias) y = Add().run(c, {&y[0], bias});
                    if (residual) y = Add().run(c, {&y[0], residual});
                    if (relu) y = Relu().run(c, {&y[0]});
                    return y;
                };
            } else if (n.op_type == "MatMulInteger") {
                auto op = std::make_shared<MatMulInteger>();
                std::string scale;
                long cast = opt_.fuse ? sole_user(out_name, "Cast") : -1;
                long mul = cast >= 0 && m.nodes[(size_t)cast].get_int("to", 0) == onnx::FLOAT ? sole_user(m.nodes[(size_t)cast].outputs[0], "Mul") : -1;
                if (mul >= 0) {
                    const std::string sc = other_input(m.nodes[(size_t)mul], m.nodes[(size_t)cast].outputs[0]);
                    scale = sc; dead[(size_t)cast] = dead[(size_t)mul] = true; fused_away_ += 2; out_name = m.nodes[(size_t)mul].outputs[0]; st.pos = (size_t)mul;
                }
                while (st.in.size() < 4) st.in.push_back(-1);
                if (opt_.fuse && st.in[3] >= 0 && consts_.count(st.in[3])) { // an all-zero constant RHS zero point: dropped (see ConvInteger)
                    const Tensor &z = consts_.at(st.in[3]);
                    bool all_zero = z.len() > 0 && (z.dtype() == DType::I8 || z.dtype() == DType::U8);
                    if (all_zero) for (uint8_t b : z.to_host<uint8_t>()) if (b) { all_zero = false; break; }
                    if (all_zero) st.in[3] = -1;
                }
                st.in.push_back(scale.empty() ? -1 : id_of(scale));
                if (!scale.empty()) st.kind_name = "MatMulIntegerToFloat";
                // Graph::prepack_weights (src/graph.rs:488-562): a constant RHS is staged once at load (PackedBMatrix)
                const Tensor *packed = nullptr;
                if (opt_.prepack && is_const(n.inputs.at(1))) {
                    const Tensor &w = consts_.at(ids_.at(n.inputs.at(1)));
                    packed = packed_for(n.name.empty() ? n.outputs.at(0) : n.name, [&] { return op->prepack(ctx_, w); });
                }
                // MatMulIntegerToFloatFusion (fusions.rs:960-1009) needs a scale of rank <= 1; the operator then needs length 1 or N.
                // Decided per run (shapes are run-time facts); otherwise MatMulInteger -> Cast -> Mul as the graph spells it.
                st.run = [op, packed](Context &c, const InputList &in) {
                    const Tensor *scale = in[4];
                    if (scale) {
                        const Tensor &b = require(in, 1);
                        const int64_t ncols = b.ndim() > 1 ? b.size(b.ndim() - 1) : 1;
                        const bool ok = scale->dtype() == DType::F32 && scale->ndim() <= 1 && (scale->len() == 1 || scale->len() == ncols);
                        if (!ok) {
                            OutputList acc = op->run_scaled(c, InputList(in.begin(), in.begin() + 4), nullptr, packed);
                            Cast cast; cast.to = DType::F32;
                            OutputList f = cast.run(c, {&acc[0]});
                            return Mul().run(c, {&f[0], scale});
                        }
                    }
                    return op->run_scaled(c, InputList(in.begin(), in.begin() + 4), scale, packed);
                };
            } else if (n.op_type == "Gemm") {
                auto op = std::make_shared<Gemm>();
                op->alpha = n.get_float("alpha", 1.f); op->beta = n.get_float("beta", 1.f);
                op->transpose_a = n.get_int("transA", 0) != 0; op->transpose_b = n.get_int("transB", 0) != 0;
                auto plan = std::make_shared<GemmPlan>();
                st.gemm_plan = plan;
                st.run = [op, plan](Context &c, const InputList &in) { PlanScope scope(c, *plan); return op->run(c, in); };
            } else if (n.op_type == "MatMul") {
                // MatMul (+ Mul / Div by a constant scalar = alpha, MatMulScale fusion) (+ Add of a constant 1-D bias,
                // MatMulAddFusion) (+ Gelu / Relu as the GEMM epilogue): FusedMatMul (src/ops/matmul.rs:455-510)
                auto op = std::make_shared<FusedMatMul>();
                std::string bias;
                auto const_scalar = [&](const std::string &v, float &out) {
                    if (!is_const(v)) return false;
                    const Tensor &t = consts_.at(ids_.at(v));
                    if (t.len() != 1 || t.dtype() != DType::F32) return false;
                    out = t.to_host<float>()[0];
                    return true;
                };
                if (opt_.fuse) {
                    for (const char *sop : {"Div", "Mul"}) {
                        long d = sole_user(out_name, sop);
                        float c = 0.f;
                        if (d >= 0 && m.nodes[(size_t)d].inputs[0] == out_name && const_scalar(m.nodes[(size_t)d].inputs[1], c) && op->alpha == 1.f) {
                            op->alpha = std::string(sop) == "Div" ? 1.0f / c : c;
                            dead[(size_t)d] = true; fused_away_++; out_name = m.nodes[(size_t)d].outputs[0]; st.pos = (size_t)d;
                        }
                    }
                    long a = op->alpha == 1.f ? sole_user(out_name, "Add") : -1;
                    if (a >= 0) {
                        const std::string other = other_input(m.nodes[(size_t)a], out_name);
                        if (is_const(other) && consts_.at(ids_.at(other)).ndim() == 1 && consts_.at(ids_.at(other)).dtype() == DType::F32) {
                            bias = other; dead[(size_t)a] = true; fused_away_++; out_name = m.nodes[(size_t)a].outputs[0]; st.pos = (size_t)a;
                            long g = sole_user(out_name, "Gelu");
                            if (g >= 0 && !m.nodes[(size_t)g].attr("approximate")) { op->act = RTEN_HIP_ACT_GELU; dead[(size_t)g] = true; fused_away_++; out_name = m.nodes[(size_t)g].outputs[0]; st.pos = (size_t)g; }
                            long r = g < 0 ? sole_user(out_name, "Relu") : -1;
                            if (r >= 0) { op->act = RTEN_HIP_ACT_RELU; dead[(size_t)r] = true; fused_away_++; out_name = m.nodes[(size_t)r].outputs[0]; st.pos = (size_t)r; }
                        }
                    }
                }
                st.in.resize(2);
                st.in.push_back(bias.empty() ? -1 : id_of(bias));
                st.kind_name = std::string(op->alpha != 1.f || !bias.empty() ? "FusedMatMul" : "MatMul") + (op->act == RTEN_HIP_ACT_GELU ? "+Gelu" : op->act == RTEN_HIP_ACT_RELU ? "+Relu" : "");
                auto plan = std::make_shared<GemmPlan>();
                st.gemm_plan = plan;
                st.run = [op, plan](Context &c, const InputList &in) {
                    const Tensor *bias = in[2];
                    if (bias && bias->len() != require(in, 1).size(require(in, 1).ndim() - 1)) throw OpError(OpError::IncompatibleInputShapes, "Cannot broadcast bias to output shape");
                    PlanScope scope(c, *plan);
                    return op->run(c, in);
                };
            } else if (n.op_type == "Add" && opt_.fuse && sole_user(out_name, "LayerNormalization") >= 0 &&
                       m.nodes[(size_t)sole_user(out_name, "LayerNormalization")].get_int("axis", -1) == -1 &&
                       m.nodes[(size_t)sole_user(out_name, "LayerNormalization")].inputs[0] == out_name) {
                // Add(residual) -> LayerNormalization(last axis) as one kernel
                const long ln = sole_user(out_name, "LayerNormalization");
                const onnx::Node &lnn = m.nodes[(size_t)ln];
                auto fused = std::make_shared<AddLayerNormalization>();
                fused->epsilon = lnn.get_float("epsilon", 1e-5f);
                auto plain = std::make_shared<LayerNormalization>();
                plain->epsilon = fused->epsilon;
                dead[(size_t)ln] = true; fused_away_++; out_name = lnn.outputs[0]; st.pos = (size_t)ln;
                st.in.push_back(id_of(lnn.inputs.at(1)));
                st.in.push_back(lnn.inputs.size() > 2 ? id_of(lnn.inputs[2]) : -1);
                st.kind_name = "Add+LayerNormalization";
                st.run = [fused, plain](Context &c, const InputList &in) {
                    if (require(in, 0).shape() == require(in, 1).shape()) return fused->run(c, in);
                    OutputList sum = Add().run(c, {in[0], in[1]}); // broadcasting Add: separate kernels
                    return plain->run(c, {&sum[0], in[2], in[3]});
                };
            } else if (n.op_type == "Add" && opt_.fuse && sole_user(out_name, "Softmax") >= 0 &&
                       m.nodes[(size_t)sole_user(out_name, "Softmax")].get_int("axis", -1) == -1) {
                // Add -> Softmax(last axis) = AddSoftmax (src/ops/attention.rs:94-156)
                const long sm = sole_user(out_name, "Softmax");
                dead[(size_t)sm] = true; fused_away_++; out_name = m.nodes[(size_t)sm].outputs[0]; st.pos = (size_t)sm;
                st.kind_name = "AddSoftmax";
                st.run = [](Context &c, const InputList &in) {
                    const Tensor &a = require(in, 0), &b = require(in, 1);
                    AddSoftmax fused;
                    try {
                        return a.len() >= b.len() ? fused.run(c, {&a, &b}) : fused.run(c, {&b, &a});
                    } catch (const OpError &e) { // a broadcast the fused kernel does not cover: Add, then Softmax
                        if (e.kind != OpError::IncompatibleInputShapes) throw;
                        OutputList sum = Add().run(c, in);
                        return Softmax().run(c, {&sum[0]});
                    }
                };
            } else if (n.op_type == "LayerNormalization") {
                auto op = std::make_shared<LayerNormalization>();
                op->axis = (int)n.get_int("axis", -1);
                op->epsilon = n.get_float("epsilon", 1e-5f);
                st.batch_coupled = op->axis == 0;
                const std::string nm = st.name;
                st.run = [this, op, nm](Context &c, const InputList &in) { note_axis("LayerNormalization", nm, op->axis, require(in, 0)); return op->run(c, in); };
            } else if (n.op_type == "Gelu" && n.attr("approximate") && n.attr("approximate")->s != "none") {
                throw GraphError("Gelu " + st.name + ": approximate=\"" + n.attr("approximate")->s + "\" is not supported");
            } else if (n.op_type == "MaxPool" || n.op_type == "AveragePool") {
                std::vector<int> k = n.get_ints("kernel_shape", {});
                if (k.size() != 2) throw GraphError(n.op_type + " " + st.name + ": kernel_shape must have 2 values");
                const std::vector<int> strides = n.get_ints("strides", {1, 1});
                const Padding pad = padding_of(n, n.op_type.c_str());
                const bool ceil = n.get_int("ceil_mode", 0) != 0;
                if (n.outputs.size() > 1 && !n.outputs[1].empty()) throw GraphError("MaxPool " + st.name + ": the Indices output is not supported");
                if (n.op_type == "MaxPool") {
                    auto op = std::make_shared<MaxPool>();
                    op->kernel_size = k; op->strides = strides; op->padding = pad; op->ceil_mode = ceil;
                    st.run = [op](Context &c, const InputList &in) { return op->run(c, in); };
                    st.maxpool = op;
                } else {
                    auto op = std::make_shared<AveragePool>();
                    op->kernel_size = k; op->strides = strides; op->padding = pad; op->ceil_mode = ceil;
                    op->count_include_pad = n.get_int("count_include_pad", 0) != 0;
                    st.run = [op](Context &c, const InputList &in) { return op->run(c, in); };
                }
            } else if (n.op_type == "Einsum") {
                auto op = std::make_shared<Einsum>();
                if (!n.attr("equation")) throw GraphError("Einsum " + st.name + ": the equation attribute is missing");
                op->equation = n.attr("equation")->s;
                st.run = [op](Context &c, const InputList &in) { return op->run(c, in); };
            } else if (n.op_type == "ReduceSum" || n.op_type == "ReduceMean") {
                auto op = std::make_shared<ReduceSum>();
                op->mean = n.op_type == "ReduceMean";
                op->keep_dims = n.get_int("keepdims", 1) != 0;
                op->noop_with_empty_axes = n.get_int("noop_with_empty_axes", 0) != 0;
                op->axes = n.get_ints("axes", {});
                if (n.inputs.size() > 1 && !n.inputs[1].empty()) { // opset >= 13: axes as a (constant) input
                    auto it = ids_.find(n.inputs[1]);
                    if (it == ids_.end() || !consts_.count(it->second)) throw GraphError(n.op_type + " " + st.name + ": the axes input must be a constant");
                    for (int32_t v : consts_.at(it->second).to_host<int32_t>()) op->axes.push_back(v);
                    st.in.resize(1);
                }
                st.batch_coupled = op->axes.empty() ? !op->noop_with_empty_axes : std::count(op->axes.begin(), op->axes.end(), 0) != 0; // sums over dim 0
                const std::string kind = n.op_type, nm = st.name;
                st.run = [this, op, kind, nm](Context &c, const InputList &in) { for (int a : op->axes) note_axis(kind, nm, a, require(in, 0)); return op->run(c, in); };
            } else if (n.op_type == "Softmax") {
                auto op = std::make_shared<Softmax>();
                op->axis = (int)n.get_int("axis", -1);
                st.batch_coupled = op->axis == 0;
                const std::string nm = st.name;
                st.run = [this, op, nm](Context &c, const InputList &in) { note_axis("Softmax", nm, op->axis, require(in, 0)); return op->run(c, in); };
            } else if (n.op_type == "Flatten" || n.op_type == "Reshape" || n.op_type == "Squeeze" || n.op_type == "Unsqueeze" || n.op_type == "Identity" ||
                       n.op_type == "Dropout") {
                make_view_step(st, n, m);
            } else if (make_layout_step(st, n)) {
                // Shape / ConstantOfShape / NonZero / Range / Slice / Concat / Expand / Where / comparisons / logic / integer arithmetic / Cast / Gather /
                // Transpose: host-evaluated when their operands are host values, device kernels otherwise
            } else {
                static const OpRegistry reg = OpRegistry::with_all_ops();
                if (!reg.contains(n.op_type))
                    throw GraphError("node " + st.name + ": operator " + n.op_type + " is not available on the HIP backend (no CPU fallback)");
                std::shared_ptr<Operator> op(reg.create(n.op_type).release());
                st.run = [op](Context &c, const InputList &in) { return op->run(c, in); };
            }
            st.out.push_back(id_of(out_name));
            for (size_t k = 1; k < n.outputs.size(); k++) st.out.push_back(n.outputs[k].empty() ? -1 : id_of(n.outputs[k]));
            steps_.push_back(std::move(st));
        }
        for (auto &o : m.outputs)
            if (!ids_.count(o.name)) throw GraphError("graph output " + o.name + " is not produced by any node");
        std::stable_sort(steps_.begin(), steps_.end(), [](const Step &a, const Step &b) { return a.pos < b.pos; });
        for (auto &st : steps_) if (st.view && st.out[0] >= 0) view_values_.insert(st.out[0]);
        if (opt_.fuse) { plan_int8_staging(); plan_dql_loaders(); plan_conv_pairs(); }
        for (auto &st : steps_) if (st.kind_name.find("DynamicQuantizeLinear") != std::string::npos) st.batch_coupled = true; // min / max over the whole tensor
        plan_liveness();
    }

    // Layout / logic operators around the hot path (src/ops/layout.rs, slice.rs, concat.rs, gather.rs, convert.rs, binary_elementwise.rs, non_zero.rs,
    // generate.rs): each runs on the host when its operands are host values and as a device kernel otherwise.  Returns false for other operator types.
    bool make_layout_step(Step &st, const onnx::Node &n) {
        const std::string op = n.op_type, name = st.name;
        static const std::map<std::string, int> ew = {{"And", RTEN_HIP_EW_AND}, {"Or", RTEN_HIP_EW_OR}, {"Xor", RTEN_HIP_EW_XOR}, {"Equal", RTEN_HIP_EW_EQUAL},
                                                      {"Less", RTEN_HIP_EW_LESS}, {"LessOrEqual", RTEN_HIP_EW_LESS_EQ}, {"Greater", RTEN_HIP_EW_GREATER},
                                                      {"GreaterOrEqual", RTEN_HIP_EW_GREATER_EQ}};
        static const std::map<std::string, int> arith = {{"Add", RTEN_HIP_EW_IADD}, {"Sub", RTEN_HIP_EW_ISUB}, {"Mul", RTEN_HIP_EW_IMUL}, {"Div", RTEN_HIP_EW_IDIV}};
        if (op == "Shape") { // src/ops/layout.rs:475-520 (start / end attributes of opset 15)
            const int64_t start = n.get_int("start", 0);
            const bool has_end = n.attr("end") != nullptr;
            const int64_t end = n.get_int("end", 0);
            auto cache = std::make_shared<HostCache>();
            st.run = [start, has_end, end, cache](Context &c, const InputList &in) {
                const Tensor &x = require(in, 0);
                const int64_t nd = x.ndim();
                int64_t s = start < 0 ? start + nd : start, e = has_end ? (end < 0 ? end + nd : end) : nd;
                s = std::min(std::max<int64_t>(s, 0), nd); e = std::min(std::max<int64_t>(e, 0), nd);
                std::vector<int64_t> dims;
                for (int64_t d = s; d < e; d++) dims.push_back(x.size((int)d));
                OutputList o;
                o.push_back(materialize(c, hostops::make_ints({(int64_t)dims.size()}, dims), *cache));
                return o;
            };
            return true;
        }
        if (ew.count(op) || arith.count(op)) {
            const bool is_arith = arith.count(op) != 0;
            const int code = is_arith ? arith.at(op) : ew.at(op);
            std::shared_ptr<Operator> fop;
            if (is_arith) { static const OpRegistry reg = OpRegistry::with_all_ops(); fop.reset(reg.create(op).release()); }
            auto dev = std::make_shared<ElementwiseNd>(code, is_arith ? "IntegerArithmetic" : "Logical");
            make_hostable(st,
                          [op](const std::vector<const HostVal *> &v, HostVal &out) { return v.size() == 2 && v[0] && v[1] && hostops::binary(op, *v[0], *v[1], out); },
                          [dev, fop, is_arith](Context &c, const InputList &in) {
                              if (is_arith && !(require(in, 0).dtype() == DType::I32 && require(in, 1).dtype() == DType::I32)) return fop->run(c, in); // the f32 kernels
                              if (is_arith && dev->code == RTEN_HIP_EW_IDIV && in[1]->host())
                                  for (int64_t d : in[1]->host()->i) if (d == 0) throw OpError(OpError::InvalidValue, "integer division by zero");
                              return dev->run(c, in);
                          });
            return true;
        }
        if (op == "Not") {
            auto dev = std::make_shared<ElementwiseNd>(RTEN_HIP_EW_NOT, "Not");
            make_hostable(st,
                          [](const std::vector<const HostVal *> &v, HostVal &out) {
                              if (v.size() != 1 || !v[0] || v[0]->is_float) return false;
                              out.shape = v[0]->shape;
                              for (int64_t x : v[0]->i) out.i.push_back(x == 0);
                              return true;
                          },
                          [dev](Context &c, const InputList &in) { return dev->run(c, in); });
            return true;
        }
        if (op == "Where") {
            auto dev = std::make_shared<Where>();
            make_hostable(st, [](const std::vector<const HostVal *> &v, HostVal &out) { return v.size() == 3 && v[0] && v[1] && v[2] && hostops::where(*v[0], *v[1], *v[2], out); },
                          [dev](Context &c, const InputList &in) { return dev->run(c, in); });
            return true;
        }
        if (op == "Cast") {
            auto dev = std::make_shared<Cast>();
            dev->to = dtype_of((int)n.get_int("to", onnx::FLOAT), name);
            const DType to = dev->to;
            make_hostable(st, [to](const std::vector<const HostVal *> &v, HostVal &out) { if (v.size() != 1 || !v[0] || (to != DType::F32 && to != DType::I32)) return false; out = hostops::cast(*v[0], to); return true; },
                          [dev](Context &c, const InputList &in) { return dev->run(c, in); });
            return true;
        }
        if (op == "Transpose") {
            auto dev = std::make_shared<Transpose>();
            dev->perm = n.get_ints("perm", {});
            const std::vector<int> perm = dev->perm;
            make_hostable(st, [perm](const std::vector<const HostVal *> &v, HostVal &out) { if (v.size() != 1 || !v[0]) return false; out = hostops::transpose(*v[0], perm); return true; },
                          [this, dev, name](Context &c, const InputList &in) {
                              const int nd = require(in, 0).ndim(); // a device transpose that moves dim 0 mixes the rows sub-batch chains would split
                              const int p0 = dev->perm.empty() ? nd - 1 : (dev->perm[0] < 0 ? dev->perm[0] + nd : dev->perm[0]);
                              if (nd > 1 && p0 != 0 && runtime_coupled_.empty()) runtime_coupled_ = "Transpose \"" + name + "\" (moves dim 0)";
                              return dev->run(c, in);
                          });
            return true;
        }
        if (op == "Gather") {
            auto dev = std::make_shared<Gather>();
            dev->axis = (int)n.get_int("axis", 0);
            const int axis = dev->axis;
            make_hostable(st, [axis](const std::vector<const HostVal *> &v, HostVal &out) { if (v.size() != 2 || !v[0] || !v[1] || v[1]->is_float) return false; out = hostops::gather(*v[0], *v[1], axis); return true; },
                          [dev](Context &c, const InputList &in) { return dev->run(c, in); });
            return true;
        }
        if (op == "Concat") {
            if (!n.attr("axis")) throw GraphError("Concat " + name + ": the axis attribute is missing");
            const int axis = (int)n.get_int("axis", 0);
            make_hostable(st,
                          [axis](const std::vector<const HostVal *> &v, HostVal &out) { for (auto *x : v) if (!x) return false; if (v.empty()) return false; out = hostops::concat(v, axis); return true; },
                          [axis](Context &c, const InputList &in) { OutputList o; o.push_back(concat_tensors(c, in, axis)); return o; });
            return true;
        }
        if (op == "Slice") { // opset >= 10: starts, ends, axes, steps as inputs (host values); opset 1: attributes
            std::vector<int64_t> a_starts, a_ends, a_axes;
            if (const onnx::Attr *a = n.attr("starts")) a_starts = a->ints;
            if (const onnx::Attr *a = n.attr("ends")) a_ends = a->ints;
            if (const onnx::Attr *a = n.attr("axes")) a_axes = a->ints;
            const bool attr_form = n.attr("starts") != nullptr;
            auto cache = std::make_shared<HostCache>();
            st.run = [attr_form, a_starts, a_ends, a_axes, cache](Context &c, const InputList &in) {
                const Tensor &x = require(in, 0);
                const std::vector<int64_t> starts = attr_form ? a_starts : host_ints(in.size() > 1 ? in[1] : nullptr, "Slice: starts"),
                                           ends = attr_form ? a_ends : host_ints(in.size() > 2 ? in[2] : nullptr, "Slice: ends"),
                                           axes = attr_form ? a_axes : host_ints(in.size() > 3 ? in[3] : nullptr, "Slice: axes"),
                                           steps = attr_form ? std::vector<int64_t>{} : host_ints(in.size() > 4 ? in[4] : nullptr, "Slice: steps");
                const std::vector<SliceRange> r = resolve_slice(x.shape(), starts, ends, axes, steps);
                OutputList o;
                if (x.host()) o.push_back(materialize(c, hostops::slice(*x.host(), r), *cache));
                else o.push_back(slice_tensor(c, x, r));
                return o;
            };
            return true;
        }
        if (op == "Expand") {
            auto cache = std::make_shared<HostCache>();
            st.run = [cache](Context &c, const InputList &in) {
                const Tensor &x = require(in, 0);
                const std::vector<int64_t> target = host_ints(&require(in, 1), "Expand: the shape input");
                OutputList o;
                if (x.host() && detail::prod(target, 0, target.size()) <= hostops::kMaxHostElems) o.push_back(materialize(c, hostops::expand(*x.host(), target), *cache));
                else o.push_back(expand_to(c, x, target));
                return o;
            };
            return true;
        }
        if (op == "ConstantOfShape") { // src/ops/generate.rs: a tensor of `shape` filled with the `value` attribute (default float 0)
            HostVal fill;
            fill.is_float = true; fill.f = {0.f};
            if (const onnx::Attr *v = n.attr("value")) {
                const onnx::TensorProto &t = v->t;
                if (t.len() != 1) throw GraphError("ConstantOfShape " + name + ": value must hold one element");
                if (t.data_type == onnx::FLOAT && t.raw.size() == 4) std::memcpy(&fill.f[0], t.raw.data(), 4);
                else if (t.data_type == onnx::INT64 && t.raw.size() == 8) { int64_t x; std::memcpy(&x, t.raw.data(), 8); fill.is_float = false; fill.f.clear(); fill.i = {hostops::wrap32(x)}; }
                else if (t.data_type == onnx::INT32 && t.raw.size() == 4) { int32_t x; std::memcpy(&x, t.raw.data(), 4); fill.is_float = false; fill.f.clear(); fill.i = {x}; }
                else if (t.data_type == onnx::BOOL && t.raw.size() == 1) { fill.is_float = false; fill.f.clear(); fill.i = {t.raw[0] != 0}; }
                else throw GraphError("ConstantOfShape " + name + ": unsupported value type");
            }
            auto cache = std::make_shared<HostCache>();
            st.run = [fill, cache](Context &c, const InputList &in) {
                const std::vector<int64_t> shape = host_ints(&require(in, 0), "ConstantOfShape: the shape input");
                for (int64_t d : shape) if (d < 0) throw OpError(OpError::InvalidValue, "ConstantOfShape: negative dimension");
                OutputList o;
                if (detail::prod(shape, 0, shape.size()) <= hostops::kMaxHostElems) { o.push_back(materialize(c, hostops::expand(fill, shape), *cache)); return o; }
                Tensor one = materialize(c, HostVal(fill), *cache); // a large fill: expand the one-element device constant
                o.push_back(expand_to(c, one, shape));
                return o;
            };
            return true;
        }
        if (op == "NonZero") { // the output's shape depends on the input's VALUES: only host values (exporters write arange(n) as NonZero(ConstantOfShape(n)))
            auto cache = std::make_shared<HostCache>();
            st.run = [cache](Context &c, const InputList &in) {
                const Tensor &x = require(in, 0);
                if (!x.host()) throw OpError(OpError::UnsupportedValue, "NonZero of device data: the output shape would depend on values (only shape-derived operands are supported)");
                OutputList o;
                o.push_back(materialize(c, hostops::nonzero(*x.host()), *cache));
                return o;
            };
            return true;
        }
        if (op == "Range") { // src/ops/generate.rs: start, limit, delta scalars (host values)
            auto cache = std::make_shared<HostCache>();
            st.run = [cache](Context &c, const InputList &in) {
                const Tensor &a = require(in, 0), &b = require(in, 1), &d = require(in, 2);
                if (!a.host() || !b.host() || !d.host()) throw OpError(OpError::UnsupportedValue, "Range: start / limit / delta must be constants or computable from the input shapes");
                HostVal out;
                if (a.host()->is_float) {
                    const float s = a.host()->f.at(0), l = b.host()->f.at(0), dl = d.host()->f.at(0);
                    if (dl == 0.f) throw OpError(OpError::InvalidValue, "delta must be non-zero");
                    out.is_float = true;
                    const int64_t cnt = std::max<int64_t>((int64_t)std::ceil((l - s) / dl), 0);
                    for (int64_t k = 0; k < cnt; k++) out.f.push_back(s + (float)k * dl);
                    out.shape = {cnt};
                } else {
                    const int64_t s = a.host()->i.at(0), l = b.host()->i.at(0), dl = d.host()->i.at(0);
                    if (dl == 0) throw OpError(OpError::InvalidValue, "delta must be non-zero");
                    for (int64_t v = s; dl > 0 ? v < l : v > l; v += dl) out.i.push_back(v);
                    out.shape = {(int64_t)out.i.size()};
                }
                OutputList o;
                o.push_back(materialize(c, std::move(out), *cache));
                return o;
            };
            return true;
        }
        return false;
    }

    // Shape-only operators: the output aliases the input's buffer (src/ops/layout.rs reshapes in place when it can).
    void make_view_step(Step &st, const onnx::Node &n, const onnx::Model &m) {
        (void)m;
        const std::string kind = n.op_type;
        const int axis = (int)n.get_int("axis", 1);
        std::vector<int64_t> spec; // Reshape target / (Un)Squeeze axes from a constant input (opset >= 13) or the attribute
        bool have_spec = false;
        bool runtime_spec = false; // the shape / axes operand is computed by the graph (shape arithmetic): read from its host value at run time
        if (n.inputs.size() > 1 && !n.inputs[1].empty()) {
            auto it = ids_.find(n.inputs[1]);
            if (it == ids_.end() || !consts_.count(it->second)) {
                runtime_spec = true;
                st.in.resize(2);
            } else {
                const Tensor &t = consts_.at(it->second);
                for (int32_t v : t.to_host<int32_t>()) spec.push_back(v);
                st.in.resize(1);
            }
            have_spec = true;
        } else if (const onnx::Attr *a = n.attr("axes")) { spec = a->ints; have_spec = true; }
        else if (const onnx::Attr *a2 = n.attr("shape")) { spec = a2->ints; have_spec = true; }
        const bool allowzero = n.get_int("allowzero", 0) != 0;
        st.kind_name = kind + "(view)";
        const std::string step_name = st.name;
        st.run = [kind, axis, spec_const = spec, have_spec, allowzero, runtime_spec, step_name](Context &, const InputList &in) {
            const Tensor &x = require(in, 0);
            const std::vector<int64_t> spec_rt = runtime_spec ? host_ints(&require(in, 1), kind + " " + step_name + ": the shape / axes input") : std::vector<int64_t>();
            const std::vector<int64_t> &spec = runtime_spec ? spec_rt : spec_const;
            std::vector<int64_t> s = x.shape();
            const int nd = (int)s.size();
            if (kind == "Flatten") {
                const int a = axis < 0 ? axis + nd : axis;
                if (a < 0 || a > nd) throw OpError(OpError::InvalidValue, "Axis is invalid");
                s = {detail::prod(x.shape(), 0, (size_t)a), detail::prod(x.shape(), (size_t)a, (size_t)nd)};
            } else if (kind == "Reshape") {
                if (!have_spec) throw OpError(OpError::MissingInputs, "");
                s.assign(spec.begin(), spec.end());
                int64_t known = 1;
                int infer = -1;
                for (size_t i = 0; i < s.size(); i++) {
                    if (s[i] == 0 && !allowzero) { if (i >= (size_t)nd) throw OpError(OpError::InvalidValue, "Input and output element counts do not match"); s[i] = x.size((int)i); }
                    if (s[i] == -1) { if (infer >= 0) throw OpError(OpError::InvalidValue, "Multiple dimensions in new shape set to -1"); infer = (int)i; }
                    else known *= s[i];
                }
                if (infer >= 0) {
                    if (known == 0 || x.len() % known != 0) throw OpError(OpError::InvalidValue, "Input length must be a multiple of specified dimensions");
                    s[(size_t)infer] = x.len() / known;
                }
                if (detail::prod(s, 0, s.size()) != x.len()) throw OpError(OpError::InvalidValue, "Input and output element counts do not match");
            } else if (kind == "Squeeze") {
                std::vector<int64_t> o;
                for (int i = 0; i < nd; i++) {
                    bool drop = have_spec ? false : s[(size_t)i] == 1;
                    for (int64_t a : spec) if ((a < 0 ? a + nd : a) == i) drop = true;
                    if (drop && s[(size_t)i] != 1) throw OpError(OpError::InvalidValue, "Can only remove dimensions of size 1");
                    if (!drop) o.push_back(s[(size_t)i]);
                }
                s = o;
            } else if (kind == "Unsqueeze") {
                if (!have_spec) throw OpError(OpError::MissingInputs, "");
                const int out_nd = nd + (int)spec.size();
                std::vector<int64_t> o((size_t)out_nd, 0);
                for (int64_t a : spec) {
                    const int64_t p = a < 0 ? a + out_nd : a;
                    if (p < 0 || p >= out_nd || o[(size_t)p] == 1) throw OpError(OpError::InvalidValue, "Axes must be unique and in range");
                    o[(size_t)p] = 1;
                }
                int src = 0;
                for (auto &d : o) if (d == 0) d = s[(size_t)src++];
                s = o;
            }
            OutputList out;
            out.push_back(Tensor::view_of(x, s));
            if (x.host()) out.back().set_host(std::make_shared<const HostVal>(hostops::reshaped(*x.host(), s))); // the same values under the new shape
            return out;
        };
        st.view = true;
    }

    // Reference counts per value (Graph::run_plan's temp value refcounts): a value's buffer returns to the pool after its
    // last consumer; a view adds one use to its base that is released together with the view itself.
    void plan_liveness() {
        uses_.assign(names_.size(), 0);
        std::vector<int> base_of(names_.size(), -1);
        for (auto &o : outputs_) uses_[(size_t)ids_.at(o.name)] += 1 << 20; // never released
        for (auto &st : steps_) {
            for (int id : st.in) if (id >= 0) uses_[(size_t)id]++;
            if (st.view && st.in[0] >= 0 && st.out[0] >= 0) {
                int base = st.in[0];
                while (base_of[(size_t)base] >= 0) base = base_of[(size_t)base];
                base_of[(size_t)st.out[0]] = base;
                uses_[(size_t)base]++; // held by the view
            }
        }
        std::vector<int> left(uses_);
        for (auto &st : steps_) {
            for (int id : st.in) {
                if (id < 0 || consts_.count(id)) continue;
                st.release_after.push_back(id);
                if (--left[(size_t)id] == 0 && base_of[(size_t)id] >= 0) { // the view died: drop its hold on the base
                    st.release_after.push_back(base_of[(size_t)id]);
                    left[(size_t)base_of[(size_t)id]]--;
                }
            }
            // outputs nobody reads (e.g. unused secondary outputs) die immediately
            for (int id : st.out) if (id >= 0 && uses_[(size_t)id] == 0) { uses_[(size_t)id] = 1; st.release_after.push_back(id); }
        }
    }
};

} // namespace rten_hip
