// Host-side shape arithmetic of the executor (include/rten_hip_graph.hpp, namespace hostops): the values exporter-written graphs compute from input shapes
// and constants.  Checked against hand-computed / numpy-defined results; needs no GPU.  The cases are the idioms of a transformers-exported BERT:
// arange(n) as NonZero(ConstantOfShape(n)), index arithmetic with broadcasting, comparisons, Where / Equal building an Expand shape, Shape -> Gather ->
// Unsqueeze -> Concat, Slice with clamping and negative steps.
#include <cstdio>

#include "rten_hip_graph.hpp"

using namespace rten_hip;
static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static HostVal ints(std::vector<int64_t> shape, std::vector<int64_t> v) { return hostops::make_ints(std::move(shape), std::move(v)); }
static HostVal floats(std::vector<int64_t> shape, std::vector<float> v) { HostVal h; h.shape = std::move(shape); h.is_float = true; h.f = std::move(v); return h; }

int main() {
    // arange(5) = Cast(Squeeze(Transpose(NonZero(ConstantOfShape([5], value = 1)))))
    HostVal one = ints({1}, {1});
    HostVal ones = hostops::expand(one, {5});
    CHECK(ones.shape == std::vector<int64_t>{5} && ones.i == std::vector<int64_t>({1, 1, 1, 1, 1}));
    HostVal nz = hostops::nonzero(ones);
    CHECK(nz.shape == (std::vector<int64_t>{1, 5}) && nz.i == std::vector<int64_t>({0, 1, 2, 3, 4}));
    HostVal tr = hostops::transpose(nz, {1, 0});
    CHECK(tr.shape == (std::vector<int64_t>{5, 1}) && tr.i == nz.i);
    // NonZero of a 2-D value: indices per axis, row-major order of the hits
    HostVal nz2 = hostops::nonzero(ints({2, 3}, {0, 7, 0, 1, 0, -2}));
    CHECK(nz2.shape == (std::vector<int64_t>{2, 3}) && nz2.i == std::vector<int64_t>({0, 1, 1, 1, 0, 2}));
    // batch_idx [2,1,1,1] * seq + kv_idx [1,1,1,4] -> [2,1,1,4]
    HostVal out;
    CHECK(hostops::binary("Mul", ints({2, 1, 1, 1}, {0, 1}), ints({}, {4}), out) && out.shape == (std::vector<int64_t>{2, 1, 1, 1}) && out.i == std::vector<int64_t>({0, 4}));
    HostVal sum;
    CHECK(hostops::binary("Add", ints({1, 1, 1, 4}, {0, 1, 2, 3}), out, sum) && sum.shape == (std::vector<int64_t>{2, 1, 1, 4}) && sum.i == std::vector<int64_t>({0, 1, 2, 3, 4, 5, 6, 7}));
    // wrapping int32 arithmetic, truncating division
    CHECK(hostops::binary("Add", ints({1}, {2147483647}), ints({1}, {1}), out) && out.i[0] == -2147483648LL);
    CHECK(hostops::binary("Div", ints({2}, {-7, 7}), ints({1}, {2}), out) && out.i == std::vector<int64_t>({-3, 3}));
    bool threw = false;
    try { hostops::binary("Div", ints({1}, {1}), ints({1}, {0}), out); } catch (const OpError &) { threw = true; }
    CHECK(threw);
    // comparisons and logic give 0 / 1
    CHECK(hostops::binary("Less", ints({3}, {-1, 0, 1}), ints({}, {0}), out) && out.i == std::vector<int64_t>({1, 0, 0}) && !out.is_float);
    CHECK(hostops::binary("GreaterOrEqual", floats({2}, {0.5f, -0.5f}), floats({}, {0.f}), out) && out.i == std::vector<int64_t>({1, 0}) && !out.is_float);
    CHECK(hostops::binary("And", ints({2, 1}, {1, 0}), ints({1, 2}, {5, 0}), out) && out.shape == (std::vector<int64_t>{2, 2}) && out.i == std::vector<int64_t>({1, 0, 0, 0}));
    CHECK(!hostops::binary("Add", ints({1}, {1}), floats({1}, {1.f}), out)); // mixed types: declined, the device path reports the type error
    CHECK(hostops::binary("Mul", floats({2}, {1.5f, -2.f}), floats({}, {-1.f}), out) && out.is_float && out.f == std::vector<float>({-1.5f, 2.f}));
    // the Expand-shape idiom: Where(Equal(shape, -1 * ones), ones, shape)
    HostVal shp = ints({4}, {2, -1, 8, 8}), onesv = ints({4}, {1, 1, 1, 1}), neg, eq, w;
    CHECK(hostops::binary("Mul", onesv, ints({}, {-1}), neg) && hostops::binary("Equal", shp, neg, eq) && eq.i == std::vector<int64_t>({0, 1, 0, 0}));
    CHECK(hostops::where(eq, onesv, shp, w) && w.i == std::vector<int64_t>({2, 1, 8, 8}));
    // Shape -> Gather(1) -> Unsqueeze -> Concat with constants
    HostVal dims = ints({2}, {32, 128});
    HostVal g = hostops::gather(dims, ints({}, {1}), 0);
    CHECK(g.shape.empty() && g.i == std::vector<int64_t>({128}));
    HostVal g1 = hostops::reshaped(g, {1});
    const HostVal c0 = ints({1}, {-1}), c2 = ints({1}, {12}), c3 = ints({1}, {64});
    HostVal cat = hostops::concat({&c0, &g1, &c2, &c3}, 0);
    CHECK(cat.shape == std::vector<int64_t>{4} && cat.i == std::vector<int64_t>({-1, 128, 12, 64}));
    // Gather along axis 1 with negative indices; out-of-range is the reference's error
    HostVal g2 = hostops::gather(ints({2, 3}, {1, 2, 3, 4, 5, 6}), ints({2}, {-1, 0}), 1);
    CHECK(g2.shape == (std::vector<int64_t>{2, 2}) && g2.i == std::vector<int64_t>({3, 1, 6, 4}));
    threw = false;
    try { hostops::gather(dims, ints({}, {2}), 0); } catch (const OpError &e) { threw = e.msg == "Entry in indices is out of range"; }
    CHECK(threw);
    // Slice: position ids [1, 512] -> [:, 0:seq]; clamping; negative step
    HostVal pos = ints({1, 6}, {0, 1, 2, 3, 4, 5});
    HostVal s1 = hostops::slice(pos, resolve_slice(pos.shape, {0}, {4}, {1}, {1}));
    CHECK(s1.shape == (std::vector<int64_t>{1, 4}) && s1.i == std::vector<int64_t>({0, 1, 2, 3}));
    HostVal s2 = hostops::slice(pos, resolve_slice(pos.shape, {-2}, {9223372036854775807LL}, {-1}, {}));
    CHECK(s2.shape == (std::vector<int64_t>{1, 2}) && s2.i == std::vector<int64_t>({4, 5}));
    HostVal s3 = hostops::slice(pos, resolve_slice(pos.shape, {-1}, {-9223372036854775807LL}, {1}, {-2}));
    CHECK(s3.shape == (std::vector<int64_t>{1, 3}) && s3.i == std::vector<int64_t>({5, 3, 1}));
    HostVal s4 = hostops::slice(pos, resolve_slice(pos.shape, {4}, {2}, {1}, {1}));
    CHECK(s4.shape == (std::vector<int64_t>{1, 0}) && s4.i.empty());
    // Cast: Rust `as` (float -> int truncates toward zero, saturates, NaN -> 0; int -> float)
    HostVal ci = hostops::cast(floats({5}, {1.9f, -1.9f, 3e10f, -3e10f, std::nanf("")}), DType::I32);
    CHECK(ci.i == std::vector<int64_t>({1, -1, 2147483647, -2147483648LL, 0}) && !ci.is_float);
    HostVal cf = hostops::cast(ints({2}, {3, -4}), DType::F32);
    CHECK(cf.is_float && cf.f == std::vector<float>({3.f, -4.f}));
    // transpose of a 3-D value
    HostVal t3 = hostops::transpose(ints({2, 1, 3}, {1, 2, 3, 4, 5, 6}), {2, 0, 1});
    CHECK(t3.shape == (std::vector<int64_t>{3, 2, 1}) && t3.i == std::vector<int64_t>({1, 4, 2, 5, 3, 6}));
    // broadcasting errors are the reference's
    threw = false;
    try { hostops::binary("Add", ints({2}, {1, 2}), ints({3}, {1, 2, 3}), out); } catch (const OpError &e) { threw = e.kind == OpError::IncompatibleInputShapes; }
    CHECK(threw);
    // HostVal equality is bitwise on floats (the cache key of a materialised constant)
    CHECK(floats({1}, {0.f}) == floats({1}, {0.f}) && !(floats({1}, {0.f}) == floats({1}, {-0.f})) && !(ints({1}, {0}) == floats({1}, {0.f})));
    // which dims of a value are "uniform" (every slice along the dim equal): what lets an expanded padding mask [B, 1, S, T] be read as one row per batch item
    CHECK(host_uniform_along(ints({1, 1, 3, 1}, {1, 1, 1}), 2) && !host_uniform_along(ints({1, 1, 3, 1}, {1, 0, 1}), 2));
    CHECK(host_uniform_along(ints({2, 3}, {5, 5, 5, 7, 7, 7}), 1) && !host_uniform_along(ints({2, 3}, {5, 5, 5, 7, 7, 7}), 0));
    CHECK(host_uniform_along(floats({2, 2}, {0.f, 1.f, 0.f, 1.f}), 0) && !host_uniform_along(floats({2, 2}, {0.f, 1.f, -0.f, 1.f}), 0)); // bitwise
    if (failures) { std::printf("%d FAILED\n", failures); return 1; }
    std::printf("ALL OK\n");
    return 0;
}
