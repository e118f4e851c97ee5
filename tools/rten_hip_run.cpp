// rten_hip_run -- run an ONNX model on the HIP backend, in the manner of `rten model.onnx` (rten-cli/src/main.rs).
//
//   rten_hip_run [options] model.onnx
//     -n, --n-iters N        timed runs after one warm-up (default 1); prints "#i - t ms" and mean / min / max / std like
//                            rten-cli/src/main.rs:301-347
//     -s, --size name=value  value of a symbolic input dimension (default 1), e.g. -s batch=32
//     --input name=path      raw little-endian file for an input (otherwise floats are U[0,1) from a fixed-seed generator
//                            and integers are 0, rten-cli/src/input_generator.rs:101-144)
//     -i, --inputs FILE      input tensors from a Safetensors file; tensor names are input names (rten-cli --inputs)
//     --check-outputs FILE   compare the outputs with the same-named tensors of a Safetensors file and print
//                            `Output "name" vs expected: max diff d` like rten-cli/src/main.rs:366-403
//     --max-diff D           with --check-outputs: exit 3 if any max diff exceeds D (or a shape / dtype differs)
//     --save-outputs FILE    write the outputs as a Safetensors file (goldens for `rten --check-outputs`)
//     --dump name=path       write an output tensor as raw bytes
//     --no-fuse              run the graph node by node (no fusion passes)
//     -t, --timing           per-operator table (each operator followed by a sync)
//     --tune                 time the candidate launch plans of every f32 convolution at load and keep the fastest
//     --graph                capture one run into a hipGraph and replay it for the timed runs
//     --parse-only           print the model summary and exit (needs no GPU)
//     --safetensors-info F   list the tensors of a Safetensors file (with --save-outputs: re-write it); no model, no GPU
//
// There is no CPU fallback: without an MI355X the tool reports BackendUnavailable and exits 2.
#include <cinttypes>
#include <cmath>

#include "rten_hip_graph.hpp"
#include "rten_hip_safetensors.hpp"

using namespace rten_hip;

static std::string shape_str(const std::vector<int64_t> &s) {
    std::string o = "[";
    for (size_t i = 0; i < s.size(); i++) o += (i ? ", " : "") + std::to_string(s[i]);
    return o + "]";
}
static const char *type_str(int t) {
    switch (t) { case onnx::FLOAT: return "f32"; case onnx::UINT8: return "u8"; case onnx::INT8: return "i8"; case onnx::INT32: return "i32"; case onnx::INT64: return "i64"; default: return "?"; }
}

int main(int argc, char **argv) {
    std::string path;
    int iters = 1;
    bool fuse = true, timing = false, parse_only = false, tune = false, use_graph = false;
    std::map<std::string, int64_t> sizes;
    std::map<std::string, std::string> input_files, dumps;
    std::string st_inputs, st_check, st_save, st_info;
    double max_diff_allowed = -1.0;
    auto kv = [](const std::string &a, std::string &k, std::string &v) {
        const size_t e = a.find('=');
        if (e == std::string::npos) return false;
        k = a.substr(0, e); v = a.substr(e + 1);
        return true;
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        std::string k, v;
        if ((a == "-n" || a == "--n-iters") && i + 1 < argc) iters = std::atoi(argv[++i]);
        else if ((a == "-s" || a == "--size") && i + 1 < argc && kv(argv[++i], k, v)) sizes[k] = std::atoll(v.c_str());
        else if (a == "--input" && i + 1 < argc && kv(argv[++i], k, v)) input_files[k] = v;
        else if (a == "--dump" && i + 1 < argc && kv(argv[++i], k, v)) dumps[k] = v;
        else if ((a == "-i" || a == "--inputs") && i + 1 < argc) st_inputs = argv[++i];
        else if (a == "--check-outputs" && i + 1 < argc) st_check = argv[++i];
        else if (a == "--save-outputs" && i + 1 < argc) st_save = argv[++i];
        else if (a == "--safetensors-info" && i + 1 < argc) st_info = argv[++i];
        else if (a == "--max-diff" && i + 1 < argc) max_diff_allowed = std::atof(argv[++i]);
        else if (a == "--no-fuse") fuse = false;
        else if (a == "-t" || a == "--timing") timing = true;
        else if (a == "--parse-only") parse_only = true;
        else if (a == "--tune") tune = true;
        else if (a == "--graph") use_graph = true;
        else if (!a.empty() && a[0] != '-') path = a;
        else { std::fprintf(stderr, "unknown or incomplete option %s\n", a.c_str()); return 1; }
    }
    if (!st_info.empty()) { // list (and with --save-outputs re-write) a Safetensors file: needs no model and no GPU
        try {
            const auto tensors = safetensors::read(st_info);
            std::vector<std::pair<std::string, safetensors::Entry>> all(tensors.begin(), tensors.end());
            for (auto &t : all) {
                uint64_t sum = 1469598103934665603ull; // FNV-1a of the bytes
                for (unsigned char c : t.second.data) { sum ^= c; sum *= 1099511628211ull; }
                std::printf("  %s: %s %s fnv1a=%016llx\n", t.first.c_str(), t.second.dtype.c_str(), shape_str(t.second.shape).c_str(), (unsigned long long)sum);
            }
            if (!st_save.empty()) safetensors::write(st_save, all);
            return 0;
        } catch (const std::exception &e) {
            std::fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
    if (path.empty()) { std::fprintf(stderr, "usage: rten_hip_run [-n N] [-s dim=value] [--input name=file] [--dump name=file] [--no-fuse] [-t] [--parse-only] model.onnx\n"); return 1; }

    try {
        const onnx::Model m = onnx::load(path);
        std::map<std::string, int> hist;
        for (auto &n : m.nodes) hist[n.op_type]++;
        size_t const_bytes = 0;
        for (auto &t : m.initializers) const_bytes += t.raw.size();
        std::printf("Model: %s (graph \"%s\", ir_version %" PRId64 ", opset %" PRId64 ", producer %s)\n", path.c_str(), m.graph_name.c_str(), m.ir_version, m.opset, m.producer.c_str());
        std::printf("  %zu nodes, %zu initializers (%.1f MB)\n  operators:", m.nodes.size(), m.initializers.size(), const_bytes / 1e6);
        for (auto &h : hist) std::printf(" %s x%d", h.first.c_str(), h.second);
        std::printf("\n");
        for (auto &in : m.inputs) {
            std::string dims = "[";
            for (size_t i = 0; i < in.dims.size(); i++) dims += (i ? ", " : "") + (in.dims[i] < 0 ? in.dim_params[i] : std::to_string(in.dims[i]));
            std::printf("  input  %s: %s %s]\n", in.name.c_str(), type_str(in.elem_type), dims.c_str());
        }
        for (auto &o : m.outputs) std::printf("  output %s: %s\n", o.name.c_str(), type_str(o.elem_type));
        {
            const onnx::Model c = Graph::canonical_form(m);
            if (c.nodes.size() != m.nodes.size()) {
                std::map<std::string, int> chist;
                for (auto &n : c.nodes) chist[n.op_type]++;
                std::printf("  canonical form (Constant nodes, Gelu / LayerNormalization patterns): %zu nodes:", c.nodes.size());
                for (auto &h : chist) std::printf(" %s x%d", h.first.c_str(), h.second);
                std::printf("\n");
            }
        }
        if (parse_only) return 0;

        Context ctx(0);
        ctx.enable_pool();
        Graph::Options opt;
        opt.fuse = fuse;
        Graph g(ctx, m, opt);
        std::printf("Plan: %zu steps (%zu nodes folded into fused steps)\n", g.num_steps(), g.num_fused_away());
        if (g.num_staged_quantizers())
            std::printf("  int8: %zu DynamicQuantizeLinear write the staged layout directly; %zu producer statistics blocks\n", g.num_staged_quantizers(),
                        g.num_stats_blocks());

        // inputs
        std::map<std::string, safetensors::Entry> given;
        if (!st_inputs.empty()) given = safetensors::read(st_inputs);
        std::vector<Tensor> feeds_store;
        std::vector<std::pair<std::string, const Tensor *>> feeds;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        auto next_f32 = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) & 0xffffff) / 16777216.0f; };
        feeds_store.reserve(g.inputs().size());
        for (auto &in : g.inputs()) {
            std::vector<int64_t> shape;
            auto gv = given.find(in.name);
            if (gv != given.end()) {
                shape = gv->second.shape; // the file decides the symbolic dimensions
                if (shape.size() != in.dims.size()) throw GraphError("input " + in.name + ": the Safetensors tensor has a different rank");
                for (size_t i = 0; i < shape.size(); i++)
                    if (in.dims[i] >= 0 && in.dims[i] != shape[i]) throw GraphError("input " + in.name + ": the Safetensors tensor does not match the model's fixed dimensions");
            } else {
                for (size_t i = 0; i < in.dims.size(); i++) {
                    int64_t d = in.dims[i];
                    if (d < 0) { auto it = sizes.find(in.dim_params[i]); d = it == sizes.end() ? 1 : it->second; }
                    shape.push_back(d);
                }
            }
            int64_t n = 1;
            for (int64_t d : shape) n *= d;
            const bool is_f32 = in.elem_type == onnx::FLOAT;
            const size_t esz = is_f32 || in.elem_type == onnx::INT32 || in.elem_type == onnx::INT64 ? 4 : 1;
            std::vector<uint8_t> host((size_t)n * esz, 0);
            auto f = input_files.find(in.name);
            if (gv != given.end()) {
                const safetensors::Entry &e = gv->second;
                if (e.dtype == "I64" && esz == 4 && !is_f32) { // int64 inputs are narrowed to int32, as the reference does at load
                    if (e.data.size() != (size_t)n * 8) throw GraphError("input " + in.name + ": Safetensors entry holds " + std::to_string(e.data.size()) + " bytes, the model input needs " + std::to_string(n * 8));
                    for (int64_t i = 0; i < n; i++) { int64_t v; std::memcpy(&v, e.data.data() + 8 * i, 8); const int32_t x = (int32_t)v; std::memcpy(host.data() + 4 * i, &x, 4); }
                } else if (e.data.size() == host.size() && ((is_f32 && e.dtype == "F32") || (!is_f32 && e.dtype != "F32"))) {
                    std::memcpy(host.data(), e.data.data(), host.size());
                } else throw GraphError("input " + in.name + ": Safetensors dtype " + e.dtype + " does not fit the model input");
            } else if (f != input_files.end()) {
                std::ifstream fi(f->second, std::ios::binary);
                if (!fi) throw GraphError("cannot open input file " + f->second);
                fi.read((char *)host.data(), (std::streamsize)host.size());
                if ((size_t)fi.gcount() != host.size()) throw GraphError("input file " + f->second + " is shorter than the tensor");
            } else if (is_f32) {
                float *p = (float *)host.data();
                for (int64_t i = 0; i < n; i++) p[i] = next_f32();
            }
            const DType dt = is_f32 ? DType::F32 : (in.elem_type == onnx::UINT8 ? DType::U8 : in.elem_type == onnx::INT8 ? DType::I8 : DType::I32);
            Tensor t(ctx, shape, dt);
            if (t.bytes()) ctx.check(rten_hip_memcpy_h2d(ctx.raw(), t.ptr(), host.data(), t.bytes()));
            std::printf("  Input \"%s\" %s shape: %s\n", in.name.c_str(), gv != given.end() ? "read, " : "generated", shape_str(shape).c_str());
            feeds_store.push_back(std::move(t));
            feeds.emplace_back(in.name, &feeds_store.back());
        }

        // warm-up (first run allocates the pool), then timed runs
        if (tune) {
            const auto t0 = std::chrono::steady_clock::now();
            const size_t n = g.autotune(feeds);
            std::printf("  Tuned the launch plan of %zu convolution / MatMul steps in %.2fs\n", n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        std::vector<Tensor> outs;
        const std::vector<Tensor> *result = &outs;
        if (use_graph) {
            result = &g.capture(feeds);
            std::printf("  Captured the plan into a hipGraph\n");
        } else {
            outs = g.run(feeds);
        }
        ctx.sync();
        std::vector<double> ms;
        for (int i = 0; i < iters; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            if (use_graph) g.replay();
            else outs = g.run(feeds);
            ctx.sync();
            ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            std::printf("  #%d - %.3fms\n", i + 1, ms.back());
        }
        if (!ms.empty()) {
            double mean = 0, mn = ms[0], mx = ms[0], var = 0;
            for (double v : ms) { mean += v; mn = std::min(mn, v); mx = std::max(mx, v); }
            mean /= ms.size();
            for (double v : ms) var += (v - mean) * (v - mean);
            std::printf("  Duration stats: mean %.3fms, min %.3fms, max %.3fms, std %.3fms\n", mean, mn, mx, std::sqrt(var / ms.size()));
        }
        if (timing) {
            std::vector<Graph::Timing> tm;
            if (use_graph) throw GraphError("--timing and --graph exclude each other");
            outs = g.run(feeds, &tm);
            std::map<std::string, std::pair<int, double>> by_op;
            double total = 0;
            for (auto &t : tm) { by_op[t.op].first++; by_op[t.op].second += t.ms; total += t.ms; }
            std::printf("  Operator timing (each operator followed by a sync; total %.3fms):\n", total);
            for (auto &kvp : by_op) std::printf("    %-40s x%-4d %8.3fms %5.1f%%\n", kvp.first.c_str(), kvp.second.first, kvp.second.second, 100.0 * kvp.second.second / total);
        }
        const std::vector<Tensor> &outs_ref = *result;
        for (size_t i = 0; i < outs_ref.size(); i++) {
            const std::string &name = g.outputs()[i].name;
            std::printf("  Output \"%s\" resolved shape %s\n", name.c_str(), shape_str(outs_ref[i].shape()).c_str());
            auto d = dumps.find(name);
            if (d != dumps.end()) {
                std::vector<uint8_t> host(outs_ref[i].bytes());
                if (outs_ref[i].bytes()) ctx.check(rten_hip_memcpy_d2h(ctx.raw(), host.data(), outs_ref[i].ptr(), outs_ref[i].bytes()));
                std::ofstream fo(d->second, std::ios::binary);
                fo.write((const char *)host.data(), (std::streamsize)host.size());
            }
        }
        auto st_dtype = [](DType t) { return t == DType::F32 ? "F32" : t == DType::I32 ? "I32" : t == DType::U8 ? "U8" : "I8"; };
        auto fetch = [&](const Tensor &t) {
            safetensors::Entry e;
            e.dtype = st_dtype(t.dtype());
            e.shape = t.shape();
            e.data.resize(t.bytes());
            if (t.bytes()) ctx.check(rten_hip_memcpy_d2h(ctx.raw(), &e.data[0], t.ptr(), t.bytes()));
            return e;
        };
        if (!st_save.empty()) {
            std::vector<std::pair<std::string, safetensors::Entry>> all;
            for (size_t i = 0; i < outs_ref.size(); i++) all.emplace_back(g.outputs()[i].name, fetch(outs_ref[i]));
            safetensors::write(st_save, all);
            std::printf("  Saved %zu outputs to %s\n", all.size(), st_save.c_str());
        }
        bool check_failed = false;
        if (!st_check.empty()) {
            const std::map<std::string, safetensors::Entry> expected = safetensors::read(st_check);
            for (size_t i = 0; i < outs_ref.size(); i++) {
                const std::string &name = g.outputs()[i].name;
                auto ex = expected.find(name);
                if (ex == expected.end()) { std::printf("  Output \"%s\" has no expected value\n", name.c_str()); continue; }
                const safetensors::Entry got = fetch(outs_ref[i]);
                if (got.shape != ex->second.shape) { std::printf("  Output \"%s\" shape %s does not match expected %s\n", name.c_str(), shape_str(got.shape).c_str(), shape_str(ex->second.shape).c_str()); check_failed = true; continue; }
                if (got.dtype != ex->second.dtype) { std::printf("  Output \"%s\" dtype %s does not match expected %s\n", name.c_str(), got.dtype.c_str(), ex->second.dtype.c_str()); check_failed = true; continue; }
                if (got.dtype != "F32") { std::fprintf(stderr, "  Unable to compare outputs. Unsupported tensor types.\n"); continue; }
                float max_diff = 0.f;
                size_t bit_diffs = 0, nan_diffs = 0; // a NaN on one side only (or NaNs that differ in their bits) is sticky: it fails the gate
                const float *a = (const float *)got.data.data(), *b = (const float *)ex->second.data.data();
                for (int64_t k = 0; k < got.len(); k++) {
                    const bool same_bits = std::memcmp(a + k, b + k, 4) == 0;
                    bit_diffs += !same_bits;
                    if (same_bits) continue; // identical bits (NaNs included) are equal
                    const float d = std::fabs(a[k] - b[k]);
                    if (std::isnan(d)) nan_diffs++;
                    else max_diff = std::max(max_diff, d);
                }
                std::printf("  Output \"%s\" vs expected: max diff %.6f (%zu of %lld elements differ in their bits%s)\n", name.c_str(), max_diff, bit_diffs, (long long)got.len(),
                            nan_diffs ? (", " + std::to_string(nan_diffs) + " NaN mismatches").c_str() : "");
                if (max_diff_allowed >= 0 && (nan_diffs || !(max_diff <= max_diff_allowed))) check_failed = true;
            }
        }
        return check_failed && max_diff_allowed >= 0 ? 3 : 0;
    } catch (const safetensors::Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    } catch (const OpError &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return e.kind == OpError::BackendUnavailable ? 2 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
