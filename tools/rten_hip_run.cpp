// rten_hip_run -- run an ONNX model on the HIP backend, in the manner of `rten model.onnx` (rten-cli/src/main.rs).
//
//   rten_hip_run [options] model.onnx
//     -n, --n-iters N        timed runs after one warm-up (default 1); prints "#i - t ms" and mean / min / max / std like
//                            rten-cli/src/main.rs:301-347
//     -s, --size name=value  value of a symbolic input dimension (default 1), e.g. -s batch=32
//     --input name=path      raw little-endian file for an input (otherwise floats are U[0,1) from a fixed-seed generator
//                            and integers are 0, rten-cli/src/input_generator.rs:101-144)
//     --dump name=path       write an output tensor as raw bytes
//     --no-fuse              run the graph node by node (no fusion passes)
//     -t, --timing           per-operator table (each operator followed by a sync)
//     --tune                 time the candidate launch plans of every f32 convolution at load and keep the fastest
//     --graph                capture one run into a hipGraph and replay it for the timed runs
//     --parse-only           print the model summary and exit (needs no GPU)
//
// There is no CPU fallback: without an MI355X the tool reports BackendUnavailable and exits 2.
#include <cinttypes>
#include <cmath>

#include "rten_hip_graph.hpp"

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
        else if (a == "--no-fuse") fuse = false;
        else if (a == "-t" || a == "--timing") timing = true;
        else if (a == "--parse-only") parse_only = true;
        else if (a == "--tune") tune = true;
        else if (a == "--graph") use_graph = true;
        else if (!a.empty() && a[0] != '-') path = a;
        else { std::fprintf(stderr, "unknown or incomplete option %s\n", a.c_str()); return 1; }
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
        std::vector<Tensor> feeds_store;
        std::vector<std::pair<std::string, const Tensor *>> feeds;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        auto next_f32 = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) & 0xffffff) / 16777216.0f; };
        feeds_store.reserve(g.inputs().size());
        for (auto &in : g.inputs()) {
            std::vector<int64_t> shape;
            for (size_t i = 0; i < in.dims.size(); i++) {
                int64_t d = in.dims[i];
                if (d < 0) { auto it = sizes.find(in.dim_params[i]); d = it == sizes.end() ? 1 : it->second; }
                shape.push_back(d);
            }
            int64_t n = 1;
            for (int64_t d : shape) n *= d;
            const bool is_f32 = in.elem_type == onnx::FLOAT;
            const size_t esz = is_f32 || in.elem_type == onnx::INT32 || in.elem_type == onnx::INT64 ? 4 : 1;
            std::vector<uint8_t> host((size_t)n * esz, 0);
            auto f = input_files.find(in.name);
            if (f != input_files.end()) {
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
            std::printf("  Input \"%s\" generated shape: %s\n", in.name.c_str(), shape_str(shape).c_str());
            feeds_store.push_back(std::move(t));
            feeds.emplace_back(in.name, &feeds_store.back());
        }

        // warm-up (first run allocates the pool), then timed runs
        if (tune) {
            const auto t0 = std::chrono::steady_clock::now();
            const size_t n = g.autotune(feeds);
            std::printf("  Tuned the launch plan of %zu convolution steps in %.2fs\n", n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
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
        return 0;
    } catch (const OpError &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return e.kind == OpError::BackendUnavailable ? 2 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
