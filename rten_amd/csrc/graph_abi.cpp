// The plan executor (include/rten_hip_graph.hpp: ONNX in, every value resident in HBM, fused steps, committed launch plan, hipGraph replay,
// independent sub-batch chains) behind the C ABI: rten_hip_model_load / _bind / _prepare / _run / _output / _destroy.
//
// Why it is in the library: a Rust host cannot keep values on the device between operators without a `Value::Device` variant (a change to the
// reference's `Value`, src/value.rs:487).  It CAN wrap a maximal run of accelerated nodes as ONE `Operator` that owns a graph -- the reference's own
// `SubgraphOperator` (src/operator.rs:630-646) is the precedent -- and for that it needs the executor through `extern "C"`, not a C++ header
// (`HipSubgraph`).  `bench.py --via-executor` times exactly this path, so the measured path is the product path.
//
// What a graph object owns: `chains` contexts (one stream each, GPU_MAX_HW_QUEUES permitting one hardware queue each), one compiled
// rten_hip::Graph per chain (constants uploaded and prepacked per chain), the resident full-batch input / output buffers (a chain works on its
// dim-0 slice in place) and one captured hipGraph per chain.
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rten_hip_graph.hpp"

#define RTEN_EXPORT extern "C" __attribute__((visibility("default")))

using namespace rten_hip;

namespace {

// ---- the launch-plan files of profiles/plans/ are JSON objects of {step name: [variant, split mode, K groups, tile order]}, optionally one level
// deeper keyed by sub-batch size ({"8": {...}}).  A reader for exactly that subset (objects, arrays, strings, integers).
struct PlanJson {
    const char *p, *e;
    std::string err;
    std::map<std::string, std::vector<std::string>> names; // lists of names, by key: the int8 plan's edge lists ({"qout": ["s0b0c1", ...]})
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool lit(char c) { ws(); if (p < e && *p == c) { p++; return true; } return false; }
    bool str(std::string &out) {
        ws();
        if (p >= e || *p != '"') return false;
        p++;
        out.clear();
        while (p < e && *p != '"') { if (*p == '\\' && p + 1 < e) p++; out.push_back(*p++); }
        if (p >= e) return false;
        p++;
        return true;
    }
    bool integer(long long &v) {
        ws();
        const char *s = p;
        if (p < e && (*p == '-' || *p == '+')) p++;
        while (p < e && *p >= '0' && *p <= '9') p++;
        if (p == s) return false;
        v = std::strtoll(std::string(s, p).c_str(), nullptr, 10);
        return true;
    }
    // {name: [ints]} -> table; nested objects are returned under their key
    bool object(std::map<std::string, std::vector<long long>> &leaves, std::map<std::string, std::map<std::string, std::vector<long long>>> &nested) {
        if (!lit('{')) return false;
        if (lit('}')) return true;
        do {
            std::string key;
            if (!str(key) || !lit(':')) return false;
            ws();
            if (p < e && *p == '{') {
                std::map<std::string, std::map<std::string, std::vector<long long>>> deeper;
                if (!object(nested[key], deeper)) return false;
            } else if (p < e && *p == '[') {
                p++;
                std::vector<long long> v;
                if (!lit(']')) {
                    do {
                        long long x;
                        std::string sname;
                        ws();
                        if (p < e && *p == '"') { if (!str(sname)) return false; names[key].push_back(sname); v.push_back(0); }
                        else if (!integer(x)) return false;
                        else v.push_back(x);
                    } while (lit(','));
                    if (!lit(']')) return false;
                }
                leaves[key] = v;
            } else {
                return false;
            }
        } while (lit(','));
        return lit('}');
    }
};

std::map<std::string, GemmPlan> plan_table(const std::map<std::string, std::vector<long long>> &leaves) {
    std::map<std::string, GemmPlan> t;
    for (auto &kv : leaves) {
        if (kv.second.size() < 3) continue;
        GemmPlan g;
        g.set = true;
        g.variant = (int)kv.second[0];
        g.mode = (int)kv.second[1];
        g.groups = (int)kv.second[2];
        g.order = kv.second.size() > 3 ? (int)kv.second[3] : 0;
        t[kv.first] = g;
    }
    return t;
}

DType elem_dtype(int32_t onnx_type) {
    switch (onnx_type) {
    case onnx::FLOAT: return DType::F32;
    case onnx::INT32: case onnx::INT64: case onnx::BOOL: return DType::I32; // int64 and bool are int32 at the API (onnx_loader.rs:332-339)
    case onnx::UINT8: return DType::U8;
    default: return DType::I8;
    }
}

} // namespace

struct rten_hip_model {
    rten_hip_ctx *caller = nullptr; // the caller's context: its stream is ordered before / after a run, errors are reported on it
    int chains = 1;
    std::vector<std::unique_ptr<Context>> ctxs;
    std::vector<std::unique_ptr<Graph>> graphs;
    std::vector<onnx::ValueInfo> inputs, outputs;
    std::map<std::string, std::vector<long long>> plan_flat;
    std::map<std::string, std::map<std::string, std::vector<long long>>> plan_by_batch;
    bool have_plan = false;
    // bound state
    std::vector<std::unique_ptr<Tensor>> full_in, full_out; // resident full-batch buffers (owned by ctxs[0])
    std::vector<std::vector<Tensor>> chain_in;              // [chain][input]: dim-0 slices of full_in
    std::vector<std::vector<int64_t>> out_shape;
    std::vector<int64_t> sub, start;                        // sub-batch sizes / first rows
    size_t planned_steps = 0, tuned_steps = 0;
    bool prepared = false;
    void *arena_ptr = nullptr; // chain 0's coalesced constants (owned by graphs[0]; a clone: its origin's)
    size_t arena_bytes = 0;
    std::shared_ptr<onnx::Model> parsed; // kept for rten_hip_model_clone
    Graph::Options opts;
    rten_hip_model *origin = nullptr;    // a clone shares its origin's constants and must be destroyed first
    int clones = 0;
    std::string last_error;
};

namespace {
// text of the calling thread's last FAILED rten_hip_model_load / _load_ex (there is no model object to carry it): rten_hip_model_load_error()
thread_local std::string tls_load_error;
int32_t load_fail(int32_t code, const std::string &msg) {
    tls_load_error = msg;
    return code;
}
int32_t fail(rten_hip_model *g, int32_t code, const std::string &msg) {
    if (g) g->last_error = msg;
    return code;
}
int32_t code_of(const OpError &e) {
    switch (e.kind) {
    case OpError::InvalidValue: return RTEN_HIP_ERR_INVALID_VALUE;
    case OpError::IncompatibleInputShapes: return RTEN_HIP_ERR_INCOMPATIBLE_SHAPES;
    case OpError::UnsupportedValue: case OpError::UnsupportedType: return RTEN_HIP_ERR_UNSUPPORTED;
    case OpError::BackendUnavailable: return RTEN_HIP_ERR_NO_DEVICE;
    default: return RTEN_HIP_ERR_HIP;
    }
}
} // namespace

// Parses `onnx` (the bytes of an ONNX ModelProto), compiles it `chains` times (constants uploaded, conv weights prepacked, fusions applied) and
// records the launch plan.  `plan_json`: NULL (the backend's automatic plans, or rten_hip_model_prepare(tune = 1)) or the text of a plan file.
// Every chain lives on `ctx`'s device.  flags: RTEN_HIP_MODEL_RECEIVE_WEIGHTS = this process will receive the weight arena
// (rten_hip_model_weight_arena) by broadcast from the process that loaded the file for real: large initializers are not uploaded.
// A failed load leaves its reason in rten_hip_model_load_error() (per calling thread).
RTEN_EXPORT int32_t rten_hip_model_load_ex(rten_hip_ctx *ctx, const void *onnx_bytes, size_t onnx_len, const char *plan_json, int32_t chains, uint32_t flags,
                                           rten_hip_model **out_graph) {
    if (!out_graph) return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_load: out_model is NULL");
    *out_graph = nullptr;
    if (!ctx || !onnx_bytes || !onnx_len || chains < 1 || chains > 16 || (flags & ~(uint32_t)RTEN_HIP_MODEL_RECEIVE_WEIGHTS))
        return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_load: null context / empty model / chains outside 1..16 / unknown flag bits");
    std::unique_ptr<rten_hip_model> g(new rten_hip_model());
    g->caller = ctx;
    g->chains = chains;
    const int32_t device_id = rten_hip_device_id(ctx);
    try {
        g->parsed = std::make_shared<onnx::Model>(onnx::parse((const uint8_t *)onnx_bytes, onnx_len));
        const onnx::Model &m = *g->parsed;
        Graph::Options &opts = g->opts;
        opts.skip_large_uploads = (flags & RTEN_HIP_MODEL_RECEIVE_WEIGHTS) != 0;
        if (plan_json && *plan_json) {
            PlanJson pj{plan_json, plan_json + std::strlen(plan_json), {}};
            if (!pj.object(g->plan_flat, g->plan_by_batch)) return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_load: the plan file is not the JSON subset of profiles/plans/ (objects of integer / name arrays)");
            g->have_plan = true;
            // quantized-output launches (and quantize-on-load layers) are opt-in per edge; the former need the device to themselves: one chain only.
            // Asking for them with several chains is an error, not something to drop silently.
            if ((pj.names.count("qout") && !pj.names["qout"].empty()) && chains != 1)
                return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_load: the plan lists quantized-output edges (\"qout\"), which need chains == 1");
            if (pj.names.count("qout")) opts.qout.insert(pj.names["qout"].begin(), pj.names["qout"].end());
            // "qout2": the same kind of edge in the recompute form (two launches, no exchange): allowed with replicas
            if (pj.names.count("qout2")) opts.qout_recompute.insert(pj.names["qout2"].begin(), pj.names["qout2"].end());
            if (pj.names.count("fused_dql")) opts.fused_dql.insert(pj.names["fused_dql"].begin(), pj.names["fused_dql"].end());
            if (pj.names.count("pairs")) opts.pairs.insert(pj.names["pairs"].begin(), pj.names["pairs"].end());
            if (pj.names.count("pair_shortcuts")) opts.pair_shortcuts.insert(pj.names["pair_shortcuts"].begin(), pj.names["pair_shortcuts"].end());
        }
        for (int c = 0; c < chains; c++) {
            // chain 0 runs on the CALLER's context (its stream): a model with N chains owns N - 1 streams.  One stream more than chains costs real
            // time on this runtime -- streams map onto hardware queues, and a fifth active stream collides with one of the four chains
            // (measured: 3.80 ms with an idle caller stream + 4 chain streams, 2.7 ms with 4 streams in all; profiles/r07/executor_vs_runner.txt)
            if (c == 0) g->ctxs.emplace_back(new Context(ctx, Context::Borrow()));
            else g->ctxs.emplace_back(new Context(device_id));
            g->ctxs.back()->enable_pool(true);
            // chains 1.. share chain 0's device constants and prepacked weights (one copy of the weight set per model, as in the Python runner's arena)
            if (c == 0) {
                g->graphs.emplace_back(new Graph(*g->ctxs.back(), m, opts));
                // every device constant in ONE allocation (before anybody aliases them): the arena a sharded deployment broadcasts once
                const auto arena = g->graphs[0]->coalesce_constants();
                g->arena_ptr = arena.first;
                g->arena_bytes = arena.second;
                // a graph whose rows are coupled through dim 0 cannot run as independent sub-batch chains: refused, never silently different
                const std::string coupled = g->graphs[0]->batch_coupled_step();
                if (chains > 1 && !coupled.empty())
                    return load_fail(RTEN_HIP_ERR_UNSUPPORTED, "model_load: chains > 1, but " + coupled + " couples the rows of dim 0 (its result for one row depends on the others)");
            } else {
                g->graphs.emplace_back(new Graph(*g->ctxs.back(), m, opts, *g->graphs[0]));
            }
        }
        g->inputs = g->graphs[0]->inputs();
        g->outputs = g->graphs[0]->outputs();
    } catch (const onnx::ParseError &e) {
        while (!g->graphs.empty()) g->graphs.pop_back();
        return load_fail(RTEN_HIP_ERR_INVALID_VALUE, std::string("model_load: ") + e.what());
    } catch (const OpError &e) {
        while (!g->graphs.empty()) g->graphs.pop_back();
        return load_fail(code_of(e), "model_load: " + OpError::kind_name(e.kind) + ": " + e.msg);
    } catch (const std::exception &e) { // GraphError: an operator outside the registry, an attribute form that is not covered, ...
        while (!g->graphs.empty()) g->graphs.pop_back();
        return load_fail(RTEN_HIP_ERR_INVALID_VALUE, std::string("model_load: ") + e.what());
    }
    tls_load_error.clear();
    *out_graph = g.release();
    return RTEN_HIP_OK;
}

// The round-4 entry point: `device_id` must be the device `ctx` was created on (the chains are created there; a mismatch used to put chains 1..
// on another device than chain 0 and the shared constants).
RTEN_EXPORT int32_t rten_hip_model_load(rten_hip_ctx *ctx, const void *onnx_bytes, size_t onnx_len, const char *plan_json, int32_t chains, int32_t device_id,
                                        rten_hip_model **out_graph) {
    if (out_graph) *out_graph = nullptr;
    if (ctx && device_id != rten_hip_device_id(ctx)) return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_load: device_id is not the device of `ctx`");
    return rten_hip_model_load_ex(ctx, onnx_bytes, onnx_len, plan_json, chains, 0u, out_graph);
}

RTEN_EXPORT const char *rten_hip_model_load_error(void) { return tls_load_error.c_str(); }

// Another REPLICA of a loaded model on another context (stream): the same graph, plan and options, its own buffers and hipGraphs, and the ORIGIN's
// device constants and prepacked weights (no second copy of the weight arena).  What a host that serves independent batches keeps one of per request
// stream ("lanes": consecutive batches on different replicas overlap -- bench.py --lanes).  `ctx` must live on the origin's
// device and, like the origin's context, outlive the replica; replicas are destroyed BEFORE their origin (rten_hip_model_destroy(origin) refuses
// while one is alive).  Bind inputs and prepare the replica like any model.
RTEN_EXPORT int32_t rten_hip_model_clone(rten_hip_model *src, rten_hip_ctx *ctx, rten_hip_model **out_model) {
    if (!out_model) return RTEN_HIP_ERR_INVALID_VALUE;
    *out_model = nullptr;
    if (!src || !ctx || !src->parsed) return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_clone: null model / context");
    if (src->origin) src = src->origin; // a clone of a clone shares the same origin
    if (rten_hip_device_id(ctx) != rten_hip_device_id(src->caller)) return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_clone: the context lives on another device than the model");
    // Replicas run side by side on their own streams; a quantized-output launch ("qout" edges of the plan) is a grid-wide exchange that needs the device to
    // itself (rten_hip.h, rten_hip_conv2d_int8_qout: time-out contract, sticky fault) -- the same reason rten_hip_model_load_ex refuses them with chains != 1.
    // Refused here, not dropped silently: load the origin from a plan without "qout" (profiles/plans/int8_lanes.json) when replicas are wanted.
    if (!src->opts.qout.empty())
        return load_fail(RTEN_HIP_ERR_INVALID_VALUE, "model_clone: the model's plan lists quantized-output edges (\"qout\"), which need the device to themselves; "
                                                      "replicas run concurrently -- load the origin from a plan without them");
    std::unique_ptr<rten_hip_model> g(new rten_hip_model());
    g->caller = ctx;
    g->chains = src->chains;
    g->parsed = src->parsed;
    g->opts = src->opts;
    g->plan_flat = src->plan_flat;
    g->plan_by_batch = src->plan_by_batch;
    g->have_plan = src->have_plan;
    g->arena_ptr = src->arena_ptr;
    g->arena_bytes = src->arena_bytes;
    try {
        const int32_t device_id = rten_hip_device_id(ctx);
        for (int c = 0; c < g->chains; c++) {
            if (c == 0) g->ctxs.emplace_back(new Context(ctx, Context::Borrow()));
            else g->ctxs.emplace_back(new Context(device_id));
            g->ctxs.back()->enable_pool(true);
            g->graphs.emplace_back(new Graph(*g->ctxs.back(), *g->parsed, g->opts, *src->graphs[0])); // every chain of a replica shares the origin's constants
        }
        g->inputs = g->graphs[0]->inputs();
        g->outputs = g->graphs[0]->outputs();
    } catch (const OpError &e) {
        while (!g->graphs.empty()) g->graphs.pop_back();
        return load_fail(code_of(e), "model_clone: " + OpError::kind_name(e.kind) + ": " + e.msg);
    } catch (const std::exception &e) {
        while (!g->graphs.empty()) g->graphs.pop_back();
        return load_fail(RTEN_HIP_ERR_INVALID_VALUE, std::string("model_clone: ") + e.what());
    }
    g->origin = src;
    src->clones++;
    tls_load_error.clear();
    *out_model = g.release();
    return RTEN_HIP_OK;
}

// The model's weight arena: ONE device allocation holding every constant of the graph (initializers, constants derived at load, prepacked weights) in a
// layout that depends only on the model and the load options -- what rank 0 of a batch-sharded job broadcasts once (rten_hip_broadcast: RCCL over xGMI)
// to ranks that loaded with RTEN_HIP_MODEL_RECEIVE_WEIGHTS.  Valid until rten_hip_model_destroy.
RTEN_EXPORT int32_t rten_hip_model_weight_arena(rten_hip_model *g, void **dev_ptr, size_t *bytes) {
    if (!g) return RTEN_HIP_ERR_INVALID_VALUE;
    if (dev_ptr) *dev_ptr = g->arena_ptr;
    if (bytes) *bytes = g->arena_bytes;
    return RTEN_HIP_OK;
}

RTEN_EXPORT const char *rten_hip_model_last_error(const rten_hip_model *g) { return g ? g->last_error.c_str() : "null graph"; }

// Counts and names: n_inputs / n_outputs / plan steps (how many convolution steps took an entry of the plan file, after _prepare) / total steps.
RTEN_EXPORT int32_t rten_hip_model_info(const rten_hip_model *g, int32_t *n_inputs, int32_t *n_outputs, int32_t *n_steps, int32_t *n_planned_steps) {
    if (!g) return RTEN_HIP_ERR_INVALID_VALUE;
    if (n_inputs) *n_inputs = (int32_t)g->inputs.size();
    if (n_outputs) *n_outputs = (int32_t)g->outputs.size();
    if (n_steps) *n_steps = (int32_t)g->graphs[0]->num_steps();
    if (n_planned_steps) *n_planned_steps = (int32_t)(g->planned_steps + g->tuned_steps + g->graphs[0]->num_qout_edges() + g->graphs[0]->num_dql_loader_steps() + g->graphs[0]->num_conv_pairs());
    return RTEN_HIP_OK;
}
RTEN_EXPORT const char *rten_hip_model_input_name(const rten_hip_model *g, int32_t i) { return (g && i >= 0 && (size_t)i < g->inputs.size()) ? g->inputs[(size_t)i].name.c_str() : nullptr; }
RTEN_EXPORT const char *rten_hip_model_output_name(const rten_hip_model *g, int32_t i) { return (g && i >= 0 && (size_t)i < g->outputs.size()) ? g->outputs[(size_t)i].name.c_str() : nullptr; }

// Declares input `i`'s FULL-batch shape (dim 0 = batch, split over the chains) and allocates its resident buffer; `*dev_ptr` is where the caller
// writes the input (device memory, valid until _destroy).
RTEN_EXPORT int32_t rten_hip_model_bind_input(rten_hip_model *g, int32_t i, const int64_t *shape, int32_t ndim, void **dev_ptr) {
    if (!g || i < 0 || (size_t)i >= g->inputs.size() || !shape || ndim < 1 || ndim > 8) return RTEN_HIP_ERR_INVALID_VALUE;
    if (g->prepared) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "bind_input after prepare");
    try {
        const int64_t batch = shape[0];
        if (batch < g->chains) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "fewer rows than chains");
        if (g->sub.empty()) {
            const int64_t q = batch / g->chains, r = batch % g->chains;
            int64_t at = 0;
            for (int c = 0; c < g->chains; c++) { g->sub.push_back(q + (c < r ? 1 : 0)); g->start.push_back(at); at += g->sub.back(); }
        } else if (g->sub.size() && g->start.back() + g->sub.back() != batch) {
            return fail(g, RTEN_HIP_ERR_INCOMPATIBLE_SHAPES, "every input must have the same dim 0 (the batch the chains split)");
        }
        if (g->full_in.size() < g->inputs.size()) { g->full_in.resize(g->inputs.size()); g->chain_in.resize((size_t)g->chains); for (auto &v : g->chain_in) v.resize(g->inputs.size()); }
        std::vector<int64_t> full(shape, shape + ndim);
        const DType dt = elem_dtype(g->inputs[(size_t)i].elem_type);
        g->full_in[(size_t)i].reset(new Tensor(*g->ctxs[0], full, dt));
        int64_t row = 1;
        for (int d = 1; d < ndim; d++) row *= shape[d];
        for (int c = 0; c < g->chains; c++) {
            std::vector<int64_t> s = full;
            s[0] = g->sub[(size_t)c];
            g->chain_in[(size_t)c][(size_t)i] = Tensor::view_at(*g->full_in[(size_t)i], (size_t)(g->start[(size_t)c] * row) * dtype_size(dt), s);
        }
        if (dev_ptr) *dev_ptr = g->full_in[(size_t)i]->ptr();
    } catch (const OpError &e) {
        return fail(g, code_of(e), e.msg);
    }
    return RTEN_HIP_OK;
}

// Applies the launch plan (or, with tune != 0 and no plan file, times the candidates once per distinct sub-batch size), runs every chain once
// (buffer pool and scratch reach their steady state) and captures each chain into a hipGraph.
RTEN_EXPORT int32_t rten_hip_model_prepare(rten_hip_model *g, int32_t tune) {
    if (!g) return RTEN_HIP_ERR_INVALID_VALUE;
    if (g->full_in.size() != g->inputs.size()) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "prepare: bind every input first");
    for (auto &t : g->full_in) if (!t) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "prepare: bind every input first");
    try {
        std::map<int64_t, std::map<std::string, GemmPlan>> tuned; // per sub-batch size
        g->full_out.clear();
        g->out_shape.clear();
        for (int c = 0; c < g->chains; c++) {
            Graph &gr = *g->graphs[(size_t)c];
            Context &cx = *g->ctxs[(size_t)c];
            Graph::Feeds feeds;
            for (size_t i = 0; i < g->inputs.size(); i++) feeds.emplace_back(g->inputs[i].name, &g->chain_in[(size_t)c][i]);
            const int64_t b = g->sub[(size_t)c];
            if (g->have_plan) {
                auto it = g->plan_by_batch.find(std::to_string(b));
                // a plan keyed by sub-batch size that has no entry for THIS chain's size plans nothing for it: an error, not a silent default
                bool keyed = false;
                for (auto &kv : g->plan_by_batch) if (!kv.first.empty() && kv.first.find_first_not_of("0123456789") == std::string::npos) keyed = true;
                if (keyed && it == g->plan_by_batch.end() && g->plan_flat.empty())
                    return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "prepare: the plan file is keyed by sub-batch size and has no entry for a chain of " + std::to_string(b) + " rows");
                // "shapes": entries by product shape for the MatMul-family steps no name matches (another exporter's file of the same model); set first -- it also
                // takes back what a previous table left on steps -- then the entries by name
                auto sh = g->plan_by_batch.find("shapes");
                gr.set_shape_plans(sh != g->plan_by_batch.end() ? plan_table(sh->second) : std::map<std::string, GemmPlan>());
                const size_t n = gr.apply_plan(plan_table(it != g->plan_by_batch.end() ? it->second : g->plan_flat));
                if (c == 0) g->planned_steps = n;
            } else if (tune) {
                if (!tuned.count(b)) { const size_t n = gr.autotune(feeds); tuned[b] = gr.plans(); if (c == 0) g->tuned_steps = n; }
                else gr.apply_plan(tuned[b]);
            }
            // A one-image chain of a larger batch: the reference's products have several rows (the whole batch), so the chain's one-row products must
            // NOT take the reference's vector-matrix (gemv) order -- the decision is made on the host while the launches are recorded, hence around
            // the probe run and the capture only (the context's setting is restored: chain 0 runs on the caller's context).
            const bool lone_row = b == 1 && g->start.back() + g->sub.back() > 1;
            struct Restore { // the caller's knobs come back as they were (chain 0 IS the caller's context), not as defaults
                Context &c; int32_t saved[8]; bool on = false;
                explicit Restore(Context &cx_) : c(cx_) { on = rten_hip_tuning_save(c.raw(), saved) == RTEN_HIP_OK; }
                ~Restore() { if (on) rten_hip_tuning_restore(c.raw(), saved); }
            } restore(cx);
            if (lone_row) cx.check(rten_hip_set_gemv_order(cx.raw(), 0, 0));
            if (c == 0) { // resident full-batch outputs, shaped from chain 0's (un-captured) first run
                const std::vector<Tensor> probe = gr.run(feeds);
                // what only a run can tell about dim-0 coupling (negative axes resolved against the ranks just seen, device transposes that move dim 0)
                if (g->chains > 1 && !gr.runtime_batch_coupled_step().empty())
                    return fail(g, RTEN_HIP_ERR_UNSUPPORTED, "prepare: chains > 1, but " + gr.runtime_batch_coupled_step() + " couples the rows of dim 0");
                for (size_t o = 0; o < probe.size(); o++) {
                    std::vector<int64_t> s = probe[o].shape();
                    if (s.empty()) return fail(g, RTEN_HIP_ERR_UNSUPPORTED, "a scalar graph output cannot be split over chains");
                    // the chains' rows are assembled along dim 0: an output whose dim 0 is not the chain's sub-batch (a Reshape to [N*k, ...], a
                    // batch-independent vector) would be copied past the end of the full-batch buffer on every replay
                    if (s[0] != g->sub[0])
                        return fail(g, RTEN_HIP_ERR_UNSUPPORTED, "output " + g->outputs[o].name + ": dim 0 (" + std::to_string(s[0]) + ") is not the chain's sub-batch (" +
                                                                      std::to_string(g->sub[0]) + "): the outputs of this graph cannot be assembled from dim-0 slices");
                    s[0] = g->start.back() + g->sub.back();
                    g->out_shape.push_back(s);
                    g->full_out.emplace_back(new Tensor(*g->ctxs[0], s, probe[o].dtype()));
                }
                cx.sync();
                g->planned_steps += gr.num_shape_planned(); // (the probe run resolved the shape-keyed entries)
            }
            // a chain's rows go into the resident full-batch outputs INSIDE its captured graph (logits-sized copies): a run is then `chains` graph
            // launches and nothing else
            gr.capture(feeds, [&](const std::vector<Tensor> &outs) {
                if (getenv("RTEN_MODEL_NO_GATHER")) return; // (diagnostic: timing without the output copies; the outputs are then not assembled)
                for (size_t o = 0; o < outs.size(); o++) {
                    if (outs[o].shape().empty() || outs[o].shape()[0] != g->sub[(size_t)c] ||
                        !std::equal(outs[o].shape().begin() + 1, outs[o].shape().end(), g->out_shape[o].begin() + 1, g->out_shape[o].end()))
                        throw OpError(OpError::UnsupportedValue, "output " + g->outputs[o].name + " of chain " + std::to_string(c) + " does not have the chain's rows on dim 0");
                    int64_t row = 1;
                    for (size_t d = 1; d < g->out_shape[o].size(); d++) row *= g->out_shape[o][d];
                    const size_t off = (size_t)(g->start[(size_t)c] * row) * dtype_size(outs[o].dtype());
                    if (outs[o].bytes()) cx.check(rten_hip_memcpy_d2d(cx.raw(), (char *)g->full_out[o]->ptr() + off, outs[o].ptr(), outs[o].bytes()));
                }
            });
        }
        for (auto &c : g->ctxs) c->sync();
        g->prepared = true;
        g->last_error.clear();
        if (g->have_plan && g->planned_steps + g->graphs[0]->num_qout_edges() + g->graphs[0]->num_dql_loader_steps() + g->graphs[0]->num_conv_pairs() == 0)
            g->last_error = "warning: the plan file matched no step of this graph (every launch runs the backend's automatic plan)";
    } catch (const OpError &e) {
        return fail(g, code_of(e), e.msg);
    } catch (const std::exception &e) {
        return fail(g, RTEN_HIP_ERR_INVALID_VALUE, e.what());
    }
    return RTEN_HIP_OK;
}

// The launch plan the model runs under, as the text of a plan file ({"<sub-batch>": {step: [variant, split mode, K groups, tile order]}}; what
// rten_hip_model_prepare(tune = 1) chose, or what the plan file / the backend's defaults left on the steps): write it to profiles/plans/ and every later
// process -- and every rank of a sharded job -- launches the same kernels without tuning.  Returns RTEN_HIP_ERR_INVALID_VALUE when `buf` is too small
// (`*needed` = bytes including the terminator).
RTEN_EXPORT int32_t rten_hip_model_plan_json(rten_hip_model *g, char *buf, size_t buf_len, size_t *needed) {
    if (!g || !g->prepared) return RTEN_HIP_ERR_INVALID_VALUE;
    std::string out = "{";
    std::set<int64_t> seen;
    for (int c = 0; c < g->chains; c++) {
        const int64_t b = g->sub[(size_t)c];
        if (!seen.insert(b).second) continue;
        if (out.size() > 1) out += ", ";
        out += "\"" + std::to_string(b) + "\": {";
        bool first = true;
        for (auto &kv : g->graphs[(size_t)c]->plans()) {
            if (!first) out += ", ";
            first = false;
            out += "\"" + kv.first + "\": [" + std::to_string(kv.second.variant) + ", " + std::to_string(kv.second.mode) + ", " + std::to_string(kv.second.groups) + ", " +
                   std::to_string(kv.second.order) + "]";
        }
        out += "}";
    }
    // the load-time lists the model was built with travel with the step tables: a plan written out and loaded again builds the same steps
    auto names = [&](const char *key, const std::set<std::string> &v) {
        if (v.empty()) return;
        if (out.size() > 1) out += ", ";
        out += std::string("\"") + key + "\": [";
        bool first = true;
        for (auto &n : v) { out += std::string(first ? "" : ", ") + "\"" + n + "\""; first = false; }
        out += "]";
    };
    names("pairs", g->opts.pairs);
    names("pair_shortcuts", g->opts.pair_shortcuts);
    names("fused_dql", g->opts.fused_dql);
    names("qout", g->opts.qout);
    names("qout2", g->opts.qout_recompute);
    out += "}";
    if (needed) *needed = out.size() + 1;
    if (!buf) return RTEN_HIP_ERR_INVALID_VALUE; // a size query: `*needed` is the answer, nothing to report
    if (buf_len < out.size() + 1) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "plan_json: buffer too small");
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return RTEN_HIP_OK;
}

// Replaces the launch plan of a loaded model (the text of a plan file, as for rten_hip_model_load_ex; the step tables only: "qout" / "fused_dql" / "pairs" / "pair_shortcuts" lists
// are load-time choices and are ignored here) and marks the model unprepared: the next rten_hip_model_prepare applies it and re-captures the chains.
// What a tuner that measures whole-model throughput under its real schedule (several replicas side by side: tools/tune_lanes.py) calls between runs.
RTEN_EXPORT int32_t rten_hip_model_set_plan(rten_hip_model *g, const char *plan_json) {
    if (!g || !plan_json || !*plan_json) return RTEN_HIP_ERR_INVALID_VALUE;
    std::map<std::string, std::vector<long long>> flat;
    std::map<std::string, std::map<std::string, std::vector<long long>>> keyed;
    PlanJson pj{plan_json, plan_json + std::strlen(plan_json), {}};
    if (!pj.object(flat, keyed)) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "set_plan: not the JSON subset of profiles/plans/");
    for (auto &c : g->ctxs) c->sync();
    g->plan_flat = std::move(flat);
    g->plan_by_batch = std::move(keyed);
    g->have_plan = true;
    g->prepared = false;
    return RTEN_HIP_OK;
}

// Instrumented pass (measurement aid; `bench.py`'s per-kernel roofline figures): every chain runs its plan EAGERLY `steps` times, chain after chain
// (serialised launches: clean per-kernel durations at the sub-batch shapes actually launched), with the backend's per-launch HIP-event profiler on
// (rten_hip_profile_*).  Result: a JSON array with one rten_hip_profile_report array per chain.  The captured graphs are untouched (an eager run
// takes the same pooled buffers the replay uses, so nothing else may run on the model meanwhile).
RTEN_EXPORT int32_t rten_hip_model_profile(rten_hip_model *g, int32_t steps, char *buf, size_t buf_len, size_t *needed) {
    if (!g || !g->prepared || steps < 1) return RTEN_HIP_ERR_INVALID_VALUE;
    std::string out = "[";
    try {
        for (auto &c : g->ctxs) c->sync();
        for (int c = 0; c < g->chains; c++) {
            Graph &gr = *g->graphs[(size_t)c];
            Context &cx = *g->ctxs[(size_t)c];
            Graph::Feeds feeds;
            for (size_t i = 0; i < g->inputs.size(); i++) feeds.emplace_back(g->inputs[i].name, &g->chain_in[(size_t)c][i]);
            struct Restore {
                Context &c; int32_t saved[8]; bool on = false;
                explicit Restore(Context &cx_) : c(cx_) { on = rten_hip_tuning_save(c.raw(), saved) == RTEN_HIP_OK; }
                ~Restore() { if (on) rten_hip_tuning_restore(c.raw(), saved); rten_hip_profile_enable(c.raw(), 0); }
            } restore(cx);
            if (g->sub[(size_t)c] == 1 && g->start.back() + g->sub.back() > 1) cx.check(rten_hip_set_gemv_order(cx.raw(), 0, 0)); // as in prepare
            cx.check(rten_hip_profile_reset(cx.raw()));
            cx.check(rten_hip_profile_enable(cx.raw(), 1));
            for (int s = 0; s < steps; s++) gr.run(feeds);
            cx.sync();
            cx.check(rten_hip_profile_enable(cx.raw(), 0));
            std::vector<char> rep((size_t)1 << 18);
            cx.check(rten_hip_profile_report(cx.raw(), rep.data(), (int32_t)rep.size()));
            if (c) out += ",";
            out += rep.data();
        }
    } catch (const OpError &e) {
        return fail(g, code_of(e), e.msg);
    } catch (const std::exception &e) {
        return fail(g, RTEN_HIP_ERR_INVALID_VALUE, e.what());
    }
    out += "]";
    if (needed) *needed = out.size() + 1;
    if (!buf || buf_len < out.size() + 1) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "model_profile: buffer too small");
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return RTEN_HIP_OK;
}

// One inference over the bound inputs.  flags bit 0: the inputs were written on the CALLER's stream since the last run (the chains then wait for
// that stream first; leave it clear when the inputs are already resident and visible).  On return the caller's stream is ordered after every chain,
// unless bit 1 is set (the caller synchronises the model itself before reading: back-to-back runs then never touch the caller's stream).
RTEN_EXPORT int32_t rten_hip_model_run(rten_hip_model *g, uint32_t flags) {
    if (!g || !g->prepared) return RTEN_HIP_ERR_INVALID_VALUE;
    try {
        for (int c = 0; c < g->chains; c++) {
            Context &cx = *g->ctxs[(size_t)c];
            if ((flags & 1u) && c > 0) cx.check(rten_hip_stream_wait(cx.raw(), g->caller));
            g->graphs[(size_t)c]->replay();
        }
        if (!(flags & 2u)) // bit 1: the caller will rten_hip_model_sync before it reads the outputs: no per-run event wait on its stream
            for (int c = 1; c < g->chains; c++) // (chain 0 IS the caller's stream)
                if (rten_hip_stream_wait(g->caller, g->ctxs[(size_t)c]->raw()) != RTEN_HIP_OK) return fail(g, RTEN_HIP_ERR_HIP, "stream_wait failed");
    } catch (const OpError &e) {
        return fail(g, code_of(e), e.msg);
    } catch (const std::exception &e) {
        return fail(g, RTEN_HIP_ERR_INVALID_VALUE, e.what());
    }
    return RTEN_HIP_OK;
}

// Waits for every chain (and reports a sticky device fault of any of them).
RTEN_EXPORT int32_t rten_hip_model_sync(rten_hip_model *g) {
    if (!g) return RTEN_HIP_ERR_INVALID_VALUE;
    for (auto &c : g->ctxs) {
        const int32_t rc = rten_hip_sync(c->raw());
        if (rc) return fail(g, rc, rten_hip_last_error(c->raw()));
    }
    return RTEN_HIP_OK;
}

namespace {
int32_t dtype_code(DType t) { return t == DType::F32 ? RTEN_HIP_DTYPE_F32 : t == DType::I32 ? RTEN_HIP_DTYPE_I32 : t == DType::U8 ? RTEN_HIP_DTYPE_U8 : RTEN_HIP_DTYPE_I8; }
} // namespace

RTEN_EXPORT int32_t rten_hip_model_input_dtype(const rten_hip_model *g, int32_t i, int32_t *dtype) {
    if (!g || !dtype || i < 0 || (size_t)i >= g->inputs.size()) return RTEN_HIP_ERR_INVALID_VALUE;
    *dtype = dtype_code(elem_dtype(g->inputs[(size_t)i].elem_type));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_model_output_dtype(const rten_hip_model *g, int32_t i, int32_t *dtype) {
    if (!g || !dtype || !g->prepared || i < 0 || (size_t)i >= g->full_out.size()) return RTEN_HIP_ERR_INVALID_VALUE;
    *dtype = dtype_code(g->full_out[(size_t)i]->dtype());
    return RTEN_HIP_OK;
}

// Output `i` after a run: device pointer of the resident full-batch tensor, its shape (up to 8 dims) and rank.
RTEN_EXPORT int32_t rten_hip_model_output(rten_hip_model *g, int32_t i, const void **dev_ptr, int64_t *shape, int32_t *ndim) {
    if (!g || !g->prepared || i < 0 || (size_t)i >= g->full_out.size()) return RTEN_HIP_ERR_INVALID_VALUE;
    if (dev_ptr) *dev_ptr = g->full_out[(size_t)i]->ptr();
    if (ndim) *ndim = (int32_t)g->out_shape[(size_t)i].size();
    if (shape) for (size_t d = 0; d < g->out_shape[(size_t)i].size() && d < 8; d++) shape[d] = g->out_shape[(size_t)i][d];
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_model_destroy(rten_hip_model *g) {
    if (!g) return RTEN_HIP_OK;
    if (g->clones > 0) return fail(g, RTEN_HIP_ERR_INVALID_VALUE, "model_destroy: replicas (rten_hip_model_clone) of this model are still alive: destroy them first");
    if (g->origin) g->origin->clones--;
    for (auto &c : g->ctxs) rten_hip_sync(c->raw());
    g->chain_in.clear();
    g->full_in.clear();
    g->full_out.clear();
    while (!g->graphs.empty()) g->graphs.pop_back(); // sharers before the donor (chain 0), graphs before their contexts
    g->ctxs.clear();
    delete g;
    return RTEN_HIP_OK;
}
