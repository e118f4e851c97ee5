// Context, device memory, stream/event/graph plumbing of librten_hip.so.
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "internal.h"

// Error text is per calling host thread (several Model::run threads may share a context): a thread always reads the
// message of ITS last failing call, and the returned pointer stays valid until that thread's next failing call.
static thread_local std::string tls_last_error;

int32_t rten_set_error(rten_hip_ctx *ctx, int32_t code, const char *fmt, ...) {
    (void)ctx;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    tls_last_error = buf;
    return code;
}

int32_t rten_check_hip(rten_hip_ctx *ctx, hipError_t e, const char *what) {
    if (e == hipSuccess) return RTEN_HIP_OK;
    return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// Grow-only scratch buffers and live hipGraphs.  A captured graph holds the scratch / aux pointer its launches were recorded with, so a
// buffer that a live graph of this context may replay from is never freed by a later, larger request: while `live_graphs` > 0 the old
// buffer is RETIRED (kept allocated, freed when the last graph of the context is destroyed, or with the context) and a new one is
// allocated for the eager caller.  (one Arc<HipContext> shared by several HipSubgraph models and the per-op wrappers.)
static void retire_or_free(rten_hip_ctx *ctx, void *buf) {
    if (!buf) return;
    if (ctx->live_graphs > 0) { ctx->retired.push_back(buf); return; }
    hipStreamSynchronize(ctx->stream);
    hipFree(buf);
}

void *rten_scratch(rten_hip_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return ctx->scratch;
    if (ctx->capturing) return nullptr; // cannot grow during capture
    retire_or_free(ctx, ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    // 288 GB of HBM make a generous floor cheap: growth (and therefore retirement) is rare.
    size_t want = bytes + bytes / 4;
    const size_t floor_bytes = (size_t)512 << 20;
    if (want < floor_bytes) want = floor_bytes;
    if (hipMalloc(&ctx->scratch, want) != hipSuccess) return nullptr;
    ctx->scratch_bytes = want;
    return ctx->scratch;
}

void *rten_aux_scratch(rten_hip_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->aux_bytes) return ctx->aux;
    if (ctx->capturing) return nullptr;
    retire_or_free(ctx, ctx->aux);
    ctx->aux = nullptr;
    ctx->aux_bytes = 0;
    size_t want = bytes + bytes / 4;
    const size_t floor_bytes = (size_t)16 << 20; // a floor here too: fewer distinct buffers over a context's life
    if (want < floor_bytes) want = floor_bytes;
    if (hipMalloc(&ctx->aux, want) != hipSuccess) return nullptr;
    ctx->aux_bytes = want;
    return ctx->aux;
}

static hipEvent_t get_event(rten_hip_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(rten_hip_ctx *c, const char *n, double fl, double by) : ctx(c), name(n), flops(fl), bytes(by) {
    if (ctx->profiling && !ctx->capturing) {
        e0 = get_event(ctx);
        e1 = get_event(ctx);
        hipEventRecord(e0, ctx->stream);
    }
}

ProfScope::~ProfScope() {
    if (e0) {
        hipEventRecord(e1, ctx->stream);
        ProfEntry &pe = ctx->prof[name];
        pe.pending.emplace_back(e0, e1);
        pe.pending_work.emplace_back(flops, bytes);
    }
}

static void prof_resolve(rten_hip_ctx *ctx) {
    hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->prof) {
        ProfEntry &pe = kv.second;
        for (size_t i = 0; i < pe.pending.size(); i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pe.pending[i].first, pe.pending[i].second) == hipSuccess) {
                pe.launches++;
                pe.ms += ms;
                pe.flops += pe.pending_work[i].first;
                pe.bytes += pe.pending_work[i].second;
            }
            ctx->event_pool.push_back(pe.pending[i].first);
            ctx->event_pool.push_back(pe.pending[i].second);
        }
        pe.pending.clear();
        pe.pending_work.clear();
    }
}

RTEN_EXPORT int32_t rten_hip_abi_version(void) { return RTEN_HIP_ABI_VERSION; }


RTEN_EXPORT int32_t rten_hip_init(int32_t device_id, void *external_stream, rten_hip_ctx **out_ctx) {
    if (!out_ctx) return RTEN_HIP_ERR_INVALID_VALUE;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RTEN_HIP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return RTEN_HIP_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return RTEN_HIP_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return RTEN_HIP_ERR_NO_DEVICE;
    // This library carries gfx950 code objects only.
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RTEN_HIP_ERR_NO_DEVICE;
    rten_hip_ctx *ctx = new rten_hip_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount;
    if (const char *dbg = getenv("RTEN_HIP_DEBUG")) ctx->debug = (int)strtoul(dbg, nullptr, 0); // (decimal or 0x...)
#ifndef RTEN_ABLATION // bits 24-31 select ablation instantiations (wrong results, timing only): they exist in -DRTEN_ABLATION builds only
    if ((unsigned)ctx->debug >> 24) { fprintf(stderr, "rten_hip: RTEN_HIP_DEBUG bits 24-31 (ablation kernels) are ignored: this library was not built with -DRTEN_ABLATION\n"); ctx->debug &= 0x00ffffff; }
#endif
    if (external_stream) {
        ctx->stream = (hipStream_t)external_stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return RTEN_HIP_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    // arrival counters of the split-K producers (gemm_f32.hip, split_finish): zeroed ON THE CONTEXT'S STREAM -- stream order puts the
    // memset before any producer that can arrive on them -- and left zero by every launch.  (No null-stream memset + device-wide
    // synchronise: creating a context must not stall the other contexts' work, nor break a capture active on this thread.)
    // A caller-provided stream that is CAPTURING would record that memset into the caller's graph instead of executing it (hipMalloc memory
    // is not zeroed: eager split-K launches would then find garbage counters): such a stream gets a blocking memset on the null stream with
    // this thread's capture mode relaxed for the duration.
    if (hipMalloc((void **)&ctx->split_counters, rten_hip_ctx::kSplitCounters * sizeof(unsigned)) == hipSuccess) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        const bool capturing = external_stream && hipStreamIsCapturing(ctx->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
        hipError_t e;
        if (capturing) {
            hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
            (void)hipThreadExchangeStreamCaptureMode(&mode);
            e = hipMemset(ctx->split_counters, 0, rten_hip_ctx::kSplitCounters * sizeof(unsigned));
            (void)hipThreadExchangeStreamCaptureMode(&mode);
        } else {
            e = hipMemsetAsync(ctx->split_counters, 0, rten_hip_ctx::kSplitCounters * sizeof(unsigned), ctx->stream);
        }
        if (e != hipSuccess) { hipFree(ctx->split_counters); ctx->split_counters = nullptr; }
    } else {
        ctx->split_counters = nullptr; // the fixup-kernel path needs none
    }
    (void)hipGetLastError();
    // sticky fault word: pinned, mapped; the device writes it only when a kernel gives up
    unsigned *fh = nullptr;
    if (hipHostMalloc((void **)&fh, 64, hipHostMallocMapped) == hipSuccess) {
        *fh = 0u;
        void *fd = nullptr;
        if (hipHostGetDevicePointer(&fd, fh, 0) == hipSuccess) { ctx->fault_host = fh; ctx->fault_dev = (unsigned *)fd; }
        else hipHostFree(fh);
    }
    (void)hipGetLastError();
    *out_ctx = ctx;
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_destroy(rten_hip_ctx *ctx) {
    if (!ctx) return RTEN_HIP_OK;
    hipSetDevice(ctx->device);
    // A capture nobody ended: the capture lock belongs to the thread that called rten_hip_graph_begin.  On that thread the capture is
    // aborted here (never delete a locked mutex); from any other thread -- a Rust `Drop` may run anywhere -- taking the lock would block
    // forever and releasing it would be undefined behaviour, so the context is refused (and leaked by a caller that ignores the status)
    // rather than deadlocked: end or abort the capture on its own thread first.
    if (ctx->capturing && ctx->capture_locks > 0) {
        if (ctx->capture_thread != std::this_thread::get_id()) {
            fprintf(stderr, "rten_hip_destroy: context %p is capturing on another thread; not destroyed\n", (void *)ctx);
            return RTEN_HIP_ERR_INVALID_VALUE;
        }
        rten_hip_graph_abort(ctx);
    }
    hipStreamSynchronize(ctx->stream);
    if (ctx->fault_host) hipHostFree((void *)ctx->fault_host);
    for (auto &kv : ctx->prof)
        for (auto &p : kv.second.pending) {
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
    for (hipEvent_t e : ctx->event_pool) hipEventDestroy(e);
    for (hipEvent_t e : ctx->sync_events) hipEventDestroy(e);
    for (auto &t : ctx->timers)
        for (hipEvent_t e : t)
            if (e) hipEventDestroy(e);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->aux) hipFree(ctx->aux);
    for (void *b : ctx->retired) hipFree(b);
    if (ctx->split_counters) hipFree(ctx->split_counters);
    for (auto &kv : ctx->luts) hipFree(kv.second);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return RTEN_HIP_OK;
}

RTEN_EXPORT const char *rten_hip_last_error(rten_hip_ctx *ctx) { return ctx ? tls_last_error.c_str() : "null context"; }

int32_t rten_check_fault(rten_hip_ctx *ctx) {
    if (ctx->fault_host && *ctx->fault_host)
        return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "a quantized-output launch gave up waiting for its grid (workgroups not all resident: the device was shared "
                                                      "with other work): every result since is void (NaN scale written); rten_hip_grid_sync_reset clears the fault (code %u)",
                              (unsigned)*ctx->fault_host);
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_sync(rten_hip_ctx *ctx) {
    RTEN_CHECK_CTX(ctx);
    RTEN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return rten_check_fault(ctx); // sticky: a kernel that gave up is reported by every sync until the fault is reset
}

RTEN_EXPORT int32_t rten_hip_device_id(const rten_hip_ctx *ctx) { return ctx ? ctx->device : -1; }

RTEN_EXPORT int32_t rten_hip_device_info(rten_hip_ctx *ctx, char *name_buf, int32_t name_len, int32_t *compute_units,
                                         int32_t *clock_mhz, int64_t *total_mem_bytes) {
    RTEN_CHECK_CTX(ctx);
    hipDeviceProp_t prop;
    RTEN_HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name_buf && name_len > 0) snprintf(name_buf, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (total_mem_bytes) *total_mem_bytes = (int64_t)prop.totalGlobalMem;
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_malloc(rten_hip_ctx *ctx, size_t bytes, void **out_dptr) {
    RTEN_CHECK_CTX(ctx);
    if (!out_dptr) return RTEN_HIP_ERR_INVALID_VALUE;
    *out_dptr = nullptr;
    if (bytes == 0) bytes = 16;
    RTEN_HIP_TRY(ctx, hipMalloc(out_dptr, bytes));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_free(rten_hip_ctx *ctx, void *dptr) {
    RTEN_CHECK_CTX(ctx);
    if (!dptr) return RTEN_HIP_OK;
    RTEN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    RTEN_HIP_TRY(ctx, hipFree(dptr));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_memcpy_h2d(rten_hip_ctx *ctx, void *dst, const void *src_host, size_t bytes) {
    RTEN_CHECK_CTX(ctx);
    if (bytes == 0) return RTEN_HIP_OK;
    // Pageable host memory: the async copy is staged by the runtime; sync so the caller may reuse src.
    RTEN_HIP_TRY(ctx, hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    RTEN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_memcpy_d2h(rten_hip_ctx *ctx, void *dst_host, const void *src, size_t bytes) {
    RTEN_CHECK_CTX(ctx);
    if (bytes == 0) return RTEN_HIP_OK;
    RTEN_HIP_TRY(ctx, hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RTEN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_memcpy_d2d(rten_hip_ctx *ctx, void *dst, const void *src, size_t bytes) {
    RTEN_CHECK_CTX(ctx);
    if (bytes == 0) return RTEN_HIP_OK;
    RTEN_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_memset(rten_hip_ctx *ctx, void *dst, int32_t byte_value, size_t bytes) {
    RTEN_CHECK_CTX(ctx);
    if (bytes == 0) return RTEN_HIP_OK;
    RTEN_HIP_TRY(ctx, hipMemsetAsync(dst, byte_value, bytes, ctx->stream));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_timer_start(rten_hip_ctx *ctx, int32_t slot) {
    RTEN_CHECK_CTX(ctx);
    if (slot < 0 || slot >= 64) return RTEN_HIP_ERR_INVALID_VALUE;
    for (int i = 0; i < 2; i++)
        if (!ctx->timers[slot][i]) RTEN_HIP_TRY(ctx, hipEventCreate(&ctx->timers[slot][i]));
    RTEN_HIP_TRY(ctx, hipEventRecord(ctx->timers[slot][0], ctx->stream));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_timer_stop(rten_hip_ctx *ctx, int32_t slot) {
    RTEN_CHECK_CTX(ctx);
    if (slot < 0 || slot >= 64 || !ctx->timers[slot][1]) return RTEN_HIP_ERR_INVALID_VALUE;
    RTEN_HIP_TRY(ctx, hipEventRecord(ctx->timers[slot][1], ctx->stream));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_timer_elapsed_ms(rten_hip_ctx *ctx, int32_t slot, float *out_ms) {
    RTEN_CHECK_CTX(ctx);
    if (slot < 0 || slot >= 64 || !ctx->timers[slot][1] || !out_ms) return RTEN_HIP_ERR_INVALID_VALUE;
    RTEN_HIP_TRY(ctx, hipEventSynchronize(ctx->timers[slot][1]));
    RTEN_HIP_TRY(ctx, hipEventElapsedTime(out_ms, ctx->timers[slot][0], ctx->timers[slot][1]));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_graph_begin(rten_hip_ctx *ctx) {
    RTEN_CHECK_CTX(ctx);
    if (ctx->capturing) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "graph capture already active");
    RTEN_HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    ctx->capture_thread = std::this_thread::get_id();
    // the capturing thread keeps the context until rten_hip_graph_end: launches of other host threads must not be
    // recorded into this graph (they block on the mutex instead)
    ctx->mu.lock();
    ctx->capture_locks++;
    return RTEN_HIP_OK;
}

// Ends the capture on EVERY path (argument errors and instantiation failures included): the capture lock taken by graph_begin is
// released and the stream leaves capture mode, so a failed capture cannot leave the context locked for its other threads.
// Must be called by the thread that called rten_hip_graph_begin (the lock is owned by it).
RTEN_EXPORT int32_t rten_hip_graph_end(rten_hip_ctx *ctx, uint64_t *out_graph) {
    RTEN_CHECK_CTX(ctx);
    if (!ctx->capturing) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "no active capture");
    ctx->capturing = false;
    if (ctx->capture_locks > 0) { ctx->capture_locks--; ctx->mu.unlock(); } // the guard of this call still holds it
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(ctx->stream, &graph);
    if (ec != hipSuccess) return rten_check_hip(ctx, ec, "hipStreamEndCapture");
    if (!out_graph) {
        if (graph) hipGraphDestroy(graph);
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "graph_end: out_graph is NULL (the capture was ended and dropped)");
    }
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return rten_check_hip(ctx, e, "hipGraphInstantiate");
    *out_graph = (uint64_t)(uintptr_t)exec;
    ctx->live_graphs++; // its launches hold this context's scratch / aux pointers (rten_scratch)
    return RTEN_HIP_OK;
}

// Abandons an active capture (an operator failed while capturing and the caller gives up): ends it, drops the recorded graph,
// releases the capture lock.  No-op without an active capture.  Same-thread rule as rten_hip_graph_end.
RTEN_EXPORT int32_t rten_hip_graph_abort(rten_hip_ctx *ctx) {
    RTEN_CHECK_CTX(ctx);
    if (!ctx->capturing) return RTEN_HIP_OK;
    ctx->capturing = false;
    if (ctx->capture_locks > 0) { ctx->capture_locks--; ctx->mu.unlock(); }
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph); // an invalidated capture reports an error here: it is over either way
    if (graph) hipGraphDestroy(graph);
    (void)e;
    (void)hipGetLastError();
    return RTEN_HIP_OK;
}

// Cross-stream ordering between two contexts on the same device: work enqueued on `waiter` after this call runs
// after everything enqueued on `signaler` so far.  During graph capture this is how a second context joins (and
// later re-joins) the capturing context's graph, giving parallel branches.
RTEN_EXPORT int32_t rten_hip_stream_wait(rten_hip_ctx *waiter, rten_hip_ctx *signaler) {
    if (!waiter) return RTEN_HIP_ERR_INVALID_VALUE;
    if (!signaler || signaler->device != waiter->device)
        return rten_set_error(waiter, RTEN_HIP_ERR_INVALID_VALUE, "stream_wait: contexts must share a device");
    if (waiter == signaler) return RTEN_HIP_OK;
    std::unique_lock<std::recursive_mutex> lk_w(waiter->mu, std::defer_lock), lk_s(signaler->mu, std::defer_lock);
    std::lock(lk_w, lk_s); // both contexts' capture flags are touched below
    rten_bind_device(waiter);
    if (waiter->sync_events.size() < 64) {
        hipEvent_t e = nullptr;
        RTEN_HIP_TRY(waiter, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        waiter->sync_events.push_back(e);
    }
    hipEvent_t ev = waiter->sync_events[waiter->sync_next++ % waiter->sync_events.size()];
    RTEN_HIP_TRY(waiter, hipEventRecord(ev, signaler->stream));
    RTEN_HIP_TRY(waiter, hipStreamWaitEvent(waiter->stream, ev, 0));
    // capture bookkeeping: a context forked from a capturing one must not allocate either; the join hands it back
    if (signaler->capturing && !waiter->capturing) { waiter->capturing = true; waiter->capture_origin = signaler->capture_origin ? signaler->capture_origin : signaler; }
    else if (signaler->capturing && waiter->capturing && signaler->capture_origin == waiter) { signaler->capturing = false; signaler->capture_origin = nullptr; }
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_graph_launch(rten_hip_ctx *ctx, uint64_t graph) {
    RTEN_CHECK_CTX(ctx);
    if (!graph) return RTEN_HIP_ERR_INVALID_VALUE;
    if (const int32_t rc = rten_check_fault(ctx)) return rc;
    RTEN_HIP_TRY(ctx, hipGraphLaunch((hipGraphExec_t)(uintptr_t)graph, ctx->stream));
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_graph_destroy(rten_hip_ctx *ctx, uint64_t graph) {
    RTEN_CHECK_CTX(ctx);
    if (!graph) return RTEN_HIP_OK;
    RTEN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    RTEN_HIP_TRY(ctx, hipGraphExecDestroy((hipGraphExec_t)(uintptr_t)graph));
    if (ctx->live_graphs > 0 && --ctx->live_graphs == 0) { // nothing replays from the retired scratch buffers any more
        for (void *b : ctx->retired) hipFree(b);
        ctx->retired.clear();
    }
    return RTEN_HIP_OK;
}

// Snapshot / restore of every sticky tuning knob of a context (GEMM variant override, split-K plan, tile order, gemv order + thread
// assumption, int8 path, attention path): a library-level caller that changes knobs around its own launches -- the plan executor on a
// BORROWED context -- puts back what the owner had set instead of the defaults.
RTEN_EXPORT int32_t rten_hip_tuning_save(rten_hip_ctx *ctx, int32_t state[8]) {
    RTEN_CHECK_CTX(ctx);
    if (!state) return RTEN_HIP_ERR_INVALID_VALUE;
    state[0] = ctx->gemm_variant_override; state[1] = ctx->split_mode; state[2] = ctx->split_s; state[3] = ctx->tile_order;
    state[4] = ctx->gemv_order; state[5] = (int32_t)ctx->gemv_threads; state[6] = ctx->int8_path; state[7] = ctx->sdpa_path;
    return RTEN_HIP_OK;
}
RTEN_EXPORT int32_t rten_hip_tuning_restore(rten_hip_ctx *ctx, const int32_t state[8]) {
    RTEN_CHECK_CTX(ctx);
    if (!state) return RTEN_HIP_ERR_INVALID_VALUE;
    // every knob is put back even if one setter objects (the values were valid when they were saved; a borrowed context must not be left half restored):
    // the first error is what the call returns
    int32_t first = RTEN_HIP_OK;
    auto keep = [&](int32_t rc) { if (rc != RTEN_HIP_OK && first == RTEN_HIP_OK) first = rc; };
    keep(rten_hip_set_gemm_variant_override(ctx, state[0]));
    keep(rten_hip_set_gemm_split(ctx, state[1], state[2]));
    keep(rten_hip_set_gemm_order(ctx, state[3]));
    keep(rten_hip_set_gemv_order(ctx, state[4], state[5]));
    ctx->int8_path = state[6];
    ctx->sdpa_path = state[7];
    return first;
}

RTEN_EXPORT int32_t rten_hip_profile_enable(rten_hip_ctx *ctx, int32_t on) {
    RTEN_CHECK_CTX(ctx);
    ctx->profiling = on != 0;
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_profile_reset(rten_hip_ctx *ctx) {
    RTEN_CHECK_CTX(ctx);
    prof_resolve(ctx);
    ctx->prof.clear();
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_profile_report(rten_hip_ctx *ctx, char *buf, int32_t buf_len) {
    RTEN_CHECK_CTX(ctx);
    if (!buf || buf_len <= 2) return RTEN_HIP_ERR_INVALID_VALUE;
    prof_resolve(ctx);
    std::string s = "[";
    bool first = true;
    for (auto &kv : ctx->prof) {
        char line[512];
        snprintf(line, sizeof line, "%s{\"kernel\":\"%s\",\"launches\":%d,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}",
                 first ? "" : ",", kv.first.c_str(), kv.second.launches, kv.second.ms, kv.second.flops,
                 kv.second.bytes);
        s += line;
        first = false;
    }
    s += "]";
    if ((int)s.size() + 1 > buf_len) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "report buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return RTEN_HIP_OK;
}

// calc_output_size_and_padding -- src/ops/pooling.rs:63-159.  Host-side shape logic shared by the
// conv and pooling operators; error strings are the reference's, verbatim.
static const char *axis_out_pad(int64_t in, int64_t k, int64_t stride, bool same, int64_t ps, int64_t pe, int64_t dil,
                                bool ceil_mode, int32_t *out, int32_t *pad_s, int32_t *pad_e) {
    if (dil <= 0) return "Dilations must be > 0";
    if (k <= 0) return "Kernel size must be > 0";
    if (stride <= 0) return "Strides must be > 0";
    if (same) {
        int64_t o = (in + stride - 1) / stride;
        int64_t need = (o - 1) * stride + (k - 1) * dil + 1;
        int64_t total = need > in ? need - in : 0;
        *out = (int32_t)o;
        *pad_s = (int32_t)(total / 2);
        *pad_e = (int32_t)((total + 1) / 2);
        return nullptr;
    }
    int64_t padded = in + ps + pe;
    int64_t dk = k + (k - 1) * (dil - 1);
    if (padded < dk) return "Input too small for kernel size";
    int64_t win = padded - dil * (k - 1) - 1;
    int64_t o = ceil_mode ? (win + stride - 1) / stride + 1 : win / stride + 1;
    if (ceil_mode && (o - 1) * stride >= in + ps) o -= 1;
    *out = (int32_t)o;
    *pad_s = (int32_t)ps;
    *pad_e = (int32_t)pe;
    return nullptr;
}

RTEN_EXPORT int32_t rten_hip_calc_output_size_and_padding(int32_t in_h, int32_t in_w, int32_t k_h, int32_t k_w,
                                                          int32_t stride_h, int32_t stride_w, int32_t same_padding,
                                                          const int32_t pads[4], int32_t dil_h, int32_t dil_w,
                                                          int32_t ceil_mode, int32_t out_hw[2], int32_t out_pads[4],
                                                          const char **err_msg) {
    if (!out_hw || !out_pads) return RTEN_HIP_ERR_INVALID_VALUE;
    if (!same_padding && !pads) return RTEN_HIP_ERR_INVALID_VALUE;
    int64_t pt = same_padding ? 0 : pads[0], pl = same_padding ? 0 : pads[1];
    int64_t pb = same_padding ? 0 : pads[2], pr = same_padding ? 0 : pads[3];
    const char *e = axis_out_pad(in_h, k_h, stride_h, same_padding != 0, pt, pb, dil_h, ceil_mode != 0, &out_hw[0],
                                 &out_pads[0], &out_pads[2]);
    if (!e)
        e = axis_out_pad(in_w, k_w, stride_w, same_padding != 0, pl, pr, dil_w, ceil_mode != 0, &out_hw[1],
                         &out_pads[1], &out_pads[3]);
    if (err_msg) *err_msg = e;
    return e ? RTEN_HIP_ERR_INVALID_VALUE : RTEN_HIP_OK;
}
