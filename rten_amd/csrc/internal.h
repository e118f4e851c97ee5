// Internal declarations shared by the translation units of librten_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/rten_hip.h"

#define RTEN_EXPORT extern "C" __attribute__((visibility("default")))

struct ProfEntry {
    int launches = 0;
    double ms = 0.0;
    double flops = 0.0;
    double bytes = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; // events not yet resolved
    std::vector<std::pair<double, double>> pending_work;    // (flops, bytes) per pending launch
};

struct rten_hip_ctx {
    // Serialises host-side entry into the context (rten_hip.h, "Thread safety"): every exported function takes it for
    // the duration of the call; recursive because entry points call each other (conv -> gemm, graph capture holds it).
    std::recursive_mutex mu;
    int capture_locks = 0; // times rten_hip_graph_begin locked `mu` on behalf of the capturing thread
    std::thread::id capture_thread; // the thread that owns those locks (only it may end / abort the capture)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t timers[64][2] = {};
    bool capturing = false;
    rten_hip_ctx *capture_origin = nullptr;  // set on contexts that joined another context's capture (stream_wait)
    std::vector<hipEvent_t> sync_events;     // cross-context ordering events (round robin)
    size_t sync_next = 0;
    // profiling
    bool profiling = false;
    std::map<std::string, ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
    // scratch (grown on demand, never during capture)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *aux = nullptr; // second grow-only buffer for operators that call the GEMM (which owns `scratch`) on an intermediate of their own
    size_t aux_bytes = 0;
    int live_graphs = 0;          // executable graphs captured on this context and not yet destroyed (they hold scratch / aux pointers)
    std::vector<void *> retired;  // scratch / aux buffers replaced while a graph was alive: freed with the last graph or the context
    std::map<std::string, void *> luts; // im2col lookup tables, keyed by conv geometry (gemm_f32.hip)
    int gemm_variant_override = -1;
    int pipeline = 1; // conv paths: 0 register-staged, 1 LDS-DMA, 2 LDS-DMA + wave specialisation, 3 four stages, 4 fragments first, 5 16x16x4 MFMAs, 6 one wave per 64x64 tile (gemm_f32_wave.hip)
    int wave_flavour = 0; // pipeline 6: 0 = k-tiles of 16 x 2 LDS stages, 1 = 8 x 4, 2 = 16 x 3
    int num_cus = 256;
    int sdpa_path = 0;  // 0 automatic (fused attention kernel when it covers the shape), 1 composed path only
    int int8_path = 0;  // 0 automatic (fast staging path when it covers the call), 1 generic kernel only
    int int8_tile = -1; // rten_hip_set_int8_tile: -1 = per-shape rule, 0..3 = 128x128 / 128x64 / 64x128 / 64x64
    int tile_order = 0; // workgroup -> tile order bits (rten_hip_set_gemm_order)
    int split_mode = 3, split_s = 1; // exact split-K plan: 0 off, 1 tail tiles, 2 all tiles, 3 automatic (gemm_f32.hip)
    int gemv_order = 1;          // m == 1 products of rten_hip_gemm_f32 follow the reference's gemv kernels (gemv_f32.hip); 0: the blocked order
    long long gemv_threads = 0;  // reference thread count assumed for its column blocks (0: at least n / 128)
    static constexpr long long kSplitCounters = 1 << 16;
    unsigned *split_counters = nullptr; // arrival counters of the split-K producers (zero between launches), allocated with the context
    // Sticky device-fault word (pinned host memory mapped into the device): a kernel that has to give up -- today: a quantized-output launch whose
    // grid-wide exchange timed out because its workgroups were not all resident (int8_fast.hip) -- stores a non-zero code here; the next
    // rten_hip_sync / rten_hip_graph_launch on the context fails with it until rten_hip_grid_sync_reset clears it.  Reading it costs the host nothing.
    volatile unsigned *fault_host = nullptr;
    unsigned *fault_dev = nullptr;
    int debug = 0; // RTEN_HIP_DEBUG ablation bits (tuning only)
};

// Padding value semantics of an integer convolution.  Depthwise geometries (groups == C == O, not the groups == 1 pointwise
// case) run the reference's depthwise kernel (conv.rs:269-284 -> conv/depthwise.rs:148-190), which skips padded taps -- they
// contribute 0 whatever the platform's im2col quirk is -- so the requested pad mode is overridden for them.
inline int rten_effective_pad_mode(const rten_hip_conv2d_int8_desc *di) {
    const rten_hip_conv2d_desc &d = di->conv;
    const bool pw = d.kh == 1 && d.kw == 1 && d.groups == 1 && d.stride_h == 1 && d.stride_w == 1 && d.dil_h == 1 && d.dil_w == 1 && d.pads[0] == 0 &&
                    d.pads[1] == 0 && d.pads[2] == 0 && d.pads[3] == 0;
    return (!pw && d.c == d.o && d.groups == d.c) ? RTEN_HIP_PAD_ZERO_POINT : di->pad_mode;
}

int32_t rten_set_error(rten_hip_ctx *ctx, int32_t code, const char *fmt, ...);
int32_t rten_check_hip(rten_hip_ctx *ctx, hipError_t e, const char *what);
void *rten_scratch(rten_hip_ctx *ctx, size_t bytes);
// non-zero: the context's sticky device fault as an error (see rten_hip_ctx::fault_host)
int32_t rten_check_fault(rten_hip_ctx *ctx);
// gemv_f32.hip: the m == 1 product in the reference's gemv order (called by rten_hip_gemm_f32)
int32_t rten_gemv_f32(rten_hip_ctx *ctx, const rten_hip_gemm_desc *d, const float *a, const float *b, const float *bias, float *c);
// gemm_f32.hip: rten_hip_gemm_f32 without the gemv dispatch (operators whose reference form is not a gemm_impl call on unpacked operands)
int32_t rten_gemm_f32_blocked(rten_hip_ctx *ctx, const rten_hip_gemm_desc *d, const float *a, const float *b, const float *bias, float *c);
void *rten_aux_scratch(rten_hip_ctx *ctx, size_t bytes);
// gemm_f32_wave.hip: one launch of the wave-tile kernels; `args` = the caller's GemmArgs (gemm_f32_common.h)
int32_t rten_launch_gemm_f32_wave(rten_hip_ctx *ctx, const void *args, unsigned grid_x, unsigned grid_z, int bl, int mode, int flavour);
// gemm_f32_patch.hip: 3x3 / stride 1 / padding 1 convolutions with B staged as image patches; RTEN_HIP_ERR_UNSUPPORTED when the launch is not covered
int32_t rten_launch_gemm_f32_patch(rten_hip_ctx *ctx, const void *args, unsigned grid_x, unsigned grid_z, int mode);

// Profiling bracket around one kernel launch.
struct ProfScope {
    rten_hip_ctx *ctx;
    const char *name;
    double flops, bytes;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(rten_hip_ctx *c, const char *n, double fl, double by);
    ~ProfScope();
};

// Binds the context's device to the calling host thread (the current device is per thread in HIP: a second
// Model::run thread starts on device 0).
inline void rten_bind_device(const rten_hip_ctx *ctx) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) hipSetDevice(ctx->device);
}

// First statement of every entry point that takes a context: NULL check, lock for the whole call, device binding.
#define RTEN_CHECK_CTX(ctx)                                                  \
    if (!(ctx)) return RTEN_HIP_ERR_INVALID_VALUE;                           \
    std::lock_guard<std::recursive_mutex> rten_ctx_guard_((ctx)->mu);        \
    rten_bind_device(ctx)

#define RTEN_HIP_TRY(ctx, expr)                                  \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return rten_check_hip(ctx, _e, #expr); \
    } while (0)

#define RTEN_LAUNCH_CHECK(ctx, what)                             \
    do {                                                         \
        hipError_t _e = hipGetLastError();                       \
        if (_e != hipSuccess) return rten_check_hip(ctx, _e, what); \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Exact division by a run-time divisor as multiply + shift.  q = n / d for 0 <= n < 2^L as (n * mul) >> shift with mul = ceil(2^(L+s) / d),
// s = ceil(log2 d): the error term n * (mul * d - 2^(L+s)) stays below 2^(L+s), so the quotient is exact on the whole range (Granlund &
// Montgomery, "Division by invariant integers using multiplication").  The kernels' prologues decode tile / pixel indices with run-time
// divisors; the compiler's integer division is ~40 VALU instructions per quotient, paid by every wave of every workgroup.
struct RtenDiv { unsigned mul; int shift; };
inline RtenDiv rten_make_div(long long n_max, long long d) {
    if (d < 1) d = 1;
    if (n_max < 1) n_max = 1;
    int L = 1;
    while (L < 31 && ((long long)1 << L) <= n_max) L++;
    int sft = 0;
    while (((long long)1 << sft) < d) sft++;
    RtenDiv r;
    if (sft > 31) { r.mul = 0; r.shift = 0; return r; } // divisor above every n in range: quotient 0
    const unsigned long long two = (unsigned long long)1 << (L + sft);
    r.mul = (unsigned)((two + (unsigned long long)d - 1) / (unsigned long long)d);
    r.shift = L + sft;
    return r;
}
__host__ __device__ __forceinline__ int rten_div(int n, const RtenDiv &d) { return (int)(((unsigned long long)(unsigned)n * d.mul) >> d.shift); }

// All cache lines of the kernel-argument segment requested at once, one wait.  The compiler fetches a large by-value argument struct piecemeal, field
// groups where they are first used, each group a dependent scalar-cache miss (five in a row in the int8 convolution kernels: ~2400 cycles before the first
// address can be formed -- tools/debug/i8_trace.py); behind this prefetch its loads hit the scalar cache.  It matters where a launch is ONE wave of workgroups
// (the int8 graph, batch-1 latency): nothing else hides a prologue there.  BYTES = size of the explicit arguments: every line requested lies inside them.
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
    constexpr int NLINES = (BYTES + 63) / 64;
    static_assert(NLINES == 3 || NLINES == 5 || NLINES == 7, "argument blocks of 3, 5 or 7 cache lines");
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned d0, d1, d2, d3, d4, d5, d6;
    if constexpr (NLINES == 7) {
        asm volatile("s_load_dword %0, %7, 0x0\n\t"
                     "s_load_dword %1, %7, 0x40\n\t"
                     "s_load_dword %2, %7, 0x80\n\t"
                     "s_load_dword %3, %7, 0xc0\n\t"
                     "s_load_dword %4, %7, 0x100\n\t"
                     "s_load_dword %5, %7, 0x140\n\t"
                     "s_load_dword %6, %7, 0x180\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6)
                     : "s"(ka)
                     : "memory");
    } else if constexpr (NLINES == 5) {
        asm volatile("s_load_dword %0, %5, 0x0\n\t"
                     "s_load_dword %1, %5, 0x40\n\t"
                     "s_load_dword %2, %5, 0x80\n\t"
                     "s_load_dword %3, %5, 0xc0\n\t"
                     "s_load_dword %4, %5, 0x100\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4)
                     : "s"(ka)
                     : "memory");
    } else {
        asm volatile("s_load_dword %0, %3, 0x0\n\t"
                     "s_load_dword %1, %3, 0x40\n\t"
                     "s_load_dword %2, %3, 0x80\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2)
                     : "s"(ka)
                     : "memory");
    }
}
