// Shared pieces of the f32 GEMM / implicit-GEMM translation units (gemm_f32.hip: the 4-wave tile kernels and every host-side launch plan;
// gemm_f32_wave.hip: the one-wave-per-tile kernel family): kernel-argument block, buffer-load helpers, the reference's fold / epilogue
// (rten-gemm/src/lib.rs:1008-1013,1221-1255; simd_generic.rs:378-414), the exact split-K last-arrival fold and the -DRTEN_TRACE stamps.
// Everything here has internal linkage (anonymous namespace): each translation unit compiles its own copy.
#pragma once
#include "internal.h"
#include "vecmath.h"

#ifdef RTEN_TRACE
struct RtenTraceHost { unsigned long long *buf = nullptr; unsigned cap = 0, next = 0; }; // 16 x u64 per record; `next`: slots handed out so far
extern RtenTraceHost g_trace_host; // defined in gemm_f32.hip (a captured launch keeps its slots for every replay)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 16;             // k-tile depth
constexpr int KC_TILES = 256 / BK; // reference depth block (kc = 256 for f32)
constexpr int NTHREADS = 256;
// Ablation switches (RTEN_HIP_DEBUG bits) exist only in -DRTEN_ABLATE tuning builds: a runtime test inside the K loop
// costs the production kernels accumulator copies and exec-mask branches.
#ifdef RTEN_ABLATE
#define ABLATE(p) ((p).debug)
#else
#define ABLATE(p) 0
#endif
constexpr unsigned OOB = 0x80000000u; // byte offset beyond every buffer (< 2 GiB): buffer loads return 0

#ifdef RTEN_TRACE
// -DRTEN_TRACE builds only (tools/debug/f32_trace.py): every workgroup of the LDS-DMA kernel appends one record of wall-clock stamps
// (s_memrealtime, 100 MHz, the same counter on every XCD) at its phase boundaries, with its compute unit.  Compiled out of the product build.
#define TR_DECL unsigned long long tr_t[6] = {0, 0, 0, 0, 0, 0}; unsigned tr_trips = 0; const unsigned long long tr_c0 = __builtin_readcyclecounter(), tr_r0 = __builtin_amdgcn_s_memrealtime();
#define TR_STAMP(i) tr_t[i] = __builtin_amdgcn_s_memrealtime();
#else
#define TR_DECL
#define TR_STAMP(i)
#endif

enum ALoad { A_M4 = 0, A_K4 = 1, A_SCALAR = 2 };
enum BLoad { B_N4 = 0, B_K4 = 1, B_SCALAR = 2, B_IM2COL = 3, B_IM2COL_TAPS = 4 }; // TAPS: <= 31 kernel taps, per-lane validity bitmask

struct GemmArgs {
    const float *A;
    const float *B;
    float *C;
    const float *bias;
    const float *res;
    const i32x2 *lut; // im2col: per k {element offset c*HW + ky*dy*W + kx*dx, (ky*dy) | (kx*dx) << 16}; padded rows fail the bounds test
    int M, N, K;
    long long a_rs, a_cs, a_bs;       // A[z*a_bs + m*a_rs + k*a_cs]
    long long b_rs, b_cs, b_ns, b_bs; // B[z*b_bs + k*b_rs + (n/Pn)*b_ns + (n%Pn)*b_cs]
    long long c_rs, c_ns, c_bs;       // C[z*c_bs + m*c_rs + (n/Pn)*c_ns + (n%Pn)]
    long long bias_bs;
    long long a_bsi, b_bsi, c_bsi;    // inner batch strides
    unsigned a_bytes, b_bytes;        // extent (bytes) of one batch slice of A / B from its base: buffer num_records
    int batch_inner;                  // z -> (z / batch_inner, z % batch_inner); <= 1: single level
    int Pn;
    float alpha, beta;
    int bias_kind, act;
    int tiles_m, tiles_n;
    int a_dir_m, b_dir_n; // scalar loaders: lanes run along m / n (1) or along k (0)
    int H, W, OW, sy, sx, pt, pl; // im2col geometry
    int KH, KW, dy, dx;           // kernel taps / dilation (B_IM2COL_TAPS validity masks)
    int debug; // ablation switches for tuning runs (RTEN_HIP_DEBUG): 1 = skip the in-loop DMA, 2 = skip the MFMAs, 4 = skip the epilogue
    // exact split-K (LDS-DMA kernel, MODE 2): tiles >= split_t1 are cut along K at depth-block (kc) boundaries into
    // split_s groups of split_g blocks; each block's raw accumulator is parked in slab slot `blk` of its tile and
    // the fixup kernel replays the unsplit fold over the slots in block order -> bit-identical to the unsplit chain.
    float *slab;
    unsigned *split_counters; // arrival counters, one per split tile (zero between launches); NULL: the fixup kernel folds
    int split_t1, split_s, split_g, split_slots, split_ntail;
    int order; // bit 0: tiles walk n fastest (default m fastest); bit 1: split workgroups walk tiles fastest, K groups slowest
    int n_lo;  // thin-tile kernel: first column of its share (the whole-round tiles of the same call cover [0, n_lo))
#ifdef RTEN_TRACE
    unsigned long long *trace_buf; // NULL: off
    unsigned trace_cap;
    unsigned trace_base; // first record slot of this launch (host counter: a launch's workgroups own slots base + blockIdx)
    unsigned long long trace_pad[8]; // (the argument block then spans 7 cache lines: kernarg_prefetch takes 3, 5 or 7)
#endif
};


__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// Coherent (sc0 sc1) accesses: the store writes through to memory, the load bypasses this XCD's L2.  Used for the split-K slab
// when the tile is folded in the same launch by a workgroup that may sit on another XCD (split_finish): the eight L2s are not
// coherent with each other inside a kernel, and fencing instead (L2 write-back + invalidate per workgroup) throws away the
// weights and activations every other workgroup has cached -- measured: the whole forward pass 2.8 -> 3.6 ms.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void coherent_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 17);
}
__device__ __forceinline__ f32x4 coherent_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 17));
}


// ---- fold / epilogue helpers shared by all kernels.  Every uniform condition (alpha/beta form, bias kind,
// residual, activation) is tested ONCE per 32x32 accumulator block around straight-line code, and the block's 16
// loads are issued back to back before the first use -- a per-element chain of uniform branches serialises every
// load behind an s_waitcnt vmcnt(0).
// Row of accumulator register r inside a 32x32 MFMA block (lanes 32..63 sit 4 rows lower: part of the lane's base).
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// First depth block: out = alpha*acc + beta*C, then the bias (rten-gemm/src/lib.rs:1008-1013,1221-1255;
// the four store forms of simd_generic.rs:378-414).  `out` may alias `acc`.
template <int TM, int TN>
__device__ __forceinline__ void fold_first(const GemmArgs &p, int z, f32x16 (&acc)[TM][TN], f32x16 (&out)[TM][TN], int mb, int nb0,
                                           long long c_zoff) {
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void *)((p.bias ? p.bias : p.C) + (long long)z * p.bias_bs), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + c_zoff), 0, 0x7ffffffc, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mrow = mb + i * 32;
        float brow[16];
        if (p.bias_kind == RTEN_HIP_BIAS_PER_ROW) {
#pragma unroll
            for (int r = 0; r < 16; r++) brow[r] = buf_load1(rsBias, mrow + acc_row(r) < p.M ? (unsigned)(mrow + acc_row(r)) << 2 : OOB, 0);
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = nb0 + j * 32;
            const bool cok = n < p.N;
            f32x16 v = acc[i][j];
            if (p.beta == 0.f) {
                if (p.alpha != 1.f) {
#pragma unroll
                    for (int r = 0; r < 16; r++) v[r] = v[r] * p.alpha;
                }
            } else {
                const int nn = cok ? n : 0;
                const int nb = nn / p.Pn, np = nn - nb * p.Pn;
                const unsigned col = (unsigned)((long long)nb * p.c_ns + np);
                float cin[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mrow + acc_row(r);
                    cin[r] = buf_load1(rsC, (m < p.M && cok) ? (col + (unsigned)m * (unsigned)p.c_rs) << 2 : OOB, 0);
                }
                if (p.beta == 1.f && p.alpha == 1.f) {
#pragma unroll
                    for (int r = 0; r < 16; r++) v[r] = cin[r] + v[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) v[r] = vm::fma(v[r], p.alpha, cin[r] * p.beta);
                }
            }
            if (p.bias_kind == RTEN_HIP_BIAS_PER_ROW) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = v[r] + brow[r];
            } else if (p.bias_kind == RTEN_HIP_BIAS_PER_COL) {
                const float bcol = buf_load1(rsBias, cok ? (unsigned)n << 2 : OOB, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = v[r] + bcol;
            }
            out[i][j] = v;
        }
    }
}

// Later depth blocks: tot = tot + alpha*acc with the reference's beta = 1 store forms (lib.rs:1008-1013).
template <int TM, int TN>
__device__ __forceinline__ void fold_next(const GemmArgs &p, f32x16 (&acc)[TM][TN], f32x16 (&tot)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            if (p.alpha == 1.f) {
#pragma unroll
                for (int r = 0; r < 16; r++) tot[i][j][r] = tot[i][j][r] + acc[i][j][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) tot[i][j][r] = vm::fma(acc[i][j][r], p.alpha, tot[i][j][r]);
            }
        }
}

// Residual Add, activation and the NCHW / row-major store of finished values.  Buffer loads/stores with 32-bit
// offsets: the lane part (column, first row of the block) is one VGPR per block, the register's row rides in the
// scalar offset; rows >= M / columns >= N get an out-of-range lane offset (store dropped, load returns 0).
template <int TM, int TN>
__device__ __forceinline__ void store_out(const GemmArgs &p, f32x16 (&val)[TM][TN], int mb, int nb0, long long c_zoff) {
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(p.C + c_zoff), 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)((p.res ? p.res : p.C) + c_zoff), 0, 0x7ffffffc, 0x00020000);
    const bool has_res = p.res != nullptr;
    const unsigned rs4 = (unsigned)p.c_rs << 2;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = nb0 + j * 32;
        const bool cok = n < p.N;
        const int nn = cok ? n : 0;
        const int nb = nn / p.Pn, np = nn - nb * p.Pn;
        const unsigned col = (unsigned)((long long)nb * p.c_ns + np);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mrow = mb + i * 32;
            const unsigned base = cok ? (col + (unsigned)mrow * (unsigned)p.c_rs) << 2 : OOB;
            f32x16 v = val[i][j];
            unsigned voff[16];
#pragma unroll
            for (int r = 0; r < 16; r++) voff[r] = mrow < p.M - acc_row(r) ? base : OOB;
            if (has_res) {
                float rr[16];
#pragma unroll
                for (int r = 0; r < 16; r++) rr[r] = buf_load1(rsR, voff[r], (unsigned)acc_row(r) * rs4);
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = v[r] + rr[r];
            }
            if (p.act == RTEN_HIP_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = vm::relu(v[r]);
            } else if (p.act == RTEN_HIP_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = vm::gelu(v[r]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float x = v[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rsC, (int)voff[r], (int)((unsigned)acc_row(r) * rs4), 0);
            }
        }
    }
}


// Split-K, last arrival folds: a producer workgroup that has parked its depth blocks in the slab announces itself on the
// tile's counter; the workgroup that finds all the others already there replays the unsplit fold over the slots in
// depth-block order (exactly what igemm_f32_fixup_kernel does: first block beta * C + bias, later blocks separate adds, then
// the shared epilogue) and clears the counter for the next launch.  Which workgroup arrives last varies from run to run;
// the order of the additions does not.  Visibility across the eight XCDs' L2s: the slab is written with write-through stores and
// read back with L2-bypassing loads (coherent_store4 / coherent_load4); a wave's stores are acknowledged (vmcnt 0) before its
// workgroup arrives on the counter, an agent-scope atomic.  No cache-wide fence.
// Saves the fixup launch (ResNet-50 batch 1: 34 of 90 launches) and its dependency gap.  `flag` = one LDS word.
template <int BM, int BN, int TM, int TN, int WM = 2, int WN = 2> // WM x WN waves per workgroup (1 x 1: the one-wave tile kernels, wq == 0)
__device__ __forceinline__ void split_finish(const GemmArgs &p, int z, int tile, int wq, int lane, int m0, int n0, long long c_zoff, int *flag) {
    const int ti = tile - p.split_t1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through slab stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.split_counters + (long long)z * p.split_ntail + ti, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == (unsigned)p.split_s - 1u;
    }
    __syncthreads();
    if (!*flag) return;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wq / WN) * (BM / WM), wn0 = (wq % WN) * (BN / WN);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.slab + ((long long)z * p.split_ntail + ti) * p.split_slots * (long long)(BM * BN)), 0, (int)((unsigned)p.split_slots * (BM * BN) * 4u), 0x00020000);
    const unsigned loff = (unsigned)(wq * (TM * TN * 16 * 64) + lane * 4) * 4u;
    constexpr int U = TM * TN >= 2 ? 1 : 2; // slots fetched per batch: kept small, the fold shares the producer kernel's register budget (occupancy): their loads are all in flight before the first fold
    f32x16 acc[U][TM][TN], tot[TM][TN];
    auto load_raw = [&](f32x16 (&v)[TM][TN], int slot) {
        const unsigned b = loff + (unsigned)slot * (BM * BN) * 4u;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x4 o = coherent_load4(rs, b + (unsigned)(((i * TN + j) * 4 + q) * 256) * 4u);
                    v[i][j][4 * q] = o[0]; v[i][j][4 * q + 1] = o[1]; v[i][j][4 * q + 2] = o[2]; v[i][j][4 * q + 3] = o[3];
                }
    };
    const int mb = m0 + wm0 + 4 * half, nb0 = n0 + wn0 + l31;
    // slot index past the end: the buffer's range check returns zeros and the value is never folded
#pragma unroll
    for (int u = 0; u < U; u++) load_raw(acc[u], u);
    fold_first<TM, TN>(p, z, acc[0], tot, mb, nb0, c_zoff);
#pragma unroll
    for (int u = 1; u < U; u++)
        if (u < p.split_slots) fold_next<TM, TN>(p, acc[u], tot);
    for (int s = U; s < p.split_slots; s += U) {
#pragma unroll
        for (int u = 0; u < U; u++) load_raw(acc[u], s + u);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (s + u < p.split_slots) fold_next<TM, TN>(p, acc[u], tot);
    }
    store_out<TM, TN>(p, tot, mb, nb0, c_zoff);
    if (threadIdx.x == 0) p.split_counters[(long long)z * p.split_ntail + ti] = 0u;
}

#ifdef RTEN_TRACE
__device__ __forceinline__ void trace_write(const GemmArgs &p, unsigned kid, int tile, int grp, unsigned trips, const unsigned long long (&t)[6], unsigned long long cycles, unsigned long long ticks) {
    if (threadIdx.x != 0 || p.trace_buf == nullptr) return;
    const unsigned i = p.trace_base + blockIdx.y * gridDim.x + blockIdx.x; // no shared counter: 130k same-address atomics per step serialise (first version: 2.7 -> 11.7 ms)
    if (i >= p.trace_cap) return;
    unsigned long long *r = p.trace_buf + (unsigned long long)i * 16;
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    r[0] = kid | ((unsigned long long)blockIdx.x << 32);
    r[1] = hwid | ((unsigned long long)xcc << 32);
    for (int k = 0; k < 6; k++) r[2 + k] = t[k];
    r[8] = (unsigned)p.M | ((unsigned long long)(unsigned)p.K << 32);
    r[9] = (unsigned)p.N | ((unsigned long long)gridDim.x << 32);
    r[10] = trips | ((unsigned long long)(unsigned)tile << 32);
    r[11] = (unsigned)(grp + 1) | ((unsigned long long)(unsigned)p.split_s << 32);
    r[12] = (unsigned long long)p.C;
    r[13] = blockIdx.y;
    r[14] = cycles | (ticks << 40); // shader cycles (s_memtime) and 10 ns ticks over the workgroup's life: cycles / ticks x 100 = the shader clock in MHz while it ran
}
#define TR_WRITE(kid, tile, grp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TR_STAMP(5) trace_write(p, kid, tile, grp, tr_trips, tr_t, __builtin_readcyclecounter() - tr_c0, __builtin_amdgcn_s_memrealtime() - tr_r0); }
#else
#define TR_WRITE(kid, tile, grp)
#endif

#ifdef RTEN_TRACE
// host: hand the launch its record slots (called right before every instrumented launch)
inline void trace_assign(GemmArgs &a, unsigned workgroups) {
    a.trace_buf = g_trace_host.buf;
    a.trace_cap = g_trace_host.cap;
    a.trace_base = g_trace_host.next;
    g_trace_host.next += workgroups;
}
#define TRACE_ASSIGN(a, n) trace_assign(a, n)
#else
#define TRACE_ASSIGN(a, n)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


} // namespace
