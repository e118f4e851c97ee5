// Probe (round 4): the k-loop of a ONE-WAVE workgroup that owns a whole 64x64 output tile -- four independent 32x32 accumulator blocks per wave
// (consecutive MFMAs never hit the same accumulator), a private LDS ring filled by the wave's own LDS-DMA, no s_barrier anywhere.
//
// Why: tools/debug/f32_trace.py (profiles/r07) shows the production 64x64 / 4-wave kernels keep the matrix pipe only 67-71 % busy INSIDE their
// k-loops (one accumulator block per wave: every MFMA depends on the previous one, and anything issued between two such MFMAs costs extra;
// a barrier per 8 MFMAs), while the 128x128 variant (four blocks per wave) runs at ~100 % in the same trace -- but 128x128 tiles quantise badly on
// ResNet's layer shapes.  A wave-sized workgroup keeps the 64x64 granularity AND the four independent chains.
//
//   W<dense|gather, BK, stages>: per k-tile a wave issues BK/4 (A) + BK/4 (dense B) dwordx4 DMAs or BK dword gathers (3x3 im2col, offsets
//   formed like the production kernel: LUT entry by scalar load, 3 VALU per gather), waits with a counted vmcnt, and runs BK/2 k-pairs of
//   {2 A-fragment + 2 B-fragment ds_read_b32, 4 MFMAs}.  Waves per compute unit are set by padding the LDS request.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_tile wave_tile.hip && /tmp/wave_tile
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
constexpr int TILE = 64;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// VAR (gather only): 0 = all DMA at the top of the k-tile, offsets from LUT entries loaded at use; 1 = offsets are loop-invariant registers (no LUT, no VALU);
// 2 = LUT entries fetched one k-tile ahead; 3 = as 2, and the gathers spread between the k-pairs' MFMA groups (2 per k-pair)
template <int GATHER, int BKW, int NST, int VAR = 0>
__global__ __launch_bounds__(64) void kloop_wave(const float *src, const i32x2 *lut, float *sink, unsigned long long *clocks, int iters, unsigned src_bytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = BKW * 2 * TILE; // floats: A [BKW][64] then B [BKW][64]
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    for (int i = lane; i < NST * STAGE; i += 64) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)src_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    typedef const __attribute__((address_space(4))) i32x2 *lut_ptr_t;
    const lut_ptr_t lc = (lut_ptr_t)(unsigned long long)lut;
    const unsigned voff = (unsigned)((blockIdx.x * 4096u + lane * 16u) % (src_bytes - (1u << 20)));
    const unsigned pix = voff >> 2;
    const unsigned inv = 0x80000000u | (lane == 5 ? 0x11u : 0u); // per-lane padding mask of the production gather
    constexpr int NA = BKW / 4, NB = GATHER ? BKW : BKW / 4;
    constexpr int PER_TILE = NA + NB;
    [[maybe_unused]] unsigned gconst[GATHER ? BKW : 1];
    if constexpr (GATHER && VAR == 1) {
#pragma unroll
        for (int r = 0; r < BKW; r++) gconst[r] = ((inv << (31 - r % 9)) & 0x80000000u) | ((pix + 40503u * r) << 2);
    }
    [[maybe_unused]] i32x2 lutE[GATHER ? BKW : 1];
    [[maybe_unused]] auto fetch_lut = [&](int kt) {
        if constexpr (GATHER) {
#pragma unroll
            for (int r = 0; r < BKW; r++) lutE[r] = lc[(kt & 63) * BKW + r];
        }
    };
    auto issue_a = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        const unsigned soff = (unsigned)(kt & 63) * 4096u;
#pragma unroll
        for (int j = 0; j < NA; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(As + j * 256), 16, (int)(voff + j * 1024u), (int)soff, 0, 0);
    };
    auto issue_b_row = [&](int kt, int stage, int r) { // one gather instruction (row r of the B tile)
        float *Bs = smem + stage * STAGE + BKW * TILE;
        unsigned go;
        if constexpr (VAR == 1) go = gconst[r];
        else {
            const i32x2 e = VAR >= 2 ? lutE[r] : lc[(kt & 63) * BKW + r];
            go = ((inv << e[1]) & 0x80000000u) | ((pix + (unsigned)e[0]) << 2);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + r * TILE), 4, (int)go, VAR == 1 ? (int)((kt & 63) * 4096u) : 0, 0, 0);
    };
    auto issue = [&](int kt, int stage) {
        float *As = smem + stage * STAGE, *Bs = As + BKW * TILE;
        const unsigned soff = (unsigned)(kt & 63) * 4096u;
        issue_a(kt, stage);
        if constexpr (!GATHER) {
#pragma unroll
            for (int j = 0; j < NB; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + j * 256), 16, (int)(voff + 65536u + j * 1024u), (int)soff, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < NB; r++) issue_b_row(kt, stage, r);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NST - 1; s++) { fetch_lut(s); issue(s, s); }
    fetch_lut(NST - 1);
    int stage = 0;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int kt = 0; kt < iters; kt++) {
        wait_vmcnt<PER_TILE *(NST - 2)>(); // tile kt has landed; NST-2 younger tiles stay in flight.  No barrier: the ring is this wave's own.
        const int sp = stage == 0 ? NST - 1 : stage - 1;
        if constexpr (GATHER && VAR == 3) issue_a(kt + NST - 1, sp);
        else issue(kt + NST - 1, sp);
        if constexpr (GATHER && VAR == 2) fetch_lut(kt + NST);
        const float *As = smem + stage * STAGE + l31 + half * TILE, *Bs = smem + stage * STAGE + BKW * TILE + l31 + half * TILE;
        float af[2][2], bf[2][2];
        af[0][0] = As[0]; af[0][1] = As[32]; bf[0][0] = Bs[0]; bf[0][1] = Bs[32];
#pragma unroll
        for (int kk = 0; kk < BKW / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BKW / 2) {
                af[nxt][0] = As[2 * (kk + 1) * TILE]; af[nxt][1] = As[2 * (kk + 1) * TILE + 32];
                bf[nxt][0] = Bs[2 * (kk + 1) * TILE]; bf[nxt][1] = Bs[2 * (kk + 1) * TILE + 32];
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            if constexpr (GATHER && VAR == 3) {
                issue_b_row(kt + NST - 1, sp, 2 * kk);
                issue_b_row(kt + NST - 1, sp, 2 * kk + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (GATHER && VAR == 3) fetch_lut(kt + NST);
        else __builtin_amdgcn_iglp_opt(0);
        stage = stage == NST - 1 ? 0 : stage + 1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    wait_vmcnt<0>();
    if (lane == 0) { clocks[blockIdx.x] = t1 - t0; clocks[gridDim.x + blockIdx.x] = r1 - r0; }
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) keep += acc[i][j][r];
    if (keep == 12345.678f) sink[0] = keep;
}

template <int GATHER, int BKW, int NST, int VAR = 0>
void run(const char *name, const float *src, const i32x2 *lut, float *sink, unsigned long long *clocks, int cus, unsigned src_bytes) {
    const int iters = 32768 / BKW * 4; // the same K per wave for every BK
    constexpr int kLds = NST * BKW * 2 * TILE * 4;
    for (int per_cu : {4, 8, 12}) {
        if (kLds * per_cu > 160 * 1024) continue;
        const int grid = cus * per_cu;
        const int dyn = (160 * 1024 / per_cu - 256) & ~1023; // exactly per_cu waves fit a compute unit
        hipFuncSetAttribute((const void *)kloop_wave<GATHER, BKW, NST, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((kloop_wave<GATHER, BKW, NST, VAR>), dim3(grid), dim3(64), (size_t)dyn, 0, src, lut, sink, clocks, iters, src_bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kloop_wave<GATHER, BKW, NST, VAR>), dim3(grid), dim3(64), (size_t)dyn, 0, src, lut, sink, clocks, iters, src_bytes);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid * 2);
        hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int i = 0; i < grid; i++) { cyc += (double)h[i]; rt += (double)h[grid + i]; }
        const double flops = (double)grid * iters * (BKW / 2) * 4 * 2.0 * 32 * 32 * 2;
        // s_memtime vs s_memrealtime (100 MHz): which clock do "cycles" count, and how fast does the shader clock really run under this load?
        printf("%-52s %2d waves/CU: %7.1f s_memtime ticks per 16 k per wave (2048 = matrix pipe alone)  %6.1f TFLOP/s  | s_memtime runs at %.1f MHz; loop wall %.0f us of kernel %.0f us\n",
               name, per_cu, cyc / grid / iters * 16.0 / BKW, flops / (ms * 1e-3) / 1e12, cyc / rt * 100.0, rt / grid / 100.0, ms * 1e3);
        fflush(stdout);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const unsigned src_bytes = 64u << 20;
    float *src, *sink;
    i32x2 *lut;
    unsigned long long *clocks;
    hipMalloc(&src, src_bytes);
    hipMemset(src, 0x3c, src_bytes);
    hipMalloc(&sink, 16);
    hipMalloc(&clocks, (size_t)cus * 16 * 8 * 2);
    std::vector<i32x2> hl(64 * 16);
    for (size_t i = 0; i < hl.size(); i++) hl[i] = i32x2{(int)((i * 40503u) % 200000u), 31 - (int)(i % 9)};
    hipMalloc(&lut, hl.size() * sizeof(i32x2));
    hipMemcpy(lut, hl.data(), hl.size() * sizeof(i32x2), hipMemcpyHostToDevice);
    run<0, 16, 2>("W  dense DMA, BK 16 x 2 stages", src, lut, sink, clocks, cus, src_bytes);
    run<0, 8, 4>("W  dense DMA, BK 8 x 4 stages", src, lut, sink, clocks, cus, src_bytes);
    run<1, 16, 2, 0>("W  3x3 gathers, BK 16 x 2, LUT at use, DMA first", src, lut, sink, clocks, cus, src_bytes);
    run<1, 16, 2, 1>("W  3x3 gathers, BK 16 x 2, constant offsets (no LUT / VALU)", src, lut, sink, clocks, cus, src_bytes);
    run<1, 16, 2, 2>("W  3x3 gathers, BK 16 x 2, LUT one tile ahead", src, lut, sink, clocks, cus, src_bytes);
    run<1, 16, 2, 3>("W  3x3 gathers, BK 16 x 2, LUT ahead, gathers between k-pairs", src, lut, sink, clocks, cus, src_bytes);
    run<1, 8, 4, 3>("W  3x3 gathers, BK 8 x 4, LUT ahead, gathers between k-pairs", src, lut, sink, clocks, cus, src_bytes);
    return 0;
}
