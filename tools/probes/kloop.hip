// Probe: what does one k-tile of the f32 implicit-GEMM loop cost a wave, piece by piece?
//
// A 64x64 tile / 4 waves / BK = 16 k-loop (8 dependent v_mfma_f32_32x32x2_f32 per wave per k-tile = 512 matrix-pipe cycles) is
// built up in steps and timed per k-tile with s_memtime, for 1..3 workgroups per compute unit:
//   A  MFMAs on register operands only
//   B  + operand fragments from LDS (16 ds_read_b32 per k-tile, next k-pair's reads behind the current MFMA, as the kernels do)
//   C  + s_barrier per k-tile
//   D  + tile DMA: 2 buffer_load_dwordx4 ... lds per wave per k-tile (dense 1x1 layer), counted vmcnt wait, 3 LDS stages
//   E  as D with 1 dwordx4 + 4 dword gathers per wave per k-tile (3x3 layer)
//   F  as D, but all fragments first and the 8 MFMAs back to back
// hipcc --offload-arch=gfx950 -O3 -o /tmp/kloop kloop.hip && /tmp/kloop
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 16, BM = 64, BN = 64, NSTAGE = 3, STAGE = BK * (BM + BN);
constexpr unsigned OOB = 0x80000000u;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(256) void kloop(const float *src, float *sink, unsigned long long *clocks, int iters, unsigned src_bytes) {
    __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wq = t >> 6, wm0 = (wq >> 1) * 32, wn0 = (wq & 1) * 32;
    for (int i = t; i < NSTAGE * STAGE; i += 256) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)src_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned voff = (unsigned)((blockIdx.x * 4096u + wave * 1024u + lane * 16u) % (src_bytes - 65536u));
    auto issue = [&](int kt, int stage) {
        float *As = smem + stage * STAGE, *Bs = As + BK * BM;
        const unsigned soff = (unsigned)(kt & 63) * 1024u;
        if constexpr (MODE == 3 || MODE == 5) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(As + wave * 256), 16, (int)voff, (int)soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + wave * 256), 16, (int)(voff + 4096u), (int)soff, 0, 0);
        } else if constexpr (MODE == 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(As + wave * 256), 16, (int)voff, (int)soff, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + (wave * 4 + r) * BN), 4, (int)(voff / 4 + 8192u * r + 4 * lane), (int)soff, 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    constexpr int PER_TILE = MODE == 4 ? 5 : 2;
    if constexpr (MODE >= 3) {
        issue(0, 0);
        issue(1, 1);
    }
    int stage = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < iters; kt++) {
        if constexpr (MODE >= 3) wait_vmcnt<PER_TILE>();
        if constexpr (MODE >= 2) __builtin_amdgcn_s_barrier();
        if constexpr (MODE >= 3) issue(kt + 2, stage == 0 ? NSTAGE - 1 : stage - 1);
        if constexpr (MODE == 0) {
            const float fa = (float)kt, fb = (float)lane;
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        } else if constexpr (MODE == 5) {
            const float *As = smem + stage * STAGE + wm0 + l31 + half * BM, *Bs = smem + stage * STAGE + BK * BM + wn0 + l31;
            float af[BK / 2], bf[BK / 2];
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) { af[kk] = As[2 * kk * BM]; bf[kk] = Bs[(2 * kk + half) * BN]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            const float *As = smem + stage * STAGE + wm0 + l31 + half * BM, *Bs = smem + stage * STAGE + BK * BM + wn0 + l31;
            float af[2], bf[2];
            af[0] = As[0];
            bf[0] = Bs[half * BN];
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk++) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < BK / 2) { af[nxt] = As[2 * (kk + 1) * BM]; bf[nxt] = Bs[(2 * (kk + 1) + half) * BN]; }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur], acc, 0, 0, 0);
            }
            __builtin_amdgcn_iglp_opt(0);
        }
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if constexpr (MODE >= 3) wait_vmcnt<0>();
    if (lane == 0 && wave == 0) clocks[blockIdx.x] = t1 - t0;
    float keep = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) keep += acc[r];
    if (keep == 12345.678f) sink[0] = keep;
}

template <int MODE>
void run(const char *name, const float *src, float *sink, unsigned long long *clocks, int cus, unsigned src_bytes) {
    const int iters = 4000;
    for (int per_cu = 1; per_cu <= 3; per_cu++) {
        const int grid = cus * per_cu;
        // pad the LDS request so that exactly per_cu workgroups fit a compute unit
        int dyn = (160 * 1024 / per_cu - NSTAGE * STAGE * 4 - 256) & ~1023;
        if (NSTAGE * STAGE * 4 + dyn > 64 * 1024) hipFuncSetAttribute((const void *)kloop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((kloop<MODE>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kloop<MODE>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid);
        hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (auto c : h) cyc += (double)c;
        const double flops = (double)grid * 4 * iters * 8.0 * 2.0 * 32 * 32 * 2;
        printf("%-46s %d WG/CU: %7.1f cycles per k-tile per wave (512 = matrix pipe alone)  %6.1f TFLOP/s\n", name, per_cu, cyc / grid / iters, flops / (ms * 1e-3) / 1e12);
    }
}

// G: the 3x3 convolution's k-loop on channels-last-by-4 activations ([N][C/4][H][W][4]): a k-UNIT is one channel quad x 9 taps = 36 k
// (18 MFMAs per wave, 1152 matrix-pipe cycles).  Per unit a workgroup moves A = 36 x 64 f32 (9 dwordx4 DMA instructions, k-major weights)
// and B = 9 taps x 64 pixels x 16 B (9 dwordx4 instructions, one per tap: a lane's piece is the 4 channels of ITS pixel) -- 18 instructions over
// 4 waves (5 / 5 / 4 / 4) against the 45 of the NCHW gather loop for the same 36 k.  B fragments: 9 ds_read_b128 give a lane all 36 values of
// its column, the k-pair of MFMA s is picked by register index (k = 9 j + tap, even k in the lower half-wave, odd k in the upper).
constexpr int UK = 36, USTAGE = UK * BM + 9 * BN * 4;
typedef float f32x4p __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kloop_cl4(const float *src, float *sink, unsigned long long *clocks, int iters, unsigned src_bytes) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float *smem = dsm;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wq = t >> 6, wm0 = (wq >> 1) * 32, wn0 = (wq & 1) * 32;
    for (int i = t; i < NSTAGE * USTAGE; i += 256) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)src_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned voff = (unsigned)((blockIdx.x * 4096u + wave * 1024u + lane * 16u) % (src_bytes - (1u << 20)));
    const int n_dma = wave < 2 ? 5 : 4; // instructions 0..8 = A rows (4 k rows each), 9..17 = B taps
    auto issue = [&](int u, int stage) {
        float *As = smem + stage * USTAGE, *Bs = As + UK * BM;
        const unsigned soff = (unsigned)(u & 63) * 4096u;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const int id = wave + 4 * q; // 0 .. 19; ids >= 18 do not exist
            if (q < 4 || wave < 2) {
                float *dst = id < 9 ? As + id * 256 : Bs + (id - 9) * 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, (int)(voff + (unsigned)id * 8192u), (int)soff, 0, 0);
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    issue(0, 0);
    issue(1, 1);
    int stage = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int u = 0; u < iters; u++) {
        if (n_dma == 5) wait_vmcnt<5>(); else wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();
        issue(u + 2, stage == 0 ? NSTAGE - 1 : stage - 1);
        const float *As = smem + stage * USTAGE + wm0 + l31, *Bs = smem + stage * USTAGE + UK * BM + (wn0 + l31) * 4;
        f32x4p b[9];
#pragma unroll
        for (int tap = 0; tap < 9; tap++) b[tap] = *reinterpret_cast<const f32x4p *>(Bs + tap * BN * 4);
        float af[2];
        af[0] = As[half * BM];
#pragma unroll
        for (int s = 0; s < 18; s++) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < 18) af[nxt] = As[(2 * (s + 1) + half) * BM];
            const int k0 = 2 * s, k1 = 2 * s + 1; // k = 9 j + tap
            const float bv = half ? b[k1 % 9][k1 / 9] : b[k0 % 9][k0 / 9];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bv, acc, 0, 0, 0);
        }
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    wait_vmcnt<0>();
    if (lane == 0 && wave == 0) clocks[blockIdx.x] = t1 - t0;
    float keep = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) keep += acc[r];
    if (keep == 12345.678f) sink[0] = keep;
}

void run_cl4(const float *src, float *sink, unsigned long long *clocks, int cus, unsigned src_bytes) {
    const int iters = 2000;
    for (int per_cu = 1; per_cu <= 2; per_cu++) {
        const int grid = cus * per_cu;
        const int base = NSTAGE * USTAGE * 4;
        int dyn = (160 * 1024 / per_cu - 256) & ~1023;
        if (dyn < base) dyn = base;
        hipFuncSetAttribute((const void *)kloop_cl4, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(kloop_cl4, dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kloop_cl4, dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid);
        hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (auto c : h) cyc += (double)c;
        const double flops = (double)grid * 4 * iters * 18.0 * 2.0 * 32 * 32 * 2;
        printf("%-46s %d WG/CU: %7.1f cycles per 36-k unit per wave (1152 = matrix pipe alone)  %6.1f TFLOP/s\n", "G  3x3 on channels-last-by-4: 36-k units", per_cu,
               cyc / grid / iters, flops / (ms * 1e-3) / 1e12);
    }
}

// H / I: 128 x 128 tile, four waves of 64 x 64 (2 x 2 accumulator blocks per wave: every fragment feeds two MFMAs), BK = 16: 32 MFMAs per wave per
// k-tile (2048 matrix-pipe cycles) behind ONE barrier, 16 KB of tile DMA per k-tile.  H = dense operands (4 dwordx4 per wave), I = 3x3 gathers on the
// B side (2 dwordx4 for A + 8 dword gathers per wave).  One workgroup per CU (128 accumulator registers + a second set for the depth-block fold).
constexpr int BM2 = 128, BN2 = 128, STAGE2 = BK * (BM2 + BN2);
template <int GATHER>
__global__ __launch_bounds__(256, 1) void kloop128(const float *src, float *sink, unsigned long long *clocks, int iters, unsigned src_bytes) {
    extern __shared__ __attribute__((aligned(16))) float dsm2[];
    float *smem = dsm2;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wq = t >> 6, wm0 = (wq >> 1) * 64, wn0 = (wq & 1) * 64;
    for (int i = t; i < NSTAGE * STAGE2; i += 256) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)src_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned voff = (unsigned)((blockIdx.x * 4096u + wave * 1024u + lane * 16u) % (src_bytes - (1u << 20)));
    auto issue = [&](int kt, int stage) {
        float *As = smem + stage * STAGE2, *Bs = As + BK * BM2;
        const unsigned soff = (unsigned)(kt & 63) * 2048u;
#pragma unroll
        for (int q = 0; q < 2; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(As + (wave * 2 + q) * 256), 16, (int)(voff + q * 16384u), (int)soff, 0, 0);
        if constexpr (GATHER) {
#pragma unroll
            for (int r = 0; r < 8; r++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + (wave * 8 + r) * 64), 4, (int)(voff / 4 + 8192u * r + 4 * lane), (int)soff, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 2; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + (wave * 2 + q) * 256), 16, (int)(voff + 65536u + q * 16384u), (int)soff, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    constexpr int PER_TILE = GATHER ? 10 : 4;
    issue(0, 0);
    issue(1, 1);
    int stage = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < iters; kt++) {
        wait_vmcnt<PER_TILE>();
        __builtin_amdgcn_s_barrier();
        issue(kt + 2, stage == 0 ? NSTAGE - 1 : stage - 1);
        const float *As = smem + stage * STAGE2 + wm0 + l31 + half * BM2, *Bs = smem + stage * STAGE2 + BK * BM2 + wn0 + l31;
        float af[2][2], bf[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++) { af[0][i] = As[i * 32]; bf[0][i] = Bs[half * BN2 + i * 32]; }
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < 2; i++) { af[nxt][i] = As[2 * (kk + 1) * BM2 + i * 32]; bf[nxt][i] = Bs[(2 * (kk + 1) + half) * BN2 + i * 32]; }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_iglp_opt(0);
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    wait_vmcnt<0>();
    if (lane == 0 && wave == 0) clocks[blockIdx.x] = t1 - t0;
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) keep += acc[i][j][r];
    if (keep == 12345.678f) sink[0] = keep;
}

template <int GATHER>
void run128(const char *name, const float *src, float *sink, unsigned long long *clocks, int cus, unsigned src_bytes) {
    const int iters = 1500, grid = cus;
    const int dyn = NSTAGE * STAGE2 * 4 + 96 * 1024; // > 80 KB: one workgroup per CU
    hipFuncSetAttribute((const void *)kloop128<GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kloop128<GATHER>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kloop128<GATHER>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid);
    hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto c : h) cyc += (double)c;
    const double flops = (double)grid * 4 * iters * 32.0 * 2.0 * 32 * 32 * 2;
    printf("%-46s 1 WG/CU: %7.1f cycles per k-tile per wave (2048 = matrix pipe alone)  %6.1f TFLOP/s\n", name, cyc / grid / iters, flops / (ms * 1e-3) / 1e12);
}

// J: as D (64x64 tile, dense DMA) with the synchronisation point moved to the MIDDLE of a k-tile's MFMA sequence: MFMAs 0-3 of tile kt, then the
// counted wait + barrier that publishes tile kt+1 and frees the stage tile kt-1 was read from, the DMA of tile kt+3 issued between MFMAs 4-7, and the
// first fragments of tile kt+1 fetched before tile kt ends -- no barrier, no DMA issue and no cold fragment read at the tile boundary.  Four LDS stages.
constexpr int NST4 = 4;
template <int GATHER>
__global__ __launch_bounds__(256) void kloop_mid(const float *src, float *sink, unsigned long long *clocks, int iters, unsigned src_bytes) {
    extern __shared__ __attribute__((aligned(16))) float dsm3[];
    float *smem = dsm3;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wq = t >> 6, wm0 = (wq >> 1) * 32, wn0 = (wq & 1) * 32;
    for (int i = t; i < NST4 * STAGE; i += 256) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)src_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const unsigned voff = (unsigned)((blockIdx.x * 4096u + wave * 1024u + lane * 16u) % (src_bytes - 65536u));
    auto issue_a = [&](int kt, int stage) {
        float *As = smem + stage * STAGE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(As + wave * 256), 16, (int)voff, (int)((unsigned)(kt & 63) * 1024u), 0, 0);
    };
    auto issue_b = [&](int kt, int stage) {
        float *Bs = smem + stage * STAGE + BK * BM;
        const unsigned soff = (unsigned)(kt & 63) * 1024u;
        if constexpr (GATHER) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + (wave * 4 + r) * BN), 4, (int)(voff / 4 + 8192u * r + 4 * lane), (int)soff, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(Bs + wave * 256), 16, (int)(voff + 4096u), (int)soff, 0, 0);
        }
    };
    constexpr int PER_TILE = GATHER ? 5 : 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    for (int i = 0; i < 3; i++) { issue_a(i, i); issue_b(i, i); }
    wait_vmcnt<2 * PER_TILE>();
    __syncthreads(); // tile 0 visible
    int stage = 0;
    float af[2], bf[2];
    {
        const float *As = smem + wm0 + l31 + half * BM, *Bs = smem + BK * BM + wn0 + l31;
        af[0] = As[0];
        bf[0] = Bs[half * BN];
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < iters; kt++) {
        const float *As = smem + stage * STAGE + wm0 + l31 + half * BM, *Bs = smem + stage * STAGE + BK * BM + wn0 + l31;
        const int sn = stage == NST4 - 1 ? 0 : stage + 1;
        const float *An = smem + sn * STAGE + wm0 + l31 + half * BM, *Bn = smem + sn * STAGE + BK * BM + wn0 + l31;
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 2) { af[nxt] = As[2 * (kk + 1) * BM]; bf[nxt] = Bs[(2 * (kk + 1) + half) * BN]; }
            else { af[nxt] = An[0]; bf[nxt] = Bn[half * BN]; } // first fragments of the next tile (published by the barrier below)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur], bf[cur], acc, 0, 0, 0);
            if (kk == 3) {
                wait_vmcnt<PER_TILE>(); // tile kt+1 has landed (kt+2 may still be in flight)
                __builtin_amdgcn_s_barrier();
                const int sp = stage == 0 ? NST4 - 1 : stage - 1; // the stage of tile kt-1: every wave is past it
                issue_a(kt + 3, sp);
            }
            if (kk == 5) issue_b(kt + 3, stage == 0 ? NST4 - 1 : stage - 1);
        }
        stage = sn;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    wait_vmcnt<0>();
    if (lane == 0 && wave == 0) clocks[blockIdx.x] = t1 - t0;
    float keep = af[0] + bf[0];
#pragma unroll
    for (int r = 0; r < 16; r++) keep += acc[r];
    if (keep == 12345.678f) sink[0] = keep;
}

template <int GATHER>
void run_mid(const char *name, const float *src, float *sink, unsigned long long *clocks, int cus, unsigned src_bytes) {
    const int iters = 4000;
    for (int per_cu = 1; per_cu <= 2; per_cu++) {
        const int grid = cus * per_cu;
        int dyn = (160 * 1024 / per_cu - 256) & ~1023;
        hipFuncSetAttribute((const void *)kloop_mid<GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((kloop_mid<GATHER>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kloop_mid<GATHER>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, src_bytes);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid);
        hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (auto c : h) cyc += (double)c;
        const double flops = (double)grid * 4 * iters * 8.0 * 2.0 * 32 * 32 * 2;
        printf("%-46s %d WG/CU: %7.1f cycles per k-tile per wave (512 = matrix pipe alone)  %6.1f TFLOP/s\n", name, per_cu, cyc / grid / iters, flops / (ms * 1e-3) / 1e12);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const unsigned src_bytes = 64u << 20;
    float *src, *sink;
    unsigned long long *clocks;
    hipMalloc(&src, src_bytes);
    hipMemset(src, 0x3c, src_bytes);
    hipMalloc(&sink, 16);
    hipMalloc(&clocks, (size_t)cus * 4 * 8);
    run_cl4(src, sink, clocks, cus, src_bytes);
    run_mid<0>("J  64x64, mid-tile barrier, dense DMA", src, sink, clocks, cus, src_bytes);
    run_mid<1>("K  64x64, mid-tile barrier, 3x3 gathers", src, sink, clocks, cus, src_bytes);
    run128<0>("H  128x128 tile, 64x64 per wave, dense DMA", src, sink, clocks, cus, src_bytes);
    run128<1>("I  128x128 tile, 64x64 per wave, 3x3 gathers", src, sink, clocks, cus, src_bytes);
    run<0>("A  MFMA on register operands", src, sink, clocks, cus, src_bytes);
    run<1>("B  + fragments from LDS (interleaved)", src, sink, clocks, cus, src_bytes);
    run<2>("C  + s_barrier per k-tile", src, sink, clocks, cus, src_bytes);
    run<3>("D  + tile DMA, 2 dwordx4 per wave (dense)", src, sink, clocks, cus, src_bytes);
    run<4>("E  + tile DMA, 1 dwordx4 + 4 dword (gather)", src, sink, clocks, cus, src_bytes);
    run<5>("F  as D, fragments first, MFMAs back to back", src, sink, clocks, cus, src_bytes);
    return 0;
}
