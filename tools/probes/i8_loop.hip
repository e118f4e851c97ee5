// Probe: the k-loop of the int8 implicit-GEMM kernel (64x64 tile, 64-byte k-tiles: one 1 KiB A piece + one 1 KiB B piece per
// wave per k-tile, counted vmcnt wait, s_barrier, 2 LDS fragment reads + 2 i8 MFMAs), stripped of everything else, with the
// operand streams placed like an under-filled ResNet stage-4 layer: 200 workgroups, 25 per XCD share their A pieces.
//   variants: NSTAGE 3 / 6; A shared by the workgroups of an XCD or private; with / without the barrier; with / without MFMAs
// hipcc --offload-arch=gfx950 -O3 -o /tmp/i8_loop i8_loop.hip && /tmp/i8_loop
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NSTAGE, bool SHARED_A, bool BARRIER, bool MFMA, bool A_RES = false, bool B_RES = false, bool ROTATE = false>
__global__ __launch_bounds__(256) void i8_loop(const unsigned char *A, const unsigned char *B, int *sink, unsigned long long *clocks, int nk, unsigned a_bytes, unsigned b_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = 128 * 64;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)B, 0, (int)b_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3;
    // A: [chunk][M = 512][16 B]; the XCD's M-tile; B: per workgroup 64 "pixels" x K bytes, chunk-major [chunk][N][16 B]
    const unsigned a_row0 = (unsigned)(SHARED_A ? xcd : (blockIdx.x % 8)) * 64u;
    const unsigned a_voff = (a_row0 + lane) * 16u + (SHARED_A ? 0u : (unsigned)wg_in_xcd * 512u * 16u * 4u * 80u);
    const unsigned b_voff = ((unsigned)blockIdx.x * 64u + lane) * 16u;
    const unsigned n_total = gridDim.x * 64u;
    // ROTATE: every workgroup of an XCD starts its K walk somewhere else (integer accumulation is order-free), so that the 25
    // workgroups sharing an A slice do not all ask the same L2 channel for the same 1 KiB piece at the same time
    const int nch = nk * 4;
    int ch = wave + (ROTATE ? (wg_in_xcd * nk / ((int)gridDim.x / 8)) * 4 : 0);
    auto issue = [&](int stage) {
        unsigned char *As = smem + stage * STAGE + wave * 1024, *Bs = smem + stage * STAGE + 4096 + wave * 1024;
        // *_RES: the stream cycles over its first 16 chunks (4 k-tiles), i.e. hits the L2 after the first trip
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)As, 16, (int)a_voff, (int)((unsigned)(A_RES ? ch & 15 : ch) * 16u * 512u), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)Bs, 16, (int)b_voff, (int)((unsigned)(B_RES ? ch & 15 : ch) * 16u * n_total), 0, 0);
        ch += 4;
        if (ROTATE && ch >= nch) ch -= nch;
    };
    i32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0;
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; i++) issue(i);
    int stage = 0;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < nk; kt++) {
        wait_vmcnt<2 * (NSTAGE - 2)>();
        if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
        issue(stage == 0 ? NSTAGE - 1 : stage - 1);
        if constexpr (MFMA) {
            const unsigned char *As = smem + stage * STAGE + (wm0 + l31) * 16, *Bs = smem + stage * STAGE + 4096 + (wn0 + l31) * 16;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const i32x4 af = *reinterpret_cast<const i32x4 *>(As + (2 * s + half) * 1024);
                const i32x4 bf = *reinterpret_cast<const i32x4 *>(Bs + (2 * s + half) * 1024);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc, 0, 0, 0);
            }
        }
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    wait_vmcnt<0>();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave == 0) clocks[blockIdx.x] = t1 - t0;
    int keep = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) keep += acc[r];
    if (keep == 0x12345678) sink[0] = keep;
}

template <int NSTAGE, bool SHARED_A, bool BARRIER, bool MFMA, bool A_RES = false, bool B_RES = false, bool ROTATE = false>
void run(const char *name, const unsigned char *A, const unsigned char *B, int *sink, unsigned long long *clocks, int grid, int nk, unsigned a_bytes, unsigned b_bytes, char *flush, size_t flush_bytes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = (size_t)NSTAGE * 8192;
    float best = 1e9f;
    std::vector<unsigned long long> h((size_t)grid);
    double cyc = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipMemsetAsync(flush, rep, flush_bytes, 0); // push the operands out of the L2s / Infinity Cache, like the layers in between do
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((i8_loop<NSTAGE, SHARED_A, BARRIER, MFMA, A_RES, B_RES, ROTATE>), dim3(grid), dim3(256), lds, 0, A, B, sink, clocks, nk, a_bytes, b_bytes);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
            cyc = 0;
            for (auto c : h) cyc += (double)c;
            cyc /= grid;
        }
    }
    printf("%-58s grid %4d nk %3d: %7.1f us  %6.1f cycles per k-tile (%5.1f ns)\n", name, grid, nk, best * 1e3, cyc / nk, cyc / nk / 2.4);
}

int main() {
    const unsigned a_bytes = 512u * 4608u * 100u, b_bytes = 1024u * 64u * 4608u;
    unsigned char *A, *B;
    char *flush;
    int *sink;
    unsigned long long *clocks;
    const size_t flush_bytes = 1ull << 30;
    hipMalloc(&A, a_bytes);
    hipMalloc(&B, b_bytes);
    hipMalloc(&flush, flush_bytes);
    hipMemset(A, 1, a_bytes);
    hipMemset(B, 2, b_bytes);
    hipMalloc(&sink, 16);
    hipMalloc(&clocks, 1024 * 8);
    for (int grid : {200, 512}) {
        run<3, true, true, true>("3 stages, A shared per XCD, barrier, MFMA (the kernel)", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<6, true, true, true>("6 stages, A shared per XCD, barrier, MFMA", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, false, true, true>("3 stages, A private, barrier, MFMA", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<6, false, true, true>("6 stages, A private, barrier, MFMA", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, true, false, true>("3 stages, A shared per XCD, NO barrier, MFMA", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, true, true, false>("3 stages, A shared per XCD, barrier, NO MFMA", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<12, true, true, true>("12 stages, A shared per XCD, barrier, MFMA", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, true, true, true, false, true>("3 stages, A shared cold, B L2-resident", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, true, true, true, true, false>("3 stages, A L2-resident, B cold", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, true, true, true, true, true>("3 stages, A and B L2-resident", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<6, true, true, true, false, true>("6 stages, A shared cold, B L2-resident", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<6, true, true, true, true, true>("6 stages, A and B L2-resident", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, true, true, true, false, false, true>("3 stages, A shared cold, B cold, K ROTATED per workgroup", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<6, true, true, true, false, false, true>("6 stages, A shared cold, B cold, K ROTATED", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<3, true, true, true, false, true, true>("3 stages, A shared cold, B resident, K ROTATED", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<6, true, true, true, false, true, true>("6 stages, A shared cold, B resident, K ROTATED", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
        run<12, true, true, true, false, false, true>("12 stages, A shared cold, B cold, K ROTATED", A, B, sink, clocks, grid, 72, a_bytes, b_bytes, flush, flush_bytes);
    }
    return 0;
}
