// Probe: how fast can ONE CU pull L2-resident bytes into LDS, by path and by address pattern?
//   path: LDS-DMA (buffer_load_dwordx4 ... lds / dword ... lds) or global_load_dwordx4 -> VGPR -> ds_write_b128
//   pattern: lane-contiguous 16 B (one 1 KiB run per wave instruction) or one 16 B piece per 576 B row (k-contiguous rows)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/p lds_fill_rate.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each wave moves J KiB-pieces per tile; a tile is 4*J KiB per workgroup; 3 LDS stages, counted waits
template <int MODE, int J>
__global__ __launch_bounds__(256) void fill(const unsigned char *g, long long win, int iters, int shared_win, unsigned *sink) {
    constexpr int TILE = 4 * J * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * TILE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned char *base = g + (shared_win ? 0 : (long long)blockIdx.x * win);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)win, 0x00020000);
    const unsigned wmask = (unsigned)win - 1;
    constexpr bool STRIDED = (MODE & 1) != 0;
    constexpr bool DMA = MODE < 2 || MODE == 4;
    unsigned pos = 0; // tile cursor within the window
    i32x4 regs[3][J];
    auto issue = [&](int stage, int slot) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            unsigned voff, soff;
            if (STRIDED) { // piece (wave, j): 64 rows of 576 B, 16 B each; successive pieces step 16 B along the row
                voff = (unsigned)lane * 576u;
                soff = (pos + (unsigned)((wave * J + j) * 16)) & wmask & ~15u;
                if (soff + 64u * 576u > (unsigned)win) soff = 0;
            } else {
                voff = (unsigned)lane * (MODE == 4 ? 4u : 16u);
                soff = (pos + (unsigned)((wave * J + j) * 1024)) & wmask;
            }
            unsigned char *dst = smem + stage * TILE + (wave * J + j) * 1024;
            if (MODE == 4) {
#pragma unroll
                for (int q = 0; q < 4; q++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + q * 256), 4, (int)voff, (int)(soff + q * 256), 0, 0);
            } else if (DMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, (int)voff, (int)soff, 0, 0);
            } else {
                regs[slot][j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
            }
        }
        pos = (pos + (STRIDED ? (unsigned)(4 * J * 16) : (unsigned)TILE)) & wmask;
    };
    auto land = [&](int stage, int slot) { // register path: park the tile in LDS
        if (!DMA) {
#pragma unroll
            for (int j = 0; j < J; j++) *reinterpret_cast<i32x4 *>(smem + stage * TILE + (wave * J + j) * 1024 + lane * 16) = regs[slot][j];
        }
    };
    constexpr int PER = (MODE == 4 ? 4 : 1) * J;
    issue(0, 0);
    issue(1, 1);
    unsigned acc = 0;
    int stage = 0;
#pragma unroll 3
    for (int it = 0; it < iters; it++) {
        if (DMA) wait_vmcnt<PER>();
        else { wait_vmcnt<J>(); land(stage, it % 3); }
        __builtin_amdgcn_s_barrier();
        issue(stage == 0 ? 2 : stage - 1, (it + 2) % 3);
        acc += *reinterpret_cast<const unsigned *>(smem + stage * TILE + ((t * 16 + it * 4) & (TILE - 1) & ~3)); // one consumer read per tile
        stage = stage == 2 ? 0 : stage + 1;
    }
    wait_vmcnt<0>();
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int J>
void run(const char *name, const unsigned char *g, long long win, int wgs, int shared_win, unsigned *sink) {
    const int iters = 4000 / J;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill<MODE, J>), dim3(wgs), dim3(256), 0, 0, g, win, 50, shared_win, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<MODE, J>), dim3(wgs), dim3(256), 0, 0, g, win, iters, shared_win, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * iters * 4.0 * J * 1024;
    const int cus = wgs < 256 ? wgs : 256;
    printf("%-44s J=%d wgs=%4d %s  %7.3f ms  %7.1f GB/s chip  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.4 GHz)\n", name, J, wgs, shared_win ? "shared-window" : "own-window   ", ms,
           bytes / ms / 1e6, bytes / ms / 1e6 / cus, bytes / ms / 1e6 / cus / 2.4);
}

int main() {
    const long long win = 64 * 1024;
    unsigned char *g;
    unsigned *sink;
    hipMalloc(&g, win * 1024);
    hipMemset(g, 1, win * 1024);
    hipMalloc(&sink, 64);
    for (int wgs : {64, 256, 512, 1024}) {
        run<0, 4>("LDS-DMA b128, lane-contiguous", g, win, wgs, 0, sink);
        run<1, 4>("LDS-DMA b128, 16 B per 576 B row", g, win, wgs, 0, sink);
        run<2, 4>("global_load_b128 -> ds_write, contiguous", g, win, wgs, 0, sink);
        run<3, 4>("global_load_b128 -> ds_write, row-strided", g, win, wgs, 0, sink);
        run<4, 4>("LDS-DMA b32, lane-contiguous", g, win, wgs, 0, sink);
    }
    run<0, 4>("LDS-DMA b128, lane-contiguous", g, win, 256, 1, sink);
    run<2, 4>("global_load_b128 -> ds_write, contiguous", g, win, 256, 1, sink);
    run<0, 2>("LDS-DMA b128, lane-contiguous", g, win, 256, 0, sink);
    run<0, 8>("LDS-DMA b128, lane-contiguous", g, win, 256, 0, sink);
    return 0;
}
