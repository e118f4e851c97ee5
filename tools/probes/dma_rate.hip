// Probe: how fast can one compute unit pull tile data into LDS with buffer_load ... lds (dwordx4 per lane = 1 KiB per wave
// instruction), as a function of the number of loads a wave keeps in flight, the workgroups per CU and where the data lives
// (a 2 MiB window = L2 hits, 96 MiB = Infinity Cache, 2 GiB = HBM)?  The int8 conv kernels move (BM + BN) * 64 bytes per k-tile
// for two 32-cycle MFMAs per wave, so their k-loop runs at whatever this path sustains.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate dma_rate.hip && /tmp/dma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: LDS-DMA dwordx4; MODE 1: global_load_dwordx4 into registers (consumed by an xor)
template <int DEPTH, int MODE>
__global__ __launch_bounds__(256) void dma_stream(const uint4 *src, unsigned *sink, unsigned long long *clocks, int iters, unsigned window_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)window_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    // every wave walks its own 1 KiB pieces through the window with a large odd stride (no two waves share a line at a time)
    unsigned pos = ((blockIdx.x * 4u + wave) * 2654435761u) % (window_bytes / 1024u);
    const unsigned nblk = window_bytes / 1024u;
    const unsigned step = 7919u % nblk; // odd, < nblk: the walk stays inside the window
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 ring[MODE == 1 ? DEPTH : 1];
    auto issue = [&](int slot) {
        const unsigned off = pos * 1024u;
        if constexpr (MODE == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (wave * DEPTH + slot) * 1024), 16, (int)(lane * 16u), (int)off, 0, 0);
        } else {
            ring[slot] = src[(off >> 4) + lane];
        }
        pos += step;
        if (pos >= nblk) pos -= nblk;
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; s++) issue(s);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; s++) {
            issue((s + DEPTH - 1) % DEPTH);
            if constexpr (MODE == 0) {
                wait_vmcnt<DEPTH - 1>();
            } else {
                acc.x ^= ring[s].x; acc.y ^= ring[s].y; acc.z ^= ring[s].z; acc.w ^= ring[s].w;
            }
        }
    }
    wait_vmcnt<0>();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave == 0) clocks[blockIdx.x] = t1 - t0;
    if constexpr (MODE == 0) {
        __syncthreads();
        if (smem[t * 16] == 0x7f && iters < 0) sink[0] = 1;
    } else if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int DEPTH, int MODE>
void run(const uint4 *src, unsigned *sink, unsigned long long *clocks, int cus, unsigned window_bytes, const char *where) {
    const int iters = 2048 / DEPTH * DEPTH;
    for (int per_cu : {1, 2, 4}) {
        const int grid = cus * per_cu;
        const int ring_bytes = 4 * DEPTH * 1024;
        int dyn = (160 * 1024 / per_cu - 512) & ~1023;
        if (dyn < ring_bytes) continue;
        hipFuncSetAttribute((const void *)dma_stream<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((dma_stream<DEPTH, MODE>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, window_bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((dma_stream<DEPTH, MODE>), dim3(grid), dim3(256), (size_t)dyn, 0, src, sink, clocks, iters, window_bytes);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid);
        hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (auto c : h) cyc += (double)c;
        cyc /= grid;
        const double bytes = (double)grid * 4 * iters * 1024.0;
        printf("%-6s %s depth %2d  %d WG/CU: %7.1f cycles per 1 KiB wave load  %6.1f B/clk/CU  %7.2f TB/s aggregate\n", where, MODE == 0 ? "lds-dma " : "to-vgpr ", DEPTH, per_cu,
               cyc / iters, (double)per_cu * 4 * iters * 1024.0 / cyc, bytes / (ms * 1e-3) / 1e12);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t cap = 2ull << 30;
    uint4 *src;
    unsigned *sink;
    unsigned long long *clocks;
    hipMalloc(&src, cap);
    hipMemset(src, 0x11, cap);
    hipMalloc(&sink, 16);
    hipMalloc(&clocks, (size_t)cus * 8 * 8);
    struct { unsigned bytes; const char *name; } wins[] = {{2u << 20, "L2"}, {96u << 20, "MALL"}, {0x7ff00000u, "HBM"}};
    for (auto w : wins) {
        run<2, 0>(src, sink, clocks, cus, w.bytes, w.name);
        run<4, 0>(src, sink, clocks, cus, w.bytes, w.name);
        run<8, 0>(src, sink, clocks, cus, w.bytes, w.name);
        run<16, 0>(src, sink, clocks, cus, w.bytes, w.name);
        run<4, 1>(src, sink, clocks, cus, w.bytes, w.name);
        run<8, 1>(src, sink, clocks, cus, w.bytes, w.name);
    }
    return 0;
}
