// Probe: what does the f32 matrix pipe SUSTAIN on this chip?  (roofline context for the f32 conv / GEMM kernels)
//
// MI355X_MICROARCH.md quotes 157.3 TFLOP/s for v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 (64 FLOP/clk/SIMD at 2.4 GHz) and
// notes that the chip clocks to its power budget ("DVFS give-back").  This probe runs nothing but MFMAs on random (non-zero,
// full-mantissa) register operands -- no loads, no LDS, no barriers -- on every SIMD for tens of milliseconds and reports
//   * TFLOP/s from the host's event timing, and
//   * the effective shader clock = s_memtime ticks (shader cycles) / wall_clock64 ticks (100 MHz constant) measured inside
//     the kernel,
// for 1, 2 and 4 waves per SIMD and 1 / 4 independent accumulators per wave.  hipcc --offload-arch=gfx950 -O3 -o /tmp/p mfma_sustained.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool M16>
__global__ __launch_bounds__(256) void mfma_stream(const float *seed, float *sink, unsigned long long *clocks, int iters) {
    const int lane = threadIdx.x & 63;
    float a = seed[lane], b = seed[64 + lane];
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float keep = 0.f;
    if constexpr (M16) {
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = f32x4{a, b, a, b};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16 / NACC; u++)
#pragma unroll
                for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) keep += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][r] = (r & 1) ? a : b;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16 / NACC; u++)
#pragma unroll
                for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) keep += acc[i][0] + acc[i][15];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (lane == 0 && (threadIdx.x >> 6) == 0) {
        clocks[2 * blockIdx.x] = t1 - t0;
        clocks[2 * blockIdx.x + 1] = w1 - w0;
    }
    if (keep == 12345.678f) sink[0] = keep; // keep the chain alive
}

template <int NACC, bool M16>
void run(const char *name, int wgs_per_cu, const float *seed, float *sink, unsigned long long *clocks, int cus) {
    const int iters = 20000; // 16 MFMAs per iteration
    const int grid = cus * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int warm = 0; warm < 2; warm++) hipLaunchKernelGGL((mfma_stream<NACC, M16>), dim3(grid), dim3(256), 0, 0, seed, sink, clocks, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 4;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((mfma_stream<NACC, M16>), dim3(grid), dim3(256), 0, 0, seed, sink, clocks, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)2 * grid);
    hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < grid; i++) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    const double flops_per_mfma = M16 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2;
    const double total = (double)reps * grid * 4 /*waves*/ * iters * 16.0 * flops_per_mfma;
    const double mhz = wall > 0 ? cyc / wall * 100.0 : 0.0; // wall_clock64 ticks at 100 MHz
    printf("%-34s %d wave(s)/SIMD  %7.1f TFLOP/s  (%.1f ms for %d launches)  effective shader clock %.0f MHz  cycles per MFMA per wave %.1f\n", name, wgs_per_cu,
           total / (ms * 1e-3) / 1e12, ms, reps, mhz, cyc / grid / ((double)iters * 16.0));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, nominal %d MHz: f32 MFMA nominal peak %.1f TFLOP/s\n", prop.gcnArchName, cus, prop.clockRate / 1000,
           cus * 4 * 64.0 * prop.clockRate * 1e3 / 1e12);
    float hseed[128];
    srand(3);
    for (auto &x : hseed) x = ((float)rand() / RAND_MAX - 0.5f) * 1.9f; // full-mantissa operands: the matrix pipe's switching activity is data dependent
    float *seed, *sink;
    unsigned long long *clocks;
    hipMalloc(&seed, sizeof hseed);
    hipMalloc(&sink, 16);
    hipMalloc(&clocks, (size_t)cus * 8 * 2 * 8);
    hipMemcpy(seed, hseed, sizeof hseed, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<4, false>("32x32x2, 4 accumulators / wave", w, seed, sink, clocks, cus);
        run<1, false>("32x32x2, 1 accumulator / wave", w, seed, sink, clocks, cus);
        run<4, true>("16x16x4, 4 accumulators / wave", w, seed, sink, clocks, cus);
        run<1, true>("16x16x4, 1 accumulator / wave", w, seed, sink, clocks, cus);
    }
    // the same stream on zero operands: the clock the power manager allows when the multipliers do not toggle
    float zeros[128] = {0};
    hipMemcpy(seed, zeros, sizeof zeros, hipMemcpyHostToDevice);
    run<4, false>("32x32x2, 4 acc, ZERO operands", 2, seed, sink, clocks, cus);
    return 0;
}
