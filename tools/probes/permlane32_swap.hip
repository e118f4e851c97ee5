// Probe: lane semantics of v_permlane32_swap on gfx950 (used by attention_fused.hip).  hipcc --offload-arch=gfx950 -o /tmp/p probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    const unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("swap(a=1000+lane, b=2000+lane): r[0] lanes 0,1,31,32,33,63 = %u %u %u %u %u %u\n", h[0], h[1], h[31], h[32], h[33], h[63]);
    printf("                                 r[1] lanes 0,1,31,32,33,63 = %u %u %u %u %u %u\n", h[64], h[65], h[95], h[96], h[97], h[127]);
    return 0;
}
