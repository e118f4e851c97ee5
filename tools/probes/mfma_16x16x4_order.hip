// Probe: is v_mfma_f32_16x16x4_f32 a k-ordered fmaf chain (like 32x32x2, which the f32 GEMM kernels rely on)?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/p probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D) {
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16]; // A[i][k]
    const float b = B[(l / 16) * 16 + l % 16]; // B[k][j]
    f32x4 c;
    for (int r = 0; r < 4; r++) c[r] = C[(4 * (l / 16) + r) * 16 + l % 16];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * (l / 16) + r) * 16 + l % 16] = c[r];
}
int main() {
    float hA[64], hB[64], hC[256], hD[256];
    srand(7);
    int bad_fwd = 0, bad_rev = 0, bad_pair = 0;
    for (int trial = 0; trial < 200; trial++) {
        for (auto &x : hA) x = (float)rand() / RAND_MAX - 0.5f;
        for (auto &x : hB) x = (float)rand() / RAND_MAX - 0.5f;
        for (auto &x : hC) x = ((float)rand() / RAND_MAX - 0.5f) * 1e-3f;
        float *dA, *dB, *dC, *dD;
        hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            float f = hC[i * 16 + j], r = hC[i * 16 + j];
            for (int kk = 0; kk < 4; kk++) f = fmaf(hA[i * 4 + kk], hB[kk * 16 + j], f);
            for (int kk = 3; kk >= 0; kk--) r = fmaf(hA[i * 4 + kk], hB[kk * 16 + j], r);
            float p = hC[i * 16 + j]; // two independent pair sums then added?
            p = fmaf(hA[i * 4 + 1], hB[16 + j], fmaf(hA[i * 4], hB[j], p));
            p = fmaf(hA[i * 4 + 3], hB[48 + j], fmaf(hA[i * 4 + 2], hB[32 + j], p));
            if (hD[i * 16 + j] != f) bad_fwd++;
            if (hD[i * 16 + j] != r) bad_rev++;
            if (hD[i * 16 + j] != p) bad_pair++;
        }
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    printf("16x16x4 f32 vs fmaf chain k=0..3: %d mismatches of %d; reversed chain: %d\n", bad_fwd, 200 * 256, bad_rev);
    return 0;
}
