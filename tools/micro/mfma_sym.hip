// Is v_mfma_f32_32x32x16_f16 symmetric under operand swap, bit for bit?  D1 = A B^T (A first), D2 = B A^T (B first):
// D1[i][j] == D2[j][i] ?   Large-magnitude f16 operands so that fp32 accumulation rounds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
__global__ void k(const _Float16* A, const _Float16* B, float* D1, float* D2, int steps) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    f32x16 c1, c2;
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; c2[r] = 0.f; }
    for (int s = 0; s < steps; ++s) {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = A[(l31 * steps + s) * 16 + 8 * half + e]; b[e] = B[(l31 * steps + s) * 16 + 8 * half + e]; }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);    // rows = A rows, lane = B row
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0);    // rows = B rows, lane = A row
    }
    for (int r = 0; r < 16; ++r) {
        D1[acc_row(r, half) * 32 + l31] = c1[r];      // D1[i = A row][j = B row]
        D2[acc_row(r, half) * 32 + l31] = c2[r];      // D2[j = B row][i = A row]
    }
}
int main() {
    const int steps = 4, n = 32 * steps * 16;
    _Float16 *hA = (_Float16*)malloc(n * 2), *hB = (_Float16*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) { hA[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4096.f); hB[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4096.f); }
    _Float16 *dA, *dB; float *d1, *d2;
    hipMalloc(&dA, n * 2); hipMalloc(&dB, n * 2); hipMalloc(&d1, 4096); hipMalloc(&d2, 4096);
    hipMemcpy(dA, hA, n * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB, n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d1, d2, steps);
    float h1[1024], h2[1024];
    hipMemcpy(h1, d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 4096, hipMemcpyDeviceToHost);
    int bad = 0; double maxrel = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float x = h1[i * 32 + j], y = h2[j * 32 + i];
        if (x != y) { ++bad; double r = fabs((double)x - y) / fabs((double)x); if (r > maxrel) maxrel = r; }
    }
    printf("mfma_f32_32x32x16_f16 operand-swap symmetry: %d of 1024 elements differ, max rel %.3e (sample %g vs %g)\n", bad, maxrel, h1[5], h2[5 * 32]);
    return 0;
}
