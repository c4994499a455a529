// What the chip sustains on f16 MFMAs alone, by instruction shape: v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16, one wave
// per SIMD and two, operands in registers (random data: the clock the chip holds depends on the data, MI355X_MICROARCH.md "DVFS
// give-back"), >= 0.3 s per measurement.  Prints TFLOP/s, cycles per MFMA (s_memtime) and the in-kernel clock (s_memtime /
// s_memrealtime).  The practical ceiling every MFMA kernel of this repo is priced against in DESIGN.md section 10.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const f16x8* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    extern __shared__ float pad[];             // 100 KB: one workgroup per CU, whatever the dispatcher would otherwise pack
    const int tid = threadIdx.x;
    if (iters < 0) pad[tid] = 1.f;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(blockIdx.x * 512 + tid) * 8 + i]; b[i] = in[(blockIdx.x * 512 + tid) * 8 + 4 + i]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    if (SHAPE == 32) {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sink += c[i][r];
    } else {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[i & 3], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) sink += c[i][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 512 + tid] = sink;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(int threads, const f16x8* din, float* dout, unsigned long long* dclk, const char* what) {
    const int blocks = 256;
    int iters = 40000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 8; ++l) hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 100 * 1024, 0, din, dout, dclk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512];
    hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost);
    const double waves = blocks * (threads / 64.0);
    const double mfma_per_wave = (SHAPE == 32 ? 16.0 : 32.0) * iters;
    const double flop = 8.0 * waves * mfma_per_wave * (SHAPE == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    printf("%-34s %7.1f TFLOP/s  %6.2f cycles per MFMA per SIMD  in-kernel clock %.2f GHz  (%.0f ms)\n", what, flop / (ms * 1e-3) / 1e12,
           (double)h[0] / mfma_per_wave / (threads / 256.0), ghz, ms);
}

int main() {
    const size_t n = 256 * 512 * 8;
    f16x8* h = (f16x8*)malloc(n * sizeof(f16x8));
    srand(3);
    for (size_t i = 0; i < n; ++i) for (int e = 0; e < 8; ++e) h[i][e] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    f16x8* din; float* dout; unsigned long long* dclk;
    hipMalloc(&din, n * sizeof(f16x8)); hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dclk, 512 * 8);
    hipMemcpy(din, h, n * sizeof(f16x8), hipMemcpyHostToDevice);
    run<32>(256, din, dout, dclk, "32x32x16 f16, one wave per SIMD");
    run<16>(256, din, dout, dclk, "16x16x32 f16, one wave per SIMD");
    run<32>(512, din, dout, dclk, "32x32x16 f16, two waves per SIMD");
    run<16>(512, din, dout, dclk, "16x16x32 f16, two waves per SIMD");
    return 0;
}
