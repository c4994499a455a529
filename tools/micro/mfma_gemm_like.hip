// The wave-level inner loop of the fp16x3 GEMMs without the memory system behind it: a 64 x 128 wave tile (128 accumulator
// registers), fragments read from a static LDS image at the kernels' ratio, three MFMA terms per product -- in the 32x32x16
// shape (12 ds_read_b128 per 24 MFMAs, one k-step of 16) and in the 16x16x32 shape (24 reads per 96 MFMAs, one k-step of 32),
// at one and two waves per SIMD.  Prints executed TFLOP/s and the clock the chip holds.  (does the advantage of
// the 16x16x32 shape in tools/micro/mfma_shape.hip survive fragment traffic and a full-size accumulator?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const f16x8* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    extern __shared__ f16x8 lds[];             // 96 KB: one workgroup per CU; 6144 fragments of 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 6144; i += blockDim.x) lds[i] = in[(blockIdx.x * 6144 + i) % (256 * 512)];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    const f16x8* base = lds + (wave & 3) * 1024 + lane;
    if (SHAPE == 32) {
        f32x16 c[2][4];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const f16x8* p = base + (it & 7) * 64;
            f16x8 a[2][2], b[2][4];                        // [plane][block]
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[pl][i] = p[(pl * 2 + i) * 64 % 512];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[pl][j] = p[(4 + pl * 4 + j) * 64 % 512 + 512];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x16 acc = c[i][j];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc, 0, 0, 0);
                    c[i][j] = acc;
                }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sink += c[i][j][r];
    } else {
        f32x4 c[4][8];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) c[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const f16x8* p = base + (it & 7) * 64;
            f16x8 a[2][4];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[pl][i] = p[(pl * 4 + i) * 64 % 512];
            // the B fragments two column blocks at a time (8 registers in flight instead of 64)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f16x8 b0 = p[(8 + j) * 64 % 512 + 512], b1 = p[(16 + j) * 64 % 512 + 512];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 acc = c[i][j];
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][i], b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i], b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][i], b0, acc, 0, 0, 0);
                    c[i][j] = acc;
                }
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) sink += c[i][j][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 512 + tid] = sink;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(int threads, const f16x8* din, float* dout, unsigned long long* dclk, const char* what) {
    const int blocks = 256;
    const int iters = SHAPE == 32 ? 40000 : 10000;         // 24 resp. 96 MFMAs per iteration
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 8; ++l) hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 96 * 1024, 0, din, dout, dclk, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512];
    hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost);
    const double waves = blocks * (threads / 64.0);
    const double mfma_per_wave = (SHAPE == 32 ? 24.0 : 96.0) * iters;
    const double flop = 8.0 * waves * mfma_per_wave * 32768.0 / (SHAPE == 32 ? 1.0 : 2.0);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    printf("%-40s %7.1f TFLOP/s  %6.2f cycles per MFMA per SIMD  in-kernel clock %.2f GHz  (%.0f ms)\n", what, flop / (ms * 1e-3) / 1e12,
           (double)h[0] / mfma_per_wave / (threads / 256.0), ghz, ms);
}

int main() {
    const size_t n = 256 * 512;
    f16x8* h = (f16x8*)malloc(n * sizeof(f16x8));
    srand(3);
    for (size_t i = 0; i < n; ++i) for (int e = 0; e < 8; ++e) h[i][e] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    f16x8* din; float* dout; unsigned long long* dclk;
    hipMalloc(&din, n * sizeof(f16x8)); hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dclk, 512 * 8);
    hipMemcpy(din, h, n * sizeof(f16x8), hipMemcpyHostToDevice);
    run<32>(256, din, dout, dclk, "32x32x16 + LDS fragments, one wave per SIMD");
    run<16>(256, din, dout, dclk, "16x16x32 + LDS fragments, one wave per SIMD");
    run<32>(512, din, dout, dclk, "32x32x16 + LDS fragments, two waves per SIMD");
    run<16>(512, din, dout, dclk, "16x16x32 + LDS fragments, two waves per SIMD");
    return 0;
}
