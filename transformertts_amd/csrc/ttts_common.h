// Shared device/host helpers for the Transformer-TTS HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ttts_hip.h"

namespace ttts {

// ---------------------------------------------------------------- host-side error plumbing
void set_error(const char* fmt, ...);

#define TTTS_REQUIRE(cond, ...)                     \
    do {                                            \
        if (!(cond)) {                              \
            ::ttts::set_error(__VA_ARGS__);         \
            return TTTS_ERR_INVALID;                \
        }                                           \
    } while (0)

#define TTTS_LAUNCH_CHECK(name)                                              \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            ::ttts::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return TTTS_ERR_LAUNCH;                                          \
        }                                                                    \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// out0[c] (+)= sum_r ws[r*ld + c] for c < n0, out1[c - n0] for the rest (reduce.hip); fixed summation order
int launch_reduce_rows(const float* ws, long ld, int nrows, long ncols, float* out0, long n0, float* out1, int accumulate,
                       hipStream_t stream);

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(seed, idx) is a pure function of the 64-bit site seed and the flat element index, so the
// backward kernels regenerate exactly the forward mask without storing it.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rand_u32(uint64_t seed, uint64_t idx) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t x = mix32(lo ^ (uint32_t)seed);
    x = mix32(x + hi * 0x9E3779B1u + (uint32_t)(seed >> 32));
    return x;
}
// threshold = round(p * 2^32) clipped; keep iff u32 >= threshold  (P[keep] = 1-p)
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0.0) t = 0.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}
__device__ __forceinline__ bool keep_elem(uint64_t seed, uint64_t idx, uint32_t thr) {
    return rand_u32(seed, idx) >= thr;
}

// ---------------------------------------------------------------- wave helpers (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// row index inside a 32x32 MFMA accumulator tile held by (lane, reg)
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

}  // namespace ttts
