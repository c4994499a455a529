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

// p[0 .. nbytes) = 0 by a fill kernel (elementwise.hip); pointer and size multiples of 4 bytes
int launch_zero(void* p, size_t nbytes, hipStream_t stream);

// out0[c] (+)= sum_r ws[r*ld + c] for c < n0, out1[c - n0] for the rest (reduce.hip); fixed summation order
// `queue` != NULL: nothing later in the same pass reads the result (parameter gradients), so the reduction is appended to
// the caller's queue (ttts_reduce_queue_flush runs it) instead of being launched now
int launch_reduce_rows(const float* ws, long ld, int nrows, long ncols, float* out0, long n0, float* out1, int accumulate,
                       hipStream_t stream, ttts_reduce_queue* queue = nullptr);
// weight-matrix partials and the matching bias partials in one launch (reduce.hip)
int launch_reduce_rows_pair(const float* ws, long ld, int nrows, long ncols, float* out, const float* ws2, long ld2,
                            long ncols2, float* out2, int accumulate, hipStream_t stream, ttts_reduce_queue* queue = nullptr);
// conv weight-gradient partials ws[split][tap][co][ci] -> dw[co][ci][tap] (reduce.hip)
int launch_conv_wgrad_reduce(const float* ws, float* dw, int cout, int cin, int taps, int nsplit, int accumulate,
                             hipStream_t stream, ttts_reduce_queue* queue = nullptr);

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(seed, idx) is a pure function of the 64-bit site seed and the element index, so the backward kernels
// regenerate exactly the forward mask without storing it.  One mixed 32-bit word serves FOUR neighbouring elements
// (each sees the top 16 bits of word * its own odd constant): the keep test is `r16 >= round(p * 65536)`, i.e. p is
// honoured to 1.5e-5.  hash_pair is the stronger two-round mixer (still used for the scheduled-sampling draw).
__device__ __forceinline__ uint32_t hash_pair(uint64_t seed, uint32_t a, uint32_t b) {
    uint32_t x = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u) ^ (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0xC2B2AE3Du);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// Effective seed of a dropout site in this step: the site seed passed by value, XORed with a per-step 64-bit word read
// from device memory (NULL: none).  A captured HIP graph replays the same kernel arguments every step; the per-step
// word is what the host (one 8-byte copy per step) changes so that every replay draws fresh masks.
__device__ __forceinline__ uint64_t site_seed(uint64_t seed, const uint64_t* step_seed) {
    return step_seed != nullptr ? (seed ^ *step_seed) : seed;
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    double t = (double)p * 65536.0 + 0.5;
    if (t < 0.0) t = 0.0;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}
__device__ __forceinline__ bool keep_from_hash(uint32_t h, uint32_t odd, uint32_t thr) {
    return (odd ? (h >> 16) : (h & 0xFFFFu)) >= thr;
}
// attention-weight form: element (row, key) of the (B*H*Tq, Tk) weight matrix.  Keys 4j .. 4j+3 of a row share ONE mixed
// 32-bit word (attn_quad_hash); key 4j+e keeps its weight iff the top 16 bits of word * ATTN_DROP_MULT[e] reach the
// threshold, i.e. iff (word * mult) >= (thr << 16) -- one multiply and one compare per weight, no field extraction.  The
// forward / dq kernels hold four neighbouring keys in four registers of a lane (one hash per four weights); in the dk/dv
// kernels a lane is a key and a quad of lanes shares the word through a DPP broadcast.
__device__ __forceinline__ uint32_t attn_quad_hash(uint64_t seed, uint32_t row, uint32_t key_quad) {
    uint32_t x = (row * 0x9E3779B1u) ^ ((key_quad + 0x632BE5ABu) * 0x85EBCA77u) ^ (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0xC2B2AE3Du);
    x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu;
    return x;
}
__device__ __forceinline__ uint32_t attn_drop_mult(uint32_t key) {
    const uint32_t e = key & 3u;
    return e == 0u ? 0x9E3779B1u : e == 1u ? 0xC2B2AE3Du : e == 2u ? 0x27D4EB2Fu : 0x165667B1u;
}
// thr16 = thr << 16 (thr = drop_threshold(p) < 65536)
__device__ __forceinline__ bool attn_keep_word(uint32_t quad_hash, uint32_t mult, uint32_t thr16) { return quad_hash * mult >= thr16; }
__device__ __forceinline__ bool attn_keep(uint64_t seed, uint32_t row, uint32_t key, uint32_t thr16) {
    return attn_keep_word(attn_quad_hash(seed, row, key >> 2), attn_drop_mult(key), thr16);
}
// flat-index form (GEMM epilogues, element-wise kernels, BatchNorm): the same scheme on the element index -- elements
// 4i .. 4i+3 share one mixed word (a float4 of outputs costs one hash, four multiplies and four compares)
__device__ __forceinline__ uint32_t elem_quad_hash(uint64_t seed, uint64_t idx) {
    return attn_quad_hash(seed, (uint32_t)(idx >> 2), (uint32_t)(idx >> 34));
}
__device__ __forceinline__ bool keep_elem(uint64_t seed, uint64_t idx, uint32_t thr) {
    return attn_keep_word(elem_quad_hash(seed, idx), attn_drop_mult((uint32_t)idx), thr << 16);
}
// idx4: a multiple of 4; k[e] = keep_elem(seed, idx4 + e, thr)
__device__ __forceinline__ void keep_quad(uint64_t seed, uint64_t idx4, uint32_t thr, bool (&k)[4]) {
    const uint32_t w = elem_quad_hash(seed, idx4), thr16 = thr << 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = attn_keep_word(w, attn_drop_mult((uint32_t)e), thr16);
}

// ---------------------------------------------------------------- two-way f16 split of a pair of fp32 values
// hi = f16(x * scale), lo = f16(x * scale - hi), both pairs packed (element 0 in the low half).  Four mixed-precision
// FMAs (v_fma_mixlo/mixhi_f16: fp32 x fp32 [+ f16] -> f16 written into one half of the destination) instead of
// multiply + pack-convert + two conversions back + subtract + pack-convert: same bits (scale is a power of two, so
// x * scale is exact and each value is rounded once), a third fewer instructions.  Used by the weight-gradient loader
// (-4 % kernel time); in the attention kernels and the forward GEMM staging the same swap measured 3-7 % SLOWER (the
// compiler schedules its own conversion sequence around the MFMAs better than opaque asm), so they keep split2_pair.
__device__ __forceinline__ void split2_scaled(float x0, float x1, float scale, uint32_t& hi, uint32_t& lo) {
    uint32_t h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "v"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(scale), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(scale), "v"(h));
    hi = h; lo = l;
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

// Publish a wave's max|value| of a gradient tensor it has just written: slot-wise atomic max on the float's bit pattern
// (non-negative floats order like unsigned integers, so the result does not depend on arrival order).  `slots` is a
// caller-zeroed array of TTTS_AMAX_SLOTS floats -- the partial-maxima array the fp16x3 gradient GEMMs take as `dy_amax`.
// (Measured on the element-wise producers, 14 000 - 16 000 waves per launch: 1024 slots and a plain no-return atomic per wave
// cost a LayerNorm +1 us; 256 slots +5 us, and a look-before-you-add load in front of the atomic +10 us -- the slots'
// cache lines become the bottleneck either way.  So: many slots, fire and forget.)
__device__ __forceinline__ void amax_publish(float m, float* __restrict__ slots, int slot) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(slots) + (slot & (TTTS_AMAX_SLOTS - 1)), __float_as_uint(m));
}

// Power-of-two pre-scale of an fp16x3 operand whose largest magnitude is m: scale * m lands in [2^11, 2^12), i.e. every
// element within 2^-15 of the largest keeps 22 significant bits in its (hi, lo) f16 pair and smaller ones an absolute
// error of 2^-37 * m.  inv = 1 / scale (exact).  Zero, denormal-small (< 2^-103) and non-finite maxima use 1: nothing
// to scale, or the infinity / NaN travels through the product visibly.
__host__ __device__ __forceinline__ void h3_pow2_scale(float m, float& scale, float& inv) {
    union { float f; uint32_t u; } v;
    v.f = m;
    const uint32_t e = (v.u >> 23) & 0xffu;
    if (e >= 24u && e < 255u) {
        v.u = (265u - e) << 23; scale = v.f;               // m in [2^(e-127), 2^(e-126)): scale = 2^(138 - e)
        v.u = (e - 11u) << 23; inv = v.f;                  // 2^(e - 138)
    } else {
        scale = 1.0f; inv = 1.0f;
    }
}
// the maximum over an operand's partial maxima (TTTS_AMAX_SLOTS floats from ttts_amax_partials or a producer's
// `*_amax_out`; a weight image carries one value); every lane of the wave gets it.  A full array is four independent
// 16-byte loads per lane (one L2 round trip), not sixteen dword loads.  Kernels with thousands of small workgroups share
// the read between their waves instead (attention.hip).
__device__ __forceinline__ float h3_partials_max(const float* __restrict__ partials, int n, int lane) {
    float m = 0.f;
    static_assert(TTTS_AMAX_SLOTS == 1024, "four float4 per lane");
    if (n == TTTS_AMAX_SLOTS && (reinterpret_cast<uintptr_t>(partials) & 15) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(partials);
        const float4 a = p4[lane], b = p4[lane + 64], c = p4[lane + 128], d = p4[lane + 192];
        m = fmaxf(fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))),
                  fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w))));
    } else {
        for (int i = lane; i < n; i += 64) m = fmaxf(m, partials[i]);
    }
    return wave_max(m);
}
// (the result is wave-uniform: handed back in scalar registers, it costs the kernels no VGPRs)
__device__ __forceinline__ float h3_uniform(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ void h3_operand_scale(const float* __restrict__ partials, int n, int lane, float& scale, float& inv) {
    float s, i;
    h3_pow2_scale(h3_partials_max(partials, n, lane), s, i);
    scale = h3_uniform(s);
    inv = h3_uniform(i);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- in-kernel clock (diagnostic builds: -DTTTS_CLOCK_STAMPS)
// The clock a kernel actually ran at = delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz), stamped once around
// the whole kernel by wave 0 of each workgroup (MI355X_MICROARCH.md, DVFS give-back item 6: board power and sysfs sclk are not
// the test).  The values go to a buffer of their own that nothing else reads; tools/gemm_clock.py reports the median.
#ifdef TTTS_CLOCK_STAMPS
#define TTTS_CLOCK_BEGIN() const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime()
#define TTTS_CLOCK_END(buf, cap)                                                                                        \
    do {                                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < (cap)) {                                                                   \
            (buf)[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_c0;                                              \
            (buf)[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;                                      \
        }                                                                                                               \
    } while (0)
#else
#define TTTS_CLOCK_BEGIN()
#define TTTS_CLOCK_END(buf, cap)
#endif

// ---------------------------------------------------------------- split precision (bf16 x 3)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> (hi, mid, lo) bf16 pairs, each packed in one dword (element 0 in the low half).  Written on vector
// types so that the compiler emits v_cvt_pk_bf16_f32 / v_pk_add_f32: 9 VALU operations per pair.
__device__ __forceinline__ void split3_pair(f32x2 x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
    f32x2 f = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    const f32x2 r1 = x - f;
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    f = f32x2{__uint_as_float(mid << 16), __uint_as_float(mid & 0xffff0000u)};
    const f32x2 r2 = r1 - f;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
}

__device__ __forceinline__ void split3_pack4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
    split3_pair(f32x2{v.x, v.y}, hi.x, mid.x, lo.x);
    split3_pair(f32x2{v.z, v.w}, hi.y, mid.y, lo.y);
}

// row index inside a 32x32 MFMA accumulator tile held by (lane, reg)
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

}  // namespace ttts
