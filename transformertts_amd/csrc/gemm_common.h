// GemmArgs and the raw-buffer load helpers shared by the GEMM translation units (gemm.hip, gemm_h3.hip).
#pragma once
#include "ttts_common.h"

namespace ttts {

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long lda, ldb, ldc;
    // implicit row shift (conv taps / go-frame): rows are (b*T + t)
    int T;           // 0: no utterance clipping (shift must be 0)
    int cin;         // A_KC: channels per tap (K = taps*cin)
    int shift0, shift_step;
    int ztaps;       // !B_KC: number of taps spread over blockIdx.z (z = split*ztaps + tap)
    // split over the reduction dimension
    int kt_per_split;
    long c_zstride;  // C offset per z slice
    // epilogue (applied only when kt range covers all of K, i.e. no split)
    const float* bias;
    int act;         // 0 none, 1 relu
    float drop_scale;
    uint32_t drop_thr;
    uint64_t seed;
    const uint64_t* step_seed;   // per-step word XORed into seed (NULL: none); see site_seed()
    const float* residual;
    long ldr;
    // data-gradient form: the output is the gradient w.r.t. an activation h = drop(relu(.)) whose forward value is
    // relu_out (same shape as C): v = relu_out > 0 ? v * relu_scale : 0   (fused relu / dropout backward mask)
    const float* relu_out;
    float relu_scale;
    // weight-gradient form only: per-split column sums of A (= bias gradient partials), [zsplit][M]
    float* colsum;
    // operand extents in bytes (< 4 GiB): loads go through raw buffer descriptors, so an out-of-range offset returns
    // zeros in hardware -- row / column / tap clipping costs a select on the offset instead of a branch around the load
    uint32_t a_bytes, b_bytes;
    // fp16x3 kernels only: f16 has no exponent range to spare, so both operands are pre-scaled by powers of two taken from
    // their measured maxima.  a_amax / b_amax point at a_amax_n / b_amax_n partial maxima of |A| / |B| (the arrays of
    // ttts_amax_partials or of a producer's `*_amax_out`; a weight-plane image carries ONE value, its tail): every workgroup
    // reduces them and scales the operand by the power of two that puts its maximum in [2^11, 2^12).  Weight planes were
    // written with that scale already (weight_split_h3_one), activations are scaled while they are staged.
    const float* a_amax;
    int a_amax_n;
    const float* b_amax;
    int b_amax_n;
    // fp16x3 forward of a convolution that feeds BatchNorm: NULL, or the workspace of ttts_bn_train_stats.  Every wave
    // leaves, per output column, the (count, mean, M2) of the rows of its wave tile at bn_ws[((tile_row * WM + wave_m) * 3
    // + {0,1,2}) * N + column] -- the row-chunk partials bn_stats_final_kernel merges -- so the statistics cost no pass
    // over y.  Only tiles whose lanes keep one column group (h3_bn_blocks() > 0).
    float* bn_ws;
    // fp16x3 kernel: NULL, or a caller-zeroed TTTS_AMAX_SLOTS-slot array that receives max|C| (amax_publish): the output is a
    // gradient that another fp16x3 GEMM will consume
    float* c_amax;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t OOB = 0xFFFFFFFFu;
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// ... with a wave-uniform scalar byte offset on top of the per-lane one, and the matching 16-byte store (an offset at or
// beyond the descriptor's extent reads 0 / writes nothing)
__device__ __forceinline__ float4 buf_load4s(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, uint32_t soff) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, (int)soff, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ void buf_store4s(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, uint32_t soff, float4 x) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, (int)soff, 0);
}

// ---- fp16x3 ("h3") split: constants and the weight-plane image (see gemm_h3.hip)
constexpr int HBK = 32;                 // k-tile depth
constexpr int H3_AMAX_PARTIALS = TTTS_AMAX_SLOTS;  // length of every partial-maxima array (include/ttts_hip.h)

// byte size of the fp16x3 image of a rows x cols weight: two f16 planes, then a 16-byte tail whose first float is max|w|
// (written by the split, read by every GEMM that takes the planes: the scale the planes were written with follows from it)
// The image is cut in 32-deep k-tiles that must not straddle a convolution tap, so the channels of a tap (all `cols` of a
// linear weight) are PADDED with zeros to the next multiple of 32: the 80-channel mel side runs as 96 channels whose last
// 16 multiply whatever the activation row's neighbour holds by zero (h3_image_cols = the padded column count).
__host__ __device__ __forceinline__ int h3_pad32(int c) { return (c + HBK - 1) / HBK * HBK; }
__host__ __device__ __forceinline__ long h3_image_cols(long cols, int c2, int taps) {
    return (c2 > 0 && taps > 0) ? (long)taps * h3_pad32(c2) : (long)h3_pad32((int)cols);
}
__host__ __device__ __forceinline__ size_t h3_plane_bytes(long rows, long image_cols) { return (size_t)rows * image_cols * 4; }
__host__ __device__ __forceinline__ const float* h3_plane_tail(const void* planes, long rows, long image_cols) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(planes) + h3_plane_bytes(rows, image_cols));
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// two fp32 (already pre-scaled) -> (hi, lo) f16 pairs, each packed in one dword (element 0 in the low half)
__device__ __forceinline__ void split2_pair(f32x2 x, uint32_t& hi, uint32_t& lo) {
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(h, f32x2);          // exact: h is within 2^-11 of x
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}

// |w| maximum of one weight into the tail of its plane image (atomic max on the bit pattern; the tail was zeroed first).
// Called by ONE wave per 256 elements (blockDim 64): a float4 per lane, a wave reduction, and a look at the tail before the
// atomic -- every wave of a weight aims at the same word, and after the first few arrivals almost none has anything to add.
__device__ __forceinline__ void weight_amax_h3_one(const float* __restrict__ w, unsigned short* __restrict__ planes, int R,
                                                   int C, long image_cols, long i0) {
    const long n = (long)R * C;
    const long i = i0 + 4 * (long)(threadIdx.x & 63);
    float m = 0.f;
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(w + i);
        m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    } else {
        for (int e = 0; e < 4; ++e)
            if (i + e < n) m = fmaxf(m, fabsf(w[i + e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        unsigned int* tail = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(planes) + h3_plane_bytes(R, image_cols));
        if (__float_as_uint(m) > __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tail, __float_as_uint(m));
    }
}

// B[r][c] of weight_split (gemm.hip) as two f16 planes of w * scale, stored [c'/32][plane][r][c'%32] with c' = the column
// in the padded image (tap * pad32(c2) + channel; c for a linear weight); scale = the power of two of h3_pow2_scale(max|w|),
// max|w| read from the tail (weight_amax_h3_one has run).  Thread i takes SOURCE element i; the thread of a tap's last
// channel also writes that tap's zero padding.
__device__ __forceinline__ void weight_split_h3_one(const float* __restrict__ w, unsigned short* __restrict__ planes, int R,
                                                    int C, int mode, int c2, int taps, long i) {
    const long n = (long)R * C;
    if (i >= n) return;
    const long Cp = h3_image_cols(C, mode >= 2 ? c2 : 0, mode >= 2 ? taps : 0);
    float w_scale, w_inv;
    h3_pow2_scale(*reinterpret_cast<const float*>(reinterpret_cast<const char*>(planes) + h3_plane_bytes(R, Cp)), w_scale, w_inv);
    const int r = (int)(i / C), c = (int)(i % C);
    float v;
    int cp = c, npad = 0;                                   // column in the padded image; zeros to write behind it
    if (mode == 0) v = w[i];
    else if (mode == 1) v = w[(long)c * R + r];
    else {
        const int tap = c / c2, ch = c % c2;
        if (mode == 2) v = w[((long)r * c2 + ch) * taps + tap];
        else v = w[((long)ch * R + r) * taps + tap];
        cp = tap * h3_pad32(c2) + ch;
        if (ch == c2 - 1) npad = h3_pad32(c2) - c2;
    }
    if (mode < 2 && c == C - 1) npad = (int)(Cp - C);
    v *= w_scale;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const long o = ((long)(cp >> 5) * 2 * R + r) * 32 + (cp & 31);
    planes[o] = __builtin_bit_cast(unsigned short, h);
    planes[o + (long)R * 32] = __builtin_bit_cast(unsigned short, l);
    for (int e = 1; e <= npad; ++e) {                       // (cp + e stays inside the k-tile of cp: the pad ends on a multiple of 32)
        planes[o + e] = 0;
        planes[o + e + (long)R * 32] = 0;
    }
}

enum { TILE_AUTO = 0, TILE_64 = 1, TILE_128 = 2, TILE_64x128 = 3, TILE_128x96 = 4, TILE_96x128 = 5,
       H3_TILE_256 = 6, H3_TILE_256x128 = 7,     // 8-wave tiles of the fp16x3 kernels
       H3_TILE_256x128_PAIR = 8 };                 // 4-wave 256x128 tile, one LDS stage, TWO workgroups per CU

// fp16x3 split-precision GEMM (gemm_h3.hip): true when it can take this problem (shape constraints of its 32-deep
// k-tiles); the caller falls back to the bf16x6 kernel otherwise
bool h3_supports(const GemmArgs& g);
int h3_tile_choice(long M, long N, long K);
void launch_weight_split_h3(const float* w, void* planes, int rows, int cols, int mode, int c2, int taps, hipStream_t stream);
int dispatch_h3(const GemmArgs& g, hipStream_t stream);
// row-chunk partials the fp16x3 forward writes into GemmArgs::bn_ws for an M x N x K problem (0: its tile shape cannot)
int h3_bn_blocks(long M, long N, long K);
int dispatch_wgrad_h3(const GemmArgs& g, int zdim, int tile, hipStream_t stream);

}  // namespace ttts
