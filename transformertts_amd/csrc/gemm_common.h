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
    // image-operand kernel (gemm_h3i.hip) only: A is an activation IMAGE (rows of K/16 groups {16 f16 hi, 16 f16 lo} of
    // x * 2^e_row) and a_row_inv[row] = 2^-e_row, the factor its output row is scaled back by; NULL for every other kernel
    const float* a_row_inv;
    // image-operand kernel only, HEAD-IMAGE output (ttts_linear_fwd_h3d_img): C is not fp32 but, per row and 64-column head group,
    // {64 f16 hi, 64 f16 lo} of (x W^T + b) * 2^e(row, group) at the byte offset the fp32 columns would have (row stride ldc * 4
    // bytes), with a per-(row, group) power of two that puts the group's maximum in [2^11, 2^12); c_row_inv[group * M + row] =
    // 2^-e.  c_amax then is an array of N / c_amax_sec partial-maxima arrays, one per section of c_amax_sec columns (q / k / v).
    float* c_row_inv;
    int c_amax_sec;
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
// STORE-DATA HAZARD: a 16-byte buffer store fetches its data registers a few cycles after it issues; a VALU
// instruction that overwrites one of them needs two wait states behind the store (tools/micro/store_war.hip: with fewer, 5 % of
// the stored dwords are the NEW value).  hipcc pads for that itself -- except when the store's soffset is an SGPR
// (GCNHazardRecognizer::createsVALUHazard exempts that form, a rule from older chips; gfx950 is not exempt: round 5's head-image
// corruption).  Every wide store with a scalar offset therefore goes through this helper: the statement behind the store reads
// the four data registers two wait states later, so nothing can recycle them earlier; tools/isa_hazard_scan.py checks the ISA.
__device__ __forceinline__ void store_data_guard(u32x4 v) { asm volatile("s_nop 1" :: "v"(v) : "memory"); }
__device__ __forceinline__ void buf_store4s(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, uint32_t soff, float4 x) {
    const u32x4 v = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, (int)soff, 0);
    store_data_guard(v);
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

// ---- activation IMAGE (the operand format of gemm_h3i.hip): K16-major [K/16][M][{16 f16 hi, 16 f16 lo}] of x * 2^e_row with a
// per-row power of two (the row's maximum lands in [2^11, 2^12)); row_inv[row] = 2^-e_row.  One wave holds one row, a lane NV
// float4 (columns 4 * (lane + 64 v) ..): the producers (LayerNorm forward / backward, ttts_act_image) call this with the row
// they have just computed.  `live`: the lane's v-th float4 lies inside the row.
template <int NV>
__device__ __forceinline__ void image_emit_row(const float4 (&o)[NV], int lane, long row, long M, int K,
                                               unsigned short* __restrict__ img, float* __restrict__ row_inv) {
    float m = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
        if ((lane + 64 * v) * 4 < K) m = fmaxf(fmaxf(m, fmaxf(fabsf(o[v].x), fabsf(o[v].y))), fmaxf(fabsf(o[v].z), fabsf(o[v].w)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float sc, inv;
    h3_pow2_scale(m, sc, inv);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int kk = (lane + 64 * v) * 4;
        if (kk < K) {
            uint2 hi, lo;
            split2_pair(f32x2{o[v].x, o[v].y} * sc, hi.x, lo.x);
            split2_pair(f32x2{o[v].z, o[v].w} * sc, hi.y, lo.y);
            unsigned short* p = img + ((long)(kk >> 4) * M + row) * 32 + (kk & 15);
            *reinterpret_cast<uint2*>(p) = hi;
            *reinterpret_cast<uint2*>(p + 16) = lo;
        }
    }
    if (lane == 0) row_inv[row] = inv;
}

// ---- the split of one weight into its fp16x3 image, in WORK UNITS of one 256-thread workgroup each: unit u is a run of
// H3_SPLIT_RUN 32-row x 32-channel tiles along the channels (row block u / channel runs, channel run u % channel runs) of the
// weight, all taps of them.  (One tile per workgroup made the batched refresh a latency chain per workgroup -- table
// bisection, tail, tile -- of 23 000 workgroups: 75 us per step for 190 MB.)
// h3_split_units = how many an entry has (host and device agree: the prefix sums of the batched table are built from it).
// ttts_weight_split modes 4-7 and 8-11 are the fp16x3 images of modes 0-3: base = which matrix, k16 = which layout
__host__ __device__ __forceinline__ int h3_mode_base(long mode) { return (int)((mode - 4) & 3); }
__host__ __device__ __forceinline__ bool h3_mode_k16(long mode) { return mode >= 8; }
constexpr int H3_SPLIT_TAPS = 4;        // taps staged per pass of a tile (longer kernels take several passes; 17 KB of LDS: 8 workgroups per CU)
constexpr int H3_SPLIT_RUN = 4;         // 32-channel tiles per work unit
__host__ __device__ __forceinline__ long h3_split_units(long R, long C, int mode, int c2) {
    const long chans = (mode & 3) >= 2 ? c2 : C;
    return ((R + 31) / 32) * (((chans + 31) / 32 + H3_SPLIT_RUN - 1) / H3_SPLIT_RUN);
}

// |w| maximum of one weight into the tail of its plane image (atomic max on the bit pattern; the tail was zeroed first).
// The source is R * C contiguous floats whatever the mode: unit u of `units` takes the u-th chunk of it, a float4 per lane
// and trip.  One atomic per wave, behind a look at the tail -- every wave of a weight aims at the same word, and after the
// first few arrivals almost none has anything to add.
// `tail`: where the maximum goes when it is not behind THIS weight's own R x image_cols planes (a stacked image: several
// weights publish into the one tail of the image they share)
__device__ __forceinline__ void weight_amax_h3_unit(const float* __restrict__ w, unsigned short* __restrict__ planes, int R,
                                                    int C, long image_cols, long unit, long units, float* tail_at = nullptr) {
    const long n = (long)R * C;
    const long chunk = ((n + units - 1) / units + 3) & ~3L;
    const long i0 = unit * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    float m = 0.f;
    if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        long i = i0 + 4 * (long)threadIdx.x;
        for (; i + 3 < i1; i += 4 * (long)blockDim.x) {
            const float4 v = *reinterpret_cast<const float4*>(w + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (; i < i1; ++i) m = fmaxf(m, fabsf(w[i]));      // (at most one lane: the last 1-3 elements of the weight)
    } else {
        for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        unsigned int* tail = tail_at != nullptr ? reinterpret_cast<unsigned int*>(tail_at)
                                                : reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(planes) + h3_plane_bytes(R, image_cols));
        if (__float_as_uint(m) > __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(tail, __float_as_uint(m));
    }
}

// B[r][c] of weight_split (gemm.hip) as two f16 planes of w * scale, stored [c'/32][plane][r][c'%32] with c' = the column
// in the padded image (tap * pad32(c2) + channel; c for a linear weight); scale = the power of two of h3_pow2_scale(max|w|),
// max|w| read from the tail (weight_amax_h3_unit has run).  One workgroup of 256 threads per unit: the tile is read in the
// order the SOURCE is contiguous in (rows of channels, columns of rows, or a row's / a channel's run over all taps), goes
// through LDS (t: H3_SPLIT_TAPS x 32 x 33 floats) and leaves as 2 KB runs of each plane -- one k-tile's 32 rows, 8 bytes per
// lane.  Channels past the tap's last one are staged as zeros, which writes the image's padding.
// k16 = true (modes 8-11 of ttts_weight_split): the K16-MAJOR image the image-operand kernel (gemm_h3i.hip) stages by LDS-DMA --
// [c'/16][r][{16 f16 hi, 16 f16 lo}], i.e. one 64-byte group per (16-deep k-tile, row) and the 16 KB a 256-row tile needs per
// k-tile in ONE contiguous run of whole 128-byte lines.  Same bytes, same padding, same tail.
// STACKED images (linear weights only; the fused cross-attention K/V projection of all decoder layers): the R x C block this
// call splits is a WINDOW of a larger image of Rimg rows -- image rows r_off .. r_off + R - 1 (the forward image: the output
// columns of several weights side by side) or image columns c_off .. (the data-gradient image: their reduction indices one
// behind the other; c_off a multiple of 32).  Rimg = 0: the image is the block itself.  All windows of an image share its
// tail, i.e. one scale (`amax` then points at it).
__device__ __forceinline__ void weight_split_h3_tile(const float* __restrict__ w, unsigned short* __restrict__ planes, int R,
                                                     int C, int mode, int c2, int taps, long unit, float (*t)[32][33],
                                                     bool k16 = false, const float* amax = nullptr, int Rimg = 0, int r_off = 0,
                                                     int c_off = 0) {
    const int chans = mode >= 2 ? c2 : C;
    if (mode < 2) taps = 1;
    if (Rimg <= 0) { Rimg = R; r_off = 0; c_off = 0; }
    const int ngc = (chans + 31) / 32, nrun = (ngc + H3_SPLIT_RUN - 1) / H3_SPLIT_RUN;
    const int r0 = (int)(unit / nrun) * 32, cb0 = (int)(unit % nrun) * H3_SPLIT_RUN;
    const int nr = min(32, R - r0);
    const int padc = h3_pad32(chans);
    float w_scale, w_inv;
    // (amax: where max|w| stands when it is not this image's own tail -- the batched refresh measures a weight once for all its images)
    h3_pow2_scale(amax != nullptr ? *amax : *h3_plane_tail(planes, R, (long)taps * padc), w_scale, w_inv);
    const int tid = threadIdx.x;
    if (mode == 0 && (C & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        // a linear weight in its own orientation: image rows are weight rows, so a lane's four channels are one 16-byte load and
        // nothing has to turn in LDS; the run's loads are all requested before the first conversion
        const int rr = tid >> 3, q = (tid & 7) * 4;
        float4 v4[H3_SPLIT_RUN];
#pragma unroll
        for (int u = 0; u < H3_SPLIT_RUN; ++u) {
            const int c = (cb0 + u) * 32 + q;
            v4[u] = (rr < nr && c < C) ? *reinterpret_cast<const float4*>(w + (long)(r0 + rr) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < H3_SPLIT_RUN; ++u) {
            const int cp = c_off + (cb0 + u) * 32 + q;
            if (rr < nr && cb0 + u < ngc) {
                const float vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
                unsigned short h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = vv[e] * w_scale;
                    const _Float16 hh = (_Float16)v;
                    const _Float16 ll = (_Float16)(v - (float)hh);
                    h[e] = __builtin_bit_cast(unsigned short, hh);
                    l[e] = __builtin_bit_cast(unsigned short, ll);
                }
                const long o = k16 ? ((long)(cp >> 4) * Rimg + r_off + r0 + rr) * 32 + (cp & 15)
                                   : ((long)(cp >> 5) * 2 * Rimg + r_off + r0 + rr) * 32 + (cp & 31);
                *reinterpret_cast<uint2*>(planes + o) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<uint2*>(planes + o + (k16 ? 16L : (long)Rimg * 32)) =
                    make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
            }
        }
        return;
    }
    for (int cb = cb0; cb < cb0 + H3_SPLIT_RUN && cb < ngc; ++cb) {
    const int ch0 = cb * 32;
    const int nc = min(32, chans - ch0);
    for (int tc0 = 0; tc0 < taps; tc0 += H3_SPLIT_TAPS) {
        const int tcn = min(H3_SPLIT_TAPS, taps - tc0), run = 32 * tcn;
        if (mode == 0) {
            for (int idx = tid; idx < 1024; idx += 256) {
                const int rr = idx >> 5, cc = idx & 31;
                t[0][rr][cc] = (rr < nr && cc < nc) ? w[(long)(r0 + rr) * C + ch0 + cc] : 0.f;
            }
        } else if (mode == 1) {
            for (int idx = tid; idx < 1024; idx += 256) {
                const int cc = idx >> 5, rr = idx & 31;
                t[0][rr][cc] = (rr < nr && cc < nc) ? w[(long)(ch0 + cc) * R + r0 + rr] : 0.f;
            }
        } else if (mode == 2) {                             // w[co][ci][tap]: a row's channels x taps are one run
            for (int idx = tid; idx < 32 * run; idx += 256) {
                const int rr = idx / run, j = idx - rr * run, cc = j / tcn, tt = j - cc * tcn;
                t[tt][rr][cc] = (rr < nr && cc < nc) ? w[((long)(r0 + rr) * c2 + ch0 + cc) * taps + tc0 + tt] : 0.f;
            }
        } else {                                            // w[co][ci][tap] read as [ci] rows: a channel's rows x taps are one run
            for (int idx = tid; idx < 32 * run; idx += 256) {
                const int cc = idx / run, j = idx - cc * run, rr = j / tcn, tt = j - rr * tcn;
                t[tt][rr][cc] = (rr < nr && cc < nc) ? w[((long)(ch0 + cc) * R + r0 + rr) * taps + tc0 + tt] : 0.f;
            }
        }
        __syncthreads();
        const int rr = tid >> 3, q = (tid & 7) * 4;
        if (rr < nr) {
            for (int tt = 0; tt < tcn; ++tt) {
                const int cp = c_off + (mode >= 2 ? (tc0 + tt) * padc : 0) + ch0 + q;
                unsigned short h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = t[tt][rr][q + e] * w_scale;
                    const _Float16 hh = (_Float16)v;
                    const _Float16 ll = (_Float16)(v - (float)hh);
                    h[e] = __builtin_bit_cast(unsigned short, hh);
                    l[e] = __builtin_bit_cast(unsigned short, ll);
                }
                const long o = k16 ? ((long)(cp >> 4) * Rimg + r_off + r0 + rr) * 32 + (cp & 15)
                                   : ((long)(cp >> 5) * 2 * Rimg + r_off + r0 + rr) * 32 + (cp & 31);
                *reinterpret_cast<uint2*>(planes + o) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<uint2*>(planes + o + (k16 ? 16L : (long)Rimg * 32)) =
                    make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
            }
        }
        __syncthreads();
    }
    }
}

enum { TILE_AUTO = 0, TILE_64 = 1, TILE_128 = 2, TILE_64x128 = 3, TILE_128x96 = 4, TILE_96x128 = 5,
       H3_TILE_256 = 6, H3_TILE_256x128 = 7,     // 8-wave tiles of the fp16x3 kernels
       H3_TILE_256x128_PAIR = 8 };                 // 4-wave 256x128 tile, one LDS stage, TWO workgroups per CU

// fp16x3 split-precision GEMM (gemm_h3.hip): true when it can take this problem (shape constraints of its 32-deep
// k-tiles); the caller falls back to the bf16x6 kernel otherwise
bool h3_supports(const GemmArgs& g);
int h3_tile_choice(long M, long N, long K);
void launch_weight_split_h3(const float* w, void* planes, int rows, int cols, int mode, int c2, int taps, hipStream_t stream,
                            bool k16 = false);
int dispatch_h3(const GemmArgs& g, hipStream_t stream);
// row-chunk partials the fp16x3 forward writes into GemmArgs::bn_ws for an M x N x K problem (0: its tile shape cannot)
int h3_bn_blocks(long M, long N, long K);
int h3_bn_chunk_rows(long M, long N, long K);
int dispatch_wgrad_h3(const GemmArgs& g, int zdim, int tile, hipStream_t stream);
// XCD-aware numbering of the weight-gradient grids: workgroups are dealt round-robin to the 8 XCDs in x-fastest order, so the
// tiles of one row split -- which read the same rows of dy and x -- would land on different L2s and each fetch its
// operands from HBM again (a 1024 x 256 weight asks for 3.2x the unique bytes).  Renumbered, XCD e works on one
// contiguous run of (split, tile) pairs: the tiles of a split start together on the CUs of one XCD and share its L2.
__device__ __forceinline__ int xcd_renumber(int bid, int total) {
    const int per = total >> 3, rem = total & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    return xcd * per + min(xcd, rem) + slot;
}
// GROUPED weight-gradient launches (wgrad_h3_group_kernel, wgrad_dma_group_kernel): up to WG_GROUP_MAX independent problems as
// one grid; first[p] = first flat workgroup of problem p (first[n] = total); inside a problem the numbering is (tile x, tile y,
// split) as in the one-problem kernels
constexpr int WG_GROUP_MAX = 4;
struct WgradGroupArgs {
    GemmArgs g[WG_GROUP_MAX];
    int first[WG_GROUP_MAX + 1];
    int n;
};
// n <= 4 independent problems on the 4-wave 128 x 128 tile as ONE grid (gemm_h3.hip, wgrad_h3_group_kernel)
int launch_wgrad_h3_group(const GemmArgs* gs, const int* zdims, int n, int tile, hipStream_t stream);
// the same problem with LDS-DMA staged operand rows (wgrad_dma.hip): whole 256 x 256 tiles only
bool wgrad_dma_supports(const GemmArgs& g, int tile);
int launch_wgrad_dma(const GemmArgs& g, int zdim, hipStream_t stream);
int launch_wgrad_dma_group(const GemmArgs* gs, const int* zdims, int n, hipStream_t stream);
// image-operand fp16x3 GEMM (gemm_h3i.hip): both operands staged by LDS-DMA, 128 x 256 tile, two workgroups per CU
bool h3i_supports(const GemmArgs& g);
int dispatch_h3i(const GemmArgs& g, hipStream_t stream);

}  // namespace ttts
