// fp32 MFMA GEMM family for gfx950: linear / 1x1-conv FFN / k-tap Conv1d as implicit GEMM,
// forward (x.W^T), data-gradient (dy.W) and weight-gradient (dy^T.x, split over rows).
//
// One workgroup = 4 waves (256 threads) computes a BM x BN tile with v_mfma_f32_32x32x2_f32
// (exact fp32, k-ordered fma chain).  Operand tiles are staged global -> registers -> LDS
// (double-buffered, one barrier per 16-deep k-tile); each wave owns a (BM/WM) x (BN/WN) sub-tile
// made of 32x32 MFMA accumulators.
//
// Operand addressing modes (template flags):
//   A_KC  : A is [M][K] row-major, K contiguous ("activation rows").  Supports an implicit
//           im2col: K = taps*cin, k-tile -> (tap, channel block), source row = m + shift(tap),
//           zero outside the utterance [0,T) the row belongs to.  This serves Conv1d forward,
//           Conv1d data-gradient (flipped taps) and the decoder's go-frame shift (one tap, shift -1).
//   !A_KC : A is stored [K][M] (reduction index is the row) -- dy^T for weight gradients.
//   B_KC  : B is [N][K] row-major, K contiguous (nn.Linear weight layout, packed conv weight).
//   !B_KC : B is stored [K][N] (reduction index is the row) -- W for data-gradients, x for weight
//           gradients (with the same per-tap row shift / utterance clipping as above).
#include <stdlib.h>

#include "gemm_common.h"

namespace ttts {

#ifndef TTTS_GEMM_BK
#define TTTS_GEMM_BK 16
#endif
#ifndef TTTS_X6_XCD
#define TTTS_X6_XCD 1
#endif
#ifndef TTTS_GEMM_MINWAVES
#define TTTS_GEMM_MINWAVES 3
#endif
constexpr int BK = TTTS_GEMM_BK;          // k-tile depth (floats)
constexpr int KC_LD = BK + 1;   // LDS row stride for K-contiguous tiles (odd -> conflict-free b32 reads)


template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, TTTS_GEMM_MINWAVES) void gemm_f32_kernel(GemmArgs g) {
    const uint64_t seed_eff = site_seed(g.seed, g.step_seed);
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");
    constexpr int A_ELEMS = A_KC ? BM * KC_LD : BK * BM;
    constexpr int B_ELEMS = B_KC ? BN * KC_LD : BK * BN;
    constexpr int NLA = (BM * 4 + 255) / 256;   // float4 loads per thread for the A tile
    constexpr int NLB = (BN * 4 + 255) / 256;

    __shared__ __attribute__((aligned(16))) float lds[2 * (A_ELEMS + B_ELEMS)];
    constexpr int STAGE = A_ELEMS + B_ELEMS;   // stage `buf`: A at lds + buf*STAGE, B right behind it

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int ztap = (g.ztaps > 1) ? (z % g.ztaps) : 0;
    const int zsplit = (g.ztaps > 1) ? (z / g.ztaps) : z;

    const int nkt = g.K / BK + ((g.K % BK) ? 1 : 0);
    const int kt_begin = zsplit * g.kt_per_split;
    int kt_end = kt_begin + g.kt_per_split;
    if (kt_end > nkt) kt_end = nkt;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);

    float4 ra[NLA], rb[NLB];
    float4 csum[NLA];   // running sum of this thread's A elements over the k-tiles (bias gradient, !A_KC form)
#pragma unroll
    for (int i = 0; i < NLA; ++i) csum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- per-thread static parts of the loaders: byte offsets of this thread's float4s at k-tile 0
    // K-contiguous tiles: thread -> (row = idx>>2, chunk = idx&3); natural tiles: (krow = idx / (X/4), col4)
    int a_t[NLA];          // A_KC: t index of the row inside its utterance (for shift clipping)
    uint32_t a_off[NLA];   // OOB when the row / column is outside the matrix
    uint32_t b_off[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        int idx = tid + i * 256;
        if (A_KC) {
            int row = idx >> 2, ch = idx & 3;
            int m = m0 + row;
            a_off[i] = (row < BM && m < g.M) ? (uint32_t)(((long)m * g.lda + ch * 4) * 4) : OOB;
            a_t[i] = (g.T > 0) ? (m % g.T) : 0;
        } else {
            int kr = idx / (BM / 4), c4 = idx % (BM / 4);
            int m = m0 + c4 * 4;
            a_off[i] = (idx < BK * (BM / 4) && m < g.M) ? (uint32_t)(((long)kr * g.lda + m) * 4) : OOB;
            a_t[i] = kr;
        }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        int idx = tid + i * 256;
        if (B_KC) {
            int row = idx >> 2, ch = idx & 3;
            int n = n0 + row;
            b_off[i] = (row < BN && n < g.N) ? (uint32_t)(((long)n * g.ldb + ch * 4) * 4) : OOB;
        } else {
            int kr = idx / (BN / 4), c4 = idx % (BN / 4);
            int n = n0 + c4 * 4;
            b_off[i] = (idx < BK * (BN / 4) && n < g.N) ? (uint32_t)(((long)kr * g.ldb + n) * 4) : OOB;
        }
    }

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        // ---------------- A
        if (A_KC) {
            const int tap = k0 / g.cin;
            const int c0 = k0 - tap * g.cin;
            const int shift = g.shift0 + tap * g.shift_step;
            const uint32_t koff = (uint32_t)(((long)shift * g.lda + c0) * 4);
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                bool ok = a_off[i] != OOB;
                if (g.T > 0) ok = ok && ((unsigned)(a_t[i] + shift) < (unsigned)g.T);
                ra[i] = buf_load4(rsrcA, ok ? a_off[i] + koff : OOB);
            }
        } else {
            const uint32_t koff = (uint32_t)((long)k0 * g.lda * 4);
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                bool ok = a_off[i] != OOB && (k0 + a_t[i]) < g.K;
                ra[i] = buf_load4(rsrcA, ok ? a_off[i] + koff : OOB);
                csum[i].x += ra[i].x; csum[i].y += ra[i].y; csum[i].z += ra[i].z; csum[i].w += ra[i].w;
            }
        }
        // ---------------- B
        if (B_KC) {
            const uint32_t koff = (uint32_t)(k0 * 4);
#pragma unroll
            for (int i = 0; i < NLB; ++i) rb[i] = buf_load4(rsrcB, b_off[i] != OOB ? b_off[i] + koff : OOB);
        } else {
            const int shift = g.shift0 + ztap * g.shift_step;
            const uint32_t koff = (uint32_t)((long)(k0 + shift) * g.ldb * 4);
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                int idx = tid + i * 256;
                int k = k0 + idx / (BN / 4);
                bool ok = b_off[i] != OOB && k < g.K;
                if (g.T > 0) ok = ok && ((unsigned)((k % g.T) + shift) < (unsigned)g.T);
                rb[i] = buf_load4(rsrcB, ok ? b_off[i] + koff : OOB);
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* as = lds + buf * STAGE;
        float* bs = as + A_ELEMS;
        if (A_KC) {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                int idx = tid + i * 256;
                int row = idx >> 2, ch = idx & 3;
                if (row < BM) {
                    float* d = as + row * KC_LD + ch * 4;
                    d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                int idx = tid + i * 256;
                if (idx < BK * (BM / 4)) {
                    int kr = idx / (BM / 4), c4 = idx % (BM / 4);
                    *reinterpret_cast<float4*>(as + kr * BM + c4 * 4) = ra[i];
                }
            }
        }
        if (B_KC) {
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                int idx = tid + i * 256;
                int row = idx >> 2, ch = idx & 3;
                if (row < BN) {
                    float* d = bs + row * KC_LD + ch * 4;
                    d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLB; ++i) {
                int idx = tid + i * 256;
                if (idx < BK * (BN / 4)) {
                    int kr = idx / (BN / 4), c4 = idx % (BN / 4);
                    *reinterpret_cast<float4*>(bs + kr * BN + c4 * 4) = rb[i];
                }
            }
        }
    };

    auto compute = [&](int buf) {
        const float* as = lds + buf * STAGE;
        const float* bs = as + A_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
            const int k = kk * 2 + half;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int m = wm * WTM + i * 32 + l31;
                a[i] = A_KC ? as[m * KC_LD + k] : as[k * BM + m];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = wn * WTN + j * 32 + l31;
                b[j] = B_KC ? bs[n * KC_LD + k] : bs[k * BN + n];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
        __syncthreads();
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = (kt + 1) < kt_end;
            if (more) load_tiles(kt + 1);
            compute(buf);
            if (more) store_tiles(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    if constexpr (!A_KC && !B_KC) {
        // bias-gradient partials: sum over this split's rows of dy for the BM output channels of the tile
        if (g.colsum != nullptr && blockIdx.x == 0 && ztap == 0) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NLA; ++i) {
                int idx = tid + i * 256;
                if (idx < BK * (BM / 4)) {
                    int kr = idx / (BM / 4), c4 = idx % (BM / 4);
                    *reinterpret_cast<float4*>(lds + kr * BM + c4 * 4) = csum[i];
                }
            }
            __syncthreads();
            if (tid < BM && m0 + tid < g.M) {
                float sacc = 0.f;
#pragma unroll
                for (int kr = 0; kr < BK; ++kr) sacc += lds[kr * BM + tid];
                g.colsum[(long)zsplit * g.M + m0 + tid] = sacc;
            }
        }
    }

    // ---------------- epilogue
    float* C = g.C + (long)z * g.c_zstride;
    const bool do_drop = g.drop_thr != 0u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WTN + j * 32 + l31;
            const bool col_ok = col < g.N;
            const float bv = (g.bias != nullptr && col_ok) ? g.bias[col] : 0.f;
            const int row0 = m0 + wm * WTM + i * 32;
            float res[16];
            // all residual loads of the sub-tile are issued before the first store (one wait, not sixteen)
            if (g.residual != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + acc_row(r, half);
                    res[r] = (row < g.M && col_ok) ? g.residual[(long)row * g.ldr + col] : 0.f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) res[r] = 0.f;
            }
            float gate[16];
            if (g.relu_out != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + acc_row(r, half);
                    gate[r] = (row < g.M && col_ok && g.relu_out[(long)row * g.ldc + col] > 0.f) ? g.relu_scale : 0.f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) gate[r] = 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + acc_row(r, half);
                float v = acc[i][j][r] + bv;
                if (g.act == 1) v = fmaxf(v, 0.f);
                if (do_drop) {
                    uint64_t idx = (uint64_t)row * (uint64_t)g.N + (uint64_t)col;
                    v = keep_elem(seed_eff, idx, g.drop_thr) ? v * g.drop_scale : 0.f;
                }
                v = v * gate[r] + res[r];
                if (row < g.M && col_ok) C[(long)row * g.ldc + col] = v;
            }
        }
    }
}

// w[co][ci][tap] -> wp[co][tap][ci]  (forward: K-contiguous rows per output channel)
//                -> wt[ci][tap][co]  (data gradient: K-contiguous rows per input channel)
__global__ void conv_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wt,
                                        int cout, int cin, int taps) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long n = (long)cout * cin * taps;
    if (i >= n) return;
    int tap = (int)(i % taps);
    long r = i / taps;
    int ci = (int)(r % cin);
    int co = (int)(r / cin);
    float v = w[i];
    if (wp) wp[((long)co * taps + tap) * cin + ci] = v;
    if (wt) wt[((long)ci * taps + tap) * cout + co] = v;
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC>
static int launch_gemm(const GemmArgs& g, int zdim, hipStream_t stream) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), zdim);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, A_KC, B_KC>), grid, dim3(256), 0, stream, g);
    TTTS_LAUNCH_CHECK("gemm_f32_kernel");
    return TTTS_OK;
}


// Pick the tile that minimises the busiest CU's work: ceil(tiles / 256 CUs) workgroups in sequence, each costing
// its area divided by how efficiently that tile shape runs.  E.g. M = 55 680, N = 256 in fp32: 870 tiles of 128x128
// are 4 deep on some CUs (65 536), 1 740 tiles of 64x128 are 7 deep at 0.93 efficiency (61 660) -> 64x128.
// The split-precision kernel is latency / traffic bound, so small tiles cost it more (second efficiency table).
static int choose_tile(long M, long N, long zdim, bool x6 = false) {
    struct Cand { int tile, bm, bn; float eff_f32, eff_x6; };
    const Cand cands[] = {{TILE_128, 128, 128, 1.00f, 1.00f}, {TILE_64x128, 64, 128, 0.93f, 0.75f}, {TILE_64, 64, 64, 0.80f, 0.60f}};
    int best = TILE_128;
    float best_cost = 1e30f;
    for (const Cand& c : cands) {
        long tiles = (long)cdiv(M, c.bm) * cdiv(N, c.bn) * zdim;
        float cost = (float)((tiles + 255) / 256) * (float)(c.bm * c.bn) / (x6 ? c.eff_x6 : c.eff_f32);
        if (cost < best_cost) { best_cost = cost; best = c.tile; }
    }
    return best;
}

template <bool A_KC, bool B_KC>
static int dispatch_gemm(const GemmArgs& g, int zdim, int tile, hipStream_t stream) {
    if (tile == TILE_AUTO) {
        if (g.N <= 96 && (long)cdiv(g.M, 128) * zdim >= 384) tile = TILE_128x96;   // 80-wide mel GEMMs
        else tile = choose_tile(g.M, g.N, zdim);
    }
    switch (tile) {
        case TILE_64: return launch_gemm<64, 64, 2, 2, A_KC, B_KC>(g, zdim, stream);
        case TILE_64x128: return launch_gemm<64, 128, 2, 2, A_KC, B_KC>(g, zdim, stream);
        case TILE_128x96: return launch_gemm<128, 96, 4, 1, A_KC, B_KC>(g, zdim, stream);
        default: return launch_gemm<128, 128, 2, 2, A_KC, B_KC>(g, zdim, stream);
    }
}

// ===============================================================================================================
// Split-precision variant ("bf16x6"): fp32-accurate products on the 16x faster bf16 MFMA.
// Every fp32 operand value is written as hi + mid + lo, three bf16 numbers (8 mantissa bits each, so 24 bits: the
// split is exact up to 2^-25 relative), and the six leading cross products
//     a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a2*b2 + a3*b1)
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact in fp32).  The dropped
// terms are O(2^-24) relative: measured error of a K = 256..1024 GEMM against fp64 is 1.1e-7, BELOW the 2.9e-7 of
// a plain fp32 fma chain, at 6/16 of the fp32-MFMA cycle count.
// Activations (A) are split on the fly while they are staged into LDS; weights (B) are split once per call by
// weight_split_kernel into bf16 planes laid k-tile-major ([K/16][plane][N][16]), so the B loader moves whole lines.
// LDS image per operand plane: [row][16 k] bf16 = 32-byte rows; the two 16-byte chunks of a row are swapped on rows
// with bit 3 set (chunk ^= (row >> 3) & 1), which makes the ds_read_b128 of the MFMA fragments conflict-free.

// B[r][c] (R rows of C bf16 per plane p = hi, mid, lo; stored as [c/16][p][r][c%16]) from a weight tensor:
//   mode 0: linear forward      B[r][c] = w[r*C + c]                      (w is (R, C))
//   mode 1: linear data-grad    B[r][c] = w[c*R + r]                      (w is (C, R); B = w^T)
//   mode 2: conv forward        B[co][tap*cin + ci] = w[(co*cin + ci)*taps + tap]       (R = cout, C = taps*cin)
//   mode 3: conv data-grad      B[ci][tap*cout + co] = w[(co*cin + ci)*taps + tap]      (R = cin,  C = taps*cout)
__device__ __forceinline__ void weight_split_one(const float* __restrict__ w, unsigned short* __restrict__ planes, int R,
                                                 int C, int mode, int c2, int taps, long i) {
    const long n = (long)R * C;
    if (i >= n) return;
    const int r = (int)(i / C), c = (int)(i % C);
    float v;
    if (mode == 0) v = w[i];
    else if (mode == 1) v = w[(long)c * R + r];
    else {
        const int tap = c / c2, ch = c % c2;          // c2 = channels per tap
        if (mode == 2) v = w[((long)r * c2 + ch) * taps + tap];          // r = co, ch = ci, c2 = cin
        else v = w[((long)ch * R + r) * taps + tap];                     // r = ci, ch = co, R = cin
    }
    __bf16 b1 = (__bf16)v;
    float r1 = v - (float)b1;
    __bf16 b2 = (__bf16)r1;
    float r2 = r1 - (float)b2;
    __bf16 b3 = (__bf16)r2;
    // k-tile-major image: [c / 16][plane][r][c % 16], so the BN x 16 block a workgroup stages per k-step and plane is
    // one contiguous run (whole 128-byte lines; a [r][c] image gave 32-byte row pieces and 4x the L2->L1 traffic)
    const long o = ((long)(c >> 4) * 3 * R + r) * 16 + (c & 15);
    const long ps = (long)R * 16;
    planes[o] = __builtin_bit_cast(unsigned short, b1);
    planes[o + ps] = __builtin_bit_cast(unsigned short, b2);
    planes[o + 2 * ps] = __builtin_bit_cast(unsigned short, b3);
}
__global__ __launch_bounds__(256) void weight_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes,
                                                           int R, int C, int mode, int c2, int taps) {
    weight_split_one(w, planes, R, C, mode, c2, taps, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// every weight of the model in one launch (after an optimizer step): descriptor i = 8 int64
// {w, planes, R, C, mode, c2, taps, first block}; a workgroup finds its descriptor by bisection on the block starts
// descriptor of the weight that block `blk` of a batched launch works on (first_block ascending)
__device__ __forceinline__ const long* batched_desc(const long* __restrict__ descs, int n, long blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid * 8 + 7] <= blk) lo = mid; else hi = mid - 1;
    }
    return descs + lo * 8;
}

// fp16x3 images (modes 4-7) carry max|w| in their tail: zero the tails, then one atomic max per wave
// A LINEAR entry (modes 4, 5, 8, 9) may describe a WINDOW of a stacked image (weight_split_h3_tile): d[5] = (image rows << 32) |
// image columns, d[6] = (first image row << 32) | first image column of the window; d[5] = 0: the entry is its own image.
struct DescGeom { long Rimg, Cimg; int r_off, c_off; bool stacked; };
__device__ __forceinline__ DescGeom desc_geom(const long* d) {
    DescGeom g;
    const bool conv = h3_mode_base(d[4]) >= 2;
    g.stacked = !conv && d[5] != 0;
    g.Rimg = g.stacked ? (d[5] >> 32) : d[2];
    g.Cimg = g.stacked ? (d[5] & 0xffffffffL) : d[3];
    g.r_off = g.stacked ? (int)(d[6] >> 32) : 0;
    g.c_off = g.stacked ? (int)(d[6] & 0xffffffffL) : 0;
    return g;
}
__device__ __forceinline__ float* batched_tail(const long* d) {
    const bool conv = h3_mode_base(d[4]) >= 2;
    const DescGeom g = desc_geom(d);
    return reinterpret_cast<float*>(reinterpret_cast<char*>(d[1]) +
                                    h3_plane_bytes(g.Rimg, h3_image_cols(g.Cimg, conv ? (int)d[5] : 0, conv ? (int)d[6] : 0)));
}
__global__ __launch_bounds__(256) void weight_tail_zero_batched_kernel(const long* __restrict__ descs, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long* d = descs + i * 8;
        if (d[4] >= 4) *batched_tail(d) = 0.f;
    }
}
// The images of ONE weight (forward / data-gradient, row-major / K16-major: up to four table entries, adjacent because the table
// is built parameter by parameter) share max|w|: the first of them -- the leader -- measures it, the others copy it while they
// are split (weight_split_batched_kernel).  Same source address and element count = same tensor.  Windows of stacked images
// share only among themselves: the tail they publish into holds the maximum over EVERY weight of the stack (each window's
// leader adds its own), which a plain image of one of those weights must not take for its own scale.
__device__ __forceinline__ const long* batched_leader(const long* __restrict__ descs, const long* d) {
    const long* l = d;
    const bool stacked = desc_geom(d).stacked;
    while (l > descs && (l - 8)[0] == d[0] && (l - 8)[2] * (l - 8)[3] == d[2] * d[3] && (l - 8)[4] >= 4 &&
           desc_geom(l - 8).stacked == stacked)
        l -= 8;
    return l;
}

__global__ __launch_bounds__(256) void weight_amax_batched_kernel(const long* __restrict__ descs, int n) {
    const long blk = blockIdx.x;
    const long* d = batched_desc(descs, n, blk);
    if (d[4] >= 4 && batched_leader(descs, d) == d) {
        const bool conv = h3_mode_base(d[4]) >= 2;
        weight_amax_h3_unit(reinterpret_cast<const float*>(d[0]), reinterpret_cast<unsigned short*>(d[1]), (int)d[2], (int)d[3],
                            h3_image_cols(d[3], conv ? (int)d[5] : 0, conv ? (int)d[6] : 0), blk - d[7],
                            h3_split_units(d[2], d[3], h3_mode_base(d[4]), conv ? (int)d[5] : 0), batched_tail(d));
    }
}

__global__ __launch_bounds__(256) void weight_split_batched_kernel(const long* __restrict__ descs, int n) {
    __shared__ float t[H3_SPLIT_TAPS][32][33];
    const long blk = blockIdx.x;
    const long* d = batched_desc(descs, n, blk);
    if (d[4] >= 4) {    // modes 4-7: the fp16x3 image of modes 0-3 (block-uniform branch), a 32 x 32 tile per workgroup
        const long* lead = batched_leader(descs, d);
        const float* amax = batched_tail(lead);
        // the GEMMs read the image's own tail (for the windows of a stacked image: every window writes the same value)
        if (lead != d && blk == d[7] && threadIdx.x == 0) *batched_tail(d) = *amax;
        const DescGeom g = desc_geom(d);
        const bool conv = h3_mode_base(d[4]) >= 2;
        weight_split_h3_tile(reinterpret_cast<const float*>(d[0]), reinterpret_cast<unsigned short*>(d[1]), (int)d[2],
                             (int)d[3], h3_mode_base(d[4]), conv ? (int)d[5] : 0, conv ? (int)d[6] : 0, blk - d[7], t, h3_mode_k16(d[4]), amax,
                             g.stacked ? (int)g.Rimg : 0, g.r_off, g.c_off);
    } else
        weight_split_one(reinterpret_cast<const float*>(d[0]), reinterpret_cast<unsigned short*>(d[1]), (int)d[2], (int)d[3],
                         (int)d[4], (int)d[5], (int)d[6], (blk - d[7]) * 256 + threadIdx.x);
}

template <int BM, int BN, int WM, int WN, bool CLIP>
__global__ __launch_bounds__(256, TTTS_GEMM_MINWAVES) void gemm_bf16x6_kernel(GemmArgs g) {
    const uint64_t seed_eff = site_seed(g.seed, g.step_seed);
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && BK == 16, "4 waves per workgroup, one MFMA k-step per k-tile");
    static_assert((BM * 4) % 256 == 0, "every thread stages whole float4s of the A tile");
    constexpr int A_PLANE = BM * 8, B_PLANE = BN * 8;          // dwords per plane (32-byte rows)
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
    constexpr int NLA = (BM * 4 + 255) / 256;                  // float4 loads per thread for the A tile
    constexpr bool B_ALL = (BN * 2 >= 256);                    // B tile: BN rows x 2 half-chunks of 16 B per plane

    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
#if TTTS_X6_XCD
    // Workgroups are dispatched x-fastest and dealt round-robin to the 8 XCDs, each with its own L2.  Renumber so
    // that XCD e works on one contiguous run of tiles (column blocks of a row panel first): the nx workgroups that
    // read the same A rows then share an L2 (1 miss + nx-1 hits instead of nx fetches from the Infinity Cache / HBM,
    // whose per-CU rate -- 10-14 B/clk -- is what bounds this kernel otherwise).
    const int nx = gridDim.x;
    const int ntiles = nx * gridDim.y;
    const int bid = blockIdx.y * nx + blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int t = xcd * per + min(xcd, rem) + slot;
    const int ty = t / nx, tx = t - ty * nx;
    const int m0 = ty * BM, n0 = tx * BN;
#else
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
#endif
    const int nkt = g.K / BK;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);
    const uint32_t b_plane_bytes = (uint32_t)g.N * 32u;           // one plane of one k-tile: N rows of 16 bf16

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NLA];
    u32x4 rb[3];

    // Loader state.  No per-load validity arithmetic except the utterance clipping of shifted rows (CLIP): rows past the
    // end of an operand fall outside its buffer descriptor (hardware returns 0), weight rows past N only feed accumulator
    // columns that are never stored.  The k position (tap, channel offset, row shift) advances incrementally: loads are
    // requested in k order, each k-tile exactly once.
    int a_t[NLA];
    uint32_t a_off[NLA];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int idx = tid + i * 256;
        const int row = idx >> 2, ch = idx & 3;
        const int m = m0 + row;
        a_off[i] = (uint32_t)(((long)m * g.lda + ch * 4) * 4);
        a_t[i] = CLIP ? (m % g.T) : 0;
    }
    // B: thread -> (row = tid>>1, half-chunk = tid&1): 8 bf16 = 16 bytes per plane
    const int b_row = tid >> 1, b_hc = tid & 1;
    uint32_t b_cur = (uint32_t)((n0 + b_row) * 32 + b_hc * 16);
    int k_c0 = 0, k_shift = g.shift0;
    uint32_t k_off = (uint32_t)((long)g.shift0 * g.lda * 4);
    const uint32_t tap_step = (uint32_t)(((long)g.shift_step * g.lda - g.cin) * 4);

    auto load_a = [&](int) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            uint32_t off = a_off[i] + k_off;
            if (CLIP) off = ((unsigned)(a_t[i] + k_shift) < (unsigned)g.T) ? off : OOB;
            ra[i] = buf_load4(rsrcA, off);
        }
        k_c0 += BK;
        k_off += BK * 4;
        if (k_c0 == g.cin) { k_c0 = 0; k_shift += g.shift_step; k_off += tap_step; }
    };
    auto load_b = [&](int) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            rb[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)(b_cur + p * b_plane_bytes), 0, 0);
        b_cur += 3u * b_plane_bytes;
    };

    auto store_a = [&](int buf, int i) {
        uint32_t* as = lds + buf * STAGE;
        int idx = tid + i * 256;
        int row = idx >> 2, ch = idx & 3;
        if ((BM * 4) % 256 == 0 || row < BM) {
            uint2 hi, mid, lo;
            split3_pack4(ra[i], hi, mid, lo);
            const int d = row * 8 + (((ch >> 1) ^ ((row >> 3) & 1)) * 4) + (ch & 1) * 2;
            *reinterpret_cast<uint2*>(as + d) = hi;
            *reinterpret_cast<uint2*>(as + A_PLANE + d) = mid;
            *reinterpret_cast<uint2*>(as + 2 * A_PLANE + d) = lo;
        }
    };
    auto store_b = [&](int buf) {
        uint32_t* bs = lds + buf * STAGE + 3 * A_PLANE;
        if (B_ALL || b_row < BN) {
            const int d = b_row * 8 + ((b_hc ^ ((b_row >> 3) & 1)) * 4);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(bs + p * B_PLANE + d) = rb[p];
        }
    };

    // One k-step.  LDS[buf] holds tile kt, the staging registers hold tile kt+1 (requested in the middle of the previous
    // step).  After the first accumulator tile's MFMAs the registers are split / written to LDS[buf^1] and immediately
    // re-used to request tile kt+2: every load has a whole k-step to land and the staging VALU / LDS work sits in the
    // shadow of the remaining MFMAs, on ONE register set.
    auto step = [&](int buf, int kt_next2, bool stage, bool fetch) {
        const uint32_t* as = lds + buf * STAGE;
        const uint32_t* bs = as + 3 * A_PLANE;
        const int cw = ((half ^ ((l31 >> 3) & 1)) * 4);       // swizzled 16-byte chunk of this lane's 8 k values
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[p][i] = *reinterpret_cast<const bf16x8*>(as + p * A_PLANE + (wm * WTM + i * 32 + l31) * 8 + cw);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[p][j] = *reinterpret_cast<const bf16x8*>(bs + p * B_PLANE + (wn * WTN + j * 32 + l31) * 8 + cw);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);   // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
                acc[i][j] = c;
                if (i == 0 && j == 0) {
                    if (stage) {
#pragma unroll
                        for (int q = 0; q < NLA; ++q) store_a(buf ^ 1, q);
                        store_b(buf ^ 1);
                    }
                    if (fetch) { load_a(kt_next2); load_b(kt_next2); }
                }
            }
    };

    if (nkt > 0) {
        load_a(0);
        load_b(0);
#pragma unroll
        for (int q = 0; q < NLA; ++q) store_a(0, q);
        store_b(0);
        if (nkt > 1) { load_a(1); load_b(1); }
        __syncthreads();
        int buf = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            step(buf, kt + 2, kt + 1 < nkt, kt + 2 < nkt);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---------------- epilogue.  The auxiliary operands (residual; the forward activation whose relu / dropout mask gates a
    // data gradient) are read through buffer descriptors: rows past M fall outside the descriptor and columns past N get
    // an out-of-range offset, so no load sits behind a branch -- a predicated global load compiles to "branch, load,
    // s_waitcnt vmcnt(0)" and serialises every one of the 16 loads of a tile.
    float* C = g.C;
    const bool do_drop = g.drop_thr != 0u;
    const bool has_gate = g.relu_out != nullptr, has_res = g.residual != nullptr;
    const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_gate ? g.relu_out : g.A), 0, has_gate ? (uint32_t)((long)g.M * g.ldc * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_res ? g.residual : g.A), 0, has_res ? (uint32_t)((long)g.M * g.ldr * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcBias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.bias != nullptr ? g.bias : g.A), 0, g.bias != nullptr ? (uint32_t)g.N * 4u : 0u, 0x00020000);
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)      // columns past N (and a null bias) fall outside the descriptor: 0
        bias_v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcBias, (n0 + wn * WTN + j * 32 + l31) * 4, 0, 0));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WTN + j * 32 + l31;
            const bool col_ok = col < g.N;
            const float bv = bias_v[j];
            const int row0 = m0 + wm * WTM + i * 32;
            float res[16], gsrc[16];
            if (has_res) {                 // block-uniform: one scalar branch around the whole group of loads
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = row0 + acc_row(r, half);
                    res[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsrcR, (int)(col_ok ? (uint32_t)((row * g.ldr + col) * 4) : OOB), 0, 0));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) res[r] = 0.f;
            }
            if (has_gate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = row0 + acc_row(r, half);
                    gsrc[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsrcG, (int)(col_ok ? (uint32_t)((row * g.ldc + col) * 4) : OOB), 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + acc_row(r, half);
                float v = acc[i][j][r] + bv;
                if (g.act == 1) v = fmaxf(v, 0.f);
                if (do_drop) {
                    uint64_t idx = (uint64_t)row * (uint64_t)g.N + (uint64_t)col;
                    v = keep_elem(seed_eff, idx, g.drop_thr) ? v * g.drop_scale : 0.f;
                }
                if (has_gate) v = gsrc[r] > 0.f ? v * g.relu_scale : 0.f;
                v += res[r];
                if (row < g.M && col_ok) C[(long)row * g.ldc + col] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_split(const GemmArgs& g, hipStream_t stream) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), 1);
    if (g.T > 0)
        hipLaunchKernelGGL((gemm_bf16x6_kernel<BM, BN, WM, WN, true>), grid, dim3(256), 0, stream, g);
    else
        hipLaunchKernelGGL((gemm_bf16x6_kernel<BM, BN, WM, WN, false>), grid, dim3(256), 0, stream, g);
    TTTS_LAUNCH_CHECK("gemm_bf16x6_kernel");
    return TTTS_OK;
}

static int dispatch_split(const GemmArgs& g, hipStream_t stream) {
    int tile = (g.N <= 96 && (long)cdiv(g.M, 128) >= 384) ? TILE_128x96 : choose_tile(g.M, g.N, 1, true);
    switch (tile) {
        case TILE_64: return launch_split<64, 64, 2, 2>(g, stream);
        case TILE_64x128: return launch_split<64, 128, 2, 2>(g, stream);
        case TILE_128x96: return launch_split<128, 96, 4, 1>(g, stream);
        default: return launch_split<128, 128, 2, 2>(g, stream);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients in the split-precision form: C[n][k] (per tap, per row split) = sum_m dy[m][n] * x[m + shift][k].
// Both operands are activations whose REDUCTION index (the row m) is the slow one in memory, so the MFMA fragments
// (8 consecutive reduction indices per lane) are columns of the stored matrices.  The loader turns them: a thread
// fetches 8 consecutive rows of ONE column with 8 dword loads (a wave covers 64 adjacent columns: 256-byte runs),
// splits the 8 values into hi / mid / lo and writes three 16-byte pieces -- exactly the [row = column][16 k] LDS
// image of the kernel above, so fragments, MFMA order and bank behaviour are shared with it.
// Same GemmArgs meaning as gemm_f32_kernel<.., false, false>: A = dy (K x M), B = x (K x N), K = rows, z = split * ztaps
// + tap, colsum = per-split column sums of dy (bias gradient).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, TTTS_GEMM_MINWAVES) void wgrad_bf16x6_kernel(GemmArgs g) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && BK == 16 && BM <= 128 && BN <= 128, "4 waves, 16 rows per step, <= 128 columns per operand");
    constexpr int A_PLANE = BM * 8, B_PLANE = BN * 8;          // dwords per plane (32-byte rows)
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);

    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const int ztap = (g.ztaps > 1) ? (z % g.ztaps) : 0;
    const int zsplit = (g.ztaps > 1) ? (z / g.ztaps) : z;
    const int nkt = (g.K + BK - 1) / BK;
    const int kt_begin = zsplit * g.kt_per_split;
    int kt_end = kt_begin + g.kt_per_split;
    if (kt_end > nkt) kt_end = nkt;
    const int shift = g.shift0 + ztap * g.shift_step;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // loader roles: thread -> (column, which 8 of the 16 rows).  2*BM (2*BN) threads take part.
    //               When both operands fit side by side (64x64 tile) A and B slots go to different waves.
    // No per-load validity arithmetic: rows past the end of the operand fall outside the buffer descriptor (hardware
    // returns 0), columns past the matrix edge only feed accumulator rows / columns that are never stored, and the
    // utterance clipping of shifted rows is applied to the loaded VALUES from one 8-bit mask per k-step.
    constexpr int B_T0 = (2 * BM + 2 * BN <= 256) ? 2 * BM : 0;
    const int tb = tid - B_T0;
    const bool a_slot = tid < 2 * BM, b_slot = tb >= 0 && tb < 2 * BN;
    const int a_col = tid % BM, a_rh = (tid / BM) & 1;
    const int b_col = (tb & 0xffff) % BN, b_rh = ((tb & 0xffff) / BN) & 1;
    const uint32_t a_row = (uint32_t)(g.lda * 4), b_row = (uint32_t)(g.ldb * 4);
    // byte offsets of this thread's first row at k-step kt_begin; advanced by 16 rows per step (wrap-around of the
    // shifted B offset below zero lands beyond the descriptor, i.e. reads 0)
    uint32_t a_cur = (uint32_t)(((long)kt_begin * BK + a_rh * 8) * g.lda + m0 + a_col) * 4u;
    uint32_t b_cur = (uint32_t)((((long)kt_begin * BK + b_rh * 8 + shift) * g.ldb + n0 + b_col) * 4);
    // position of this thread's first B row inside its utterance (clipping of shifted rows), kept incrementally
    int b_t = (g.T > 0) ? (int)(((long)kt_begin * BK + b_rh * 8) % g.T) : 0;
    const bool clip = g.T > 0 && shift != 0;

    float av[8], bv[8];
    float csum = 0.f;   // bias gradient: running sum of this thread's dy values

    auto load_tiles = [&](int) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            av[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcA, (int)(a_cur + j * a_row), 0, 0));
#pragma unroll
        for (int j = 0; j < 8; ++j)
            bv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcB, (int)(b_cur + j * b_row), 0, 0));
        a_cur += BK * a_row;
        b_cur += BK * b_row;
    };
    // bit j set = row j of this thread's 8 is clipped: its position t_j = (b_t + j) mod T has t_j + shift outside [0, T)
    auto clip_mask = [&]() -> uint32_t {
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int t = b_t + j;
            if (g.T >= BK) { if (t >= g.T) t -= g.T; } else t %= g.T;
            m |= ((unsigned)(t + shift) >= (unsigned)g.T) ? (1u << j) : 0u;
        }
        b_t += BK;
        if (g.T >= BK) { if (b_t >= g.T) b_t -= g.T; } else b_t %= g.T;
        return m;
    };

    auto store_tiles = [&](int buf) {
        uint32_t* as = lds + buf * STAGE;
        uint32_t* bs = as + 3 * A_PLANE;
        if (a_slot) {
            u32x4 hi, mid, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t h, m, l;
                split3_pair(f32x2{av[2 * q], av[2 * q + 1]}, h, m, l);
                hi[q] = h; mid[q] = m; lo[q] = l;
            }
            csum += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
            const int d = a_col * 8 + ((a_rh ^ ((a_col >> 3) & 1)) * 4);
            *reinterpret_cast<u32x4*>(as + d) = hi;
            *reinterpret_cast<u32x4*>(as + A_PLANE + d) = mid;
            *reinterpret_cast<u32x4*>(as + 2 * A_PLANE + d) = lo;
        }
        if (b_slot) {
            if (clip) {                      // block-uniform: only shifted taps pay for the clipping
                const uint32_t cm = clip_mask();
                if (cm != 0u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bv[j] = ((cm >> j) & 1u) ? 0.f : bv[j];
                }
            }
            u32x4 hi, mid, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t h, m, l;
                split3_pair(f32x2{bv[2 * q], bv[2 * q + 1]}, h, m, l);
                hi[q] = h; mid[q] = m; lo[q] = l;
            }
            const int d = b_col * 8 + ((b_rh ^ ((b_col >> 3) & 1)) * 4);
            *reinterpret_cast<u32x4*>(bs + d) = hi;
            *reinterpret_cast<u32x4*>(bs + B_PLANE + d) = mid;
            *reinterpret_cast<u32x4*>(bs + 2 * B_PLANE + d) = lo;
        }
    };

    auto compute = [&](int buf) {
        const uint32_t* as = lds + buf * STAGE;
        const uint32_t* bs = as + 3 * A_PLANE;
        const int cw = ((half ^ ((l31 >> 3) & 1)) * 4);
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[p][i] = *reinterpret_cast<const bf16x8*>(as + p * A_PLANE + (wm * WTM + i * 32 + l31) * 8 + cw);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[p][j] = *reinterpret_cast<const bf16x8*>(bs + p * B_PLANE + (wn * WTN + j * 32 + l31) * 8 + cw);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };

    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
        __syncthreads();
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = (kt + 1) < kt_end;
            if (more) load_tiles(kt + 1);
            compute(buf);
            if (more) store_tiles(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    if (g.colsum != nullptr && blockIdx.x == 0 && ztap == 0) {
        float* fs = reinterpret_cast<float*>(lds);
        __syncthreads();
        if (a_slot) fs[a_rh * BM + a_col] = csum;
        __syncthreads();
        if (tid < BM && m0 + tid < g.M) g.colsum[(long)zsplit * g.M + m0 + tid] = fs[tid] + fs[BM + tid];
    }

    float* C = g.C + (long)z * g.c_zstride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WTN + j * 32 + l31;
            const int row0 = m0 + wm * WTM + i * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + acc_row(r, half);
                if (row < g.M && col < g.N) C[(long)row * g.ldc + col] = acc[i][j][r];
            }
        }
}

template <int BM, int BN, int WM, int WN>
static int launch_wgrad_split(const GemmArgs& g, int zdim, hipStream_t stream) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), zdim);
    hipLaunchKernelGGL((wgrad_bf16x6_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, stream, g);
    TTTS_LAUNCH_CHECK("wgrad_bf16x6_kernel");
    return TTTS_OK;
}

static int dispatch_wgrad_split(const GemmArgs& g, int zdim, int tile, hipStream_t stream) {
    switch (tile) {
        case TILE_64: return launch_wgrad_split<64, 64, 2, 2>(g, zdim, stream);
        case TILE_128x96: return launch_wgrad_split<128, 96, 4, 1>(g, zdim, stream);
        case TILE_96x128: return launch_wgrad_split<96, 128, 1, 4>(g, zdim, stream);
        default: return launch_wgrad_split<128, 128, 2, 2>(g, zdim, stream);
    }
}

static GemmArgs base_args() {
    GemmArgs g;
    g.A = g.B = nullptr; g.C = nullptr;
    g.M = g.N = g.K = 0; g.lda = g.ldb = g.ldc = 0;
    g.T = 0; g.cin = 1; g.shift0 = 0; g.shift_step = 0; g.ztaps = 1;
    g.kt_per_split = 1 << 30; g.c_zstride = 0;
    g.bias = nullptr; g.act = 0; g.drop_scale = 1.f; g.drop_thr = 0; g.seed = 0; g.step_seed = nullptr;
    g.residual = nullptr; g.ldr = 0; g.colsum = nullptr; g.a_bytes = 0; g.b_bytes = 0;
    g.relu_out = nullptr; g.relu_scale = 1.f;
    g.a_amax = nullptr; g.a_amax_n = 0; g.b_amax = nullptr; g.b_amax_n = 0; g.c_amax = nullptr; g.bn_ws = nullptr;
    g.a_row_inv = nullptr; g.c_row_inv = nullptr; g.c_amax_sec = 0;
    return g;
}

static int aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// how the row (reduction) dimension of a weight gradient is split.  The fp32-MFMA / bf16x6 kernels aim at about 768
// workgroups (3 per CU).  The fp16x3 kernel holds two workgroups per CU, so 768 is one full round plus a half-empty one;
// it gets ONE round -- 512 workgroups for outputs of more than 12 tiles over long row ranges, else 256: every split costs
// a partial tile written and read again, which for the small outputs weighs more than a second workgroup per CU brings
// (measured on the step's shapes, tools/wgrad_bench.py: 256x256 118 -> 134, 768x256 157 -> 188, 256x1024 189 -> 214 TF).
struct WgradPlan { int tile; int nsplit; int kt_per_split; };
static WgradPlan plan_wgrad(int64_t M, int N, int K, int taps, bool x6 = false, int bk = BK) {
    WgradPlan p;
    long tiles = (long)cdiv(N, 128) * cdiv(K, 128) * taps;
    p.tile = (K <= 96) ? TILE_128x96 : TILE_128;
    long nkt = (M + bk - 1) / bk;
    if (x6) {
        // the split-precision kernel wants the large tile (its per-thread staging work is fixed per k-step); small
        // outputs get more row splits instead, capped so that the partial sums stay a few tens of MB
        if (N <= 64 && K <= 64) { p.tile = TILE_64; tiles = (long)cdiv(N, 64) * cdiv(K, 64) * taps; }
        else if (N <= 96 && K > 96) p.tile = TILE_96x128;          // 80-wide mel outputs
    } else if (tiles < 16) {
        p.tile = TILE_64; tiles = (long)cdiv(N, 64) * cdiv(K, 64) * taps;
    }
    constexpr long forced = 0;
    constexpr int big = 1;
    long target = 768;
    if (bk == HBK) {
        // The kernel's speed is set by the operand bytes its workgroups request (tiles x rows x (BM + BN) x 4 B at about
        // 7.7 TB/s over the chip: 1024x256 with 128-wide tiles asks for 913 MB = 119 us, measured 119), so outputs of three
        // or more 256 x 256 tiles take the 8-wave 256-wide tile (half the bytes per product) on one workgroup per CU.  Smaller
        // outputs stay on the 128-wide tile: at 256 row splits their partial sums (256 KB each, written and read back)
        // would cost what the operands save.
        const long tiles256 = (long)cdiv(N, 256) * cdiv(K, 256) * taps;
        if (big && p.tile == TILE_128 && N >= 256 && K >= 256 && tiles256 >= 3 && nkt >= 800) {
            p.tile = H3_TILE_256;
            tiles = tiles256;
            target = 256;
        } else {
            target = (tiles > 12 && nkt >= 800) ? 512 : 256;
        }
    }
    if (forced > 0) target = forced;
    long want = target / tiles;
    if (want < 1) want = 1;
    if (x6 && want > 128) want = 128;
    if (want > nkt) want = nkt;
    long per = (nkt + want - 1) / want;
    if (per < 8 && nkt >= 8) per = 8;
    p.kt_per_split = (int)per;
    p.nsplit = (int)((nkt + per - 1) / per);
    return p;
}

// the split-precision weight-gradient kernel pays off when its 128-wide (or 96-wide, for the 80-channel mel side) tiles
// are reasonably full
static bool wgrad_use_x6(int N, int K) { return (N >= 128 || N == 80 || N == 96) && (K >= 128 || K == 80 || K == 96) && (N >= 128 || K >= 128); }

}  // namespace ttts

using namespace ttts;

extern "C" {

int ttts_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int64_t M,
                    int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int row_shift, int T, void* stream) {
    TTTS_REQUIRE(x && w && y, "linear_fwd: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_fwd: bad dims M=%lld N=%d K=%d", (long long)M, N, K);
    TTTS_REQUIRE(K % BK == 0, "linear_fwd: K=%d must be a multiple of %d", K, BK);
    TTTS_REQUIRE(aligned16(x) && aligned16(w), "linear_fwd: x/w must be 16-byte aligned");
    TTTS_REQUIRE(act == 0 || act == 1, "linear_fwd: act must be 0 (none) or 1 (relu)");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear_fwd: dropout p=%f out of [0,1)", drop_p);
    TTTS_REQUIRE(row_shift == 0 || (T > 0 && M % T == 0), "linear_fwd: row_shift needs T>0 and M %% T == 0");
    GemmArgs g = base_args();
    g.A = x; g.B = w; g.C = y; g.M = (int)M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    TTTS_REQUIRE((uint64_t)M * K * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_fwd: operand larger than 4 GiB");
    g.a_bytes = (uint32_t)((uint64_t)M * K * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 4);
    g.cin = K; g.shift0 = row_shift; g.T = (row_shift != 0) ? T : 0;
    g.bias = bias; g.act = act;
    if (drop_p > 0.f) { g.drop_thr = drop_threshold(drop_p); g.drop_scale = 1.f / (1.f - drop_p); g.seed = seed; g.step_seed = step_seed; }
    g.residual = residual; g.ldr = N;
    return dispatch_gemm<true, true>(g, 1, TILE_AUTO, (hipStream_t)stream);
}

int ttts_linear_bwd_data(const float* dy, const float* w, const float* residual, float* dx, int64_t M, int N, int K,
                         const float* relu_out, float relu_scale, void* stream) {
    // dx[M,K] = dy[M,N] . w[N,K] (+ residual)   (reduction over N; w consumed in its natural [N][K] layout)
    TTTS_REQUIRE(dy && w && dx, "linear_bwd_data: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_bwd_data: bad dims");
    TTTS_REQUIRE(N % BK == 0 && K % 4 == 0, "linear_bwd_data: N=%d must be a multiple of 16 and K=%d of 4", N, K);
    TTTS_REQUIRE(aligned16(dy) && aligned16(w), "linear_bwd_data: dy/w must be 16-byte aligned");
    GemmArgs g = base_args();
    g.A = dy; g.B = w; g.C = dx; g.M = (int)M; g.N = K; g.K = N;
    g.lda = N; g.ldb = K; g.ldc = K; g.cin = N;
    TTTS_REQUIRE((uint64_t)M * N * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_bwd_data: operand larger than 4 GiB");
    g.a_bytes = (uint32_t)((uint64_t)M * N * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 4);
    g.residual = residual; g.ldr = K;
    g.relu_out = relu_out; g.relu_scale = relu_scale;
    return dispatch_gemm<true, false>(g, 1, TILE_AUTO, (hipStream_t)stream);
}

size_t ttts_wgrad_workspace_bytes(int64_t M, int N, int K, int taps) {
    // covers both kernel forms (their row-split counts differ)
    WgradPlan p = plan_wgrad(M, N, K, taps, false), q = plan_wgrad(M, N, K, taps, true), r = plan_wgrad(M, N, K, taps, true, HBK);
    size_t ns = (size_t)(p.nsplit > q.nsplit ? p.nsplit : q.nsplit);
    if ((size_t)r.nsplit > ns) ns = (size_t)r.nsplit;
    return (ns * taps * N * K + ns * N) * sizeof(float);
}

static int wgrad_common(const float* dy, const float* x, float* ws, float* colsum_ws, int64_t M, int N, int K, int taps,
                        int T, int shift0, int shift_step, WgradPlan* plan_out, bool x6, hipStream_t stream,
                        const float* dy_amax = nullptr, const float* x_amax = nullptr) {
    // dy_amax != NULL selects the fp16x3 kernel (32-row k-steps, dynamic pre-scales of dy and x); x6 must be true then
    WgradPlan p = plan_wgrad(M, N, K, taps, x6, dy_amax ? HBK : BK);
    GemmArgs g = base_args();
    // C[N][K] (per tap) = dy^T[N][M] . xshift[M][K]
    g.A = dy; g.B = x; g.C = ws; g.M = N; g.N = K; g.K = (int)M;
    g.lda = N; g.ldb = K; g.ldc = K;
    if (!((uint64_t)M * N * 4 < (1ull << 32) && (uint64_t)M * K * 4 < (1ull << 32))) {
        set_error("weight gradient: operand larger than 4 GiB");
        return TTTS_ERR_INVALID;
    }
    g.a_bytes = (uint32_t)((uint64_t)M * N * 4); g.b_bytes = (uint32_t)((uint64_t)M * K * 4);
    g.T = T; g.shift0 = shift0; g.shift_step = shift_step; g.ztaps = taps;
    g.kt_per_split = p.kt_per_split; g.c_zstride = (long)N * K;
    g.colsum = colsum_ws;
    *plan_out = p;
    if (dy_amax) {
        g.a_amax = dy_amax; g.a_amax_n = H3_AMAX_PARTIALS;
        g.b_amax = x_amax; g.b_amax_n = H3_AMAX_PARTIALS;
        return dispatch_wgrad_h3(g, p.nsplit * taps, p.tile, stream);
    }
    if (x6) return dispatch_wgrad_split(g, p.nsplit * taps, p.tile, stream);
    return dispatch_gemm<false, false>(g, p.nsplit * taps, p.tile, stream);
}

static int linear_bwd_weight_impl(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes,
                                  int64_t M, int N, int K, int row_shift, int T, int accumulate, bool x6,
                                  ttts_reduce_queue* queue, void* stream_, const float* dy_amax = nullptr,
                                  const float* x_amax = nullptr) {
    // dw[N,K] (+)= dy[M,N]^T . x[M,K] ; dbias[N] (+)= column sums of dy
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(dy && x && dw && ws, "linear_bwd_weight: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_bwd_weight: bad dims");
    TTTS_REQUIRE(N % 4 == 0 && K % 4 == 0, "linear_bwd_weight: N=%d and K=%d must be multiples of 4", N, K);
    TTTS_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(ws), "linear_bwd_weight: pointers must be 16-byte aligned");
    TTTS_REQUIRE(ws_bytes >= ttts_wgrad_workspace_bytes(M, N, K, 1), "linear_bwd_weight: workspace too small");
    TTTS_REQUIRE(row_shift == 0 || (T > 0 && M % T == 0), "linear_bwd_weight: row_shift needs T>0 and M %% T == 0");
    WgradPlan p;
    long n = (long)N * K;
    x6 = x6 && wgrad_use_x6(N, K);
    if (!x6) dy_amax = nullptr;
    float* colsum_ws = dbias ? ws + (size_t)plan_wgrad(M, N, K, 1, x6, dy_amax ? HBK : BK).nsplit * n : nullptr;
    int rc = wgrad_common(dy, x, ws, colsum_ws, M, N, K, 1, row_shift != 0 ? T : 0, row_shift, 0, &p, x6, stream, dy_amax, x_amax);
    if (rc) return rc;
    return launch_reduce_rows_pair(ws, n, p.nsplit, n, dw, colsum_ws, N, N, dbias, accumulate != 0, stream, queue);
}

int ttts_linear_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes,
                           int64_t M, int N, int K, int row_shift, int T, int accumulate, ttts_reduce_queue* queue, void* stream) {
    return linear_bwd_weight_impl(dy, x, dw, dbias, ws, ws_bytes, M, N, K, row_shift, T, accumulate, false, queue, stream);
}
int ttts_linear_bwd_weight_x6(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes,
                              int64_t M, int N, int K, int row_shift, int T, int accumulate, ttts_reduce_queue* queue, void* stream) {
    return linear_bwd_weight_impl(dy, x, dw, dbias, ws, ws_bytes, M, N, K, row_shift, T, accumulate, true, queue, stream);
}
int ttts_linear_bwd_weight_h3(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes,
                              int64_t M, int N, int K, int row_shift, int T, int accumulate, const float* dy_amax,
                              const float* x_amax, ttts_reduce_queue* queue, void* stream) {
    TTTS_REQUIRE(dy_amax && x_amax, "linear_bwd_weight_h3: dy_amax and x_amax (partial maxima of dy and x) are required");
    return linear_bwd_weight_impl(dy, x, dw, dbias, ws, ws_bytes, M, N, K, row_shift, T, accumulate, true, queue, stream, dy_amax,
                                  x_amax);
}

int ttts_linear_bwd_weight_h3_parts(const float* dy, const float* x, float* const* dw_parts, float* const* dbias_parts, int nparts,
                                    float* ws, size_t ws_bytes, int64_t M, int N, int K, int accumulate, const float* dy_amax,
                                    const float* x_amax, ttts_reduce_queue* queue, void* stream_) {
    // ONE weight-gradient GEMM dy^T x (N x K) whose N rows belong to `nparts` different weights (equal row blocks): block i is
    // reduced into dw_parts[i] ((N / nparts) x K, contiguous) and its column sums into dbias_parts[i] -- the fused K/V
    // projection of all decoder layers' cross-attention (model/layers.py:54-74 runs it once per layer on the same memory)
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(dy && x && dw_parts && ws && dy_amax && x_amax, "linear_bwd_weight_h3_parts: null pointer");
    TTTS_REQUIRE(nparts > 0 && nparts <= 64 && N % nparts == 0 && (N / nparts) % 4 == 0, "linear_bwd_weight_h3_parts: N=%d does not split in %d blocks of whole float4s", N, nparts);
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31) && K % 4 == 0, "linear_bwd_weight_h3_parts: bad dims");
    TTTS_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(ws), "linear_bwd_weight_h3_parts: pointers must be 16-byte aligned");
    TTTS_REQUIRE(ws_bytes >= ttts_wgrad_workspace_bytes(M, N, K, 1), "linear_bwd_weight_h3_parts: workspace too small");
    for (int i = 0; i < nparts; ++i) TTTS_REQUIRE(dw_parts[i] != nullptr, "linear_bwd_weight_h3_parts: null destination");
    WgradPlan p;
    const long n = (long)N * K;
    const bool x6 = wgrad_use_x6(N, K);
    const bool want_b = dbias_parts != nullptr && dbias_parts[0] != nullptr;
    float* colsum_ws = want_b ? ws + (size_t)plan_wgrad(M, N, K, 1, x6, x6 ? HBK : BK).nsplit * n : nullptr;
    int rc = wgrad_common(dy, x, ws, colsum_ws, M, N, K, 1, 0, 0, 0, &p, x6, stream, x6 ? dy_amax : nullptr, x_amax);
    if (rc) return rc;
    const int Np = N / nparts;
    const long np = (long)Np * K;
    for (int i = 0; i < nparts; ++i) {
        rc = launch_reduce_rows_pair(ws + (size_t)i * np, n, p.nsplit, np, dw_parts[i], want_b ? colsum_ws + (size_t)i * Np : nullptr, N, Np,
                                     want_b ? dbias_parts[i] : nullptr, accumulate != 0, stream, queue);
        if (rc) return rc;
    }
    return TTTS_OK;
}

int ttts_wgrad_group_ok(int64_t M, int N, int K, int taps) {
    // can dw[N,K(,tap)] = dy[M,N]^T x[M(+tap shift),K] be a member of a grouped launch, and of which CLASS (members of one launch
    // share it)?  1: the fp16x3 form on the 4-wave 128 x 128 tile (small outputs); 2: whole 256 x 256 tiles on the 8-wave LDS-DMA
    // kernel (long row ranges); 3 / 4: the 96-wide tiles of the 80-channel mel side; 0: none (the fp32-MFMA shapes, ragged
    // 256-tiles keep launches of their own).  taps = 1: a linear weight; taps > 1: a convolution's (cout = N, cin = K).
    if (!(M > 0 && N > 0 && K > 0 && taps >= 1 && taps <= 64 && N % 4 == 0 && K % 4 == 0 && wgrad_use_x6(N, K))) return 0;
    const int tile = plan_wgrad(M, N, K, taps, true, HBK).tile;
    if (tile == TILE_128) return 1;
    if (tile == H3_TILE_256 && N % 256 == 0 && K % 256 == 0) return 2;
    if (tile == TILE_128x96) return 3;            // 80 input channels: the decoder pre-net's first linear, the post-net's first convolution
    if (tile == TILE_96x128) return 4;            // 80 output channels: the mel head, the post-net's last convolution
    return 0;
}

int ttts_wgrad_group(int n, const float* const* dy, const float* const* x, float* const* dw, float* const* dbias,
                     float* const* ws, const size_t* ws_bytes, const int64_t* M, const int* N, const int* K, const int* taps,
                     const int* T, const int* row_shift, int accumulate, const float* const* dy_amax, const float* const* x_amax,
                     ttts_reduce_queue* queue, void* stream_) {
    // n <= 4 independent weight gradients as ONE grid (all members of one class, ttts_wgrad_group_ok).  Member i with taps_i = 1:
    // dw_i[N_i,K_i] (+)= dy_i^T x_i (a linear weight; row_shift_i != 0: x_i read row_shift_i rows further inside utterances of T_i
    // rows, as ttts_linear_bwd_weight's row_shift -- the decoder pre-net's go-frame shift); taps_i > 1: dw_i[N_i,K_i,tap] (+)= sum over (b, t) of dy_i[b,t,:] x_i[b,
    // t + tap - (taps-1)/2, :] with utterances of T_i rows (a same-padded convolution; M_i = B T_i).  dbias_i (+)= column sums of
    // dy_i.  The row splits are planned for the group (same chip-filling target as a single launch, shared by the members), so
    // every member writes 1/n of the partial sums a launch of its own would and its workgroups walk n times the rows.  Each
    // member's partial sums go to its own workspace and are reduced (queued) as usual.
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(n >= 1 && n <= 4 && dy && x && dw && ws && ws_bytes && M && N && K && taps && T && row_shift && dy_amax && x_amax, "wgrad_group: bad arguments");
    GemmArgs gs[4];
    int zd[4], ns[4];
    float* colsum[4];
    long tiles_total = 0, nkt_max = 0;
    const int cls = ttts_wgrad_group_ok(M[0], N[0], K[0], taps[0]);
    const int edge_n = cls == 2 ? 256 : cls == 4 ? 96 : 128, edge_k = cls == 2 ? 256 : cls == 3 ? 96 : 128;      // tile: output rows x columns
    const int tile = cls == 3 ? TILE_128x96 : cls == 4 ? TILE_96x128 : TILE_128;
    for (int i = 0; i < n; ++i) {
        TTTS_REQUIRE(dy[i] && x[i] && dw[i] && ws[i] && dy_amax[i] && x_amax[i], "wgrad_group: null pointer (member %d)", i);
        TTTS_REQUIRE(cls != 0 && ttts_wgrad_group_ok(M[i], N[i], K[i], taps[i]) == cls && M[i] < (1LL << 31),
                     "wgrad_group: member %d (M=%lld N=%d K=%d taps=%d) is not of the group's class %d", i, (long long)M[i], N[i], K[i], taps[i], cls);
        TTTS_REQUIRE((taps[i] == 1 && row_shift[i] == 0) || (T[i] > 0 && M[i] % T[i] == 0 && (cls != 2 || T[i] >= 16)),
                     "wgrad_group: member %d: a convolution / a shifted linear needs its utterance length", i);
        TTTS_REQUIRE(taps[i] == 1 || row_shift[i] == 0, "wgrad_group: member %d: a convolution takes no row shift of its own", i);
        TTTS_REQUIRE(aligned16(dy[i]) && aligned16(x[i]) && aligned16(ws[i]), "wgrad_group: pointers must be 16-byte aligned");
        TTTS_REQUIRE((uint64_t)M[i] * N[i] * 4 < (1ull << 32) && (uint64_t)M[i] * K[i] * 4 < (1ull << 32), "wgrad_group: operand larger than 4 GiB");
        tiles_total += (long)cdiv(N[i], edge_n) * cdiv(K[i], edge_k) * taps[i];
        const long nkt = (M[i] + HBK - 1) / HBK;
        nkt_max = nkt > nkt_max ? nkt : nkt_max;
    }
    // plan_wgrad's rule for the class, applied to the group as a whole: the 8-wave tile holds one workgroup per CU, the 4-wave
    // tile two (a second round only for many tiles over long row ranges)
    // (the 96-wide tiles' workgroups are long -- 10 + 2 tiles of the mel side over 55 680 rows were 140 us on 252 workgroups -- and
    // their partial tiles small: both workgroup slots of every CU from 8 tiles on)
    const long many = (cls == 3 || cls == 4) ? 8 : 13;
    // (row ranges from 200 k-tiles on -- 6 400 rows: at 13 920 rows a group on ONE workgroup slot per CU ran its 109 k-tiles per
    // workgroup at a fifth of the MFMA rate, nothing covering its operand latencies: batch 16 -0.08 ms, `profiles/r06_ab_wgrad_target.txt`)
#ifndef TTTS_WG_NKT512
#define TTTS_WG_NKT512 200
#endif
    const long target = cls == 2 ? 256 : ((tiles_total >= many && nkt_max >= TTTS_WG_NKT512) ? 512 : 256);
    long want = target / tiles_total;
    if (want < 1) want = 1;
    for (int i = 0; i < n; ++i) {
        const long nkt = (M[i] + HBK - 1) / HBK;
        long w = want > nkt ? nkt : want;
        long per = (nkt + w - 1) / w;
        if (per < 8 && nkt >= 8) per = 8;
        const int nsplit = (int)((nkt + per - 1) / per);
        const long nk = (long)N[i] * K[i];
        TTTS_REQUIRE(ws_bytes[i] >= ((size_t)nsplit * taps[i] * nk + (size_t)nsplit * N[i]) * sizeof(float), "wgrad_group: workspace %d too small", i);
        GemmArgs g = base_args();
        g.A = dy[i]; g.B = x[i]; g.C = ws[i]; g.M = N[i]; g.N = K[i]; g.K = (int)M[i];
        g.lda = N[i]; g.ldb = K[i]; g.ldc = K[i];
        g.a_bytes = (uint32_t)((uint64_t)M[i] * N[i] * 4); g.b_bytes = (uint32_t)((uint64_t)M[i] * K[i] * 4);
        const bool conv = taps[i] > 1;
        g.T = (conv || row_shift[i] != 0) ? T[i] : 0; g.shift0 = conv ? -((taps[i] - 1) / 2) : row_shift[i]; g.shift_step = conv ? 1 : 0;
        g.ztaps = taps[i];
        g.kt_per_split = (int)per; g.c_zstride = nk;
        colsum[i] = (dbias != nullptr && dbias[i] != nullptr) ? ws[i] + (size_t)nsplit * taps[i] * nk : nullptr;
        g.colsum = colsum[i];
        g.a_amax = dy_amax[i]; g.a_amax_n = H3_AMAX_PARTIALS;
        g.b_amax = x_amax[i]; g.b_amax_n = H3_AMAX_PARTIALS;
        gs[i] = g;
        ns[i] = nsplit;
        zd[i] = nsplit * taps[i];
    }
    int rc = cls == 2 ? launch_wgrad_dma_group(gs, zd, n, stream) : launch_wgrad_h3_group(gs, zd, n, tile, stream);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const long nk = (long)N[i] * K[i];
        if (taps[i] == 1) {
            rc = launch_reduce_rows_pair(ws[i], nk, ns[i], nk, dw[i], colsum[i], N[i], N[i], colsum[i] ? dbias[i] : nullptr, accumulate != 0,
                                         stream, queue);
        } else {
            rc = launch_conv_wgrad_reduce(ws[i], dw[i], N[i], K[i], taps[i], ns[i], accumulate != 0, stream, queue);
            if (rc == TTTS_OK && colsum[i])
                rc = launch_reduce_rows(colsum[i], N[i], ns[i], N[i], dbias[i], N[i], nullptr, accumulate != 0, stream, queue);
        }
        if (rc) return rc;
    }
    return TTTS_OK;
}

size_t ttts_conv1d_pack_bytes(int cout, int cin, int taps) { return (size_t)cout * cin * taps * sizeof(float); }

int ttts_conv1d_pack_weight(const float* w, float* w_fwd, float* w_bwd, int cout, int cin, int taps, void* stream) {
    TTTS_REQUIRE(w && (w_fwd || w_bwd), "conv1d_pack_weight: null pointer");
    long n = (long)cout * cin * taps;
    hipLaunchKernelGGL(conv_pack_weight_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, w_fwd, w_bwd,
                       cout, cin, taps);
    TTTS_LAUNCH_CHECK("conv_pack_weight_kernel");
    return TTTS_OK;
}

int ttts_conv1d_fwd(const float* x, const float* w_fwd, const float* bias, float* y, int B, int T, int cin, int cout,
                    int taps, void* stream) {
    // y[b,t,co] = bias[co] + sum_{tap,ci} x[b,t+tap-pad,ci] * w[co,ci,tap],  pad = (taps-1)/2, zero outside [0,T)
    TTTS_REQUIRE(x && w_fwd && y, "conv1d_fwd: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && cin > 0 && cout > 0 && taps > 0 && (taps & 1), "conv1d_fwd: bad dims");
    TTTS_REQUIRE(cin % BK == 0, "conv1d_fwd: cin=%d must be a multiple of %d", cin, BK);
    TTTS_REQUIRE((long)B * T < (1L << 31), "conv1d_fwd: B*T too large");
    TTTS_REQUIRE(aligned16(x) && aligned16(w_fwd), "conv1d_fwd: pointers must be 16-byte aligned");
    GemmArgs g = base_args();
    g.A = x; g.B = w_fwd; g.C = y; g.M = B * T; g.N = cout; g.K = taps * cin;
    g.lda = cin; g.ldb = (long)taps * cin; g.ldc = cout;
    TTTS_REQUIRE((uint64_t)B * T * cin * 4 < (1ull << 32), "conv1d_fwd: activation larger than 4 GiB");
    g.a_bytes = (uint32_t)((uint64_t)B * T * cin * 4); g.b_bytes = (uint32_t)((uint64_t)cout * taps * cin * 4);
    g.T = T; g.cin = cin; g.shift0 = -((taps - 1) / 2); g.shift_step = 1;
    g.bias = bias;
    return dispatch_gemm<true, true>(g, 1, TILE_AUTO, (hipStream_t)stream);
}

int ttts_conv1d_bwd_data(const float* dy, const float* w_bwd, float* dx, int B, int T, int cin, int cout, int taps,
                         void* stream) {
    // dx[b,t,ci] = sum_{tap,co} dy[b,t-tap+pad,co] * w[co,ci,tap]   (w_bwd = w packed as [ci][tap][co])
    TTTS_REQUIRE(dy && w_bwd && dx, "conv1d_bwd_data: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && cin > 0 && cout > 0 && taps > 0 && (taps & 1), "conv1d_bwd_data: bad dims");
    TTTS_REQUIRE(cout % BK == 0, "conv1d_bwd_data: cout=%d must be a multiple of %d", cout, BK);
    TTTS_REQUIRE(aligned16(dy) && aligned16(w_bwd), "conv1d_bwd_data: pointers must be 16-byte aligned");
    GemmArgs g = base_args();
    g.A = dy; g.B = w_bwd; g.C = dx; g.M = B * T; g.N = cin; g.K = taps * cout;
    g.lda = cout; g.ldb = (long)taps * cout; g.ldc = cin;
    TTTS_REQUIRE((uint64_t)B * T * cout * 4 < (1ull << 32), "conv1d_bwd_data: activation larger than 4 GiB");
    g.a_bytes = (uint32_t)((uint64_t)B * T * cout * 4); g.b_bytes = (uint32_t)((uint64_t)cin * taps * cout * 4);
    g.T = T; g.cin = cout; g.shift0 = (taps - 1) / 2; g.shift_step = -1;
    return dispatch_gemm<true, true>(g, 1, TILE_AUTO, (hipStream_t)stream);
}

static int conv1d_bwd_weight_impl(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes, int B,
                                  int T, int cin, int cout, int taps, int accumulate, bool x6, ttts_reduce_queue* queue,
                                  void* stream_, const float* dy_amax = nullptr, const float* x_amax = nullptr) {
    // dw[co,ci,tap] (+)= sum_{b,t} dy[b,t,co] * x[b,t+tap-pad,ci] ; dbias[co] (+)= sum dy
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(dy && x && dw && ws, "conv1d_bwd_weight: null pointer");
    TTTS_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "conv1d_bwd_weight: channel counts must be multiples of 4");
    TTTS_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(ws), "conv1d_bwd_weight: pointers must be 16-byte aligned");
    int64_t M = (int64_t)B * T;
    TTTS_REQUIRE(ws_bytes >= ttts_wgrad_workspace_bytes(M, cout, cin, taps), "conv1d_bwd_weight: workspace too small");
    WgradPlan p;
    long n = (long)cout * cin * taps;
    x6 = x6 && wgrad_use_x6(cout, cin);
    if (!x6) dy_amax = nullptr;
    float* colsum_ws = dbias ? ws + (size_t)plan_wgrad(M, cout, cin, taps, x6, dy_amax ? HBK : BK).nsplit * n : nullptr;
    int rc = wgrad_common(dy, x, ws, colsum_ws, M, cout, cin, taps, T, -((taps - 1) / 2), 1, &p, x6, stream, dy_amax, x_amax);
    if (rc) return rc;
    rc = launch_conv_wgrad_reduce(ws, dw, cout, cin, taps, p.nsplit, accumulate != 0, stream, queue);
    if (rc == TTTS_OK && dbias)
        rc = launch_reduce_rows(colsum_ws, cout, p.nsplit, cout, dbias, cout, nullptr, accumulate != 0, stream, queue);
    return rc;
}

int ttts_conv1d_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes, int B,
                           int T, int cin, int cout, int taps, int accumulate, ttts_reduce_queue* queue, void* stream) {
    return conv1d_bwd_weight_impl(dy, x, dw, dbias, ws, ws_bytes, B, T, cin, cout, taps, accumulate, false, queue, stream);
}
int ttts_conv1d_bwd_weight_x6(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes, int B,
                              int T, int cin, int cout, int taps, int accumulate, ttts_reduce_queue* queue, void* stream) {
    return conv1d_bwd_weight_impl(dy, x, dw, dbias, ws, ws_bytes, B, T, cin, cout, taps, accumulate, true, queue, stream);
}

int ttts_conv1d_bwd_weight_h3(const float* dy, const float* x, float* dw, float* dbias, float* ws, size_t ws_bytes, int B,
                              int T, int cin, int cout, int taps, int accumulate, const float* dy_amax, const float* x_amax,
                              ttts_reduce_queue* queue, void* stream) {
    TTTS_REQUIRE(dy_amax && x_amax, "conv1d_bwd_weight_h3: dy_amax and x_amax (partial maxima of dy and x) are required");
    return conv1d_bwd_weight_impl(dy, x, dw, dbias, ws, ws_bytes, B, T, cin, cout, taps, accumulate, true, queue, stream, dy_amax,
                                  x_amax);
}

int ttts_gemm_tile_choice(int64_t M, int N, int K, int x6) {
    // which tile the forward / data-gradient dispatch picks for an M x N output: 1 = 64x64, 2 = 128x128, 3 = 64x128,
    // 4 = 128x96 (profiling aid: lets a host-side probe attribute a launch to its kernel instantiation);
    // x6 = 0: fp32 MFMA kernel, 1: bf16x6, 2: fp16x3
    if (x6 == 2) return h3_tile_choice(M, N, K);
    if (N <= 96 && (long)cdiv(M, 128) >= 384) return TILE_128x96;
    return choose_tile(M, N, 1, x6 != 0);
}

size_t ttts_split_bytes(int64_t rows, int64_t cols) { return (size_t)3 * (size_t)rows * (size_t)cols * 2; }

size_t ttts_split_image_bytes(int64_t rows, int64_t cols, int mode, int channels_per_tap, int taps) {
    // bytes of the image ttts_weight_split writes in `mode`: three bf16 planes (modes 0-3), or two f16 planes of the
    // channel-padded image plus the 16-byte tail (modes 4-7)
    if (mode < 4) return ttts_split_bytes(rows, cols);
    const bool conv = h3_mode_base(mode) >= 2;
    return h3_plane_bytes(rows, h3_image_cols(cols, conv ? channels_per_tap : 0, conv ? taps : 0)) + 16;
}

int ttts_weight_split(const float* w, void* planes, int rows, int cols, int mode, int channels_per_tap, int taps,
                      void* stream) {
    // planes[3][rows][cols] bf16 (hi, mid, lo) of a weight re-laid as the K-contiguous B operand; modes in gemm.hip
    TTTS_REQUIRE(w && planes && rows > 0 && cols > 0, "weight_split: bad arguments");
    TTTS_REQUIRE(mode >= 0 && mode <= 11, "weight_split: mode must be 0..11");
    TTTS_REQUIRE(cols % (mode >= 4 ? 4 : BK) == 0, "weight_split: cols=%d must be a multiple of %d", cols, mode >= 4 ? 4 : BK);
    TTTS_REQUIRE((mode < 4 ? mode : h3_mode_base(mode)) < 2 || (channels_per_tap > 0 && taps > 0 && cols == channels_per_tap * taps),
                 "weight_split: conv modes need cols == channels_per_tap * taps");
    if (mode >= 4) {
        // (the fp16x3 image pads the channels of a tap -- all columns of a linear weight -- to a multiple of 32 with zeros:
        // size it with ttts_split_image_bytes)
        TTTS_REQUIRE(h3_mode_base(mode) < 2 || channels_per_tap % 4 == 0, "weight_split: fp16x3 conv image needs channels %% 4 == 0");
        launch_weight_split_h3(w, planes, rows, cols, h3_mode_base(mode), channels_per_tap, taps, (hipStream_t)stream, h3_mode_k16(mode));
        TTTS_LAUNCH_CHECK("weight_split_h3_kernel");
        return TTTS_OK;
    }
    long n = (long)rows * cols;
    hipLaunchKernelGGL(weight_split_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)planes, rows, cols, mode, channels_per_tap, taps);
    TTTS_LAUNCH_CHECK("weight_split_kernel");
    return TTTS_OK;
}

int64_t ttts_weight_split_units(int64_t rows, int64_t cols, int mode, int channels_per_tap) {
    // workgroups an entry of ttts_weight_split_batched takes: 256 elements each (bf16x6 images), a 32-row x 32-channel
    // tile over all taps each (fp16x3 images)
    // (linear entries may carry the geometry of a stacked image in channels_per_tap / taps: no part of their unit count)
    return mode >= 4 ? h3_split_units(rows, cols, h3_mode_base(mode), h3_mode_base(mode) >= 2 ? channels_per_tap : 0) : (rows * cols + 255) / 256;
}

int ttts_weight_split_batched(const int64_t* descs, int n, int64_t total_blocks, void* stream) {
    // descs (device): n x 8 int64 {w, planes, rows, cols, mode, channels_per_tap, taps, first_block}; first_block are the
    // prefix sums of ttts_weight_split_units(); the caller guarantees the per-entry constraints of ttts_weight_split
    TTTS_REQUIRE(descs && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "weight_split_batched: bad arguments");
    hipLaunchKernelGGL(weight_tail_zero_batched_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long*>(descs), n);
    hipLaunchKernelGGL(weight_amax_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long*>(descs), n);
    hipLaunchKernelGGL(weight_split_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long*>(descs), n);
    TTTS_LAUNCH_CHECK("weight_split_batched_kernel");
    return TTTS_OK;
}

int ttts_linear_fwd_x6(const float* x, const void* w_planes, const float* bias, const float* residual, float* y,
                       int64_t M, int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int row_shift, int T, void* stream) {
    TTTS_REQUIRE(x && w_planes && y, "linear_fwd_x6: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_fwd_x6: bad dims");
    TTTS_REQUIRE(K % BK == 0, "linear_fwd_x6: K=%d must be a multiple of %d", K, BK);
    TTTS_REQUIRE(aligned16(x) && aligned16(w_planes), "linear_fwd_x6: x / planes must be 16-byte aligned");
    TTTS_REQUIRE(act == 0 || act == 1, "linear_fwd_x6: act must be 0 (none) or 1 (relu)");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear_fwd_x6: dropout p out of [0,1)");
    TTTS_REQUIRE(row_shift == 0 || (T > 0 && M % T == 0), "linear_fwd_x6: row_shift needs T>0 and M %% T == 0");
    TTTS_REQUIRE((uint64_t)M * K * 4 < (1ull << 32) && (uint64_t)N * K * 6 < (1ull << 32), "linear_fwd_x6: operand larger than 4 GiB");
    GemmArgs g = base_args();
    g.A = x; g.B = (const float*)w_planes; g.C = y; g.M = (int)M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.a_bytes = (uint32_t)((uint64_t)M * K * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 6);
    g.cin = K; g.shift0 = row_shift; g.T = (row_shift != 0) ? T : 0;
    g.bias = bias; g.act = act;
    if (drop_p > 0.f) { g.drop_thr = drop_threshold(drop_p); g.drop_scale = 1.f / (1.f - drop_p); g.seed = seed; g.step_seed = step_seed; }
    g.residual = residual; g.ldr = N;
    return dispatch_split(g, (hipStream_t)stream);
}

int ttts_linear_fwd_h3(const float* x, const void* w_planes, const float* bias, const float* residual, float* y,
                       int64_t M, int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int row_shift,
                       int T, const float* x_amax, float* y_amax_out, void* stream) {
    TTTS_REQUIRE(x && w_planes && y && x_amax, "linear_fwd_h3: null pointer (x_amax, the partial maxima of |x|, is required)");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_fwd_h3: bad dims");
    TTTS_REQUIRE(K % 4 == 0 && N % 4 == 0, "linear_fwd_h3: K=%d and N=%d must be multiples of 4", K, N);
    TTTS_REQUIRE(aligned16(x) && aligned16(w_planes), "linear_fwd_h3: x / planes must be 16-byte aligned");
    TTTS_REQUIRE(act == 0 || act == 1, "linear_fwd_h3: act must be 0 (none) or 1 (relu)");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear_fwd_h3: dropout p out of [0,1)");
    TTTS_REQUIRE(row_shift == 0 || (T > 0 && M % T == 0), "linear_fwd_h3: row_shift needs T>0 and M %% T == 0");
    TTTS_REQUIRE((uint64_t)M * K * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_fwd_h3: operand larger than 4 GiB");
    GemmArgs g = base_args();
    // (K is padded to the image's 32-deep k-tiles: the loader reads up to 28 floats past a row's end -- the next row, or
    // zeros past the buffer -- against zero weight columns)
    const int Kp = h3_pad32(K);
    g.A = x; g.B = (const float*)w_planes; g.C = y; g.M = (int)M; g.N = N; g.K = Kp;
    g.lda = K; g.ldb = Kp; g.ldc = N;
    g.a_bytes = (uint32_t)((uint64_t)M * K * 4); g.b_bytes = (uint32_t)((uint64_t)N * Kp * 4);
    g.cin = Kp; g.shift0 = row_shift; g.T = (row_shift != 0) ? T : 0;
    g.bias = bias; g.act = act;
    if (drop_p > 0.f) { g.drop_thr = drop_threshold(drop_p); g.drop_scale = 1.f / (1.f - drop_p); g.seed = seed; g.step_seed = step_seed; }
    g.residual = residual; g.ldr = N;
    g.a_amax = x_amax; g.a_amax_n = H3_AMAX_PARTIALS;
    g.b_amax = h3_plane_tail(w_planes, N, Kp); g.b_amax_n = 1;
    g.c_amax = y_amax_out;
    return dispatch_h3(g, (hipStream_t)stream);
}

int ttts_conv1d_fwd_h3_bn_blocks(int B, int T, int cin, int cout, int taps) {
    return h3_bn_blocks((long)B * T, cout, (long)taps * h3_pad32(cin));
}

int ttts_conv1d_fwd_h3_bn_chunk_rows(int B, int T, int cin, int cout, int taps) {
    // rows of (b, t) one partial covers (0: no partials for this shape): partial i holds rows i * chunk .. of the B T rows, so a
    // caller may finish the statistics of a row range that is a whole number of chunks on its own (the two forwards of a
    // training step run as one batch: ops.ConvBNFn, twin batches)
    return h3_bn_blocks((long)B * T, cout, (long)taps * h3_pad32(cin)) > 0 ? h3_bn_chunk_rows((long)B * T, cout, (long)taps * h3_pad32(cin)) : 0;
}

int ttts_conv1d_fwd_h3(const float* x, const void* planes_fwd, const float* bias, float* y, int B, int T, int cin, int cout,
                       int taps, const float* x_amax, float* bn_partials, void* stream) {
    TTTS_REQUIRE(x && planes_fwd && y && x_amax, "conv1d_fwd_h3: null pointer (x_amax, the partial maxima of |x|, is required)");
    TTTS_REQUIRE(B > 0 && T > 0 && cin > 0 && cout > 0 && taps > 0 && (taps & 1), "conv1d_fwd_h3: bad dims");
    TTTS_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "conv1d_fwd_h3: cin=%d and cout=%d must be multiples of 4", cin, cout);
    TTTS_REQUIRE((uint64_t)B * T * cin * 4 < (1ull << 32), "conv1d_fwd_h3: activation larger than 4 GiB");
    TTTS_REQUIRE(aligned16(x) && aligned16(planes_fwd), "conv1d_fwd_h3: pointers must be 16-byte aligned");
    GemmArgs g = base_args();
    const int cinp = h3_pad32(cin);                  // channels per tap of the padded image (see ttts_linear_fwd_h3)
    g.A = x; g.B = (const float*)planes_fwd; g.C = y; g.M = B * T; g.N = cout; g.K = taps * cinp;
    g.lda = cin; g.ldb = (long)taps * cinp; g.ldc = cout;
    g.a_bytes = (uint32_t)((uint64_t)B * T * cin * 4); g.b_bytes = (uint32_t)((uint64_t)cout * taps * cinp * 4);
    g.T = T; g.cin = cinp; g.shift0 = -((taps - 1) / 2); g.shift_step = 1;
    g.bias = bias;
    g.a_amax = x_amax; g.a_amax_n = H3_AMAX_PARTIALS;
    g.b_amax = h3_plane_tail(planes_fwd, cout, (long)taps * cinp); g.b_amax_n = 1;
    TTTS_REQUIRE(bn_partials == nullptr || (((uintptr_t)bn_partials) & 15) == 0, "conv1d_fwd_h3: bn_partials must be 16-byte aligned");
    g.bn_ws = bn_partials;
    return dispatch_h3(g, (hipStream_t)stream);
}

int ttts_linear_bwd_data_h3(const float* dy, const void* wt_planes, const float* residual, float* dx, int64_t M, int N,
                            int K, const float* relu_out, float relu_scale, const float* dy_amax, float* dx_amax_out,
                            void* stream) {
    // dx[M,K] = dy[M,N] . w[N,K] (+ residual) in the fp16x3 form; wt_planes = weight_split mode 5 (w^T as [K][N] rows);
    // dy_amax = the partial maxima of |dy| written by ttts_amax_partials (the dynamic pre-scale of the gradient operand)
    TTTS_REQUIRE(dy && wt_planes && dx && dy_amax, "linear_bwd_data_h3: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_bwd_data_h3: bad dims");
    TTTS_REQUIRE(N % 4 == 0 && K % 4 == 0, "linear_bwd_data_h3: N=%d and K=%d must be multiples of 4", N, K);
    TTTS_REQUIRE(aligned16(dy) && aligned16(wt_planes), "linear_bwd_data_h3: pointers must be 16-byte aligned");
    TTTS_REQUIRE((uint64_t)M * N * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_bwd_data_h3: operand larger than 4 GiB");
    GemmArgs g = base_args();
    const int Np = h3_pad32(N);                      // reduction depth of the padded image (see ttts_linear_fwd_h3)
    g.A = dy; g.B = (const float*)wt_planes; g.C = dx; g.M = (int)M; g.N = K; g.K = Np;
    g.lda = N; g.ldb = Np; g.ldc = K; g.cin = Np;
    g.a_bytes = (uint32_t)((uint64_t)M * N * 4); g.b_bytes = (uint32_t)((uint64_t)Np * K * 4);
    g.residual = residual; g.ldr = K;
    g.relu_out = relu_out; g.relu_scale = relu_scale;
    g.a_amax = dy_amax; g.a_amax_n = H3_AMAX_PARTIALS;
    g.b_amax = h3_plane_tail(wt_planes, K, Np); g.b_amax_n = 1;
    g.c_amax = dx_amax_out;
    return dispatch_h3(g, (hipStream_t)stream);
}

int ttts_conv1d_bwd_data_h3(const float* dy, const void* planes_bwd, float* dx, int B, int T, int cin, int cout, int taps,
                            const float* dy_amax, void* stream) {
    TTTS_REQUIRE(dy && planes_bwd && dx && dy_amax, "conv1d_bwd_data_h3: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && cin > 0 && cout > 0 && taps > 0 && (taps & 1), "conv1d_bwd_data_h3: bad dims");
    TTTS_REQUIRE(cout % 4 == 0 && cin % 4 == 0, "conv1d_bwd_data_h3: cout=%d and cin=%d must be multiples of 4", cout, cin);
    TTTS_REQUIRE((uint64_t)B * T * cout * 4 < (1ull << 32), "conv1d_bwd_data_h3: activation larger than 4 GiB");
    TTTS_REQUIRE(aligned16(dy) && aligned16(planes_bwd), "conv1d_bwd_data_h3: pointers must be 16-byte aligned");
    GemmArgs g = base_args();
    const int coutp = h3_pad32(cout);                // channels per tap of the padded image (see ttts_linear_fwd_h3)
    g.A = dy; g.B = (const float*)planes_bwd; g.C = dx; g.M = B * T; g.N = cin; g.K = taps * coutp;
    g.lda = cout; g.ldb = (long)taps * coutp; g.ldc = cin;
    g.a_bytes = (uint32_t)((uint64_t)B * T * cout * 4); g.b_bytes = (uint32_t)((uint64_t)cin * taps * coutp * 4);
    g.T = T; g.cin = coutp; g.shift0 = (taps - 1) / 2; g.shift_step = -1;
    g.a_amax = dy_amax; g.a_amax_n = H3_AMAX_PARTIALS;
    g.b_amax = h3_plane_tail(planes_bwd, cin, (long)taps * coutp); g.b_amax_n = 1;
    return dispatch_h3(g, (hipStream_t)stream);
}

int ttts_linear_bwd_data_x6(const float* dy, const void* wt_planes, const float* residual, float* dx, int64_t M, int N,
                            int K, const float* relu_out, float relu_scale, void* stream) {
    // dx[M,K] = dy[M,N] . w[N,K] (+ residual); wt_planes = split of w^T, i.e. [K][N] rows (weight_split mode 1)
    TTTS_REQUIRE(dy && wt_planes && dx, "linear_bwd_data_x6: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_bwd_data_x6: bad dims");
    TTTS_REQUIRE(N % BK == 0, "linear_bwd_data_x6: N=%d must be a multiple of 16", N);
    TTTS_REQUIRE(aligned16(dy) && aligned16(wt_planes), "linear_bwd_data_x6: pointers must be 16-byte aligned");
    TTTS_REQUIRE((uint64_t)M * N * 4 < (1ull << 32) && (uint64_t)N * K * 6 < (1ull << 32), "linear_bwd_data_x6: operand larger than 4 GiB");
    GemmArgs g = base_args();
    g.A = dy; g.B = (const float*)wt_planes; g.C = dx; g.M = (int)M; g.N = K; g.K = N;
    g.lda = N; g.ldb = N; g.ldc = K; g.cin = N;
    g.a_bytes = (uint32_t)((uint64_t)M * N * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 6);
    g.residual = residual; g.ldr = K;
    g.relu_out = relu_out; g.relu_scale = relu_scale;
    return dispatch_split(g, (hipStream_t)stream);
}

int ttts_conv1d_fwd_x6(const float* x, const void* planes_fwd, const float* bias, float* y, int B, int T, int cin, int cout,
                       int taps, void* stream) {
    TTTS_REQUIRE(x && planes_fwd && y, "conv1d_fwd_x6: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && cin > 0 && cout > 0 && taps > 0 && (taps & 1), "conv1d_fwd_x6: bad dims");
    TTTS_REQUIRE(cin % BK == 0, "conv1d_fwd_x6: cin=%d must be a multiple of %d", cin, BK);
    TTTS_REQUIRE((uint64_t)B * T * cin * 4 < (1ull << 32), "conv1d_fwd_x6: activation larger than 4 GiB");
    TTTS_REQUIRE(aligned16(x) && aligned16(planes_fwd), "conv1d_fwd_x6: pointers must be 16-byte aligned");
    GemmArgs g = base_args();
    g.A = x; g.B = (const float*)planes_fwd; g.C = y; g.M = B * T; g.N = cout; g.K = taps * cin;
    g.lda = cin; g.ldb = (long)taps * cin; g.ldc = cout;
    g.a_bytes = (uint32_t)((uint64_t)B * T * cin * 4); g.b_bytes = (uint32_t)((uint64_t)cout * taps * cin * 6);
    g.T = T; g.cin = cin; g.shift0 = -((taps - 1) / 2); g.shift_step = 1;
    g.bias = bias;
    return dispatch_split(g, (hipStream_t)stream);
}

int ttts_conv1d_bwd_data_x6(const float* dy, const void* planes_bwd, float* dx, int B, int T, int cin, int cout, int taps,
                            void* stream) {
    TTTS_REQUIRE(dy && planes_bwd && dx, "conv1d_bwd_data_x6: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && cin > 0 && cout > 0 && taps > 0 && (taps & 1), "conv1d_bwd_data_x6: bad dims");
    TTTS_REQUIRE(cout % BK == 0, "conv1d_bwd_data_x6: cout=%d must be a multiple of %d", cout, BK);
    TTTS_REQUIRE((uint64_t)B * T * cout * 4 < (1ull << 32), "conv1d_bwd_data_x6: activation larger than 4 GiB");
    TTTS_REQUIRE(aligned16(dy) && aligned16(planes_bwd), "conv1d_bwd_data_x6: pointers must be 16-byte aligned");
    GemmArgs g = base_args();
    g.A = dy; g.B = (const float*)planes_bwd; g.C = dx; g.M = B * T; g.N = cin; g.K = taps * cout;
    g.lda = cout; g.ldb = (long)taps * cout; g.ldc = cin;
    g.a_bytes = (uint32_t)((uint64_t)B * T * cout * 4); g.b_bytes = (uint32_t)((uint64_t)cin * taps * cout * 6);
    g.T = T; g.cin = cout; g.shift0 = (taps - 1) / 2; g.shift_step = -1;
    return dispatch_split(g, (hipStream_t)stream);
}

}  // extern "C"
