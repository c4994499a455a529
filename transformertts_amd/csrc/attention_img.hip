// Attention on HEAD-IMAGE operands (fp16x3 products, head_dim 64): q / k / v arrive as the in-projection's epilogue wrote them
// (gemm_h3i.hip, IMG): per row and 64-column head, {64 f16 "hi", 64 f16 "lo"} of x * 2^e(row, head) in the 256 bytes the fp32
// columns would occupy, with 2^-e in `*_inv[head * rows + row]`.  An in-projection's output has no reader but attention, so the
// split costs no extra traffic, and here it removes what the fp32-operand kernels (attention.hip) spent most of their time
// on: a K / V tile was loaded to registers, split (16 VALU cycles per score element), written to LDS and only then multiplied,
// between two barriers, with the tile's global latency exposed (38-41 % of wave time parked).  Now
//   * K / V tiles (64 keys x {hi, lo} x 128 B) go global -> LDS by `buffer_load_dwordx4 ... lds`, TWO tiles deep: the tile of
//     step t+1 is requested before the products of step t and waited for behind them -- one barrier per tile, no staging
//     registers, no split arithmetic;
//   * the LDS image is one layout for both kinds of read: 128-byte plane rows whose 16-byte chunk index is XORed with
//     f(row) = ((row & 2) << 1) | ((row >> 2) & 3) -- applied to the SOURCE address of each DMA lane (the DMA writes
//     lane-linearly).  Row reads (`ds_read_b128`: K for the scores, V for dP, Q / dO in the dK / dV kernel) and transposed
//     reads (`ds_read_b64_tr_b16`: V^T for O += P V, K^T for dQ, Q^T / dO^T) are both conflict-free on it, so no operand is
//     stored twice;
//   * the per-(row, head) scales are what makes the image writable by a GEMM epilogue (a tile owns whole head rows but not the
//     tensor's maximum).  A query's scale is lane-local here (the query is on the lane) and rides in the exp2's multiplier;
//     a key's scale multiplies its score register (16 multiplies per 32 x 32 tile); V's folds into the probabilities before
//     their split, normalised by the power of two of V's tensor maximum (which the producer publishes) so that no factor
//     exceeds 1.  All are powers of two: the scores are bit-identical functions of the same MFMA accumulators in the forward
//     and both backward kernels, as before (rowstat).
// Orientation, masks, dropout, row statistics and the one-hot handling are those of attention.hip's fp16x3 kernels.
#include "attention_common.h"

namespace ttts {

struct AttnImgArgs {
    const void* q; const void* k; const void* v;          // head images, already offset to (section, head 0): 4-byte cells
    const float* q_inv; const float* k_inv; const float* v_inv;   // inverse scales of head 0 of the section: [head][rows]
    long q_rows, k_rows;                                  // rows per head plane of the q-side / key-side inverse scales (B * Tq, B * Tk,
    //                                                       or more: a batch that is the first part of a larger image)
    long stat_plane;                                      // floats per row-statistic plane (B * H * Tq, or more, likewise)
    float* o; float* lse; float* attn;
    const float* dout; float* delta; float* dq; float* dk; float* dv;
    const int64_t* key_lens;
    int B, H, Tq, Tk;
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;             // row strides in 4-byte cells
    float qscale;
    float drop_scale; uint32_t thr; uint64_t seed; const uint64_t* step_seed;
    const float* v_amax;                                  // TTTS_AMAX_SLOTS partial maxima of |v| (the producer's section array)
    const float* do_amax;
    float* amax_dq; float* amax_dkv;
    // dK / dV kernel with the QUERY range split over gridDim.z workgroups (cross-attention: one 128-key block per (batch, head) is
    // 256 workgroups for 256 CUs, each streaming every query): split z writes its partial sums to dkv_part + z * part_stride in
    // the layout of the packed (dk, dv) gradient, and attn_dkv_reduce_kernel adds the splits in fixed order
    float* dkv_part; long part_stride;
    int ldp;                                              // row stride of the partial sums: 2 H 64 (dk | dv packed), whatever lddk is
    float* o_amax;
    // (5, B, H, Tq): 0 = exponent subtrahend mcs, 1 = log2 of the row sum, 2 = one-hot flag (as attention.hip), 3 = the row's
    // exponent multiplier c2 = 2^-e_q * qscale * log2(e), 4 = its score multiplier c = 2^-e_q * qscale
    float* rowstat;
};

__device__ __forceinline__ uint32_t lds_addr_a(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));
// 64 lanes x 16 bytes (descriptor, per-lane byte offset + scalar byte offset) -> LDS at lds_dst + 16 * lane (see gemm_h3i.hip)
__device__ __forceinline__ void dma16a(u32x4a rsrc, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
// 64 lanes x 4 bytes -> LDS at lds_dst + 4 * lane
__device__ __forceinline__ void dma4a(u32x4a rsrc, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ u32x4a make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return u32x4a{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}

// ---- the tile image: plane p (0 hi, 1 lo) of a 64-row tile = 64 rows x 128 bytes; 16-byte chunk c of row r sits at
//      r * 128 + ((c ^ isw(r)) << 4)
constexpr int IMG_PLANE = 64 * 128;                 // bytes
constexpr int IMG_TILE = 2 * IMG_PLANE;             // one operand tile, both planes: 16 KB
__device__ __forceinline__ int isw(int r) { return ((r & 2) << 1) | ((r >> 2) & 3); }
typedef short s16x4v __attribute__((ext_vector_type(4)));
// transposed read: per 16-lane group a block of 4 rows x 16 columns of f16, delivered column-major (lane i of the group gets
// column i, rows 0..3); the lane passes the address of (row q = (lane & 15) >> 2, columns 4 (lane & 3) ..)
__device__ __forceinline__ s16x4v lds_tr4(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4v*)(uintptr_t)lds_addr_a(p));
}
__device__ __forceinline__ f16x8v join_tr(s16x4v a, s16x4v b) {
    typedef short s16x8v __attribute__((ext_vector_type(8)));
    const s16x8v j = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8v, j);
}

// E = the power of two that puts max|v| (over the tensor) in [2^11, 2^12): the partial maxima are read ONCE per workgroup, a
// quarter per wave (attention.hip, attn_wg_quarter_max); `red` = 4 floats of LDS
__device__ __forceinline__ void img_tensor_scale(const float* __restrict__ partials, int lane, int wave, float* red, float& scale,
                                                 float& inv) {
    static_assert(TTTS_AMAX_SLOTS == 1024, "four waves x 64 lanes x float4");
    const float4 a = reinterpret_cast<const float4*>(partials)[wave * 64 + lane];
    const float m = wave_max(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    float s, i;
    h3_pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), s, i);
    scale = h3_uniform(s);
    inv = h3_uniform(i);
}

// development ablation (never defined in the product build; results are WRONG, timing only): -DTTTS_AIMG_ABL_SAMEKV makes every
// workgroup stream the K / V (dK / dV kernel: the Q / dO) of (batch 0, head 0) -- every tile request then hits L2 --, the upper
// bound of what ANY K / V reuse scheme between a head's query blocks could buy
#ifdef TTTS_AIMG_ABL_SAMEKV
#define AIMG_SB(b) 0
#define AIMG_SH(h) 0
#else
#define AIMG_SB(b) (b)
#define AIMG_SH(h) (h)
#endif

#ifdef TTTS_AIMG_STAMPS
// development aid (tools/aimg_stamps.py; never defined in the product build): per (workgroup, wave) sums of s_memtime ticks of the
// causal forward: 0 waiting for a tile (vmcnt + barrier), 1 issuing the next tile's DMAs, 2 score products + key scales,
// 3 softmax + dropout, 4 P split + V^T reads + products, 5 whole kernel, 6 sub-tiles done, 7 prologue
__device__ unsigned long long ttts_aimg_stamps[2048 * 4 * 8];
#define ASTAMP() __builtin_amdgcn_s_memtime()
#define AACC(slot, v) do { if ((threadIdx.x & 63) == 0) st_acc[slot] += (v); } while (0)
#else
#define ASTAMP() 0ull
#define AACC(slot, v)
#endif

// ===================================================================================== forward
// LDS: two stages of {K planes, V planes, key scales, value scales} of KT keys; the stages double as the per-wave fp32 scratch of
// the epilogue.  KT = 64: 16 KB per operand tile, two workgroups per CU; KT = 32: half of that and THREE workgroups per CU (the
// kernel's 156 registers allow three waves per SIMD) -- twice the barriers per key, but a third more waves to cover the
// latencies a wave cannot cover itself (exp, LDS, MFMA results): causal self-attention 64 x 4 x 870, back to back, 131-138 us
// against 152 (tools/aimg_time.py).  Measured and not kept: the score products of tile t+1 issued BEFORE the softmax of tile t
// (a ring of three 32-key stages, 150 registers, still three workgroups per CU): 133-138 us -- with three waves per SIMD
// the hardware already runs one wave's products under another's arithmetic.
constexpr int FI_STAGE = 2 * IMG_TILE + 512;          // (KT = 64; the backward kernels)
constexpr int OUT_LD = 68;                            // floats per row of the epilogue's scratch: 16-byte rows, 4-bank skew
constexpr int OUT_BYTES = 4 * 32 * OUT_LD * 4;
// write a 2 x (32x32) accumulator pair holding X^T[d][row] (row on the lane) as rows of 64 floats, 16 bytes per lane and store
__device__ __forceinline__ void wave_store_rows4(const f32x16 (&acc)[2], float* scratch, float* gbase, long row0, long nrows_total,
                                                 int ld, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[l31 * OUT_LD + blk * 32 + acc_row(r, half)] = acc[blk][r];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4), c4 = (lane & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(scratch + row * OUT_LD + c4);
        if (row0 + row < nrows_total) *reinterpret_cast<float4*>(gbase + (row0 + row) * ld + c4) = v;
    }
    wave_lds_sync();
}
template <bool CAUSAL, bool WRITE_A, int KT>
__global__ __launch_bounds__(256, KT == 32 ? 3 : 2) void attn_fwd_img_kernel(AttnImgArgs a) {
    constexpr int PL = KT * 128;                      // bytes of one plane of a tile
    constexpr int TILE = 2 * PL, STG = 2 * TILE + 512, NSUB = KT / 32;
    constexpr int XS = 2 * STG > OUT_BYTES ? 2 * STG : OUT_BYTES;
#ifdef TTTS_AIMG_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = ASTAMP();
#endif
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) char xs[XS];
    __shared__ float ptile_all[WRITE_A ? 4 * 32 * 17 : 1];   // per wave: 32 queries x 16 keys (+1 pad)
    __shared__ float red4[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    float* ptile = ptile_all + (WRITE_A ? wave * 32 * 17 : 0);
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = reinterpret_cast<float*>(xs) + wave * 32 * OUT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst_live = (kend + KT - 1) / KT;
    const int nst = WRITE_A ? (a.Tk + KT - 1) / KT : nst_live;
    int wave_kend = WRITE_A ? a.Tk : kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    float Ev, inv_Ev;
    img_tensor_scale(a.v_amax, lane, wave, red4, Ev, inv_Ev);

    // ---- Q fragments straight from the image: lane (query l31, half) holds Q'[q][16 s + 8 half + 0..7], hi and lo
    const int qrow = qg < a.Tq ? qg : a.Tq - 1;
    f16x8v qf[4][2];
    {
        const char* qp = reinterpret_cast<const char*>(a.q) + ((long)(b * a.Tq + qrow) * a.ldq + h * HD) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p) qf[s][p] = *reinterpret_cast<const f16x8v*>(qp + p * 128 + (2 * s + half) * 16);
    }
    // accumulator units -> true score: 2^-e_q * qscale (the key's own 2^-e_k multiplies the accumulator first)
    const float c_q = a.q_inv[(long)h * a.q_rows + (long)b * a.Tq + qrow] * a.qscale;
    const float c2_q = c_q * 1.4426950408889634f;

    // ---- K / V tiles by LDS-DMA.  This wave moves rows (KT / 4) w .. of each plane in 8-row pieces: lane -> (row lane / 8 of the
    // piece, chunk slot lane & 7), and the chunk it FETCHES is slot ^ isw(row).  Rows past the last key are clamped to it (finite data the masks
    // remove; nothing of another allocation is touched).
    const u32x4a rsK = make_rsrc(reinterpret_cast<const char*>(a.k) + ((long)AIMG_SB(b) * a.Tk * a.ldk + AIMG_SH(h) * HD) * 4, (uint32_t)a.Tk * (uint32_t)a.ldk * 4u);
    const u32x4a rsV = make_rsrc(reinterpret_cast<const char*>(a.v) + ((long)AIMG_SB(b) * a.Tk * a.ldv + AIMG_SH(h) * HD) * 4, (uint32_t)a.Tk * (uint32_t)a.ldv * 4u);
    const u32x4a rsKi = make_rsrc(a.k_inv + (long)AIMG_SH(h) * a.k_rows + (long)AIMG_SB(b) * a.Tk, (uint32_t)a.Tk * 4u);
    const u32x4a rsVi = make_rsrc(a.v_inv + (long)AIMG_SH(h) * a.k_rows + (long)AIMG_SB(b) * a.Tk, (uint32_t)a.Tk * 4u);
    const uint32_t lds0 = lds_addr_a(xs);
    const int ld_r = lane >> 3;                                       // row within an 8-row piece
    const uint32_t ld_c0 = (uint32_t)((lane & 7) ^ isw(ld_r)) * 16u;  // chunk fetched for piece nn = 0; nn = 1: ^ 32 bytes
    auto issue = [&](int t, int stage, bool with_v) {
        const uint32_t dst = lds0 + (uint32_t)stage * STG;
#pragma unroll
        for (int nn = 0; nn < KT / 32; ++nn) {
            const int piece_row = (KT / 4) * wave + 8 * nn;           // first row of this 8-row piece inside the tile
            int row = t * KT + piece_row + ld_r;
            row = row < a.Tk ? row : a.Tk - 1;
            const uint32_t cb = ld_c0 ^ (uint32_t)(((piece_row >> 3) & 1) * 32);      // isw of the piece's rows: bit 3 of the row flips chunk bit 1
            const uint32_t ko = (uint32_t)row * (uint32_t)(a.ldk * 4) + cb, vo = (uint32_t)row * (uint32_t)(a.ldv * 4) + cb;
            const uint32_t piece = (uint32_t)piece_row * 128u;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                dma16a(rsK, ko, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + (uint32_t)p * PL + piece));
                if (with_v) dma16a(rsV, vo, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + TILE + (uint32_t)p * PL + piece));
            }
        }
        if (wave < 2) {                                               // wave 0: the tile's key scales; wave 1: its value scales
            int key = t * KT + lane;                                  // (64 lanes: the KT = 32 tile takes its neighbour's along)
            key = key < a.Tk ? key : a.Tk - 1;
            if (wave == 0) dma4a(rsKi, (uint32_t)key * 4u, __builtin_amdgcn_readfirstlane(dst + 2 * TILE));
            else if (with_v) dma4a(rsVi, (uint32_t)key * 4u, __builtin_amdgcn_readfirstlane(dst + 2 * TILE + 256));
        }
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };

    // fragment addresses (bytes inside a stage)
    const int fsw = isw(l31);
    uint32_t k_off[4];                 // K row l31 (+ 32 sub), chunk 2 st + half
#pragma unroll
    for (int st = 0; st < 4; ++st) k_off[st] = (uint32_t)(l31 * 128 + (((2 * st + half) ^ fsw) << 4));
    // V^T by transposed reads: lane -> (row 4 half + q4, 8-byte piece pc of 16-column group g16) of a 4-row block
    const int q4 = (lane & 15) >> 2, pc = lane & 3, g16 = (lane >> 4) & 1;
    uint32_t v_off[2][2];              // [i2][second]: d block 32 i2, keys +0..3 / +8..11 of the 16-key step
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            const int r = 8 * sc + 4 * half + q4;                    // row inside the 16-key step (isw of the full row: same bits)
            const int ch = 4 * i2 + 2 * g16 + (pc >> 1);
            v_off[i2][sc] = (uint32_t)(TILE + r * 128 + ((ch ^ isw(r)) << 4) + (pc & 1) * 8);
        }

    float m = NEG_INF, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);

    // scores of one 32-key sub-tile in accumulator units of (q', k), times the keys' 2^-e_k: t = 2^e_q (q . k)
    auto scores = [&](const char* st_, int sub, f32x16& s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            f16x8v kf[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) kf[p] = *reinterpret_cast<const f16x8v*>(st_ + p * PL + sub * 4096 + k_off[st]);
            mfma_h3(s, kf, qf[st]);
        }
        const float* ki = reinterpret_cast<const float*>(st_ + 2 * TILE) + sub * 32 + 4 * half;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 f = *reinterpret_cast<const float4*>(ki + 8 * g4);
            s[4 * g4] *= f.x; s[4 * g4 + 1] *= f.y; s[4 * g4 + 2] *= f.z; s[4 * g4 + 3] *= f.w;
        }
    };
    auto alive = [&](int key_g) -> bool { return key_g < klen && (!CAUSAL || key_g <= qg); };
    auto drop16 = [&](float (&p)[16], int key0) {
#pragma unroll
        for (int r = 0; r < 16; r += 4) {     // registers r .. r+3 are four neighbouring keys: one hash
            const uint32_t qh = attn_quad_hash(seed_eff, rowid, (uint32_t)(key0 + acc_row(r, half)) >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool keep = attn_keep_word(qh, attn_drop_mult(e), thr16);
                p[r + e] = keep ? (WRITE_A ? p[r + e] * a.drop_scale : p[r + e]) : 0.f;
            }
        }
    };

    if (WRITE_A) {
        // ---------------- pass 1: row max / row sum only (K tiles alone)
        if (nst_live > 0) issue(0, 0, false);
        landed();
        for (int t = 0; t < nst_live; ++t) {
            if (t + 1 < nst_live) issue(t + 1, (t + 1) & 1, false);
            const char* st_ = xs + (t & 1) * STG;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int key0 = t * KT + sub * 32;
                if (key0 >= kend) break;
                f32x16 s;
                scores(st_, sub, s);
                float mx = NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = alive(key0 + acc_row(r, half)) ? s[r] : NEG_INF;
                    mx = fmaxf(mx, s[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float m_new = fmaxf(m, mx);
                float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                float alpha = fast_exp2((m - m_use) * c2_q);
                const float mc = m_use * c2_q;
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += fast_exp2(__builtin_fmaf(s[r], c2_q, -mc));
                l = l * alpha + ps;
                m = m_new;
            }
            landed();
        }
        l = l + __shfl_xor(l, 32, 64);
    }

    const float m_fin = (m == NEG_INF) ? 0.f : m;
    const float inv_l = (l > 0.f) ? 1.f / l : 0.f;
    const float mcs_fin = m_fin * c2_q;         // ONE rounded product per row, kept as it is (rowstat): see attention.hip

    // ---------------- main pass
    AACC(7, ASTAMP() - st_begin);
    if (nst > 0) issue(0, 0, true);
    landed();
    for (int t = 0; t < nst; ++t) {
        // the tile of step t+1 goes into the stage every wave finished reading before the barrier that ended step t-1
        [[maybe_unused]] const unsigned long long s_i0 = ASTAMP();
        if (t + 1 < nst) issue(t + 1, (t + 1) & 1, true);
        AACC(1, ASTAMP() - s_i0);
        const char* st_ = xs + (t & 1) * STG;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int key0 = t * KT + sub * 32;
            if (key0 >= wave_kend) break;
            f32x16 s;
            [[maybe_unused]] const unsigned long long s_a = ASTAMP();
            scores(st_, sub, s);
            [[maybe_unused]] const unsigned long long s_b = ASTAMP();
            float p[16];
            const bool full = (key0 + 32 <= klen) && (!CAUSAL || key0 + 31 <= qw0);     // wave-uniform: no mask arithmetic
            if (WRITE_A) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    p[r] = alive(key0 + acc_row(r, half)) ? fast_exp2(__builtin_fmaf(s[r], c2_q, -mcs_fin)) * inv_l : 0.f;
            } else {
                float mx = NEG_INF;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s[r] = alive(key0 + acc_row(r, half)) ? s[r] : NEG_INF;
                        mx = fmaxf(mx, s[r]);
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float m_new = fmaxf(m, mx);
                float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                float alpha = fast_exp2((m - m_use) * c2_q);
                // the weights are born pre-scaled by 2^10 (the f16 split scale rides in the exponent): l sums them scaled
                const float mc = __builtin_fmaf(m_use, c2_q, -10.f);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = fast_exp2(__builtin_fmaf(s[r], c2_q, -mc)); ps += p[r]; }
                l = l * alpha + ps;
                m = m_new;
                if (__any(alpha != 1.f)) {   // the running maximum rarely moves after the first tiles
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                }
            }
            if (a.thr != 0u) drop16(p, key0);
            if (WRITE_A) {
                // transpose through LDS in two halves of 16 keys so that the weights leave as 64-byte row segments
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ptile[l31 * 17 + (acc_row(8 * g2 + e, half) & 15)] = p[8 * g2 + e];
                    wave_lds_sync();
#pragma unroll 4
                    for (int i = 0; i < 8; ++i) {
                        const int qr = 4 * i + (lane >> 4), kc = lane & 15;
                        const float v = ptile[qr * 17 + kc];
                        const int q_g = qw0 + qr, key_g = key0 + 16 * g2 + kc;
                        if (q_g < a.Tq && key_g < a.Tk) a.attn[(arow + q_g) * a.Tk + key_g] = v;
                    }
                    wave_lds_sync();
                }
            }
            [[maybe_unused]] const unsigned long long s_c = ASTAMP();
            // O^T[d][q] += V'^T[d][key] P''^T[key][q], P'' = P (2^10) 2^-e_v(key) E: two 16-key steps, registers 8 t2 .. 8 t2 + 7 of
            // the lane are its B fragment and the transposed reads deliver V'^T in exactly that key order
            const float* vi = reinterpret_cast<const float*>(st_ + 2 * TILE + 256) + sub * 32 + 4 * half;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float4 f0 = *reinterpret_cast<const float4*>(vi + 16 * t2), f1 = *reinterpret_cast<const float4*>(vi + 16 * t2 + 8);
                // 2^-e_v(key) E <= 1 for every row the maximum covers; the cap only meets rows that carry no scale of their own
                // (all zero, or below 2^-103: h3_pow2_scale leaves 1) under a large E -- their V' is 0 and must not meet an infinity
                const float pscale = WRITE_A ? H3A_P * Ev : Ev, cap = WRITE_A ? H3A_P : 1.f;
                const float vf8[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = p[8 * t2 + e] * fminf(vf8[e] * pscale, cap);
                f16x8v pf[2];
                split_frag8_h3(x, 1.0f, pf[0], pf[1]);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    f16x8v vf[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const char* vb = st_ + pl * PL + sub * 4096 + t2 * 2048;
                        vf[pl] = join_tr(lds_tr4(vb + v_off[i2][0]), lds_tr4(vb + v_off[i2][1]));
                    }
                    mfma_h3(o[i2], vf, pf);
                }
            }
#ifdef TTTS_AIMG_STAMPS
            asm volatile("" :: "v"(o[0]), "v"(o[1]));
            { const unsigned long long s_d = ASTAMP(); AACC(2, s_b - s_a); AACC(3, s_c - s_b); AACC(4, s_d - s_c); AACC(6, 1); }
#endif
        }
        [[maybe_unused]] const unsigned long long s_w = ASTAMP();
        landed();
        AACC(0, ASTAMP() - s_w);
    }

    float out_scale = inv_Ev * (1.0f / H3A_P);          // the O accumulator holds (V')^T (P 2^10 E 2^-e_v)^T
    float lse_v;
    if (WRITE_A) {
        lse_v = m_fin * c_q + __logf(l > 0.f ? l : 1.f);
    } else {
        float lt = l + __shfl_xor(l, 32, 64);                 // = 2^10 * the sum of the weights
        out_scale = (lt > 0.f) ? (a.drop_scale * inv_Ev) / lt : 0.f;
        lse_v = ((m == NEG_INF) ? 0.f : m) * c_q + __logf(lt > 0.f ? lt * (1.0f / H3A_P) : 1.f);
    }
    if (a.lse != nullptr && half == 0 && qg < a.Tq) a.lse[arow + qg] = lse_v;
    const float lsum = WRITE_A ? l : (l + __shfl_xor(l, 32, 64));
    if (a.rowstat != nullptr && half == 0 && qg < a.Tq) {
        const long plane = a.stat_plane;
        const float mcs_w = WRITE_A ? mcs_fin : ((m == NEG_INF) ? -10.f : __builtin_fmaf(m, c2_q, -10.f));
        a.rowstat[arow + qg] = mcs_w;
        a.rowstat[plane + arow + qg] = lsum > 0.f ? __log2f(lsum) : 0.f;
        // one-hot row: its sum IS its largest term (attention.hip, attn_fwd_h3_kernel)
        const float top = (m == NEG_INF) ? 0.f : fast_exp2(__builtin_fmaf(m, c2_q, -mcs_w));
        a.rowstat[2 * plane + arow + qg] = (lsum > 0.f && lsum == top) ? 1.f : 0.f;
        a.rowstat[3 * plane + arow + qg] = c2_q;
        a.rowstat[4 * plane + arow + qg] = c_q;
    }
    float omax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o[0][r] *= out_scale; o[1][r] *= out_scale;
        omax = fmaxf(omax, fmaxf(fabsf(o[0][r]), fabsf(o[1][r])));
    }
    if (a.o_amax != nullptr) amax_publish(qg < a.Tq ? omax : 0.f, a.o_amax, blockIdx.y * gridDim.x + blockIdx.x);
    // (every wave passed the loop's last barrier after its last read of the stages: the scratch that aliases them is free)
    wave_store_rows4(o, scratch, a.o + (long)b * a.Tq * a.ldo + h * HD, qw0, a.Tq, a.ldo, lane);
#ifdef TTTS_AIMG_STAMPS
    st_acc[5] = ASTAMP() - st_begin;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if ((threadIdx.x & 63) == 0 && wg < 2048)
        for (int i = 0; i < 8; ++i) ttts_aimg_stamps[(wg * 4 + (threadIdx.x >> 6)) * 8 + i] = st_acc[i];
#endif
}

// ===================================================================================== forward with the weights written, Tk <= 128
// Cross-attention returns its per-head weights (model/layers.py:68-73), normalised by the whole row's sum -- the generic kernel
// streams K twice for that (row maximum / sum first).  A phoneme sequence is at most 128 keys here: all of K and V (two 64-key
// tiles = the two stages) sit in LDS at once and a lane's whole score row (4 x 16 registers) stays in registers, so ONE pass
// does it: every score product once, one request of K and V.
__global__ __launch_bounds__(256, 2) void attn_fwd_img_maps_kernel(AttnImgArgs a) {
    constexpr int KT = 64, PL = KT * 128, TILE = 2 * PL, STG = 2 * TILE + 512;
    constexpr int XS = 2 * STG > OUT_BYTES ? 2 * STG : OUT_BYTES;
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) char xs[XS];
    __shared__ float red4[4];
    constexpr int PT_LD = 36;                          // floats per row of a wave's 32 x 32 weight tile (16-byte rows, 4-bank skew)
    static_assert(4 * 32 * PT_LD * 4 <= XS, "the weight tiles reuse the stages");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    float* ptile = reinterpret_cast<float*>(xs) + wave * 32 * PT_LD;     // (after the products: the stages are free then)
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int qw0 = blockIdx.y * QB + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = reinterpret_cast<float*>(xs) + wave * 32 * OUT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    const int nsub = (a.Tk + 31) / 32;                 // <= 4 (the launcher checks Tk <= 128)

    float Ev, inv_Ev;
    img_tensor_scale(a.v_amax, lane, wave, red4, Ev, inv_Ev);

    const int qrow = qg < a.Tq ? qg : a.Tq - 1;
    f16x8v qf[4][2];
    {
        const char* qp = reinterpret_cast<const char*>(a.q) + ((long)(b * a.Tq + qrow) * a.ldq + h * HD) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p) qf[s][p] = *reinterpret_cast<const f16x8v*>(qp + p * 128 + (2 * s + half) * 16);
    }
    const float c_q = a.q_inv[(long)h * a.q_rows + (long)b * a.Tq + qrow] * a.qscale;
    const float c2_q = c_q * 1.4426950408889634f;

    // both tiles at once (rows past the last key clamped to it: finite data the masks remove)
    {
        const u32x4a rsK = make_rsrc(reinterpret_cast<const char*>(a.k) + ((long)b * a.Tk * a.ldk + h * HD) * 4, (uint32_t)a.Tk * (uint32_t)a.ldk * 4u);
        const u32x4a rsV = make_rsrc(reinterpret_cast<const char*>(a.v) + ((long)b * a.Tk * a.ldv + h * HD) * 4, (uint32_t)a.Tk * (uint32_t)a.ldv * 4u);
        const u32x4a rsKi = make_rsrc(a.k_inv + (long)h * a.k_rows + (long)b * a.Tk, (uint32_t)a.Tk * 4u);
        const u32x4a rsVi = make_rsrc(a.v_inv + (long)h * a.k_rows + (long)b * a.Tk, (uint32_t)a.Tk * 4u);
        const uint32_t lds0 = lds_addr_a(xs);
        const int ld_r = lane >> 3;
        const uint32_t ld_c0 = (uint32_t)((lane & 7) ^ isw(ld_r)) * 16u;
        const int ntile = (a.Tk + KT - 1) / KT;
        for (int t = 0; t < ntile; ++t) {
            const uint32_t dst = lds0 + (uint32_t)t * STG;
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                int row = t * KT + 16 * wave + 8 * nn + ld_r;
                row = row < a.Tk ? row : a.Tk - 1;
                const uint32_t cb = ld_c0 ^ (uint32_t)(nn * 32);
                const uint32_t ko = (uint32_t)row * (uint32_t)(a.ldk * 4) + cb, vo = (uint32_t)row * (uint32_t)(a.ldv * 4) + cb;
                const uint32_t piece = (uint32_t)(16 * wave + 8 * nn) * 128u;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    dma16a(rsK, ko, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + (uint32_t)p * PL + piece));
                    dma16a(rsV, vo, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + TILE + (uint32_t)p * PL + piece));
                }
            }
            if (wave < 2) {
                int key = t * KT + lane;
                key = key < a.Tk ? key : a.Tk - 1;
                if (wave == 0) dma4a(rsKi, (uint32_t)key * 4u, __builtin_amdgcn_readfirstlane(dst + 2 * TILE));
                else dma4a(rsVi, (uint32_t)key * 4u, __builtin_amdgcn_readfirstlane(dst + 2 * TILE + 256));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }

    const int fsw = isw(l31);
    uint32_t k_off[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) k_off[st] = (uint32_t)(l31 * 128 + (((2 * st + half) ^ fsw) << 4));
    const int q4 = (lane & 15) >> 2, pc = lane & 3, g16 = (lane >> 4) & 1;
    uint32_t v_off[2][2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            const int r = 8 * sc + 4 * half + q4;
            const int ch = 4 * i2 + 2 * g16 + (pc >> 1);
            v_off[i2][sc] = (uint32_t)(TILE + r * 128 + ((ch ^ isw(r)) << 4) + (pc & 1) * 8);
        }

    // ---- every score of the row, key scales applied, masked
    f32x16 s[4];
    float mx = NEG_INF;
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[sb][r] = NEG_INF;
        if (sb < nsub) {
            const char* st_ = xs + (sb >> 1) * STG;
            const int sub = sb & 1;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8v kf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) kf[p] = *reinterpret_cast<const f16x8v*>(st_ + p * PL + sub * 4096 + k_off[st]);
                mfma_h3(acc, kf, qf[st]);
            }
            const float* ki = reinterpret_cast<const float*>(st_ + 2 * TILE) + sub * 32 + 4 * half;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 f = *reinterpret_cast<const float4*>(ki + 8 * g4);
                const float kf4[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const float v = (sb * 32 + acc_row(r, half) < klen) ? acc[r] * kf4[e] : NEG_INF;
                    s[sb][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_fin = (mx == NEG_INF) ? 0.f : mx;
    const float mcs_fin = m_fin * c2_q;                // ONE rounded product per row, kept as it is (rowstat)
    float l = 0.f;
#pragma unroll
    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {                 // exp2(-inf) = 0 for the masked keys
            s[sb][r] = fast_exp2(__builtin_fmaf(s[sb][r], c2_q, -mcs_fin));
            l += s[sb][r];
        }
    l = l + __shfl_xor(l, 32, 64);
    const float inv_l = (l > 0.f) ? 1.f / l : 0.f;

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
        if (sb >= nsub) break;
        const char* st_ = xs + (sb >> 1) * STG;
        const int sub = sb & 1, key0 = sb * 32;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = s[sb][r] * inv_l;
        if (a.thr != 0u) {
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const uint32_t qh = attn_quad_hash(seed_eff, rowid, (uint32_t)(key0 + acc_row(r, half)) >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) p[r + e] = attn_keep_word(qh, attn_drop_mult(e), thr16) ? p[r + e] * a.drop_scale : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[sb][r] = p[r];          // kept for the write-out below
        const float* vi = reinterpret_cast<const float*>(st_ + 2 * TILE + 256) + sub * 32 + 4 * half;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const float4 f0 = *reinterpret_cast<const float4*>(vi + 16 * t2), f1 = *reinterpret_cast<const float4*>(vi + 16 * t2 + 8);
            const float vf8[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = p[8 * t2 + e] * fminf(vf8[e] * (H3A_P * Ev), H3A_P);
            f16x8v pf[2];
            split_frag8_h3(x, 1.0f, pf[0], pf[1]);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                f16x8v vf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const char* vb = st_ + pl * PL + sub * 4096 + t2 * 2048;
                    vf[pl] = join_tr(lds_tr4(vb + v_off[i2][0]), lds_tr4(vb + v_off[i2][1]));
                }
                mfma_h3(o[i2], vf, pf);
            }
        }
    }

    // ---- the weights: a workgroup's 128 queries x Tk keys are ONE contiguous block of the map (rows of Tk floats follow each
    // other), so with 16-byte rows (Tk % 4 == 0) the block is assembled in the freed stage memory exactly as it lies in memory
    // (a lane writes its four consecutive keys of a query as one 16-byte piece) and leaves as a linear copy: every store
    // instruction writes 1 KB of consecutive bytes.  (Storing the 32 x 32 tiles directly put 128-byte segments at a 400-byte
    // stride: 1.8 TB/s.)  Other Tk: the tiles go out row by row, tail keys one by one.
    __syncthreads();                                   // every wave is done with K / V
    const bool rows16 = (a.Tk & 3) == 0;
    if (rows16) {
        float* blk = reinterpret_cast<float*>(xs);     // [128 queries][Tk]
        static_assert(QB * 128 * 4 <= XS, "the block of weights fits the stages");
        const int q_l = wave * 32 + l31;
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            if (sb >= nsub) break;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int key = sb * 32 + acc_row(r, half);
                if (key < a.Tk)
                    *reinterpret_cast<float4*>(blk + q_l * a.Tk + key) = make_float4(s[sb][r], s[sb][r + 1], s[sb][r + 2], s[sb][r + 3]);
            }
        }
        __syncthreads();
        const int q0 = blockIdx.y * QB;
        const int nq = (a.Tq - q0) < QB ? (a.Tq - q0) : QB;
        const int n4 = nq * (a.Tk >> 2);
        float4* dst = reinterpret_cast<float4*>(a.attn + (arow + q0) * a.Tk);
        for (int i = tid; i < n4; i += 256) dst[i] = reinterpret_cast<const float4*>(blk)[i];
    } else {
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            if (sb >= nsub) break;
            const int key0 = sb * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) ptile[l31 * PT_LD + acc_row(r, half)] = s[sb][r];
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qr = 8 * i + (lane >> 3), k4 = (lane & 7) * 4;
                const float4 v = *reinterpret_cast<const float4*>(ptile + qr * PT_LD + k4);
                const int q_g = qw0 + qr, key_g = key0 + k4;
                if (q_g < a.Tq && key_g < a.Tk) {
                    float* dst = a.attn + (arow + q_g) * a.Tk + key_g;
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (key_g + e < a.Tk) dst[e] = vv[e];
                }
            }
            wave_lds_sync();
        }
    }

    const float out_scale = inv_Ev * (1.0f / H3A_P);
    if (a.lse != nullptr && half == 0 && qg < a.Tq) a.lse[arow + qg] = m_fin * c_q + __logf(l > 0.f ? l : 1.f);
    if (a.rowstat != nullptr && half == 0 && qg < a.Tq) {
        const long plane = a.stat_plane;
        a.rowstat[arow + qg] = mcs_fin;
        a.rowstat[plane + arow + qg] = l > 0.f ? __log2f(l) : 0.f;
        const float top = (mx == NEG_INF) ? 0.f : fast_exp2(__builtin_fmaf(mx, c2_q, -mcs_fin));
        a.rowstat[2 * plane + arow + qg] = (l > 0.f && l == top) ? 1.f : 0.f;
        a.rowstat[3 * plane + arow + qg] = c2_q;
        a.rowstat[4 * plane + arow + qg] = c_q;
    }
    float omax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o[0][r] *= out_scale; o[1][r] *= out_scale;
        omax = fmaxf(omax, fmaxf(fabsf(o[0][r]), fabsf(o[1][r])));
    }
    if (a.o_amax != nullptr) amax_publish(qg < a.Tq ? omax : 0.f, a.o_amax, blockIdx.y * gridDim.x + blockIdx.x);
    __syncthreads();                                   // every wave is done with K / V: the scratch aliases them
    wave_store_rows4(o, scratch, a.o + (long)b * a.Tq * a.ldo + h * HD, qw0, a.Tq, a.ldo, lane);
}

// ===================================================================================== backward: dQ (+ delta)
// One workgroup = 128 queries of a (batch, head); K / V tiles stream through the same two-stage ring as in the forward.  Per
// 32-key sub-tile: S^T = K' Q'^T and dP^T = V' dO'^T (row reads), P from the forward's row statistics, dS = P (dP - delta), and
// dQ^T += K'^T dS''^T with K'^T by transposed reads of the SAME K planes; dS'' = dS 2^-e_k(key) carries the key's scale, and the
// lane-local power-of-two pre-scale of attention.hip's fp16x3 backward (attn_h3_track_scale) takes care of its range.
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_img_kernel(AttnImgArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) char xs[2 * FI_STAGE];
    __shared__ float red4[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = reinterpret_cast<float*>(xs) + wave * 32 * KT_LD;
    float* scratch4 = reinterpret_cast<float*>(xs) + wave * 32 * OUT_LD;     // the epilogue's (16-byte rows)

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst = (kend + KB - 1) / KB;
    int wave_kend = kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    const float* ob_ = a.o + (long)b * a.Tq * a.ldo + h * HD;
    const float* gb_ = a.dout + (long)b * a.Tq * a.ldo + h * HD;
    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);

    // dO is a gradient: its pre-scale is the power of two that puts max|dO| (over the whole tensor) in [2^11, 2^12)
    float s_g, inv_g;
    img_tensor_scale(a.do_amax, lane, wave, red4, s_g, inv_g);

    const int qrow = qg < a.Tq ? qg : a.Tq - 1;
    f16x8v qf[4][2], gf[4][2];
    {
        const char* qp = reinterpret_cast<const char*>(a.q) + ((long)(b * a.Tq + qrow) * a.ldq + h * HD) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p) qf[s][p] = *reinterpret_cast<const f16x8v*>(qp + p * 128 + (2 * s + half) * 16);
    }
    wave_stage_tile(ob_, qw0, a.Tq, a.ldo, lane, scratch, 1.f);
    wave_lds_sync();
    float orow[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) orow[j] = scratch[l31 * KT_LD + 2 * j + half];
    wave_lds_sync();
    wave_stage_tile(gb_, qw0, a.Tq, a.ldo, lane, scratch, 1.f);
    wave_lds_sync();
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) delta += scratch[l31 * KT_LD + 2 * j + half] * orow[j];
    delta += __shfl_xor(delta, 32, 64);
    load_lane_frags_h3(scratch, l31, half, s_g, gf);
    // row statistics of this query, as the forward left them
    const long plane = a.stat_plane;
    const float m_q = a.rowstat[arow + qrow], l2_row = a.rowstat[plane + arow + qrow];
    const bool saturated = a.rowstat[2 * plane + arow + qrow] != 0.f;      // one-hot row: dS is the exact zero it is
    // ... which costs nothing per element: the row's log-sum becomes +inf, its recomputed weights exp2(-inf) = 0, and with them dS
    const float l2_q = saturated ? __builtin_inff() : l2_row;
    const float c2_q = a.rowstat[3 * plane + arow + qrow];
    if (half == 0 && qg < a.Tq) {
        a.delta[arow + qg] = saturated ? -0.f : delta;
        // the dK / dV kernel multiplies dS by the row's score multiplier (plane 4): a one-hot row's becomes zero here, and with it
        // that row's dS there -- no test per element (nothing else reads plane 4 after the forward; zeroing it twice is the same)
        if (saturated) a.rowstat[4 * plane + arow + qg] = 0.f;
    }
    const float dp_unscale = inv_g * a.drop_scale;      // dP accumulator units -> true dP (times the key's 2^-e_v), times 1/(1-p)
    float sds = 0.f;                                    // this query's dS pre-scale (power of two), set / lowered on the fly

    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    const u32x4a rsK = make_rsrc(reinterpret_cast<const char*>(a.k) + ((long)AIMG_SB(b) * a.Tk * a.ldk + AIMG_SH(h) * HD) * 4, (uint32_t)a.Tk * (uint32_t)a.ldk * 4u);
    const u32x4a rsV = make_rsrc(reinterpret_cast<const char*>(a.v) + ((long)AIMG_SB(b) * a.Tk * a.ldv + AIMG_SH(h) * HD) * 4, (uint32_t)a.Tk * (uint32_t)a.ldv * 4u);
    const u32x4a rsKi = make_rsrc(a.k_inv + (long)AIMG_SH(h) * a.k_rows + (long)AIMG_SB(b) * a.Tk, (uint32_t)a.Tk * 4u);
    const u32x4a rsVi = make_rsrc(a.v_inv + (long)AIMG_SH(h) * a.k_rows + (long)AIMG_SB(b) * a.Tk, (uint32_t)a.Tk * 4u);
    const uint32_t lds0 = lds_addr_a(xs);
    const int ld_r = lane >> 3;
    const uint32_t ld_c0 = (uint32_t)((lane & 7) ^ isw(ld_r)) * 16u;
    auto issue = [&](int t, int stage) {
        const uint32_t dst = lds0 + (uint32_t)stage * FI_STAGE;
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            int row = t * KB + 16 * wave + 8 * nn + ld_r;
            row = row < a.Tk ? row : a.Tk - 1;
            const uint32_t cb = ld_c0 ^ (uint32_t)(nn * 32);
            const uint32_t ko = (uint32_t)row * (uint32_t)(a.ldk * 4) + cb, vo = (uint32_t)row * (uint32_t)(a.ldv * 4) + cb;
            const uint32_t piece = (uint32_t)(16 * wave + 8 * nn) * 128u;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                dma16a(rsK, ko, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + (uint32_t)p * IMG_PLANE + piece));
                dma16a(rsV, vo, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + IMG_TILE + (uint32_t)p * IMG_PLANE + piece));
            }
        }
        if (wave < 2) {
            int key = t * KB + lane;
            key = key < a.Tk ? key : a.Tk - 1;
            if (wave == 0) dma4a(rsKi, (uint32_t)key * 4u, __builtin_amdgcn_readfirstlane(dst + 2 * IMG_TILE));
            else dma4a(rsVi, (uint32_t)key * 4u, __builtin_amdgcn_readfirstlane(dst + 2 * IMG_TILE + 256));
        }
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };

    const int fsw = isw(l31);
    uint32_t k_off[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) k_off[st] = (uint32_t)(l31 * 128 + (((2 * st + half) ^ fsw) << 4));
    const int q4 = (lane & 15) >> 2, pc = lane & 3, g16 = (lane >> 4) & 1;
    uint32_t kt_off[2][2];             // K'^T by transposed reads of the K planes: [i2][second]
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            const int r = 8 * sc + 4 * half + q4;
            const int ch = 4 * i2 + 2 * g16 + (pc >> 1);
            kt_off[i2][sc] = (uint32_t)(r * 128 + ((ch ^ isw(r)) << 4) + (pc & 1) * 8);
        }

    __syncthreads();                   // the per-wave scratch aliases the stages: every wave is done with it
    if (nst > 0) issue(0, 0);
    landed();
    for (int t = 0; t < nst; ++t) {
        if (t + 1 < nst) issue(t + 1, (t + 1) & 1);
        const char* st_ = xs + (t & 1) * FI_STAGE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = t * KB + sub * 32;
            if (key0 >= wave_kend) break;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8v kf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) kf[p] = *reinterpret_cast<const f16x8v*>(st_ + p * IMG_PLANE + sub * 4096 + k_off[st]);
                mfma_h3(s, kf, qf[st]);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8v vf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) vf[p] = *reinterpret_cast<const f16x8v*>(st_ + IMG_TILE + p * IMG_PLANE + sub * 4096 + k_off[st]);
                mfma_h3(dp, vf, gf[st]);
            }
            const float* ki = reinterpret_cast<const float*>(st_ + 2 * IMG_TILE) + sub * 32 + 4 * half;
            float kiv[16], viv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 f = *reinterpret_cast<const float4*>(ki + 8 * g4), g = *reinterpret_cast<const float4*>(ki + 64 + 8 * g4);
                kiv[4 * g4] = f.x; kiv[4 * g4 + 1] = f.y; kiv[4 * g4 + 2] = f.z; kiv[4 * g4 + 3] = f.w;
                viv[4 * g4] = g.x; viv[4 * g4 + 1] = g.y; viv[4 * g4 + 2] = g.z; viv[4 * g4 + 3] = g.w;
            }
            float ds[16];
            const bool full = (key0 + 32 <= klen) && (!CAUSAL || key0 + 31 <= qw0);
            // (no run-time test around a single element: hipcc turns `if (a.thr != 0u)` inside the unrolled loop into a scalar
            // branch PER ELEMENT -- 41 branches per sub-tile in the ISA of round 5's first build, every one a basic-block
            // boundary the MFMAs cannot be scheduled across.  thr16 == 0 keeps every weight, so the dropout arithmetic is
            // unconditional; the key / causal masks are a property of the whole sub-tile.)
            float pw[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)    // the forward's own exponent: fma(s' 2^-e_k, c2, -mcs), on bit-identical accumulators s'
                pw[r] = fast_exp2(__builtin_fmaf(s[r] * kiv[r], c2_q, -m_q) - l2_q);
            if (!full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kg = key0 + acc_row(r, half);
                    pw[r] = (kg < klen && (!CAUSAL || kg <= qg)) ? pw[r] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int key_g = key0 + acc_row(r, half);
                const uint32_t qh = attn_quad_hash(seed_eff, rowid, (uint32_t)key_g >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = dp[r + e] * (viv[r + e] * dp_unscale);
                    g = attn_keep_word(qh, attn_drop_mult(e), thr16) ? g : 0.f;
                    // dS carries the key's 2^-e_k from here on: the K'^T it meets below is K 2^e_k
                    ds[r + e] = pw[r + e] * (g - delta) * kiv[r + e];
                }
            }
            attn_h3_track_scale(ds, sds, dq);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = ds[8 * t2 + e];
                f16x8v dsf[2];
                split_frag8_h3(x, sds, dsf[0], dsf[1]);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    f16x8v ktf[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const char* kb = st_ + pl * IMG_PLANE + sub * 4096 + t2 * 2048;
                        ktf[pl] = join_tr(lds_tr4(kb + kt_off[i2][0]), lds_tr4(kb + kt_off[i2][1]));
                    }
                    mfma_h3(dq[i2], ktf, dsf);
                }
            }
        }
        landed();
    }
    {
        const float fin = (sds > 0.f) ? a.qscale / sds : 0.f;    // accumulator: K^T (dS sds) -> dQ (the 1 / sqrt(d) of q's pre-scale)
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dq[0][r] *= fin; dq[1][r] *= fin;
            mx = fmaxf(mx, fmaxf(fabsf(dq[0][r]), fabsf(dq[1][r])));
        }
        if (a.amax_dq != nullptr) amax_publish(qg < a.Tq ? mx : 0.f, a.amax_dq, blockIdx.y * gridDim.x + blockIdx.x);
    }
    wave_store_rows4(dq, scratch4, a.dq + (long)b * a.Tq * a.lddq + h * HD, qw0, a.Tq, a.lddq, lane);
}

// ===================================================================================== backward: dK, dV
// One workgroup = 128 keys of a (batch, head), key on the lane (K', V' fragments lane-resident, straight from the image); stages
// of 32 queries stream through a two-deep ring: Q' planes by LDS-DMA from the image, dO as raw fp32 rows by LDS-DMA into the very
// bytes its two planes occupy (a row's first 128 bytes land in its "hi" row, the rest in its "lo" row) and split IN PLACE by
// the wave that requested them (as gemm_h3i's convert_rows) -- under the other waves' products, with one barrier per stage.
// S = Q' K'^T and dP = dO' V'^T read the planes by rows; dV^T += dO'^T P and dK^T += Q'^T dS'' read the same planes transposed.
constexpr int DKI_QS = 32;                          // queries per stage
constexpr int DKI_PLANE = DKI_QS * 128;             // 4 KB
constexpr int DKI_STAGE = 4 * DKI_PLANE + 5 * 256;  // Q hi, Q lo, dO hi, dO lo, five row-statistic arrays of 64 floats (32 used)
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_img_kernel(AttnImgArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) char xs[(2 * DKI_STAGE > OUT_BYTES) ? 2 * DKI_STAGE : OUT_BYTES];
    __shared__ float red4[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int kblk = blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int k0 = kblk * QB, kw0 = k0 + wave * 32;
    const int kg = kw0 + l31;
    const uint32_t key_mult = attn_drop_mult((uint32_t)kg);
    float* scratch4 = reinterpret_cast<float*>(xs) + wave * 32 * OUT_LD;     // the epilogue's scratch (aliases the stages)

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    const long arow = ((long)(b * a.H + h) * a.Tq);

    float s_g, inv_g;
    img_tensor_scale(a.do_amax, lane, wave, red4, s_g, inv_g);

    // K', V' fragments of this lane's key, and its two scales
    const int krow = kg < a.Tk ? kg : a.Tk - 1;
    f16x8v kf[4][2], vf[4][2];
    {
        const char* kp = reinterpret_cast<const char*>(a.k) + ((long)(b * a.Tk + krow) * a.ldk + h * HD) * 4;
        const char* vp = reinterpret_cast<const char*>(a.v) + ((long)(b * a.Tk + krow) * a.ldv + h * HD) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                kf[s][p] = *reinterpret_cast<const f16x8v*>(kp + p * 128 + (2 * s + half) * 16);
                vf[s][p] = *reinterpret_cast<const f16x8v*>(vp + p * 128 + (2 * s + half) * 16);
            }
    }
    const float kinv = a.k_inv[(long)h * a.k_rows + (long)b * a.Tk + krow];
    const float vinv = a.v_inv[(long)h * a.k_rows + (long)b * a.Tk + krow];

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

    const float dp_unscale = inv_g * vinv * a.drop_scale;       // dP accumulator units -> true dP, times the 1/(1-p) of kept weights
    float sds = 0.f;
    // (query stages of this workgroup: all of them, or the z-th of gridDim.z equal ranges)
    const int nqs_all = (a.Tq + DKI_QS - 1) / DKI_QS;
    const int per_z = (nqs_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int nqs = min(nqs_all, ((int)blockIdx.z + 1) * per_z);
    int qs_begin = max(CAUSAL ? (k0 / DKI_QS) : 0, (int)blockIdx.z * per_z);
    if (k0 >= klen || qs_begin > nqs) qs_begin = nqs;

    // ---- the ring.  This wave moves rows 8 w .. 8 w + 7 of a stage: one 1-KB piece per Q plane (source chunks swizzled) and two
    // raw pieces of dO; waves 0 / 1 also the row statistics.  Rows past Tq are clamped (finite data; masked below).
    const u32x4a rsQ = make_rsrc(reinterpret_cast<const char*>(a.q) + ((long)AIMG_SB(b) * a.Tq * a.ldq + AIMG_SH(h) * HD) * 4, (uint32_t)a.Tq * (uint32_t)a.ldq * 4u);
    const u32x4a rsG = make_rsrc(a.dout + (long)AIMG_SB(b) * a.Tq * a.ldo + AIMG_SH(h) * HD, (uint32_t)a.Tq * (uint32_t)a.ldo * 4u);
    const long plane = a.stat_plane;
    const u32x4a rsS0 = make_rsrc(a.rowstat + arow, (uint32_t)a.Tq * 4u), rsS1 = make_rsrc(a.rowstat + plane + arow, (uint32_t)a.Tq * 4u);
    const u32x4a rsS3 = make_rsrc(a.rowstat + 3 * plane + arow, (uint32_t)a.Tq * 4u), rsS4 = make_rsrc(a.rowstat + 4 * plane + arow, (uint32_t)a.Tq * 4u);
    const u32x4a rsD = make_rsrc(a.delta + arow, (uint32_t)a.Tq * 4u);
    const uint32_t lds0 = lds_addr_a(xs);
    const int ld_r = lane >> 3;
    const uint32_t ld_cq = (uint32_t)((lane & 7) ^ isw(8 * (wave & 1) + ld_r)) * 16u;    // row 8 w + ld_r: isw sees w only through bit 3
    auto issue = [&](int qt0, int stage) {
        const uint32_t dst = lds0 + (uint32_t)stage * DKI_STAGE;
        int row = qt0 + 8 * wave + ld_r;
        row = row < a.Tq ? row : a.Tq - 1;
        const uint32_t qo = (uint32_t)row * (uint32_t)(a.ldq * 4) + ld_cq;
        const uint32_t go = (uint32_t)row * (uint32_t)(a.ldo * 4) + (uint32_t)(lane & 7) * 16u;
        const uint32_t piece = (uint32_t)(8 * wave) * 128u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            dma16a(rsQ, qo, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + (uint32_t)p * DKI_PLANE + piece));
            dma16a(rsG, go, (uint32_t)p * 128u, __builtin_amdgcn_readfirstlane(dst + (uint32_t)(2 + p) * DKI_PLANE + piece));
        }
        if (wave < 2) {
            int q = qt0 + l31;
            q = q < a.Tq ? q : a.Tq - 1;
            const uint32_t so = (uint32_t)q * 4u, sd = dst + 4 * DKI_PLANE;
            if (wave == 0) {
                dma4a(rsS0, so, __builtin_amdgcn_readfirstlane(sd));
                dma4a(rsS1, so, __builtin_amdgcn_readfirstlane(sd + 256));
                dma4a(rsD, so, __builtin_amdgcn_readfirstlane(sd + 512));
            } else {
                dma4a(rsS3, so, __builtin_amdgcn_readfirstlane(sd + 768));
                dma4a(rsS4, so, __builtin_amdgcn_readfirstlane(sd + 1024));
            }
        }
    };
    // this wave's own raw dO rows -> f16 planes of dO * s_g, in place: lane -> (row 8 w + lane / 8, columns 8 c .. 8 c + 7, c = lane & 7)
    auto convert = [&](int stage) {
        char* st_ = xs + stage * DKI_STAGE;
        const int row = 8 * wave + ld_r, c = lane & 7;
        const char* src = st_ + (2 + (c >> 2)) * DKI_PLANE + row * 128 + (c & 3) * 32;
        const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every lane has its raw values before any lane overwrites the rows
        __builtin_amdgcn_wave_barrier();
        u32x4a hi, lo;
        uint32_t hh, ll;
        split2_pair_h(f32x2{x0.x, x0.y} * s_g, hh, ll); hi.x = hh; lo.x = ll;
        split2_pair_h(f32x2{x0.z, x0.w} * s_g, hh, ll); hi.y = hh; lo.y = ll;
        split2_pair_h(f32x2{x1.x, x1.y} * s_g, hh, ll); hi.z = hh; lo.z = ll;
        split2_pair_h(f32x2{x1.z, x1.w} * s_g, hh, ll); hi.w = hh; lo.w = ll;
        const int off = row * 128 + ((c ^ isw(row)) << 4);
        *reinterpret_cast<u32x4a*>(st_ + 2 * DKI_PLANE + off) = hi;
        *reinterpret_cast<u32x4a*>(st_ + 3 * DKI_PLANE + off) = lo;
    };

    const int fsw = isw(l31);
    uint32_t r_off[4];                 // row l31 of a stage plane, chunk 2 st + half
#pragma unroll
    for (int st = 0; st < 4; ++st) r_off[st] = (uint32_t)(l31 * 128 + (((2 * st + half) ^ fsw) << 4));
    const int q4 = (lane & 15) >> 2, pc = lane & 3, g16 = (lane >> 4) & 1;
    uint32_t t_off[2][2];              // transposed reads: [i2][second] of the 16-query step t2 = 0; t2 = 1 is 2 KB further (isw sees
#pragma unroll                         // rows 16 apart alike)
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            const int r = 8 * sc + 4 * half + q4;
            const int ch = 4 * i2 + 2 * g16 + (pc >> 1);
            t_off[i2][sc] = (uint32_t)(r * 128 + ((ch ^ isw(r)) << 4) + (pc & 1) * 8);
        }

    if (qs_begin < nqs) {
        issue(qs_begin * DKI_QS, qs_begin & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert(qs_begin & 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int qs = qs_begin; qs < nqs; ++qs) {
        const int qt0 = qs * DKI_QS;
        if (qs + 1 < nqs) issue(qt0 + DKI_QS, (qs + 1) & 1);       // in flight while this stage is multiplied
        const char* st_ = xs + (qs & 1) * DKI_STAGE;
        const bool skip = (CAUSAL && qt0 + 31 < kw0) || kw0 >= klen;    // every query precedes this wave's keys / all padding (wave-uniform)
        if (!skip) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8v qfr[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) qfr[p] = *reinterpret_cast<const f16x8v*>(st_ + p * DKI_PLANE + r_off[st]);
                // the same three products in the same order as the forward formed them (there K was the first operand): the
                // accumulators are bit-identical to the forward's
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr[0], kf[st][1], s, 0, 0, 0);      // q_hi k_lo
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr[1], kf[st][0], s, 0, 0, 0);      // q_lo k_hi
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr[0], kf[st][0], s, 0, 0, 0);
            }
            const bool full = (kw0 + 32 <= klen) && (!CAUSAL || kw0 + 31 <= qt0) && (qt0 + DKI_QS <= a.Tq);   // wave-uniform
            const float* stat = reinterpret_cast<const float*>(st_ + 4 * DKI_PLANE);
            float pd[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {         // registers 4 g4 .. 4 g4 + 3 are four consecutive query rows: 16-byte statistic reads
                const float4 m4 = *reinterpret_cast<const float4*>(stat + 8 * g4 + 4 * half);
                const float4 l4 = *reinterpret_cast<const float4*>(stat + 64 + 8 * g4 + 4 * half);
                const float4 e4 = *reinterpret_cast<const float4*>(stat + 192 + 8 * g4 + 4 * half);
                const float mq4[4] = {m4.x, m4.y, m4.z, m4.w}, l24[4] = {l4.x, l4.y, l4.z, l4.w}, c24[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const int q_g = qt0 + acc_row(r, half);
                    pd[r] = fast_exp2(__builtin_fmaf(s[r] * kinv, c24[e], -mq4[e]) - l24[e]);
                }
            }
            if (!full) {                                 // (one test per sub-tile, not one per element: see the dQ kernel)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q_g = qt0 + acc_row(r, half);
                    pd[r] = (kg < klen && (!CAUSAL || kg <= q_g) && q_g < a.Tq) ? pd[r] : 0.f;
                }
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8v gfr[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) gfr[p] = *reinterpret_cast<const f16x8v*>(st_ + (2 + p) * DKI_PLANE + r_off[st]);
                mfma_h3(dp, gfr, vf[st]);
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                uint32_t hq[4];
                {                        // a quad of lanes (keys 4 j .. 4 j + 3) shares one hash word per query row (attention.hip);
                    //                      unconditional: thr16 == 0 keeps every weight (see the dQ kernel)
                    const int rr = r + (lane & 3);
                    const uint32_t mine = attn_quad_hash(seed_eff, (uint32_t)(arow + qt0 + acc_row(rr, half)), (uint32_t)kg >> 2);
                    hq[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x00, 0xF, 0xF, true);
                    hq[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x55, 0xF, 0xF, true);
                    hq[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xAA, 0xF, 0xF, true);
                    hq[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xFF, 0xF, 0xF, true);
                }
                const float4 d4 = *reinterpret_cast<const float4*>(stat + 128 + 2 * r + 4 * half);
                const float4 f4 = *reinterpret_cast<const float4*>(stat + 256 + 2 * r + 4 * half);
                const float dl4[4] = {d4.x, d4.y, d4.z, d4.w}, cq4[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = dp[r + e] * dp_unscale;
                    float pk = pd[r + e];
                    {
                        const bool keep = attn_keep_word(hq[e], key_mult, thr16);
                        g = keep ? g : 0.f;
                        pk = keep ? pk : 0.f;
                    }
                    // (one-hot row: exact zero -- the dQ kernel zeroed its multiplier cq and left delta = -0.0); dS carries the query's
                    // 2^-e_q / sqrt(d) from here on: the Q'^T it meets below is Q 2^e_q
                    ds[r + e] = pd[r + e] * (g - dl4[e]) * cq4[e];
                    pd[r + e] = pk;
                }
            }
            // dV^T[d][key] += dO'^T[d][q] P[q][key],  dK^T[d][key] += Q'^T[d][q] dS''[q][key]
            attn_h3_track_scale(ds, sds, dk);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                float x[8];
                f16x8v f[2];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = pd[8 * t2 + e];
                split_frag8_h3(x, H3A_P, f[0], f[1]);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    f16x8v af[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const char* gb = st_ + (2 + pl) * DKI_PLANE + t2 * 2048;
                        af[pl] = join_tr(lds_tr4(gb + t_off[i2][0]), lds_tr4(gb + t_off[i2][1]));
                    }
                    mfma_h3(dv[i2], af, f);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = ds[8 * t2 + e];
                split_frag8_h3(x, sds, f[0], f[1]);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    f16x8v af[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const char* qb = st_ + pl * DKI_PLANE + t2 * 2048;
                        af[pl] = join_tr(lds_tr4(qb + t_off[i2][0]), lds_tr4(qb + t_off[i2][1]));
                    }
                    mfma_h3(dk[i2], af, f);
                }
            }
        }
        // the next stage: this wave's own pieces have landed -> its dO rows are split in place; then everybody meets
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (qs + 1 < nqs) convert((qs + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    {
        const float fk = (sds > 0.f) ? 1.f / sds : 0.f;                 // dk accumulator: Q^T (dS / sqrt(d) * sds)
        const float fv = inv_g / H3A_P * a.drop_scale;                    // dv accumulator: (dO * s_g)^T (kept P * 2^10) / (1-p)
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[0][r] *= fk; dk[1][r] *= fk; dv[0][r] *= fv; dv[1][r] *= fv;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(dk[0][r]), fabsf(dk[1][r]))), fmaxf(fabsf(dv[0][r]), fabsf(dv[1][r])));
        }
        if (a.amax_dkv != nullptr && gridDim.z == 1) amax_publish(kg < a.Tk ? mx : 0.f, a.amax_dkv, blockIdx.y * gridDim.x + blockIdx.x);
    }
    if (gridDim.z == 1) {
        wave_store_rows4(dk, scratch4, a.dk + (long)b * a.Tk * a.lddk + h * HD, kw0, a.Tk, a.lddk, lane);
        wave_store_rows4(dv, scratch4, a.dv + (long)b * a.Tk * a.lddv + h * HD, kw0, a.Tk, a.lddv, lane);
    } else {            // partial sums: packed rows of (dk | dv), 2 H 64 floats each
        float* part = a.dkv_part + (long)blockIdx.z * a.part_stride + (long)b * a.Tk * a.ldp + h * HD;
        wave_store_rows4(dk, scratch4, part, kw0, a.Tk, a.ldp, lane);
        wave_store_rows4(dv, scratch4, part + a.H * HD, kw0, a.Tk, a.ldp, lane);
    }
}

// out[row][c] = part[0][row][c] + part[1][row][c] + ... (fixed order), 16 bytes per lane; the partial sums are packed rows of
// cols4 float4s, the output has a row stride of its own (a layer's window of the stacked K/V gradient); max|out| published
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n4, int zs,
                                                              long stride, float* __restrict__ amax, int cols4, long out_ld) {
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 acc = reinterpret_cast<const float4*>(part)[i];
        for (int z = 1; z < zs; ++z) {
            const float4 v = reinterpret_cast<const float4*>(part + (long)z * stride)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const long row = i / cols4;
        *reinterpret_cast<float4*>(out + row * out_ld + (i - row * cols4) * 4) = acc;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    }
    if (amax != nullptr) amax_publish(mx, amax, blockIdx.x * 4 + (threadIdx.x >> 6));
}

#ifndef TTTS_AIMG_KT
#define TTTS_AIMG_KT 32
#endif

static int check_img(const char* name, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, float drop_p) {
    TTTS_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "%s: bad dims", name);
    TTTS_REQUIRE((long)B * H < (1L << 31) && cdiv(Tq, QB) <= 65535 && cdiv(Tk, QB) <= 65535, "%s: grid too large", name);
    TTTS_REQUIRE(ldq >= H * HD && ldk >= H * HD && ldv >= H * HD && ldo >= H * HD, "%s: row strides must be >= H*64", name);
    TTTS_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "%s: row strides must be multiples of 4", name);
    TTTS_REQUIRE((uint64_t)Tq * ldq * 4 < (1ull << 32) && (uint64_t)Tk * ldk * 4 < (1ull << 32) && (uint64_t)Tk * ldv * 4 < (1ull << 32),
                 "%s: one utterance's operand exceeds 4 GiB", name);
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "%s: bad dropout p", name);
    return TTTS_OK;
}

}  // namespace ttts

using namespace ttts;

#ifdef TTTS_AIMG_STAMPS
extern "C" int ttts_dbg_aimg_read_stamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ttts::ttts_aimg_stamps), n * sizeof(unsigned long long));
}
#endif

/* Scaled dot-product attention forward on head-image operands (ttts_linear_fwd_h3d_img / ttts_head_image): replaces the same
 * call sites as ttts_attention_fwd_h3 (torch F.scaled_dot_product_attention inside nn.MultiheadAttention,
 * torch/nn/functional.py:6576-6629, reached from model/layers.py:54-74 and torch _sa_block).  q / k / v point at head 0 of
 * their section inside the image (row strides ld* in 4-byte cells), *_inv at the [head][rows] inverse scales of that section
 * (q_inv_rows / k_inv_rows rows per head plane: 0 = B * Tq / B * Tk, more when the batch is the first part of a larger image;
 * stat_plane: floats per plane of rowstat_out, 0 = B * H * Tq, likewise); v_amax: TTTS_AMAX_SLOTS partial maxima of |v|.  Outputs as
 * ttts_attention_fwd_h3; rowstat_out has FIVE planes (B, H, Tq) -- the backward on images needs them all. */
extern "C" int ttts_attention_fwd_img(const void* q, const void* k, const void* v, const float* q_inv, const float* k_inv,
                                      const float* v_inv, float* o, float* lse, float* attn, const int64_t* key_lens, int B, int H,
                                      int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int causal, float q_scale, float drop_p,
                                      uint64_t seed, const uint64_t* step_seed, const float* v_amax, float* o_amax_out,
                                      float* rowstat_out, int64_t q_inv_rows, int64_t k_inv_rows, int64_t stat_plane, void* stream) {
    TTTS_REQUIRE(q && k && v && q_inv && k_inv && v_inv && o && key_lens && v_amax, "attention_fwd_img: null pointer");
    int rc = check_img("attention_fwd_img", B, H, Tq, Tk, ldq, ldk, ldv, ldo, drop_p);
    if (rc) return rc;
    TTTS_REQUIRE(!(causal && attn), "attention_fwd_img: weights output is only for non-causal (cross) attention");
    TTTS_REQUIRE(!causal || Tq == Tk, "attention_fwd_img: causal form needs Tq == Tk");
    TTTS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)v_amax) & 15) == 0,
                 "attention_fwd_img: q/k/v/o/v_amax must be 16-byte aligned");
    AttnImgArgs a = {};
    a.q = q; a.k = k; a.v = v; a.q_inv = q_inv; a.k_inv = k_inv; a.v_inv = v_inv;
    TTTS_REQUIRE((q_inv_rows == 0 || q_inv_rows >= (int64_t)B * Tq) && (k_inv_rows == 0 || k_inv_rows >= (int64_t)B * Tk) &&
                 (stat_plane == 0 || stat_plane >= (int64_t)B * H * Tq), "attention_fwd_img: plane strides smaller than the batch");
    a.q_rows = q_inv_rows ? q_inv_rows : (long)B * Tq; a.k_rows = k_inv_rows ? k_inv_rows : (long)B * Tk;
    a.stat_plane = stat_plane ? stat_plane : (long)B * H * Tq;
    a.o = o; a.lse = lse; a.attn = attn; a.key_lens = key_lens;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    a.drop_scale = 1.f / (1.f - drop_p);
    a.qscale = q_scale;
    a.seed = seed; a.step_seed = step_seed;
    a.v_amax = v_amax; a.o_amax = o_amax_out; a.rowstat = rowstat_out;
    dim3 grid(B * H, cdiv(Tq, QB), 1);
    if (causal)
        hipLaunchKernelGGL((attn_fwd_img_kernel<true, false, TTTS_AIMG_KT>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (!attn)
        hipLaunchKernelGGL((attn_fwd_img_kernel<false, false, TTTS_AIMG_KT>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (Tk <= 128)
        hipLaunchKernelGGL(attn_fwd_img_maps_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((attn_fwd_img_kernel<false, true, 64>), grid, dim3(256), 0, (hipStream_t)stream, a);
    TTTS_LAUNCH_CHECK("attn_fwd_img_kernel");
    return TTTS_OK;
}

/* dq, dk, dv (fp32, packed or separate: strides ldd*) from d_o on head-image operands; o, d_o fp32 as the forward wrote / the
 * out-projection's data gradient left them; rowstat = the five planes ttts_attention_fwd_img wrote; do_amax = partial maxima
 * of |d_o|; delta (B,H,Tq) is scratch; plane 4 of rowstat (the rows' score multipliers) is ZEROED for one-hot rows by this call (the
 * dK / dV kernel reads it after the dQ kernel; nothing else does); dq_amax_out / dkv_amax_out: NULL, or zeroed TTTS_AMAX_SLOTS floats; dkv_partials / q_splits:
 * NULL / 1, or a workspace of q_splits x B x Tk x 2 H 64 floats: the dK / dV kernel then splits the QUERY range over q_splits
 * workgroups per key block and a fixed-order reduction adds the partial sums (cross-attention's one key block per (batch, head)
 * otherwise leaves three quarters of the chip's wave slots empty; causal self-attention at small batches is as long as its first
 * key block's walk over every query); dv = dk + H 64 with one row stride >= 2 H 64 (a
 * layer's window of a wider gradient tensor: the fused K/V projection of all decoder layers).  Replaces the same
 * call sites as ttts_attention_bwd_h3 (autograd of F.scaled_dot_product_attention / the explicit softmax path,
 * torch/nn/functional.py:6576-6629). */
extern "C" int ttts_attention_bwd_img(const void* q, const void* k, const void* v, const float* q_inv, const float* k_inv,
                                      const float* v_inv, const float* o, const float* d_o, const float* rowstat, float* delta,
                                      float* dq, float* dk, float* dv, const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq,
                                      int ldk, int ldv, int ldo, int lddq, int lddk, int lddv, int causal, float q_scale,
                                      float drop_p, uint64_t seed, const uint64_t* step_seed, const float* do_amax,
                                      float* dq_amax_out, float* dkv_amax_out, float* dkv_partials, int q_splits,
                                      int64_t q_inv_rows, int64_t k_inv_rows, int64_t stat_plane, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(q && k && v && q_inv && k_inv && v_inv && o && d_o && rowstat && delta && dq && dk && dv && key_lens && do_amax,
                 "attention_bwd_img: null pointer");
    int rc = check_img("attention_bwd_img", B, H, Tq, Tk, ldq, ldk, ldv, ldo, drop_p);
    if (rc) return rc;
    TTTS_REQUIRE(lddq >= H * HD && lddk >= H * HD && lddv >= H * HD, "attention_bwd_img: gradient strides must be >= H*64");
    TTTS_REQUIRE(!causal || Tq == Tk, "attention_bwd_img: causal form needs Tq == Tk");
    TTTS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)do_amax) & 15) == 0,
                 "attention_bwd_img: q/k/v/o/d_o/do_amax must be 16-byte aligned");
    TTTS_REQUIRE((uint64_t)Tq * ldo * 4 < (1ull << 32), "attention_bwd_img: one utterance's d_o exceeds 4 GiB");
    AttnImgArgs a = {};
    a.q = q; a.k = k; a.v = v; a.q_inv = q_inv; a.k_inv = k_inv; a.v_inv = v_inv;
    TTTS_REQUIRE((q_inv_rows == 0 || q_inv_rows >= (int64_t)B * Tq) && (k_inv_rows == 0 || k_inv_rows >= (int64_t)B * Tk) &&
                 (stat_plane == 0 || stat_plane >= (int64_t)B * H * Tq), "attention_bwd_img: plane strides smaller than the batch");
    a.q_rows = q_inv_rows ? q_inv_rows : (long)B * Tq; a.k_rows = k_inv_rows ? k_inv_rows : (long)B * Tk;
    a.stat_plane = stat_plane ? stat_plane : (long)B * H * Tq;
    a.o = const_cast<float*>(o); a.dout = d_o; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.key_lens = key_lens;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    a.drop_scale = 1.f / (1.f - drop_p);
    a.qscale = q_scale;
    a.seed = seed; a.step_seed = step_seed;
    a.do_amax = do_amax; a.amax_dq = dq_amax_out; a.amax_dkv = dkv_amax_out;
    a.rowstat = const_cast<float*>(rowstat);
    // q_splits > 1: dk / dv as partial sums over q_splits query ranges in dkv_partials (q_splits x B x Tk x 2 H 64 floats), then
    // one reduction into rows that hold dk then dv
    TTTS_REQUIRE(q_splits >= 1 && q_splits <= 16, "attention_bwd_img: q_splits out of 1..16");
    if (q_splits > 1) {
        TTTS_REQUIRE(dkv_partials && dv == dk + H * HD && lddk >= 2 * H * HD && lddv == lddk,
                     "attention_bwd_img: query splits need a workspace and a (dk | dv) gradient whose rows hold dk then dv");
        a.ldp = 2 * H * HD;
        a.dkv_part = dkv_partials; a.part_stride = (long)B * Tk * a.ldp;
    }
    dim3 gq(B * H, cdiv(Tq, QB), 1), gk(B * H, cdiv(Tk, QB), q_splits);
    if (causal) {
        hipLaunchKernelGGL((attn_bwd_dq_img_kernel<true>), gq, dim3(256), 0, stream, a);
        TTTS_LAUNCH_CHECK("attn_bwd_dq_img_kernel");
        hipLaunchKernelGGL((attn_bwd_dkv_img_kernel<true>), gk, dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_img_kernel<false>), gq, dim3(256), 0, stream, a);
        TTTS_LAUNCH_CHECK("attn_bwd_dq_img_kernel");
        hipLaunchKernelGGL((attn_bwd_dkv_img_kernel<false>), gk, dim3(256), 0, stream, a);
    }
    TTTS_LAUNCH_CHECK("attn_bwd_dkv_img_kernel");
    if (q_splits > 1) {
        const long n4 = a.part_stride / 4;
        const long blocks = (n4 + 255) / 256;
        hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, dkv_partials, dk, n4,
                           q_splits, a.part_stride, dkv_amax_out, a.ldp / 4, (long)lddk);
        TTTS_LAUNCH_CHECK("attn_dkv_reduce_kernel");
    }
    return TTTS_OK;
}
