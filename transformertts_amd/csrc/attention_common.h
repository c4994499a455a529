// Pieces shared by the attention translation units (attention.hip: fp32 operands; attention_img.hip: head-image operands).
#pragma once
#include "ttts_common.h"

namespace ttts {

constexpr int HD = 64;            // head dim
constexpr int KT_LD = HD + 1;     // LDS row stride (odd: conflict-free "row per lane" reads)
constexpr int QB = 128;           // rows per workgroup (4 waves x 32)
constexpr float NEG_INF = -__builtin_inff();
// v_exp_f32 as it is: exp2f() wraps it in a range reduction for results below 2^-126 (compare, select, add, ldexp: four more
// instructions per element in kernels whose speed is set by their VALU count); such weights are zero at fp32 anyway
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// Row loads go through a raw buffer descriptor spanning `nrows` rows of the (batch, head) slice: a row index past the
// end gives an offset outside the descriptor and the hardware returns zeros.  A predicated global load ("if (row <
// nrows) v = *p") instead compiles to branch + load + s_waitcnt vmcnt(0) and serialises the loads of a stage.
struct RowSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    int ld;
};
__device__ __forceinline__ RowSrc row_src(const float* base, long nrows, int ld) {
    RowSrc r;
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (uint32_t)(nrows * ld * 4), 0x00020000);
    r.ld = ld;
    return r;
}
__device__ __forceinline__ float4 row_load4(const RowSrc& s, long row, int c4) {
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (int)(uint32_t)((row * s.ld + c4 * 4) * 4), 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

struct AttnArgs {
    const float* q; const float* k; const float* v;
    float* o; float* lse; float* attn;
    const float* dout; float* delta; float* dq; float* dk; float* dv;
    const int64_t* key_lens;
    int B, H, Tq, Tk;
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float qscale;                             // q is multiplied by this before q.k^T: sqrt(1 / head_dim) (torch/nn/functional.py:6578)
    float drop_scale; uint32_t thr; uint64_t seed; const uint64_t* step_seed;
    const float* do_amax; int do_amax_n;      // fp16x3 backward: partial maxima of |dout| (ttts_amax_partials)
    float* amax_dq; float* amax_dkv;          // fp16x3 backward: NULL, or caller-zeroed TTTS_AMAX_SLOTS-slot arrays receiving max|dq| / max|dk, dv|
    // fp16x3 forms: TTTS_AMAX_SLOTS partial maxima of |q|, |k|, |v| each (the same array three times for a packed projection output):
    // the operands' dynamic pre-scales.  o_amax (forward): NULL, or a caller-zeroed TTTS_AMAX_SLOTS-slot array receiving max|o|.
    const float* q_amax; const float* k_amax; const float* v_amax;
    float* o_amax;
    // fp16x3 forms: per-row softmax statistics in the forward's own units, (2, B, H, Tq): plane 0 = the subtrahend mcs of the
    // weights' exponents fma(s, c2, -mcs) exactly as the forward used it with the final row maximum (one rounded product
    // per row), plane 1 = log2 of the row sum of those weights.  The backward re-forms bit-identical score accumulators s and
    // the same exponents, so its probabilities ARE the forward's whatever the scores' magnitude; from lse (one float,
    // natural units) they are only good to ulp(lse): 6 % at scores of 1e6.
    float* rowstat;
};


constexpr int KB = 64;   // rows staged per barrier pair (two 32-row MFMA sub-tiles)

// one wave stages its own 32 x 64 tile (rows beyond nrows_total -> 0) into `dst` (stride KT_LD)
__device__ __forceinline__ void wave_stage_tile(const float* base, long row0, long nrows_total, int ld, int lane,
                                                float* dst, float scale) {
    const RowSrc src = row_src(base, nrows_total, ld);
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int row = (lane >> 4) + 4 * i, c4 = lane & 15;
        v[i] = row_load4(src, row0 + row, c4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int row = (lane >> 4) + 4 * i, c4 = lane & 15;
        float* d = dst + row * KT_LD + c4 * 4;
        d[0] = v[i].x * scale; d[1] = v[i].y * scale; d[2] = v[i].z * scale; d[3] = v[i].w * scale;
    }
}
__device__ __forceinline__ void wave_lds_sync() {
    // LDS traffic of one wave: make the writes above visible to the reads below (same wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// write a 2 x (32x32) accumulator pair holding X^T[d][row] (row on the lane) as rows of 64 floats
__device__ __forceinline__ void wave_store_rows(const f32x16 (&acc)[2], float* scratch, float* gbase, long row0,
                                                long nrows_total, int ld, int lane, float scale) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[l31 * KT_LD + blk * 32 + acc_row(r, half)] = acc[blk][r] * scale;
    wave_lds_sync();
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        float v = scratch[i * KT_LD + lane];
        if (row0 + i < nrows_total) gbase[(row0 + i) * ld + lane] = v;
    }
    wave_lds_sync();
}

// LDS budget shared by the three kernels: two staged 64-row tiles, re-used as per-wave 32x65 scratch in the
// prologue / epilogue (4 waves x 8320 B = 33280 B)
constexpr int SMEM_FLOATS = 4 * 32 * KT_LD;
static_assert(2 * KB * KT_LD <= SMEM_FLOATS, "staging buffers must fit the shared scratch");


typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
constexpr float H3A_P = 1024.0f;

__device__ __forceinline__ void split2_pair_h(f32x2 x, uint32_t& hi, uint32_t& lo) {
    const f16x2v h = __builtin_convertvector(x, f16x2v);
    const f32x2 r = x - __builtin_convertvector(h, f32x2);          // exact
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2v));
}

// 8 fp32 -> two f16x8 fragments of x * scale
__device__ __forceinline__ void split_frag8_h3(const float (&x)[8], float scale, f16x8v& hi, f16x8v& lo) {
    u32x4v h, l;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        uint32_t a, b;
        split2_pair_h(f32x2{x[2 * u], x[2 * u + 1]} * scale, a, b);
        h[u] = a; l[u] = b;
    }
    hi = __builtin_bit_cast(f16x8v, h);
    lo = __builtin_bit_cast(f16x8v, l);
}
// c += a (hi,lo) x b (hi,lo), three products, smallest terms first
__device__ __forceinline__ void mfma_h3(f32x16& c, const f16x8v (&a)[2], const f16x8v (&b)[2]) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
}


// ds: this lane's 16 dS values of the tile (true units); sds: the lane's current pre-scale (0 = unset); acc: the
// accumulator pair the products land in (lane-local column).  Both half-waves hold halves of the same column.
__device__ __forceinline__ void attn_h3_track_scale(const float (&ds)[16], float& sds, f32x16 (&acc)[2]) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(ds[r]));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const uint32_t e = (__float_as_uint(m) >> 23) & 0xffu;
    const bool change = e >= 24u && e < 255u && (sds == 0.f || m * sds > 8192.f);
    if (__any(change)) {
        const float snew = change ? __uint_as_float((265u - e) << 23) : sds;
        const float f = (change && sds != 0.f) ? snew / sds : 1.f;        // a power of two <= 2^-1 when it applies
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] *= f; acc[1][r] *= f; }
        sds = snew;
    }
}

// lane-resident B operand: the 64 values of this lane's row (query or key) x scale, split in two, for the four 16-deep steps
__device__ __forceinline__ void load_lane_frags_h3(const float* scratch, int l31, int half, float scale, f16x8v (&f)[4][2]) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = scratch[l31 * KT_LD + 16 * st + 8 * half + e];
        split_frag8_h3(x, scale, f[st][0], f[st][1]);
    }
}


}  // namespace ttts
