// Scaled dot-product attention (head_dim 64) forward + backward on fp32 MFMA, gfx950.
//
// All three uses of the path share these kernels: encoder self-attention (key-padding mask), decoder
// masked self-attention (causal + key-padding) and encoder-decoder cross-attention (key-padding, per-head
// post-dropout weights written out).  Masks are computed from `key_lens` in-kernel; no mask / score tensor
// ever reaches HBM except the mandatory cross-attention weights.
//
// Orientation ("key on the accumulator rows, query on the lane"): scores are produced transposed,
//   S^T[key][query] = K . (Q*scale)^T      (v_mfma_f32_32x32x2_f32, A = K tile from LDS, B = Q in registers)
// so each lane owns ONE query column: its 16 accumulator registers are 16 keys of that query, the other
// half-wave holds the other 16.  Row max / row sum are therefore lane-local plus one lane^32 exchange
// (wavefront shuffle), and the probabilities are already in the B-operand layout of the next product
//   O^T[d][query] += V^T[d][key] . P^T[key][query]
// with no cross-lane movement at all.  The backward kernels use the same trick in both orientations.
#include "attention_common.h"

namespace ttts {

// minimum waves per SIMD the register allocator must leave room for (launch bounds), per kernel
#ifndef TTTS_DKV_W
#define TTTS_DKV_W 2
#endif
#ifndef TTTS_DQ_W
#define TTTS_DQ_W 2
#endif
#ifndef TTTS_FWD_W
#define TTTS_FWD_W 3
#endif

// ---- cooperative staging (256 threads): KB rows x 64 floats from global straight into LDS; rows beyond
// `nrows_total` are zero.  No register prefetch across the compute phase: with 3-4 workgroups per CU the other
// workgroups cover the load latency, and the registers are worth more as occupancy.
template <bool PADDED>
__device__ __forceinline__ void stage_rows(const float* base, long row0, long nrows_total, int ld, int tid, float* dst,
                                           float scale) {
    constexpr int LDD = PADDED ? KT_LD : HD;
    const RowSrc src = row_src(base, nrows_total, ld);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = (tid >> 4) + 16 * i, c4 = tid & 15;
        long gr = row0 + row;
        v[i] = row_load4(src, gr, c4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = (tid >> 4) + 16 * i, c4 = tid & 15;
        float* d = dst + row * LDD + c4 * 4;
        if (PADDED) {
            d[0] = v[i].x * scale; d[1] = v[i].y * scale; d[2] = v[i].z * scale; d[3] = v[i].w * scale;
        } else {
            *reinterpret_cast<float4*>(d) = make_float4(v[i].x * scale, v[i].y * scale, v[i].z * scale, v[i].w * scale);
        }
    }
}
// =====================================================================================  forward
template <bool CAUSAL, bool WRITE_A>
__global__ __launch_bounds__(256, TTTS_FWD_W) void attn_fwd_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ float ptile_all[WRITE_A ? 4 * 32 * 33 : 1];
    float* Ks = smem;                  // [KB][65]
    float* Vs = smem + KB * KT_LD;     // [KB][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    float* ptile = ptile_all + (WRITE_A ? wave * 32 * 33 : 0);
    // grid = (B*H, query blocks): x (fastest in dispatch order) walks the (batch, head) pairs, y the query blocks, so the
    // whole launch runs heaviest-first for the causal form (last query block = longest key range)
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = smem + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst_live = (kend + KB - 1) / KB;
    const int nst = WRITE_A ? (a.Tk + KB - 1) / KB : nst_live;
    int wave_kend = WRITE_A ? a.Tk : kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;

    // Q fragment: lane (query l31, half) holds Q[q][2j+half] * sqrt(1/64)
    float qreg[32];
    wave_stage_tile(qb_, qw0, a.Tq, a.ldq, lane, scratch, a.qscale);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; ++j) qreg[j] = scratch[l31 * KT_LD + 2 * j + half];

    float m = NEG_INF, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

    const long arow = ((long)(b * a.H + h) * a.Tq);   // row base of the (B,H,Tq,*) outputs
    const uint32_t rowid = (uint32_t)(arow + qg);

    auto scores = [&](const float* ks, f32x16& s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float kf = ks[l31 * KT_LD + 2 * j + half];
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, qreg[j], s, 0, 0, 0);
        }
    };
    auto alive = [&](int key_g) -> bool { return key_g < klen && (!CAUSAL || key_g <= qg); };
    // dropout on the 16 weights of this lane: keys (r, r+1) with r even are neighbours and share one hash
    auto drop16 = [&](float (&p)[16], int key0) {
#pragma unroll
        for (int r = 0; r < 16; r += 4) {     // registers r .. r+3 are four neighbouring keys: one hash
            const uint32_t qh = attn_quad_hash(seed_eff, rowid, (uint32_t)(key0 + acc_row(r, half)) >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) p[r + e] = attn_keep_word(qh, attn_drop_mult(e), thr16) ? p[r + e] * a.drop_scale : 0.f;
        }
    };

    if (WRITE_A) {
        // ---------------- pass 1: row max / row sum only
        for (int t = 0; t < nst_live; ++t) {
            __syncthreads();
            stage_rows<true>(kb_, (long)t * KB, a.Tk, a.ldk, tid, Ks, 1.f);
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int key0 = t * KB + sub * 32;
                if (key0 >= kend) break;
                f32x16 s;
                scores(Ks + sub * 32 * KT_LD, s);
                float mx = NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = alive(key0 + acc_row(r, half)) ? s[r] : NEG_INF;
                    mx = fmaxf(mx, s[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float m_new = fmaxf(m, mx);
                float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                float alpha = __expf(m - m_use);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += __expf(s[r] - m_use);
                l = l * alpha + ps;
                m = m_new;
            }
        }
        l = l + __shfl_xor(l, 32, 64);
    }

    const float m_fin = (m == NEG_INF) ? 0.f : m;
    const float inv_l = (l > 0.f) ? 1.f / l : 0.f;

    // ---------------- main pass
    for (int t = 0; t < nst; ++t) {
        __syncthreads();
        stage_rows<true>(kb_, (long)t * KB, a.Tk, a.ldk, tid, Ks, 1.f);
        stage_rows<false>(vb_, (long)t * KB, a.Tk, a.ldv, tid, Vs, 1.f);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = t * KB + sub * 32;
            if (key0 >= wave_kend) break;      // sub-tile entirely above this wave's causal frontier / past the keys
            const float* ks = Ks + sub * 32 * KT_LD;
            const float* vs = Vs + sub * 32 * HD;
            f32x16 s;
            scores(ks, s);
            float p[16];
            if (WRITE_A) {
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = alive(key0 + acc_row(r, half)) ? __expf(s[r] - m_fin) * inv_l : 0.f;
            } else {
                float mx = NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = alive(key0 + acc_row(r, half)) ? s[r] : NEG_INF;
                    mx = fmaxf(mx, s[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float m_new = fmaxf(m, mx);
                float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                float alpha = __expf(m - m_use);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = __expf(s[r] - m_use); ps += p[r]; }
                l = l * alpha + ps;
                m = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            }
            if (a.thr != 0u) drop16(p, key0);
            if (WRITE_A) {
                // transpose the 32(key) x 32(query) tile through a small per-wave LDS buffer so every weight row leaves
                // as a 128-B segment
#pragma unroll
                for (int r = 0; r < 16; ++r) ptile[l31 * 33 + acc_row(r, half)] = p[r];
                wave_lds_sync();
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const int qrow = 2 * i + half;
                    const float v = ptile[qrow * 33 + l31];
                    const int q_g = qw0 + qrow, key_g = key0 + l31;
                    if (q_g < a.Tq && key_g < a.Tk) a.attn[(arow + q_g) * a.Tk + key_g] = v;
                }
                wave_lds_sync();
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = acc_row(r, half);
                float v0 = vs[krow * HD + l31];
                float v1 = vs[krow * HD + 32 + l31];
                o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, p[r], o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, p[r], o[1], 0, 0, 0);
            }
        }
    }

    float out_scale = 1.f;
    float lse_v;
    if (WRITE_A) {
        lse_v = m_fin + __logf(l > 0.f ? l : 1.f);
    } else {
        float lt = l + __shfl_xor(l, 32, 64);
        out_scale = (lt > 0.f) ? 1.f / lt : 0.f;
        lse_v = ((m == NEG_INF) ? 0.f : m) + __logf(lt > 0.f ? lt : 1.f);
    }
    if (a.lse != nullptr && half == 0 && qg < a.Tq) a.lse[arow + qg] = lse_v;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= out_scale; o[1][r] *= out_scale; }
    __syncthreads();
    wave_store_rows(o, scratch, a.o + (long)b * a.Tq * a.ldo + h * HD, qw0, a.Tq, a.ldo, lane, 1.f);
}

// =====================================================================================  backward: dQ (+ delta)
template <bool CAUSAL>
__global__ __launch_bounds__(256, TTTS_DQ_W) void attn_bwd_dq_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* Ks = smem;                  // [KB][65]
    float* Vs = smem + KB * KT_LD;     // [KB][65]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = smem + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst = (kend + KB - 1) / KB;
    int wave_kend = kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;
    const float* ob_ = a.o + (long)b * a.Tq * a.ldo + h * HD;
    const float* gb_ = a.dout + (long)b * a.Tq * a.ldo + h * HD;
    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);

    float qreg[32], greg[32];
    wave_stage_tile(qb_, qw0, a.Tq, a.ldq, lane, scratch, a.qscale);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; ++j) qreg[j] = scratch[l31 * KT_LD + 2 * j + half];
    wave_lds_sync();
    wave_stage_tile(gb_, qw0, a.Tq, a.ldo, lane, scratch, 1.f);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; ++j) greg[j] = scratch[l31 * KT_LD + 2 * j + half];
    wave_lds_sync();
    wave_stage_tile(ob_, qw0, a.Tq, a.ldo, lane, scratch, 1.f);
    wave_lds_sync();
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) delta += greg[j] * scratch[l31 * KT_LD + 2 * j + half];
    delta += __shfl_xor(delta, 32, 64);
    if (half == 0 && qg < a.Tq) a.delta[arow + qg] = delta;
    const float lse_q = (qg < a.Tq) ? a.lse[arow + qg] : 0.f;

    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    for (int t = 0; t < nst; ++t) {
        __syncthreads();
        stage_rows<true>(kb_, (long)t * KB, a.Tk, a.ldk, tid, Ks, 1.f);
        stage_rows<true>(vb_, (long)t * KB, a.Tk, a.ldv, tid, Vs, 1.f);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = t * KB + sub * 32;
            if (key0 >= wave_kend) break;
            const float* ks = Ks + sub * 32 * KT_LD;
            const float* vs = Vs + sub * 32 * KT_LD;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float kf = ks[l31 * KT_LD + 2 * j + half];
                float vf = vs[l31 * KT_LD + 2 * j + half];
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, qreg[j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, greg[j], dp, 0, 0, 0);
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int key_g = key0 + acc_row(r, half);
                uint32_t qh = 0;
                if (a.thr != 0u) qh = attn_quad_hash(seed_eff, rowid, (uint32_t)key_g >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kg = key_g + e;
                    bool live = kg < klen && (!CAUSAL || kg <= qg);
                    float p = live ? __expf(s[r + e] - lse_q) : 0.f;
                    float g = dp[r + e];
                    if (a.thr != 0u) g = attn_keep_word(qh, attn_drop_mult(e), thr16) ? g * a.drop_scale : 0.f;
                    ds[r + e] = p * (g - delta);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = acc_row(r, half);
                float k0 = ks[krow * KT_LD + l31];
                float k1 = ks[krow * KT_LD + 32 + l31];
                dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(k0, ds[r], dq[0], 0, 0, 0);
                dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(k1, ds[r], dq[1], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    wave_store_rows(dq, scratch, a.dq + (long)b * a.Tq * a.lddq + h * HD, qw0, a.Tq, a.lddq, lane, a.qscale);
}

// =====================================================================================  backward: dK, dV
template <bool CAUSAL>
__global__ __launch_bounds__(256, TTTS_DKV_W) void attn_bwd_dkv_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ float lse_s[KB], delta_s[KB];
    float* Qs = smem;                  // [KB][65]
    float* Gs = smem + KB * KT_LD;     // [KB][65]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int kblk = blockIdx.y;          // ascending = heaviest first for the causal form (key block 0 meets every query)
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int k0 = kblk * QB, kw0 = k0 + wave * 32;
    const int kg = kw0 + l31;
    float* scratch = smem + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;
    const float* gb_ = a.dout + (long)b * a.Tq * a.ldo + h * HD;
    const long arow = ((long)(b * a.H + h) * a.Tq);

    float kreg[32], vreg[32];
    wave_stage_tile(kb_, kw0, a.Tk, a.ldk, lane, scratch, 1.f);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; ++j) kreg[j] = scratch[l31 * KT_LD + 2 * j + half];
    wave_lds_sync();
    wave_stage_tile(vb_, kw0, a.Tk, a.ldv, lane, scratch, 1.f);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; ++j) vreg[j] = scratch[l31 * KT_LD + 2 * j + half];

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

    const int nqs = (a.Tq + KB - 1) / KB;
    int qs_begin = CAUSAL ? (k0 / KB) : 0;        // queries below the block's first key never see it
    if (k0 >= klen) qs_begin = nqs;               // whole key block is padding: gradients are zero

    for (int qs = qs_begin; qs < nqs; ++qs) {
        __syncthreads();
        stage_rows<true>(qb_, (long)qs * KB, a.Tq, a.ldq, tid, Qs, a.qscale);
        stage_rows<true>(gb_, (long)qs * KB, a.Tq, a.ldo, tid, Gs, 1.f);
        if (tid < KB) {
            int q = qs * KB + tid;
            lse_s[tid] = (q < a.Tq) ? a.lse[arow + q] : 0.f;
            delta_s[tid] = (q < a.Tq) ? a.delta[arow + q] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int qt0 = qs * KB + sub * 32;
            if (qt0 >= a.Tq) break;
            if (CAUSAL && qt0 + 31 < kw0) continue;   // every query of the sub-tile precedes this wave's keys
            const float* qsub = Qs + sub * 32 * KT_LD;
            const float* gsub = Gs + sub * 32 * KT_LD;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float qf = qsub[l31 * KT_LD + 2 * j + half];
                float gf = gsub[l31 * KT_LD + 2 * j + half];
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qf, kreg[j], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gf, vreg[j], dp, 0, 0, 0);
            }
            float pd[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = acc_row(r, half);
                const int q_g = qt0 + qrow;
                bool live = kg < klen && (!CAUSAL || kg <= q_g) && q_g < a.Tq;
                float p = live ? __expf(s[r] - lse_s[sub * 32 + qrow]) : 0.f;
                float g = dp[r];
                float pk = p;
                if (a.thr != 0u) {
                    bool keep = attn_keep(seed_eff, (uint32_t)(arow + q_g), (uint32_t)kg, thr16);
                    g = keep ? g * a.drop_scale : 0.f;
                    pk = keep ? p * a.drop_scale : 0.f;
                }
                pd[r] = pk;
                ds[r] = p * (g - delta_s[sub * 32 + qrow]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qrow = acc_row(r, half);
                float g0 = gsub[qrow * KT_LD + l31];
                float g1 = gsub[qrow * KT_LD + 32 + l31];
                float q0f = qsub[qrow * KT_LD + l31];
                float q1f = qsub[qrow * KT_LD + 32 + l31];
                dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, pd[r], dv[0], 0, 0, 0);
                dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, pd[r], dv[1], 0, 0, 0);
                dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0f, ds[r], dk[0], 0, 0, 0);
                dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1f, ds[r], dk[1], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    wave_store_rows(dk, scratch, a.dk + (long)b * a.Tk * a.lddk + h * HD, kw0, a.Tk, a.lddk, lane, 1.f);
    wave_store_rows(dv, scratch, a.dv + (long)b * a.Tk * a.lddv + h * HD, kw0, a.Tk, a.lddv, lane, 1.f);
}

// =====================================================================================================================
// Split-precision forms ("bf16x6", see gemm.hip): the same algorithm with every product formed on
// v_mfma_f32_32x32x16_bf16 from hi/mid/lo bf16 splits of the fp32 operands (six products, fp32-grade accuracy at 6/16 of
// the fp32-MFMA cycles).  What changes is data movement:
//  * K is split while it is staged: three 64 x 64 bf16 planes, 128-byte rows, 16-byte chunks XOR-swizzled with
//    (row >> 1) & 7 so that the fragment reads (ds_read_b128, 16 lanes per clock) are conflict-free.
//  * V is staged TRANSPOSED ([d][key]) because it is the A operand of O^T += V^T P^T and its contraction index (key)
//    must be the fast one.  The probabilities come out of the S^T accumulator with a lane holding keys
//    4*half + (r & 3) + 8*(r >> 2); a 16-deep MFMA step takes registers 8t..8t+7, i.e. keys 16t + {0..3, 8..11} + 4*half.
//    The contraction order is free, so V^T is stored in exactly that key order and P needs no cross-lane movement.
//  * Q (B operand, per lane its query's 64 values) is split once into registers.
constexpr int XP = 64 * 32;     // dwords per 64 x 64 bf16 plane
__device__ __forceinline__ int xsw(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

// 64 rows x 64 floats (row-major, d fast) -> planes[3][64][64]; rows beyond nrows_total are zero
__device__ __forceinline__ void stage_split_rows(const float* base, long row0, long nrows_total, int ld, int tid,
                                                 uint32_t* dst) {
    const RowSrc src = row_src(base, nrows_total, ld);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 4) + 16 * i, c4 = tid & 15;
        v[i] = row_load4(src, row0 + row, c4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 4) + 16 * i, c4 = tid & 15;
        uint2 hi, mid, lo;
        split3_pack4(v[i], hi, mid, lo);
        const int d = xsw(row, c4 >> 1) + (c4 & 1) * 2;
        *reinterpret_cast<uint2*>(dst + d) = hi;
        *reinterpret_cast<uint2*>(dst + XP + d) = mid;
        *reinterpret_cast<uint2*>(dst + 2 * XP + d) = lo;
    }
}
// 64 rows (keys) x 64 floats -> TRANSPOSED planes[3][64 d][64 key positions], key order as described above
__device__ __forceinline__ void stage_split_cols(const float* base, long row0, long nrows_total, int ld, int tid,
                                                 uint32_t* dst) {
    const int dq = tid & 15, kq = tid >> 4;
    const RowSrc src = row_src(base, nrows_total, ld);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = row_load4(src, row0 + 4 * kq + i, dq);
    // keys 4kq..4kq+3 sit at positions pos..pos+3 of their row
    const int pos = (kq >> 3) * 32 + ((kq >> 2) & 1) * 16 + (kq & 1) * 8 + ((kq >> 1) & 1) * 4;
    const float x[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                           {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int d = 4 * dq + c;
        uint2 hi, mid, lo;
        split3_pair(f32x2{x[c][0], x[c][1]}, hi.x, mid.x, lo.x);
        split3_pair(f32x2{x[c][2], x[c][3]}, hi.y, mid.y, lo.y);
        const int dd = xsw(d, pos >> 3) + ((pos >> 2) & 1) * 2;
        *reinterpret_cast<uint2*>(dst + dd) = hi;
        *reinterpret_cast<uint2*>(dst + XP + dd) = mid;
        *reinterpret_cast<uint2*>(dst + 2 * XP + dd) = lo;
    }
}
// 8 fp32 -> three bf16x8 fragments
__device__ __forceinline__ void split_frag8(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    u32x4v h, m, l;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        uint32_t a, b, c;
        split3_pair(f32x2{x[2 * u], x[2 * u + 1]}, a, b, c);
        h[u] = a; m[u] = b; l[u] = c;
    }
    hi = __builtin_bit_cast(bf16x8, h);
    mid = __builtin_bit_cast(bf16x8, m);
    lo = __builtin_bit_cast(bf16x8, l);
}
// c += a (hi,mid,lo) x b (hi,mid,lo), six products, smallest terms first
__device__ __forceinline__ void mfma_x6(f32x16& c, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
}

// waves per SIMD the allocator must leave room for: 3 costs the plain forward a ten-dword spill outside its inner loop
// and is 7 % faster than 2; the weights-writing form is LDS-limited to 2 workgroups per CU anyway
#ifndef TTTS_FWDX_W
#define TTTS_FWDX_W 3
#endif
constexpr int XSMEM = 6 * XP;   // dwords: K planes + V^T planes of one 64-key stage (48 KB)
static_assert(XSMEM >= SMEM_FLOATS, "per-wave fp32 scratch must fit the stage buffers");

template <bool CAUSAL, bool WRITE_A>
__global__ __launch_bounds__(256, WRITE_A ? 2 : TTTS_FWDX_W) void attn_fwd_x6_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) uint32_t xs[XSMEM];
    __shared__ float ptile_all[WRITE_A ? 4 * 32 * 17 : 1];   // per wave: 32 queries x 16 keys (+1 pad)
    uint32_t* Kp = xs;              // [3][64 keys][64 d]
    uint32_t* Vt = xs + 3 * XP;     // [3][64 d][64 key positions]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    float* ptile = ptile_all + (WRITE_A ? wave * 32 * 17 : 0);
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = reinterpret_cast<float*>(xs) + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst_live = (kend + KB - 1) / KB;
    const int nst = WRITE_A ? (a.Tk + KB - 1) / KB : nst_live;
    int wave_kend = WRITE_A ? a.Tk : kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;

    // Q fragments: lane (query l31, half) holds Q[q][16 s + 8 half + 0..7] / 8 for the four d-steps s, split in three
    bf16x8 qf[4][3];
    wave_stage_tile(qb_, qw0, a.Tq, a.ldq, lane, scratch, a.qscale);
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = scratch[l31 * KT_LD + 16 * s + 8 * half + e];
        split_frag8(x, qf[s][0], qf[s][1], qf[s][2]);
    }

    float m = NEG_INF, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);

    auto scores = [&](int sub, f32x16& s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            bf16x8 kf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                kf[p] = *reinterpret_cast<const bf16x8*>(Kp + p * XP + xsw(sub * 32 + l31, 2 * st + half));
            mfma_x6(s, kf, qf[st]);
        }
    };
    auto alive = [&](int key_g) -> bool { return key_g < klen && (!CAUSAL || key_g <= qg); };
    auto drop16 = [&](float (&p)[16], int key0) {
#pragma unroll
        for (int r = 0; r < 16; r += 4) {     // registers r .. r+3 are four neighbouring keys: one hash
            const uint32_t qh = attn_quad_hash(seed_eff, rowid, (uint32_t)(key0 + acc_row(r, half)) >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) p[r + e] = attn_keep_word(qh, attn_drop_mult(e), thr16) ? p[r + e] * a.drop_scale : 0.f;
        }
    };

    if (WRITE_A) {
        // ---------------- pass 1: row max / row sum only
        for (int t = 0; t < nst_live; ++t) {
            __syncthreads();
            stage_split_rows(kb_, (long)t * KB, a.Tk, a.ldk, tid, Kp);
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int key0 = t * KB + sub * 32;
                if (key0 >= kend) break;
                f32x16 s;
                scores(sub, s);
                float mx = NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = alive(key0 + acc_row(r, half)) ? s[r] : NEG_INF;
                    mx = fmaxf(mx, s[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float m_new = fmaxf(m, mx);
                float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                float alpha = __expf(m - m_use);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += __expf(s[r] - m_use);
                l = l * alpha + ps;
                m = m_new;
            }
        }
        l = l + __shfl_xor(l, 32, 64);
    }

    const float m_fin = (m == NEG_INF) ? 0.f : m;
    const float inv_l = (l > 0.f) ? 1.f / l : 0.f;

    // ---------------- main pass
    for (int t = 0; t < nst; ++t) {
        __syncthreads();
        stage_split_rows(kb_, (long)t * KB, a.Tk, a.ldk, tid, Kp);
        stage_split_cols(vb_, (long)t * KB, a.Tk, a.ldv, tid, Vt);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = t * KB + sub * 32;
            if (key0 >= wave_kend) break;
            f32x16 s;
            scores(sub, s);
            float p[16];
            // a tile every lane sees in full needs no mask arithmetic (wave-uniform test)
            const bool full = (key0 + 32 <= klen) && (!CAUSAL || key0 + 31 <= qw0);
            if (WRITE_A) {
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = alive(key0 + acc_row(r, half)) ? __expf(s[r] - m_fin) * inv_l : 0.f;
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
                float alpha = __expf(m - m_use);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = __expf(s[r] - m_use); ps += p[r]; }
                l = l * alpha + ps;
                m = m_new;
                if (__any(alpha != 1.f)) {   // the running maximum rarely moves after the first tiles
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                }
            }
            if (a.thr != 0u) drop16(p, key0);
            if (WRITE_A) {
                // transpose through LDS in two halves of 16 keys (registers 8g..8g+7 are keys 16g..16g+15) so that the
                // weights leave as 64-byte row segments
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ptile[l31 * 17 + (acc_row(8 * g2 + e, half) & 15)] = p[8 * g2 + e];
                    wave_lds_sync();
#pragma unroll 4
                    for (int i = 0; i < 8; ++i) {
                        const int qrow = 4 * i + (lane >> 4), kc = lane & 15;
                        const float v = ptile[qrow * 17 + kc];
                        const int q_g = qw0 + qrow, key_g = key0 + 16 * g2 + kc;
                        if (q_g < a.Tq && key_g < a.Tk) a.attn[(arow + q_g) * a.Tk + key_g] = v;
                    }
                    wave_lds_sync();
                }
            }
            // O^T[d][q] += V^T[d][key] P^T[key][q]: two 16-key steps, registers 8t..8t+7 of the lane are its B fragment
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = p[8 * t2 + e];
                bf16x8 pf[3];
                split_frag8(x, pf[0], pf[1], pf[2]);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    bf16x8 vf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        vf[pl] = *reinterpret_cast<const bf16x8*>(Vt + pl * XP + xsw(32 * i2 + l31, 4 * sub + 2 * t2 + half));
                    mfma_x6(o[i2], vf, pf);
                }
            }
        }
    }

    float out_scale = 1.f;
    float lse_v;
    if (WRITE_A) {
        lse_v = m_fin + __logf(l > 0.f ? l : 1.f);
    } else {
        float lt = l + __shfl_xor(l, 32, 64);
        out_scale = (lt > 0.f) ? 1.f / lt : 0.f;
        lse_v = ((m == NEG_INF) ? 0.f : m) + __logf(lt > 0.f ? lt : 1.f);
    }
    if (a.lse != nullptr && half == 0 && qg < a.Tq) a.lse[arow + qg] = lse_v;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= out_scale; o[1][r] *= out_scale; }
    __syncthreads();
    wave_store_rows(o, scratch, a.o + (long)b * a.Tq * a.ldo + h * HD, qw0, a.Tq, a.ldo, lane, 1.f);
}

// =====================================================================================================================
// fp16x3 form of the forward kernel ("h3", see gemm_h3.hip): the same algorithm with every product formed from THREE
// f16 x f16 MFMA terms of two-way hi/lo f16 splits (a_hi b_hi + a_hi b_lo + a_lo b_hi) instead of six bf16 terms.  f16
// has 11 significand bits but a narrow exponent range, so operands are pre-scaled by powers of two: Q/8, K and V by the
// power of two that puts their measured maximum in [2^11, 2^12) (h3_pow2_scale on the partial maxima their producer
// left: nothing is assumed about their magnitude), probabilities -- which lie in [0, 1/(1-p)] whatever the inputs -- by
// 2^10 (absolute error 3e-11 for the small ones).  The scores stay in accumulator units and the scale rides in the exp2's
// fused multiply-add; the output accumulator is scaled back once at the end.  Two planes per staged operand: 32 KB of
// LDS per workgroup instead of 48.
// the dynamic pre-scales of one launch (wave-uniform, held in scalar registers)
struct H3Scales {
    float sq, sk, sv;          // Q / 8, K, V pre-scales
    float inv_sq, inv_sk, inv_sv;
    float c;                   // accumulator units -> true score: 1 / (sq * sk)
    float c2;                  // ... -> base-2 exponent
    float sg, inv_sg;          // backward only: dO pre-scale (max|dO| over the whole tensor -> [2^11, 2^12))
};
__device__ __forceinline__ float sgpr(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
// Maximum over one TTTS_AMAX_SLOTS array, read ONCE per workgroup: each of the four waves takes a quarter (one 16-byte load
// per lane) and the quarters meet in LDS.  These kernels run thousands of small workgroups (1 792 for a cross-attention
// forward); with every wave reading whole arrays the maxima alone were a quarter of the kernel's L2 traffic (+17 us).
__device__ __forceinline__ float attn_wg_quarter_max(const float* __restrict__ partials, int lane, int wave) {
    static_assert(TTTS_AMAX_SLOTS == 1024, "four waves x 64 lanes x float4");
    const float4 a = reinterpret_cast<const float4*>(partials)[wave * 64 + lane];
    return wave_max(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
}
__device__ __forceinline__ H3Scales attn_h3_scales(const float* q_amax, const float* k_amax, const float* v_amax, int lane,
                                                   float qscale, const float* g_amax = nullptr) {
    H3Scales h;
    float s, i;
    __shared__ float quarter[4][4];
    const int wave = threadIdx.x >> 6;
    // a packed projection output passes the same array three times: read it once (block-uniform branches)
    const bool k_own = k_amax != q_amax, v_own = v_amax != k_amax && v_amax != q_amax;
    const float pq = attn_wg_quarter_max(q_amax, lane, wave);
    const float pk = k_own ? attn_wg_quarter_max(k_amax, lane, wave) : 0.f;
    const float pv = v_own ? attn_wg_quarter_max(v_amax, lane, wave) : 0.f;
    const float pg = g_amax != nullptr ? attn_wg_quarter_max(g_amax, lane, wave) : 0.f;
    if (lane == 0) { quarter[0][wave] = pq; quarter[1][wave] = pk; quarter[2][wave] = pv; quarter[3][wave] = pg; }
    __syncthreads();
    h3_pow2_scale(fmaxf(fmaxf(quarter[3][0], quarter[3][1]), fmaxf(quarter[3][2], quarter[3][3])), s, i);
    h.sg = sgpr(s); h.inv_sg = sgpr(i);
    const float mq = fmaxf(fmaxf(quarter[0][0], quarter[0][1]), fmaxf(quarter[0][2], quarter[0][3]));
    const float mk = k_own ? fmaxf(fmaxf(quarter[1][0], quarter[1][1]), fmaxf(quarter[1][2], quarter[1][3])) : mq;
    const float mv = v_own ? fmaxf(fmaxf(quarter[2][0], quarter[2][1]), fmaxf(quarter[2][2], quarter[2][3]))
                           : (v_amax == k_amax ? mk : mq);
    h3_pow2_scale(qscale * mq, s, i);                                    // Q is multiplied by qscale before it is split
    h.sq = sgpr(s); h.inv_sq = sgpr(i);
    h3_pow2_scale(mk, s, i);
    h.sk = sgpr(s); h.inv_sk = sgpr(i);
    h3_pow2_scale(mv, s, i);
    h.sv = sgpr(s); h.inv_sv = sgpr(i);
    h.c = h.inv_sq * h.inv_sk;
    h.c2 = h.c * 1.4426950408889634f;
    return h;
}
#ifndef TTTS_FWDH_W
#define TTTS_FWDH_W 3
#endif
constexpr int HSMEM = (4 * XP > SMEM_FLOATS) ? 4 * XP : SMEM_FLOATS;   // K + V^T planes (32 KB), re-used as per-wave fp32 scratch

// 64 rows x 64 floats (row-major, d fast) -> planes[2][64][64] of x * scale; rows beyond nrows_total are zero
__device__ __forceinline__ void stage_split_rows_h3(const float* base, long row0, long nrows_total, int ld, int tid,
                                                    uint32_t* dst, float scale) {
    const RowSrc src = row_src(base, nrows_total, ld);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 4) + 16 * i, c4 = tid & 15;
        v[i] = row_load4(src, row0 + row, c4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 4) + 16 * i, c4 = tid & 15;
        uint2 hi, lo;
        split2_pair_h(f32x2{v[i].x, v[i].y} * scale, hi.x, lo.x);
        split2_pair_h(f32x2{v[i].z, v[i].w} * scale, hi.y, lo.y);
        const int d = xsw(row, c4 >> 1) + (c4 & 1) * 2;
        *reinterpret_cast<uint2*>(dst + d) = hi;
        *reinterpret_cast<uint2*>(dst + XP + d) = lo;
    }
}
// 64 rows (keys) x 64 floats -> TRANSPOSED planes[2][64 d][64 key positions], key order as in stage_split_cols
__device__ __forceinline__ void stage_split_cols_h3(const float* base, long row0, long nrows_total, int ld, int tid,
                                                    uint32_t* dst, float scale) {
    const int dq = tid & 15, kq = tid >> 4;
    const RowSrc src = row_src(base, nrows_total, ld);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = row_load4(src, row0 + 4 * kq + i, dq);
    const int pos = (kq >> 3) * 32 + ((kq >> 2) & 1) * 16 + (kq & 1) * 8 + ((kq >> 1) & 1) * 4;
    const float x[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                           {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int d = 4 * dq + c;
        uint2 hi, lo;
        split2_pair_h(f32x2{x[c][0], x[c][1]} * scale, hi.x, lo.x);
        split2_pair_h(f32x2{x[c][2], x[c][3]} * scale, hi.y, lo.y);
        const int dd = xsw(d, pos >> 3) + ((pos >> 2) & 1) * 2;
        *reinterpret_cast<uint2*>(dst + dd) = hi;
        *reinterpret_cast<uint2*>(dst + XP + dd) = lo;
    }
}
template <bool CAUSAL, bool WRITE_A>
__global__ __launch_bounds__(256, WRITE_A ? 2 : TTTS_FWDH_W) void attn_fwd_h3_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    __shared__ __attribute__((aligned(16))) uint32_t xs[HSMEM];
    __shared__ float ptile_all[WRITE_A ? 4 * 32 * 17 : 1];   // per wave: 32 queries x 16 keys (+1 pad)
    uint32_t* Kp = xs;              // [2][64 keys][64 d]      f16 hi / lo planes of K * 2^4
    uint32_t* Vt = xs + 2 * XP;     // [2][64 d][64 key positions]   of V * 2^4

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    float* ptile = ptile_all + (WRITE_A ? wave * 32 * 17 : 0);
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = reinterpret_cast<float*>(xs) + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst_live = (kend + KB - 1) / KB;
    const int nst = WRITE_A ? (a.Tk + KB - 1) / KB : nst_live;
    int wave_kend = WRITE_A ? a.Tk : kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;
    const H3Scales hs = attn_h3_scales(a.q_amax, a.k_amax, a.v_amax, lane, a.qscale);
    const float H3A_C = hs.c, H3A_C2 = hs.c2;

    // Q fragments: lane (query l31, half) holds Q[q][16 s + 8 half + 0..7] / 8 * sq for the four d-steps s, split in two
    f16x8v qf[4][2];
    wave_stage_tile(qb_, qw0, a.Tq, a.ldq, lane, scratch, a.qscale);
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = scratch[l31 * KT_LD + 16 * s + 8 * half + e];
        split_frag8_h3(x, hs.sq, qf[s][0], qf[s][1]);
    }

    float m = NEG_INF, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);

    auto scores = [&](int sub, f32x16& s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            f16x8v kf[2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
                kf[p] = *reinterpret_cast<const f16x8v*>(Kp + p * XP + xsw(sub * 32 + l31, 2 * st + half));
            mfma_h3(s, kf, qf[st]);
        }
    };
    auto alive = [&](int key_g) -> bool { return key_g < klen && (!CAUSAL || key_g <= qg); };
    auto drop16 = [&](float (&p)[16], int key0) {
#pragma unroll
        for (int r = 0; r < 16; r += 4) {     // registers r .. r+3 are four neighbouring keys: one hash
            const uint32_t qh = attn_quad_hash(seed_eff, rowid, (uint32_t)(key0 + acc_row(r, half)) >> 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (the 1/(1-p) of the kept weights rides in the final output scale unless the weights themselves are returned)
                const bool keep = attn_keep_word(qh, attn_drop_mult(e), thr16);
                p[r + e] = keep ? (WRITE_A ? p[r + e] * a.drop_scale : p[r + e]) : 0.f;
            }
        }
    };

    if (WRITE_A) {
        // ---------------- pass 1: row max / row sum only
        for (int t = 0; t < nst_live; ++t) {
            __syncthreads();
            stage_split_rows_h3(kb_, (long)t * KB, a.Tk, a.ldk, tid, Kp, hs.sk);
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int key0 = t * KB + sub * 32;
                if (key0 >= kend) break;
                f32x16 s;
                scores(sub, s);
                float mx = NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = alive(key0 + acc_row(r, half)) ? s[r] : NEG_INF;
                    mx = fmaxf(mx, s[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float m_new = fmaxf(m, mx);
                float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                float alpha = fast_exp2((m - m_use) * H3A_C2);
                const float mc = m_use * H3A_C2;
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ps += fast_exp2(__builtin_fmaf(s[r], H3A_C2, -mc));
                l = l * alpha + ps;
                m = m_new;
            }
        }
        l = l + __shfl_xor(l, 32, 64);
    }

    const float m_fin = (m == NEG_INF) ? 0.f : m;
    const float inv_l = (l > 0.f) ? 1.f / l : 0.f;
    // The exponent of a weight is fma(s, c2, -mcs) with mcs = fl(max * c2): ONE rounded product per row, kept as it is
    // (rowstat) so that the backward forms the very same exponents.  Its rounding error is the same for every key of the
    // row and cancels against the row sum -- here and there -- whatever the scores' magnitude.
    const float mcs_fin = m_fin * H3A_C2;
    float mcs_last = 0.f;        // online form: the subtrahend of the last tile (= of the final maximum)

    // ---------------- main pass
    for (int t = 0; t < nst; ++t) {
        __syncthreads();
        stage_split_rows_h3(kb_, (long)t * KB, a.Tk, a.ldk, tid, Kp, hs.sk);
        stage_split_cols_h3(vb_, (long)t * KB, a.Tk, a.ldv, tid, Vt, hs.sv);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = t * KB + sub * 32;
            if (key0 >= wave_kend) break;
            f32x16 s;
            scores(sub, s);
            float p[16];
            // a tile every lane sees in full needs no mask arithmetic (wave-uniform test)
            const bool full = (key0 + 32 <= klen) && (!CAUSAL || key0 + 31 <= qw0);
            if (WRITE_A) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    p[r] = alive(key0 + acc_row(r, half)) ? fast_exp2(__builtin_fmaf(s[r], H3A_C2, -mcs_fin)) * inv_l : 0.f;
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
                float alpha = fast_exp2((m - m_use) * H3A_C2);
                // the weights are born pre-scaled by 2^10 (the f16 split scale rides in the exponent): l sums them scaled
                const float mc = __builtin_fmaf(m_use, H3A_C2, -10.f);
                mcs_last = mc;
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = fast_exp2(__builtin_fmaf(s[r], H3A_C2, -mc)); ps += p[r]; }
                l = l * alpha + ps;
                m = m_new;
                if (__any(alpha != 1.f)) {   // the running maximum rarely moves after the first tiles
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                }
            }
            if (a.thr != 0u) drop16(p, key0);
            if (WRITE_A) {
                // transpose through LDS in two halves of 16 keys (registers 8g..8g+7 are keys 16g..16g+15) so that the
                // weights leave as 64-byte row segments
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ptile[l31 * 17 + (acc_row(8 * g2 + e, half) & 15)] = p[8 * g2 + e];
                    wave_lds_sync();
#pragma unroll 4
                    for (int i = 0; i < 8; ++i) {
                        const int qrow = 4 * i + (lane >> 4), kc = lane & 15;
                        const float v = ptile[qrow * 17 + kc];
                        const int q_g = qw0 + qrow, key_g = key0 + 16 * g2 + kc;
                        if (q_g < a.Tq && key_g < a.Tk) a.attn[(arow + q_g) * a.Tk + key_g] = v;
                    }
                    wave_lds_sync();
                }
            }
            // O^T[d][q] += V^T[d][key] P^T[key][q]: two 16-key steps, registers 8t..8t+7 of the lane are its B fragment
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = p[8 * t2 + e];
                f16x8v pf[2];
                split_frag8_h3(x, WRITE_A ? H3A_P : 1.0f, pf[0], pf[1]);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    f16x8v vf[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        vf[pl] = *reinterpret_cast<const f16x8v*>(Vt + pl * XP + xsw(32 * i2 + l31, 4 * sub + 2 * t2 + half));
                    mfma_h3(o[i2], vf, pf);
                }
            }
        }
    }

    float out_scale = hs.inv_sv * (1.0f / H3A_P);       // the O accumulator holds (V * sv)^T (P * 2^10)^T
    float lse_v;
    if (WRITE_A) {
        lse_v = m_fin * H3A_C + __logf(l > 0.f ? l : 1.f);
    } else {
        float lt = l + __shfl_xor(l, 32, 64);                 // = 2^10 * the sum of the weights
        out_scale = (lt > 0.f) ? (a.drop_scale * hs.inv_sv) / lt : 0.f;
        lse_v = ((m == NEG_INF) ? 0.f : m) * H3A_C + __logf(lt > 0.f ? lt * (1.0f / H3A_P) : 1.f);
    }
    if (a.lse != nullptr && half == 0 && qg < a.Tq) a.lse[arow + qg] = lse_v;
    // row statistics for the backward: the subtrahend the weights were formed with and log2 of their sum (in the online form
    // both carry the 2^10 the weights are born with; a row whose last tile did not move the maximum used the same value)
    const float lsum = WRITE_A ? l : (l + __shfl_xor(l, 32, 64));
    if (a.rowstat != nullptr && half == 0 && qg < a.Tq) {
        const long plane = (long)a.B * a.H * a.Tq;
        const float mcs_w = WRITE_A ? mcs_fin : ((m == NEG_INF) ? -10.f : __builtin_fmaf(m, H3A_C2, -10.f));
        a.rowstat[arow + qg] = mcs_w;
        a.rowstat[plane + arow + qg] = lsum > 0.f ? __log2f(lsum) : 0.f;
        // third plane: 1 for a ONE-HOT row -- its sum IS its largest term (the same instruction, on the same operands, that
        // formed the maximum's own term in the loop: the running maximum is one of the scores, and the subtrahend of every
        // later tile is the one it set), i.e. every other exponential vanished in fp32.  The backward takes such a row's
        // softmax gradient as the exact zero it is (attn_bwd_dq_h3_kernel, `saturated`).
        const float top = (m == NEG_INF) ? 0.f : fast_exp2(__builtin_fmaf(m, H3A_C2, -(WRITE_A ? mcs_fin : mcs_w)));
        a.rowstat[2 * plane + arow + qg] = (lsum > 0.f && lsum == top) ? 1.f : 0.f;
    }
    (void)mcs_last;
    float omax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o[0][r] *= out_scale; o[1][r] *= out_scale;
        omax = fmaxf(omax, fmaxf(fabsf(o[0][r]), fabsf(o[1][r])));
    }
    // max|o| for the fp16x3 out-projection that consumes it (rows past Tq are not stored)
    if (a.o_amax != nullptr) amax_publish(qg < a.Tq ? omax : 0.f, a.o_amax, blockIdx.y * gridDim.x + blockIdx.x);
    __syncthreads();
    wave_store_rows(o, scratch, a.o + (long)b * a.Tq * a.ldo + h * HD, qw0, a.Tq, a.ldo, lane, 1.f);
}

// ---- backward, split-precision.  Both kernels need one streamed operand in BOTH orientations (row-major for the score
// products, transposed for the gradient products), so a thread stages a 4 x 4 patch: split once, packed twice (the
// transposed pairs are re-packed from the row-major ones with v_perm_b32 instead of being split again).
__device__ __forceinline__ void patch_load(const float* base, long row0, long nrows_total, int ld, int rq, int dq,
                                           float scale, float4 (&v)[4]) {
    const RowSrc src = row_src(base, nrows_total, ld);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = row_load4(src, row0 + 4 * rq + i, dq);
        v[i].x *= scale; v[i].y *= scale; v[i].z *= scale; v[i].w *= scale;
    }
}
// rows 4rq..4rq+3, columns 4dq..4dq+3 -> row-major planes `rows` (stride XPR dwords per plane, 128-byte rows) and, when
// TR, transposed planes `cols` ([64 d][NPOS positions], plane stride XPC; `pos` = position of row 4rq in the permuted
// order; chunk swizzle for 128-byte rows: xsw, for 64-byte rows: xsw4)
__device__ __forceinline__ int xsw4(int row, int chunk) { return row * 16 + ((chunk ^ ((row >> 2) & 3)) << 2); }
template <bool TR, int NPOS>
__device__ __forceinline__ void patch_split_store(const float4 (&v)[4], int rq, int dq, int pos, uint32_t* rows, int XPR,
                                                  uint32_t* cols, int XPC) {
    uint2 hi[4], mid[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        split3_pack4(v[i], hi[i], mid[i], lo[i]);
        const int d = xsw(4 * rq + i, dq >> 1) + (dq & 1) * 2;
        *reinterpret_cast<uint2*>(rows + d) = hi[i];
        *reinterpret_cast<uint2*>(rows + XPR + d) = mid[i];
        *reinterpret_cast<uint2*>(rows + 2 * XPR + d) = lo[i];
    }
    if (TR) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t sel = (c & 1) ? 0x07060302u : 0x05040100u;   // high / low halves of the two source dwords
            uint2 th, tm, tl;
            if (c < 2) {
                th.x = __builtin_amdgcn_perm(hi[1].x, hi[0].x, sel); th.y = __builtin_amdgcn_perm(hi[3].x, hi[2].x, sel);
                tm.x = __builtin_amdgcn_perm(mid[1].x, mid[0].x, sel); tm.y = __builtin_amdgcn_perm(mid[3].x, mid[2].x, sel);
                tl.x = __builtin_amdgcn_perm(lo[1].x, lo[0].x, sel); tl.y = __builtin_amdgcn_perm(lo[3].x, lo[2].x, sel);
            } else {
                th.x = __builtin_amdgcn_perm(hi[1].y, hi[0].y, sel); th.y = __builtin_amdgcn_perm(hi[3].y, hi[2].y, sel);
                tm.x = __builtin_amdgcn_perm(mid[1].y, mid[0].y, sel); tm.y = __builtin_amdgcn_perm(mid[3].y, mid[2].y, sel);
                tl.x = __builtin_amdgcn_perm(lo[1].y, lo[0].y, sel); tl.y = __builtin_amdgcn_perm(lo[3].y, lo[2].y, sel);
            }
            const int d = 4 * dq + c;
            const int dd = (NPOS == 64 ? xsw(d, pos >> 3) : xsw4(d, pos >> 3)) + ((pos >> 2) & 1) * 2;
            *reinterpret_cast<uint2*>(cols + dd) = th;
            *reinterpret_cast<uint2*>(cols + XPC + dd) = tm;
            *reinterpret_cast<uint2*>(cols + 2 * XPC + dd) = tl;
        }
    }
}
// position (in the permuted contraction order) of row 4*rq within its 32-row group: 16 t + 8 h + 4 (e >> 2), see above
__device__ __forceinline__ int perm_pos(int rq) { return ((rq >> 2) & 1) * 16 + (rq & 1) * 8 + ((rq >> 1) & 1) * 4; }

// lane-resident B operand: the 64 values of this lane's row (query or key), split, for the four 16-deep steps
__device__ __forceinline__ void load_lane_frags(const float* scratch, int l31, int half, bf16x8 (&f)[4][3]) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = scratch[l31 * KT_LD + 16 * st + 8 * half + e];
        split_frag8(x, f[st][0], f[st][1], f[st][2]);
    }
}

#ifndef TTTS_DQX_W
#define TTTS_DQX_W 2
#endif
#ifndef TTTS_DKVX_W
#define TTTS_DKVX_W 2
#endif
constexpr int DQX_SMEM = 9 * XP * 4;   // bytes: K rows, V rows, K^T of one 64-key stage (72 KB, dynamic)

template <bool CAUSAL>
__global__ __launch_bounds__(256, TTTS_DQX_W) void attn_bwd_dq_x6_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t xsd[];
    uint32_t* Kr = xsd;              // [3][64 keys][64 d]
    uint32_t* Vr = xsd + 3 * XP;     // [3][64 keys][64 d]
    uint32_t* Kt = xsd + 6 * XP;     // [3][64 d][64 key positions]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = reinterpret_cast<float*>(xsd) + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst = (kend + KB - 1) / KB;
    int wave_kend = kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;
    const float* ob_ = a.o + (long)b * a.Tq * a.ldo + h * HD;
    const float* gb_ = a.dout + (long)b * a.Tq * a.ldo + h * HD;
    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);

    bf16x8 qf[4][3], gf[4][3];
    wave_stage_tile(qb_, qw0, a.Tq, a.ldq, lane, scratch, a.qscale);
    wave_lds_sync();
    load_lane_frags(scratch, l31, half, qf);
    wave_lds_sync();
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
    load_lane_frags(scratch, l31, half, gf);
    if (half == 0 && qg < a.Tq) a.delta[arow + qg] = delta;
    const float lse_q = (qg < a.Tq) ? a.lse[arow + qg] : 0.f;

    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    const int rq = tid >> 4, dqd = tid & 15;
    const int kpos = (rq >> 3) * 32 + perm_pos(rq);
    for (int t = 0; t < nst; ++t) {
        __syncthreads();
        {
            float4 v[4];
            patch_load(kb_, (long)t * KB, a.Tk, a.ldk, rq, dqd, 1.f, v);
            patch_split_store<true, 64>(v, rq, dqd, kpos, Kr, XP, Kt, XP);
            patch_load(vb_, (long)t * KB, a.Tk, a.ldv, rq, dqd, 1.f, v);
            patch_split_store<false, 64>(v, rq, dqd, 0, Vr, XP, nullptr, 0);
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int key0 = t * KB + sub * 32;
            if (key0 >= wave_kend) break;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                bf16x8 kf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    kf[p] = *reinterpret_cast<const bf16x8*>(Kr + p * XP + xsw(sub * 32 + l31, 2 * st + half));
                mfma_x6(s, kf, qf[st]);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                bf16x8 vf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    vf[p] = *reinterpret_cast<const bf16x8*>(Vr + p * XP + xsw(sub * 32 + l31, 2 * st + half));
                mfma_x6(dp, vf, gf[st]);
            }
            float ds[16];
            // a tile every lane sees in full needs no mask arithmetic (wave-uniform test)
            const bool full = (key0 + 32 <= klen) && (!CAUSAL || key0 + 31 <= qw0);
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int key_g = key0 + acc_row(r, half);
                uint32_t qh = 0;
                if (a.thr != 0u) qh = attn_quad_hash(seed_eff, rowid, (uint32_t)key_g >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kg = key_g + e;
                    float p = __expf(s[r + e] - lse_q);
                    if (!full) p = (kg < klen && (!CAUSAL || kg <= qg)) ? p : 0.f;
                    float g = dp[r + e];
                    if (a.thr != 0u) g = attn_keep_word(qh, attn_drop_mult(e), thr16) ? g * a.drop_scale : 0.f;
                    ds[r + e] = p * (g - delta);
                }
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = ds[8 * t2 + e];
                bf16x8 dsf[3];
                split_frag8(x, dsf[0], dsf[1], dsf[2]);
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    bf16x8 ktf[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        ktf[p] = *reinterpret_cast<const bf16x8*>(Kt + p * XP + xsw(32 * i2 + l31, 4 * sub + 2 * t2 + half));
                    mfma_x6(dq[i2], ktf, dsf);
                }
            }
        }
    }
    __syncthreads();
    wave_store_rows(dq, scratch, a.dq + (long)b * a.Tq * a.lddq + h * HD, qw0, a.Tq, a.lddq, lane, a.qscale);
}

// dK, dV: key on the lane; one stage = 32 queries in both orientations (Q, dO row-major for S / dP, transposed for dK / dV).
// The raw fp32 rows of stage t+1 are fetched by LDS-DMA (global_load_lds, no registers) while stage t is multiplied,
// so the split / re-pack pass at the top of a stage reads LDS instead of waiting for HBM or L2.
constexpr int QS = 32;                  // queries per stage
constexpr int XPQ = QS * 32;            // dwords per row-major plane of a stage: 32 rows x 128 B
constexpr int XPT = 64 * 16;            // dwords per transposed plane: 64 d x 32 positions
constexpr int RAWQ = QS * 64;           // dwords of one raw fp32 stage tile (8 KB)
constexpr int DKVX_DW = 6 * XPQ + 6 * XPT + 2 * RAWQ + 256;   // planes 48 KB + raw Q, dO 16 KB + 2 x (lse, delta) rows
static_assert(DKVX_DW >= SMEM_FLOATS, "per-wave fp32 scratch must fit the stage buffers");
constexpr int DKVX_SMEM = DKVX_DW * 4;   // 66 048 bytes: dynamic (above the 64 KB static limit)

// LDS-DMA through inline assembly: with the builtin, hipcc treats an in-flight DMA as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of the NEXT LDS READ of any address -- i.e. it waits for the prefetch at the first fragment
// read of the stage it was supposed to overlap.  An asm statement is outside hipcc's counter bookkeeping; the kernel
// waits for its DMAs itself (vmcnt(0) before the barrier that publishes the stage).  M0 (the LDS destination base) is
// saved and restored inside the statement.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// source = wave-uniform base pointer (SGPR pair) + per-lane 32-bit byte offset: one VGPR per request
__device__ __forceinline__ void dma16(const float* base, uint32_t lane_off, uint32_t* lds_wave_base) {
    // 64 lanes x 16 bytes: lane i lands at lds_wave_base + 16 i (the destination is wave-uniform base + lane * size)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(lds_addr(lds_wave_base)), "s"(base) : "memory");
}
__device__ __forceinline__ void dma4(const float* base, uint32_t lane_off, uint32_t* lds_wave_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(lds_addr(lds_wave_base)), "s"(base) : "memory");
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, TTTS_DKVX_W) void attn_bwd_dkv_x6_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t xsd[];
    uint32_t* xs = xsd;
    uint32_t* Qr = xs;                      // [3][32 q][64 d]
    uint32_t* Gr = xs + 3 * XPQ;
    uint32_t* Qt = xs + 6 * XPQ;            // [3][64 d][32 q positions]
    uint32_t* Gt = xs + 6 * XPQ + 3 * XPT;
    uint32_t* rawQ = xs + 6 * XPQ + 6 * XPT;   // [32][64] fp32, filled by DMA
    uint32_t* rawG = rawQ + RAWQ;
    float* stat_s = reinterpret_cast<float*>(rawG + RAWQ);  // [2 buffers][lse 64 | delta 64] (32 of each 64 used)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int kblk = blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int k0 = kblk * QB, kw0 = k0 + wave * 32;
    const int kg = kw0 + l31;
    const uint32_t key_mult = attn_drop_mult((uint32_t)kg);
    float* scratch = reinterpret_cast<float*>(xs) + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;
    const float* gb_ = a.dout + (long)b * a.Tq * a.ldo + h * HD;
    const long arow = ((long)(b * a.H + h) * a.Tq);

    bf16x8 kf[4][3], vf[4][3];
    wave_stage_tile(kb_, kw0, a.Tk, a.ldk, lane, scratch, 1.f);
    wave_lds_sync();
    load_lane_frags(scratch, l31, half, kf);
    wave_lds_sync();
    wave_stage_tile(vb_, kw0, a.Tk, a.ldv, lane, scratch, 1.f);
    wave_lds_sync();
    load_lane_frags(scratch, l31, half, vf);

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

    const int nqs = (a.Tq + QS - 1) / QS;
    int qs_begin = CAUSAL ? (k0 / QS) : 0;
    if (k0 >= klen) qs_begin = nqs;

    // staging roles: threads 0..127 take Q, 128..255 take dO; each a 4 x 4 patch of the 32 x 64 stage
    const int st_t = tid & 127, rq = st_t >> 4, dqd = st_t & 15;
    const bool st_q = tid < 128;
    const int qpos = perm_pos(rq);

    // DMA of one stage: wave w moves rows 8w..8w+7 of Q and of dO (two 1-KB instructions each), wave 0 also lse / delta.
    // Rows past Tq are clamped to the last row here and zeroed when they are split.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // scalar copy: the DMA destinations must not cost VGPRs
    const int dma_row = 8 * wave + (lane >> 4);
    const uint32_t dma_col = (uint32_t)(lane & 15) * 16u;
    const float* lse_b = a.lse + arow;
    const float* delta_b = a.delta + arow;
    auto fetch = [&](int qt0, int sb) {
        if (wave_u == 0) {     // first: whatever the allocator does to this address must not wait on the big requests
            int q = qt0 + l31;
            if (q > a.Tq - 1) q = a.Tq - 1;
            dma4(lse_b, (uint32_t)q * 4u, reinterpret_cast<uint32_t*>(stat_s + sb * 128));
            dma4(delta_b, (uint32_t)q * 4u, reinterpret_cast<uint32_t*>(stat_s + sb * 128 + 64));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int gr = qt0 + dma_row + 4 * j;
            if (gr > a.Tq - 1) gr = a.Tq - 1;
            dma16(qb_, (uint32_t)gr * (uint32_t)(a.ldq * 4) + dma_col, rawQ + (8 * wave_u + 4 * j) * 64);
            dma16(gb_, (uint32_t)gr * (uint32_t)(a.ldo * 4) + dma_col, rawG + (8 * wave_u + 4 * j) * 64);
        }
    };

    __syncthreads();                       // the per-wave scratch (aliasing the planes) is dead from here on
    if (qs_begin < nqs) fetch(qs_begin * QS, qs_begin & 1);

    for (int qs = qs_begin; qs < nqs; ++qs) {
        const int qt0 = qs * QS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the stage have landed
        __syncthreads();                                    // ... and everybody else's; previous planes are free
        {
            const uint32_t* raw = st_q ? rawQ : rawG;
            const float sc = st_q ? a.qscale : 1.f;
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * rq + i;
                const u32x4v u = *reinterpret_cast<const u32x4v*>(raw + row * 64 + dqd * 4);
                const bool ok = qt0 + row < a.Tq;
                v[i] = make_float4(ok ? __uint_as_float(u[0]) * sc : 0.f, ok ? __uint_as_float(u[1]) * sc : 0.f,
                                   ok ? __uint_as_float(u[2]) * sc : 0.f, ok ? __uint_as_float(u[3]) * sc : 0.f);
            }
            if (st_q) patch_split_store<true, 32>(v, rq, dqd, qpos, Qr, XPQ, Qt, XPT);
            else patch_split_store<true, 32>(v, rq, dqd, qpos, Gr, XPQ, Gt, XPT);
        }
        const float* lse_s = stat_s + (qs & 1) * 128;
        const float* delta_s = lse_s + 64;
        __syncthreads();
        if (qs + 1 < nqs) fetch(qt0 + QS, (qs + 1) & 1);   // in flight while this stage is multiplied
        if (CAUSAL && qt0 + 31 < kw0) continue;     // every query of the stage precedes this wave's keys (wave-uniform)
        if (kw0 >= klen) continue;                  // this wave's keys are all padding

        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            bf16x8 qfr[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) qfr[p] = *reinterpret_cast<const bf16x8*>(Qr + p * XPQ + xsw(l31, 2 * st + half));
            mfma_x6(s, qfr, kf[st]);
        }
        float pd[16];
        const bool full = (kw0 + 32 <= klen) && (!CAUSAL || kw0 + 31 <= qt0) && (qt0 + QS <= a.Tq);   // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q_g = qt0 + acc_row(r, half);
            float p = __expf(s[r] - lse_s[acc_row(r, half)]);
            if (!full) p = (kg < klen && (!CAUSAL || kg <= q_g) && q_g < a.Tq) ? p : 0.f;
            pd[r] = p;
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            bf16x8 gfr[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) gfr[p] = *reinterpret_cast<const bf16x8*>(Gr + p * XPQ + xsw(l31, 2 * st + half));
            mfma_x6(dp, gfr, vf[st]);
        }
        float ds[16];
        // dropout: keys 4j .. 4j+3 (a quad of lanes) share one hash word per query row -- lane 4j+i computes it for registers
        // r+i of each group of four, quad-permute DPP broadcasts hand every lane the word of each row, and the lane's own
        // multiplier (its key & 3) picks its 16 bits
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            uint32_t hq[4] = {0u, 0u, 0u, 0u};
            if (a.thr != 0u) {
                const int rr = r + (lane & 3);
                const uint32_t mine = attn_quad_hash(seed_eff, (uint32_t)(arow + qt0 + acc_row(rr, half)), (uint32_t)kg >> 2);
                hq[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]
                hq[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x55, 0xF, 0xF, true);
                hq[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xAA, 0xF, 0xF, true);
                hq[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xFF, 0xF, 0xF, true);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = dp[r + e];
                float pk = pd[r + e];
                if (a.thr != 0u) {
                    const bool keep = attn_keep_word(hq[e], key_mult, thr16);
                    g = keep ? g * a.drop_scale : 0.f;
                    pk = keep ? pk * a.drop_scale : 0.f;
                }
                ds[r + e] = pd[r + e] * (g - delta_s[acc_row(r + e, half)]);
                pd[r + e] = pk;
            }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            float x[8];
            bf16x8 f[3];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = pd[8 * t2 + e];
            split_frag8(x, f[0], f[1], f[2]);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8*>(Gt + p * XPT + xsw4(32 * i2 + l31, 2 * t2 + half));
                mfma_x6(dv[i2], af, f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ds[8 * t2 + e];
            split_frag8(x, f[0], f[1], f[2]);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8*>(Qt + p * XPT + xsw4(32 * i2 + l31, 2 * t2 + half));
                mfma_x6(dk[i2], af, f);
            }
        }
    }
    __syncthreads();
    wave_store_rows(dk, scratch, a.dk + (long)b * a.Tk * a.lddk + h * HD, kw0, a.Tk, a.lddk, lane, 1.f);
    wave_store_rows(dv, scratch, a.dv + (long)b * a.Tk * a.lddv + h * HD, kw0, a.Tk, a.lddv, lane, 1.f);
}

// =====================================================================================================================
// fp16x3 forms of the backward kernels.  Q / 8, K, V get the forward's dynamic pre-scales (attn_h3_scales, from the same
// partial maxima), P x 2^10; the two gradient operands:
//   * dO: the power of two that puts max|dO| over the whole tensor in [2^11, 2^12), from the partial maxima of
//     ttts_amax_partials (AttnArgs.do_amax).  It is one value for the tensor because dO is contracted over its rows in
//     dV^T += dO^T P and over its columns in dP = dO V^T.
//   * dS = P (dP - delta): produced and consumed in registers as the lane's own accumulator column (its query in the dQ
//     kernel, its key in the dK/dV kernel), so its pre-scale is lane-local -- but the products of all tiles accumulate into
//     the same registers, so it must be one value per lane for the whole loop.  attn_h3_track_scale sets it from the first
//     non-zero tile (max -> [2^11, 2^12)) and, when a later tile would exceed 2^13, lowers it and rescales the lane's
//     accumulator column by the same power of two (the online-softmax trick applied to a scale instead of a maximum).
#ifndef TTTS_DQH_W
#define TTTS_DQH_W 2
#endif
#ifndef TTTS_DKVH_W
#define TTTS_DKVH_W 2
#endif
constexpr int DQH_SMEM = ((6 * XP > SMEM_FLOATS) ? 6 * XP : SMEM_FLOATS) * 4;   // K rows, V rows, K^T: 2 planes each (48 KB)
constexpr int DKVH_DW = 4 * XPQ + 4 * XPT + 2 * RAWQ + 384;                    // planes 32 KB + raw Q, dO 16 KB + stats
static_assert(DKVH_DW >= SMEM_FLOATS, "per-wave fp32 scratch must fit the stage buffers");
constexpr int DKVH_SMEM = DKVH_DW * 4;

// rows 4rq..4rq+3, columns 4dq..4dq+3 (already pre-scaled) -> row-major f16 planes `rows` and, when TR, transposed planes
// `cols`; layouts, swizzles and the v_perm re-pack as patch_split_store
template <bool TR, int NPOS>
__device__ __forceinline__ void patch_split_store_h3(const float4 (&v)[4], int rq, int dq, int pos, uint32_t* rows, int XPR,
                                                     uint32_t* cols, int XPC) {
    uint2 hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        split2_pair_h(f32x2{v[i].x, v[i].y}, hi[i].x, lo[i].x);
        split2_pair_h(f32x2{v[i].z, v[i].w}, hi[i].y, lo[i].y);
        const int d = xsw(4 * rq + i, dq >> 1) + (dq & 1) * 2;
        *reinterpret_cast<uint2*>(rows + d) = hi[i];
        *reinterpret_cast<uint2*>(rows + XPR + d) = lo[i];
    }
    if (TR) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t sel = (c & 1) ? 0x07060302u : 0x05040100u;   // high / low halves of the two source dwords
            uint2 th, tl;
            if (c < 2) {
                th.x = __builtin_amdgcn_perm(hi[1].x, hi[0].x, sel); th.y = __builtin_amdgcn_perm(hi[3].x, hi[2].x, sel);
                tl.x = __builtin_amdgcn_perm(lo[1].x, lo[0].x, sel); tl.y = __builtin_amdgcn_perm(lo[3].x, lo[2].x, sel);
            } else {
                th.x = __builtin_amdgcn_perm(hi[1].y, hi[0].y, sel); th.y = __builtin_amdgcn_perm(hi[3].y, hi[2].y, sel);
                tl.x = __builtin_amdgcn_perm(lo[1].y, lo[0].y, sel); tl.y = __builtin_amdgcn_perm(lo[3].y, lo[2].y, sel);
            }
            const int d = 4 * dq + c;
            const int dd = (NPOS == 64 ? xsw(d, pos >> 3) : xsw4(d, pos >> 3)) + ((pos >> 2) & 1) * 2;
            *reinterpret_cast<uint2*>(cols + dd) = th;
            *reinterpret_cast<uint2*>(cols + XPC + dd) = tl;
        }
    }
}
template <bool CAUSAL>
__global__ __launch_bounds__(256, TTTS_DQH_W) void attn_bwd_dq_h3_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t xsd[];
    uint32_t* Kr = xsd;              // [2][64 keys][64 d]       f16 hi / lo of K * 2^4
    uint32_t* Vr = xsd + 2 * XP;     // [2][64 keys][64 d]       of V * 2^4
    uint32_t* Kt = xsd + 4 * XP;     // [2][64 d][64 key positions]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int qblk = CAUSAL ? (gridDim.y - 1 - blockIdx.y) : blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int q0 = qblk * QB, qw0 = q0 + wave * 32;
    const int qg = qw0 + l31;
    float* scratch = reinterpret_cast<float*>(xsd) + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;
    int kend = klen;
    if (CAUSAL && kend > q0 + QB) kend = q0 + QB;
    const int nst = (kend + KB - 1) / KB;
    int wave_kend = kend;
    if (CAUSAL && wave_kend > qw0 + 32) wave_kend = qw0 + 32;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;
    const float* ob_ = a.o + (long)b * a.Tq * a.ldo + h * HD;
    const float* gb_ = a.dout + (long)b * a.Tq * a.ldo + h * HD;
    const long arow = ((long)(b * a.H + h) * a.Tq);
    const uint32_t rowid = (uint32_t)(arow + qg);

    // dO is a gradient: its pre-scale is the power of two that puts max|dO| (over the whole tensor) in [2^11, 2^12)
    const H3Scales hs = attn_h3_scales(a.q_amax, a.k_amax, a.v_amax, lane, a.qscale, a.do_amax);
    const float s_g = hs.sg, inv_g = hs.inv_sg;
    const float H3A_C2 = hs.c2;
    f16x8v qf[4][2], gf[4][2];
    wave_stage_tile(qb_, qw0, a.Tq, a.ldq, lane, scratch, a.qscale);
    wave_lds_sync();
    load_lane_frags_h3(scratch, l31, half, hs.sq, qf);
    wave_lds_sync();
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
    // row statistics of this query: the forward's exponent subtrahend and log2 of its row sum (rowstat), or lse alone
    float m_q = 0.f, l2_q = 0.f;
    bool saturated = false;
    if (qg < a.Tq) {
        if (a.rowstat != nullptr) {
            m_q = a.rowstat[arow + qg];
            l2_q = a.rowstat[(long)a.B * a.H * a.Tq + arow + qg];
            // A row whose sum is exactly its largest term (every other exponential vanished in fp32) is ONE-HOT.
            // torch keeps the probabilities and its softmax backward is then exactly zero (1 * (dP_k - dP_k)); a backward that
            // recomputes P forms P (dP - delta) with delta = rowsum(dO * O), two roundings of the same number that do not
            // cancel -- 2^-22 |dO| |V| of noise on a gradient that should vanish, multiplied by |K|, |Q| (scores ~1e6 come
            // from inputs ~1e3: 4-5 % of the pre-net's whole gradient).  Such a row's dS is taken as the exact zero it is; the
            // dK / dV kernel learns of it through a sentinel in `delta` (-0.0: a genuine delta of that value belongs to a row
            // whose dS is zero anyway).
            saturated = a.rowstat[2 * (long)a.B * a.H * a.Tq + arow + qg] != 0.f;     // the forward's verdict: sum == largest term
        } else {
            l2_q = a.lse[arow + qg] * 1.4426950408889634f;
        }
    }
    if (half == 0 && qg < a.Tq) a.delta[arow + qg] = saturated ? -0.f : delta;
    const float dp_unscale = inv_g * hs.inv_sv * a.drop_scale;   // dP accumulator units -> true dP, times the 1/(1-p) of kept weights
    float sds = 0.f;                              // this query's dS pre-scale (power of two), set / lowered on the fly

    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    const int rq = tid >> 4, dqd = tid & 15;
    const int kpos = (rq >> 3) * 32 + perm_pos(rq);
    for (int t = 0; t < nst; ++t) {
        __syncthreads();
        {
            float4 v[4];
            patch_load(kb_, (long)t * KB, a.Tk, a.ldk, rq, dqd, hs.sk, v);
            patch_split_store_h3<true, 64>(v, rq, dqd, kpos, Kr, XP, Kt, XP);
            patch_load(vb_, (long)t * KB, a.Tk, a.ldv, rq, dqd, hs.sv, v);
            patch_split_store_h3<false, 64>(v, rq, dqd, 0, Vr, XP, nullptr, 0);
        }
        __syncthreads();
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
                for (int p = 0; p < 2; ++p)
                    kf[p] = *reinterpret_cast<const f16x8v*>(Kr + p * XP + xsw(sub * 32 + l31, 2 * st + half));
                mfma_h3(s, kf, qf[st]);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                f16x8v vf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    vf[p] = *reinterpret_cast<const f16x8v*>(Vr + p * XP + xsw(sub * 32 + l31, 2 * st + half));
                mfma_h3(dp, vf, gf[st]);
            }
            float ds[16];
            // a tile every lane sees in full needs no mask arithmetic (wave-uniform test)
            const bool full = (key0 + 32 <= klen) && (!CAUSAL || key0 + 31 <= qw0);
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int key_g = key0 + acc_row(r, half);
                uint32_t qh = 0;
                if (a.thr != 0u) qh = attn_quad_hash(seed_eff, rowid, (uint32_t)key_g >> 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kg = key_g + e;
                    float p = fast_exp2(__builtin_fmaf(s[r + e], H3A_C2, -m_q) - l2_q);
                    if (!full) p = (kg < klen && (!CAUSAL || kg <= qg)) ? p : 0.f;
                    float g = dp[r + e] * dp_unscale;
                    if (a.thr != 0u) g = attn_keep_word(qh, attn_drop_mult(e), thr16) ? g : 0.f;
                    ds[r + e] = saturated ? 0.f : p * (g - delta);
                }
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q].  dS is the lane's own column (its query), so its f16 pre-scale is
            // lane-local; it has to be ONE value over all key tiles because they accumulate into the same dq registers,
            // so it is lowered when a tile outgrows it and the (lane-local) accumulator column is rescaled with it
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
                    for (int p = 0; p < 2; ++p)
                        ktf[p] = *reinterpret_cast<const f16x8v*>(Kt + p * XP + xsw(32 * i2 + l31, 4 * sub + 2 * t2 + half));
                    mfma_h3(dq[i2], ktf, dsf);
                }
            }
        }
    }
    __syncthreads();
    {
        const float fin = (sds > 0.f) ? a.qscale * hs.inv_sk / sds : 0.f;    // accumulator units -> dQ (incl. the 1/8 of q / 8)
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dq[0][r] *= fin; dq[1][r] *= fin;
            mx = fmaxf(mx, fmaxf(fabsf(dq[0][r]), fabsf(dq[1][r])));
        }
        if (a.amax_dq != nullptr) amax_publish(qg < a.Tq ? mx : 0.f, a.amax_dq, blockIdx.y * gridDim.x + blockIdx.x);
    }
    wave_store_rows(dq, scratch, a.dq + (long)b * a.Tq * a.lddq + h * HD, qw0, a.Tq, a.lddq, lane, 1.f);
}


template <bool CAUSAL>
__global__ __launch_bounds__(256, TTTS_DKVH_W) void attn_bwd_dkv_h3_kernel(AttnArgs a) {
    const uint64_t seed_eff = site_seed(a.seed, a.step_seed);
    const uint32_t thr16 = a.thr << 16;
    extern __shared__ __attribute__((aligned(16))) uint32_t xsd[];
    uint32_t* xs = xsd;
    uint32_t* Qr = xs;                      // [2][32 q][64 d]     f16 hi / lo of Q / 8 * 2^4
    uint32_t* Gr = xs + 2 * XPQ;            //                     of dO * s_g
    uint32_t* Qt = xs + 4 * XPQ;            // [2][64 d][32 q positions]
    uint32_t* Gt = xs + 4 * XPQ + 2 * XPT;
    uint32_t* rawQ = xs + 4 * XPQ + 4 * XPT;   // [32][64] fp32, filled by DMA
    uint32_t* rawG = rawQ + RAWQ;
    float* stat_s = reinterpret_cast<float*>(rawG + RAWQ);  // [2 buffers][m or lse 64 | log2 l 64 | delta 64] (32 of each 64 used)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int kblk = blockIdx.y;
    const int h = blockIdx.x % a.H, b = blockIdx.x / a.H;
    const int k0 = kblk * QB, kw0 = k0 + wave * 32;
    const int kg = kw0 + l31;
    const uint32_t key_mult = attn_drop_mult((uint32_t)kg);
    float* scratch = reinterpret_cast<float*>(xs) + wave * 32 * KT_LD;

    int klen = (int)a.key_lens[b];
    if (klen > a.Tk) klen = a.Tk;
    if (klen < 0) klen = 0;

    const float* qb_ = a.q + (long)b * a.Tq * a.ldq + h * HD;
    const float* kb_ = a.k + (long)b * a.Tk * a.ldk + h * HD;
    const float* vb_ = a.v + (long)b * a.Tk * a.ldv + h * HD;
    const float* gb_ = a.dout + (long)b * a.Tq * a.ldo + h * HD;
    const long arow = ((long)(b * a.H + h) * a.Tq);

    const H3Scales hs = attn_h3_scales(a.q_amax, a.k_amax, a.v_amax, lane, a.qscale, a.do_amax);
    const float s_g = hs.sg, inv_g = hs.inv_sg;
    const float H3A_C2 = hs.c2;
    f16x8v kf[4][2], vf[4][2];
    wave_stage_tile(kb_, kw0, a.Tk, a.ldk, lane, scratch, 1.f);
    wave_lds_sync();
    load_lane_frags_h3(scratch, l31, half, hs.sk, kf);
    wave_lds_sync();
    wave_stage_tile(vb_, kw0, a.Tk, a.ldv, lane, scratch, 1.f);
    wave_lds_sync();
    load_lane_frags_h3(scratch, l31, half, hs.sv, vf);

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

    const float dp_unscale = inv_g * hs.inv_sv * a.drop_scale;  // ... times the 1/(1-p) of kept weights (1 without dropout)
    float sds = 0.f;                        // this key's dS pre-scale (power of two), see attn_bwd_dq_h3_kernel
    const int nqs = (a.Tq + QS - 1) / QS;
    int qs_begin = CAUSAL ? (k0 / QS) : 0;
    if (k0 >= klen) qs_begin = nqs;

    // staging roles: threads 0..127 take Q, 128..255 take dO; each a 4 x 4 patch of the 32 x 64 stage
    const int st_t = tid & 127, rq = st_t >> 4, dqd = st_t & 15;
    const bool st_q = tid < 128;
    const int qpos = perm_pos(rq);

    // DMA of one stage: wave w moves rows 8w..8w+7 of Q and of dO (two 1-KB instructions each), wave 0 also lse / delta.
    // Rows past Tq are clamped to the last row here and zeroed when they are split.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // scalar copy: the DMA destinations must not cost VGPRs
    const int dma_row = 8 * wave + (lane >> 4);
    const uint32_t dma_col = (uint32_t)(lane & 15) * 16u;
    const bool has_rs = a.rowstat != nullptr;       // row statistics in accumulator units (see AttnArgs), else lse alone
    const float* lse_b = (has_rs ? a.rowstat : a.lse) + arow;
    const float* l2_b = has_rs ? a.rowstat + (long)a.B * a.H * a.Tq + arow : a.lse + arow;
    const float* delta_b = a.delta + arow;
    auto fetch = [&](int qt0, int sb) {
        if (wave_u == 0) {     // first: whatever the allocator does to this address must not wait on the big requests
            int q = qt0 + l31;
            if (q > a.Tq - 1) q = a.Tq - 1;
            dma4(lse_b, (uint32_t)q * 4u, reinterpret_cast<uint32_t*>(stat_s + sb * 192));
            dma4(l2_b, (uint32_t)q * 4u, reinterpret_cast<uint32_t*>(stat_s + sb * 192 + 64));
            dma4(delta_b, (uint32_t)q * 4u, reinterpret_cast<uint32_t*>(stat_s + sb * 192 + 128));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int gr = qt0 + dma_row + 4 * j;
            if (gr > a.Tq - 1) gr = a.Tq - 1;
            dma16(qb_, (uint32_t)gr * (uint32_t)(a.ldq * 4) + dma_col, rawQ + (8 * wave_u + 4 * j) * 64);
            dma16(gb_, (uint32_t)gr * (uint32_t)(a.ldo * 4) + dma_col, rawG + (8 * wave_u + 4 * j) * 64);
        }
    };

    __syncthreads();                       // the per-wave scratch (aliasing the planes) is dead from here on
    if (qs_begin < nqs) fetch(qs_begin * QS, qs_begin & 1);

    for (int qs = qs_begin; qs < nqs; ++qs) {
        const int qt0 = qs * QS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the stage have landed
        __syncthreads();                                    // ... and everybody else's; previous planes are free
        {
            const uint32_t* raw = st_q ? rawQ : rawG;
            const float sc = st_q ? a.qscale * hs.sq : s_g;
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * rq + i;
                const u32x4v u = *reinterpret_cast<const u32x4v*>(raw + row * 64 + dqd * 4);
                const bool ok = qt0 + row < a.Tq;
                v[i] = make_float4(ok ? __uint_as_float(u[0]) * sc : 0.f, ok ? __uint_as_float(u[1]) * sc : 0.f,
                                   ok ? __uint_as_float(u[2]) * sc : 0.f, ok ? __uint_as_float(u[3]) * sc : 0.f);
            }
            if (st_q) patch_split_store_h3<true, 32>(v, rq, dqd, qpos, Qr, XPQ, Qt, XPT);
            else patch_split_store_h3<true, 32>(v, rq, dqd, qpos, Gr, XPQ, Gt, XPT);
        }
        const float* lse_s = stat_s + (qs & 1) * 192;
        const float* l2_s = lse_s + 64;
        const float* delta_s = lse_s + 128;
        __syncthreads();
        if (qs + 1 < nqs) fetch(qt0 + QS, (qs + 1) & 1);   // in flight while this stage is multiplied
        if (CAUSAL && qt0 + 31 < kw0) continue;     // every query of the stage precedes this wave's keys (wave-uniform)
        if (kw0 >= klen) continue;                  // this wave's keys are all padding

        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            f16x8v qfr[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) qfr[p] = *reinterpret_cast<const f16x8v*>(Qr + p * XPQ + xsw(l31, 2 * st + half));
            // the same three products in the same order as the forward formed them (there K was the first operand): the
            // accumulators are then bit-identical to the forward's and s - m below is exact
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr[0], kf[st][1], s, 0, 0, 0);      // q_hi k_lo
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr[1], kf[st][0], s, 0, 0, 0);      // q_lo k_hi
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr[0], kf[st][0], s, 0, 0, 0);
        }
        float pd[16];
        const bool full = (kw0 + 32 <= klen) && (!CAUSAL || kw0 + 31 <= qt0) && (qt0 + QS <= a.Tq);   // wave-uniform
        // the row statistics of this lane's 16 queries: registers 4g .. 4g+3 are four consecutive rows, so each array is four
        // 16-byte LDS reads (broadcast within a half-wave), not sixteen dword reads
        float mq_r[16], l2_r[16], dl_r[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 m4 = *reinterpret_cast<const float4*>(lse_s + 8 * g4 + 4 * half);
            const float4 l4 = *reinterpret_cast<const float4*>(l2_s + 8 * g4 + 4 * half);
            const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 8 * g4 + 4 * half);
            mq_r[4 * g4] = m4.x; mq_r[4 * g4 + 1] = m4.y; mq_r[4 * g4 + 2] = m4.z; mq_r[4 * g4 + 3] = m4.w;
            l2_r[4 * g4] = l4.x; l2_r[4 * g4 + 1] = l4.y; l2_r[4 * g4 + 2] = l4.z; l2_r[4 * g4 + 3] = l4.w;
            dl_r[4 * g4] = d4.x; dl_r[4 * g4 + 1] = d4.y; dl_r[4 * g4 + 2] = d4.z; dl_r[4 * g4 + 3] = d4.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q_g = qt0 + acc_row(r, half);
            const float mq = has_rs ? mq_r[r] : 0.f;
            const float l2 = has_rs ? l2_r[r] : l2_r[r] * 1.4426950408889634f;
            float p = fast_exp2(__builtin_fmaf(s[r], H3A_C2, -mq) - l2);
            if (!full) p = (kg < klen && (!CAUSAL || kg <= q_g) && q_g < a.Tq) ? p : 0.f;
            pd[r] = p;
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            f16x8v gfr[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) gfr[p] = *reinterpret_cast<const f16x8v*>(Gr + p * XPQ + xsw(l31, 2 * st + half));
            mfma_h3(dp, gfr, vf[st]);
        }
        float ds[16];
        // dropout: keys 4j .. 4j+3 (a quad of lanes) share one hash word per query row -- lane 4j+i computes it for registers
        // r+i of each group of four, quad-permute DPP broadcasts hand every lane the word of each row, and the lane's own
        // multiplier (its key & 3) picks its 16 bits
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            uint32_t hq[4] = {0u, 0u, 0u, 0u};
            if (a.thr != 0u) {
                const int rr = r + (lane & 3);
                const uint32_t mine = attn_quad_hash(seed_eff, (uint32_t)(arow + qt0 + acc_row(rr, half)), (uint32_t)kg >> 2);
                hq[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]
                hq[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x55, 0xF, 0xF, true);
                hq[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xAA, 0xF, 0xF, true);
                hq[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xFF, 0xF, 0xF, true);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = dp[r + e] * dp_unscale;
                float pk = pd[r + e];
                if (a.thr != 0u) {
                    const bool keep = attn_keep_word(hq[e], key_mult, thr16);
                    g = keep ? g : 0.f;                  // the 1/(1-p) factors ride in dp_unscale and in dv's final scale
                    pk = keep ? pk : 0.f;
                }
                // (one-hot row: exact zero, flagged by the dQ kernel's -0.0 sentinel in delta -- see there)
                ds[r + e] = (__float_as_uint(dl_r[r + e]) == 0x80000000u) ? 0.f : pd[r + e] * (g - dl_r[r + e]);
                pd[r + e] = pk;
            }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]
        attn_h3_track_scale(ds, sds, dk);       // lane-local dS pre-scale (this lane's key), lowered on the fly with dk
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
                for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const f16x8v*>(Gt + p * XPT + xsw4(32 * i2 + l31, 2 * t2 + half));
                mfma_h3(dv[i2], af, f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ds[8 * t2 + e];
            split_frag8_h3(x, sds, f[0], f[1]);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                f16x8v af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const f16x8v*>(Qt + p * XPT + xsw4(32 * i2 + l31, 2 * t2 + half));
                mfma_h3(dk[i2], af, f);
            }
        }
    }
    __syncthreads();
    {
        const float fk = (sds > 0.f) ? hs.inv_sq / sds : 0.f;            // dk accumulator: (Q / 8 * sq)^T (dS * sds)
        const float fv = inv_g / H3A_P * a.drop_scale;                    // dv accumulator: (dO * s_g)^T (kept P * 2^10) / (1-p)
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[0][r] *= fk; dk[1][r] *= fk; dv[0][r] *= fv; dv[1][r] *= fv;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(dk[0][r]), fabsf(dk[1][r]))), fmaxf(fabsf(dv[0][r]), fabsf(dv[1][r])));
        }
        if (a.amax_dkv != nullptr) amax_publish(kg < a.Tk ? mx : 0.f, a.amax_dkv, blockIdx.y * gridDim.x + blockIdx.x);
    }
    wave_store_rows(dk, scratch, a.dk + (long)b * a.Tk * a.lddk + h * HD, kw0, a.Tk, a.lddk, lane, 1.f);
    wave_store_rows(dv, scratch, a.dv + (long)b * a.Tk * a.lddv + h * HD, kw0, a.Tk, a.lddv, lane, 1.f);
}

static int check_common(const char* name, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, float drop_p) {
    TTTS_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "%s: bad dims", name);
    TTTS_REQUIRE((long)B * H < (1L << 31) && cdiv(Tq, QB) <= 65535 && cdiv(Tk, QB) <= 65535, "%s: grid too large", name);
    TTTS_REQUIRE(ldq >= H * HD && ldk >= H * HD && ldv >= H * HD && ldo >= H * HD, "%s: row strides must be >= H*64", name);
    TTTS_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0, "%s: row strides must be multiples of 4", name);
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "%s: bad dropout p", name);
    return TTTS_OK;
}

}  // namespace ttts

using namespace ttts;

template <bool CAUSAL>
static int launch_bwd_x6(const AttnArgs& a, dim3 gq, dim3 gk, hipStream_t stream) {
    static bool configured = false;   // more than 64 KB of LDS per workgroup needs an explicit opt-in, once per kernel
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_x6_kernel<CAUSAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DQX_SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_x6_kernel<CAUSAL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, DKVX_SMEM);
        if (e != hipSuccess) {
            set_error("attention_bwd: cannot reserve %d bytes of LDS: %s", DQX_SMEM, hipGetErrorString(e));
            return TTTS_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL((attn_bwd_dq_x6_kernel<CAUSAL>), gq, dim3(256), DQX_SMEM, stream, a);
    TTTS_LAUNCH_CHECK("attn_bwd_dq_x6_kernel");
    hipLaunchKernelGGL((attn_bwd_dkv_x6_kernel<CAUSAL>), gk, dim3(256), DKVX_SMEM, stream, a);
    TTTS_LAUNCH_CHECK("attn_bwd_dkv_x6_kernel");
    return TTTS_OK;
}

template <bool CAUSAL>
static int launch_bwd_h3(const AttnArgs& a, dim3 gq, dim3 gk, hipStream_t stream) {
    static bool configured = false;   // dynamic LDS sizes are registered once per kernel
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_h3_kernel<CAUSAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DQH_SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_h3_kernel<CAUSAL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, DKVH_SMEM);
        if (e != hipSuccess) {
            set_error("attention_bwd_h3: cannot reserve %d bytes of LDS: %s", DKVH_SMEM, hipGetErrorString(e));
            return TTTS_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL((attn_bwd_dq_h3_kernel<CAUSAL>), gq, dim3(256), DQH_SMEM, stream, a);
    TTTS_LAUNCH_CHECK("attn_bwd_dq_h3_kernel");
    hipLaunchKernelGGL((attn_bwd_dkv_h3_kernel<CAUSAL>), gk, dim3(256), DKVH_SMEM, stream, a);
    TTTS_LAUNCH_CHECK("attn_bwd_dkv_h3_kernel");
    return TTTS_OK;
}

extern "C" {

static int attention_fwd_impl(const float* q, const float* k, const float* v, float* o, float* lse, float* attn,
                              const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                              int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, int form, void* stream_,
                              const float* q_amax = nullptr, const float* k_amax = nullptr, const float* v_amax = nullptr,
                              float* o_amax_out = nullptr, float* rowstat_out = nullptr) {
    // form: 0 = fp32 MFMA, 1 = bf16x6, 2 = fp16x3 (needs the partial maxima of q, k, v)
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(q && k && v && o && key_lens, "attention_fwd: null pointer");
    TTTS_REQUIRE(form != 2 || (q_amax && k_amax && v_amax), "attention_fwd_h3: q_amax / k_amax / v_amax are required");
    int rc = check_common("attention_fwd", B, H, Tq, Tk, ldq, ldk, ldv, ldo, drop_p);
    if (rc) return rc;
    TTTS_REQUIRE(!(causal && attn), "attention_fwd: weights output is only provided for the non-causal (cross) form");
    TTTS_REQUIRE(!causal || Tq == Tk, "attention_fwd: causal form needs Tq == Tk");
    TTTS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "attention_fwd: q/k/v must be 16-byte aligned");
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.attn = attn; a.key_lens = key_lens;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    a.drop_scale = 1.f / (1.f - drop_p);
    a.qscale = q_scale;
    a.seed = seed; a.step_seed = step_seed;
    a.q_amax = q_amax; a.k_amax = k_amax; a.v_amax = v_amax; a.o_amax = o_amax_out; a.rowstat = rowstat_out;
    dim3 grid(B * H, cdiv(Tq, QB), 1);
    if (form == 2) {
        if (causal)
            hipLaunchKernelGGL((attn_fwd_h3_kernel<true, false>), grid, dim3(256), 0, stream, a);
        else if (attn)
            hipLaunchKernelGGL((attn_fwd_h3_kernel<false, true>), grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((attn_fwd_h3_kernel<false, false>), grid, dim3(256), 0, stream, a);
        TTTS_LAUNCH_CHECK("attn_fwd_h3_kernel");
        return TTTS_OK;
    }
    if (form == 1) {
        if (causal)
            hipLaunchKernelGGL((attn_fwd_x6_kernel<true, false>), grid, dim3(256), 0, stream, a);
        else if (attn)
            hipLaunchKernelGGL((attn_fwd_x6_kernel<false, true>), grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((attn_fwd_x6_kernel<false, false>), grid, dim3(256), 0, stream, a);
        TTTS_LAUNCH_CHECK("attn_fwd_x6_kernel");
        return TTTS_OK;
    }
    if (causal)
        hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, dim3(256), 0, stream, a);
    else if (attn)
        hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, dim3(256), 0, stream, a);
    TTTS_LAUNCH_CHECK("attn_fwd_kernel");
    return TTTS_OK;
}

int ttts_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, float* attn,
                       const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                       int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream) {
    return attention_fwd_impl(q, k, v, o, lse, attn, key_lens, B, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, q_scale, drop_p, seed, step_seed, 0,
                              stream);
}
int ttts_attention_fwd_x6(const float* q, const float* k, const float* v, float* o, float* lse, float* attn,
                          const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream) {
    return attention_fwd_impl(q, k, v, o, lse, attn, key_lens, B, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, q_scale, drop_p, seed, step_seed, 1,
                              stream);
}
int ttts_attention_fwd_h3(const float* q, const float* k, const float* v, float* o, float* lse, float* attn,
                          const int64_t* key_lens, int B, int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, const float* q_amax,
                          const float* k_amax, const float* v_amax, float* o_amax_out, float* rowstat_out, void* stream) {
    return attention_fwd_impl(q, k, v, o, lse, attn, key_lens, B, H, Tq, Tk, ldq, ldk, ldv, ldo, causal, q_scale, drop_p, seed, step_seed, 2,
                              stream, q_amax, k_amax, v_amax, o_amax_out, rowstat_out);
}

static int attention_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* do_,
                              const float* lse, float* delta, float* dq, float* dk, float* dv, const int64_t* key_lens, int B,
                              int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                              int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, int form, void* stream_,
                              const float* do_amax = nullptr, float* dq_amax_out = nullptr, float* dkv_amax_out = nullptr,
                              const float* q_amax = nullptr, const float* k_amax = nullptr, const float* v_amax = nullptr,
                              const float* rowstat = nullptr) {
    // form: 0 = fp32 MFMA, 1 = bf16x6, 2 = fp16x3 (needs do_amax and the partial maxima of q, k, v)
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(q && k && v && o && do_ && lse && delta && dq && dk && dv && key_lens, "attention_bwd: null pointer");
    TTTS_REQUIRE(form != 2 || (do_amax && q_amax && k_amax && v_amax),
                 "attention_bwd_h3: do_amax, q_amax, k_amax and v_amax (partial maxima of the operands) are required");
    int rc = check_common("attention_bwd", B, H, Tq, Tk, ldq, ldk, ldv, ldo, drop_p);
    if (rc) return rc;
    TTTS_REQUIRE(lddq >= H * HD && lddk >= H * HD && lddv >= H * HD, "attention_bwd: gradient strides must be >= H*64");
    TTTS_REQUIRE(!causal || Tq == Tk, "attention_bwd: causal form needs Tq == Tk");
    TTTS_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)do_) & 15) == 0,
                 "attention_bwd: q/k/v/o/do must be 16-byte aligned");
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = const_cast<float*>(o); a.dout = do_; a.lse = const_cast<float*>(lse);
    a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.key_lens = key_lens;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    a.drop_scale = 1.f / (1.f - drop_p);
    a.qscale = q_scale;
    a.seed = seed; a.step_seed = step_seed;
    dim3 gq(B * H, cdiv(Tq, QB), 1), gk(B * H, cdiv(Tk, QB), 1);
    if (form == 2) {
        a.do_amax = do_amax; a.do_amax_n = TTTS_AMAX_SLOTS;
        a.amax_dq = dq_amax_out; a.amax_dkv = dkv_amax_out;
        a.q_amax = q_amax; a.k_amax = k_amax; a.v_amax = v_amax;
        a.rowstat = const_cast<float*>(rowstat);
        return causal ? launch_bwd_h3<true>(a, gq, gk, stream) : launch_bwd_h3<false>(a, gq, gk, stream);
    }
    if (form == 1) return causal ? launch_bwd_x6<true>(a, gq, gk, stream) : launch_bwd_x6<false>(a, gq, gk, stream);
    if (causal) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), gq, dim3(256), 0, stream, a);
        TTTS_LAUNCH_CHECK("attn_bwd_dq_kernel");
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true>), gk, dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<false>), gq, dim3(256), 0, stream, a);
        TTTS_LAUNCH_CHECK("attn_bwd_dq_kernel");
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<false>), gk, dim3(256), 0, stream, a);
    }
    TTTS_LAUNCH_CHECK("attn_bwd_dkv_kernel");
    return TTTS_OK;
}

int ttts_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* do_,
                       const float* lse, float* delta, float* dq, float* dk, float* dv, const int64_t* key_lens, int B,
                       int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                       int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream) {
    return attention_bwd_impl(q, k, v, o, do_, lse, delta, dq, dk, dv, key_lens, B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddq, lddk,
                              lddv, causal, q_scale, drop_p, seed, step_seed, 0, stream);
}
int ttts_attention_bwd_x6(const float* q, const float* k, const float* v, const float* o, const float* do_,
                          const float* lse, float* delta, float* dq, float* dk, float* dv, const int64_t* key_lens, int B,
                          int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, void* stream) {
    return attention_bwd_impl(q, k, v, o, do_, lse, delta, dq, dk, dv, key_lens, B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddq, lddk,
                              lddv, causal, q_scale, drop_p, seed, step_seed, 1, stream);
}
int ttts_attention_bwd_h3(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                          const float* lse, float* delta, float* dq, float* dk, float* dv, const int64_t* key_lens, int B,
                          int H, int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                          int causal, float q_scale, float drop_p, uint64_t seed, const uint64_t* step_seed, const float* do_amax,
                          float* dq_amax_out, float* dkv_amax_out, const float* q_amax, const float* k_amax,
                          const float* v_amax, const float* rowstat, void* stream) {
    return attention_bwd_impl(q, k, v, o, d_o, lse, delta, dq, dk, dv, key_lens, B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddq, lddk,
                              lddv, causal, q_scale, drop_p, seed, step_seed, 2, stream, do_amax, dq_amax_out, dkv_amax_out, q_amax, k_amax,
                              v_amax, rowstat);
}

}  // extern "C"
