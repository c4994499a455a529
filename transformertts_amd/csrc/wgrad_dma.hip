// Weight gradients in the fp16x3 form with the operand rows staged by LDS-DMA (the 8-wave 256 x 256 tile):
//     C[n][k] (per tap, per row split) = sum_m dy[m][n] * x[m + shift][k]        (GemmArgs as wgrad_h3_kernel, gemm_h3.hip)
// wgrad_h3_kernel turns both operands in REGISTERS (a thread fetches a 4-column x 8-row block, splits it and writes [column][32 k]
// pieces): eight 16-byte loads held in registers across the products of a step, 4.25 VALU instructions per MFMA, and on the
// 8-wave tile room for only one register set, so every step waits for its loads (round 5 PMC: matrix pipe 34.5 % busy, waves
// 57 % "waiting to issue").  Here
//   * a step is 16 ROWS of dy and of x (one MFMA k-step), fp32, row-major as they lie in memory: global -> LDS by
//     `buffer_load_dwordx4 ... lds` (one instruction = 1 KB = one row of a 256-column tile), requested FOUR steps ahead into a ring
//     of five 32 KB stages -- no staging registers, 96 KB per CU in flight;
//   * the wave that requested a row converts it IN PLACE two steps before its products: the 4 B x 256 raw row becomes {256 f16
//     hi | 256 f16 lo} of value * 2^e (e from the operand's measured maximum, as before) in the same bytes; rows past the end of
//     the operand, and rows of a shifted tap that cross an utterance boundary, are written as zeros here (their DMA source is
//     clamped to a valid row).  The conversion's arithmetic sits between the MFMAs of the step;
//   * both MFMA operands want 8 consecutive ROWS of one column per lane -- a transposed read (`ds_read_b64_tr_b16`: a 16-lane
//     group fetches 4 rows x 16 columns and gets them column-major), two per fragment.  Plane rows are 1 KB apart, a multiple of
//     the 256-byte bank period, so the four rows of a group would collide; the 64-byte block index of a plane row is XORed with
//     (row & 3): the 32 lanes of a half-wave then cover 32 distinct 8-byte slots of one 256-byte period (SQ_LDS_BANK_CONFLICT
//     = 0, profiles/r05_wgrad_dma_pmc.txt);
//   * the fragments of step t+1 are read during the products of step t (a block row's registers as soon as its products are
//     issued), and the wait for the landed rows sits behind the first half of a step's products;
//   * accumulator rows / columns are matrix rows / columns in natural order, so a 32-lane half stores 128 contiguous bytes per
//     partial-sum row (the register-turning kernel scattered 4-byte stores at a 16-byte stride).
// One barrier per step.  Both tile dimensions must divide the output (256 | M and 256 | N); the caller falls back to
// wgrad_h3_kernel otherwise.  What it buys and what it does not (profiles/r05_wgrad_dma*.txt): 8 % fewer busy
// cycles than the register-turning kernel, matrix pipe saturated inside the product phase -- and, run back to back, nearly the
// same time: the socket power cap clocks it down to 1.4 GHz (1.9 GHz with idle gaps between launches; same cycle count).
// Inside the training step, where the cap binds less, the step is 0.1-0.2 ms shorter (same-box A/B).
#include "gemm_common.h"

namespace ttts {

namespace {

typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef short s16x8w __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t lds_addr_w(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// 64 lanes x 16 bytes (descriptor + per-lane byte offset) -> LDS at lds_dst + 16 * lane
__device__ __forceinline__ void dma16w(u32x4 rsrc, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc_w(const void* base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a),
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu)), bytes, 0x00020000u};
}
__device__ __forceinline__ f16x8 tr_frag(uint32_t a0, uint32_t a1) {
    const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)a0);
    const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)a1);
    const s16x8w j = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, j);
}

}  // namespace

// TTTS_WG_ABL (development builds only, results are wrong): 1 = no products, 2 = no conversion, 4 = no requests inside the loop
#ifndef TTTS_WG_ABL
#define TTTS_WG_ABL 0
#endif

#ifdef TTTS_WG_STAMPS
// development aid (tools/wgrad_stamps.py; never defined in the product build): per (workgroup, wave) sums of s_memtime ticks:
// 0 requesting + waiting for landed rows, 1 products + conversion + fragment loads, 2 barrier, 3 whole kernel, 4 steps,
// 5 prologue, 6 epilogue, 7 whole kernel in s_memrealtime ticks (100 MHz)
__device__ unsigned long long ttts_wg_stamps[2048 * 8 * 8];
#define WSTAMP() __builtin_amdgcn_s_memtime()
#define WACC(slot, v) do { st_acc[slot] += (v); } while (0)
#else
#define WSTAMP() 0ull
#define WACC(slot, v)
#endif

// (the kernel's body, shared by the one-problem kernel and the grouped one: workgroup (bx, by, z) of problem g)
template <int BT, int NW>
__device__ __forceinline__ void wgrad_dma_body(const GemmArgs& g, int bx, int by, int z) {
    constexpr int WM = 2, WN = NW / 2;
    constexpr int WTM = BT / WM, WTN = BT / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int KS = 16;                               // rows per step = one MFMA k-step
    constexpr int NST = (BT == 256) ? 5 : 4;             // stages: NST - 2 landing, one being converted, one whose fragments are being loaded
    constexpr int ROWB = BT * 4, HALFB = BT * 2;         // bytes of a raw row = of its two plane rows
    constexpr int OPB = KS * ROWB, STAGE = 2 * OPB;      // one operand / both operands of a step
    constexpr int RPW = KS / NW;                         // rows per wave and operand: 2 (8 waves) or 4 (4 waves)
    constexpr int LPR = ROWB / 16, RPI = 64 / LPR;       // DMA lanes per row, rows per DMA instruction
    constexpr int NI = RPW / RPI;                        // DMA instructions per wave, operand and step
    static_assert(NI == 2 && (BT == 256 || BT == 128) && (NW == 8 || NW == 4), "a wave owns 2 KB of each operand per step");

    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
#ifdef TTTS_WG_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_t0 = WSTAMP(), st_r0 = __builtin_amdgcn_s_memrealtime();
    const int st_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif
    const int m0 = by * BT, n0 = bx * BT;
    const int ztap = (g.ztaps > 1) ? (z % g.ztaps) : 0;
    const int zsplit = (g.ztaps > 1) ? (z / g.ztaps) : z;
    const int nkt = (g.K + HBK - 1) / HBK;
    const int kt_begin = zsplit * g.kt_per_split;
    int kt_end = kt_begin + g.kt_per_split;
    if (kt_end > nkt) kt_end = nkt;
    const int st_begin = kt_begin * (HBK / KS), st_end = kt_end * (HBK / KS);   // 16-row steps of this split
    const int shift = g.shift0 + ztap * g.shift_step;
    // utterance clipping of a shifted tap: row t of an utterance keeps its x row iff 0 <= t + shift < T (T >= 16, see
    // wgrad_dma_supports); without a shift every row passes
    const bool clip = g.T > 0 && shift != 0;
    const int Tm = clip ? g.T : (1 << 30), sh = clip ? shift : 0;

    float a_scale, b_scale, out_scale;
    {
        float a_inv, b_inv;
        h3_operand_scale(g.a_amax, g.a_amax_n, lane, a_scale, a_inv);      // dy
        h3_operand_scale(g.b_amax, g.b_amax_n, lane, b_scale, b_inv);      // x
        out_scale = a_inv * b_inv;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const u32x4 rsA = make_rsrc_w(g.A, g.a_bytes), rsB = make_rsrc_w(g.B, g.b_bytes);
    const uint32_t lds0 = lds_addr_w(smem);
    const uint32_t a_row_bytes = (uint32_t)(g.lda * 4), b_row_bytes = (uint32_t)(g.ldb * 4);
    const int last_row = g.K - 1;

    // ---- request the 16 rows of step st into stage s: this wave's RPW rows of each operand
    const int dma_row = wave * RPW + lane / LPR;                               // + j * RPI
    const uint32_t dma_col_a = (uint32_t)(m0 * 4 + (lane % LPR) * 16), dma_col_b = (uint32_t)(n0 * 4 + (lane % LPR) * 16);
    auto request = [&](int st, int s) {
        const int row0 = st * KS;
        const uint32_t dst = lds0 + (uint32_t)(s * STAGE + wave * RPW * ROWB);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int r = row0 + dma_row + j * RPI;
            const int ra = min(r, last_row), rb = min(max(r + shift, 0), last_row);
            dma16w(rsA, (uint32_t)ra * a_row_bytes + dma_col_a, __builtin_amdgcn_readfirstlane(dst + (uint32_t)(j * RPI * ROWB)));
            dma16w(rsB, (uint32_t)rb * b_row_bytes + dma_col_b, __builtin_amdgcn_readfirstlane(dst + (uint32_t)(OPB + j * RPI * ROWB)));
        }
    };

    // ---- conversion of this wave's landed rows of a stage into planes, in place.  float4 f = i * 64 + lane of the wave's 2 KB:
    // row f / LPR, columns 4 (f % LPR) ..; the hi pairs go to byte 8 (f % LPR) of the row's first half, the lo pairs to the
    // second half, 64-byte blocks XORed with (row & 3).  Every read of the wave precedes its first write (LDS serves a wave in
    // order), and nobody else touches these rows before the barrier.  Split in two so that the arithmetic sits between the
    // products of the step: `fetch` issues the reads, `emit` converts and writes.
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const int cv_row = lane / LPR, cv_c4 = lane % LPR;                         // row within the wave's rows = i * RPI + cv_row
    // `live`: the step exists, i.e. its rows were requested.  A step past the split's end was NOT -- its stage holds whatever an
    // earlier kernel left in LDS, and `garbage * 0` is NaN when the garbage is an infinity or a NaN (round 6: a NaN in one bias
    // gradient of the 768 x 256 case after tests that fill their outputs with NaN had run): such a step reads nothing.
    auto fetch = [&](int s, bool live, float4 (&va)[NI], float4 (&vb)[NI]) {
        const char* base = smem + s * STAGE + wave * RPW * ROWB;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            va[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            vb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (live) {                                     // (wave-uniform)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                va[i] = *reinterpret_cast<const float4*>(base + (i * 64 + lane) * 16);
                vb[i] = *reinterpret_cast<const float4*>(base + OPB + (i * 64 + lane) * 16);
            }
        }
    };
    // `st` = the step the rows belong to, `live` = it exists (the step after the last converts nothing real), tb = position of
    // the wave's first row within its utterance
    auto emit = [&](int st, int s, bool live, int tb, const float4 (&va)[NI], const float4 (&vb)[NI]) {
        char* base = smem + s * STAGE + wave * RPW * ROWB;
        const int row0 = st * KS + wave * RPW + cv_row;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int rl = i * RPI + cv_row;                                   // row within the wave's rows
            const int r = wave * RPW + rl;                                     // row within the stage
            const bool in = live && (row0 + i * RPI) < g.K;
            int t = tb + rl;
            t = (t >= Tm) ? t - Tm : t;
            const bool keep = in && (unsigned)(t + sh) < (unsigned)Tm;
            const float sa = in ? a_scale : 0.f, sb = keep ? b_scale : 0.f, one = in ? 1.f : 0.f;
            csum[0] = fmaf(va[i].x, one, csum[0]); csum[1] = fmaf(va[i].y, one, csum[1]);
            csum[2] = fmaf(va[i].z, one, csum[2]); csum[3] = fmaf(va[i].w, one, csum[3]);
            uint2 ah, al, bh, bl;
#ifdef TTTS_WG_MIX
            split2_scaled(va[i].x, va[i].y, sa, ah.x, al.x);
            split2_scaled(va[i].z, va[i].w, sa, ah.y, al.y);
            split2_scaled(vb[i].x, vb[i].y, sb, bh.x, bl.x);
            split2_scaled(vb[i].z, vb[i].w, sb, bh.y, bl.y);
#else
            split2_pair(f32x2{va[i].x, va[i].y} * sa, ah.x, al.x);
            split2_pair(f32x2{va[i].z, va[i].w} * sa, ah.y, al.y);
            split2_pair(f32x2{vb[i].x, vb[i].y} * sb, bh.x, bl.x);
            split2_pair(f32x2{vb[i].z, vb[i].w} * sb, bh.y, bl.y);
#endif
            char* p = base + rl * ROWB + ((cv_c4 * 8) ^ ((r & 3) << 6));
            *reinterpret_cast<uint2*>(p) = ah;
            *reinterpret_cast<uint2*>(p + HALFB) = al;
            *reinterpret_cast<uint2*>(p + OPB) = bh;
            *reinterpret_cast<uint2*>(p + OPB + HALFB) = bl;
        }
    };

    // ---- fragments of one stage: per 32-column block, hi and lo by two transposed reads each
    const int q4 = (lane & 15) >> 2, pc = lane & 3, g16 = (lane >> 4) & 1;
    const uint32_t frag_row = lds0 + (uint32_t)((8 * half + q4) * ROWB);
    uint32_t off_a[TM], off_b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) off_a[i] = frag_row + (uint32_t)((64 * (wm * TM + i) + 32 * g16 + 8 * pc) ^ (q4 << 6));
#pragma unroll
    for (int j = 0; j < TN; ++j) off_b[j] = frag_row + (uint32_t)(OPB + ((64 * (wn * TN + j) + 32 * g16 + 8 * pc) ^ (q4 << 6)));

    if (st_begin < st_end) {
        constexpr int D = NST - 1;                      // a step's rows are requested D steps before its products
        // position of the wave's first row of the step being converted within its utterance
        int tb = clip ? (st_begin * KS + wave * RPW) % g.T : 0;
        // prologue: D steps requested, the first two converted, the fragments of the first loaded
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (st_begin + d < st_end) request(st_begin + d, d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float4 va[NI], vb[NI];
            fetch(d, st_begin + d < st_end, va, vb);
            emit(st_begin + d, d, st_begin + d < st_end, tb, va, vb);
            tb += KS;
            tb = (tb >= Tm) ? tb - Tm : tb;
        }
        __syncthreads();
        auto load_a = [&](f16x8 (&a)[2][TM], int i, uint32_t so) {
#pragma unroll
            for (int p = 0; p < 2; ++p) a[p][i] = tr_frag(off_a[i] + so + p * HALFB, off_a[i] + so + p * HALFB + 4 * ROWB);
        };
        auto load_b = [&](f16x8 (&b)[2][TN], uint32_t so) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < TN; ++j) b[p][j] = tr_frag(off_b[j] + so + p * HALFB, off_b[j] + so + p * HALFB + 4 * ROWB);
        };
        f16x8 A0[2][TM], B0[2][TN], A1[2][TM], B1[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) load_a(A0, i, 0u);
        load_b(B0, 0u);
        int s = 0;                                      // stage of the step being multiplied
        // One step: products of step st from the fragments in (a, b); meanwhile the rows of step st + 2 (requested two or three
        // steps ago) are converted, the fragments of step st + 1 (converted during the previous step, published by its barrier)
        // are loaded into (an, bn) -- a block row's registers as soon as its products are issued -- and step st + D is requested.
        WACC(5, WSTAMP() - st_t0);
        auto step = [&](int st, f16x8 (&a)[2][TM], f16x8 (&b)[2][TN], f16x8 (&an)[2][TM], f16x8 (&bn)[2][TN]) {
            [[maybe_unused]] const unsigned long long w0 = WSTAMP();
#if !(TTTS_WG_ABL & 4)
            if (st + D < st_end) request(st + D, (s + D) % NST);
#endif
            const int s1 = (s + 1) % NST, s2 = (s + 2) % NST;
            const uint32_t so1 = (uint32_t)(s1 * STAGE);
            auto block_row = [&](int i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
#if TTTS_WG_ABL & 1
                    c[0] += (float)(a[1][i][0] + b[0][j][0] + a[0][i][1] + b[1][j][1]);
#else
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], c, 0, 0, 0);
#endif
                    acc[i][j] = c;
                }
                load_a(an, i, so1);
            };
            // the first half of the products does not need the rows in flight: their wait sits behind it
#pragma unroll
            for (int i = 0; i < TM / 2; ++i) block_row(i);
            __builtin_amdgcn_sched_barrier(0);
            if (st + D < st_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (D - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            [[maybe_unused]] const unsigned long long w1 = WSTAMP();
            float4 va[NI], vb[NI];
#if !(TTTS_WG_ABL & 2)
            fetch(s2, st + 2 < st_end, va, vb);
#endif
#pragma unroll
            for (int i = TM / 2; i < TM; ++i) block_row(i);
#if !(TTTS_WG_ABL & 2)
            emit(st + 2, s2, st + 2 < st_end, tb, va, vb);
#endif
            tb += KS;
            tb = (tb >= Tm) ? tb - Tm : tb;
            load_b(bn, so1);
#ifdef TTTS_WG_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long w2 = WSTAMP();
#endif
            __syncthreads();
#ifdef TTTS_WG_STAMPS
            { const unsigned long long w3 = WSTAMP(); WACC(0, w1 - w0); WACC(1, w2 - w1); WACC(2, w3 - w2); WACC(4, 1); }
#endif
            s = s1;
        };
        for (int st = st_begin; st < st_end; st += 2) {
            step(st, A0, B0, A1, B1);
            if (st + 1 < st_end) step(st + 1, A1, B1, A0, B0);
        }
    }

    [[maybe_unused]] const unsigned long long st_e0 = WSTAMP();
    if (g.colsum != nullptr && bx == 0 && ztap == 0) {
        // csum[e] = this wave's rows of column 4 cv_c4 + e (4-wave tile: lanes l and l + 32 hold the same columns)
        float* fs = reinterpret_cast<float*>(smem);
        if constexpr (LPR == 32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], 32, 64);
        }
        if (lane < LPR) *reinterpret_cast<float4*>(fs + wave * BT + 4 * cv_c4) = make_float4(csum[0], csum[1], csum[2], csum[3]);
        __syncthreads();
        if (tid < BT && m0 + tid < g.M) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += fs[w * BT + tid];
            g.colsum[(long)zsplit * g.M + m0 + tid] = sum;
        }
    }

    float* C = g.C + (long)z * g.c_zstride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WTM + i * 32 + acc_row(r, half);
                C[(long)row * g.ldc + col] = acc[i][j][r] * out_scale;
            }
        }
#ifdef TTTS_WG_STAMPS
    {
        const unsigned long long e = WSTAMP();
        st_acc[6] = e - st_e0; st_acc[3] = e - st_t0; st_acc[7] = __builtin_amdgcn_s_memrealtime() - st_r0;
        if (lane == 0 && st_wg < 2048)
            for (int i = 0; i < 8; ++i) ttts_wg_stamps[(st_wg * 8 + wave) * 8 + i] = st_acc[i];
    }
#endif
}

// XCD-aware numbering, as wgrad_h3_kernel: the tiles (and taps) of one row split read the same rows and share an L2
template <int BT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void wgrad_dma_kernel(GemmArgs g) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int t = xcd_renumber(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * (int)gridDim.z);
    wgrad_dma_body<BT, NW>(g, t % gx, (t / gx) % gy, t / (gx * gy));
}
// GROUPED launch (WgradGroupArgs, gemm_common.h): the 256-wide weight gradients of one decoder layer -- FFN1, FFN2, the packed
// self-attention in-projection -- as ONE grid with the row splits planned for the group: 11 output tiles share the chip's 256
// workgroup slots, 23 splits each instead of 64-85, a third of the partial sums (65 MB per launch before) and of the epilogues.
template <int BT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void wgrad_dma_group_kernel(WgradGroupArgs gg) {
    const int t = xcd_renumber(blockIdx.x, gg.first[gg.n]);
    int p = 0;
#pragma unroll
    for (int i = 1; i < WG_GROUP_MAX; ++i)
        if (i < gg.n && t >= gg.first[i]) p = i;
    p = __builtin_amdgcn_readfirstlane(p);
    const GemmArgs& g = gg.g[p];
    const int local = t - gg.first[p];
    const int gx = g.N / BT, gy = g.M / BT;
    wgrad_dma_body<BT, NW>(g, local % gx, (local / gx) % gy, local / (gx * gy));
}

// shapes the DMA kernel takes: whole 256 x 256 tiles, 16-byte aligned rows, utterances of at least one step (operands the 32-bit
// DMA offsets can reach are checked by the caller)
bool wgrad_dma_supports(const GemmArgs& g, int tile) {
    if (tile != H3_TILE_256) return false;
    return g.M % 256 == 0 && g.N % 256 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && g.K >= 1 && (g.T == 0 || g.T >= 16) &&
           ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15) == 0;
}

int launch_wgrad_dma_group(const GemmArgs* gs, const int* zdims, int n, hipStream_t stream) {
    WgradGroupArgs gg = {};
    if (n < 1 || n > WG_GROUP_MAX) {
        set_error("grouped weight gradient: %d problems (1..%d)", n, WG_GROUP_MAX);
        return TTTS_ERR_INVALID;
    }
    int total = 0;
    for (int i = 0; i < n; ++i) {
        if (!wgrad_dma_supports(gs[i], H3_TILE_256)) {
            set_error("grouped weight gradient: member %d does not take the 256 x 256 LDS-DMA tile", i);
            return TTTS_ERR_INVALID;
        }
        gg.g[i] = gs[i];
        gg.first[i] = total;
        total += (gs[i].N / 256) * (gs[i].M / 256) * zdims[i];
    }
    for (int i = n; i <= WG_GROUP_MAX; ++i) gg.first[i] = total;
    gg.n = n;
    hipLaunchKernelGGL((wgrad_dma_group_kernel<256, 8>), dim3((unsigned)total), dim3(512), 0, stream, gg);
    TTTS_LAUNCH_CHECK("wgrad_dma_group_kernel");
    return TTTS_OK;
}

int launch_wgrad_dma(const GemmArgs& g, int zdim, hipStream_t stream) {
    dim3 grid(g.N / 256, g.M / 256, zdim);
    hipLaunchKernelGGL((wgrad_dma_kernel<256, 8>), grid, dim3(512), 0, stream, g);
    TTTS_LAUNCH_CHECK("wgrad_dma_kernel");
    return TTTS_OK;
}

}  // namespace ttts

#ifdef TTTS_WG_STAMPS
extern "C" int ttts_dbg_wg_read_stamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ttts::ttts_wg_stamps), n * sizeof(unsigned long long));
}
#endif
