// fp16x3 GEMM whose ACTIVATION operand arrives already split ("image" operand): gemm_h3i_kernel.
//
// gemm_h3.hip splits the fp32 activation while it stages it: the operand has to pass through registers (3.5 VALU
// instructions per MFMA in the loader, 64 KB of staging registers per workgroup, ONE k-tile of requests in flight per CU),
// and DESIGN.md section 9.7 shows the three costs of a K = 256 tile -- operand round trips, MFMAs, the 256 KB store burst --
// running one after the other.  Here both operands are bytes:
//   * the activation is an IMAGE written by its producer (LayerNorm forward / backward, ttts_act_image): K16-MAJOR,
//     [K/16][M][64 B], the 64 bytes of (k-tile, row) being 16 f16 "hi" then 16 f16 "lo" of x * 2^e_row -- the same 4 bytes
//     per element as fp32 -- with a PER-ROW power-of-two scale (the row's maximum lands in [2^11, 2^12)); 2^-e_row sits in a
//     float per row.  A row scale of A factors out of C's row, so it is undone where the accumulator leaves the registers (in
//     the transposed accumulator layout a lane IS an output row);
//   * the weight image has the same layout ([K/16][N][64 B]: ttts_weight_split modes 8-11), so every LDS-DMA instruction of
//     either operand moves ONE contiguous KB = eight whole 128-byte lines (a first version read 64-byte pieces of row-major
//     activations and 32-byte pieces of the [K/32][plane][N][32] planes: twice the lines per byte, and no faster than gemm_h3);
//   * both are staged by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no VALU), three 16-deep k-tiles in an
//     LDS ring (3 x 24 KB), two k-tiles in flight behind a COUNTED s_waitcnt vmcnt and one raw s_barrier per k-tile;
//   * the tile is 128 x 256 with 4 waves (64 x 128 per wave, 128 accumulator registers), so TWO workgroups share a CU:
//     one's store burst and LDS turn run under the other's MFMAs -- the overlap a single 256 x 256 workgroup per CU cannot
//     have (vmcnt is one in-order counter per wave: a wave cannot wait for loads issued behind its own stores);
//   * the operand stream does not stop at a tile boundary: the first two k-tiles of the workgroup's next tile are requested
//     (and the first one is read into registers) before the epilogue, so the next tile starts on landed data.
// N = 256 outputs (out-projection, FFN2) keep whole rows inside one tile.
//
// LDS stage (24 KB): activation rows [128][64 B] then weight rows [256][64 B]; a row's 64 bytes are four 16-byte chunks
// {hi k0-7, hi k8-15, lo k0-7, lo k8-15}, chunk index XORed with (row >> 2) & 3 -- the DMA writes lane-linearly, so the
// permutation is applied to the SOURCE address of each lane and again by the fragment reads (conflict-free ds_read_b128).
#include "gemm_common.h"
#include <type_traits>

namespace ttts {

constexpr int IBN = 256, IBK = 16;
#ifndef TTTS_H3I_SPREAD
#define TTTS_H3I_SPREAD 0
#endif
#ifndef TTTS_H3I_BIG_NST
#define TTTS_H3I_BIG_NST 3
#endif
constexpr int I_B_BYTES = IBN * 64;
// TM rows per tile: 128 (4 waves, two workgroups per CU: 24 KB stages) or 256 (8 waves, one workgroup per CU: 32 KB stages -- half
// the weight bytes per flop through the CU's load path, which is what bounds the long-K and wide-N shapes)
template <int TM> struct H3IGeo {
    static constexpr int NW = TM / 32;                  // waves: a wave owns 64 x 128 of the tile
    static constexpr int A_BYTES = TM * 64, STAGE = A_BYTES + I_B_BYTES;
    static constexpr int B_PIECES = IBN / NW / 16;      // 1-KB weight pieces per wave and k-tile
    static constexpr int LOADS = 2 + B_PIECES;          // LDS-DMA instructions per wave and k-tile (2 activation pieces)
    static constexpr int NST = TM == 128 ? 3 : TTTS_H3I_BIG_NST;    // ring stages: NST - 1 k-tiles requested ahead of the one being multiplied
};

__device__ __forceinline__ uint32_t lds_addr_i(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// 64 lanes x 16 bytes from (descriptor, per-lane byte offset + scalar byte offset) to LDS at lds_dst + 16 * lane.  Inline
// assembly: outside hipcc's counter bookkeeping (with the builtin it drains vmcnt(0) before the next LDS read of any address);
// the kernel counts its own.  M0 = LDS destination base, saved and restored inside the statement.
__device__ __forceinline__ void dma16b(u32x4 rsrc, uint32_t voff, uint32_t soff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(N) : "memory");
}

#ifdef TTTS_H3I_STAMPS
// development aid (tools/h3i_stamps.py; never defined in the product build): per (workgroup, wave) sums of s_memtime ticks spent
// 0 waiting for a k-tile (vmcnt + barrier), 1 issuing the next k-tile's DMAs, 2 fragment reads + products, 3 epilogue,
// 4 whole kernel, 5 tiles done, 6 k-tiles done, 7 the wait of each tile's FIRST k-tile
__device__ unsigned long long ttts_h3i_stamps[512 * 8 * 8];
#define ISTAMP() __builtin_amdgcn_s_memtime()
#define IACC(slot, v) do { if ((threadIdx.x & 63) == 0) st_acc[slot] += (v); } while (0)
#else
#define ISTAMP() 0ull
#define IACC(slot, v)
#endif

// HAS_RES / HAS_GATE / DROP: the epilogue's optional operands are TEMPLATE parameters of the kernel, not branches inside it.  With
// the six variants behind run-time branches in one kernel, hipcc's s_waitcnt insertion merged their counter states at the tile
// loop's head and put `s_waitcnt vmcnt(0)` in front of the main loop's first fragment read -- a wait for the DMAs issued a
// few instructions earlier, i.e. no prefetch at all (2.2k instead of 1.1k cycles per k-tile).
// A_RAW: the activation is plain fp32 (row-major, row stride lda) with a per-TENSOR scale from its partial maxima, as gemm_h3
// takes it -- no producer has to write an image.  The raw 16-deep k-tile (128 rows x 64 B) is DMA'd into the stage's activation
// area exactly like an image k-tile; the wave that requested a 32-row piece converts it IN PLACE once its own requests have
// landed (two 16-byte reads per lane, the split, two 16-byte writes: the raw row and its {hi, lo} planes are the same 64 bytes),
// in front of the k-tile's barrier.  ~1.3 VALU instructions per MFMA instead of the 3.5 of gemm_h3's loader, no staging
// registers, and everything global still moves by LDS-DMA, so the counted waits stay within one completion order.
// IMG: the output leaves as a HEAD IMAGE (GemmArgs::c_row_inv): attention is the only reader of an in-projection's output, and it
// wants f16 hi / lo planes it can stage by LDS-DMA -- the same 4 bytes per element as fp32, written here instead of fp32, with a
// power-of-two scale per (row, 64-column head) because the tile owns whole head rows (bias only; no residual / gate / dropout).
#ifdef TTTS_CLOCK_STAMPS
__device__ unsigned long long ttts_clock_h3i[2 * 512];
#endif
template <int TM, bool A_RAW, bool HAS_RES, bool HAS_GATE, bool DROP, bool IMG = false>
__global__ __launch_bounds__(TM * 2, TM == 128 ? 2 : 1) void gemm_h3i_kernel(GemmArgs g) {
    using Geo = H3IGeo<TM>;
    constexpr int IBM = TM, I_A_BYTES = Geo::A_BYTES, I_STAGE = Geo::STAGE, I_LOADS = Geo::LOADS, NW = Geo::NW, INST = Geo::NST;
    constexpr int I_AHEAD = (INST - 2) * I_LOADS;        // requests that may still be in flight when the NEXT k-tile must have landed
    static_assert(!IMG || TM == 128, "the head-image epilogue parks its bias behind four slabs of a 24 KB stage");
    TTTS_CLOCK_BEGIN();
#ifdef TTTS_H3I_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_begin = ISTAMP();
#endif
    __shared__ __attribute__((aligned(16))) uint32_t lds[INST * I_STAGE / 4];
    const uint64_t seed_eff = site_seed(g.seed, g.step_seed);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int nkt = g.K / IBK;
    const int nx = (g.N + IBN - 1) / IBN;
    const int ntiles = nx * ((g.M + IBM - 1) / IBM);
    float w_inv, a_scale = 1.f, a_inv = 1.f;
    {
        float w_scale;
        h3_operand_scale(g.b_amax, g.b_amax_n, lane, w_scale, w_inv);       // the planes already carry w_scale
        if (A_RAW) h3_operand_scale(g.a_amax, g.a_amax_n, lane, a_scale, a_inv);
        asm volatile("" :: "s"(w_inv), "s"(a_scale));     // the wait for these loads sits HERE, in front of the first DMA
    }
    auto tile_coords = [&](int bid, int& m0, int& n0) {     // XCD-aware numbering of gemm_h3_kernel: column tiles of a row panel share an L2
        const int xcd = bid & 7, slot = bid >> 3;
        const int per = ntiles >> 3, rem = ntiles & 7;
        const int t = xcd * per + min(xcd, rem) + slot;
        const int ty = t / nx;
        m0 = ty * IBM;
        n0 = (t - ty * nx) * IBN;
    };
    // ---- loader: lane -> (row r16 of a 16-row piece, LDS chunk slot); the chunk it FETCHES is slot ^ swizzle(row)
    // Both images are K16-MAJOR: [k / 16][row][64 B], so the 16 rows of a piece are ONE contiguous KB (eight whole lines).
    const int r16 = lane >> 2, c = (lane & 3) ^ ((r16 >> 2) & 3);
    const uint32_t lane_off = (uint32_t)r16 * 64u + (uint32_t)c * 16u;
    // (A_RAW: fp32 rows of lda floats, the lane's chunk unswizzled -- the in-place conversion applies the swizzle)
    const uint32_t a_row_bytes = (uint32_t)g.lda * 4u;
    const uint32_t a_lane_off = A_RAW ? (uint32_t)r16 * a_row_bytes + (uint32_t)(lane & 3) * 16u : lane_off;
    const uint32_t a_kstep = A_RAW ? 64u : (uint32_t)g.M * 64u, b_kstep = (uint32_t)g.N * 64u;     // one k-tile further
    const uint32_t lds0 = lds_addr_i(lds);
    const uint64_t a_base = reinterpret_cast<uint64_t>(g.A), b_base = reinterpret_cast<uint64_t>(g.B);
    int ld_bid = blockIdx.x, ld_kt = 0, ld_stage = 0;
    uint32_t ld_a0 = 0, ld_b0 = 0, ld_abytes = 0, ld_bbytes = 0;
    auto set_ld_tile = [&](int bid) {       // (no tile left: empty descriptors -- the loads return without touching memory, and the
        const bool live = bid < ntiles;     //  k-tile loop needs no variant that stops requesting)
        int m0, n0;
        tile_coords(live ? bid : 0, m0, n0);
        ld_a0 = A_RAW ? (uint32_t)((long)(m0 + wave * 32) * a_row_bytes) : (uint32_t)(m0 + wave * 32) * 64u;
        ld_b0 = (uint32_t)(n0 + wave * (IBN / NW)) * 64u;
        ld_abytes = live ? g.a_bytes : 0u;
        ld_bbytes = live ? g.b_bytes : 0u;
    };
    // piece e of the I_LOADS requests of the loader's current k-tile (0, 1: activation rows; 2 ..: weight rows)
    auto issue_piece = [&](int e) {
        const uint32_t dst = lds0 + (uint32_t)ld_stage * I_STAGE;
        if (e < 2) {
            const u32x4 rsrcA = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xffffu, ld_abytes, 0x00020000u};
            const uint32_t a_s = ld_a0 + (uint32_t)ld_kt * a_kstep;
            dma16b(rsrcA, a_lane_off, __builtin_amdgcn_readfirstlane(a_s + e * (A_RAW ? 16u * a_row_bytes : 1024u)),
                   __builtin_amdgcn_readfirstlane(dst + (uint32_t)(2 * wave + e) * 1024u));
        } else {
            const u32x4 rsrcB = {(uint32_t)b_base, (uint32_t)(b_base >> 32) & 0xffffu, ld_bbytes, 0x00020000u};
            const uint32_t b_s = ld_b0 + (uint32_t)ld_kt * b_kstep;
            dma16b(rsrcB, lane_off, __builtin_amdgcn_readfirstlane(b_s + (e - 2) * 1024u),
                   __builtin_amdgcn_readfirstlane(dst + I_A_BYTES + (uint32_t)(Geo::B_PIECES * wave + (e - 2)) * 1024u));
        }
    };
    auto issue_advance = [&]() {
        ld_stage = ld_stage == INST - 1 ? 0 : ld_stage + 1;
        if (++ld_kt == nkt) {
            ld_kt = 0;
            ld_bid += gridDim.x;
            set_ld_tile(ld_bid);
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int e = 0; e < I_LOADS; ++e) issue_piece(e);
        issue_advance();
    };
    // ---- fragments: lane -> row l31 of a 32-row block, chunk (plane * 2 + half) ^ swizzle
    const int sw = (l31 >> 2) & 3;
    const uint32_t ch_hi = (uint32_t)((half ^ sw) * 16), ch_lo = (uint32_t)(((2 + half) ^ sw) * 16);
    const uint32_t fa_off = (uint32_t)(wm * 64 + l31) * 64u;
    const uint32_t fb_off = I_A_BYTES + (uint32_t)(wn * 128 + l31) * 64u;
    f32x16 acc[2][4];
    // Two fragment sets: while the products of k-tile t run from one set, the fragments of k-tile t+1 are read into the other, so
    // a wave has no LDS latency in front of its first MFMA and a workgroup that is ALONE on the matrix pipe (its partner is in
    // its epilogue) still keeps it busy.  One iteration:
    //     request k-tile t+2  ->  first half of the products of t  ->  wait: t+1 has landed (counted vmcnt) + barrier
    //     ->  read the fragments of t+1  ->  second half of the products of t (the reads land under them).
    // The barrier sits between every wave's last read of stage (t-1) % 3 (issued an iteration ago, retired by the lgkmcnt(0) in
    // front of it) and the next iteration's request that overwrites that stage.
    f16x8 fa[2][2][2], fb[2][2][4];                                 // [set][plane][block]
    auto read_frags = [&](int set, int stage) {
        const char* st = reinterpret_cast<const char*>(lds) + stage * I_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fb[set][0][j] = *reinterpret_cast<const f16x8*>(st + fb_off + j * 2048 + ch_hi);
            fb[set][1][j] = *reinterpret_cast<const f16x8*>(st + fb_off + j * 2048 + ch_lo);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[set][0][i] = *reinterpret_cast<const f16x8*>(st + fa_off + i * 2048 + ch_hi);
            fa[set][1][i] = *reinterpret_cast<const f16x8*>(st + fa_off + i * 2048 + ch_lo);
        }
    };
    // (SPREAD: the k-tile's requests go out BETWEEN the products of the first row block -- an LDS-DMA instruction takes 70-90 cycles
    // to issue (tools/h3i_stamps.py), which the wave otherwise spends in front of its first MFMA with the matrix pipe idle)
    auto products = [&](int set, int i, bool spread = false) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the WEIGHT fragment is the MFMA's first operand: the accumulator block is C^T (lane = output row, registers
            // 4q .. 4q+3 = four consecutive output columns); small terms first, as gemm_h3_kernel
            f32x16 cc = acc[i][j];
            cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[set][0][j], fa[set][1][i], cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[set][1][j], fa[set][0][i], cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[set][0][j], fa[set][0][i], cc, 0, 0, 0);
            acc[i][j] = cc;
            if (spread) {       // (pinned: left alone, hipcc gathers the requests in front of the second MFMA)
                constexpr int PER = (I_LOADS + 3) / 4;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < PER; ++c)
                    if (j * PER + c < I_LOADS) issue_piece(j * PER + c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // A_RAW: lane -> (row 32 * wave + lane / 2, k-half lane & 1) of the wave's own piece: raw chunks 2h, 2h+1 -> chunks h (hi), 2 + h (lo)
    auto convert_rows = [&](int stage) {
        const int row = wave * 32 + (lane >> 1), h = lane & 1, swr = (row >> 2) & 3;
        char* p = reinterpret_cast<char*>(lds) + stage * I_STAGE + row * 64;
        const float4 x0 = *reinterpret_cast<const float4*>(p + h * 32), x1 = *reinterpret_cast<const float4*>(p + h * 32 + 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every lane has its raw values before any lane overwrites the row
        __builtin_amdgcn_wave_barrier();
        u32x4 hi, lo;
        uint32_t hh, ll;
        split2_pair(f32x2{x0.x, x0.y} * a_scale, hh, ll); hi.x = hh; lo.x = ll;
        split2_pair(f32x2{x0.z, x0.w} * a_scale, hh, ll); hi.y = hh; lo.y = ll;
        split2_pair(f32x2{x1.x, x1.y} * a_scale, hh, ll); hi.z = hh; lo.z = ll;
        split2_pair(f32x2{x1.z, x1.w} * a_scale, hh, ll); hi.w = hh; lo.w = ll;
        *reinterpret_cast<u32x4*>(p + ((h ^ swr) * 16)) = hi;
        *reinterpret_cast<u32x4*>(p + (((2 + h) ^ swr) * 16)) = lo;
    };
    int cstage = 0;                 // stage of the k-tile whose fragments sit in set 0 at the top of an (even) iteration
    bool after_ep = false;
    // one k-tile: `set` holds its fragments (compile-time), the other set receives the next k-tile's
    auto ktile = [&](auto set_c, bool first_after_ep, bool landed) {
        constexpr int SET = decltype(set_c)::value;
        [[maybe_unused]] const unsigned long long s0 = ISTAMP();
        if (first_after_ep) asm volatile("s_barrier" ::: "memory");     // every wave has left its epilogue slab (this request's target)
#if TTTS_H3I_SPREAD
        [[maybe_unused]] const unsigned long long s1 = ISTAMP();
        products(SET, 0, true);
        issue_advance();
#else
        issue();
        [[maybe_unused]] const unsigned long long s1 = ISTAMP();
        products(SET, 0);
#endif
        [[maybe_unused]] const unsigned long long s2 = ISTAMP();
        // k-tile t+1 has landed when all but the I_LOADS requests behind it have: LDS-DMAs complete in the order they were issued
        // AMONG THEMSELVES, so "at most I_LOADS operations outstanding" implies it whatever else (stores) is still in flight.
        // The first INST - 2 k-tiles after an epilogue need no count at all (`landed`): the epilogue began with vmcnt(0) (see there),
        // so the wait that also covers the epilogue's stores comes INST - 2 k-tiles behind them.
        if (A_RAW) {
            // this wave's own requests for k-tile t+1 have landed: convert its 32 raw rows in place, then meet the others
            if (!landed) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(I_AHEAD) : "memory");
            convert_rows(cstage == INST - 1 ? 0 : cstage + 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (landed) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "i"(I_AHEAD) : "memory");
        [[maybe_unused]] const unsigned long long s3 = ISTAMP();
        const int nstage = cstage == INST - 1 ? 0 : cstage + 1;
        read_frags(SET ^ 1, nstage);
        products(SET, 1);
        cstage = nstage;
        IACC(0, s3 - s2); IACC(1, s1 - s0); IACC(2, (s2 - s1) + (ISTAMP() - s3)); IACC(6, 1);
    };

    set_ld_tile(ld_bid);
#pragma unroll
    for (int e = 0; e < INST - 1; ++e) issue();     // the first k-tiles of the first tile (the requests roll over into the next tile by themselves)
    if (A_RAW) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(I_AHEAD) : "memory");
        convert_rows(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        wait_vm_barrier<I_AHEAD>();
    }
    read_frags(0, 0);
    for (int bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
        int m0, n0;
        tile_coords(bid, m0, n0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nkt; kt += 2) {                        // (K % 32 == 0: an even number of k-tiles)
            ktile(std::integral_constant<int, 0>{}, after_ep && kt == 0, after_ep && kt < INST - 2);
            ktile(std::integral_constant<int, 1>{}, false, after_ep && kt + 1 < INST - 2);
        }
        [[maybe_unused]] const unsigned long long s3 = ISTAMP();
        // cstage holds the NEXT tile's first k-tile (its fragments are in set 0 already), the stage behind it the second; the
        // stage before it is the one the last products ran on: every wave passed the last barrier after reading it -> the slabs
        const int ep_stage = cstage == 0 ? INST - 1 : cstage - 1;
        // LDS-DMAs and ordinary loads / stores do NOT complete in one common order (measured: with the next tile's k-tile still
        // in flight, hipcc's counted `s_waitcnt vmcnt(4)` in front of a residual / gate operand -- correct if everything retired
        // in issue order -- let the operand be read before it had arrived: a few hundred wrong elements per launch, only where a
        // workgroup has a next tile).  So no DMA is in flight while the epilogue's own loads and stores are counted: this wait,
        // for a k-tile requested a whole k-tile ago, costs little; the next tile's first k-tile is in registers already.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ---------------- epilogue: each 32 x 32 accumulator block is scaled by its lane's row factor, turned through a 4 KB slab
        // (row-major, 16-byte chunk ^= row & 7: conflict-free both ways) and leaves as whole 128-byte lines.  A block's auxiliary
        // operands (residual, relu gate) are requested BEHIND the previous block's stores and waited for with everything older
        // (see the note at the loads); the variants without auxiliary operands issue no load at all.
        {
            float* slab = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + ep_stage * I_STAGE) + wave * 1024;
            const int rsub = lane >> 3, ch = lane & 7;
            const int col0 = n0 + wn * 128 + ch * 4;
            const bool want_max = g.c_amax != nullptr;
            const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (uint32_t)((long)g.M * g.ldc * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rsrcBias = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(g.bias != nullptr ? g.bias : g.A), 0, g.bias != nullptr ? (uint32_t)g.N * 4u : 0u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(A_RAW ? g.A : g.a_row_inv), 0, A_RAW ? 0u : (uint32_t)g.M * 4u, 0x00020000);
            float rs[2];
            float4 bias4[4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                rs[i] = A_RAW ? a_inv * w_inv
                              : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcS, (m0 + wm * 64 + i * 32 + l31) * 4, 0, 0)) * w_inv;
#pragma unroll
            for (int j = 0; j < 4; ++j) bias4[j] = buf_load4(rsrcBias, (col0 + j * 32 < g.N) ? (uint32_t)(col0 + j * 32) * 4u : OOB);
            const float relu_lo = (g.act == 1) ? 0.f : -__builtin_inff();
            float cmax = 0.f;
            // Addressing as gemm_h3's fast epilogue: ONE per-lane 32-bit byte offset per column group (out of range when the group
            // lies past N) plus a wave-uniform scalar offset per row group -- rows past M fall outside the descriptors, so there
            // is no row test, no 64-bit arithmetic and no branch around a load or a store (output, residual and gate operand
            // share the row stride: ldr == ldc at both entry points).
            const int row_l = m0 + wm * 64 + rsub;                       // this lane's row in row group 0
            const int left = g.M - row_l;
            const int rows_left = left < 0 ? 0 : (left > IBM ? IBM : left);
            uint32_t offC[4];
            int live_rows[4];                                            // row groups rg < live_rows[j] hold stored values
            uint64_t idx0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = col0 + j * 32;
                const bool ok = col < g.N;
                offC[j] = ok ? (uint32_t)(((long)row_l * g.ldc + col) * 4) : OOB;
                live_rows[j] = ok ? rows_left : 0;
                idx0[j] = (uint64_t)row_l * (uint64_t)g.N + (uint64_t)col;
            }
            const uint32_t row_step = (uint32_t)g.ldc * 4u;
            if constexpr (IMG) {
                // lane -> (row r16 of a 16-row pass, 8-column group c8 of a 32-column block): after the slab turn a lane holds
                // 8 consecutive columns of its row in each of the head's two blocks, so the row's maximum over the head is the
                // lane's 16 values and two quad-permute steps, and each plane leaves in 16-byte pieces (4 lanes = 64 bytes of
                // a head's 128-byte plane row)
                const int r16 = lane >> 2, c8 = lane & 3;
                const __amdgpu_buffer_rsrc_t rsrcI = __builtin_amdgcn_make_buffer_rsrc(g.c_row_inv, 0, (uint32_t)((long)(g.N / 64) * g.M * 4), 0x00020000);
                // the wave's 128 bias values wait in LDS (behind the four slabs of the stage): 32 registers less than holding them
                float* bias_s = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + ep_stage * I_STAGE + 16384) + wave * 128;
                if (lane < 32) {
                    const int col = n0 + wn * 128 + lane * 4;
                    *reinterpret_cast<float4*>(bias_s + lane * 4) = buf_load4(rsrcBias, col < g.N ? (uint32_t)col * 4u : OOB);
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                const int row_i = m0 + wm * 64 + r16;                    // this lane's row in pass 0 of row block 0
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int hcol = n0 + wn * 128 + g2 * 64;        // first column of this head (wave-uniform)
                        float v[2][2][8];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * g2 + jj;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                *reinterpret_cast<float4*>(slab + l31 * 32 + (((2 * q + half) ^ (l31 & 7)) * 4)) =
                                    make_float4(acc[i][j][4 * q] * rs[i], acc[i][j][4 * q + 1] * rs[i], acc[i][j][4 * q + 2] * rs[i],
                                                acc[i][j][4 * q + 3] * rs[i]);
                            __builtin_amdgcn_wave_barrier();
                            asm volatile("" ::: "memory");
                            const float4 b0 = *reinterpret_cast<const float4*>(bias_s + j * 32 + c8 * 8);
                            const float4 b1 = *reinterpret_cast<const float4*>(bias_s + j * 32 + c8 * 8 + 4);
#pragma unroll
                            for (int ps = 0; ps < 2; ++ps) {
                                const int srow = ps * 16 + r16;
                                const float4 a0 = *reinterpret_cast<const float4*>(slab + srow * 32 + (((2 * c8) ^ (srow & 7)) * 4));
                                const float4 a1 = *reinterpret_cast<const float4*>(slab + srow * 32 + (((2 * c8 + 1) ^ (srow & 7)) * 4));
                                v[jj][ps][0] = a0.x + b0.x; v[jj][ps][1] = a0.y + b0.y; v[jj][ps][2] = a0.z + b0.z; v[jj][ps][3] = a0.w + b0.w;
                                v[jj][ps][4] = a1.x + b1.x; v[jj][ps][5] = a1.y + b1.y; v[jj][ps][6] = a1.z + b1.z; v[jj][ps][7] = a1.w + b1.w;
                            }
                            __builtin_amdgcn_wave_barrier();
                            asm volatile("" ::: "memory");
                        }
                        float hmax = 0.f;                                 // max|y| over this wave's part of the head's section
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            float m = 0.f;
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[jj][ps][e]));
                            m = fmaxf(m, __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(m), 0xB1, 0xF, 0xF, true)));   // lane ^ 1
                            m = fmaxf(m, __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(m), 0x4E, 0xF, 0xF, true)));   // lane ^ 2
                            float sc, inv;
                            h3_pow2_scale(m, sc, inv);
                            const int row = row_i + i * 32 + ps * 16;
                            const bool live = row < g.M && hcol < g.N;
                            hmax = live ? fmaxf(hmax, m) : hmax;
                            const uint32_t soff = (uint32_t)(i * 32 + ps * 16) * row_step;
                            const uint32_t loff = live ? (uint32_t)(((long)(m0 + wm * 64 + r16) * g.ldc + hcol) * 4) + (uint32_t)c8 * 16u : OOB;
                            u32x4 hi[2], lo[2];
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                uint32_t hh, ll;
                                split2_pair(f32x2{v[jj][ps][0], v[jj][ps][1]} * sc, hh, ll); hi[jj].x = hh; lo[jj].x = ll;
                                split2_pair(f32x2{v[jj][ps][2], v[jj][ps][3]} * sc, hh, ll); hi[jj].y = hh; lo[jj].y = ll;
                                split2_pair(f32x2{v[jj][ps][4], v[jj][ps][5]} * sc, hh, ll); hi[jj].z = hh; lo[jj].z = ll;
                                split2_pair(f32x2{v[jj][ps][6], v[jj][ps][7]} * sc, hh, ll); hi[jj].w = hh; lo[jj].w = ll;
                            }
                            // (a live row's offset stays inside the descriptor: no row test; dead rows / heads carry OOB)
                            const uint32_t so = live ? loff : OOB;
                            __builtin_amdgcn_raw_buffer_store_b128(hi[0], rsrcC, (int)so, (int)soff, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(hi[1], rsrcC, (int)(live ? loff + 64u : OOB), (int)soff, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(lo[0], rsrcC, (int)(live ? loff + 128u : OOB), (int)soff, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(lo[1], rsrcC, (int)(live ? loff + 192u : OOB), (int)soff, 0);
                            // STORE-DATA HAZARD (gemm_common.h): the data registers of these stores are recycled by the next
                            // row's packed multiplies (and, in some instantiations, as the next store's ADDRESS register in the very next
                            // instruction), and hipcc pads no wait states behind a wide buffer store whose soffset is an SGPR -- every
                            // store of rows i, ps != 0, 0 here.  Round 5 met it as ~0.15 % of stores whose second dword was the next
                            // product.  The statement keeps all four registers live and untouched behind the last store.
                            asm volatile("s_nop 7" :: "v"(hi[0]), "v"(hi[1]), "v"(lo[0]), "v"(lo[1]) : "memory");
                            if (c8 == 0)
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(inv), rsrcI,
                                                                     (int)(live ? (uint32_t)(((long)(hcol >> 6) * g.M + row) * 4) : OOB), 0, 0);
                        }
                        if (want_max && hcol < g.N) {       // (a head past the last column belongs to no section: wave-uniform test)
                            const int sec = g.c_amax_sec > 0 ? hcol / g.c_amax_sec : 0;
                            amax_publish(hmax, g.c_amax + (long)sec * TTTS_AMAX_SLOTS, bid);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(HAS_RES ? g.residual : g.A), 0, HAS_RES ? (uint32_t)((long)g.M * g.ldr * 4) : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(HAS_GATE ? g.relu_out : g.A), 0, HAS_GATE ? (uint32_t)((long)g.M * g.ldc * 4) : 0u, 0x00020000);
                // Auxiliary operands (residual, relu gate) are requested block by block, behind the previous block's stores, and
                // waited for with everything older.  Requesting them AHEAD of those stores and counting the stores as
                // "younger" (`s_waitcnt vmcnt(4)`, which is what hipcc emits for that order) is not safe on this chip: stores
                // retire ahead of older loads under load -- measured, a few hundred operands per launch read before they had
                // arrived (tools/h3i_repro.py) -- so a count can only be trusted to mean "everything older than the stores".
                constexpr int GB = 1;                    // blocks per group (two: 32 more registers per operand kind -> spills)
                float4 r4[HAS_RES ? 4 * GB : 1], g4[HAS_GATE ? 4 * GB : 1];
#pragma unroll
                for (int pair = 0; pair < 8 / GB; ++pair) {
                    const int i = (pair * GB) >> 2;
                    if (HAS_RES || HAS_GATE) {
#pragma unroll
                        for (int b2 = 0; b2 < GB; ++b2)
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int j = (pair * GB + b2) & 3;
                                const uint32_t soff = (uint32_t)(i * 32 + u * 8) * row_step;
                                if (HAS_RES) r4[b2 * 4 + u] = buf_load4s(rsrcR, offC[j], soff);
                                if (HAS_GATE) g4[b2 * 4 + u] = buf_load4s(rsrcG, offC[j], soff);
                            }
                    }
#pragma unroll
                    for (int b2 = 0; b2 < GB; ++b2) {
                        const int j = (pair * GB + b2) & 3;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(slab + l31 * 32 + (((2 * q + half) ^ (l31 & 7)) * 4)) =
                                make_float4(acc[i][j][4 * q] * rs[i], acc[i][j][4 * q + 1] * rs[i], acc[i][j][4 * q + 2] * rs[i],
                                            acc[i][j][4 * q + 3] * rs[i]);
                        // (LDS instructions of one wave execute in order: no wait between its writes and its reads, only the
                        // compiler must keep them in order)
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int srow = u * 8 + rsub;
                            const int rg = i * 32 + u * 8;                   // row group, wave-uniform
                            const float4 a4 = *reinterpret_cast<const float4*>(slab + srow * 32 + ((ch ^ (srow & 7)) * 4));
                            float v[4] = {a4.x + bias4[j].x, a4.y + bias4[j].y, a4.z + bias4[j].z, a4.w + bias4[j].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], relu_lo);
                            if (DROP) {
                                bool kp[4];
                                keep_quad(seed_eff, idx0[j] + (uint64_t)rg * (uint64_t)g.N, g.drop_thr, kp);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] *= kp[e] ? g.drop_scale : 0.f;        // (a product, as torch's dropout)
                            }
                            if (HAS_GATE) {
                                const float4 gq = g4[HAS_GATE ? b2 * 4 + u : 0];
                                const float gg[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] *= gg[e] > 0.f ? g.relu_scale : 0.f;
                            }
                            if (HAS_RES) {
                                const float4 rq = r4[HAS_RES ? b2 * 4 + u : 0];
                                v[0] += rq.x; v[1] += rq.y; v[2] += rq.z; v[3] += rq.w;
                            }
                            if (want_max) {          // branch-free: two v_max3, a compare and a select
                                float mx = fmaxf(fmaxf(cmax, fabsf(v[0])), fabsf(v[1]));
                                mx = fmaxf(fmaxf(mx, fabsf(v[2])), fabsf(v[3]));
                                cmax = rg < live_rows[j] ? mx : cmax;
                            }
                            // unconditional: see the addressing note above
                            buf_store4s(rsrcC, offC[j], (uint32_t)rg * row_step, make_float4(v[0], v[1], v[2], v[3]));
                        }
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (!IMG && want_max) amax_publish(cmax, g.c_amax, bid);
        }
        after_ep = true;
        IACC(3, ISTAMP() - s3); IACC(5, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // requests made for a tile that does not exist must not outlive the wave
    TTTS_CLOCK_END(ttts_clock_h3i, 512);
#ifdef TTTS_H3I_STAMPS
    st_acc[4] = ISTAMP() - st_begin;
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 512)
        for (int i = 0; i < 8; ++i) ttts_h3i_stamps[(blockIdx.x * NW + (threadIdx.x >> 6)) * 8 + i] = st_acc[i];
#endif
}

bool h3i_supports(const GemmArgs& g) {
    return (g.a_row_inv != nullptr || g.a_amax != nullptr) && g.K % 32 == 0 && g.K >= 32 && g.N % 4 == 0 && g.T <= 0 && g.bn_ws == nullptr &&
           (uint64_t)g.N * g.K * 4 < (1ull << 32);
}

// The 256-row tile (one 8-wave workgroup per CU) where it was measured ahead of the 128-row one AND of gemm_h3's 256 x 256 tile
// (tools/h3i_bench.py): never for head-image outputs (their epilogue is written for the 24 KB stage).
#ifndef TTTS_H3I_BIG
#define TTTS_H3I_BIG 1
#endif
#ifndef TTTS_H3I_BIG_GATE
#define TTTS_H3I_BIG_GATE 1      // the relu-gated data gradient (256 -> 1024) on the 256-row tile too (step -0.02 ms same-box, 3 / 3)
#endif
#ifndef TTTS_H3I_BIG_MAXK
#define TTTS_H3I_BIG_MAXK 256
#endif
static bool h3i_big_tile(const GemmArgs& g) {
#ifdef TTTS_H3I_BIG_ANY          // development A/B (tools/h3i_bench.py): every shape with enough tiles
    if (g.c_row_inv == nullptr && (long)cdiv(g.N, IBN) * cdiv(g.M, 256) >= 128) return true;
#endif
    return TTTS_H3I_BIG != 0 && g.c_row_inv == nullptr && (TTTS_H3I_BIG_GATE != 0 || g.relu_out == nullptr) && g.K <= TTTS_H3I_BIG_MAXK && g.N >= 1024 &&
           (long)cdiv(g.N, IBN) * cdiv(g.M, 256) >= 256;
}

template <int TM>
static int launch_h3i(const GemmArgs& g, hipStream_t stream) {
    const long ntiles = (long)cdiv(g.N, IBN) * cdiv(g.M, TM);
    // as many workgroups as the chip holds at once (two per CU at 128 rows, one at 256), and no more of them than level rounds
    // need (a multiple of 8: virtual ids keep their XCD)
    constexpr long slots = TM == 128 ? 512 : 256;
    const long rounds = (ntiles + slots - 1) / slots;
    long gsz = ((ntiles + rounds - 1) / rounds + 7) / 8 * 8;
    if (gsz > slots) gsz = slots;
    dim3 grid((unsigned)(ntiles < slots ? ntiles : gsz), 1, 1);
    const dim3 block(TM * 2);
    const bool res = g.residual != nullptr, gate = g.relu_out != nullptr, drop = g.drop_thr != 0u;
    if (g.c_row_inv != nullptr) {
        if constexpr (TM == 128) {
            if (g.a_row_inv != nullptr) hipLaunchKernelGGL((gemm_h3i_kernel<128, false, false, false, false, true>), grid, block, 0, stream, g);
            else hipLaunchKernelGGL((gemm_h3i_kernel<128, true, false, false, false, true>), grid, block, 0, stream, g);
        }
        TTTS_LAUNCH_CHECK("gemm_h3i_kernel<img>");
        return TTTS_OK;
    }
#define TTTS_H3I(R, G, D)                                                                                   \
    do {                                                                                                    \
        if (g.a_row_inv != nullptr) hipLaunchKernelGGL((gemm_h3i_kernel<TM, false, R, G, D>), grid, block, 0, stream, g); \
        else hipLaunchKernelGGL((gemm_h3i_kernel<TM, true, R, G, D>), grid, block, 0, stream, g);          \
    } while (0)
    if (gate) { if (res) TTTS_H3I(true, true, false); else TTTS_H3I(false, true, false); }
    else if (drop) { if (res) TTTS_H3I(true, false, true); else TTTS_H3I(false, false, true); }
    else { if (res) TTTS_H3I(true, false, false); else TTTS_H3I(false, false, false); }
#undef TTTS_H3I
    TTTS_LAUNCH_CHECK("gemm_h3i_kernel");
    return TTTS_OK;
}

int dispatch_h3i(const GemmArgs& g, hipStream_t stream) {
    const long ldmax = g.ldc > g.ldr ? g.ldc : g.ldr;
    if (((long)g.M + 256) * ldmax * 4 >= (1L << 32)) {
        set_error("fp16x3 GEMM (image operand): output larger than 4 GiB (M=%d, row stride %ld)", g.M, ldmax);
        return TTTS_ERR_INVALID;
    }
    if (!h3i_supports(g)) {
        set_error("fp16x3 GEMM (image operand): unsupported shape M=%d N=%d K=%d", g.M, g.N, g.K);
        return TTTS_ERR_INVALID;
    }
    const bool res = g.residual != nullptr, gate = g.relu_out != nullptr, drop = g.drop_thr != 0u;
    if (g.c_row_inv != nullptr && (res || gate || drop || g.act != 0 || g.N % 64 != 0)) {
        set_error("fp16x3 GEMM (head-image output): bias-only epilogue and N %% 64 == 0 required (N=%d)", g.N);
        return TTTS_ERR_INVALID;
    }
    if (gate && drop) {
        set_error("fp16x3 GEMM (image operand): a relu gate (data gradient) cannot be combined with dropout");
        return TTTS_ERR_INVALID;
    }
    return h3i_big_tile(g) ? launch_h3i<256>(g, stream) : launch_h3i<128>(g, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 (M x K) -> activation image + per-row inverse scale.  One wave per row, a lane owns NV float4 (K = 256 NV' or any
// multiple of 16 up to 1024: lanes past the row's end idle).  The stand-alone form of what the producers do in their epilogues.
template <int NV>
__global__ __launch_bounds__(256) void act_image_kernel(const float* __restrict__ x, unsigned short* __restrict__ img,
                                                        float* __restrict__ row_inv, long M, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= M) return;
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c4 = lane + 64 * k;
        v[k] = (c4 * 4 < K) ? reinterpret_cast<const float4*>(x + row * K)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    image_emit_row<NV>(v, lane, row, M, K, img, row_inv);
}

// fp32 (M rows x N columns, row stride ld floats) -> HEAD IMAGE in `img` (same row stride, in 4-byte units) + per-(row, head)
// inverse scales: the stand-alone form of the IMG epilogue (tests; operands that no GEMM of ours produced).  One wave per row
// and pass of four heads; 16 lanes hold a head.
__global__ __launch_bounds__(256) void head_image_kernel(const float* __restrict__ x, long ldx, unsigned short* __restrict__ img,
                                                         long ldi, float* __restrict__ row_inv, long M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= M) return;
    for (int c0 = 0; c0 < N; c0 += 256) {
        const int col = c0 + lane * 4;
        const bool ok = col < N;
        const float4 v = ok ? *reinterpret_cast<const float4*>(x + row * ldx + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sc, inv;
        h3_pow2_scale(m, sc, inv);
        if (ok) {
            uint2 hi, lo;
            split2_pair(f32x2{v.x, v.y} * sc, hi.x, lo.x);
            split2_pair(f32x2{v.z, v.w} * sc, hi.y, lo.y);
            const int g64 = col >> 6, d = col & 63;
            unsigned short* p = img + (row * ldi + g64 * 64) * 2 + d;
            *reinterpret_cast<uint2*>(p) = hi;
            *reinterpret_cast<uint2*>(p + 64) = lo;
            if ((lane & 15) == 0) row_inv[(long)g64 * M + row] = inv;
        }
    }
}

static GemmArgs h3i_base_args() {
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

}  // namespace ttts

using namespace ttts;

#ifdef TTTS_CLOCK_STAMPS
extern "C" int ttts_dbg_read_clock_h3i(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ttts::ttts_clock_h3i), n * sizeof(unsigned long long));
}
#endif
#ifdef TTTS_H3I_STAMPS
extern "C" int ttts_dbg_h3i_read_stamps(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ttts::ttts_h3i_stamps), n * sizeof(unsigned long long));
}
#endif

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int ttts_act_image(const float* x, void* image, float* row_inv, int64_t M, int K, void* stream) {
    TTTS_REQUIRE(x && image && row_inv, "act_image: null pointer");
    TTTS_REQUIRE(M > 0 && K > 0 && K % 16 == 0 && K <= 1024, "act_image: K=%d must be a multiple of 16, at most 1024", K);
    TTTS_REQUIRE(al16(x) && al16(image), "act_image: pointers must be 16-byte aligned");
    const dim3 grid((unsigned)cdiv(M, 4));
    if (K <= 256) hipLaunchKernelGGL((act_image_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)image, row_inv, (long)M, K);
    else if (K <= 512) hipLaunchKernelGGL((act_image_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)image, row_inv, (long)M, K);
    else hipLaunchKernelGGL((act_image_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)image, row_inv, (long)M, K);
    TTTS_LAUNCH_CHECK("act_image_kernel");
    return TTTS_OK;
}

extern "C" int ttts_linear_fwd_h3i(const void* x_image, const float* x_row_inv, const void* w_planes, const float* bias,
                                   const float* residual, float* y, int64_t M, int N, int K, int act, float drop_p, uint64_t seed,
                                   const uint64_t* step_seed, float* y_amax_out, void* stream) {
    TTTS_REQUIRE(x_image && x_row_inv && w_planes && y, "linear_fwd_h3i: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_fwd_h3i: bad dims");
    TTTS_REQUIRE(K % 32 == 0 && N % 4 == 0, "linear_fwd_h3i: K=%d must be a multiple of 32 and N=%d of 4", K, N);
    TTTS_REQUIRE(al16(x_image) && al16(w_planes) && al16(y), "linear_fwd_h3i: pointers must be 16-byte aligned");
    TTTS_REQUIRE(act == 0 || act == 1, "linear_fwd_h3i: act must be 0 (none) or 1 (relu)");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear_fwd_h3i: dropout p out of [0,1)");
    TTTS_REQUIRE((uint64_t)M * K * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_fwd_h3i: operand larger than 4 GiB");
    GemmArgs g = h3i_base_args();
    g.A = (const float*)x_image; g.B = (const float*)w_planes; g.C = y; g.M = (int)M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.a_bytes = (uint32_t)((uint64_t)M * K * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 4);
    g.cin = K;
    g.bias = bias; g.act = act;
    if (drop_p > 0.f) { g.drop_thr = drop_threshold(drop_p); g.drop_scale = 1.f / (1.f - drop_p); g.seed = seed; g.step_seed = step_seed; }
    g.residual = residual; g.ldr = N;
    g.a_row_inv = x_row_inv;
    g.b_amax = h3_plane_tail(w_planes, N, K); g.b_amax_n = 1;
    g.c_amax = y_amax_out;
    return dispatch_h3i(g, (hipStream_t)stream);
}

extern "C" int ttts_linear_bwd_data_h3i(const void* dy_image, const float* dy_row_inv, const void* wt_planes, const float* residual,
                                        float* dx, int64_t M, int N, int K, const float* relu_out, float relu_scale,
                                        float* dx_amax_out, void* stream) {
    // dx[M,K] = dy[M,N] . w[N,K] (+ residual): the gradient is the image operand; wt_planes = weight_split mode 5 (w^T)
    TTTS_REQUIRE(dy_image && dy_row_inv && wt_planes && dx, "linear_bwd_data_h3i: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_bwd_data_h3i: bad dims");
    TTTS_REQUIRE(N % 32 == 0 && K % 4 == 0, "linear_bwd_data_h3i: N=%d must be a multiple of 32 and K=%d of 4", N, K);
    TTTS_REQUIRE(al16(dy_image) && al16(wt_planes) && al16(dx), "linear_bwd_data_h3i: pointers must be 16-byte aligned");
    TTTS_REQUIRE((uint64_t)M * N * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_bwd_data_h3i: operand larger than 4 GiB");
    GemmArgs g = h3i_base_args();
    g.A = (const float*)dy_image; g.B = (const float*)wt_planes; g.C = dx; g.M = (int)M; g.N = K; g.K = N;
    g.lda = N; g.ldb = N; g.ldc = K; g.cin = N;
    g.a_bytes = (uint32_t)((uint64_t)M * N * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 4);
    g.residual = residual; g.ldr = K;
    g.relu_out = relu_out; g.relu_scale = relu_scale;
    g.a_row_inv = dy_row_inv;
    g.b_amax = h3_plane_tail(wt_planes, K, N); g.b_amax_n = 1;
    g.c_amax = dx_amax_out;
    return dispatch_h3i(g, (hipStream_t)stream);
}

/* fp32 activation (per-tensor scale from its partial maxima) on the LDS-DMA kernel: the weight image is the K16-major one
 * (ttts_weight_split modes 8 / 9); arguments otherwise as ttts_linear_fwd_h3 / ttts_linear_bwd_data_h3 without a row shift */
extern "C" int ttts_linear_fwd_h3d(const float* x, const void* w_planes, const float* bias, const float* residual, float* y,
                                   int64_t M, int N, int K, int act, float drop_p, uint64_t seed, const uint64_t* step_seed,
                                   const float* x_amax, float* y_amax_out, void* stream) {
    TTTS_REQUIRE(x && w_planes && y && x_amax, "linear_fwd_h3d: null pointer (x_amax, the partial maxima of |x|, is required)");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_fwd_h3d: bad dims");
    TTTS_REQUIRE(K % 32 == 0 && N % 4 == 0, "linear_fwd_h3d: K=%d must be a multiple of 32 and N=%d of 4", K, N);
    TTTS_REQUIRE(al16(x) && al16(w_planes) && al16(y), "linear_fwd_h3d: pointers must be 16-byte aligned");
    TTTS_REQUIRE(act == 0 || act == 1, "linear_fwd_h3d: act must be 0 (none) or 1 (relu)");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "linear_fwd_h3d: dropout p out of [0,1)");
    TTTS_REQUIRE((uint64_t)M * K * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_fwd_h3d: operand larger than 4 GiB");
    GemmArgs g = h3i_base_args();
    g.A = x; g.B = (const float*)w_planes; g.C = y; g.M = (int)M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.a_bytes = (uint32_t)((uint64_t)M * K * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 4);
    g.cin = K;
    g.bias = bias; g.act = act;
    if (drop_p > 0.f) { g.drop_thr = drop_threshold(drop_p); g.drop_scale = 1.f / (1.f - drop_p); g.seed = seed; g.step_seed = step_seed; }
    g.residual = residual; g.ldr = N;
    g.a_amax = x_amax; g.a_amax_n = H3_AMAX_PARTIALS;
    g.b_amax = h3_plane_tail(w_planes, N, K); g.b_amax_n = 1;
    g.c_amax = y_amax_out;
    return dispatch_h3i(g, (hipStream_t)stream);
}

extern "C" int ttts_linear_bwd_data_h3d(const float* dy, const void* wt_planes, const float* residual, float* dx, int64_t M, int N,
                                        int K, const float* relu_out, float relu_scale, const float* dy_amax, float* dx_amax_out,
                                        void* stream) {
    TTTS_REQUIRE(dy && wt_planes && dx && dy_amax, "linear_bwd_data_h3d: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_bwd_data_h3d: bad dims");
    TTTS_REQUIRE(N % 32 == 0 && K % 4 == 0, "linear_bwd_data_h3d: N=%d must be a multiple of 32 and K=%d of 4", N, K);
    TTTS_REQUIRE(al16(dy) && al16(wt_planes) && al16(dx), "linear_bwd_data_h3d: pointers must be 16-byte aligned");
    TTTS_REQUIRE((uint64_t)M * N * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_bwd_data_h3d: operand larger than 4 GiB");
    GemmArgs g = h3i_base_args();
    g.A = dy; g.B = (const float*)wt_planes; g.C = dx; g.M = (int)M; g.N = K; g.K = N;
    g.lda = N; g.ldb = N; g.ldc = K; g.cin = N;
    g.a_bytes = (uint32_t)((uint64_t)M * N * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 4);
    g.residual = residual; g.ldr = K;
    g.relu_out = relu_out; g.relu_scale = relu_scale;
    g.a_amax = dy_amax; g.a_amax_n = H3_AMAX_PARTIALS;
    g.b_amax = h3_plane_tail(wt_planes, K, N); g.b_amax_n = 1;
    g.c_amax = dx_amax_out;
    return dispatch_h3i(g, (hipStream_t)stream);
}

/* The in-projections of nn.MultiheadAttention (packed q/k/v, or its q and k/v row slices: torch/nn/functional.py:6206+ reached
 * from model/layers.py:54-74 and torch _sa_block) with a HEAD-IMAGE output: y_image has the geometry of the fp32 output (M rows of
 * N 4-byte cells) but holds, per row and 64-column head, {64 f16 hi, 64 f16 lo} of (x W^T + b) * 2^e(row, head);
 * y_row_inv[head * M + row] = 2^-e.  y_amax_out: NULL, or N / amax_section_cols caller-zeroed TTTS_AMAX_SLOTS-float arrays (one per
 * section: q / k / v), each receiving max|y| of its section. */
extern "C" int ttts_linear_fwd_h3d_img(const float* x, const void* w_planes, const float* bias, void* y_image, float* y_row_inv,
                                       int64_t M, int N, int K, const float* x_amax, float* y_amax_out, int amax_section_cols,
                                       void* stream) {
    TTTS_REQUIRE(x && w_planes && y_image && y_row_inv && x_amax, "linear_fwd_h3d_img: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear_fwd_h3d_img: bad dims");
    TTTS_REQUIRE(K % 32 == 0 && N % 64 == 0, "linear_fwd_h3d_img: K=%d must be a multiple of 32 and N=%d of 64", K, N);
    TTTS_REQUIRE(al16(x) && al16(w_planes) && al16(y_image), "linear_fwd_h3d_img: pointers must be 16-byte aligned");
    TTTS_REQUIRE(amax_section_cols >= 0 && (amax_section_cols == 0 || (amax_section_cols % 64 == 0 && N % amax_section_cols == 0)),
                 "linear_fwd_h3d_img: amax sections must be whole heads that divide N");
    TTTS_REQUIRE((uint64_t)M * K * 4 < (1ull << 32) && (uint64_t)N * K * 4 < (1ull << 32), "linear_fwd_h3d_img: operand larger than 4 GiB");
    GemmArgs g = h3i_base_args();
    g.A = x; g.B = (const float*)w_planes; g.C = (float*)y_image; g.M = (int)M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.a_bytes = (uint32_t)((uint64_t)M * K * 4); g.b_bytes = (uint32_t)((uint64_t)N * K * 4);
    g.cin = K;
    g.bias = bias;
    g.a_amax = x_amax; g.a_amax_n = H3_AMAX_PARTIALS;
    g.b_amax = h3_plane_tail(w_planes, N, K); g.b_amax_n = 1;
    g.c_amax = y_amax_out; g.c_amax_sec = amax_section_cols;
    g.c_row_inv = y_row_inv;
    return dispatch_h3i(g, (hipStream_t)stream);
}

/* fp32 rows (row stride ld_x floats) -> head image (row stride ld_image 4-byte cells) + inverse scales [N / 64][M] */
extern "C" int ttts_head_image(const float* x, int64_t ld_x, void* image, int64_t ld_image, float* row_inv, int64_t M, int N,
                               void* stream) {
    TTTS_REQUIRE(x && image && row_inv, "head_image: null pointer");
    TTTS_REQUIRE(M > 0 && N > 0 && N % 64 == 0 && ld_x >= N && ld_image >= N && ld_x % 4 == 0 && ld_image % 4 == 0,
                 "head_image: N=%d must be a multiple of 64 and the row strides multiples of 4 that cover it", N);
    TTTS_REQUIRE(al16(x) && al16(image), "head_image: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(head_image_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, (long)ld_x,
                       (unsigned short*)image, (long)ld_image, row_inv, (long)M, N);
    TTTS_LAUNCH_CHECK("head_image_kernel");
    return TTTS_OK;
}
