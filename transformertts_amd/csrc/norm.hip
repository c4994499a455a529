// LayerNorm and BatchNorm1d (train / eval) forward + backward for (rows, channels) fp32 activations.
// HBM-bound: every kernel streams rows with coalesced accesses across channels, reduces with
// wave shuffles / fixed-order partials (bitwise reproducible, no atomics).
#include "gemm_common.h"

namespace ttts {

// ------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; lane owns columns lane, lane+64, ... (coalesced 256-B segments per step)
constexpr int LN_MAXPER = 16;   // d <= 1024

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long M, int d, float eps, float* __restrict__ amax_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= M) return;
    float ymax = 0.f;
    const float* xr = x + row * d;
    float v[LN_MAXPER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPER; ++i) {
        v[i] = 0.f;
        if (lane + 64 * i < d) { v[i] = xr[lane + 64 * i]; s += v[i]; }          // any width up to 64 * LN_MAXPER
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPER; ++i) {
        if (lane + 64 * i < d) { float c = v[i] - mean; q += c * c; }
    }
    const float var = wave_sum(q) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
    float* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < LN_MAXPER; ++i) {
        if (lane + 64 * i < d) {
            int c = lane + 64 * i;
            const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
            yr[c] = o;
            ymax = fmaxf(ymax, fabsf(o));
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    if (amax_out != nullptr) amax_publish(ymax, amax_out, (int)row);
}

// d = 256 * NV: a lane owns NV float4 (16-byte accesses, one 1-KB wave instruction per 256 columns) and a wave walks
// over ROWS consecutive rows with all their loads in flight before the first reduction -- a quarter of the memory
// instructions and a quarter of the waves of the kernel above (23.8 -> 19.3 us for 55 680 x 256, 5.9 TB/s).
template <int NV, int ROWS>
__global__ __launch_bounds__(256) void layernorm_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               long M, float eps, float* __restrict__ amax_out,
                                                               unsigned short* __restrict__ img, float* __restrict__ row_inv) {
    constexpr int d = 256 * NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row0 = ((long)blockIdx.x * 4 + wave) * ROWS;
    if (row0 >= M) return;
    float ymax = 0.f;
    float4 v[ROWS][NV];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const long row = (row0 + r < M) ? row0 + r : M - 1;          // tail rows re-read the last row, not stored
            v[r][k] = reinterpret_cast<const float4*>(x + row * d)[lane + 64 * k];
        }
    float4 ga[NV], be[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        ga[k] = reinterpret_cast<const float4*>(gamma)[lane + 64 * k];
        be[k] = reinterpret_cast<const float4*>(beta)[lane + 64 * k];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += (v[r][k].x + v[r][k].y) + (v[r][k].z + v[r][k].w);
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const float a = v[r][k].x - mean, b = v[r][k].y - mean, c = v[r][k].z - mean, e = v[r][k].w - mean;
            q += (a * a + b * b) + (c * c + e * e);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
        if (row0 + r < M) {
            float4 ov[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float4 o = make_float4((v[r][k].x - mean) * rstd * ga[k].x + be[k].x, (v[r][k].y - mean) * rstd * ga[k].y + be[k].y,
                                             (v[r][k].z - mean) * rstd * ga[k].z + be[k].z, (v[r][k].w - mean) * rstd * ga[k].w + be[k].w);
                reinterpret_cast<float4*>(y + (row0 + r) * d)[lane + 64 * k] = o;
                ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                ov[k] = o;
            }
            // the image operand of the GEMMs that read y (gemm_h3i.hip): the row is complete in this wave's registers
            if (img != nullptr) image_emit_row<NV>(ov, lane, row0 + r, M, d, img, row_inv);
            if (lane == 0) {
                if (mean_out) mean_out[row0 + r] = mean;
                if (rstd_out) rstd_out[row0 + r] = rstd;
            }
        }
    }
    // max|y| of the rows this wave wrote, for the fp16x3 GEMM / attention that consumes y (caller-zeroed slots)
    if (amax_out != nullptr) amax_publish(ymax, amax_out, blockIdx.x * 4 + wave);
}

constexpr int LN_BWD_BLOCKS = 256;

// dx per row; per-block partial column sums of dy*xhat (dgamma) and dy (dbeta) -> ws[block][2][d]
template <int NPER>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ ws, long M, int d) {
    // d <= NPER * 64 (any width: the columns past d of the last 64-column group are skipped)
    constexpr int DP = NPER * 64;
    __shared__ float red[4][2][DP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float g[NPER], accg[NPER], accb[NPER];
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        accg[i] = 0.f; accb[i] = 0.f;
        g[i] = (lane + 64 * i < d) ? gamma[lane + 64 * i] : 0.f;
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const float* xr = x + row * d;
        const float* dr = dy + row * d;
        float xh[NPER], gd[NPER];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NPER; ++i) {
            int c = lane + 64 * i;
            const bool in = c < d;
            float dyv = in ? dr[c] : 0.f;
            xh[i] = in ? (xr[c] - mu) * rs : 0.f;
            gd[i] = dyv * g[i];
            s1 += gd[i];
            s2 += gd[i] * xh[i];
            accg[i] += dyv * xh[i];
            accb[i] += dyv;
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
        float* dxr = dx + row * d;
#pragma unroll
        for (int i = 0; i < NPER; ++i)
            if (lane + 64 * i < d) dxr[lane + 64 * i] = rs * (gd[i] - s1 - xh[i] * s2);
    }
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        red[wave][0][lane + 64 * i] = accg[i];
        red[wave][1][lane + 64 * i] = accb[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += 256) {
        int which = c / d, col = c - which * d;
        float sum = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
        ws[((long)blockIdx.x * 2 + which) * d + col] = sum;
    }
}

// d = 256 * NV form of the backward: 16-byte accesses, two rows per wave iteration in flight, NW waves per block (the
// block count is fixed by the partial-sum workspace, so occupancy comes from the block size: 16 waves per CU at d = 256)
// DROP: also write dacc = dx * keep(seed, element) / (1-p) -- the gradient behind the residual dropout of the sublayer whose
// output this LayerNorm normalised (what ttts_dropout_bwd would compute from dx in a pass of its own) -- and publish the
// maximum of |dacc| (atomic maxima into a caller-zeroed TTTS_AMAX_SLOTS-slot array)
template <int NV, int NW, bool DROP>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_v4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, float* __restrict__ dx,
                                                               float* __restrict__ ws, long M, float* __restrict__ dacc,
                                                               float drop_scale, uint32_t thr, uint64_t seed,
                                                               const uint64_t* step_seed, float* __restrict__ dacc_amax,
                                                               unsigned short* __restrict__ img, float* __restrict__ row_inv) {
    constexpr int d = 256 * NV;
    __shared__ float red[NW][2][d];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seed_eff = DROP ? site_seed(seed, step_seed) : 0;
    float amx = 0.f;
    float4 g[NV], accg[NV], accb[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        g[k] = reinterpret_cast<const float4*>(gamma)[lane + 64 * k];
        accg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        accb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long stride = (long)gridDim.x * NW;
    for (long row = (long)blockIdx.x * NW + wave; row < M; row += 2 * stride) {
        float4 dv[2][NV], xv[2][NV];
        float mu[2], rs[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long r = (row + u * stride < M) ? row + u * stride : row;       // second row may not exist: redo the first
            mu[u] = mean[r]; rs[u] = rstd[r];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                dv[u][k] = reinterpret_cast<const float4*>(dy + r * d)[lane + 64 * k];
                xv[u][k] = reinterpret_cast<const float4*>(x + r * d)[lane + 64 * k];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long r = row + u * stride;
            if (r >= M) break;
            float4 xh[NV], gd[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                xh[k] = make_float4((xv[u][k].x - mu[u]) * rs[u], (xv[u][k].y - mu[u]) * rs[u], (xv[u][k].z - mu[u]) * rs[u],
                                    (xv[u][k].w - mu[u]) * rs[u]);
                gd[k] = make_float4(dv[u][k].x * g[k].x, dv[u][k].y * g[k].y, dv[u][k].z * g[k].z, dv[u][k].w * g[k].w);
                s1 += (gd[k].x + gd[k].y) + (gd[k].z + gd[k].w);
                s2 += (gd[k].x * xh[k].x + gd[k].y * xh[k].y) + (gd[k].z * xh[k].z + gd[k].w * xh[k].w);
                accg[k].x += dv[u][k].x * xh[k].x; accg[k].y += dv[u][k].y * xh[k].y;
                accg[k].z += dv[u][k].z * xh[k].z; accg[k].w += dv[u][k].w * xh[k].w;
                accb[k].x += dv[u][k].x; accb[k].y += dv[u][k].y; accb[k].z += dv[u][k].z; accb[k].w += dv[u][k].w;
            }
            s1 = wave_sum(s1) / (float)d;
            s2 = wave_sum(s2) / (float)d;
            float4 ov[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float4 o = make_float4(rs[u] * (gd[k].x - s1 - xh[k].x * s2), rs[u] * (gd[k].y - s1 - xh[k].y * s2),
                                             rs[u] * (gd[k].z - s1 - xh[k].z * s2), rs[u] * (gd[k].w - s1 - xh[k].w * s2));
                reinterpret_cast<float4*>(dx + r * d)[lane + 64 * k] = o;
                ov[k] = o;
                if (DROP) {
                    bool kp[4];
                    keep_quad(seed_eff, (uint64_t)(r * d + 4 * (lane + 64 * k)), thr, kp);
                    const float4 da = make_float4(kp[0] ? o.x * drop_scale : 0.f, kp[1] ? o.y * drop_scale : 0.f,
                                                  kp[2] ? o.z * drop_scale : 0.f, kp[3] ? o.w * drop_scale : 0.f);
                    reinterpret_cast<float4*>(dacc + r * d)[lane + 64 * k] = da;
                    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(da.x), fabsf(da.y))), fmaxf(fabsf(da.z), fabsf(da.w)));
                    ov[k] = da;
                }
            }
            // image of what the producing Linear's backward takes as its dy: dacc with a residual dropout, dx without
            if (img != nullptr) image_emit_row<NV>(ov, lane, r, M, d, img, row_inv);
        }
    }
    if (DROP && dacc_amax != nullptr) amax_publish(amx, dacc_amax, blockIdx.x * NW + wave);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        reinterpret_cast<float4*>(&red[wave][0][0])[lane + 64 * k] = accg[k];
        reinterpret_cast<float4*>(&red[wave][1][0])[lane + 64 * k] = accb[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += 64 * NW) {
        int which = c / d, col = c - which * d;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4)
            sum += (red[w][which][col] + red[w + 1][which][col]) + (red[w + 2][which][col] + red[w + 3][which][col]);
        ws[((long)blockIdx.x * 2 + which) * d + col] = sum;
    }
}

// ------------------------------------------------------------------------------------------ BatchNorm
constexpr int BN_MAXBLK = 512;   // row chunks: enough waves (and bytes in flight) to stream at HBM rate
constexpr int BN_XROWS = 4;      // bn_stats_final_kernel: rows of the matrix itself merged with the partials, per merge lane (64 lanes)

// per (row-chunk, channel): count, mean, M2 = sum (x - mean)^2.  Block = 64 channels x 4 row-lanes; two passes over the
// chunk (the second one hits L2) so no division sits in the streaming loops; merged deterministically afterwards.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ ws, long M,
                                                               int C, int rows_per_block) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const float n = (float)(r1 - r0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {       // four independent loads in flight per thread
            s0 += x[r * C + c]; s1 += x[(r + 4) * C + c]; s2 += x[(r + 8) * C + c]; s3 += x[(r + 12) * C + c];
        }
        for (; r < r1; r += 4) s0 += x[r * C + c];
    }
    red[rl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    const float mean = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / n;
    __syncthreads();
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if (ok) {
        long r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            float a = x[r * C + c] - mean, b = x[(r + 4) * C + c] - mean, d2 = x[(r + 8) * C + c] - mean,
                  e2 = x[(r + 12) * C + c] - mean;
            q0 += a * a; q1 += b * b; q2 += d2 * d2; q3 += e2 * e2;
        }
        for (; r < r1; r += 4) { float a = x[r * C + c] - mean; q0 += a * a; }
    }
    red[rl][cl] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    if (rl == 0 && ok) {
        float* w = ws + ((long)blockIdx.y * 3) * C;
        w[c] = n;
        w[C + c] = mean;
        w[2 * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    }
}

// Merge the per-chunk (count, mean, M2) partials of a channel (Chan et al.), 16 channels x 16 merge-lanes per
// workgroup: each lane folds every 16th partial in order, lane 0 then folds the 16 lane results in order.
struct BnStatSet {          // one set of rows of a matrix: row-chunk partials [nblk][3][C] and / or nxr rows of the matrix itself
    const float* ws;
    const float* xr;
    float* mean;
    float* invstd;
    int nblk, nxr;
};

// NSETS = 2: the statistics of TWO row sets of one matrix (the halves of a twin batch) in one launch -- 256 threads per set --, the
// running statistics updated with set a's first, then set b's, by one thread per channel.
template <int NSETS>
__global__ __launch_bounds__(256 * NSETS) void bn_stats_final_kernel(BnStatSet sa, BnStatSet sb, float* __restrict__ running_mean,
                                                                     float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                                     int C, float momentum, float eps) {
    // merge of the row-chunk partials without a serial chain of divisions: N = sum n_b, mean = sum n_b mean_b / N, then
    // M2 = sum [M2_b + n_b (mean_b - mean)^2] (the pooled-variance identity, centred on the global mean).  4 channels
    // x 64 chunk-lanes per block (the loop is latency bound: short trips, many lanes), fixed summation order.
    __shared__ float sn[NSETS][64][5], sm[NSETS][64][5], sq[NSETS][64][5];
    __shared__ float fin[NSETS][2][4];
    const int set = NSETS > 1 ? (int)(threadIdx.x >> 8) : 0, t = threadIdx.x & 255;
    const BnStatSet& s = (NSETS > 1 && set) ? sb : sa;
    const float* __restrict__ ws = s.ws;
    const float* __restrict__ xr = s.xr;
    const int nblk = s.nblk, nxr = s.nxr;
    const int cl = t & 3, rl = t >> 2;
    const int c = blockIdx.x * 4 + cl;
    // a lane's partials (every 64th chunk: at most BN_MAXBLK / 64 = 8) are all requested up front and kept for the second
    // pass: the kernel is two dependent trips to L2 otherwise repeated per chunk (10 -> 4 us)
    constexpr int PER = BN_MAXBLK / 64;
    float pn[PER], pm[PER], pq[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int b = rl + 64 * i;
        const bool ok = c < C && b < nblk;
        const float* w = ws + ((long)(ok ? b : 0) * 3) * C + (ok ? c : 0);
        pn[i] = ok ? w[0] : 0.f;
        pm[i] = ok ? w[C] : 0.f;
        pq[i] = ok ? w[2 * C] : 0.f;
    }
    // xr: nxr (<= 64 * BN_XROWS) more rows of the matrix itself, each a partial (1, x, 0): the rows of a row chunk that only
    // partly belongs to this set of rows (a twin batch whose halves do not end on a chunk boundary, ops.ConvBNFn._forward_twin)
    float xv[BN_XROWS];
#pragma unroll
    for (int i = 0; i < BN_XROWS; ++i) {
        const int r = rl + 64 * i;
        xv[i] = (c < C && r < nxr) ? xr[(long)r * C + c] : 0.f;
    }
    float n = 0.f, sw = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { n += pn[i]; sw += pn[i] * pm[i]; }
#pragma unroll
    for (int i = 0; i < BN_XROWS; ++i) { n += (rl + 64 * i < nxr) ? 1.f : 0.f; sw += xv[i]; }
    sn[set][rl][cl] = n; sm[set][rl][cl] = sw;
    __syncthreads();
    n = 0.f; sw = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) { n += sn[set][i][cl]; sw += sm[set][i][cl]; }
    const float mean = (n > 0.f) ? sw / n : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float dlt = pm[i] - mean;
        m2 += pq[i] + pn[i] * dlt * dlt;
    }
#pragma unroll
    for (int i = 0; i < BN_XROWS; ++i) {
        const float dlt = xv[i] - mean;
        m2 += (rl + 64 * i < nxr) ? dlt * dlt : 0.f;
    }
    sq[set][rl][cl] = m2;
    __syncthreads();
    if (rl == 0 && c < C) {
        m2 = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) m2 += sq[set][i][cl];
        const float var = m2 / n;
        s.mean[c] = mean;
        s.invstd[c] = 1.0f / sqrtf(var + eps);
        fin[set][0][cl] = mean;
        fin[set][1][cl] = (n > 1.f) ? m2 / (n - 1.f) : var;
    }
    __syncthreads();
    if (set == 0 && rl == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < NSETS; ++k) {              // set a's update first, then set b's
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fin[k][0][cl];
            if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * fin[k][1][cl];
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += NSETS;
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ mean,
                                     float* __restrict__ invstd, int C, float eps) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mean[c] = rm[c];
        invstd[c] = 1.0f / sqrtf(rv[c] + eps);
    }
}

// z = drop(act(xhat*gamma + beta)); 4 channels per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ z, long n4,
                                                           int C, int act, float drop_scale, uint32_t thr, uint64_t seed, const uint64_t* step_seed,
                                                           float* __restrict__ amax_out) {
    seed = site_seed(seed, step_seed);
    float zmax = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const int c = (int)(e % C);
        float4 xv = *reinterpret_cast<const float4*>(x + e);
        float4 mu = *reinterpret_cast<const float4*>(mean + c);
        float4 is = *reinterpret_cast<const float4*>(invstd + c);
        float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        float4 be = *reinterpret_cast<const float4*>(beta + c);
        float o[4] = {(xv.x - mu.x) * is.x * ga.x + be.x, (xv.y - mu.y) * is.y * ga.y + be.y,
                      (xv.z - mu.z) * is.z * ga.z + be.z, (xv.w - mu.w) * is.w * ga.w + be.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (act == TTTS_ACT_TANH) o[j] = tanhf(o[j]);
            if (thr != 0u) o[j] = keep_elem(seed, (uint64_t)(e + j), thr) ? o[j] * drop_scale : 0.f;
        }
        *reinterpret_cast<float4*>(z + e) = make_float4(o[0], o[1], o[2], o[3]);
        zmax = fmaxf(fmaxf(zmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    if (amax_out != nullptr) amax_publish(zmax, amax_out, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// 1 - tanh(u)^2 for the backward pass: tanh from one v_exp and one v_rcp (absolute error ~1e-7 in tanh, i.e. relative
// error <= 2e-7 in the derivative) instead of libm's tanhf (~30 instructions incl. branches; the forward keeps tanhf)
__device__ __forceinline__ float dtanh_fast(float u) {
    const float e = __expf(2.0f * u);                       // inf for large u -> t = 1; 0 for very negative u -> t = -1
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
    return 1.0f - t * t;
}
// gradient w.r.t. the BN output (before act / dropout) for element e, channel c; `keep` = the element's dropout decision
__device__ __forceinline__ float bn_dy_pre_k(float dz, float xhat, float gamma, float beta, int act, float drop_scale, bool keep) {
    float g = keep ? dz * drop_scale : 0.f;
    if (act == TTTS_ACT_TANH) g *= dtanh_fast(xhat * gamma + beta);
    return g;
}
__device__ __forceinline__ float bn_dy_pre(float dz, float xhat, float gamma, float beta, int act, float drop_scale,
                                           uint32_t thr, uint64_t seed, uint64_t e) {
    return bn_dy_pre_k(dz, xhat, gamma, beta, act, thr != 0u ? drop_scale : 1.0f, thr == 0u || keep_elem(seed, e, thr));
}

// per (row-chunk, channel): sum dy, sum dy*xhat -> ws[blk][2C]  (block = 64 channels x 4 row-lanes)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ ws, long M, int C, int rows_per_block,
                                                             int act, float drop_scale, uint32_t thr, uint64_t seed, const uint64_t* step_seed) {
    seed = site_seed(seed, step_seed);
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
        const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
#pragma unroll 4
        for (long r = r0 + rl; r < r1; r += 4) {
            long e = r * C + c;
            float xh = (x[e] - mu) * is;
            float g = bn_dy_pre(dz[e], xh, ga, be, act, drop_scale, thr, seed, (uint64_t)e);
            s1 += g;
            s2 += g * xh;
        }
    }
    red[0][rl][cl] = s1;
    red[1][rl][cl] = s2;
    __syncthreads();
    if (rl == 0 && ok) {
        ws[((long)blockIdx.y * 2) * C + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        ws[((long)blockIdx.y * 2 + 1) * C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ sums, float* __restrict__ dx, long n4,
                                                           int C, float inv_m, int act, float drop_scale, uint32_t thr,
                                                           uint64_t seed, const uint64_t* step_seed, float* __restrict__ amax,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    seed = site_seed(seed, step_seed);
    if (blockIdx.x == 0)                                    // dgamma / dbeta (+)= the reduced sums (no launch of their own)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sums[c] : sums[c];
            if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sums[C + c] : sums[C + c];
        }
    float mx = 0.f;
    const float dsc = thr != 0u ? drop_scale : 1.0f;
    // per-channel vectors as 16-byte loads when every one of them is 16-byte aligned (parameters may sit at any 4-byte
    // offset of a flat buffer)
    const bool pv = (((uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)mean | (uintptr_t)beta | (uintptr_t)sums) & 15) == 0;
    auto ld4 = [&](const float* p) -> float4 {
        return pv ? *reinterpret_cast<const float4*>(p) : make_float4(p[0], p[1], p[2], p[3]);
    };
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const int c = (int)(e % C);                         // C % 4 == 0: the four elements are channels c .. c+3 of one row
        const float4 xv = *reinterpret_cast<const float4*>(x + e);
        const float4 dv = *reinterpret_cast<const float4*>(dz + e);
        const float4 is4 = ld4(invstd + c), ga4 = ld4(gamma + c), mu4 = ld4(mean + c), be4 = ld4(beta + c);
        const float4 s14 = ld4(sums + c), s24 = ld4(sums + C + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        const float is[4] = {is4.x, is4.y, is4.z, is4.w}, ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w};
        const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, be[4] = {be4.x, be4.y, be4.z, be4.w};
        const float s1[4] = {s14.x, s14.y, s14.z, s14.w}, s2[4] = {s24.x, s24.y, s24.z, s24.w};
        bool keep[4] = {true, true, true, true};
        if (thr != 0u) keep_quad(seed, (uint64_t)e, thr, keep);     // e is a multiple of 4: one hash for the float4
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - mu[j]) * is[j];
            const float g = bn_dy_pre_k(ds[j], xh, ga[j], be[j], act, dsc, keep[j]);
            o[j] = ga[j] * is[j] * (g - s1[j] * inv_m - xh * s2[j] * inv_m);
        }
        *reinterpret_cast<float4*>(dx + e) = make_float4(o[0], o[1], o[2], o[3]);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    if (amax != nullptr) amax_publish(mx, amax, blockIdx.x * 4 + (threadIdx.x >> 6));    // max|dx| (caller-zeroed slots)
}

static int bn_blocks(long M, int* rows_per_block) {
    int nb = BN_MAXBLK;
    int rpb = cdiv(M, nb);
    if (rpb < 1) rpb = 1;
    nb = cdiv(M, rpb);
    *rows_per_block = rpb;
    return nb;
}

}  // namespace ttts

using namespace ttts;

extern "C" {

int ttts_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int64_t M, int d, float eps, float* y_amax_out, void* y_image_out, float* y_row_inv_out, void* stream) {
    TTTS_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
    TTTS_REQUIRE((y_image_out == nullptr) == (y_row_inv_out == nullptr), "layernorm_fwd: image and row_inv come together");
    unsigned short* img = reinterpret_cast<unsigned short*>(y_image_out);
    TTTS_REQUIRE(M > 0 && d > 0 && d <= 64 * LN_MAXPER, "layernorm_fwd: d=%d must be in 1..%d", d, 64 * LN_MAXPER);
    const bool v4 = (d == 256 || d == 512 || d == 1024) &&
                    ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0;
    if (v4 && d == 256)
        hipLaunchKernelGGL((layernorm_fwd_v4_kernel<1, 4>), dim3(cdiv(M, 16)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                           mean, rstd, (long)M, eps, y_amax_out, img, y_row_inv_out);
    else if (v4 && d == 512)
        hipLaunchKernelGGL((layernorm_fwd_v4_kernel<2, 2>), dim3(cdiv(M, 8)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                           mean, rstd, (long)M, eps, y_amax_out, img, y_row_inv_out);
    else if (v4)
        hipLaunchKernelGGL((layernorm_fwd_v4_kernel<4, 1>), dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                           mean, rstd, (long)M, eps, y_amax_out, img, y_row_inv_out);
    else {
        TTTS_REQUIRE(img == nullptr, "layernorm_fwd: an image output needs d in {256, 512, 1024} and 16-byte aligned operands");
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                           rstd, (long)M, d, eps, y_amax_out);
    }
    TTTS_LAUNCH_CHECK("layernorm_fwd_kernel");
    return TTTS_OK;
}

size_t ttts_layernorm_bwd_workspace_bytes(int d) { return (size_t)LN_BWD_BLOCKS * 2 * d * sizeof(float); }

static int layernorm_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                              float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int64_t M, int d,
                              int accumulate, float* dacc, float drop_p, uint64_t seed, const uint64_t* step_seed,
                              float* dacc_amax, void* image_out, float* row_inv_out, ttts_reduce_queue* queue, hipStream_t stream) {
    TTTS_REQUIRE((image_out == nullptr) == (row_inv_out == nullptr), "layernorm_bwd: image and row_inv come together");
    unsigned short* img = reinterpret_cast<unsigned short*>(image_out);
    TTTS_REQUIRE(dy && x && mean && rstd && gamma && dx && ws, "layernorm_bwd: null pointer");
    TTTS_REQUIRE(M > 0 && d > 0 && d <= 64 * LN_MAXPER, "layernorm_bwd: d=%d must be in 1..%d", d, 64 * LN_MAXPER);
    TTTS_REQUIRE(ws_bytes >= ttts_layernorm_bwd_workspace_bytes(d), "layernorm_bwd: workspace too small");
    int nblk = LN_BWD_BLOCKS;
    if ((long)nblk * 4 > M) nblk = cdiv(M, 4);
    const bool v4 = (d == 256 || d == 512 || d == 1024) &&
                    ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)gamma) | ((uintptr_t)dacc)) & 15) == 0;
    TTTS_REQUIRE(dacc == nullptr || v4, "layernorm_bwd_drop: needs d in {256, 512, 1024} and 16-byte aligned operands");
    if (v4) {
        const uint32_t thr = (dacc && drop_p > 0.f) ? drop_threshold(drop_p) : 0u;
        const float sc = 1.f / (1.f - drop_p);
#define TTTS_LN_BWD4(NV, NW, DROP)                                                                                          \
        hipLaunchKernelGGL((layernorm_bwd_v4_kernel<NV, NW, DROP>), dim3(nblk), dim3(64 * NW), 0, stream, dy, x, mean, rstd, \
                           gamma, dx, ws, (long)M, dacc, sc, thr, seed, step_seed, dacc_amax, img, row_inv_out)
        if (dacc) {
            if (d == 256) TTTS_LN_BWD4(1, 16, true); else if (d == 512) TTTS_LN_BWD4(2, 8, true); else TTTS_LN_BWD4(4, 4, true);
        } else {
            if (d == 256) TTTS_LN_BWD4(1, 16, false); else if (d == 512) TTTS_LN_BWD4(2, 8, false); else TTTS_LN_BWD4(4, 4, false);
        }
#undef TTTS_LN_BWD4
        TTTS_LAUNCH_CHECK("layernorm_bwd_v4_kernel");
        return launch_reduce_rows(ws, 2 * d, nblk, 2 * d, dgamma, d, dbeta, accumulate != 0, stream, queue);
    }
    TTTS_REQUIRE(img == nullptr, "layernorm_bwd: an image output needs d in {256, 512, 1024} and 16-byte aligned operands");
#define TTTS_LN_BWD(NPER)                                                                                         \
    hipLaunchKernelGGL((layernorm_bwd_kernel<NPER>), dim3(nblk), dim3(256), 0, stream, dy, x, mean, rstd, gamma, dx, ws, \
                       (long)M, d)
    const int groups = (d + 63) / 64;          // the instantiation with the next power of two of 64-column groups
    if (groups <= 1) TTTS_LN_BWD(1);
    else if (groups <= 2) TTTS_LN_BWD(2);
    else if (groups <= 4) TTTS_LN_BWD(4);
    else if (groups <= 8) TTTS_LN_BWD(8);
    else TTTS_LN_BWD(16);
#undef TTTS_LN_BWD
    TTTS_LAUNCH_CHECK("layernorm_bwd_kernel");
    return launch_reduce_rows(ws, 2 * d, nblk, 2 * d, dgamma, d, dbeta, accumulate != 0, stream, queue);
}

int ttts_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int64_t M, int d,
                       int accumulate, void* dx_image_out, float* dx_row_inv_out, ttts_reduce_queue* queue, void* stream) {
    return layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, ws, ws_bytes, M, d, accumulate, nullptr, 0.f, 0,
                              nullptr, nullptr, dx_image_out, dx_row_inv_out, queue, (hipStream_t)stream);
}

int ttts_layernorm_bwd_drop(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                            float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int64_t M, int d,
                            int accumulate, float* dacc, float drop_p, uint64_t seed, const uint64_t* step_seed,
                            float* dacc_amax, void* dacc_image_out, float* dacc_row_inv_out, ttts_reduce_queue* queue,
                            void* stream) {
    TTTS_REQUIRE(dacc, "layernorm_bwd_drop: dacc is required");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd_drop: bad dropout p");
    return layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, ws, ws_bytes, M, d, accumulate, dacc, drop_p, seed,
                              step_seed, dacc_amax, dacc_image_out, dacc_row_inv_out, queue, (hipStream_t)stream);
}

size_t ttts_bn_workspace_bytes(int64_t M, int C) {
    (void)M;
    return ((size_t)BN_MAXBLK * 3 + 2) * (size_t)C * sizeof(float);
}

int ttts_bn_train_stats(const float* x, float* mean, float* invstd, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* ws, size_t ws_bytes, int64_t M, int C, float momentum,
                        float eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(x && mean && invstd && ws, "bn_train_stats: null pointer");
    TTTS_REQUIRE(M > 1 && C > 0, "bn_train_stats: need M > 1 rows (got %lld) and C > 0", (long long)M);
    TTTS_REQUIRE(ws_bytes >= ttts_bn_workspace_bytes(M, C), "bn_train_stats: workspace too small");
    int rpb;
    int nb = bn_blocks(M, &rpb);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(cdiv(C, 64), nb), dim3(256), 0, stream, x, ws, (long)M, C, rpb);
    TTTS_LAUNCH_CHECK("bn_stats_partial_kernel");
    const BnStatSet st = {ws, nullptr, mean, invstd, nb, 0};
    hipLaunchKernelGGL(bn_stats_final_kernel<1>, dim3(cdiv(C, 4)), dim3(256), 0, stream, st, st, running_mean, running_var,
                       num_batches_tracked, C, momentum, eps);
    TTTS_LAUNCH_CHECK("bn_stats_final_kernel");
    return TTTS_OK;
}

int ttts_bn_train_stats_from_partials_rows(const float* partials, int nblk, const float* rows, int n_rows, float* mean,
                                           float* invstd, float* running_mean, float* running_var,
                                           int64_t* num_batches_tracked, int C, float momentum, float eps, void* stream) {
    // the second half of ttts_bn_train_stats on row-chunk partials [nblk][3][C] = (count, mean, M2) that somebody else wrote
    // (the fp16x3 convolution's epilogue: ttts_conv1d_fwd_h3 with bn_partials), plus n_rows rows of the matrix itself (row
    // stride C) that no partial of the run covers
    TTTS_REQUIRE(mean && invstd && (partials || nblk == 0) && (rows || n_rows == 0), "bn_train_stats_from_partials: null pointer");
    TTTS_REQUIRE(nblk >= 0 && nblk <= BN_MAXBLK && C > 0, "bn_train_stats_from_partials: nblk=%d must be in 0..%d", nblk, BN_MAXBLK);
    TTTS_REQUIRE(n_rows >= 0 && n_rows <= 64 * BN_XROWS, "bn_train_stats_from_partials: n_rows=%d must be in 0..%d", n_rows, 64 * BN_XROWS);
    TTTS_REQUIRE(nblk + n_rows > 0, "bn_train_stats_from_partials: nothing to merge");
    const BnStatSet st = {partials, rows, mean, invstd, nblk, n_rows};
    hipLaunchKernelGGL(bn_stats_final_kernel<1>, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, st, st, running_mean,
                       running_var, num_batches_tracked, C, momentum, eps);
    TTTS_LAUNCH_CHECK("bn_stats_final_kernel");
    return TTTS_OK;
}

int ttts_bn_train_stats_twin(const float* partials_a, int nblk_a, const float* rows_a, int n_rows_a, float* mean_a, float* invstd_a,
                             const float* partials_b, int nblk_b, const float* rows_b, int n_rows_b, float* mean_b, float* invstd_b,
                             float* running_mean, float* running_var, int64_t* num_batches_tracked, int C, float momentum,
                             float eps, void* stream) {
    // ttts_bn_train_stats_from_partials_rows for TWO row sets of one matrix in one launch: the running statistics (and the batch
    // counter) are updated with set a's statistics first, then with set b's, as two calls in that order would
    TTTS_REQUIRE(mean_a && invstd_a && mean_b && invstd_b && (partials_a || nblk_a == 0) && (rows_a || n_rows_a == 0) &&
                 (partials_b || nblk_b == 0) && (rows_b || n_rows_b == 0), "bn_train_stats_twin: null pointer");
    TTTS_REQUIRE(nblk_a >= 0 && nblk_a <= BN_MAXBLK && nblk_b >= 0 && nblk_b <= BN_MAXBLK && C > 0, "bn_train_stats_twin: nblk out of 0..%d", BN_MAXBLK);
    TTTS_REQUIRE(n_rows_a >= 0 && n_rows_a <= 64 * BN_XROWS && n_rows_b >= 0 && n_rows_b <= 64 * BN_XROWS,
                 "bn_train_stats_twin: n_rows out of 0..%d", 64 * BN_XROWS);
    TTTS_REQUIRE(nblk_a + n_rows_a > 0 && nblk_b + n_rows_b > 0, "bn_train_stats_twin: nothing to merge");
    const BnStatSet sa = {partials_a, rows_a, mean_a, invstd_a, nblk_a, n_rows_a}, sb = {partials_b, rows_b, mean_b, invstd_b, nblk_b, n_rows_b};
    hipLaunchKernelGGL(bn_stats_final_kernel<2>, dim3(cdiv(C, 4)), dim3(512), 0, (hipStream_t)stream, sa, sb, running_mean,
                       running_var, num_batches_tracked, C, momentum, eps);
    TTTS_LAUNCH_CHECK("bn_stats_final_kernel<2>");
    return TTTS_OK;
}

int ttts_bn_train_stats_from_partials(const float* partials, int nblk, float* mean, float* invstd, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, int C, float momentum, float eps,
                                      void* stream) {
    TTTS_REQUIRE(partials && nblk > 0, "bn_train_stats_from_partials: null pointer");
    return ttts_bn_train_stats_from_partials_rows(partials, nblk, nullptr, 0, mean, invstd, running_mean, running_var,
                                                  num_batches_tracked, C, momentum, eps, stream);
}

int ttts_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* invstd, int C, float eps,
                       void* stream) {
    TTTS_REQUIRE(running_mean && running_var && mean && invstd && C > 0, "bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, running_mean,
                       running_var, mean, invstd, C, eps);
    TTTS_LAUNCH_CHECK("bn_eval_stats_kernel");
    return TTTS_OK;
}

int ttts_bn_apply_fwd(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      float* z, int64_t M, int C, int act, float drop_p, uint64_t seed, const uint64_t* step_seed,
                      float* z_amax_out, void* stream) {
    TTTS_REQUIRE(x && mean && invstd && gamma && beta && z, "bn_apply_fwd: null pointer");
    TTTS_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_apply_fwd: C=%d must be a multiple of 4", C);
    TTTS_REQUIRE(act == TTTS_ACT_NONE || act == TTTS_ACT_TANH, "bn_apply_fwd: act must be none or tanh");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "bn_apply_fwd: bad dropout p");
    long n4 = (long)M * C / 4;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta, z,
                       n4, C, act, 1.f / (1.f - drop_p), thr, seed, step_seed, z_amax_out);
    TTTS_LAUNCH_CHECK("bn_apply_fwd_kernel");
    return TTTS_OK;
}

int ttts_bn_bwd(const float* dz, const float* x, const float* mean, const float* invstd, const float* gamma,
                const float* beta, float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, int64_t M, int C,
                int act, float drop_p, uint64_t seed, const uint64_t* step_seed, int accumulate,
                float* dx_amax_partials, int batch_stats, void* stream_) {
    // batch_stats = 1: train-mode BatchNorm (mean / invstd are statistics of x itself, so dx carries their derivative);
    // 0: eval mode (running statistics are constants of the graph: dx = gamma * invstd * g)
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(dz && x && mean && invstd && gamma && beta && dx && ws, "bn_bwd: null pointer");
    TTTS_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "bn_bwd: C=%d must be a multiple of 4", C);
    TTTS_REQUIRE(ws_bytes >= ttts_bn_workspace_bytes(M, C), "bn_bwd: workspace too small");
    int rpb;
    int nb = bn_blocks(M, &rpb);
    uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    float scale = 1.f / (1.f - drop_p);
    float* sums = ws + (size_t)BN_MAXBLK * 3 * C;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cdiv(C, 64), nb), dim3(256), 0, stream, dz, x, mean, invstd, gamma, beta,
                       ws, (long)M, C, rpb, act, scale, thr, seed, step_seed);
    TTTS_LAUNCH_CHECK("bn_bwd_partial_kernel");
    int rc = launch_reduce_rows(ws, 2 * C, nb, 2 * C, sums, 2 * C, nullptr, 0, stream);
    if (rc) return rc;
    long n4 = (long)M * C / 4;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, stream, dz, x, mean, invstd, gamma, beta, sums, dx, n4,
                       C, batch_stats ? 1.0f / (float)M : 0.0f, act, scale, thr, seed, step_seed, dx_amax_partials, dgamma, dbeta,
                       accumulate);
    TTTS_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return TTTS_OK;
}

}  // extern "C"
