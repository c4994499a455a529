// Small row-wise / element-wise pieces of the path: embedding gather + scatter-add, scaled positional
// encoding, dropout / relu backward masks, the 1-wide stop-token head.  All HBM-bound; float4 accesses,
// grid-stride loops capped at 8 blocks per CU, fixed-order reductions.
#include <stdarg.h>

#include "ttts_common.h"

namespace ttts {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static inline int ew_grid(long n_items) {
    long g = (n_items + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------ zero fill
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, long n) {
    // 16-byte stores over the aligned middle, dwords at the ragged ends
    const long head = (long)(((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2) < n ? (long)(((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2) : n;
    const long n4 = (n - head) / 4;
    uint4* q = reinterpret_cast<uint4*>(p + head);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        if ((long)threadIdx.x < head) p[threadIdx.x] = 0u;
        const long tail0 = head + n4 * 4;
        if (tail0 + threadIdx.x < n) p[tail0 + threadIdx.x] = 0u;
    }
}

// A fill KERNEL, not hipMemsetAsync: the zeroing of the gradient bucket and of the partial-maxima arrays is part of the
// captured step graph, and graph memset nodes proved fragile on ROCm 7 (after ANOTHER graph's memory pool had been released
// with hipFree, replays of a surviving graph saw arrays that were not zeroed; reproduced by tools/dbg_two_models.py).
int launch_zero(void* p, size_t nbytes, hipStream_t stream) {
    TTTS_REQUIRE(p != nullptr || nbytes == 0, "zero: null pointer");
    if (nbytes == 0) return TTTS_OK;
    TTTS_REQUIRE((((uintptr_t)p) & 3) == 0 && nbytes % 4 == 0, "zero: pointer and size must be multiples of 4 bytes");
    const long n = (long)(nbytes / 4);
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(p), n);
    TTTS_LAUNCH_CHECK("zero_kernel");
    return TTTS_OK;
}

// ------------------------------------------------------------------ head padding (attention for head_dim < 64)
// dst[r][h*64 + c] = c < hd ? src[r*ld_src + h*hd + c] : 0   (pad)      dst[r*ld_dst + h*hd + c] = src[r][h*64 + c]   (unpad)
__global__ __launch_bounds__(256) void heads_pad_kernel(const float* __restrict__ src, long ld_src, float* __restrict__ dst,
                                                        long rows, int H, int hd, int unpad, long ld_dst) {
    const long n = rows * H * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 63);
        const long rh = i >> 6;
        const int h = (int)(rh % H);
        const long r = rh / H;
        if (unpad) {
            if (c < hd) dst[r * ld_dst + h * hd + c] = src[i];
        } else {
            dst[i] = c < hd ? src[r * ld_src + h * hd + c] : 0.f;
        }
    }
}

// ------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                            float* __restrict__ out, long n, int vocab, int d4,
                                                            float* __restrict__ amax_out) {
    // one wave per output row, float4 per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = 0.f;
    for (long row = (long)blockIdx.x * 4 + wave; row < n; row += (long)gridDim.x * 4) {
        long id = ids[row];
        if (id < 0) id = 0;
        if (id >= vocab) id = vocab - 1;
        const float4* src = reinterpret_cast<const float4*>(table) + id * d4;
        float4* dst = reinterpret_cast<float4*>(out) + row * d4;
        for (int c = lane; c < d4; c += 64) {
            const float4 v = src[c];
            dst[c] = v;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    if (amax_out != nullptr) amax_publish(m, amax_out, blockIdx.x * 4 + wave);
}

// one workgroup per vocabulary row.  Positions holding this id are compacted IN ORDER into LDS, 1024 ids per step (four
// independent id loads per lane, wave ballots + prefix counts), then the matching gradient rows are summed in that fixed
// order: deterministic.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                            float* __restrict__ dtable, long n, int d, int accumulate) {
    constexpr int PER = 4;                                  // ids per lane and step
    __shared__ int list[256 * PER];
    __shared__ int wcount[PER][4];
    const int v = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int MAXPER = 4;   // d <= 1024
    float acc[MAXPER] = {0.f, 0.f, 0.f, 0.f};
    for (long base = 0; base < n; base += 256 * PER) {
        bool hit[PER];
        unsigned long long m[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {                     // position base + 256 p + tid: ordered by (p, wave, lane)
            const long i = base + 256 * p + tid;
            hit[p] = (i < n) && (ids[i] == v);
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            m[p] = __ballot(hit[p]);
            if (lane == 0) wcount[p][wave] = __popcll(m[p]);
        }
        __syncthreads();
        int total = 0;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int off = total;
            for (int w = 0; w < wave; ++w) off += wcount[p][w];
            if (hit[p]) list[off + __popcll(m[p] & ((1ull << lane) - 1ull))] = 256 * p + tid;
            total += wcount[p][0] + wcount[p][1] + wcount[p][2] + wcount[p][3];
        }
        __syncthreads();
        int k = 0;
        for (; k + 8 <= total; k += 8) {                    // eight rows in flight, added in list order
            float g[8][MAXPER];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long row = base + list[k + u];
#pragma unroll
                for (int j = 0; j < MAXPER; ++j) {
                    const int c = tid + 256 * j;
                    g[u][j] = c < d ? dout[row * d + c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < MAXPER; ++j) acc[j] += g[u][j];
        }
        for (; k < total; ++k) {
            const long row = base + list[k];
#pragma unroll
            for (int j = 0; j < MAXPER; ++j) {
                int c = tid + 256 * j;
                if (c < d) acc[j] += dout[row * d + c];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < MAXPER; ++j) {
        int c = tid + 256 * j;
        if (c < d) {
            long o = (long)v * d + c;
            dtable[o] = accumulate ? dtable[o] + acc[j] : acc[j];
        }
    }
}

// ------------------------------------------------------------------ positional encoding
__global__ __launch_bounds__(256) void posenc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                         const float* __restrict__ alpha, float* __restrict__ y, long n4,
                                                         int T, int d, float drop_scale, uint32_t thr, uint64_t seed, const uint64_t* step_seed,
                                                         float* __restrict__ amax_out) {
    seed = site_seed(seed, step_seed);
    const float a = alpha[0];
    const long td = (long)T * d;
    float ymax = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const long pe_off = e % td;   // (t, c) offset inside the (T, d) table
        float4 xv = *reinterpret_cast<const float4*>(x + e);
        float4 pv = *reinterpret_cast<const float4*>(pe + pe_off);
        float o[4] = {xv.x + a * pv.x, xv.y + a * pv.y, xv.z + a * pv.z, xv.w + a * pv.w};
        if (thr != 0u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = keep_elem(seed, (uint64_t)(e + j), thr) ? o[j] * drop_scale : 0.f;
        }
        *reinterpret_cast<float4*>(y + e) = make_float4(o[0], o[1], o[2], o[3]);
        ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    if (amax_out != nullptr) amax_publish(ymax, amax_out, blockIdx.x * 4 + (threadIdx.x >> 6));
}

constexpr int PE_BWD_BLOCKS = 1024;

// dx = dy*keep/(1-p); block partial of sum(dx * pe) -> ws[block]
__global__ __launch_bounds__(256) void posenc_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pe,
                                                         float* __restrict__ dx, float* __restrict__ ws, long n4, int T,
                                                         int d, float drop_scale, uint32_t thr, uint64_t seed, const uint64_t* step_seed) {
    seed = site_seed(seed, step_seed);
    __shared__ double red[4];
    const long td = (long)T * d;
    // The sum over every position cancels heavily (d alpha is 5-50 x smaller than its terms' mass): accumulated in double, so that
    // what is left of its error is what dy itself carries, not this kernel's summation order (the pass is memory-bound either way)
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        float4 gv = *reinterpret_cast<const float4*>(dy + e);
        float4 pv = *reinterpret_cast<const float4*>(pe + (e % td));
        float g[4] = {gv.x, gv.y, gv.z, gv.w};
        if (thr != 0u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = keep_elem(seed, (uint64_t)(e + j), thr) ? g[j] * drop_scale : 0.f;
        }
        s += (double)g[0] * pv.x + (double)g[1] * pv.y + (double)g[2] * pv.z + (double)g[3] * pv.w;
        *reinterpret_cast<float4*>(dx + e) = make_float4(g[0], g[1], g[2], g[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<double*>(ws)[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void scalar_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int n, int accumulate) {
    // single wave, fixed order; double partials
    const double* w8 = reinterpret_cast<const double*>(ws);
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += w8[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + (float)s : (float)s;      // (as autograd adds two fp32 results)
}

// ------------------------------------------------------------------ masks
// Both mask kernels can also leave the partial maxima of |dx| behind (amax != NULL: a caller-zeroed TTTS_AMAX_SLOTS array,
// slot-wise atomic maxima) and save the consumer the separate pass of ttts_amax_partials.
__device__ __forceinline__ void block_amax_out(float m, float* __restrict__ amax) {
    amax_publish(m, amax, blockIdx.x * 4 + (threadIdx.x >> 6));
}

__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                               float* __restrict__ dx, long n4, float scale,
                                                               float* __restrict__ amax) {
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        float4 o = reinterpret_cast<const float4*>(out)[i];
        float4 r;
        r.x = o.x > 0.f ? g.x * scale : 0.f;
        r.y = o.y > 0.f ? g.y * scale : 0.f;
        r.z = o.z > 0.f ? g.z * scale : 0.f;
        r.w = o.w > 0.f ? g.w * scale : 0.f;
        reinterpret_cast<float4*>(dx)[i] = r;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    }
    if (amax != nullptr) block_amax_out(mx, amax);
}

__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long n4,
                                                          float scale, uint32_t thr, uint64_t seed, const uint64_t* step_seed,
                                                          float* __restrict__ amax) {
    seed = site_seed(seed, step_seed);
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        float4 r;
        r.x = keep_elem(seed, (uint64_t)e, thr) ? g.x * scale : 0.f;
        r.y = keep_elem(seed, (uint64_t)(e + 1), thr) ? g.y * scale : 0.f;
        r.z = keep_elem(seed, (uint64_t)(e + 2), thr) ? g.z * scale : 0.f;
        r.w = keep_elem(seed, (uint64_t)(e + 3), thr) ? g.w * scale : 0.f;
        reinterpret_cast<float4*>(dx)[i] = r;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    }
    if (amax != nullptr) block_amax_out(mx, amax);
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                  float* __restrict__ z, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(x)[i];
        float4 b = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(z)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                   const float* __restrict__ w, float* __restrict__ z, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        const float4 b = reinterpret_cast<const float4*>(y)[i];
        const float4 c = reinterpret_cast<const float4*>(w)[i];
        reinterpret_cast<float4*>(z)[i] = make_float4((a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z, (a.w + b.w) + c.w);
    }
}

// ------------------------------------------------------------------ stop-token head (N = 1)
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, long M, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float bias = b ? b[0] : 0.f;
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
        const float* xr = x + row * d;
        float s = 0.f;
        for (int c = lane; c < d; c += 64) s += xr[c] * w[c];
        s = wave_sum(s);
        if (lane == 0) y[row] = s + bias;
    }
}

constexpr int RD_BWD_BLOCKS = 1024;

// dx[m,:] += dy[m]*w ; per-block partials of dw[c] = sum_m dy[m]*x[m,c] and db = sum_m dy[m] -> ws[block][d+1].
// One wave per row, a float4 of 4 consecutive columns per lane (d <= 1024, d % 4 == 0), two rows in flight per wave:
// the kernel is a read-modify-write stream over dx, so what matters is independent loads in flight.
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ w, float* __restrict__ dx,
                                                         float* __restrict__ ws, long M, int d) {
    __shared__ float red[4][1025];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MAXV = 4;                       // float4s per lane: columns lane*4 + 256*i
    float4 acc[MAXV], wv[MAXV];
    float accb = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        wv[i] = (c < d) ? *reinterpret_cast<const float4*>(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long stride = (long)gridDim.x * 4;
    for (long row = (long)blockIdx.x * 4 + wave; row < M; row += 2 * stride) {
        const long row2 = row + stride;
        const bool two = row2 < M;
        const float g0 = dy[row], g1 = two ? dy[row2] : 0.f;
        accb += g0 + g1;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < d) {                           // wave-uniform per i except in the last, partial group of lanes
                const float4 x0 = *reinterpret_cast<const float4*>(x + row * d + c);
                const float4 x1 = two ? *reinterpret_cast<const float4*>(x + row2 * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc[i].x += g0 * x0.x + g1 * x1.x; acc[i].y += g0 * x0.y + g1 * x1.y;
                acc[i].z += g0 * x0.z + g1 * x1.z; acc[i].w += g0 * x0.w + g1 * x1.w;
                if (dx) {
                    float4* p0 = reinterpret_cast<float4*>(dx + row * d + c);
                    float4 a0 = *p0;
                    a0.x += g0 * wv[i].x; a0.y += g0 * wv[i].y; a0.z += g0 * wv[i].z; a0.w += g0 * wv[i].w;
                    if (two) {
                        float4* p1 = reinterpret_cast<float4*>(dx + row2 * d + c);
                        float4 a1 = *p1;
                        a1.x += g1 * wv[i].x; a1.y += g1 * wv[i].y; a1.z += g1 * wv[i].z; a1.w += g1 * wv[i].w;
                        *p1 = a1;
                    }
                    *p0 = a0;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < d) { red[wave][c] = acc[i].x; red[wave][c + 1] = acc[i].y; red[wave][c + 2] = acc[i].z; red[wave][c + 3] = acc[i].w; }
    }
    if (lane == 0) red[wave][1024] = accb;
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256)
        ws[(long)blockIdx.x * (d + 1) + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (threadIdx.x == 0)
        ws[(long)blockIdx.x * (d + 1) + d] = (red[0][1024] + red[1][1024]) + (red[2][1024] + red[3][1024]);
}

}  // namespace ttts

using namespace ttts;

extern "C" {

const char* ttts_last_error(void) { return ttts::g_err; }
int ttts_abi_version(void) { return 14; }

int ttts_zero(void* p, size_t nbytes, void* stream) { return ::ttts::launch_zero(p, nbytes, (hipStream_t)stream); }

int ttts_heads_pad(const float* src, int64_t ld_src, float* dst, int64_t rows, int H, int head_dim, void* stream) {
    TTTS_REQUIRE(src && dst && rows > 0 && H > 0 && head_dim > 0 && head_dim <= 64 && ld_src >= (int64_t)H * head_dim,
                 "heads_pad: bad arguments (head_dim %d must be in 1..64)", head_dim);
    hipLaunchKernelGGL(heads_pad_kernel, dim3(ew_grid(rows * H * 64)), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, dst,
                       (long)rows, H, head_dim, 0, 0L);
    TTTS_LAUNCH_CHECK("heads_pad_kernel");
    return TTTS_OK;
}

int ttts_heads_unpad(const float* src, float* dst, int64_t ld_dst, int64_t rows, int H, int head_dim, void* stream) {
    TTTS_REQUIRE(src && dst && rows > 0 && H > 0 && head_dim > 0 && head_dim <= 64 && ld_dst >= (int64_t)H * head_dim,
                 "heads_unpad: bad arguments (head_dim %d must be in 1..64)", head_dim);
    hipLaunchKernelGGL(heads_pad_kernel, dim3(ew_grid(rows * H * 64)), dim3(256), 0, (hipStream_t)stream, src, 0L, dst, (long)rows,
                       H, head_dim, 1, (long)ld_dst);
    TTTS_LAUNCH_CHECK("heads_pad_kernel");
    return TTTS_OK;
}

int ttts_embedding_fwd(const int64_t* ids, const float* table, float* out, int64_t n, int vocab, int d, float* out_amax_out,
                       void* stream) {
    TTTS_REQUIRE(ids && table && out, "embedding_fwd: null pointer");
    TTTS_REQUIRE(n > 0 && vocab > 0 && d > 0 && d % 4 == 0, "embedding_fwd: d=%d must be a multiple of 4", d);
    int grid = (int)((n + 3) / 4);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, table, out, (long)n, vocab,
                       d / 4, out_amax_out);
    TTTS_LAUNCH_CHECK("embedding_fwd_kernel");
    return TTTS_OK;
}

int ttts_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, int64_t n, int vocab, int d, int accumulate,
                       void* stream) {
    TTTS_REQUIRE(ids && dout && dtable, "embedding_bwd: null pointer");
    TTTS_REQUIRE(n > 0 && vocab > 0 && d > 0 && d <= 1024, "embedding_bwd: d=%d must be <= 1024", d);
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(vocab), dim3(256), 0, (hipStream_t)stream, ids, dout, dtable, (long)n, d,
                       accumulate);
    TTTS_LAUNCH_CHECK("embedding_bwd_kernel");
    return TTTS_OK;
}

int ttts_posenc_fwd(const float* x, const float* pe, const float* alpha, float* y, int B, int T, int d, float drop_p,
                    uint64_t seed, const uint64_t* step_seed, float* y_amax_out, void* stream) {
    TTTS_REQUIRE(x && pe && alpha && y, "posenc_fwd: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && d > 0 && d % 4 == 0, "posenc_fwd: d=%d must be a multiple of 4", d);
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "posenc_fwd: bad dropout p");
    long n4 = (long)B * T * d / 4;
    uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    hipLaunchKernelGGL(posenc_fwd_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, pe, alpha, y, n4, T, d,
                       1.f / (1.f - drop_p), thr, seed, step_seed, y_amax_out);
    TTTS_LAUNCH_CHECK("posenc_fwd_kernel");
    return TTTS_OK;
}

size_t ttts_posenc_bwd_workspace_bytes(void) { return (size_t)PE_BWD_BLOCKS * sizeof(double); }

int ttts_posenc_bwd(const float* dy, const float* pe, float* dx, float* dalpha, float* ws, size_t ws_bytes, int B, int T,
                    int d, float drop_p, uint64_t seed, const uint64_t* step_seed, int accumulate,
                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(dy && pe && dx && dalpha && ws, "posenc_bwd: null pointer");
    TTTS_REQUIRE(B > 0 && T > 0 && d > 0 && d % 4 == 0, "posenc_bwd: bad dims");
    TTTS_REQUIRE(ws_bytes >= ttts_posenc_bwd_workspace_bytes(), "posenc_bwd: workspace too small");
    long n4 = (long)B * T * d / 4;
    int grid = ew_grid(n4);
    if (grid > PE_BWD_BLOCKS) grid = PE_BWD_BLOCKS;
    uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    hipLaunchKernelGGL(posenc_bwd_kernel, dim3(grid), dim3(256), 0, stream, dy, pe, dx, ws, n4, T, d, 1.f / (1.f - drop_p),
                       thr, seed, step_seed);
    TTTS_LAUNCH_CHECK("posenc_bwd_kernel");
    hipLaunchKernelGGL(scalar_reduce_kernel, dim3(1), dim3(64), 0, stream, ws, dalpha, grid, accumulate);
    TTTS_LAUNCH_CHECK("scalar_reduce_kernel");
    return TTTS_OK;
}

int ttts_relu_dropout_bwd(const float* dy, const float* out, float* dx, int64_t n, float drop_p, float* amax_partials,
                          void* stream) {
    TTTS_REQUIRE(dy && out && dx && n > 0 && n % 4 == 0, "relu_dropout_bwd: bad arguments (n %% 4 must be 0)");
    TTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "relu_dropout_bwd: bad dropout p");
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                       dy, out, dx, (long)(n / 4), 1.f / (1.f - drop_p), amax_partials);
    TTTS_LAUNCH_CHECK("relu_dropout_bwd_kernel");
    return TTTS_OK;
}

int ttts_dropout_bwd(const float* dy, float* dx, int64_t n, float drop_p, uint64_t seed, const uint64_t* step_seed,
                     float* amax_partials, void* stream) {
    TTTS_REQUIRE(dy && dx && n > 0 && n % 4 == 0, "dropout_bwd: bad arguments (n %% 4 must be 0)");
    TTTS_REQUIRE(drop_p > 0.f && drop_p < 1.f, "dropout_bwd: p must be in (0,1)");
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, dy,
                       dx, (long)(n / 4), 1.f / (1.f - drop_p), drop_threshold(drop_p), seed, step_seed, amax_partials);
    TTTS_LAUNCH_CHECK("dropout_bwd_kernel");
    return TTTS_OK;
}

int ttts_add(const float* x, const float* y, float* z, int64_t n, void* stream) {
    TTTS_REQUIRE(x && y && z && n > 0 && n % 4 == 0, "add: bad arguments (n %% 4 must be 0)");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, z, (long)(n / 4));
    TTTS_LAUNCH_CHECK("add_kernel");
    return TTTS_OK;
}

int ttts_add3(const float* x, const float* y, const float* w, float* z, int64_t n, void* stream) {
    TTTS_REQUIRE(x && y && w && z && n > 0 && n % 4 == 0, "add3: bad arguments (n %% 4 must be 0)");
    hipLaunchKernelGGL(add3_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, w, z, (long)(n / 4));
    TTTS_LAUNCH_CHECK("add3_kernel");
    return TTTS_OK;
}

int ttts_rowdot_fwd(const float* x, const float* w, const float* b, float* y, int64_t M, int d, void* stream) {
    TTTS_REQUIRE(x && w && y && M > 0 && d > 0, "rowdot_fwd: bad arguments");
    int grid = (int)((M + 3) / 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, b, y, (long)M, d);
    TTTS_LAUNCH_CHECK("rowdot_fwd_kernel");
    return TTTS_OK;
}

size_t ttts_rowdot_bwd_workspace_bytes(int d) { return (size_t)RD_BWD_BLOCKS * (d + 1) * sizeof(float); }

int ttts_rowdot_bwd(const float* dy, const float* x, const float* w, float* dx_accum, float* dw, float* db, float* ws,
                    size_t ws_bytes, int64_t M, int d, int accumulate, ttts_reduce_queue* queue, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TTTS_REQUIRE(dy && x && w && ws, "rowdot_bwd: null pointer");
    TTTS_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 1024, "rowdot_bwd: d=%d must be a multiple of 4, <= 1024", d);
    TTTS_REQUIRE(ws_bytes >= ttts_rowdot_bwd_workspace_bytes(d), "rowdot_bwd: workspace too small");
    int nblk = RD_BWD_BLOCKS;
    if ((long)nblk * 4 > M) nblk = (int)((M + 3) / 4);
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(nblk), dim3(256), 0, stream, dy, x, w, dx_accum, ws, (long)M, d);
    TTTS_LAUNCH_CHECK("rowdot_bwd_kernel");
    return launch_reduce_rows(ws, d + 1, nblk, d + 1, dw, d, db, accumulate != 0, stream, queue);
}

}  // extern "C"
