// Deterministic second-stage reductions shared by the weight-gradient, bias-gradient, LayerNorm and BatchNorm
// backward kernels: partial results laid out [nrows][ncols] are summed over rows in a fixed order (no atomics),
// and either stored or added to the destination (`accumulate`: gradients land directly in a flat gradient buffer).
//
// Deferred form: a training step has ~90 of these (one per weight / bias / LayerNorm parameter pair), each a few
// microseconds of work behind a launch.  Given a caller-owned `ttts_reduce_queue` the launchers append a descriptor to it
// on the host instead of launching, and ttts_reduce_queue_flush runs them all in ONE launch per 48 descriptors (the table
// travels in the kernel-argument segment: nothing to copy, nothing to keep alive, and a captured HIP graph replays it as it
// stands).  Every form sums in the same order whether deferred or not.  The library keeps no queue of its own: a queue
// belongs to whoever created it (one per gradient bucket in the Python host code), and independent callers on different
// threads / streams use different queues.
#include "ttts_common.h"
#include <mutex>
#include <new>
#include <vector>

namespace ttts {

enum { RED_NARROW = 0, RED_WIDE = 1, RED_CONV = 2 };

struct ReduceDesc {
    const float* ws;
    float* out0;
    float* out1;       // narrow form only: columns >= n0 go here (may be NULL: dropped)
    int ld;            // row stride of ws in floats (conv form: floats per split)
    int ncols;         // narrow / conv: columns; wide: float4 columns
    int n0;            // narrow: split point between out0 and out1; conv: cout * cin
    int nrows;
    int kind;
    int accumulate;
    int first_block;
    int taps;          // conv form
    int pad;
};
static_assert(sizeof(ReduceDesc) == 64, "descriptor layout");

constexpr int RED_BATCH = 48;
struct ReduceBatch {
    ReduceDesc d[RED_BATCH];
    int n;
};

// narrow outputs (a few hundred .. few thousand columns, up to a few hundred rows): 16 columns x 16 row-lanes
__device__ __forceinline__ void reduce_narrow_body(const float* __restrict__ ws, long ld, int nrows, long ncols,
                                                   float* __restrict__ out0, long n0, float* __restrict__ out1,
                                                   int accumulate, int block, float (*red)[17]) {
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long c = (long)block * 16 + cl;
    float s = 0.f;
    if (c < ncols) {
        // eight independent loads in flight per lane (a few hundred rows: the loop used to be one dependent trip to L2 per row,
        // 11 us for BatchNorm's 511 x 512 backward sums); fixed order all the same
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = rl;
        for (; r + 7 * 16 < nrows; r += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += ws[(long)(r + 16 * u) * ld + c];
        }
        for (; r < nrows; r += 16) a[0] += ws[(long)r * ld + c];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        float* o = (c < n0) ? (out0 + c) : (out1 != nullptr ? out1 + (c - n0) : nullptr);
        if (o != nullptr) *o = accumulate ? (*o + t) : t;
    }
}

// wide outputs (weight matrices): one float4 of columns per thread, rows unrolled by 4
__device__ __forceinline__ void reduce_wide_body(const float* __restrict__ ws, long ld, int nrows, long ncols4,
                                                 float* __restrict__ out, int accumulate, int block) {
    const long i = (long)block * 256 + threadIdx.x;
    if (i >= ncols4) return;
    const float4* p = reinterpret_cast<const float4*>(ws) + i;
    const long ld4 = ld >> 2;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int r = 0;
    for (; r + 4 <= nrows; r += 4) {
        float4 v0 = p[(long)r * ld4], v1 = p[(long)(r + 1) * ld4], v2 = p[(long)(r + 2) * ld4], v3 = p[(long)(r + 3) * ld4];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; r < nrows; ++r) {
        float4 v0 = p[(long)r * ld4];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    float4 s = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                           (a0.w + a1.w) + (a2.w + a3.w));
    float4* o = reinterpret_cast<float4*>(out) + i;
    if (accumulate) {
        float4 t = *o;
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *o = s;
}

// conv weight gradient: ws[split][tap][co][ci] -> dw[co][ci][tap] (+= when accumulate)
__device__ __forceinline__ void reduce_conv_body(const float* __restrict__ ws, float* __restrict__ dw, long per, int taps,
                                                 int nsplit, int accumulate, int block) {
    const long i = (long)block * 256 + threadIdx.x;   // index into [tap][co][ci]
    const long n = per * taps;
    if (i >= n) return;
    const int tap = (int)(i / per);
    const long rem = i - (long)tap * per;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 4 <= nsplit; z += 4) {
        s0 += ws[(long)z * n + i];
        s1 += ws[(long)(z + 1) * n + i];
        s2 += ws[(long)(z + 2) * n + i];
        s3 += ws[(long)(z + 3) * n + i];
    }
    for (; z < nsplit; ++z) s0 += ws[(long)z * n + i];
    const float s = (s0 + s1) + (s2 + s3);
    const long o = rem * taps + tap;
    dw[o] = accumulate ? dw[o] + s : s;
}

__global__ __launch_bounds__(256) void reduce_rows_narrow_kernel(const float* __restrict__ ws, long ld, int nrows,
                                                                 long ncols, float* __restrict__ out0, long n0,
                                                                 float* __restrict__ out1, int accumulate) {
    __shared__ float red[16][17];
    reduce_narrow_body(ws, ld, nrows, ncols, out0, n0, out1, accumulate, blockIdx.x, red);
}

__global__ __launch_bounds__(256) void reduce_rows_wide_kernel(const float* __restrict__ ws, long ld, int nrows, long ncols4,
                                                               float* __restrict__ out, int accumulate) {
    reduce_wide_body(ws, ld, nrows, ncols4, out, accumulate, blockIdx.x);
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                int cout, int cin, int taps, int nsplit, int accumulate) {
    reduce_conv_body(ws, dw, (long)cout * cin, taps, nsplit, accumulate, blockIdx.x);
}

// a weight matrix (wide form above) AND its bias vector in ONE launch: blocks [0, nb_wide) reduce the matrix partials,
// the blocks behind them the ncols2 bias partials (16 columns x 16 row-lanes per block, as the narrow kernel)
__global__ __launch_bounds__(256) void reduce_rows_pair_kernel(const float* __restrict__ ws, long ld, int nrows, long ncols4,
                                                               float* __restrict__ out, int nb_wide,
                                                               const float* __restrict__ ws2, long ld2, long ncols2,
                                                               float* __restrict__ out2, int accumulate) {
    __shared__ float red[16][17];
    if ((int)blockIdx.x < nb_wide) reduce_wide_body(ws, ld, nrows, ncols4, out, accumulate, blockIdx.x);
    else reduce_narrow_body(ws2, ld2, nrows, ncols2, out2, ncols2, nullptr, accumulate, (int)blockIdx.x - nb_wide, red);
}

// every queued reduction of a backward pass: the block finds its descriptor (first_block ascending, wave-uniform scan
// of the kernel-argument table) and runs the body of the kernel that would have been launched for it
__global__ __launch_bounds__(256) void reduce_batched_kernel(const ReduceBatch b) {
    __shared__ float red[16][17];
    const int bi = blockIdx.x;
    int k = 0;
    for (int i = 1; i < b.n; ++i)
        if (b.d[i].first_block <= bi) k = i;
    const ReduceDesc& D = b.d[k];
    const int block = bi - D.first_block;
    if (D.kind == RED_WIDE) reduce_wide_body(D.ws, D.ld, D.nrows, D.ncols, D.out0, D.accumulate, block);
    else if (D.kind == RED_CONV) reduce_conv_body(D.ws, D.out0, D.n0, D.taps, D.nrows, D.accumulate, block);
    else reduce_narrow_body(D.ws, D.ld, D.nrows, D.ncols, D.out0, D.n0, D.out1, D.accumulate, block, red);
}

// ------------------------------------------------------------------------------------------ host side
}  // namespace ttts
// the opaque handle of include/ttts_hip.h: descriptors appended by the *_bwd_* entry points that were given this queue.
// The mutex only protects the vector: autograd runs backward nodes on worker threads while the owner flushes from its own.
struct ttts_reduce_queue {
    std::mutex mu;
    std::vector<ttts::ReduceDesc> q;
};
namespace ttts {

static bool defer_push(ttts_reduce_queue* queue, ReduceDesc d, long blocks) {
    if (queue == nullptr) return false;
    std::lock_guard<std::mutex> lock(queue->mu);
    d.first_block = (int)blocks;        // block COUNT for now; flush turns the counts into prefix sums per batch
    d.pad = 0;
    queue->q.push_back(d);
    return true;
}

static bool fits_int(long v) { return v >= 0 && v < (1L << 31); }

static bool wide_ok(const float* ws, long ld, long ncols, const float* out) {
    return ncols >= 8192 && (ncols % 4) == 0 && (ld % 4) == 0 && ((((uintptr_t)ws) | ((uintptr_t)out)) & 15) == 0;
}

static ReduceDesc narrow_desc(const float* ws, long ld, int nrows, long ncols, float* out0, long n0, float* out1, int accumulate) {
    ReduceDesc d{};
    d.ws = ws; d.out0 = out0; d.out1 = out1; d.ld = (int)ld; d.ncols = (int)ncols; d.n0 = (int)(n0 < ncols ? n0 : ncols);
    d.nrows = nrows; d.kind = RED_NARROW; d.accumulate = accumulate;
    return d;
}
static ReduceDesc wide_desc(const float* ws, long ld, int nrows, long ncols4, float* out, int accumulate) {
    ReduceDesc d{};
    d.ws = ws; d.out0 = out; d.ld = (int)ld; d.ncols = (int)ncols4; d.nrows = nrows; d.kind = RED_WIDE; d.accumulate = accumulate;
    return d;
}

// out (+)= column sums of ws [nrows][ncols] and out2 (+)= column sums of ws2 [nrows][ncols2]; one launch when the first is
// a wide (weight-matrix) reduction, two otherwise.  Same summation order as launch_reduce_rows for either part.
int launch_reduce_rows_pair(const float* ws, long ld, int nrows, long ncols, float* out, const float* ws2, long ld2,
                            long ncols2, float* out2, int accumulate, hipStream_t stream, ttts_reduce_queue* queue) {
    const bool wide = wide_ok(ws, ld, ncols, out);
    if (!wide || out2 == nullptr) {
        int rc = launch_reduce_rows(ws, ld, nrows, ncols, out, ncols, nullptr, accumulate, stream, queue);
        if (rc || out2 == nullptr) return rc;
        return launch_reduce_rows(ws2, ld2, nrows, ncols2, out2, ncols2, nullptr, accumulate, stream, queue);
    }
    const long n4 = ncols / 4;
    const int nb_wide = cdiv(n4, 256);
    if (queue != nullptr && fits_int(ld) && fits_int(ld2) && fits_int(ncols)) {
        defer_push(queue, wide_desc(ws, ld, nrows, n4, out, accumulate), nb_wide);
        defer_push(queue, narrow_desc(ws2, ld2, nrows, ncols2, out2, ncols2, nullptr, accumulate), cdiv(ncols2, 16));
        return TTTS_OK;
    }
    hipLaunchKernelGGL(reduce_rows_pair_kernel, dim3(nb_wide + cdiv(ncols2, 16)), dim3(256), 0, stream, ws, ld, nrows, n4, out,
                       nb_wide, ws2, ld2, ncols2, out2, accumulate);
    TTTS_LAUNCH_CHECK("reduce_rows_pair_kernel");
    return TTTS_OK;
}

int launch_reduce_rows(const float* ws, long ld, int nrows, long ncols, float* out0, long n0, float* out1, int accumulate,
                       hipStream_t stream, ttts_reduce_queue* queue) {
    const bool wide = out1 == nullptr && n0 >= ncols && wide_ok(ws, ld, ncols, out0);
    const bool q = queue != nullptr && fits_int(ld) && fits_int(ncols);
    if (wide) {
        long n4 = ncols / 4;
        if (q && defer_push(queue, wide_desc(ws, ld, nrows, n4, out0, accumulate), cdiv(n4, 256))) return TTTS_OK;
        hipLaunchKernelGGL(reduce_rows_wide_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, stream, ws, ld, nrows, n4, out0,
                           accumulate);
    } else {
        if (q && defer_push(queue, narrow_desc(ws, ld, nrows, ncols, out0, n0, out1, accumulate), cdiv(ncols, 16))) return TTTS_OK;
        hipLaunchKernelGGL(reduce_rows_narrow_kernel, dim3(cdiv(ncols, 16)), dim3(256), 0, stream, ws, ld, nrows, ncols,
                           out0, n0, out1, accumulate);
    }
    TTTS_LAUNCH_CHECK("reduce_rows_kernel");
    return TTTS_OK;
}

int launch_conv_wgrad_reduce(const float* ws, float* dw, int cout, int cin, int taps, int nsplit, int accumulate,
                             hipStream_t stream, ttts_reduce_queue* queue) {
    const long n = (long)cout * cin * taps;
    if (queue != nullptr && fits_int(n)) {
        ReduceDesc d{};
        d.ws = ws; d.out0 = dw; d.n0 = cout * cin; d.taps = taps; d.nrows = nsplit; d.kind = RED_CONV; d.accumulate = accumulate;
        d.ncols = (int)n;
        if (defer_push(queue, d, cdiv(n, 256))) return TTTS_OK;
    }
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, ws, dw, cout, cin, taps, nsplit,
                       accumulate);
    TTTS_LAUNCH_CHECK("conv_wgrad_reduce_kernel");
    return TTTS_OK;
}

}  // namespace ttts

using namespace ttts;

extern "C" {

ttts_reduce_queue* ttts_reduce_queue_create(void) { return new (std::nothrow) ttts_reduce_queue(); }

void ttts_reduce_queue_destroy(ttts_reduce_queue* queue) { delete queue; }

int64_t ttts_reduce_queue_pending(ttts_reduce_queue* queue) {
    if (queue == nullptr) return 0;
    std::lock_guard<std::mutex> lock(queue->mu);
    return (int64_t)queue->q.size();
}

int ttts_reduce_queue_clear(ttts_reduce_queue* queue) {
    TTTS_REQUIRE(queue, "reduce_queue_clear: null queue");
    std::lock_guard<std::mutex> lock(queue->mu);
    queue->q.clear();
    return TTTS_OK;
}

int ttts_reduce_queue_flush(ttts_reduce_queue* queue, void* stream_) {
    // run everything appended since the last flush.  Entries of one launch run concurrently, so a second reduction into a
    // destination already in the batch (a parameter used twice in the pass) starts a new launch: queue order is kept per
    // destination.
    TTTS_REQUIRE(queue, "reduce_queue_flush: null queue");
    std::vector<ReduceDesc> q;
    {
        std::lock_guard<std::mutex> lock(queue->mu);
        q.swap(queue->q);
    }
    hipStream_t stream = (hipStream_t)stream_;
    for (size_t at = 0; at < q.size();) {
        ReduceBatch b;
        b.n = 0;
        long blocks = 0;
        while (at < q.size() && b.n < RED_BATCH) {
            const ReduceDesc& d = q[at];
            bool clash = false;
            for (int i = 0; i < b.n && !clash; ++i)
                clash = b.d[i].out0 == d.out0 || (d.out1 && (b.d[i].out1 == d.out1 || b.d[i].out0 == d.out1)) ||
                        (b.d[i].out1 && b.d[i].out1 == d.out0);
            if (clash) break;
            b.d[b.n] = d;
            b.d[b.n].first_block = (int)blocks;
            blocks += d.first_block;
            ++b.n;
            ++at;
        }
        TTTS_REQUIRE(blocks > 0 && blocks < (1L << 31), "reduce_queue_flush: bad block count");
        hipLaunchKernelGGL(reduce_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b);
        TTTS_LAUNCH_CHECK("reduce_batched_kernel");
    }
    return TTTS_OK;
}

}  // extern "C"
