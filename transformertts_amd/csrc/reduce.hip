// Deterministic second-stage reductions shared by the weight-gradient, bias-gradient, LayerNorm and BatchNorm
// backward kernels: partial results laid out [nrows][ncols] are summed over rows in a fixed order (no atomics),
// and either stored or added to the destination (`accumulate`: gradients land directly in a flat gradient buffer).
#include "ttts_common.h"

namespace ttts {

int launch_reduce_rows(const float* ws, long ld, int nrows, long ncols, float* out0, long n0, float* out1, int accumulate,
                       hipStream_t stream);

// narrow outputs (a few hundred .. few thousand columns, up to a few hundred rows): 16 columns x 16 row-lanes
__global__ __launch_bounds__(256) void reduce_rows_narrow_kernel(const float* __restrict__ ws, long ld, int nrows,
                                                                 long ncols, float* __restrict__ out0, long n0,
                                                                 float* __restrict__ out1, int accumulate) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long c = (long)blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < ncols) {
        for (int r = rl; r < nrows; r += 16) s += ws[(long)r * ld + c];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        float* o = (c < n0) ? (out0 + c) : (out1 + (c - n0));
        if (o != nullptr) *o = accumulate ? (*o + t) : t;
    }
}

// wide outputs (weight matrices): one float4 of columns per thread, rows unrolled by 4
__global__ __launch_bounds__(256) void reduce_rows_wide_kernel(const float* __restrict__ ws, long ld, int nrows, long ncols4,
                                                               float* __restrict__ out, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
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

// a weight matrix (wide form above) AND its bias vector in ONE launch: blocks [0, nb_wide) reduce the matrix partials,
// the blocks behind them the ncols2 bias partials (16 columns x 16 row-lanes per block, as the narrow kernel)
__global__ __launch_bounds__(256) void reduce_rows_pair_kernel(const float* __restrict__ ws, long ld, int nrows, long ncols4,
                                                               float* __restrict__ out, int nb_wide,
                                                               const float* __restrict__ ws2, long ld2, long ncols2,
                                                               float* __restrict__ out2, int accumulate) {
    if ((int)blockIdx.x < nb_wide) {
        const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
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
        return;
    }
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long c = (long)((int)blockIdx.x - nb_wide) * 16 + cl;
    float s = 0.f;
    if (c < ncols2) {
        for (int r = rl; r < nrows; r += 16) s += ws2[(long)r * ld2 + c];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < ncols2) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        out2[c] = accumulate ? (out2[c] + t) : t;
    }
}

// out (+)= column sums of ws [nrows][ncols] and out2 (+)= column sums of ws2 [nrows][ncols2]; one launch when the first is
// a wide (weight-matrix) reduction, two otherwise.  Same summation order as launch_reduce_rows for either part.
int launch_reduce_rows_pair(const float* ws, long ld, int nrows, long ncols, float* out, const float* ws2, long ld2,
                            long ncols2, float* out2, int accumulate, hipStream_t stream) {
    const bool wide = ncols >= 8192 && (ncols % 4) == 0 && (ld % 4) == 0 && ((((uintptr_t)ws) | ((uintptr_t)out)) & 15) == 0;
    if (!wide || out2 == nullptr) {
        int rc = launch_reduce_rows(ws, ld, nrows, ncols, out, ncols, nullptr, accumulate, stream);
        if (rc || out2 == nullptr) return rc;
        return launch_reduce_rows(ws2, ld2, nrows, ncols2, out2, ncols2, nullptr, accumulate, stream);
    }
    const long n4 = ncols / 4;
    const int nb_wide = cdiv(n4, 256);
    hipLaunchKernelGGL(reduce_rows_pair_kernel, dim3(nb_wide + cdiv(ncols2, 16)), dim3(256), 0, stream, ws, ld, nrows, n4, out,
                       nb_wide, ws2, ld2, ncols2, out2, accumulate);
    TTTS_LAUNCH_CHECK("reduce_rows_pair_kernel");
    return TTTS_OK;
}

int launch_reduce_rows(const float* ws, long ld, int nrows, long ncols, float* out0, long n0, float* out1, int accumulate,
                       hipStream_t stream) {
    const bool wide = ncols >= 8192 && out1 == nullptr && n0 >= ncols && (ncols % 4) == 0 && (ld % 4) == 0 &&
                      ((((uintptr_t)ws) | ((uintptr_t)out0)) & 15) == 0;
    if (wide) {
        long n4 = ncols / 4;
        hipLaunchKernelGGL(reduce_rows_wide_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, stream, ws, ld, nrows, n4, out0,
                           accumulate);
    } else {
        hipLaunchKernelGGL(reduce_rows_narrow_kernel, dim3(cdiv(ncols, 16)), dim3(256), 0, stream, ws, ld, nrows, ncols,
                           out0, n0, out1, accumulate);
    }
    TTTS_LAUNCH_CHECK("reduce_rows_kernel");
    return TTTS_OK;
}

}  // namespace ttts
