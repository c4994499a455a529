// Input side of the path: the reference's collate_fn (dataset.py:61-103) pads every utterance of
// a batch to the longest one on the host, after __getitem__ has transposed the stored (n_mels, T) spectrogram
// (dataset.py:64).  Here the host only concatenates the utterances as they lie on disk (no transpose, no padding
// bytes over PCIe) and these two kernels produce the padded (B, Tmax, n_mels) fp32 / (B, Pmax) int64 tensors in
// HBM.  HBM-bound: each output element is written once, each input element read once.
#include "ttts_common.h"

namespace ttts {

constexpr int CT = 64;        // frames per workgroup tile
constexpr int CM_MAX = 128;   // n_mels limit

// grid (ceil(Tmax/64), B).  Source utterance b is (n_mels, len_b) row-major starting at float offset
// frame_off[b]*n_mels: reads run along t (contiguous in the source), the LDS tile turns them, and the 64*n_mels
// output floats of a tile are one contiguous run of out.
__global__ __launch_bounds__(256) void collate_mel_kernel(const float* __restrict__ ragged, const int64_t* __restrict__ frame_off,
                                                          float* __restrict__ out, int Tmax, int n_mels) {
    __shared__ float tile[CM_MAX][CT + 1];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * CT;
    const long off = frame_off[b];
    const int len = (int)(frame_off[b + 1] - off);
    const float* src = ragged + off * n_mels;
    const int tl = threadIdx.x & 63, m0 = threadIdx.x >> 6;
    const int nt = min(CT, Tmax - t0);
    if (t0 < len) {   // uniform per workgroup
        for (int m = m0; m < n_mels; m += 4) {
            const int t = t0 + tl;
            tile[m][tl] = (t < len) ? src[(long)m * len + t] : 0.f;
        }
        __syncthreads();
        float* dst = out + ((long)b * Tmax + t0) * n_mels;
        const int total = nt * n_mels;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int t = i / n_mels, m = i - t * n_mels;
            dst[i] = tile[m][t];
        }
    } else {
        float* dst = out + ((long)b * Tmax + t0) * n_mels;
        const int total = nt * n_mels;
        for (int i = threadIdx.x; i < total; i += 256) dst[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void collate_ids_kernel(const int64_t* __restrict__ ragged, const int64_t* __restrict__ off,
                                                          int64_t* __restrict__ out, int B, int Pmax) {
    const long n = (long)B * Pmax;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / Pmax), j = (int)(i - (long)b * Pmax);
        const long o = off[b];
        const int len = (int)(off[b + 1] - o);
        out[i] = (j < len) ? ragged[o + j] : 0;
    }
}

}  // namespace ttts

using namespace ttts;

extern "C" int ttts_collate_melspec(const float* ragged, const int64_t* frame_offsets, float* out, int B, int Tmax,
                                    int n_mels, void* stream) {
    TTTS_REQUIRE(B >= 0 && Tmax >= 0 && n_mels >= 1 && n_mels <= CM_MAX, "collate_melspec: need 1 <= n_mels <= %d (got %d)",
                 CM_MAX, n_mels);
    if (B == 0 || Tmax == 0) return 0;
    TTTS_REQUIRE(ragged && frame_offsets && out, "collate_melspec: null pointer");
    TTTS_REQUIRE(B <= 65535, "collate_melspec: B=%d exceeds the grid limit", B);
    dim3 grid((Tmax + CT - 1) / CT, B);
    hipLaunchKernelGGL(collate_mel_kernel, grid, dim3(256), 0, (hipStream_t)stream, ragged, frame_offsets, out, Tmax, n_mels);
    TTTS_LAUNCH_CHECK("collate_melspec");
    return 0;
}

extern "C" int ttts_collate_phoneme(const int64_t* ragged, const int64_t* offsets, int64_t* out, int B, int Pmax,
                                    void* stream) {
    TTTS_REQUIRE(B >= 0 && Pmax >= 0, "collate_phoneme: negative size");
    if (B == 0 || Pmax == 0) return 0;
    TTTS_REQUIRE(ragged && offsets && out, "collate_phoneme: null pointer");
    long n = (long)B * Pmax;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(collate_ids_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ragged, offsets, out, B, Pmax);
    TTTS_LAUNCH_CHECK("collate_phoneme");
    return 0;
}
