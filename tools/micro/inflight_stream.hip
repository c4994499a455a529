// How fast ONE workgroup per CU streams global memory as a function of the bytes it keeps in flight (development aid; the
// measurement behind DESIGN 9.7 (iii)).  A workgroup of 256 threads issues D x dwordx4 loads per thread (D x 4 KB per CU),
// waits for all of them, adds them up and goes on -- the burst-and-wait rhythm of the GEMM loader's k-tiles.
//   pattern 0: contiguous (each burst = D x 4 KB of consecutive bytes of the workgroup's own region)
//   pattern 1: the GEMM activation pattern: a burst = 256 rows x 128 B at a row pitch of 1 KB (K = 256 floats), D / 8 such
//              k-tiles side by side (D = 8: one k-tile = 32 KB; D = 16: two)
// Two footprints: 2 GB (HBM) and 96 MB read over and over (Infinity Cache).  `active` of the 256 workgroups do the work, the
// others exit (how the rate depends on how many CUs pull at once).
// build: hipcc --offload-arch=gfx950 -O3 -o inflight_stream inflight_stream.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int D, int PATTERN>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, long region_f4, int bursts, int active,
                                                     float* __restrict__ out) {
    if ((int)blockIdx.x >= active) return;
    const float4* base = src + (long)blockIdx.x * region_f4;
    const int tid = threadIdx.x;
    float acc = 0.f;
    long pos = 0;                                           // float4 offset of the burst inside the region
    for (int b = 0; b < bursts; ++b) {
        float4 v[D];
        if (PATTERN == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) v[i] = base[pos + (long)i * 256 + tid];
            pos += (long)D * 256;
            if (pos + (long)D * 256 > region_f4) pos = 0;
        } else {
            // thread -> (row = tid >> 3 (+ 32 per piece), 16-byte chunk = tid & 7) of a 256-row x 32-float k-tile; 8 pieces per
            // k-tile; row pitch 64 float4 (1 KB)
            const int row0 = tid >> 3, ch = tid & 7;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int kt = i >> 3, piece = i & 7;
                v[i] = base[pos + (long)(row0 + 32 * piece) * 64 + kt * 8 + ch];
            }
            pos += (D / 8) * 8;                             // the next k-tiles of the same rows
            if ((pos & 63) == 0) pos += 255 * 64;           // a row block done (64 float4 per row): next 256 rows
            if (pos + 256 * 64 > region_f4) pos = 0;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (acc == 12345.678f) out[blockIdx.x * 256 + tid] = acc;   // (keeps the loads alive)
}

template <int D, int PATTERN>
static void run(const float4* buf, long region_f4, int active, float* out, const char* what) {
    const int bursts = 2048 * 8 / D;                        // 64 MB per workgroup whatever D
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<D, PATTERN>), dim3(256), dim3(256), 0, 0, buf, region_f4, bursts / 8, active, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<D, PATTERN>), dim3(256), dim3(256), 0, 0, buf, region_f4, bursts, active, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)bursts * D * 4096.0;
    const double gbs_cu = bytes_cu / (ms * 1e-3) / 1e9;
    // cycles a burst takes at a nominal 2.1 GHz
    printf("%-10s pattern %d  active %3d  in flight %3d KB/CU : %6.1f GB/s per CU  %6.2f TB/s chip  burst %6.0f ns\n", what,
           PATTERN, active, D * 4, gbs_cu, gbs_cu * active / 1e3, ms * 1e6 / bursts);
}

int main() {
    const long total_bytes = 2L << 30;
    float4* buf; float* out;
    if (hipMalloc(&buf, total_bytes) != hipSuccess || hipMalloc(&out, 256 * 256 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, total_bytes);
    const long big = total_bytes / 16 / 256;                // float4 per workgroup: 8 MB regions (2 GB footprint)
    const long small = (96L << 20) / 16 / 256;              // 384 KB regions (96 MB footprint: stays in the Infinity Cache)
    for (int active : {256, 224, 128, 64}) {
        run<4, 0>(buf, big, active, out, "HBM");   run<8, 0>(buf, big, active, out, "HBM");
        run<16, 0>(buf, big, active, out, "HBM");  run<24, 0>(buf, big, active, out, "HBM");
        run<32, 0>(buf, big, active, out, "HBM");
        run<8, 1>(buf, big, active, out, "HBM");   run<16, 1>(buf, big, active, out, "HBM");
        run<32, 1>(buf, big, active, out, "HBM");
        if (active == 256 || active == 64) {
            run<8, 0>(buf, small, active, out, "InfCache");  run<16, 0>(buf, small, active, out, "InfCache");
            run<32, 0>(buf, small, active, out, "InfCache");
            run<16, 1>(buf, small, active, out, "InfCache"); run<32, 1>(buf, small, active, out, "InfCache");
        }
    }
    return 0;
}
