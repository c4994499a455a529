#include <hip/hip_runtime.h>
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_mix(float x0, float x1, float scale, uint32_t& hi, uint32_t& lo) {
    uint32_t h, l;
    // hi = f16(x * scale)  (low / high half), lo = f16(x * scale - hi)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "v"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(scale), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(scale), "v"(h));
    hi = h; lo = l;
}
__device__ __forceinline__ void split2_ref(float x0, float x1, float scale, uint32_t& hi, uint32_t& lo) {
    f32x2 x = f32x2{x0, x1} * scale;
    const f16x2v h = __builtin_convertvector(x, f16x2v);
    const f32x2 r = x - __builtin_convertvector(h, f32x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2v));
}
__global__ void k(const float* x, uint32_t* out, int n, float scale) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    uint32_t h, l, h2, l2;
    split2_mix(x[2 * i], x[2 * i + 1], scale, h, l);
    split2_ref(x[2 * i], x[2 * i + 1], scale, h2, l2);
    out[4 * i] = h; out[4 * i + 1] = l; out[4 * i + 2] = h2; out[4 * i + 3] = l2;
}
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cmath>
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n);
    srand(1);
    for (int i = 0; i < n; ++i) { float m = (float)rand() / RAND_MAX * 2 - 1; int e = rand() % 40 - 30; hx[i] = ldexpf(m, e); }
    hx[0] = 0.f; hx[1] = -0.f; hx[2] = 65504.f / 16; hx[3] = 1e-12f; hx[4] = 70000.f; hx[5] = -70000.f;
    float* dx; uint32_t* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 8);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, dout, n, 16.0f);
    std::vector<uint32_t> ho(2 * n);
    hipMemcpy(ho.data(), dout, n * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n / 2; ++i) if (ho[4 * i] != ho[4 * i + 2] || ho[4 * i + 1] != ho[4 * i + 3]) { if (bad < 10) printf("mismatch at %d: x=%g,%g mix %08x %08x ref %08x %08x\n", i, hx[2*i], hx[2*i+1], ho[4*i], ho[4*i+1], ho[4*i+2], ho[4*i+3]); ++bad; }
    printf("mismatches: %ld of %d pairs\n", bad, n / 2);
    return 0;
}
