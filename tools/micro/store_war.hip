// How soon after a `buffer_store_dwordx4` may the wave overwrite the store's DATA registers?   (gfx950, ROCm 7.2)
//
// Round 5 found gemm_h3i's head-image epilogue storing, on some lanes of some waves, the PRODUCT of the next packed multiply in
// the second dword of a 16-byte store: hipcc had placed the `v_pk_mul_f32` that recycles the data registers two VALU
// instructions behind the store -- the wait states its hazard table asks for.  Round 4's 16x16x32 GEMM port died of the same
// signature ("first dword of some float4s = the row's scale factor").  This kernel issues, from inline assembly so that nothing is
// padded for us,
//      buffer_store_dwordx4 v[100:103] ; K fillers ; v_pk_mul_f32 v[100:101] *= 2 ; v_pk_mul_f32 v[102:103] *= 2
// with K = 0..7 fillers of two kinds (s_nop 0, or an independent VALU instruction as a compiler would put there), on a chip
// whose every CU is storing, with and without LDS reads in flight, and counts the stored dwords that came out doubled.
// Round 6: the same with the store's row offset in an SGPR soffset (`sgpr_soffset` rows) -- the form for which hipcc inserts NO
// wait states at all (GCNHazardRecognizer::createsVALUHazard exempts MUBUF stores with a register soffset), which is what the
// head-image epilogue's damaged stores had in common.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_war.hip -o tools/micro/store_war && tools/micro/store_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP0(x)
#define REP1(x) x
#define REP2(x) x x
#define REP3(x) x x x
#define REP4(x) x x x x
#define REP5(x) x x x x x
#define REP6(x) x x x x x x
#define REP7(x) x x x x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP12(x) REP8(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

// MODE 0: s_nop fillers; 1: VALU fillers (v_add_f32 on an unrelated register)
#define KERNEL(NAME, FILL, SOFF, VOFF_IT)                                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, int lds_traffic) {                                   \
        __shared__ float lbuf[4096];                                                                                        \
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                         \
        const long wid = (long)blockIdx.x * 4 + wave;                                                                       \
        const uint64_t base = reinterpret_cast<uint64_t>(out + wid * (long)iters * 256);                                    \
        const u32x4 rsrc = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base),                                   \
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(base >> 32) & 0xffffu)),                  \
                            (uint32_t)iters * 1024u, 0x00020000u};                                                            \
        for (int i = threadIdx.x; i < 4096; i += 256) lbuf[i] = (float)i;                                                   \
        __syncthreads();                                                                                                    \
        float junk = 0.f, fill = 1.0f;                                                                                      \
        const f32x2 two = {2.0f, 2.0f};                                                                          \
        for (int it = 0; it < iters; ++it) {                                                                                \
            const float p0 = 1.0f + lane + 64.f * (it & 63), p1 = p0 + 0.25f, p2 = p0 + 0.5f, p3 = p0 + 0.75f;              \
            const uint32_t voff = (uint32_t)it * (VOFF_IT) + (uint32_t)lane * 16u;                                          \
            const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(it * (1024 - (VOFF_IT)));                        \
            float4 l = make_float4(0.f, 0.f, 0.f, 0.f);                                                                     \
            if (lds_traffic) l = *reinterpret_cast<const float4*>(lbuf + ((lane * 4 + it * 64) & 4092));                    \
            asm volatile("v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\t"          \
                         "s_nop 4\n\t"                                                                                     \
                         "buffer_store_dwordx4 v[100:103], %6, %7, " SOFF " offen\n\t" FILL                                        \
                         "v_pk_mul_f32 v[100:101], v[100:101], %8\n\t"                                                     \
                         "v_pk_mul_f32 v[102:103], v[102:103], %8\n\t"                                                     \
                         "v_add_f32 %0, v100, v103\n\t"                                                                      \
                         : "+v"(junk), "+v"(fill)                                                                            \
                         : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(voff), "s"(rsrc), "v"(two), "s"(soff)                       \
                         : "v100", "v101", "v102", "v103", "memory");                                                      \
            junk += l.x + l.y + l.z + l.w;                                                                                  \
        }                                                                                                                   \
        if (junk == 12345.678f) out[0] = junk + fill;                                                                       \
    }

#define NOP(k) REP##k("s_nop 0\n\t")
#define VAL(k) REP##k("v_add_f32 %1, %1, %1\n\t")
#define KX(...) KERNEL(__VA_ARGS__)
#define IMM "0", 1024u
#define SGP "%9", 0u          /* the row offset rides in an SGPR soffset: the form LLVM's hazard recogniser exempts */
KX(kn0, NOP(0), IMM) KX(kn1, NOP(1), IMM) KX(kn2, NOP(2), IMM) KX(kn3, NOP(3), IMM) KX(kn4, NOP(4), IMM) KX(kn5, NOP(5), IMM)
KX(kn6, NOP(6), IMM) KX(kn7, NOP(7), IMM) KX(kn8, NOP(8), IMM) KX(kn12, NOP(12), IMM) KX(kn16, NOP(16), IMM)
KX(kv0, VAL(0), IMM) KX(kv1, VAL(1), IMM) KX(kv2, VAL(2), IMM) KX(kv3, VAL(3), IMM) KX(kv4, VAL(4), IMM) KX(kv5, VAL(5), IMM)
KX(kv6, VAL(6), IMM) KX(kv7, VAL(7), IMM) KX(kv8, VAL(8), IMM) KX(kv12, VAL(12), IMM) KX(kv16, VAL(16), IMM)
KX(ks0, NOP(0), SGP) KX(ks1, NOP(1), SGP) KX(ks2, NOP(2), SGP) KX(ks3, NOP(3), SGP) KX(ks4, NOP(4), SGP)

typedef void (*kern_t)(float*, int, int);

int main() {
    const int blocks = 1024, iters = 128;
    const size_t n = (size_t)blocks * 4 * iters * 256;
    float* d;
    hipMalloc(&d, n * 4);
    std::vector<float> h(n);
    struct { const char* name; kern_t k; int fill; } ks[] = {
        {"s_nop", kn0, 0}, {"s_nop", kn1, 1}, {"s_nop", kn2, 2}, {"s_nop", kn3, 3}, {"s_nop", kn4, 4}, {"s_nop", kn5, 5}, {"s_nop", kn6, 6},
        {"s_nop", kn7, 7}, {"s_nop", kn8, 8}, {"s_nop", kn12, 12}, {"s_nop", kn16, 16},
        {"valu", kv0, 0}, {"valu", kv1, 1}, {"valu", kv2, 2}, {"valu", kv3, 3}, {"valu", kv4, 4}, {"valu", kv5, 5}, {"valu", kv6, 6},
        {"valu", kv7, 7}, {"valu", kv8, 8}, {"valu", kv12, 12}, {"valu", kv16, 16},
        {"sgpr_soffset+s_nop", ks0, 0}, {"sgpr_soffset+s_nop", ks1, 1}, {"sgpr_soffset+s_nop", ks2, 2}, {"sgpr_soffset+s_nop", ks3, 3},
        {"sgpr_soffset+s_nop", ks4, 4}};
    printf("# buffer_store_dwordx4 v[100:103]; K fillers; v_pk_mul_f32 v[100:101]; v_pk_mul_f32 v[102:103]  -- %d workgroups x 4 waves x %d stores\n", blocks, iters);
    printf("# filler K lds_reads  stored_dwords  doubled(dword0 dword1 dword2 dword3)  other_wrong\n");
    for (auto& e : ks)
        for (int lds = 0; lds < 2; ++lds) {
            hipMemset(d, 0, n * 4);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters, lds);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
            long dbl[4] = {0, 0, 0, 0}, other = 0;
            for (size_t w = 0; w < (size_t)blocks * 4; ++w)
                for (int it = 0; it < iters; ++it)
                    for (int lane = 0; lane < 64; ++lane) {
                        const float p0 = 1.0f + lane + 64.f * (it & 63);
                        const float* c = &h[(w * iters + it) * 256 + lane * 4];
                        for (int q = 0; q < 4; ++q) {
                            const float want = p0 + 0.25f * q;
                            if (c[q] == want) continue;
                            if (c[q] == 2.f * want) ++dbl[q]; else ++other;
                        }
                    }
            printf("%-18s %2d %d  %zu  %ld %ld %ld %ld  %ld\n", e.name, e.fill, lds, n, dbl[0], dbl[1], dbl[2], dbl[3], other);
        }
    hipFree(d);
    return 0;
}
