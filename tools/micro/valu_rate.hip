// Issue cost of a few VALU instructions on one wave per SIMD (development aid): 8 independent chains, 4096 x 8 instructions.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(64) void k(unsigned* out, unsigned seed, long long* cyc) {
    unsigned v[8];
    unsigned long long w[8];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    f32x16 acc[4]; f16x8 frag;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int r = 0; r < 8; ++r) frag[r] = (_Float16)(float)(seed & 3);
    for (int i = 0; i < 8; ++i) w[i] = seed * 3ull + i;
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 7 + i;
    unsigned c = seed | 0x9E3779B1u;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 4096; ++it) {
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(v[i]));
#define XAD(i) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(c));
#define MADU32(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(v[i]), "v"(c) : "vcc");
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
#define MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "+v"(v[i]) : "v"(c));
#define CVTF32(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
#define CVTSDWA(i) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v[i]));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c));
#define CMP(i) asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(v[i]), "v"(c) : "vcc");
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
#define MFMA(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc[i & 3]) : "v"(frag));
#define MFMA1(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc[0]) : "v"(frag));
#define MFMA2(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc[i & 1]) : "v"(frag));
        if (OP == 0) { REP8(MULLO) }
        if (OP == 1) { REP8(MULHI) }
        if (OP == 2) { REP8(MUL24) }
        if (OP == 3) { REP8(MAD24) }
        if (OP == 4) { REP8(ADD) }
        if (OP == 5) { REP8(XOR) }
        if (OP == 6) { REP8(ALIGN) }
        if (OP == 7) { REP8(XAD) }
        if (OP == 8) { REP8(LSHLADD) }
        if (OP == 9) { REP8(EXP) }
        if (OP == 10) { REP8(CVTPK) }
        if (OP == 11) { REP8(PERM) }
        if (OP == 12) { REP8(MIXLO) }
        if (OP == 13) { REP8(CVTF32) }
        if (OP == 14) { REP8(CVTSDWA) }
        if (OP == 15) { REP8(PKADD) }
        if (OP == 16) { REP8(PKFMA) }
        if (OP == 17) { REP8(MAX3) }
        if (OP == 18) { REP8(CNDMASK) }
        if (OP == 19) { REP8(CMP) }
        if (OP == 20) { REP8(FMA) }
        if (OP == 21) { REP8(MFMA) }
        if (OP == 22) { REP8(MFMA1) }
        if (OP == 23) { REP8(MFMA2) }
    }
    long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= v[i] ^ (unsigned)w[i] ^ (unsigned)(w[i] >> 32);
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s ^= __float_as_uint(acc[i][r]);
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, unsigned* out, long long* cyc) {
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, out, 12345u, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-16s %.2f clk/instr (s_memtime units)\n", name, (double)h / (4096.0 * 8));
}
int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    run<4>("v_add_u32", out, cyc); run<0>("v_mul_lo_u32", out, cyc); run<1>("v_mul_hi_u32", out, cyc);
    run<2>("v_mul_u32_u24", out, cyc); run<3>("v_mad_u32_u24", out, cyc); run<5>("v_xor_b32", out, cyc);
    run<6>("v_alignbit_b32", out, cyc); run<7>("v_xad_u32", out, cyc); run<8>("v_lshl_add_u32", out, cyc);
    run<9>("v_exp_f32", out, cyc); run<10>("v_cvt_pk_f16_f32", out, cyc); run<11>("v_perm_b32", out, cyc);
    run<12>("v_fma_mixlo_f16", out, cyc); run<13>("v_cvt_f32_f16", out, cyc); run<14>("v_cvt_f32_f16_sdwa", out, cyc);
    run<15>("v_pk_add_f32", out, cyc); run<16>("v_pk_fma_f32", out, cyc); run<17>("v_max3_f32", out, cyc);
    run<18>("v_cndmask_b32", out, cyc); run<19>("v_cmp_le_u32", out, cyc); run<20>("v_fma_f32", out, cyc);
    run<21>("v_mfma_32x32x16_f16", out, cyc);
    run<22>("mfma 32x32x16, ONE accumulator chain", out, cyc); run<23>("mfma 32x32x16, two accumulators", out, cyc);
    return 0;
}
