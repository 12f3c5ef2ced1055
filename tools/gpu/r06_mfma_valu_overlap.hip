// Does a SIMD of gfx950 run a wave's VALU instructions while an MFMA (of the same wave, or of another wave on the SIMD) is in flight?
// Each variant loops over 4 MFMAs (v_mfma_f32_32x32x16_bf16, four independent accumulators) with N independent VALU instructions
// (v_fma_f32 or v_exp_f32) behind each, one or two waves per SIMD, accumulators in AGPRs or VGPRs; cycles per MFMA from s_memtime.
//   hipcc --offload-arch=gfx950 -O2 -o tools/gpu/bin/r06_mfma_valu_overlap tools/gpu/r06_mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define REP4(X) X X X X
#define VALU1 "v_fma_f32 %[x0], %[x0], %[y], %[y]\n\t"
#define VALU2 VALU1 "v_fma_f32 %[x1], %[x1], %[y], %[y]\n\t"
#define VALU4 VALU2 "v_fma_f32 %[x2], %[x2], %[y], %[y]\n\tv_fma_f32 %[x3], %[x3], %[y], %[y]\n\t"
#define VALU6 VALU4 "v_fma_f32 %[x4], %[x4], %[y], %[y]\n\tv_fma_f32 %[x5], %[x5], %[y], %[y]\n\t"
#define VALU8 VALU6 "v_fma_f32 %[x6], %[x6], %[y], %[y]\n\tv_fma_f32 %[x7], %[x7], %[y], %[y]\n\t"
#define EXP2 "v_exp_f32 %[x0], %[x0]\n\tv_exp_f32 %[x1], %[x1]\n\t"

// MODE: 0 = MFMA only, 1 = VALU only, 2 = interleaved.  ACC: 'a' or 'v' accumulators; operands in VGPRs.
#define KERNEL(NAME, ACCC, MF, VA)                                                                                            \
    __global__ __launch_bounds__(512) void NAME(float* out, long long* cyc, int iters) {                                   \
        f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};                                                                              \
        f4v a = {1.f, 2.f, 3.f, 4.f}, b = {1.f, 1.f, 1.f, 1.f};                                                               \
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f, y = 0.5f;                \
        const long long t0 = __builtin_readcyclecounter();                                                                    \
        for (int i = 0; i < iters; ++i) {                                                                                     \
            asm volatile(MF("%[c0]") VA MF("%[c1]") VA MF("%[c2]") VA MF("%[c3]") VA                                           \
                         : [c0] "+" ACCC(c0), [c1] "+" ACCC(c1), [c2] "+" ACCC(c2), [c3] "+" ACCC(c3), [x0] "+v"(x0), [x1] "+v"(x1), \
                           [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6), [x7] "+v"(x7)             \
                         : [a] "v"(a), [b] "v"(b), [y] "v"(y));                                                               \
        }                                                                                                                     \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                                    \
        const long long t1 = __builtin_readcyclecounter();                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;    \
    }
#define MFMA(C) "v_mfma_f32_32x32x16_bf16 " C ", %[a], %[b], " C "\n\t"
#define NOMF(C) ""
#define NOVA ""
KERNEL(k_mfma_a, "a", MFMA, NOVA)
KERNEL(k_mfma_v, "v", MFMA, NOVA)
KERNEL(k_valu2, "v", NOMF, VALU2)
KERNEL(k_valu4, "v", NOMF, VALU4)
KERNEL(k_valu8, "v", NOMF, VALU8)
KERNEL(k_exp2, "v", NOMF, EXP2)
KERNEL(k_both_a2, "a", MFMA, VALU2)
KERNEL(k_both_a4, "a", MFMA, VALU4)
KERNEL(k_both_a6, "a", MFMA, VALU6)
KERNEL(k_both_a8, "a", MFMA, VALU8)
KERNEL(k_both_v4, "v", MFMA, VALU4)
KERNEL(k_both_v8, "v", MFMA, VALU8)
KERNEL(k_both_a_exp2, "a", MFMA, EXP2)
KERNEL(k_both_v_exp2, "v", MFMA, EXP2)

// dependent chains: the same four MFMAs per iteration over 1, 2 or 4 accumulators; operands from VGPRs or AGPRs
#define KCHAIN(NAME, ACCC, SRC, C0, C1, C2, C3)                                                                               \
    __global__ __launch_bounds__(512) void NAME(float* out, long long* cyc, int iters) {                                      \
        f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};                                                                              \
        f4v a = {1.f, 2.f, 3.f, 4.f}, b = {1.f, 1.f, 1.f, 1.f};                                                               \
        const long long t0 = __builtin_readcyclecounter();                                                                    \
        for (int i = 0; i < iters; ++i) {                                                                                     \
            asm volatile(MFMA(C0) MFMA(C1) MFMA(C2) MFMA(C3)                                                                  \
                         : [c0] "+" ACCC(c0), [c1] "+" ACCC(c1), [c2] "+" ACCC(c2), [c3] "+" ACCC(c3) : [a] SRC(a), [b] SRC(b)); \
        }                                                                                                                     \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                                    \
        const long long t1 = __builtin_readcyclecounter();                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];                                           \
    }
KCHAIN(k_ch1_v, "v", "v", "%[c0]", "%[c0]", "%[c0]", "%[c0]")
KCHAIN(k_ch2_v, "v", "v", "%[c0]", "%[c1]", "%[c0]", "%[c1]")
KCHAIN(k_ch4_v, "v", "v", "%[c0]", "%[c1]", "%[c2]", "%[c3]")
KCHAIN(k_ch1_a, "a", "v", "%[c0]", "%[c0]", "%[c0]", "%[c0]")
KCHAIN(k_ch2_a, "a", "v", "%[c0]", "%[c1]", "%[c0]", "%[c1]")
KCHAIN(k_ch2_v_srca, "v", "a", "%[c0]", "%[c1]", "%[c0]", "%[c1]")
KCHAIN(k_ch4_v_srca, "v", "a", "%[c0]", "%[c1]", "%[c2]", "%[c3]")
KCHAIN(k_ch4_a_srca, "a", "a", "%[c0]", "%[c1]", "%[c2]", "%[c3]")

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1024 * 1024 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    struct V { const char* name; void (*k)(float*, long long*, int); } vs[] = {
        {"mfma only, AGPR acc", k_mfma_a}, {"mfma only, VGPR acc", k_mfma_v}, {"2 fma only", k_valu2}, {"4 fma only", k_valu4}, {"8 fma only", k_valu8},
        {"2 exp only", k_exp2}, {"mfma(AGPR) + 2 fma", k_both_a2}, {"mfma(AGPR) + 4 fma", k_both_a4}, {"mfma(AGPR) + 6 fma", k_both_a6},
        {"mfma(AGPR) + 8 fma", k_both_a8}, {"mfma(VGPR) + 4 fma", k_both_v4}, {"mfma(VGPR) + 8 fma", k_both_v8}, {"mfma(AGPR) + 2 exp", k_both_a_exp2},
        {"mfma(VGPR) + 2 exp", k_both_v_exp2},
        {"1 chain, VGPR acc", k_ch1_v}, {"2 chains, VGPR acc", k_ch2_v}, {"4 chains, VGPR acc", k_ch4_v}, {"1 chain, AGPR acc", k_ch1_a},
        {"2 chains, AGPR acc", k_ch2_a}, {"2 chains VGPR, A/B AGPR", k_ch2_v_srca}, {"4 chains VGPR, A/B AGPR", k_ch4_v_srca},
        {"4 chains AGPR, A/B AGPR", k_ch4_a_srca}};
    for (int threads : {256, 512}) {
        printf("---- %d waves per SIMD (one workgroup of %d threads per CU, 256 workgroups)\n", threads / 256, threads);
        for (auto& v : vs) {
            hipLaunchKernelGGL(v.k, 256, threads, 0, 0, out, cyc, 100);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(v.k, 256, threads, 0, 0, out, cyc, iters);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            long long c = 0; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-24s %7.1f s_memtime ticks and %6.2f ns per group (1 MFMA + its VALU) of one wave\n", v.name, (double)c / iters / 4.0, ms * 1e6 / iters / 4.0);
        }
    }
    return 0;
}
