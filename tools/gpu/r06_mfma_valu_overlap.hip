// Does a SIMD of gfx950 run a wave's VALU instructions while an MFMA (of the same wave, or of another wave on the SIMD) is in flight?
// Each variant loops over 4 MFMAs (v_mfma_f32_32x32x16_bf16, four independent accumulators) with N independent VALU instructions
// (v_fma_f32 or v_exp_f32) behind each, one or two waves per SIMD, accumulators in AGPRs or VGPRs; cycles per MFMA from s_memtime.
//   hipcc --offload-arch=gfx950 -O2 -o tools/gpu/bin/r06_mfma_valu_overlap tools/gpu/r06_mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

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

// memory instructions behind the MFMAs of one wave: per 4 MFMAs, ND LDS-DMA instructions (global_load_lds_dwordx4, 1 KB each, from a
// 64 KB L2-resident source), or ND ds_read_b128
template <int KIND, int ND>
__global__ __launch_bounds__(512) void k_mem(float* out, long long* cyc, int iters, const uint4* src) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f4v a = {1.f, 2.f, 3.f, 4.f}, b = {1.f, 1.f, 1.f, 1.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* my = src + (blockIdx.x & 3) * 1024 + lane;
    uint4 sink = {0, 0, 0, 0};
    u4v wdata = {(unsigned)lane, 1u, 2u, 3u};
    const unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + wave * 8192;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        asm volatile(MFMA("%[c0]") MFMA("%[c1]") MFMA("%[c2]") MFMA("%[c3]")
                     : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3) : [a] "v"(a), [b] "v"(b));
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (KIND == 0) __builtin_amdgcn_global_load_lds((const void*)(my + 64 * ((i + d) & 15)), (__attribute__((address_space(3))) void*)(uintptr_t)(ldsb + 1024 * d), 16, 0, 0);
            else if (KIND == 3) { asm volatile("ds_write_b128 %0, %1" :: "v"(ldsb + lane * 16 + 1024 * d), "v"(wdata) : "memory"); }
            else { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ldsb + lane * 16 + 1024 * d) : "memory"); (void)v; }
        }
        if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + sink.x;
}

// the same with plain 16-byte global loads into registers (8 rotating register sets, loads waited for 4 iterations later)
template <int ND>
__global__ __launch_bounds__(512) void k_gload(float* out, long long* cyc, int iters, const uint4* src) {
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f4v a = {1.f, 2.f, 3.f, 4.f}, b = {1.f, 1.f, 1.f, 1.f};
    const int lane = threadIdx.x & 63;
    const uint4* my = src + (blockIdx.x & 3) * 1024 + lane;
    uint4 v[8][ND];
    unsigned acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile(MFMA("%[c0]") MFMA("%[c1]") MFMA("%[c2]") MFMA("%[c3]")
                         : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3) : [a] "v"(a), [b] "v"(b));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * ND) : "memory");
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[u][d]) : "v"(my + 64 * ((u + d) & 15)) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int d = 0; d < ND; ++d) acc ^= v[u][d].x;
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (float)acc;
}

// register staging: per 4 MFMAs ND plain loads into a ring of 8 register sets and ND ds_write_b128 of the set loaded 4 iterations earlier
template <int ND>
__global__ __launch_bounds__(512) void k_stage(float* out, long long* cyc, int iters, const uint4* src) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f4v a = {1.f, 2.f, 3.f, 4.f}, b = {1.f, 1.f, 1.f, 1.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* my = src + (blockIdx.x & 3) * 1024 + lane;
    const unsigned ldsb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + wave * 8192 + lane * 16;
    u4v v[8][ND];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int d = 0; d < ND; ++d) v[u][d] = u4v{(unsigned)u, (unsigned)d, 0u, 0u};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile(MFMA("%[c0]") MFMA("%[c1]") MFMA("%[c2]") MFMA("%[c3]")
                         : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3) : [a] "v"(a), [b] "v"(b));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * ND) : "memory");            // the set loaded 4 iterations ago has landed
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ldsb), "v"(v[(u + 4) & 7][d]), "n"(1024 * d) : "memory");
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[u][d]) : "v"(my + 64 * ((u + d) & 15)) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15" ::: "memory");
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int d = 0; d < ND; ++d) acc ^= v[u][d].x;
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + (float)acc;
}

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
    {
        uint4* src; (void)hipMalloc(&src, 4 * 1024 * 16 + 65536); (void)hipMemset(src, 0, 4 * 1024 * 16 + 65536);
        struct M { const char* name; void (*k)(float*, long long*, int, const uint4*); } ms[] = {
            {"4 mfma, no memory op", k_mem<0, 0>}, {"4 mfma + 1 LDS-DMA (1 KB)", k_mem<0, 1>}, {"4 mfma + 2 LDS-DMA", k_mem<0, 2>}, {"4 mfma + 4 LDS-DMA", k_mem<0, 4>},
            {"4 mfma + 1 global_load x4", k_gload<1>}, {"4 mfma + 2 global_load x4", k_gload<2>}, {"4 mfma + 4 global_load x4", k_gload<4>},
            {"4 mfma + 1 ds_write_b128", k_mem<3, 1>}, {"4 mfma + 2 ds_write_b128", k_mem<3, 2>}, {"4 mfma + 4 ds_write_b128", k_mem<3, 4>},
            {"4 mfma + 1 (load + ds_write)", k_stage<1>}, {"4 mfma + 2 (load + ds_write)", k_stage<2>},
            {"4 mfma + 2 ds_read_b128", k_mem<2, 2>}, {"4 mfma + 4 ds_read_b128", k_mem<2, 4>}, {"4 mfma + 8 ds_read_b128", k_mem<2, 8>}};
        for (int threads : {256, 512}) {
            printf("---- memory instructions behind 4 MFMAs, %d wave(s) per SIMD\n", threads / 256);
            for (auto& m : ms) {
                printf("%-28s ", m.name); fflush(stdout);
                (void)hipFuncSetAttribute((const void*)m.k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                hipLaunchKernelGGL(m.k, 256, threads, 65536, 0, out, cyc, 100, src);
                (void)hipDeviceSynchronize();
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(m.k, 256, threads, 65536, 0, out, cyc, iters, src);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms_ = 0; (void)hipEventElapsedTime(&ms_, e0, e1);
                printf("%6.2f ns per MFMA of one wave\n", ms_ * 1e6 / iters / 4.0); fflush(stdout);
            }
        }
    }
    return 0;
}
