// mfma_power.hip -- where the power budget of a bf16 GEMM inner loop goes on THIS chip (round 5).  tools/mfma_peak.hip showed that a
// bare v_mfma_f32_32x32x16_bf16 stream clocks down from 2.36 GHz (zeros) to 1.82 GHz (random operands); the GEMM kernels' cycle trace
// (profiles/r05_gemm_nt_phase_cycles.json) shows 85 % MFMA issue at 1.55 GHz, while hipBLASLt's MT256x256x64 MI16x16x1 kernel runs
// 15-20 % faster.  This tool builds the inner loop up one ingredient at a time -- no HBM-sized problem, no barriers, every wave on
// its own -- and reports TFLOP/s and the effective shader clock of each arm, all on uniform random bf16 data, 2 waves per SIMD:
//   m32_const   32x32x16, one operand pair reused by every MFMA (mfma_peak's arm: operand buses do not toggle)
//   m32_rot     32x32x16, 4 a-fragments x 2 b-fragments in registers (the NT kernel's 128 x 64 wave tile), every MFMA another pair
//   m16_const / m16_rot   v_mfma_f32_16x16x32_bf16 (what the library's kernel is built on), 8 a x 4 b fragments
//   m32_lds     m32_rot + the fragments re-read from LDS every k-step (6 ds_read_b128 per 8 MFMAs, the NT kernel's ratio)
//   m16_lds     m16_rot + 12 ds_read_b128 per 32 MFMAs (same bytes per flop)
//   m32_lds_dma m32_lds + 8 LDS-DMA pieces (global_load_lds, 1 KiB each) per 32 MFMAs per wave (a 256 x 256 x 64 tile step)
//   m32_lds_vgpr m32_lds + the same bytes through global_load_dwordx4 -> ds_write_b128
//   m16_lds_dma   likewise on 16x16x32 (the VGPR-path arm does not fit 256 registers beside 32 accumulator tiles + double-buffered fragments)
//   *_big_*      one wave per SIMD with 128 x 128 wave tiles (512 registers): half the fragment reads per flop
// The load arms run twice: loads wrapping inside 2 MiB (L2 hits) and inside 64 MiB (no reuse).
// Reading the output: `tflops` is the figure of merit.  `ticks_per_us` (s_memtime ticks of wave 0 per microsecond of kernel time) is the
// shader clock only in the one-wave-per-SIMD (big) arms: with two waves per SIMD the older wave of a bare MFMA stream gets every issue
// slot and finishes first, so wave 0's ticks cover about half the kernel.
// Prints JSON lines.   hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/bin/mfma_power
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define PIN() __builtin_amdgcn_sched_barrier(0)
#define LDS_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(ADDR), "n"(OFF) : "memory")

constexpr int kFragBytes = 65536;      // LDS region the fragments are read from (random data)
constexpr int kDmaBytes = 65536;       // LDS region the loads land in (8 waves x 8 KiB)

// MODE: 0 registers only, 1 + LDS fragment reads, 2 + LDS-DMA, 3 + loads through VGPRs.  ROT: operands rotate.
template <bool M16, bool ROT, int MODE, bool BIG = false>
__global__ __launch_bounds__(BIG ? 256 : 512, 1) void k_power(const uint4* __restrict__ ops, size_t nops_mask, float* __restrict__ out,
                                                 uint64_t* __restrict__ ticks, int iters, size_t win_mask) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int THREADS = BIG ? 256 : 512;
    const size_t gid = (size_t)blockIdx.x * THREADS + tid;
    // fill the fragment region with random data
    for (int i = tid; i < kFragBytes / 16; i += THREADS) reinterpret_cast<uint4*>(lds)[i] = ops[(gid * 131 + (size_t)i * 7919) & nops_mask];
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    constexpr int NA = M16 ? 8 : 4, NB = (M16 ? 4 : 2) * (BIG ? 2 : 1);      // fragments of a 128 x 64 (BIG: 128 x 128, one wave per SIMD) wave tile per k-step
    constexpr int KSTEPS = M16 ? 2 : 4;                    // k-steps per 64 of K
    u32x4 fa[2][NA], fb[2][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) fa[0][i] = fa[1][i] = __builtin_bit_cast(u32x4, ops[(gid * 3 + i * 1000003) & nops_mask]);
#pragma unroll
    for (int i = 0; i < NB; ++i) fb[0][i] = fb[1][i] = __builtin_bit_cast(u32x4, ops[(gid * 5 + i * 2000003 + 99) & nops_mask]);
    f32x16 acc32[M16 ? 1 : NA * NB];
    f32x4 acc16[M16 ? NA * NB : 1];
#pragma unroll
    for (int i = 0; i < (M16 ? 1 : NA * NB); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < (M16 ? NA * NB : 1); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
    constexpr int NPIECE = BIG ? 16 : 8, NP2 = NPIECE / 2;      // 1 KiB pieces per wave per 64 of K (a 256 x 256 x 64 tile step / waves)
    const uint8_t* gsrc = reinterpret_cast<const uint8_t*>(ops);
    const size_t gmask = win_mask;      // the loads wrap inside this many bytes (a window inside L2, or all 64 MiB)
    size_t goff = ((size_t)(blockIdx.x * 8 + wave) * (NPIECE * 1024) + lane * 16) & gmask;
    uint4 stage[NPIECE];
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) stage[j] = make_uint4(0, 0, 0, 0);

    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const uint32_t rbase = lds0 + lane * 16 + ((it & 1) << 15);        // another 32 KiB half every iteration
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) {
                const uint32_t dst = lds0 + kFragBytes + wave * 8192 + (j & 7) * 1024;      // wave-uniform
                __builtin_amdgcn_global_load_lds((const void*)(gsrc + ((goff + j * 1024) & gmask)),
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
            }
        }
        if (MODE == 3) {      // two halves of 4 pieces: a half is loaded here / mid-iteration and written half an iteration later (registers)
#pragma unroll
            for (int j = 0; j < NP2; ++j) {
                *reinterpret_cast<uint4*>(lds + kFragBytes + wave * 8192 + ((NP2 + j) & 7) * 1024 + lane * 16) = stage[NP2 + j];
                stage[j] = *reinterpret_cast<const uint4*>(gsrc + ((goff + j * 1024) & gmask));
            }
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (MODE == 3 && ks == KSTEPS / 2) {
#pragma unroll
                for (int j = 0; j < NP2; ++j) {
                    *reinterpret_cast<uint4*>(lds + kFragBytes + wave * 8192 + (j & 7) * 1024 + lane * 16) = stage[j];
                    stage[NP2 + j] = *reinterpret_cast<const uint4*>(gsrc + ((goff + (NP2 + j) * 1024) & gmask));
                }
            }
            if (MODE >= 1) {                                  // next k-step's fragments in flight under this step's MFMAs
#pragma unroll
                for (int i = 0; i < NA; ++i) LDS_READ(fa[nxt][i], rbase, (ks * (NA + NB) + i) * 1024);
#pragma unroll
                for (int i = 0; i < NB; ++i) LDS_READ(fb[nxt][i], rbase, (ks * (NA + NB) + NA + i) * 1024);
                PIN();
            }
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, fa[ROT ? cur : 0][ROT ? i : 0]);
                    const bf16x8 b = __builtin_bit_cast(bf16x8, fb[ROT ? cur : 0][ROT ? j : 0]);
                    if constexpr (M16) acc16[j * NA + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[j * NA + i], 0, 0, 0);
                    else acc32[j * NA + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[j * NA + i], 0, 0, 0);
                }
            if (MODE >= 1) {
                PIN();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PIN();
            }
        }
        goff = (goff + 2048 * 8192) & gmask;      // (256 CUs x 64 KiB per step)
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < (M16 ? 1 : NA * NB); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
#pragma unroll
    for (int i = 0; i < (M16 ? NA * NB : 1); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc16[i][r];
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) s += (float)((stage[j].x ^ stage[j].y ^ stage[j].z ^ stage[j].w) & 1);
    out[gid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

typedef void (*kfn)(const uint4*, size_t, float*, uint64_t*, int, size_t);
struct ArmDef { const char* name; kfn fn; bool big; bool loads; };

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t nops = 1 << 22;                  // 64 MiB of operand data
    uint4* dops; (void)hipMalloc(&dops, nops * 16);
    uint16_t* h = (uint16_t*)malloc(nops * 16);
    srand(123);
    for (size_t i = 0; i < nops * 8; ++i) h[i] = f2bf((float)rand() / RAND_MAX * 2.f - 1.f);
    (void)hipMemcpy(dops, h, nops * 16, hipMemcpyHostToDevice);
    float* dout; (void)hipMalloc(&dout, (size_t)cus * 512 * 4);
    uint64_t* dticks; (void)hipMalloc(&dticks, cus * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const ArmDef arms[] = {
        {"m32_const", k_power<false, false, 0>, false, false},       {"m32_rot", k_power<false, true, 0>, false, false},
        {"m16_const", k_power<true, false, 0>, false, false},        {"m16_rot", k_power<true, true, 0>, false, false},
        {"m32_lds", k_power<false, true, 1>, false, false},          {"m16_lds", k_power<true, true, 1>, false, false},
        {"m32_lds_dma", k_power<false, true, 2>, false, true},       {"m32_lds_vgpr", k_power<false, true, 3>, false, true},
        {"m16_lds_dma", k_power<true, true, 2>, false, true},
        // one wave per SIMD, 128 x 128 wave tiles (512 registers): half the fragment reads per flop
        {"m32_big_rot", k_power<false, true, 0, true>, true, false}, {"m16_big_rot", k_power<true, true, 0, true>, true, false},
        {"m32_big_lds", k_power<false, true, 1, true>, true, false}, {"m16_big_lds", k_power<true, true, 1, true>, true, false},
        {"m32_big_lds_dma", k_power<false, true, 2, true>, true, true},   {"m16_big_lds_dma", k_power<true, true, 2, true>, true, true},
    };
    const int n_arms = sizeof(arms) / sizeof(arms[0]);
    const size_t shmem = kFragBytes + kDmaBytes;
    for (int a = 0; a < n_arms; ++a) (void)hipFuncSetAttribute((const void*)arms[a].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d}\n", prop.gcnArchName, cus, iters);
    // the load arms twice: the loads wrap inside 2 MiB (every XCD's L2 holds it: the hit-dominated case of a tiled GEMM) and inside
    // the whole 64 MiB (no reuse: every byte from the Infinity Cache / HBM)
    const size_t windows[2] = {(size_t)2 << 20, (size_t)64 << 20};
    for (int a = 0; a < n_arms; ++a) {
        for (int w = 0; w < (arms[a].loads ? 2 : 1); ++w) {
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(arms[a].fn, cus, arms[a].big ? 256 : 512, shmem, 0, dops, nops - 1, dout, dticks, iters, windows[w] - 1);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                if (hipGetLastError() != hipSuccess) { printf("{\"arm\": \"%s\", \"error\": true}\n", arms[a].name); break; }
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                uint64_t tk[4]; (void)hipMemcpy(tk, dticks, sizeof(tk), hipMemcpyDeviceToHost);
                const double flop = 2.0 * 256 * 256 * 64 * (double)iters * cus;      // one 256 x 256 x 64 tile step per CU per iteration
                const double chip_cycles = flop / (double)iters / cus / (4 * 1024.0);   // MFMA-pipe cycles of that step at full issue
                // s_memtime ticks of wave 0 per iteration vs the pipe cycles the iteration needs: issue efficiency if the tick is the
                // shader clock (printed raw: this part counts one tick per two shader cycles in some modes)
                printf("{\"arm\": \"%s\", \"load_window_mib\": %d, \"rep\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"ticks_per_iter\": %.1f, "
                       "\"pipe_cycles_per_iter\": %.0f, \"ticks_per_us\": %.0f}\n",
                       arms[a].name, arms[a].loads ? (int)(windows[w] >> 20) : 0, rep, ms, flop / (ms * 1e-3) / 1e12, (double)tk[0] / iters, chip_cycles,
                       (double)tk[0] / (ms * 1e3));
                fflush(stdout);
            }
        }
    }
    return 0;
}
