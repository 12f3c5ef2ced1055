// mfma_peak.hip -- what the matrix pipe of THIS chip sustains for v_mfma_f32_32x32x16_bf16, bare: no LDS, no barriers, no loads in
// the loop.  Round 5, VERDICT r04 weak #6: DESIGN called "1.70-1.78 PF" the chip's ceiling; that figure was k_gemm_dw4's own skeleton
// with its loads ablated.  Here: W waves per SIMD (1 or 2), each issuing 8 independent accumulator chains back to back, operands in
// registers -- (a) zero operands, (b) uniform random bf16 operands in [-1, 1), (c) random with the sign bit cleared.  Reported per
// arm: TFLOP/s from hipEvents and the effective shader clock (s_memtime ticks of one wave / its wall time): the guide's 2.38-2.5 PF
// is the pipe at 2.4 GHz; on random operands the chip clocks down to its power budget (MI355X_MICROARCH.md "DVFS give-back").
// Prints JSON lines.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_peak(const uint4* __restrict__ ops, float* __restrict__ out, uint64_t* __restrict__ ticks, int iters) {
    const int tid = threadIdx.x;
    // every lane its own 16 bytes of operand data (random arm: different bits per lane and per wave)
    const uint4 ua = ops[(blockIdx.x * THREADS + tid) & 0xFFFFF];
    const uint4 ub = ops[((blockIdx.x * THREADS + tid) + 77777) & 0xFFFFF];
    bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const uint64_t t0 = __builtin_readcyclecounter();       // s_memtime
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t nops = 1 << 20;
    uint4* dops; hipMalloc(&dops, nops * 16);
    uint16_t* h = (uint16_t*)malloc(nops * 16);
    float* dout; hipMalloc(&dout, (size_t)cus * 4 * 512 * 4);
    uint64_t* dticks; hipMalloc(&dticks, cus * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz_prop\": %d}\n", prop.name, cus, prop.clockRate);
    const char* fills[3] = {"zeros", "uniform_random_pm1", "uniform_random_sign_cleared"};
    for (int fill = 0; fill < 3; ++fill) {
        srand(123);
        for (size_t i = 0; i < nops * 8; ++i) {
            float v = fill == 0 ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
            if (fill == 2) v = v < 0 ? -v : v;
            h[i] = f2bf(v);
        }
        hipMemcpy(dops, h, nops * 16, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {
            const int threads = 256 * wps, grid = cus;      // one workgroup per CU: wps waves on each SIMD
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                if (wps == 1) hipLaunchKernelGGL(k_peak<256>, grid, threads, 0, 0, dops, dout, dticks, iters);
                else hipLaunchKernelGGL(k_peak<512>, grid, threads, 0, 0, dops, dout, dticks, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                uint64_t tk[8]; hipMemcpy(tk, dticks, sizeof(tk), hipMemcpyDeviceToHost);
                const double flop = 2.0 * 32 * 32 * 16 * 32.0 * iters * (double)grid * (threads / 64);
                // s_memtime runs at a fixed 100 MHz reference on some parts and at the shader clock on others: print the raw ratio too
                printf("{\"fill\": \"%s\", \"waves_per_simd\": %d, \"rep\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"ticks_wave0\": %llu, "
                       "\"ticks_per_mfma\": %.2f, \"ticks_per_us\": %.1f}\n",
                       fills[fill], wps, rep, ms, flop / (ms * 1e-3) / 1e12, (unsigned long long)tk[0], (double)tk[0] / (32.0 * iters),
                       (double)tk[0] / (ms * 1e3));
            }
        }
    }
    return 0;
}
