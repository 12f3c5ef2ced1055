// mfma_probe.hip -- pins on the GPU the hardware contracts the MFMA GEMM (auto_round_amd/csrc/ar_gemm.hip) relies on:
//  (1) ds_read_b64_tr_b16: LDS holds u16 value i at index i; lane L reads from byte address 8*L; the 4 values each lane gets
//      back are printed (so the lane -> source mapping of the hardware 4x4 transposing read can be read off);
//  (2) global_load_lds 16 B: destination = wave-uniform M0 base + 16 * lane;
//  (3) v_mfma_f32_32x32x16_bf16: A row = lane & 31, B col = lane & 31, k-slots pair up identically in A and B,
//      D col = lane & 31, D row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)  -- checked against a host product.
// Prints JSON lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_tr(uint16_t* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

__global__ void k_glds(const uint4* g, uint4* out) {
    __shared__ uint4 lds[256];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_amdgcn_global_load_lds(g + threadIdx.x, (__attribute__((address_space(3))) void*)(lds + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}

__global__ void k_mfma(const uint16_t* A, const uint16_t* B, float* D) {   // A [32][16], B [16][32] row-major bf16 bits; D [32][32]
    const int l = threadIdx.x;
    bf16x8 a, b;
    uint16_t ta[8], tb[8];
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (l >> 5) + j;
        ta[j] = A[(l & 31) * 16 + k];
        tb[j] = B[k * 32 + (l & 31)];
    }
    memcpy(&a, ta, 16); memcpy(&b, tb, 16);
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k_tr, 1, 64, 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"probe\": \"ds_read_b64_tr_b16, lane L reads lds[4L..4L+3] (u16 index values)\", \"lane_values\": [");
    for (int L = 0; L < 64; ++L) printf("%s[%d,%d,%d,%d]", L ? "," : "", h[4 * L], h[4 * L + 1], h[4 * L + 2], h[4 * L + 3]);
    printf("]}\n");
    // which rule?  S2: R[4b+r][c] = M[4b+c][r]  (value index = 4*(4b+c)+r) ; S1: R[L][j] = M[4j + L/4][L%4]
    int s1 = 1, s2 = 1;
    for (int grp = 0; grp < 4; ++grp)
        for (int L = 0; L < 16; ++L)
            for (int j = 0; j < 4; ++j) {
                const int got = h[4 * (16 * grp + L) + j] - 64 * grp;
                if (got != 4 * (4 * j + L / 4) + (L % 4)) s1 = 0;
                if (got != 4 * (4 * (L / 4) + j) + (L % 4)) s2 = 0;
            }
    printf("{\"probe\": \"ds_read_b64_tr_b16 rule\", \"rule1_row_is_lane_div4\": %d, \"rule2_row_is_lane_mod4\": %d}\n", s1, s2);

    uint4 *g, *o; hipMalloc(&g, 256 * 16); hipMalloc(&o, 256 * 16);
    uint32_t hg[1024]; for (int i = 0; i < 1024; ++i) hg[i] = 0xA0000000u + i;
    hipMemcpy(g, hg, sizeof(hg), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_glds, 1, 256, 0, 0, g, o);
    uint32_t ho[1024]; hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += ho[i] != hg[i];
    printf("{\"probe\": \"global_load_lds dwordx4: dst = M0 base + 16*lane\", \"mismatched_dwords\": %d}\n", bad);

    uint16_t hA[512], hB[512]; float hD[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = f2bf((float)(rand() % 17 - 8)); hB[i] = f2bf((float)(rand() % 13 - 6)); }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += bf2f(hA[m * 16 + k]) * bf2f(hB[k * 32 + n]); ref[m * 32 + n] = s; }
    uint16_t *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, 1, 64, 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    bad = 0; for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("{\"probe\": \"v_mfma_f32_32x32x16_bf16 operand / result layout\", \"mismatched_outputs\": %d}\n", bad);
    return 0;
}
