// Does v_mfma_f32_32x32x16_bf16 care WHICH of its 16 k-slots a product sits in?  (Needed to know whether a first-party attention
// kernel must reproduce the library's k-slot <-> row mapping, or only the set of rows per instruction and the instruction order.)
// Test: D = A B + C with random bf16 operands of wide dynamic range; then the same products with the k-slots permuted consistently in
// A and B (three permutations: lane halves swapped; elements reversed within a lane; the accumulator-adoption order
// k = 8 h + e  <->  row 8 (e >> 2) + 4 h + (e & 3)).  Prints the number of differing result words.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const uint16_t* A, const uint16_t* B, const float* C, float* D, const int* perm, int n) {
    // A: [n][32 rows][16 k], B: [n][16 k][32 cols], C/D: [n][32][32]; perm: 16 ints, slot -> source k
    const int lane = threadIdx.x, h = lane >> 5, l = lane & 31;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        uint16_t a[8], b[8];
        for (int e = 0; e < 8; ++e) {
            const int kk = perm[8 * h + e];
            a[e] = A[(t * 32 + l) * 16 + kk];
            b[e] = B[(t * 16 + kk) * 32 + l];
        }
        bf8 av, bv;
        __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
        f16v c;
        for (int r = 0; r < 16; ++r) c[r] = C[(t * 32 + (8 * (r >> 2) + 4 * h + (r & 3))) * 32 + l];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D[(t * 32 + (8 * (r >> 2) + 4 * h + (r & 3))) * 32 + l] = c[r];
    }
}
// the same products with the operand ROLES swapped: D2 = B^T A^T (a-operand rows = columns of B, b-operand columns = rows of A)
__global__ void k_swapped(const uint16_t* A, const uint16_t* B, const float* C, float* D, int n) {
    const int lane = threadIdx.x, h = lane >> 5, l = lane & 31;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        uint16_t a[8], b[8];
        for (int e = 0; e < 8; ++e) {
            const int kk = 8 * h + e;
            a[e] = B[(t * 16 + kk) * 32 + l];          // a-operand row l = column l of B
            b[e] = A[(t * 32 + l) * 16 + kk];          // b-operand column l = row l of A
        }
        bf8 av, bv;
        __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
        f16v c;       // D2[m = column of the original][n = row of the original]
        for (int r = 0; r < 16; ++r) c[r] = C[(t * 32 + l) * 32 + (8 * (r >> 2) + 4 * h + (r & 3))];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D[(t * 32 + l) * 32 + (8 * (r >> 2) + 4 * h + (r & 3))] = c[r];
    }
}
static uint16_t rnd_bf16(int spread) {
    float m = 1.0f + (rand() % 128) / 128.0f;
    int e = rand() % (2 * spread + 1) - spread;
    float v = ldexpf(m, e) * ((rand() & 1) ? -1.f : 1.f);
    uint32_t u; memcpy(&u, &v, 4); return (uint16_t)(u >> 16);
}
int main() {
    const int n = 4096;
    for (int spread : {2, 8, 20}) {
        std::vector<uint16_t> A(n * 32 * 16), B(n * 16 * 32); std::vector<float> C(n * 1024);
        srand(1 + spread);
        for (auto& x : A) x = rnd_bf16(spread);
        for (auto& x : B) x = rnd_bf16(spread);
        for (auto& x : C) { uint32_t u = (uint32_t)rnd_bf16(spread) << 16 | (rand() & 0xffff); memcpy(&x, &u, 4); }
        uint16_t *dA, *dB; float *dC, *dD; int* dP;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, C.size() * 4); hipMalloc(&dP, 64);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        std::vector<std::vector<int>> perms(4, std::vector<int>(16));
        for (int s = 0; s < 16; ++s) {
            perms[0][s] = s;
            perms[1][s] = s ^ 8;                                   // lane halves swapped
            perms[2][s] = (s & 8) | (7 - (s & 7));                 // reversed within a lane
            const int hh = s >> 3, e = s & 7;
            perms[3][s] = 8 * (e >> 2) + 4 * hh + (e & 3);         // accumulator-adoption order
        }
        std::vector<std::vector<float>> R(4, std::vector<float>(n * 1024));
        for (int p = 0; p < 4; ++p) {
            hipMemcpy(dP, perms[p].data(), 64, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, 256, 64, 0, 0, dA, dB, dC, dD, dP, n);
            hipMemcpy(R[p].data(), dD, C.size() * 4, hipMemcpyDeviceToHost);
        }
        for (int p = 1; p < 4; ++p) {
            long diff = 0;
            for (size_t i = 0; i < R[0].size(); ++i) diff += memcmp(&R[0][i], &R[p][i], 4) != 0;
            printf("spread 2^+-%d  perm %d: %ld of %zu result words differ\n", spread, p, diff, R[0].size());
        }
        {
            std::vector<float> R2(n * 1024);
            hipLaunchKernelGGL(k_swapped, 256, 64, 0, 0, dA, dB, dC, dD, n);
            hipMemcpy(R2.data(), dD, C.size() * 4, hipMemcpyDeviceToHost);
            long diff = 0;
            for (size_t i = 0; i < R2.size(); ++i) diff += memcmp(&R[0][i], &R2[i], 4) != 0;
            printf("spread 2^+-%d  operand roles swapped: %ld of %zu result words differ\n", spread, diff, R2.size());
        }
        // and: one K = 16 instruction against the exact fp64 sum rounded once (how the hardware rounds)
        long exact = 0;
        for (int t = 0; t < 64; ++t)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double acc = C[(t * 32 + i) * 32 + j];
                    for (int kk = 0; kk < 16; ++kk) {
                        uint32_t ua = (uint32_t)A[(t * 32 + i) * 16 + kk] << 16, ub = (uint32_t)B[(t * 16 + kk) * 32 + j] << 16;
                        float fa, fb; memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
                        acc += (double)fa * fb;
                    }
                    float want = (float)acc;
                    exact += memcmp(&want, &R[0][(t * 32 + i) * 32 + j], 4) == 0;
                }
        printf("spread 2^+-%d: %ld of %d results equal the fp64 sum rounded once\n", spread, exact, 64 * 1024);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD); hipFree(dP);
    }
    return 0;
}
