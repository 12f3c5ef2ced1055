// streamprobe.hip -- what HBM bandwidth does this chip give to the TRAFFIC PATTERN of the quant kernels with trivial
// arithmetic?  (a) float4 copy; (b) K1 pattern: read 2 B + 4 B, write 2 B per element; (c) K2 pattern: read 2+2+4,
// write 4.  Grid-stride, 16 B per lane per access, several unroll depths / grid sizes.  Practical ceilings for roofline.frac.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ o, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        uint4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n) r[u] = a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n) o[i + u * 256] = r[u];
    }
}
// K1 pattern: chunk = 8 elements: W 16 B, V 32 B -> Wq 16 B
template <int U>
__global__ __launch_bounds__(256) void k_p1(const uint4* __restrict__ W, const float4* __restrict__ V, uint4* __restrict__ O, int64_t nchunks) {
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t c = (int64_t)blockIdx.x * 256 * U + threadIdx.x; c < nchunks; c += stride) {
        uint4 w[U]; float4 v0[U], v1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (c + u * 256 < nchunks) { w[u] = W[c + u * 256]; v0[u] = V[2 * (c + u * 256)]; v1[u] = V[2 * (c + u * 256) + 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (c + u * 256 < nchunks) {
            uint4 o = w[u];
            o.x ^= __float_as_uint(v0[u].x + v1[u].x); o.y ^= __float_as_uint(v0[u].y + v1[u].y);
            o.z ^= __float_as_uint(v0[u].z + v1[u].z); o.w ^= __float_as_uint(v0[u].w + v1[u].w);
            O[c + u * 256] = o;
        }
    }
}
// K2 pattern: dWq 16 B, W 16 B, V 32 B -> V 32 B
template <int U>
__global__ __launch_bounds__(256) void k_p2(const uint4* __restrict__ G, const uint4* __restrict__ W, float4* __restrict__ V, int64_t nchunks) {
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t c = (int64_t)blockIdx.x * 256 * U + threadIdx.x; c < nchunks; c += stride) {
        uint4 g[U], w[U]; float4 v0[U], v1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (c + u * 256 < nchunks) { g[u] = G[c + u * 256]; w[u] = W[c + u * 256]; v0[u] = V[2 * (c + u * 256)]; v1[u] = V[2 * (c + u * 256) + 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (c + u * 256 < nchunks) {
            v0[u].x += __uint_as_float((g[u].x ^ w[u].x) & 0x3f800000u); v1[u].y += __uint_as_float((g[u].y ^ w[u].w) & 0x3f800000u);
            V[2 * (c + u * 256)] = v0[u]; V[2 * (c + u * 256) + 1] = v1[u];
        }
    }
}

template <typename F> float timeit(F f, int iters = 20) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters;
}

int main() {
    const int64_t n = 218103808;           // Llama-3-8B block
    const int64_t nchunks = n / 8;
    void *W, *G, *V, *O;
    hipMalloc(&W, n * 2); hipMalloc(&G, n * 2); hipMalloc(&O, n * 2); hipMalloc(&V, n * 4);
    hipMemset(W, 1, n * 2); hipMemset(G, 2, n * 2); hipMemset(V, 0, n * 4);
    int grids[] = {2048, 4096, 8192, 0};
    for (int gi = 0; gi < 4; ++gi) {
        for (int U = 1; U <= 4; U *= 2) {
            int64_t full = (nchunks + 256 * U - 1) / (256 * U);
            int grid = grids[gi] ? grids[gi] : (int)full;
            float t0, t1, t2;
            auto run = [&](auto kc, auto k1, auto k2) {
                t0 = timeit([&] { hipLaunchKernelGGL(kc, grid, 256, 0, 0, (const uint4*)V, (uint4*)V + n / 8, n / 8); });   // copy half of V onto the other half: n*2 B read + n*2 B written
                t1 = timeit([&] { hipLaunchKernelGGL(k1, grid, 256, 0, 0, (const uint4*)W, (const float4*)V, (uint4*)O, nchunks); });
                t2 = timeit([&] { hipLaunchKernelGGL(k2, grid, 256, 0, 0, (const uint4*)G, (const uint4*)W, (float4*)V, nchunks); });
            };
            if (U == 1) run(k_copy<1>, k_p1<1>, k_p2<1>); else if (U == 2) run(k_copy<2>, k_p1<2>, k_p2<2>); else run(k_copy<4>, k_p1<4>, k_p2<4>);
            printf("{\"grid\": %d, \"unroll\": %d, \"copy_GBps\": %.0f, \"k1_pattern_GBps\": %.0f, \"k2_pattern_GBps\": %.0f}\n", grid, U,
                   4.0 * n / t0 / 1e6, 8.0 * n / t1 / 1e6, 12.0 * n / t2 / 1e6);
        }
    }
    return 0;
}
