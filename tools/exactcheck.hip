// exactcheck.hip -- exhaustive GPU verification of the two "fast but exact" primitives of the streaming kernels
// (auto_round_amd/csrc/ar_common.hpp), against the plain IEEE / integer formulations the reference arithmetic implies.
//  (1) div_fast: y = 1/s ; q0 = w*y ; r = fma(-q0,s,w) ; q = fma(r,y,q0)  (guarded by div_fast_ok) vs  w/s,
//      for ALL finite bf16 and fp16 weights x ALL fp16 scales with |s| >= fp16(1e-5)  (2^32 pairs per dtype),
//      and for the backward's second quotient (w/s)/s;  plus 2^32 pseudo-random (fp32 weight, fp16 scale) pairs.
//  (2) pack_bf16x2 (v_cvt_pk_bf16_f32) vs the integer round-to-nearest-even formula, for ALL 2^32 fp32 inputs.
// Prints one JSON object per check; every "mismatch" field must be 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define AR_FASTDIV 1
#define AR_HW_BF16 1
#include "../auto_round_amd/csrc/ar_common.hpp"
using namespace ar;

__device__ __forceinline__ float guarded_div(float w, float s, float y) { return div_fast_ok(w) ? div_fast(w, s, y) : w / s; }

template <int WDT>
__global__ void check_div(unsigned long long* out) {
    const float s = f16_to_f32(blockIdx.x);
    const float as = fabsf(s);
    if (!(as >= 1.00135803e-05f) || !(as <= 65504.f)) return;
    const float y = 1.0f / s;
    unsigned long long b1 = 0, b2 = 0, tot = 0, fastn = 0;
    for (uint32_t wb = threadIdx.x; wb < 65536; wb += blockDim.x) {
        const float w = WDT == 0 ? __uint_as_float(wb << 16) : f16_to_f32(wb);
        if (!(fabsf(w) <= 3.4e38f)) continue;
        const float q = w / s;
        if (__float_as_uint(q) != __float_as_uint(guarded_div(w, s, y))) ++b1;
        // the kernels decide fast/slow on w only and then use the same mode for x = w/s
        const float q2 = q / s;
        const float f2 = div_fast_ok(w) ? div_fast(q, s, y) : q / s;
        if (__float_as_uint(q2) != __float_as_uint(f2)) ++b2;
        fastn += div_fast_ok(w) ? 1 : 0;
        ++tot;
    }
    atomicAdd(out + 0, b1); atomicAdd(out + 1, b2); atomicAdd(out + 2, tot); atomicAdd(out + 3, fastn);
}

__global__ void check_div_f32(unsigned long long* out) {
    // 2^32 pseudo-random pairs: fp32 weight bit patterns from a counter hash, fp16 scales from the low bits
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b1 = 0, tot = 0;
    for (int rep = 0; rep < 256; ++rep) {
        uint64_t h = (gid * 256 + rep) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const float w = __uint_as_float((uint32_t)h);
        const float s = f16_to_f32((uint32_t)(h >> 32) & 0xffffu);
        const float as = fabsf(s);
        if (!(as >= 1.00135803e-05f) || !(as <= 65504.f) || !(fabsf(w) <= 3.4e38f)) continue;
        const float y = 1.0f / s;
        if (__float_as_uint(w / s) != __float_as_uint(guarded_div(w, s, y))) ++b1;
        ++tot;
    }
    atomicAdd(out + 0, b1); atomicAdd(out + 2, tot);
}

__global__ void check_bf16(unsigned long long* out) {
    const uint32_t base = ((uint32_t)blockIdx.x * blockDim.x + threadIdx.x) * 256u;
    unsigned long long bad = 0;
    for (uint32_t i = 0; i < 256; i += 2) {
        const float a = __uint_as_float(base + i), b = __uint_as_float(base + i + 1);
        const uint32_t hw = pack_bf16x2(a, b);
        const uint32_t sw = f32_to_bf16(a) | (f32_to_bf16(b) << 16);
        // NaN payloads may differ (both are quiet NaNs): compare NaN-ness only
        const bool na = (__float_as_uint(a) & 0x7fffffffu) > 0x7f800000u, nb = (__float_as_uint(b) & 0x7fffffffu) > 0x7f800000u;
        const uint32_t m = (na ? 0u : 0xffffu) | (nb ? 0u : 0xffff0000u);
        if ((hw & m) != (sw & m)) ++bad;
        if (na && ((hw & 0x7fffu) <= 0x7f80u)) ++bad;
        if (nb && (((hw >> 16) & 0x7fffu) <= 0x7f80u)) ++bad;
    }
    atomicAdd(out, bad);
}

// (3) MXFP4 shared exponent: floorf(log2f(x)) as the fp4 kernels evaluate it (csrc/ar_fp4.hip fp4_group_scale) against the host
//     libm (glibc log2f -- what the C oracle and the CPU goldens use) for every float within +-8 ulp of every power of two
//     2^k, k in [-149, 127] (the only inputs where a 1-ulp difference between two libms can move the floor), and for 2^24
//     pseudo-random positive floats.
__global__ void eval_floor_log2(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = floorf(log2f(x[i]));
}

#include <math.h>
#include <vector>
static void check_log2() {
    std::vector<float> xs;
    for (int k = -149; k <= 127; ++k) {
        const float p = ldexpf(1.0f, k);
        uint32_t b; memcpy(&b, &p, 4);
        for (int d = -8; d <= 8; ++d) {
            const int64_t bb = (int64_t)b + d;
            if (bb <= 0 || bb >= 0x7f800000ll) continue;
            const uint32_t u = (uint32_t)bb; float f; memcpy(&f, &u, 4);
            xs.push_back(f);
        }
    }
    const size_t n_edge = xs.size();
    uint64_t h = 0x1234567ull;
    for (int i = 0; i < (1 << 24); ++i) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        uint32_t u = (uint32_t)(h >> 33) % 0x7f800000u;
        if (u == 0) u = 1;
        float f; memcpy(&f, &u, 4);
        xs.push_back(f);
    }
    float *dx, *dy;
    hipMalloc(&dx, xs.size() * 4); hipMalloc(&dy, xs.size() * 4);
    hipMemcpy(dx, xs.data(), xs.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(eval_floor_log2, (unsigned)((xs.size() + 255) / 256), 256, 0, 0, dx, dy, (int)xs.size());
    std::vector<float> ys(xs.size());
    hipMemcpy(ys.data(), dy, xs.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long bad_edge = 0, bad_rand = 0, bad_vs_exponent = 0, bad_edge_normal = 0, bad_rand_normal = 0, printed = 0;
    for (size_t i = 0; i < xs.size(); ++i) {
        const float host = floorf(log2f(xs[i]));
        const bool normal = xs[i] >= 1.17549435e-38f;      // subnormal inputs: ocml's log2f flushes them, glibc does not
        if (host != ys[i]) {
            if (normal && printed++ < 8) printf("{\"log2_mismatch_normal_input\": {\"x_bits\": %u, \"gpu\": %g, \"host\": %g}}\n", *(uint32_t*)&xs[i], ys[i], host);
            if (i < n_edge) { ++bad_edge; bad_edge_normal += normal; }
            else { ++bad_rand; bad_rand_normal += normal; }
        }
        int e; frexpf(xs[i], &e);
        if (i < n_edge && ys[i] != (float)(e - 1)) ++bad_vs_exponent;      // informational: where floor(log2f) != exponent field
    }
    printf("{\"check\": \"floorf(log2f(x)) GPU (ocml) vs host (glibc)\", \"edge_inputs\": %zu, \"edge_mismatch\": %llu, "
           "\"edge_mismatch_normal_inputs\": %llu, \"random_inputs\": %d, \"random_mismatch\": %llu, \"random_mismatch_normal_inputs\": %llu, "
           "\"edge_inputs_where_gpu_floor_differs_from_exponent_field\": %llu}\n",
           n_edge, bad_edge, bad_edge_normal, 1 << 24, bad_rand, bad_rand_normal, bad_vs_exponent);
    hipFree(dx); hipFree(dy);
}

int main(int argc, char** argv) {
    unsigned long long *d, h[4];
    check_log2();
    if (argc > 1) return 0;          // "exactcheck log2": only the (fast) shared-exponent check
    hipMalloc(&d, sizeof(h));
    for (int wdt = 0; wdt < 2; ++wdt) {
        hipMemset(d, 0, sizeof(h));
        if (wdt == 0) hipLaunchKernelGGL(check_div<0>, 65536, 256, 0, 0, d); else hipLaunchKernelGGL(check_div<1>, 65536, 256, 0, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("{\"check\": \"div_fast\", \"weights\": \"%s\", \"scales\": \"fp16\", \"pairs\": %llu, \"fast_path_pairs\": %llu, "
               "\"mismatch_w_over_s\": %llu, \"mismatch_w_over_s_over_s\": %llu}\n", wdt == 0 ? "bf16" : "fp16", h[2], h[3], h[0], h[1]);
    }
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(check_div_f32, 65536, 256, 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"check\": \"div_fast\", \"weights\": \"fp32 (random bit patterns)\", \"scales\": \"fp16\", \"pairs\": %llu, \"mismatch_w_over_s\": %llu}\n", h[2], h[0]);
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(check_bf16, 65536, 256, 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"check\": \"pack_bf16x2 (v_cvt_pk_bf16_f32) vs integer RNE\", \"inputs\": 4294967296, \"mismatch\": %llu}\n", h[0]);
    return 0;
}
