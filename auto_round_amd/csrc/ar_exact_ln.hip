// ar_exact_ln.hip -- nn.LayerNorm forward / input-gradient WITH THE BITS OF TORCH'S OWN KERNELS ON THIS GPU (round 6), for the
// `exact_rounding` form of OPT-style decoder blocks (auto_round_amd/exact_opt_block.py; BASELINE configs[0], OPT-125M).
//
// What the reference runs (transformers/models/opt/modeling_opt.py `OPTDecoderLayer` under auto_round's `block_forward`,
// auto_round/compressors/utils.py:109-172, with autocast): `layer_norm` is on autocast's fp32 list, so each norm is
//     x.float()  ->  at::native::vectorized_layer_norm_kernel<float, float, false>  ->  the consuming linear's cast to bf16
// and autograd's backward   dy.float() -> at::native::layer_norm_grad_input_kernel<float, float, false> -> .to(bf16).
// The two kernels below restate those two ATen kernels -- read off the gfx950 code objects inside the installed libtorch_hip.so
// (torch 2.10.0+rocm7.0; clang-offload-bundler --unbundle, llvm-objdump), since the arithmetic that decides the bits is the
// COMPILED one: which products the compiler fused into an fma and which it left as mul + add is not visible in ATen's source.
//
// forward, one workgroup of 64 x 4 threads per row (ATen's launch: `threads(warp_size, num_threads() / warp_size)`, blocks(M)):
//   thread t = x + 64 y owns the float4 vectors t, t + 256, ... of the row and runs Welford's update over their elements in order:
//        count += 1 ; r = v_rcp_f32(count) ; delta = v - mean ; mean = fma(r, delta, mean) ; m2 = m2 + delta * (v - mean)
//        (the mean update IS an fma, the m2 update is NOT: the compiler packed the two products of neighbouring elements into a v_pk_mul);
//   64 lanes combine by shuffle-down 32, 16, ..., 1 (lane l takes lane l + off as B):
//        n = nA + nB ; c = v_rcp_f32(n) ; d = meanA - meanB ; wA = nA * c ; wB = c * nB
//        mean = fma(wB, meanB, wA * meanA) ; m2 = fma((d * d) * nB, wA, m2A + m2B)            (n == 0: all zero)
//   the four wavefronts combine through shared memory (upper half writes, lower half merges; offsets 2, 1), the same formulas except
//        mean = fma(meanA, wA, wB * meanB)                                                     (the OTHER product is the fused one here);
//   var = m2 / float(N) (IEEE division) ; rstd = v_rsq_f32(var + eps) ; y = fma(rstd * (v - mean), gamma, beta).
// backward (rows < 32768: above that ATen's ROCm build switches to cuComputeGradInput, which this file does not restate), one
//   workgroup of 256 threads per row: thread t accumulates over elements 4t .. 4t+3, 4t+1024 .. in order
//        g = gamma * dy ; s1 = s1 + g ; s2 = s2 + rstd * ((x - mean) * g)                      (plain adds: no fma)
//   both sums through cuda_utils::BlockReduceSum (shuffle-down 32 .. 1 inside a wavefront, lane 0 of each to shared memory, the first
//   wavefront reduces those the same way), then per element
//        dx = ((1 / float(N)) * rstd) * ((fma(dy, gamma * float(N), -(s2 * (rstd * (x - mean))))) - s1).
// The fp32 results are rounded to the activation dtype once (the cast the module path's next op performs); `dres` (the residual
// branch's gradient) is added as its own rounding, like autograd's accumulation.  `flags` keeps the two choices another torch build
// could make differently reachable (the caller proves a form against torch before using it): bit 0 = IEEE 1/x instead of v_rcp_f32
// (builds without USE_LAYERNORM_FAST_RECIPROCAL), bit 1 = rsqrt evaluated in double.
// This file is compiled with -ffp-contract=off like the rest of the library: every fma below is written out.
#include "ar_common.hpp"

namespace ar {

struct Welford {
    float mean, m2, count;
};

template <bool IEEE_RCP> __device__ __forceinline__ float ln_rcp(float x) {
    if constexpr (IEEE_RCP) return 1.f / x;
    else return __builtin_amdgcn_rcpf(x);
}

template <bool IEEE_RCP> __device__ __forceinline__ Welford welford_push(float v, Welford w) {
    const float delta = v - w.mean;
    const float n = w.count + 1.f;
    const float mean = fmaf(ln_rcp<IEEE_RCP>(n), delta, w.mean);
    return {mean, w.m2 + delta * (v - mean), n};
}

// wavefront stage (A = this lane, B = lane + offset)
template <bool IEEE_RCP> __device__ __forceinline__ Welford welford_merge_lanes(Welford a, Welford b) {
    const float n = a.count + b.count;
    if (!(n > 0.f)) return {0.f, 0.f, n};
    const float c = ln_rcp<IEEE_RCP>(n);
    const float d = a.mean - b.mean;
    const float wa = a.count * c;
    const float wb = c * b.count;
    return {fmaf(wb, b.mean, wa * a.mean), fmaf((d * d) * b.count, wa, a.m2 + b.m2), n};
}

// shared-memory stage (A = this wavefront's lane 0, B = the wavefront `offset` above)
template <bool IEEE_RCP> __device__ __forceinline__ Welford welford_merge_waves(Welford a, Welford b) {
    const float n = a.count + b.count;
    if (!(n > 0.f)) return {0.f, 0.f, n};
    const float c = ln_rcp<IEEE_RCP>(n);
    const float d = a.mean - b.mean;
    const float wb = b.count * c;
    const float wa = a.count * c;
    return {fmaf(a.mean, wa, wb * b.mean), fmaf(b.count * (d * d), wa, a.m2 + b.m2), n};
}

template <int DT> __device__ __forceinline__ void load4(const void* p, int64_t elem, float (&v)[4]) {
    const uint2 q = *reinterpret_cast<const uint2*>((const uint16_t*)p + elem);
    if constexpr (DT == AR_DT_BF16) {
        v[0] = bf16_lo(q.x); v[1] = bf16_hi(q.x); v[2] = bf16_lo(q.y); v[3] = bf16_hi(q.y);
    } else {
        v[0] = f16_to_f32(q.x & 0xffffu); v[1] = f16_to_f32(q.x >> 16); v[2] = f16_to_f32(q.y & 0xffffu); v[3] = f16_to_f32(q.y >> 16);
    }
}
template <int DT> __device__ __forceinline__ void store4(void* p, int64_t elem, const float (&v)[4]) {
    uint2 q;
    if constexpr (DT == AR_DT_BF16) {
        q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]);
    } else {
        q.x = f32_to_f16(v[0]) | (f32_to_f16(v[1]) << 16); q.y = f32_to_f16(v[2]) | (f32_to_f16(v[3]) << 16);
    }
    *reinterpret_cast<uint2*>((uint16_t*)p + elem) = q;
}

constexpr int kLnWaves = 4;      // ATen: num_threads() / warp_size on a 64-wide wavefront

template <int DT, bool IEEE_RCP, bool RSQRT_F64>
__global__ __launch_bounds__(kTPB) void k_x_layernorm_fwd(const void* __restrict__ x, const void* __restrict__ gamma, const void* __restrict__ beta,
                                                           void* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int hidden, float eps) {
    __shared__ float buf[kLnWaves * 3];          // [2 * wave] mean, m2 ; [2 * kLnWaves/2 ...]: ATen's layout is buf / buf + blockDim.y
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * hidden;
    const int nvec = hidden / 4;
    Welford w = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nvec; i += kTPB) {
        float v[4];
        load4<DT>(x, base + 4 * (int64_t)i, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) w = welford_push<IEEE_RCP>(v[j], w);
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const Welford b = {__shfl_down(w.mean, off, kWave), __shfl_down(w.m2, off, kWave), __shfl_down(w.count, off, kWave)};
        w = welford_merge_lanes<IEEE_RCP>(w, b);
    }
    float* ms = buf;                 // mean, m2 pairs
    float* cnt = buf + kLnWaves;     // counts
    for (int off = kLnWaves / 2; off > 0; off /= 2) {
        if (lane == 0 && wave >= off && wave < 2 * off) {
            const int s = wave - off;
            ms[2 * s] = w.mean; ms[2 * s + 1] = w.m2; cnt[s] = w.count;
        }
        __syncthreads();
        if (lane == 0 && wave < off) {
            const Welford b = {ms[2 * wave], ms[2 * wave + 1], cnt[wave]};
            w = welford_merge_waves<IEEE_RCP>(w, b);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ms[0] = w.mean;
        ms[1] = w.m2 / (float)hidden;
    }
    __syncthreads();
    const float mean = ms[0];
    const float var = ms[1] + eps;
    float rstd;
    if constexpr (RSQRT_F64) rstd = (float)rsqrt((double)var);
    else rstd = (var < 1.17549435e-38f) ? __builtin_amdgcn_rsqf(var * 16777216.f) * 4096.f : __builtin_amdgcn_rsqf(var);
    for (int i = threadIdx.x; i < nvec; i += kTPB) {
        float v[4], g[4], b[4], o[4];
        load4<DT>(x, base + 4 * (int64_t)i, v);
        load4<DT>(gamma, 4 * (int64_t)i, g);
        load4<DT>(beta, 4 * (int64_t)i, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(rstd * (v[j] - mean), g[j], b[j]);
        store4<DT>(y, base + 4 * (int64_t)i, o);
    }
    if (threadIdx.x == 0) {
        if (mean_out) mean_out[blockIdx.x] = mean;
        if (rstd_out) rstd_out[blockIdx.x] = rstd;
    }
}

// cuda_utils::BlockReduceSum over 256 threads: result valid in thread 0
__device__ __forceinline__ float aten_block_sum(float v, float* shared) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v = v + __shfl_down(v, off, kWave);
    __syncthreads();
    if (lane == 0) shared[wave] = v;
    __syncthreads();
    v = (threadIdx.x < kTPB / kWave) ? shared[lane] : 0.f;
    if (wave == 0) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) v = v + __shfl_down(v, off, kWave);
    }
    return v;
}

template <int DT, bool RES>
__global__ __launch_bounds__(kTPB) void k_x_layernorm_bwd(const void* __restrict__ dy, const void* __restrict__ x, const void* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const void* __restrict__ dres, void* __restrict__ dx, int hidden) {
    __shared__ float red[kTPB / kWave];
    __shared__ float st[2];
    const int64_t base = (int64_t)blockIdx.x * hidden;
    const float m = mean[blockIdx.x], r = rstd[blockIdx.x];
    float s1 = 0.f, s2 = 0.f;
    int l = 4 * (int)threadIdx.x;
    for (; l + 3 < hidden; l += 4 * kTPB) {
        float d[4], xv[4], g[4];
        load4<DT>(dy, base + l, d);
        load4<DT>(x, base + l, xv);
        load4<DT>(gamma, l, g);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gl = g[k] * d[k];
            s1 = s1 + gl;
            s2 = s2 + r * ((xv[k] - m) * gl);
        }
    }
    for (; l < hidden; ++l) {
        const float gl = load1<DT>(gamma, l) * load1<DT>(dy, base + l);
        s1 = s1 + gl;
        s2 = s2 + r * ((load1<DT>(x, base + l) - m) * gl);
    }
    s1 = aten_block_sum(s1, red);
    s2 = aten_block_sum(s2, red);
    if (threadIdx.x == 0) {
        st[0] = s1;
        st[1] = s2;
    }
    __syncthreads();
    s1 = st[0];
    s2 = st[1];
    const float fH = (float)hidden;
    const float term1 = r * (1.f / fH);
    for (int i = threadIdx.x; i < hidden; i += kTPB) {
        const float xv = load1<DT>(x, base + i), d = load1<DT>(dy, base + i), g = load1<DT>(gamma, i);
        float f = fmaf(d, g * fH, -(s2 * (r * (xv - m))));
        f = f - s1;
        float o = round_to<DT>(term1 * f);
        if (RES) o = o + load1<DT>(dres, base + i);
        store1<DT>(dx, base + i, o);
    }
}

}  // namespace ar

using namespace ar;

#define AR_DT_SWITCH2(dt, CALL)                    \
    switch (dt) {                                  \
        case AR_DT_BF16: CALL(AR_DT_BF16); break;  \
        case AR_DT_F16: CALL(AR_DT_F16); break;    \
        default: return AR_ERR_UNSUPPORTED;        \
    }

extern "C" int ar_layernorm_fwd_exact(const void* x, const void* gamma, const void* beta, void* y, float* mean_out, float* rstd_out,
                                      int64_t rows, int hidden, float eps, int flags, int dt, ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    // ATen takes its vectorised kernel when N % 4 == 0 and every pointer is 16-byte aligned in fp32 terms; anything else runs
    // RowwiseMomentsCUDAKernel + LayerNormForwardCUDAKernel, which this file does not restate
    if (!x || !gamma || !beta || !y || hidden < 4 || hidden % 4 || hidden > (1 << 24) || rows > 0x7fffffffLL) return AR_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 7) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)rows;
#define AR_CALL(DT)                                                                                                                         \
    switch (flags & 3) {                                                                                                                    \
        case 0: AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_layernorm_fwd<DT, false, false>), grid, kTPB, 0, st, x, gamma, beta, y, mean_out, rstd_out, hidden, eps); break; \
        case 1: AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_layernorm_fwd<DT, true, false>), grid, kTPB, 0, st, x, gamma, beta, y, mean_out, rstd_out, hidden, eps); break;  \
        case 2: AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_layernorm_fwd<DT, false, true>), grid, kTPB, 0, st, x, gamma, beta, y, mean_out, rstd_out, hidden, eps); break;  \
        default: AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_layernorm_fwd<DT, true, true>), grid, kTPB, 0, st, x, gamma, beta, y, mean_out, rstd_out, hidden, eps); break;  \
    }
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_layernorm_bwd_exact(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd, const void* dres,
                                      void* dx, int64_t rows, int hidden, int dt, ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    // rows >= 32768: ATen's ROCm build launches cuComputeGradInput instead of layer_norm_grad_input_kernel (another summation order)
    if (!dy || !x || !gamma || !mean || !rstd || !dx || hidden < 4 || hidden % 4 || rows >= 32768) return AR_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dy) & 7) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)rows;
#define AR_CALL(DT)                                                                                                                \
    if (dres) AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_layernorm_bwd<DT, true>), grid, kTPB, 0, st, dy, x, gamma, mean, rstd, dres, dx, hidden); \
    else AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_layernorm_bwd<DT, false>), grid, kTPB, 0, st, dy, x, gamma, mean, rstd, dres, dx, hidden)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}
