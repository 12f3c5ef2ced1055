// ar_fp4.hip -- MXFP4 (e8m0 shared exponent, group 32) and NVFP4 (e4m3 block scale x fp32 global scale, group 16)
// fake-quant forward and the FP4 nibble packer for gfx950.  OCP e4m3fn semantics (gfx950 is OCP, not fnuz).
//
// A group is gs/8 consecutive lanes (4 for MXFP4, 2 for NVFP4); the group absmax either comes precomputed
// (weights: constant during tuning, ar_group_absmax) or is reduced in-wave with a shuffle butterfly (activations).
#include "ar_common.hpp"

namespace ar {

// literal restatement of mxfp.quant_element(ebits=2, mbits=3, max_norm=6, "even") on |t| <= 6
// (auto_round/data_type/mxfp.py:49-85).  floor(log2|t|) clipped at 0 is 0/1/2 by comparison (exact in this range).
__device__ __forceinline__ float mx_e2m1(float t) {
    const float a0 = fabsf(t);
    const float pscale = (a0 >= 4.f) ? 4.f : ((a0 >= 2.f) ? 2.f : 1.f);   // 2^private_exp
    const float x = t / pscale * 2.0f;
    const float a = fabsf(x);
    const float h = a - 0.5f;
    const float mask = (h == 2.0f * floorf(h * 0.5f)) ? 1.f : 0.f;        // (a - 0.5) % 2 == 0
    float v = sgnf(x) * (floorf(a + 0.5f) - mask);
    v = v / 2.0f * pscale;
    return clamp3(v, -6.f, 6.f);
}
// literal restatement of nvfp.cast_to_fp4 (auto_round/data_type/nvfp.py:26-39)
__device__ __forceinline__ float nv_e2m1(float x) {
    const float a = fabsf(x);
    float o;
    if (a < 2.0f) o = __builtin_rintf(2.0f * a) / 2.0f;
    else if (a < 4.0f) o = __builtin_rintf(a);
    else o = 2.0f * __builtin_rintf(a / 2.0f);
    o = fminf(o, 6.f);
    return o * sgnf(x);
}
// OCP e4m3fn encode (RNE; input pre-clamped to +-448) and decode
__device__ __forceinline__ uint32_t f32_to_e4m3(float f) {
    const uint32_t s = (__float_as_uint(f) >> 24) & 0x80u;
    const float a = fabsf(f);
    if (!(a < 464.0f)) return s | 0x7fu;
    if (a < 0.015625f) return s | (uint32_t)__builtin_rintf(a * 512.0f);
    int ex;
    const float fr = frexpf(a, &ex);
    int e = ex - 1;
    float m = __builtin_rintf((fr * 2.0f - 1.0f) * 8.0f);
    if (m == 8.0f) { m = 0.f; e += 1; }
    const int be = e + 7;
    if (be > 15 || (be == 15 && m == 7.0f)) return s | 0x7fu;
    return s | ((uint32_t)be << 3) | (uint32_t)m;
}
__device__ __forceinline__ float e4m3_to_f32(uint32_t b) {
    const uint32_t e = (b >> 3) & 0xfu, m = b & 7u;
    float v;
    if (e == 0) v = (float)m * 0.001953125f;
    else if (e == 15 && m == 7) v = __uint_as_float(0x7fc00000u);
    else v = ldexpf(1.0f + (float)m * 0.125f, (int)e - 7);
    return (b & 0x80u) ? -v : v;
}
__device__ __forceinline__ float recip0(float x) { return x == 0.f ? 0.f : 1.0f / x; }

struct Fp4Args {
    const void* X; const float* V; const float* absmax; const float* max_s; const float* gscale;
    void* Xq; void* scale_out;
    int64_t n_groups;
    int cpg, mode;
    float init_scale, lo, hi;
};

// per-group scale pair: (sc, aux).  MX: sc = 2^e, aux = e.  NV: sc = out_scale (multiply), aux = e4m3 scale value.
__device__ __forceinline__ void fp4_group_scale(int mode, float amax, float Ms, float init_scale, float gscale, float& sc,
                                                float& aux, float& rsc) {
    if (mode == 0) {
        const float mv = amax * (init_scale * Ms);
        float se = (mv == 0.f) ? 1.0f : log2f(mv);
        se = clamp3(floorf(se) - 2.0f, -127.f, 127.f);
        sc = ldexpf(1.0f, (int)se);
        aux = se;
        rsc = sc;
    } else {
        const float vm = amax * (Ms * init_scale);
        float s = gscale * (vm * (float)(1.0 / 6.0));
        s = clamp3(s, -448.f, 448.f);
        s = e4m3_to_f32(f32_to_e4m3(s));
        sc = recip0(s * recip0(gscale));   // out_scale
        rsc = recip0(sc);
        aux = s;
    }
}

template <int XDT>
__global__ __launch_bounds__(kTPB) void k_fp4_fwd(const Fp4Args a) {
    const int cpg = a.cpg;
    const int64_t total_chunks = a.n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    const float gscale = (a.mode == 1 && a.gscale) ? *a.gscale : 1.0f;
    // total_chunks is padded up to a multiple of the wave so that every lane of a lane-group joins the butterfly
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    for (int64_t c = (int64_t)blockIdx.x * kTPB + threadIdx.x; c < limit; c += stride) {
        const bool ok = c < total_chunks;
        const int64_t g = c / cpg;
        float x[8], v[8], o[8];
        float amax = 0.f;
        if (ok) {
            unpack8<XDT>(load8_raw<XDT>(a.X, c * kEPT), x);
            if (a.V) unpack_f8(load8_f32(a.V, c * kEPT), v);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            if (a.absmax) amax = a.absmax[g];
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(x[k]));
            }
        }
        if (!a.absmax) amax = lanes_max(amax, cpg);
        if (!ok) continue;
        const float Ms = a.max_s ? clamp3(a.max_s[g], a.lo, a.hi) : 1.0f;
        float sc, aux, rsc;
        fp4_group_scale(a.mode, amax, Ms, a.init_scale, gscale, sc, aux, rsc);
        if (a.mode == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t = clamp3(x[k] / sc + v[k], -6.f, 6.f);
                o[k] = mx_e2m1(t) * sc;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t = clamp3(x[k] * sc + v[k], -6.f, 6.f);
                o[k] = nv_e2m1(t) * rsc;
            }
        }
        store8<XDT>(a.Xq, c * kEPT, o);
        if (a.scale_out && (c % cpg) == 0) {
            if (a.mode == 0) store1<XDT>(a.scale_out, g, aux);
            else ((float*)a.scale_out)[g] = aux;
        }
    }
}

// FP4 nibble packer: one lane packs 8 consecutive values -> 4 bytes (one dword store).
__device__ __forceinline__ uint32_t fp4_nibble_f32(float x) {
    // argmin_j | |x| - tab[j] | with the first minimum winning (qlinear_fp.py:246-252), fp32 table arithmetic
    const float a = fabsf(x);
    const float tab[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    int best = 0;
    float bd = INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float d = fabsf(a - tab[j]);
        if (d < bd) { bd = d; best = j; }
    }
    return (uint32_t)best | ((__float_as_uint(x) >> 28) & 8u);
}
template <int WDT> __device__ __forceinline__ uint32_t fp4_nibble_lp(float x) {
    // same argmin but with the subtraction evaluated in the (16-bit) weight dtype, as the reference does for MXFP4
    const float a = fabsf(x);
    const float tab[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    int best = 0;
    float bd = INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float d = fabsf(round_to<WDT>(a - tab[j]));
        if (d < bd) { bd = d; best = j; }
    }
    return (uint32_t)best | ((__float_as_uint(x) >> 28) & 8u);
}

template <int WDT>
__global__ __launch_bounds__(kTPB) void k_pack_fp4(const void* __restrict__ W, const void* __restrict__ scale,
                                                   const float* __restrict__ gscale_dev, int64_t n_elems, int gs, int mode,
                                                   uint32_t* __restrict__ packed, uint8_t* __restrict__ scale_bytes) {
    const int64_t n_chunks = n_elems / kEPT;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    const float gscale = (mode == 1 && gscale_dev) ? *gscale_dev : 1.0f;
    for (int64_t c = (int64_t)blockIdx.x * kTPB + threadIdx.x; c < n_chunks; c += stride) {
        float w[8];
        unpack8<WDT>(load8_raw<WDT>(W, c * kEPT), w);
        const int64_t g = (c * kEPT) / gs;   // in % gs == 0 -> flat group index == o*n_groups + ig
        uint32_t word = 0;
        if (mode == 0) {
            const float e = load1<WDT>(scale, g);
            const float p = round_to<WDT>(ldexpf(1.0f, (int)e));
#pragma unroll
            for (int k = 0; k < 8; ++k) word |= fp4_nibble_lp<WDT>(round_to<WDT>(w[k] / p)) << (4 * k);
            if ((c * kEPT) % gs == 0) scale_bytes[g] = (uint8_t)clamp3(round_to<WDT>(e + 127.f), 0.f, 255.f);
        } else {
            const float s = ((const float*)scale)[g];
            const float r = recip0(s * recip0(gscale));
#pragma unroll
            for (int k = 0; k < 8; ++k) word |= fp4_nibble_f32(nv_e2m1(clamp3(w[k] * r, -6.f, 6.f))) << (4 * k);
            if ((c * kEPT) % gs == 0) scale_bytes[g] = (uint8_t)f32_to_e4m3(s);
        }
        packed[c] = word;
    }
}

}  // namespace ar

using namespace ar;

static inline int fp4_grid(int64_t chunks) {
    int64_t b = (chunks + kTPB - 1) / kTPB;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

extern "C" int ar_qdq_fp4_fwd(const void* X, const float* V, const float* absmax, const float* max_s, float init_scale,
                              const float* global_scale_dev, void* Xq, void* scale_out, int64_t n_groups, int gs, int mode,
                              int x_dt, float lo_bound, float hi_bound, ar_stream_t stream) {
    if (!((mode == 0 && gs == 32) || (mode == 1 && gs == 16)) || n_groups < 0) return AR_ERR_UNSUPPORTED;
    if (mode == 1 && !global_scale_dev) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    Fp4Args a;
    a.X = X; a.V = V; a.absmax = absmax; a.max_s = max_s; a.gscale = global_scale_dev; a.Xq = Xq; a.scale_out = scale_out;
    a.n_groups = n_groups; a.cpg = gs / kEPT; a.mode = mode; a.init_scale = init_scale; a.lo = lo_bound; a.hi = hi_bound;
    const int grid = fp4_grid(n_groups * a.cpg);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_fp4_fwd<AR_DT_BF16>, grid, kTPB, 0, st, a); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_fp4_fwd<AR_DT_F16>, grid, kTPB, 0, st, a); break;
        case AR_DT_F32: hipLaunchKernelGGL(k_fp4_fwd<AR_DT_F32>, grid, kTPB, 0, st, a); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    return launch_status();
}

extern "C" int ar_pack_fp4(const void* Wq, const void* scale, const float* global_scale_dev, int64_t out_f, int64_t in_f,
                           int gs, int mode, int w_dt, uint8_t* packed, uint8_t* scale_bytes, ar_stream_t stream) {
    if (in_f % gs || gs % kEPT || (mode != 0 && mode != 1)) return AR_ERR_UNSUPPORTED;
    const int64_t n = out_f * in_f;
    if (n == 0) return AR_OK;
    const int grid = fp4_grid(n / kEPT);
    hipStream_t st = (hipStream_t)stream;
    switch (w_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_pack_fp4<AR_DT_BF16>, grid, kTPB, 0, st, Wq, scale, global_scale_dev, n, gs, mode, (uint32_t*)packed, scale_bytes); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_pack_fp4<AR_DT_F16>, grid, kTPB, 0, st, Wq, scale, global_scale_dev, n, gs, mode, (uint32_t*)packed, scale_bytes); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    return launch_status();
}

// fp4 backward + sign-SGD: implemented in ar_fp4_bwd.hip once its oracle is pinned; until then the entry point
// reports "unsupported" instead of silently doing nothing.
#ifndef AR_HAVE_FP4_BWD
extern "C" int ar_qdq_fp4_bwd_sgd(const void*, const void*, float*, const float*, float*, float, const float*, int64_t, int,
                                  int, int, float, float, const float*, const float*, int, const int32_t*, float*, float*,
                                  float*, float*, ar_stream_t) {
    return AR_ERR_UNSUPPORTED;
}
#endif
