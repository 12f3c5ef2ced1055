// ar_fp4.hip -- MXFP4 (e8m0 shared exponent, group 32) and NVFP4 (e4m3 block scale x fp32 global scale, group 16)
// fake-quant forward and the FP4 nibble packer for gfx950.  OCP e4m3fn semantics (gfx950 is OCP, not fnuz).
//
// A group is gs/8 consecutive lanes (4 for MXFP4, 2 for NVFP4); the group absmax either comes precomputed
// (weights: constant during tuning, ar_group_absmax) or is reduced in-wave with a shuffle butterfly (activations).
#include "ar_common.hpp"

namespace ar {

// d out / d t of the MX element rounding for the clamped element t (q = its E2M1 value), applied to dx4 = d loss / d (rounded element):
// autograd's own arithmetic, op by op (quant_element, data_type/mxfp.py:49-85).  With P = 2^pe the chain through the rounding itself
// multiplies and divides by P, 2 and +-1 only -- exact -- so it hands dx4 through; the chain through the private exponent
// pe = floor_ste(log2 |t|).clip(min = 0) is live for |t| >= 1: the two `2.0 ** pe` nodes receive dx4 * (q / P) and -(dx4 P)((t / P) / P),
// PowBackward turns each into grad * (P * ln 2) (two separate roundings, then their sum), Log2Backward divides by |t| * ln 2:
//     dt = dx4 + sign(t) * fl( fl( fl(fl(dx4 q) ln2) - fl(fl(dx4 t) ln2) ) / fl(|t| ln2) )
// (powers of two commute with every rounding).  Same value as dx4 * q / t; these are the bits torch produces.
__device__ __forceinline__ float mx_elem_grad(float dx4, float t, float q) {
    const float LN2F = 0.6931471805599453f;
    if (t == 0.f) return 0.f;
    const float at = fabsf(t);
    if (at < 1.0f) return dx4;
    const float a = (dx4 * q) * LN2F;
    const float b = (dx4 * t) * LN2F;
    const float darg = (a - b) / (at * LN2F);
    return dx4 + (t > 0.f ? darg : -darg);
}


// literal restatement of mxfp.quant_element(ebits=2, mbits=3, max_norm=6, "even") on |t| <= 6
// (auto_round/data_type/mxfp.py:49-85).  floor(log2|t|) clipped at 0 is 0/1/2 by comparison (exact in this range).
__device__ __forceinline__ float mx_e2m1(float t) {
    const float a0 = fabsf(t);
    // 2^private_exp in {1,2,4}: t / 2^pe * 2 and v / 2 * 2^pe are exact power-of-two scalings -> multiplications
    const float up = (a0 >= 4.f) ? 0.5f : ((a0 >= 2.f) ? 1.0f : 2.0f);    // 2 / 2^pe
    const float dn = (a0 >= 4.f) ? 2.0f : ((a0 >= 2.f) ? 1.0f : 0.5f);    // 2^pe / 2
    const float x = t * up;
    const float a = fabsf(x);
    // (a - 0.5) % 2 == 0  <=>  a in {0.5, 2.5, 4.5, ...}  <=>  fract(a/2) == 0.25   (a <= 12 here)
    const float mask = (__builtin_amdgcn_fractf(a * 0.5f) == 0.25f) ? 1.f : 0.f;
    const float v = sgnf(x) * (floorf(a + 0.5f) - mask) * dn;
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
    const void* X; const float* V; const float* absmax; const float* max_s; const float* gscale; const float* init_dev;
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
        rsc = ldexpf(1.0f, -(int)se);      // 1/sc, exact: x / 2^e == x * 2^-e bit for bit (also for subnormal results)
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

#ifndef AR_FP4_FWD_UNROLL
#define AR_FP4_FWD_UNROLL 1
#endif
#ifndef AR_FP4_BWD_UNROLL
#define AR_FP4_BWD_UNROLL 2
#endif
#ifndef AR_FP4_ACT_UNROLL          // chunks per lane of the activation forward (4 B/elem: less HBM time to hide the math)
#define AR_FP4_ACT_UNROLL 2
#endif
#ifndef AR_FP4_ABWD_UNROLL
#define AR_FP4_ABWD_UNROLL 1
#endif

// MODE (0 MXFP4 / 1 NVFP4) and ACT (dynamic activation fake-quant: no V / absmax / max_scale / init scale) are template
// parameters: the group width (4 or 2 lanes) and every mode branch fold at compile time, the butterflies become fixed
// DPP shuffles, and the activation instantiation carries none of the weight path's loads or selects.
template <int XDT, int U, int MODE, bool ACT>
__global__ __launch_bounds__(kTPB) void k_fp4_fwd(const Fp4Args a0) {
    Fp4Args a = a0;
    if (ACT) { a.V = nullptr; a.absmax = nullptr; a.max_s = nullptr; a.init_dev = nullptr; }
    a.mode = MODE;
    constexpr int cpg = MODE == 0 ? 4 : 2;
    const int64_t total_chunks = a.n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB * U;
    const float gscale = (a.mode == 1 && a.gscale) ? *a.gscale : 1.0f;
    // the chunk range is padded up to a multiple of the wave so that every lane of a lane-group joins the butterfly
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    for (int64_t c0 = (int64_t)blockIdx.x * kTPB * U + threadIdx.x; c0 < limit; c0 += stride) {
        Raw8<XDT> xr[U];
        F8 vr[U];
        float am[U], msr[U], inr[U];
        bool ok[U];
        // all streaming loads of this lane are issued before anything waits on them
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + (int64_t)u * kTPB;
            ok[u] = c < total_chunks;
            am[u] = 0.f; msr[u] = 1.f; inr[u] = a.init_scale;
            if (ok[u]) {
                xr[u] = load8_raw<XDT>(a.X, c * kEPT);
                if (a.V) vr[u] = load8_f32(a.V, c * kEPT);
                const int64_t g = c / cpg;
                if (a.absmax) am[u] = a.absmax[g];
                if (a.max_s) msr[u] = a.max_s[g];
                inr[u] = a.init_dev ? a.init_dev[g] : a.init_scale;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + (int64_t)u * kTPB;
            if (c >= limit) break;          // wave-uniform (limit is a multiple of the wave)
            float x[8], v[8], o[8];
            float amax = am[u];
            if (ok[u]) {
                unpack8<XDT>(xr[u], x);
                if (a.V) unpack_f8(vr[u], v);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = 0.f;
                }
                if (!a.absmax) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(x[k]));
                }
            }
            if (!a.absmax) amax = group_max<cpg>(amax);
            if (!ok[u]) continue;
            const int64_t g = c / cpg;
            const float Ms = a.max_s ? clamp3(msr[u], a.lo, a.hi) : 1.0f;
            float sc, aux, rsc;
            fp4_group_scale(a.mode, amax, Ms, inr[u], gscale, sc, aux, rsc);
            if (a.mode == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float t = clamp3(x[k] * rsc + v[k], -6.f, 6.f);    // == x / sc (power-of-two scale)
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
    int64_t b = (chunks + kTPB - 1) / kTPB;   // one pass per workgroup (measured >= capped grid-stride for these streams)
    return (int)(b < 1 ? 1 : (b > (1 << 24) ? (1 << 24) : b));
}

extern "C" int ar_qdq_fp4_fwd(const void* X, const float* V, const float* absmax, const float* max_s, float init_scale,
                              const float* init_scale_dev, const float* global_scale_dev, void* Xq, void* scale_out,
                              int64_t n_groups, int gs, int mode, int x_dt, float lo_bound, float hi_bound,
                              ar_stream_t stream) {
    if (!((mode == 0 && gs == 32) || (mode == 1 && gs == 16)) || n_groups < 0) return AR_ERR_UNSUPPORTED;
    if (mode == 1 && !global_scale_dev) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    Fp4Args a;
    a.X = X; a.V = V; a.absmax = absmax; a.max_s = max_s; a.gscale = global_scale_dev; a.Xq = Xq; a.scale_out = scale_out;
    a.init_dev = init_scale_dev;
    a.n_groups = n_groups; a.cpg = gs / kEPT; a.mode = mode; a.init_scale = init_scale; a.lo = lo_bound; a.hi = hi_bound;
    const bool act = !V && !absmax && !max_s && !init_scale_dev && init_scale == 1.0f;
    const int unroll = act ? AR_FP4_ACT_UNROLL : AR_FP4_FWD_UNROLL;
    const int grid = fp4_grid((n_groups * a.cpg + unroll - 1) / unroll);
    hipStream_t st = (hipStream_t)stream;
#define AR_FWD(DT)                                                                                                   \
    do {                                                                                                             \
        if (mode == 0) {                                                                                             \
            if (act) hipLaunchKernelGGL((k_fp4_fwd<DT, AR_FP4_ACT_UNROLL, 0, true>), grid, kTPB, 0, st, a);          \
            else AR_LAUNCH_PROF(AR_PROF_FP4_FWD, n_groups, (k_fp4_fwd<DT, AR_FP4_FWD_UNROLL, 0, false>), grid, kTPB, 0, st, a); \
        } else {                                                                                                     \
            if (act) hipLaunchKernelGGL((k_fp4_fwd<DT, AR_FP4_ACT_UNROLL, 1, true>), grid, kTPB, 0, st, a);          \
            else AR_LAUNCH_PROF(AR_PROF_FP4_FWD, n_groups, (k_fp4_fwd<DT, AR_FP4_FWD_UNROLL, 1, false>), grid, kTPB, 0, st, a); \
        }                                                                                                            \
    } while (0)
    switch (x_dt) {
        case AR_DT_BF16: AR_FWD(AR_DT_BF16); break;
        case AR_DT_F16: AR_FWD(AR_DT_F16); break;
        case AR_DT_F32: AR_FWD(AR_DT_F32); break;
        default: return AR_ERR_UNSUPPORTED;
    }
#undef AR_FWD
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

// ------------------------------------------------------------------------------------------------------------------
// fp4 backward (+ sign-SGD on V and max_scale, + best snapshot).  Closed forms of what autograd computes through
// quant_mx / nv_fp4 (derivation: oracle/ar_oracle.c oracle_qdq_fp4_bwd):
//   MXFP4: dq/dt = 0 (t==0) | 1 (|t|<1) | q/t (|t|>=1);  dV = g*sc*dq/dt*[|t_pre|<=6];
//          dsc = sum g*q - sum dV*((W/sc)/sc);  dMs = ((dsc*(sc*ln2))/(m*ln2))*amax*init
//   NVFP4: dq/dx = 0 (x==0) | 1;  dV = g*ro*dq/dx*[|x_pre|<=6];
//          dosc = sum dV*W - (sum g*q)*ro^2;  dr = -dosc*osc^2;  ds = dr/gs;  dMs = (((ds*gs)/6)*amax)*init
// Deliberate deviation: for an all-zero group the reference's autograd yields NaN for d max_scale (the unselected
// branch of torch.where(max_val==0, 1, log2(max_val)) / get_reciprocal back-propagates 0*inf) and SignSGD then poisons
// the parameter with NaN; here that gradient is 0 and the group simply stays at its RTN value.
// ------------------------------------------------------------------------------------------------------------------
namespace ar {

struct Fp4BwdArgs {
    const void* dXq; const void* X; float* V; const float* absmax; float* max_s; const float* gscale; const float* init_dev;
    const float* lr_v; const float* lr_mm; const int32_t* snap; float* best_V; float* best_max; float* dV_out; float* dmax_out;
    int64_t n_groups;
    int cpg, mode, tune_minmax;
    float init_scale, lo, hi;
};

template <int XDT, int U, int MODE>
__global__ __launch_bounds__(kTPB) void k_fp4_bwd(const Fp4BwdArgs a0) {
    Fp4BwdArgs a = a0;
    a.mode = MODE;
    constexpr int cpg = MODE == 0 ? 4 : 2;
    const int64_t total_chunks = a.n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB * U;
    const float gscale = (a.mode == 1 && a.gscale) ? *a.gscale : 1.0f;
    const float alpha_v = a.lr_v ? -(*a.lr_v) : 0.f;
    const float alpha_mm = a.lr_mm ? -(*a.lr_mm) : 0.f;
    const bool do_snap = a.snap != nullptr && *a.snap != 0;
    const float LN2 = 0.6931471805599453f;
    const float r6 = (float)(1.0 / 6.0);
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    for (int64_t c0 = (int64_t)blockIdx.x * kTPB * U + threadIdx.x; c0 < limit; c0 += stride) {
        Raw8<XDT> gr[U], wr[U];
        F8 vr[U];
        float am[U], msr[U], inr[U];
        bool okk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + (int64_t)u * kTPB;
            okk[u] = c < total_chunks;
            am[u] = 0.f; msr[u] = 1.f; inr[u] = a.init_scale;
            if (okk[u]) {
                gr[u] = load8_raw<XDT>(a.dXq, c * kEPT);
                wr[u] = load8_raw<XDT>(a.X, c * kEPT);
                if (a.V) vr[u] = load8_f32(a.V, c * kEPT);
                const int64_t g = c / cpg;
                am[u] = a.absmax[g];
                if (a.max_s) msr[u] = a.max_s[g];
                if (a.init_dev) inr[u] = a.init_dev[g];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + (int64_t)u * kTPB;
            if (c >= limit) break;          // wave-uniform
            const bool ok = okk[u];
            const int64_t g = ok ? c / cpg : 0;
            float s_gq = 0.f, s_dvw = 0.f;
            const float amax = am[u], init = inr[u];
            float Ms = 1.f, sc = 1.f, rsc = 1.f, m = 0.f, se_un = 0.f, s_pre = 0.f, r = 0.f;
            if (ok) {
                float gg[8], w[8], v[8], dv[8], tgq[8], tdw[8];
                unpack8<XDT>(gr[u], gg);
                unpack8<XDT>(wr[u], w);
                if (a.V) unpack_f8(vr[u], v);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = 0.f;
                }
                Ms = a.max_s ? clamp3(msr[u], a.lo, a.hi) : 1.0f;
                if (a.mode == 0) {
                    m = amax * (init * Ms);
                    float se = (m == 0.f) ? 1.0f : log2f(m);
                    se_un = floorf(se) - 2.0f;
                    const int sei = (int)clamp3(se_un, -127.f, 127.f);
                    sc = ldexpf(1.0f, sei);
                    rsc = ldexpf(1.0f, -sei);          // exact reciprocal of the power-of-two scale
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float ws = w[k] * rsc;
                        const float tp = ws + v[k];
                        const float t = clamp3(tp, -6.f, 6.f);
                        const float q = mx_e2m1(t);
                        const bool inside = (tp >= -6.f) && (tp <= 6.f);
                        dv[k] = inside ? mx_elem_grad(gg[k] * sc, t, q) : 0.f;
                        tgq[k] = gg[k] * q;
                        tdw[k] = dv[k] * (ws * rsc);
                    }
                } else {
                    const float vm = amax * (Ms * init);
                    s_pre = gscale * (vm * r6);
                    const float s = e4m3_to_f32(f32_to_e4m3(clamp3(s_pre, -448.f, 448.f)));
                    r = s * recip0(gscale);
                    sc = recip0(r);        // osc
                    rsc = recip0(sc);      // ro
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float xp = w[k] * sc + v[k];
                        const float x = clamp3(xp, -6.f, 6.f);
                        const float q = nv_e2m1(x);
                        const bool inside = (xp >= -6.f) && (xp <= 6.f);
                        dv[k] = (inside && x != 0.f) ? gg[k] * rsc : 0.f;
                        tgq[k] = gg[k] * q;
                        tdw[k] = dv[k] * w[k];
                    }
                }
                // rows of 32 / 16 values: torch's reduction kernel gives every element its own thread and combines neighbours
                // first -- a pure pairwise tree (sum8_torch<true>, then the lanes of the group neighbours first)
                s_gq = sum8_torch<true>(tgq);
                s_dvw = sum8_torch<true>(tdw);
                if (a.dV_out) store8_f32(a.dV_out, c * kEPT, dv);
                if (a.lr_v && a.V) {
                    if (do_snap && a.best_V) store8_f32(a.best_V, c * kEPT, v);
                    float vn[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) vn[k] = v[k] + alpha_v * sgnf(dv[k]);
                    store8_f32(a.V, c * kEPT, vn);
                }
            }
            s_gq = group_sum_asc<cpg>(s_gq);
            s_dvw = group_sum_asc<cpg>(s_dvw);
            if (ok && (c % cpg) == 0) {
                float dMs;
                if (a.mode == 0) {
                    const float dsc = s_gq - s_dvw;
                    const bool pass = (se_un >= -127.f) && (se_un <= 127.f);
                    dMs = (m == 0.f || !pass) ? 0.f : ((dsc * (sc * LN2)) / (m * LN2)) * amax * init;
                } else {
                    const float dosc = (sc == 0.f) ? 0.f : (s_dvw - s_gq * (rsc * rsc));
                    const float dr = (r == 0.f) ? 0.f : -dosc * (sc * sc);
                    float ds = dr * recip0(gscale);
                    if (!((s_pre >= -448.f) && (s_pre <= 448.f))) ds = 0.f;
                    dMs = (((ds * gscale) * r6) * amax) * init;
                }
                if (a.dmax_out) a.dmax_out[g] = dMs;
                if (a.lr_mm && a.tune_minmax && a.max_s) {
                    if (do_snap && a.best_max) a.best_max[g] = Ms;
                    a.max_s[g] = Ms + alpha_mm * sgnf(dMs);
                }
            }
        }
    }
}

}  // namespace ar

extern "C" int ar_qdq_fp4_bwd_sgd(const void* dXq, const void* X, float* V, const float* absmax, float* max_s,
                                  float init_scale, const float* init_scale_dev, const float* global_scale_dev,
                                  int64_t n_groups, int gs, int mode,
                                  int x_dt, float lo_bound, float hi_bound, const float* lr_v_dev, const float* lr_mm_dev,
                                  int tune_minmax, const int32_t* snapshot_flag, float* best_V, float* best_max,
                                  float* dV_out, float* dmax_out, ar_stream_t stream) {
    if (!((mode == 0 && gs == 32) || (mode == 1 && gs == 16)) || n_groups < 0 || !absmax) return AR_ERR_UNSUPPORTED;
    if (mode == 1 && !global_scale_dev) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    Fp4BwdArgs a;
    a.dXq = dXq; a.X = X; a.V = V; a.absmax = absmax; a.max_s = max_s; a.gscale = global_scale_dev;
    a.init_dev = init_scale_dev;
    a.lr_v = lr_v_dev; a.lr_mm = lr_mm_dev; a.snap = snapshot_flag; a.best_V = best_V; a.best_max = best_max;
    a.dV_out = dV_out; a.dmax_out = dmax_out; a.n_groups = n_groups; a.cpg = gs / kEPT; a.mode = mode;
    a.tune_minmax = tune_minmax; a.init_scale = init_scale; a.lo = lo_bound; a.hi = hi_bound;
    const int grid = fp4_grid((n_groups * a.cpg + AR_FP4_BWD_UNROLL - 1) / AR_FP4_BWD_UNROLL);
    hipStream_t st = (hipStream_t)stream;
#define AR_BWD(DT)                                                                                                   \
    do {                                                                                                             \
        if (mode == 0) AR_LAUNCH_PROF(AR_PROF_FP4_BWD, n_groups, (k_fp4_bwd<DT, AR_FP4_BWD_UNROLL, 0>), grid, kTPB, 0, st, a); \
        else AR_LAUNCH_PROF(AR_PROF_FP4_BWD, n_groups, (k_fp4_bwd<DT, AR_FP4_BWD_UNROLL, 1>), grid, kTPB, 0, st, a); \
    } while (0)
    switch (x_dt) {
        case AR_DT_BF16: AR_BWD(AR_DT_BF16); break;
        case AR_DT_F16: AR_BWD(AR_DT_F16); break;
        case AR_DT_F32: AR_BWD(AR_DT_F32); break;
        default: return AR_ERR_UNSUPPORTED;
    }
#undef AR_BWD
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// activation fake-quant backward w.r.t. the input (see oracle_fp4_act_bwd for the derivation)
// ------------------------------------------------------------------------------------------------------------------
namespace ar {
__device__ __forceinline__ int lanes_min_i(int v, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) { const int o = __shfl_xor(v, m, kWave); v = o < v ? o : v; }
    return v;
}

template <int XDT, int U, int MODE>
__global__ __launch_bounds__(kTPB) void k_fp4_act_bwd(const void* __restrict__ dXq, const void* __restrict__ X,
                                                      void* __restrict__ dX, const float* __restrict__ gscale_dev,
                                                      int64_t n_groups) {
    constexpr int cpg = MODE == 0 ? 4 : 2;
    constexpr int mode = MODE;
    const int64_t total_chunks = n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB * U;
    const float gscale = (mode == 1 && gscale_dev) ? *gscale_dev : 1.0f;
    const float LN2 = 0.6931471805599453f;
    const float r6 = (float)(1.0 / 6.0);
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    for (int64_t c0 = (int64_t)blockIdx.x * kTPB * U + threadIdx.x; c0 < limit; c0 += stride) {
      Raw8<XDT> gr[U], xr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {          // all loads first
          const int64_t cu = c0 + (int64_t)u * kTPB;
          if (cu < total_chunks) { gr[u] = load8_raw<XDT>(dXq, cu * kEPT); xr[u] = load8_raw<XDT>(X, cu * kEPT); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t c = c0 + (int64_t)u * kTPB;
        if (c >= limit) break;               // wave-uniform
        const bool ok = c < total_chunks;
        float gg[8], x[8], dx[8];
        float lmax = -1.f;
        int lk = 0;
        if (ok) {
            unpack8<XDT>(gr[u], gg);
            unpack8<XDT>(xr[u], x);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float a = fabsf(x[k]); if (a > lmax) { lmax = a; lk = k; } }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { gg[k] = 0.f; x[k] = 0.f; }
        }
        const float amax = group_max<cpg>(lmax);
        const int my_pos = ((int)(c % cpg)) * kEPT + lk;
        const int kstar = group_imin<cpg>((ok && lmax == amax) ? my_pos : 0x7fffffff);   // first index attaining the max
        float s_gq = 0.f, s_dvw = 0.f;
        float tgq[8], tdw[8];
        float sc = 1.f, rsc = 1.f, m = amax, se_un = 0.f, s_pre = 0.f, r = 0.f;
        if (mode == 0) {
            float se = (m == 0.f) ? 1.0f : log2f(m);
            se_un = floorf(se) - 2.0f;
            const int sei = (int)clamp3(se_un, -127.f, 127.f);
            sc = ldexpf(1.0f, sei);
            rsc = ldexpf(1.0f, -sei);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float ws = x[k] * rsc;
                const float t = clamp3(ws, -6.f, 6.f);
                const float q = mx_e2m1(t);
                const bool inside = (ws >= -6.f) && (ws <= 6.f);
                const float dtp = inside ? mx_elem_grad(gg[k] * sc, t, q) : 0.f;
                dx[k] = dtp * rsc;
                tgq[k] = gg[k] * q;
                tdw[k] = dtp * (ws * rsc);
            }
        } else {
            s_pre = gscale * (amax * r6);
            const float s = e4m3_to_f32(f32_to_e4m3(clamp3(s_pre, -448.f, 448.f)));
            r = s * recip0(gscale);
            sc = recip0(r);
            rsc = recip0(sc);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xp = x[k] * sc;
                const float xc = clamp3(xp, -6.f, 6.f);
                const float q = nv_e2m1(xc);
                const bool inside = (xp >= -6.f) && (xp <= 6.f);
                const float dxp = (inside && xc != 0.f) ? gg[k] * rsc : 0.f;
                dx[k] = dxp * sc;
                tgq[k] = gg[k] * q;
                tdw[k] = dxp * x[k];
            }
        }
        s_gq = group_sum_asc<cpg>(sum8_torch<true>(tgq));         // (torch's association for rows of 32 / 16: see k_fp4_bwd)
        s_dvw = group_sum_asc<cpg>(sum8_torch<true>(tdw));
        if (!ok) continue;
        float extra;
        if (mode == 0) {
            const float dsc = s_gq - s_dvw;
            const bool pass = (se_un >= -127.f) && (se_un <= 127.f);
            extra = (m == 0.f || !pass) ? 0.f : (dsc * (sc * LN2)) / (m * LN2);
        } else {
            const float dosc = (sc == 0.f) ? 0.f : (s_dvw - s_gq * (rsc * rsc));
            const float dr = (r == 0.f) ? 0.f : -dosc * (sc * sc);
            float ds = dr * recip0(gscale);
            if (!((s_pre >= -448.f) && (s_pre <= 448.f))) ds = 0.f;
            extra = (ds * gscale) * r6;
        }
        const int base = ((int)(c % cpg)) * kEPT;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool star = (base + k) == kstar;
            if (mode == 0) dx[k] = dx[k] + (star ? extra * sgnf(x[k]) : 0.f);
            else dx[k] = round_to<XDT>(dx[k]) + (star ? round_to<XDT>(extra) * sgnf(x[k]) : 0.f);
        }
        store8<XDT>(dX, c * kEPT, dx);
      }
    }
}
}  // namespace ar

extern "C" int ar_fp4_act_bwd(const void* dXq, const void* X, void* dX, const float* global_scale_dev, int64_t n_groups,
                              int gs, int mode, int x_dt, ar_stream_t stream) {
    if (!((mode == 0 && gs == 32) || (mode == 1 && gs == 16)) || n_groups < 0) return AR_ERR_UNSUPPORTED;
    if (mode == 1 && !global_scale_dev) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    const int cpg = gs / kEPT;
    const int grid = fp4_grid((n_groups * cpg + AR_FP4_ABWD_UNROLL - 1) / AR_FP4_ABWD_UNROLL);
    hipStream_t st = (hipStream_t)stream;
#define AR_ABWD(DT)                                                                                                  \
    do {                                                                                                             \
        if (mode == 0) hipLaunchKernelGGL((k_fp4_act_bwd<DT, AR_FP4_ABWD_UNROLL, 0>), grid, kTPB, 0, st, dXq, X, dX, global_scale_dev, n_groups); \
        else hipLaunchKernelGGL((k_fp4_act_bwd<DT, AR_FP4_ABWD_UNROLL, 1>), grid, kTPB, 0, st, dXq, X, dX, global_scale_dev, n_groups); \
    } while (0)
    switch (x_dt) {
        case AR_DT_BF16: AR_ABWD(AR_DT_BF16); break;
        case AR_DT_F16: AR_ABWD(AR_DT_F16); break;
        case AR_DT_F32: AR_ABWD(AR_DT_F32); break;
        default: return AR_ERR_UNSUPPORTED;
    }
#undef AR_ABWD
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// init-scale search (algorithm extension): try every candidate coefficient on every group, keep the first strictly best
// importance-weighted squared error.  One group per gs/8 lanes, the group's elements stay in registers across all
// candidates; the only cross-lane traffic is one butterfly sum per candidate.
// ------------------------------------------------------------------------------------------------------------------
namespace ar {
template <int XDT>
__global__ __launch_bounds__(kTPB) void k_search_fp4_scale(const void* __restrict__ X, const float* __restrict__ absmax,
                                                           const float* __restrict__ qw_row, int64_t groups_per_row,
                                                           const float* __restrict__ gscale_dev,
                                                           const float* __restrict__ cand, int n_cand,
                                                           float* __restrict__ best_out, int64_t n_groups, int cpg, int mode) {
    const int64_t total_chunks = n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    const float gscale = (mode == 1 && gscale_dev) ? *gscale_dev : 1.0f;
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    for (int64_t c = (int64_t)blockIdx.x * kTPB + threadIdx.x; c < limit; c += stride) {
        const bool ok = c < total_chunks;
        const int64_t g = ok ? c / cpg : 0;
        float x[8], qw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = 0.f; qw[k] = 1.f; }
        float amax = 0.f;
        if (ok) {
            unpack8<XDT>(load8_raw<XDT>(X, c * kEPT), x);
            amax = absmax[g];
            if (qw_row) unpack_f8(load8_f32(qw_row, ((g % groups_per_row) * cpg + (c % cpg)) * kEPT), qw);
        }
        float best = 0.f, best_c = 1.0f;
        for (int ci = 0; ci < n_cand; ++ci) {
            const float coeff = cand[ci];
            float sc, aux, rsc;
            fp4_group_scale(mode, amax, coeff, 1.0f, gscale, sc, aux, rsc);
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float q;
                if (mode == 0) q = mx_e2m1(clamp3(x[k] * rsc, -6.f, 6.f)) * sc;
                else q = nv_e2m1(clamp3(x[k] * sc, -6.f, 6.f)) * rsc;
                const float d = q - x[k];
                t[k] = (d * d) * qw[k];
            }
            // rows of 32 / 16 values: torch's reduction kernel adds them as a pairwise tree, neighbours first (see k_fp4_bwd) -- the
            // bits `loss < best_loss` compares in search_mx_scale / search_nvfp4_scale on the GPU
            const float loss = lanes_sum_torch(sum8_torch<true>(t), cpg);
            if (ci == 0 || loss < best) { best = loss; best_c = coeff; }
        }
        if (ok && (c % cpg) == 0) best_out[g] = best_c;
    }
}
}  // namespace ar

extern "C" int ar_search_fp4_scale(const void* X, const float* absmax, const float* qw_row, int64_t groups_per_row,
                                   const float* global_scale_dev, const float* candidates_dev, int n_candidates,
                                   float* best_out, int64_t n_groups, int gs, int mode, int x_dt, ar_stream_t stream) {
    if (!((mode == 0 && gs == 32) || (mode == 1 && gs == 16)) || n_groups < 0 || n_candidates <= 0 || !absmax) return AR_ERR_UNSUPPORTED;
    if (mode == 1 && !global_scale_dev) return AR_ERR_UNSUPPORTED;
    if (qw_row && groups_per_row <= 0) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    const int cpg = gs / kEPT;
    const int grid = fp4_grid(n_groups * cpg);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_search_fp4_scale<AR_DT_BF16>, grid, kTPB, 0, st, X, absmax, qw_row, groups_per_row, global_scale_dev, candidates_dev, n_candidates, best_out, n_groups, cpg, mode); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_search_fp4_scale<AR_DT_F16>, grid, kTPB, 0, st, X, absmax, qw_row, groups_per_row, global_scale_dev, candidates_dev, n_candidates, best_out, n_groups, cpg, mode); break;
        case AR_DT_F32: hipLaunchKernelGGL(k_search_fp4_scale<AR_DT_F32>, grid, kTPB, 0, st, X, absmax, qw_row, groups_per_row, global_scale_dev, candidates_dev, n_candidates, best_out, n_groups, cpg, mode); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    return launch_status();
}
