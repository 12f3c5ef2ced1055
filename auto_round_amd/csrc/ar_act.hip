// ar_act.hip -- dynamic INT activation fake-quant, symmetric and asymmetric (W4A8 / W8A8-style schemes) for gfx950: forward and the
// gradient w.r.t. the input, autograd mirrored op by op (derivation and dtype choreography: oracle/ar_oracle.c,
// oracle_int_act_fwd / oracle_int_act_bwd).
//
// reference: quant_tensor_sym (auto_round/data_type/int.py:165-238) as called by WrapperLinear._qdq_act
// (auto_round/wrapper.py:295-321) with v = 0, tensor_min/max = None and 0-dim act_min_scale / act_max_scale = 1: the
// range arithmetic stays in the activation dtype (a 0-dim fp32 tensor does not promote a 16-bit tensor).
//
// A group (act_group_size elements along the hidden dimension) is gs/8 consecutive lanes when that is a power of two <= 64
// (g32: 4 lanes, g128: 16 lanes), reduced with a shuffle butterfly; any other multiple of 8 (per-token groups of 4096 ...)
// takes one wave per group that strides over the group's chunks and re-reads them from L2 for the second pass.
#include "ar_common.hpp"

namespace ar {

struct ActQ { float s, s_raw, a, b, sgn, xmin, xmax, wmin, zp; int imin, imax; };

template <int ADT, bool SYM = true>
__device__ __forceinline__ void act_scale(float mn, float mx, int bits, int s_dt, float thresh, ActQ& q) {
    q.xmin = mn; q.xmax = mx;
    const float wmin = mn < 0.f ? mn : 0.f, wmax = mx > 0.f ? mx : 0.f;
    q.wmin = wmin; q.zp = 0.f;
    if (!SYM) {     // quant_tensor_asym (int.py:283-293): range in the activation dtype, zero point in fp32
        const float maxq_a = (float)((1 << bits) - 1);
        q.a = 0.f; q.b = 0.f; q.sgn = 1.f;
        q.s_raw = round_to_rt(s_dt, round_to<ADT>(round_to<ADT>(wmax - wmin) / maxq_a));
        const float ta = round_to_rt(s_dt, thresh);
        q.s = q.s_raw < ta ? ta : q.s_raw;
        q.zp = __builtin_rintf((-wmin) / q.s);
        return;
    }
    const float maxq = (float)(1 << (bits - 1));
    q.a = -wmin; q.b = wmax;
    q.sgn = (q.b < q.a) ? 1.f : -1.f;
    const float m = (q.a > q.b) ? q.a : q.b;
    q.s_raw = round_to_rt(s_dt, round_to<ADT>((q.sgn * m) / maxq));
    const float t = round_to_rt(s_dt, thresh);
    q.s = (q.s_raw < 0.f) ? ((q.s_raw > -t) ? -t : q.s_raw) : ((q.s_raw < t) ? t : q.s_raw);
}

// (value, first index) reductions over `width` lanes
__device__ __forceinline__ void lanes_argmin(float& v, int& idx, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) {
        const float ov = __shfl_xor(v, m, kWave);
        const int oi = __shfl_xor(idx, m, kWave);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}
__device__ __forceinline__ void lanes_argmax_v(float& v, int& idx, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) {
        const float ov = __shfl_xor(v, m, kWave);
        const int oi = __shfl_xor(idx, m, kWave);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// The three quotients per element (x/s, (x/s)/s, dy/s) are correctly rounded: Markstein's 3-instruction form with the group's
// reciprocal y = 1/s, exact inside the guarded exponent range (tools/exactcheck).  The fast instantiation is branch free and
// reports whether any operand left that range; the caller then redoes the chunk with IEEE divisions (wave-uniform, rare).
// qlo / qhi / zp: symmetric -maxq .. maxq-1 with zp 0; asymmetric 0 .. 2^bits-1 with the group's zero point
template <int ADT, int XR, bool FAST>
__device__ __forceinline__ bool act_fwd8_impl(const float (&x)[8], float s, float zp, float qlo, float qhi, float (&o)[8]) {
    const float y = FAST ? 1.0f / s : 0.f;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (FAST) bad = bad || !div_fast_ok(x[k]);
        const float r = round_ste_value(round_to<XR>(FAST ? div_fast(x[k], s, y) : x[k] / s) + 0.f);
        o[k] = s * (clamp3(r + zp, qlo, qhi) - zp);
    }
    return bad;
}
template <int ADT, int XR>
__device__ __forceinline__ void act_fwd8(const float (&x)[8], float s, float zp, float qlo, float qhi, float (&o)[8]) {
    if (__any(act_fwd8_impl<ADT, XR, true>(x, s, zp, qlo, qhi, o))) act_fwd8_impl<ADT, XR, false>(x, s, zp, qlo, qhi, o);
}

// per-chunk part of the backward: direct gradient + the two partial sums of the scale gradient
template <int ADT, int XR, bool FAST>
__device__ __forceinline__ bool act_bwd8_impl(const float (&g)[8], const float (&x)[8], float s, float zp, float qlo, float qhi,
                                              float (&dx)[8], float& acc1, float& acc2, float& acc_e, float& acc_dy) {
    const float y = FAST ? 1.0f / s : 0.f;
    bool bad = false;
    float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float xs = round_to<XR>(FAST ? div_fast(x[k], s, y) : x[k] / s);
        const float r = round_ste_value(xs + 0.f);
        const float tq = r + zp;
        const float qq = clamp3(tq, qlo, qhi) - zp;
        const bool inside = (tq >= qlo) && (tq <= qhi);
        const float e = round_to<XR>(g[k] * s);
        const float dy = inside ? e : 0.f;
        ae += -e; ad += dy;
        if (FAST) bad = bad || !div_fast_ok(x[k]) || !div_fast_ok(xs) || !div_fast_ok(dy);
        dx[k] = round_to<ADT>(round_to<XR>(FAST ? div_fast(dy, s, y) : dy / s)) + 0.f;   // + 0: autograd adds the dense scatter grads
        a1 += round_to<XR>(g[k] * qq);
        a2 += round_to<XR>((-dy) * round_to<XR>(FAST ? div_fast(xs, s, y) : xs / s));
    }
    acc1 = a1; acc2 = a2; acc_e = ae; acc_dy = ad;
    return bad;
}
template <int ADT, int XR>
__device__ __forceinline__ void act_bwd8(const float (&g)[8], const float (&x)[8], float s, float zp, float qlo, float qhi,
                                         float (&dx)[8], float& acc1, float& acc2, float& acc_e, float& acc_dy) {
    // measured: with three quotients per element the IEEE divisions are as fast as the Markstein form plus its range
    // bookkeeping (2.9 vs 2.7 TB/s at group 32), so the backward keeps the plain form; the forward uses the fast one
    float a1, a2, ae, ad;
    act_bwd8_impl<ADT, XR, false>(g, x, s, zp, qlo, qhi, dx, a1, a2, ae, ad);
    acc1 += a1; acc2 += a2; acc_e += ae; acc_dy += ad;
}

template <int ADT, int XR, bool SYM = true>
__device__ __forceinline__ void act_route(const ActQ& q, float sum1, float sum2, float sum_e, float sum_dy, int bits, int s_dt,
                                          float thresh, float& dmin, float& dmax) {
    const float maxq = (float)(1 << (bits - 1));
    const float c1 = round_to_rt(s_dt, round_to<XR>(sum1));
    const float c2 = round_to_rt(s_dt, round_to<XR>(sum2));
    float ds_c = round_to_rt(s_dt, c1 + c2);
    const float t = round_to_rt(s_dt, thresh);
    if (!SYM) {     // zero-point path of quant_tensor_asym, then (wmax - wmin) / maxq in the activation dtype
        const float maxq_a = (float)((1 << bits) - 1);
        const float dzp = sum_e + sum_dy;
        const float u_over_s = ((-q.wmin) / q.s) / q.s;
        ds_c = round_to_rt(s_dt, ds_c + round_to_rt(s_dt, (-dzp) * u_over_s));
        const float dsa = (q.s_raw >= t) ? ds_c : 0.f;
        const float d = round_to<ADT>(round_to<ADT>(dsa) / maxq_a);
        const float dneg = round_to<ADT>(dzp / q.s);
        dmin = (q.xmin <= 0.f) ? round_to<ADT>((-d) + (-dneg)) : 0.f;
        dmax = (q.xmax >= 0.f) ? d : 0.f;
        return;
    }
    const float ds = (q.s_raw < 0.f) ? ((q.s_raw <= -t) ? ds_c : 0.f) : ((q.s_raw >= t) ? ds_c : 0.f);
    const float dm = round_to<ADT>(round_to<ADT>(ds) / maxq) * q.sgn;
    float da, db;
    if (q.a == q.b) { da = round_to<ADT>(dm / 2.f); db = da; }
    else if (q.a > q.b) { da = dm; db = 0.f; }
    else { da = 0.f; db = dm; }
    dmin = (q.xmin <= 0.f) ? -da : 0.f;
    dmax = (q.xmax >= 0.f) ? db : 0.f;
}

// ---- lane-group kernels ----------------------------------------------------------------------------------------------
template <int ADT, int XR, bool BWD, int CPG, bool SYM>
__global__ __launch_bounds__(kTPB) void k_int_act(const void* __restrict__ dXq, const void* __restrict__ X,
                                                  void* __restrict__ Out, void* __restrict__ scale_out, int64_t n_groups,
                                                  int bits, int s_dt, float thresh) {
    constexpr int cpg = CPG;        // lanes per group: compile-time, so the butterflies are fixed DPP shuffles
    const int64_t total_chunks = n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    const float qlo = SYM ? -(float)(1 << (bits - 1)) : 0.f;
    const float qhi = SYM ? (float)(1 << (bits - 1)) - 1.f : (float)((1 << bits) - 1);
    for (int64_t c = (int64_t)blockIdx.x * kTPB + threadIdx.x; c < limit; c += stride) {
        const bool ok = c < total_chunks;
        const int cin = ok ? (int)(c % cpg) : 0;
        float x[8], g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = 0.f; g[k] = 0.f; }
        if (ok) {
            unpack8<ADT>(load8_raw<ADT>(X, c * kEPT), x);
            if (BWD) unpack8<ADT>(load8_raw<ADT>(dXq, c * kEPT), g);
        }
        float mn = x[0], mx = x[0];
        int imn = cin * 8, imx = cin * 8;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            if (x[k] < mn) { mn = x[k]; imn = cin * 8 + k; }
            if (x[k] > mx) { mx = x[k]; imx = cin * 8 + k; }
        }
        if (!ok) { mn = INFINITY; mx = -INFINITY; imn = imx = 0x7fffffff; }
        lanes_argmin(mn, imn, cpg);
        lanes_argmax_v(mx, imx, cpg);
        ActQ q;
        act_scale<ADT, SYM>(mn, mx, bits, s_dt, thresh, q);
        float o[8];
        if (!BWD) {
            act_fwd8<ADT, XR>(x, q.s, q.zp, qlo, qhi, o);
            if (ok) {
                store8<ADT>(Out, c * kEPT, o);
                if (scale_out && cin == 0) store1_rt(s_dt, scale_out, c / cpg, q.s);
            }
        } else {
            float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
            act_bwd8<ADT, XR>(g, x, q.s, q.zp, qlo, qhi, o, a1, a2, ae, ad);
            a1 = lanes_sum(a1, cpg);
            a2 = lanes_sum(a2, cpg);
            if (!SYM) { ae = lanes_sum(ae, cpg); ad = lanes_sum(ad, cpg); }
            float dmin, dmax;
            act_route<ADT, XR, SYM>(q, a1, a2, ae, ad, bits, s_dt, thresh, dmin, dmax);
            if (ok) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int pos = cin * 8 + k;
                    if (pos == imn) o[k] = round_to<ADT>(o[k] + dmin);
                    if (pos == imx) o[k] = round_to<ADT>(o[k] + dmax);
                }
                store8<ADT>(Out, c * kEPT, o);
            }
        }
    }
}

// ---- one wave per group (any gs % 8 == 0) ----------------------------------------------------------------------------
template <int ADT, int XR, bool BWD, bool SYM>
__global__ __launch_bounds__(kTPB) void k_int_act_wave(const void* __restrict__ dXq, const void* __restrict__ X,
                                                       void* __restrict__ Out, void* __restrict__ scale_out,
                                                       int64_t n_groups, int cpg, int bits, int s_dt, float thresh) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / kWave;
    const int64_t n_waves = (int64_t)gridDim.x * (kTPB / kWave);
    const float qlo = SYM ? -(float)(1 << (bits - 1)) : 0.f;
    const float qhi = SYM ? (float)(1 << (bits - 1)) - 1.f : (float)((1 << bits) - 1);
    for (int64_t gi = wave0; gi < n_groups; gi += n_waves) {
        float mn = INFINITY, mx = -INFINITY;
        int imn = 0x7fffffff, imx = 0x7fffffff;
        for (int ch = lane; ch < cpg; ch += kWave) {
            float x[8];
            unpack8<ADT>(load8_raw<ADT>(X, (gi * cpg + ch) * kEPT), x);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (x[k] < mn) { mn = x[k]; imn = ch * 8 + k; }
                if (x[k] > mx) { mx = x[k]; imx = ch * 8 + k; }
            }
        }
        lanes_argmin(mn, imn, kWave);
        lanes_argmax_v(mx, imx, kWave);
        ActQ q;
        act_scale<ADT, SYM>(mn, mx, bits, s_dt, thresh, q);
        if (!BWD) {
            if (scale_out && lane == 0) store1_rt(s_dt, scale_out, gi, q.s);
            for (int ch = lane; ch < cpg; ch += kWave) {
                float x[8], o[8];
                unpack8<ADT>(load8_raw<ADT>(X, (gi * cpg + ch) * kEPT), x);
                act_fwd8<ADT, XR>(x, q.s, q.zp, qlo, qhi, o);
                store8<ADT>(Out, (gi * cpg + ch) * kEPT, o);
            }
            continue;
        }
        float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
        for (int ch = lane; ch < cpg; ch += kWave) {
            float x[8], g[8], o[8];
            unpack8<ADT>(load8_raw<ADT>(X, (gi * cpg + ch) * kEPT), x);
            unpack8<ADT>(load8_raw<ADT>(dXq, (gi * cpg + ch) * kEPT), g);
            act_bwd8<ADT, XR>(g, x, q.s, q.zp, qlo, qhi, o, a1, a2, ae, ad);
            store8<ADT>(Out, (gi * cpg + ch) * kEPT, o);
        }
        a1 = lanes_sum(a1, kWave);
        a2 = lanes_sum(a2, kWave);
        if (!SYM) { ae = lanes_sum(ae, kWave); ad = lanes_sum(ad, kWave); }
        float dmin, dmax;
        act_route<ADT, XR, SYM>(q, a1, a2, ae, ad, bits, s_dt, thresh, dmin, dmax);
        // the two scatter additions: the lane that owns the element re-reads its own (already written) direct gradient
        if (imn / 8 % kWave == lane) {
            const int64_t e = gi * cpg * kEPT + imn;
            store1<ADT>(Out, e, round_to<ADT>(load1<ADT>(Out, e) + dmin));
        }
        if (imx / 8 % kWave == lane) {
            const int64_t e = gi * cpg * kEPT + imx;
            store1<ADT>(Out, e, round_to<ADT>(load1<ADT>(Out, e) + dmax));
        }
    }
}
}  // namespace ar

using namespace ar;

static inline int ilog2_pow2(int v) {
    for (int s = 0; s < 31; ++s) if ((1 << s) == v) return s;
    return -1;
}

template <bool BWD, bool SYM>
static int launch_int_act(const void* dXq, const void* X, void* Out, void* scale_out, int64_t n_groups, int gs, int bits,
                          int a_dt, int s_dt, float q_thresh, ar_stream_t stream) {
    if (gs <= 0 || gs % kEPT || n_groups < 0 || bits < 2 || bits > 8) return AR_ERR_UNSUPPORTED;
    if (s_dt != AR_DT_F16 && s_dt != AR_DT_BF16 && s_dt != AR_DT_F32) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    const int cpg = gs / kEPT;
    const bool lane_groups = cpg <= kWave && ilog2_pow2(cpg) >= 0;
    const bool same16 = a_dt == s_dt && a_dt != AR_DT_F32;      // fp16 activations / fp16 scale: the division stays fp16
    hipStream_t st = (hipStream_t)stream;
    const int64_t want = lane_groups ? (n_groups * cpg + kTPB - 1) / kTPB : (n_groups + kTPB / kWave - 1) / (kTPB / kWave);
    const int grid = (int)(want < 1 ? 1 : (want > (1 << 22) ? (1 << 22) : want));
#define AR_ACT_LG(ADT, XR, C)                                                                                                \
    hipLaunchKernelGGL((k_int_act<ADT, XR, BWD, C, SYM>), grid, kTPB, 0, st, dXq, X, Out, scale_out, n_groups, bits, s_dt, q_thresh)
#define AR_ACT(ADT, XR)                                                                                                      \
    do {                                                                                                                     \
        if (lane_groups) {                                                                                                   \
            switch (cpg) {                                                                                                   \
                case 1: AR_ACT_LG(ADT, XR, 1); break;                                                                        \
                case 2: AR_ACT_LG(ADT, XR, 2); break;                                                                        \
                case 4: AR_ACT_LG(ADT, XR, 4); break;                                                                        \
                case 8: AR_ACT_LG(ADT, XR, 8); break;                                                                        \
                case 16: AR_ACT_LG(ADT, XR, 16); break;                                                                      \
                case 32: AR_ACT_LG(ADT, XR, 32); break;                                                                      \
                default: AR_ACT_LG(ADT, XR, 64); break;                                                                      \
            }                                                                                                                \
        } else hipLaunchKernelGGL((k_int_act_wave<ADT, XR, BWD, SYM>), grid, kTPB, 0, st, dXq, X, Out, scale_out, n_groups, cpg,    \
                                bits, s_dt, q_thresh);                                                                        \
    } while (0)
    switch (a_dt) {
        case AR_DT_BF16: if (same16) AR_ACT(AR_DT_BF16, AR_DT_BF16); else AR_ACT(AR_DT_BF16, AR_DT_F32); break;
        case AR_DT_F16: if (same16) AR_ACT(AR_DT_F16, AR_DT_F16); else AR_ACT(AR_DT_F16, AR_DT_F32); break;
        case AR_DT_F32: AR_ACT(AR_DT_F32, AR_DT_F32); break;
        default: return AR_ERR_UNSUPPORTED;
    }
#undef AR_ACT
#undef AR_ACT_LG
    return launch_status();
}

extern "C" int ar_qdq_int_act_fwd(const void* X, void* Xq, void* scale_out, int64_t n_groups, int gs, int bits, int sym,
                                  int a_dt, int s_dt, float q_thresh, ar_stream_t stream) {
    if (sym) return launch_int_act<false, true>(nullptr, X, Xq, scale_out, n_groups, gs, bits, a_dt, s_dt, q_thresh, stream);
    return launch_int_act<false, false>(nullptr, X, Xq, scale_out, n_groups, gs, bits, a_dt, s_dt, q_thresh, stream);
}

extern "C" int ar_int_act_bwd(const void* dXq, const void* X, void* dX, int64_t n_groups, int gs, int bits, int sym, int a_dt,
                              int s_dt, float q_thresh, ar_stream_t stream) {
    if (!dXq) return AR_ERR_UNSUPPORTED;
    if (sym) return launch_int_act<true, true>(dXq, X, dX, nullptr, n_groups, gs, bits, a_dt, s_dt, q_thresh, stream);
    return launch_int_act<true, false>(dXq, X, dX, nullptr, n_groups, gs, bits, a_dt, s_dt, q_thresh, stream);
}
