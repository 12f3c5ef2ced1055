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
    const float inv_maxq = __uint_as_float((uint32_t)(128 - bits) << 23);      // 1 / 2^(bits-1): the division is exact scaling
    q.a = -wmin; q.b = wmax;
    q.sgn = (q.b < q.a) ? 1.f : -1.f;
    const float m = (q.a > q.b) ? q.a : q.b;
    q.s_raw = round_to_rt(s_dt, round_to<ADT>((q.sgn * m) * inv_maxq));
    const float t = round_to_rt(s_dt, thresh);
    q.s = (q.s_raw < 0.f) ? ((q.s_raw > -t) ? -t : q.s_raw) : ((q.s_raw < t) ? t : q.s_raw);
}

// (value, first index) reductions over `width` lanes (wave-per-group kernel)
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

// The quotients per element (x/s, (x/s)/s, dy/s) are correctly rounded.  Fast form: Markstein's three instructions with the
// group's reciprocal y = 1/s (q0 = w*y; r = fma(-q0, s, w); q = fma(r, y, q0)), exact while nothing under- or overflows
// (tools/exactcheck: every 16-bit w with |w| in [2^-64, 2^64] or w == 0, every admissible scale), written on float2 so that the
// multiplies / fmas issue as v_pk_*_f32 (two elements per instruction).  The window is checked once per chunk on the exponents
// (`chunk_in_window`); a chunk outside it is redone with IEEE divisions (wave-uniform branch, not taken for real activations).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 div_fast2(f32x2 w, f32x2 s, f32x2 y) {
    const f32x2 q0 = w * y;
    const f32x2 r = __builtin_elementwise_fma(-q0, s, w);
    return __builtin_elementwise_fma(r, y, q0);      // a zero w gives a zero of either sign: every user adds +0 or multiplies
}
// x/s and (x/s)/s stay inside the verified window when 2^-40 <= |x| <= 2^36 (or x == 0) and 2^-24 <= |s| <= 2^16
__device__ __forceinline__ bool chunk_in_window(const float (&x)[8], float s) {
    int lo = 0, hi = 0;         // frexp exponent of 0 is 0: zeros pass
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const int e0 = __builtin_amdgcn_frexp_expf(x[k]), e1 = __builtin_amdgcn_frexp_expf(x[k + 1]);
        lo = min(lo, min(e0, e1));
        hi = max(hi, max(e0, e1));
    }
    const int es = __builtin_amdgcn_frexp_expf(s);       // 0 for an infinite or NaN scale (an infinite activation): excluded below
    return lo >= -39 && hi <= 36 && es >= -23 && es <= 17 && __builtin_fabsf(s) < INFINITY;
}
// (rint(q) + 0 is what round_ste_value(q + 0.f) returns for a finite q: rint with a zero result always +0)

// qlo / qhi / zp: symmetric -maxq .. maxq-1 with zp 0; asymmetric 0 .. 2^bits-1 with the group's zero point
template <int ADT, int XR, bool SYM>
__device__ __forceinline__ void act_fwd8_fast(const float (&x)[8], float s, float y, float zp, float qlo, float qhi, float (&o)[8]) {
    const f32x2 s2 = {s, s}, y2 = {y, y}, zp2 = {zp, zp};
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x2 xp = {x[k], x[k + 1]};
        const f32x2 q = div_fast2(xp, s2, y2);
        f32x2 t;
        if constexpr (XR == AR_DT_F32) t = f32x2{__builtin_rintf(q.x), __builtin_rintf(q.y)} + f32x2{0.f, 0.f};
        else { t.x = round_ste_value(round_to<XR>(q.x) + 0.f); t.y = round_ste_value(round_to<XR>(q.y) + 0.f); }
        if constexpr (!SYM) t = t + zp2;
        f32x2 c = {clamp3(t.x, qlo, qhi), clamp3(t.y, qlo, qhi)};
        if constexpr (!SYM) c = c - zp2;
        const f32x2 w = s2 * c;
        o[k] = w.x; o[k + 1] = w.y;
    }
}
template <int ADT, int XR>
__device__ __forceinline__ void act_fwd8_ieee(const float (&x)[8], float s, float zp, float qlo, float qhi, float (&o)[8]) {
    asm volatile("; IEEE-division form" ::: "memory");      // not speculatable: the compiler must keep this behind its branch
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float r = round_ste_value(round_to<XR>(x[k] / s) + 0.f);
        o[k] = s * (clamp3(r + zp, qlo, qhi) - zp);
    }
}
template <int ADT, int XR, bool SYM>
__device__ __forceinline__ void act_fwd8(const float (&x)[8], float s, float zp, float qlo, float qhi, float (&o)[8]) {
    if (__all(chunk_in_window(x, s))) act_fwd8_fast<ADT, XR, SYM>(x, s, 1.0f / s, zp, qlo, qhi, o);
    else act_fwd8_ieee<ADT, XR>(x, s, zp, qlo, qhi, o);
}
// one chunk of the row kernels, result stored: the IEEE form re-reads its elements one at a time (no registers for the rare path)
template <int ADT, int XR, bool SYM>
__device__ __forceinline__ void act_fwd_chunk(const void* __restrict__ X, void* __restrict__ Out, int64_t at, const float (&x)[8],
                                              float s, float y, float zp, float qlo, float qhi) {
    if (__all(chunk_in_window(x, s))) {
        float o[8];
        act_fwd8_fast<ADT, XR, SYM>(x, s, y, zp, qlo, qhi, o);
        store8<ADT>(Out, at, o);
        return;
    }
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
        const float r = round_ste_value(round_to<XR>(load1<ADT>(X, at + j) / s) + 0.f);
        store1<ADT>(Out, at + j, s * (clamp3(r + zp, qlo, qhi) - zp));
    }
}

// per-chunk part of the backward: direct gradient + the partial sums of the scale (and zero-point) gradient.
// Fast form (x inside the window).  With fp32 products and a 16-bit activation dtype the direct gradient needs no division:
// e = fl(g*s) and fl(e/s) differ from g by at most 2^-23 relative while g's neighbours in the 16-bit format are 2^-11 (2^-8) away,
// so round_ADT(fl(fl(g*s)/s)) == g whenever e is a normal number (checked per element; a zero, subnormal, infinite or NaN e
// with g != 0 sends the chunk to the IEEE form).  The partial sums run as two interleaved accumulators (even / odd elements).
template <int ADT, int XR, bool SYM>
__device__ __forceinline__ bool act_bwd8_fast(const float (&g)[8], const float (&x)[8], float s, float y, float zp, float qlo,
                                              float qhi, float (&dx)[8], float& acc1, float& acc2, float& acc_e, float& acc_dy) {
    constexpr bool kNoDiv = (XR == AR_DT_F32) && (ADT != AR_DT_F32);
    const f32x2 s2 = {s, s}, y2 = {y, y}, zp2 = {zp, zp}, zero2 = {0.f, 0.f};
    f32x2 a1 = zero2, a2 = zero2, ae = zero2, ad = zero2;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x2 xp = {x[k], x[k + 1]}, gp = {g[k], g[k + 1]};
        f32x2 xs = div_fast2(xp, s2, y2);
        f32x2 t;
        if constexpr (XR == AR_DT_F32) t = f32x2{__builtin_rintf(xs.x), __builtin_rintf(xs.y)} + zero2;
        else {
            xs.x = round_to<XR>(xs.x); xs.y = round_to<XR>(xs.y);
            t.x = round_ste_value(xs.x + 0.f); t.y = round_ste_value(xs.y + 0.f);
        }
        if constexpr (!SYM) t = t + zp2;
        f32x2 c = {clamp3(t.x, qlo, qhi), clamp3(t.y, qlo, qhi)};
        const bool in0 = (c.x == t.x), in1 = (c.y == t.y);
        if constexpr (!SYM) c = c - zp2;
        f32x2 e = gp * s2;
        if constexpr (XR != AR_DT_F32) { e.x = round_to<XR>(e.x); e.y = round_to<XR>(e.y); }
        const f32x2 dy = {in0 ? e.x : 0.f, in1 ? e.y : 0.f};
        if constexpr (!SYM) { ae = ae - e; ad = ad + dy; }
        if constexpr (kNoDiv) {
            constexpr int kNotNormal = 0x3ff & ~(0x008 | 0x100);
            bad = bad || (__builtin_amdgcn_classf(e.x, kNotNormal) && gp.x != 0.f)
                      || (__builtin_amdgcn_classf(e.y, kNotNormal) && gp.y != 0.f);
            const f32x2 d = f32x2{in0 ? gp.x : 0.f, in1 ? gp.y : 0.f} + zero2;
            dx[k] = d.x; dx[k + 1] = d.y;
        } else {
            dx[k] = round_to<ADT>(round_to<XR>(dy.x / s)) + 0.f;
            dx[k + 1] = round_to<ADT>(round_to<XR>(dy.y / s)) + 0.f;
        }
        f32x2 p1 = gp * c;
        f32x2 xss = div_fast2(xs, s2, y2);
        if constexpr (XR != AR_DT_F32) { p1.x = round_to<XR>(p1.x); p1.y = round_to<XR>(p1.y); xss.x = round_to<XR>(xss.x); xss.y = round_to<XR>(xss.y); }
        f32x2 p2 = (-dy) * xss;
        if constexpr (XR != AR_DT_F32) { p2.x = round_to<XR>(p2.x); p2.y = round_to<XR>(p2.y); }
        a1 = a1 + p1;
        a2 = a2 + p2;
    }
    acc1 += a1.x + a1.y; acc2 += a2.x + a2.y;
    if constexpr (!SYM) { acc_e += ae.x + ae.y; acc_dy += ad.x + ad.y; }
    return bad;
}
// CH chunks of one lane in the fast form; returns false when the wave has to redo them with IEEE divisions (an operand outside
// the window, or a product that left the normal range)
template <int ADT, int XR, bool SYM, int CH>
__device__ __forceinline__ bool act_bwd_lane_fast(const float (&g)[CH][8], const float (&x)[CH][8], float s, float zp, float qlo,
                                                  float qhi, float (&dx)[CH][8], float& acc1, float& acc2, float& acc_e,
                                                  float& acc_dy) {
    bool in_window = true;
#pragma unroll
    for (int h = 0; h < CH; ++h) in_window = in_window && chunk_in_window(x[h], s);
    if (!__all(in_window)) return false;
    const float y = 1.0f / s;
    float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
    bool bad = false;
#pragma unroll
    for (int h = 0; h < CH; ++h) bad = act_bwd8_fast<ADT, XR, SYM>(g[h], x[h], s, y, zp, qlo, qhi, dx[h], a1, a2, ae, ad) || bad;
    if (__any(bad)) return false;
    acc1 += a1; acc2 += a2; acc_e += ae; acc_dy += ad;
    return true;
}
// The IEEE form of a lane for the lane-group kernel: one element at a time, re-read from memory and written straight to dX, so
// that the rare path costs the common one no registers (unrolled next to it, its 6 * CH * 8 divisions took the kernel from
// 94 to 156 VGPRs).  The scatter additions then go through memory as well (k_int_act).
template <int ADT, int XR, int E>
__device__ __forceinline__ void act_bwd_slow(const void* __restrict__ dXq, const void* __restrict__ X, void* __restrict__ dX,
                                             int64_t at, bool ok, float s, float zp, float qlo, float qhi, float& acc1,
                                             float& acc2, float& acc_e, float& acc_dy) {
    float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
#pragma unroll 1
    for (int j = 0; j < E; ++j) {
        const float xv = load1<ADT>(X, at + j), gv = load1<ADT>(dXq, at + j);
        const float xs = round_to<XR>(xv / s);
        const float r = round_ste_value(xs + 0.f);
        const float tq = r + zp;
        const float qq = clamp3(tq, qlo, qhi) - zp;
        const bool inside = (tq >= qlo) && (tq <= qhi);
        const float e = round_to<XR>(gv * s);
        const float dy = inside ? e : 0.f;
        ae += -e; ad += dy;
        const float d = round_to<ADT>(round_to<XR>(dy / s)) + 0.f;
        if (ok) store1<ADT>(dX, at + j, d);
        a1 += round_to<XR>(gv * qq);
        a2 += round_to<XR>((-dy) * round_to<XR>(xs / s));
    }
    acc1 += a1; acc2 += a2; acc_e += ae; acc_dy += ad;
}
// one chunk of the row kernels, direct gradient stored: fast form, else the memory IEEE form
template <int ADT, int XR, bool SYM>
__device__ __forceinline__ void act_bwd_chunk(const void* __restrict__ dXq, const void* __restrict__ X, void* __restrict__ dX,
                                              int64_t at, const float (&g)[1][8], const float (&x)[1][8], float s, float zp,
                                              float qlo, float qhi, float& acc1, float& acc2, float& acc_e, float& acc_dy) {
    float o[1][8];
    if (act_bwd_lane_fast<ADT, XR, SYM, 1>(g, x, s, zp, qlo, qhi, o, acc1, acc2, acc_e, acc_dy)) { store8<ADT>(dX, at, o[0]); return; }
    act_bwd_slow<ADT, XR, 8>(dXq, X, dX, at, true, s, zp, qlo, qhi, acc1, acc2, acc_e, acc_dy);
}

template <int ADT, int XR, bool SYM = true>
__device__ __forceinline__ void act_route(const ActQ& q, float sum1, float sum2, float sum_e, float sum_dy, int bits, int s_dt,
                                          float thresh, float& dmin, float& dmax) {
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
    const float dm = round_to<ADT>(round_to<ADT>(ds) * __uint_as_float((uint32_t)(128 - bits) << 23)) * q.sgn;
    float da, db;
    if (q.a == q.b) { da = round_to<ADT>(dm * 0.5f); db = da; }
    else if (q.a > q.b) { da = dm; db = 0.f; }
    else { da = 0.f; db = dm; }
    dmin = (q.xmin <= 0.f) ? -da : 0.f;
    dmax = (q.xmax >= 0.f) ? db : 0.f;
}

// keeps a load where it was written (the scheduler otherwise sinks the second load behind the first one's wait)
template <int DT> __device__ __forceinline__ void pin_raw(Raw8<DT>& r) {
    if constexpr (DT == AR_DT_F32)
        asm volatile("" : "+v"(r.a.x), "+v"(r.a.y), "+v"(r.a.z), "+v"(r.a.w), "+v"(r.b.x), "+v"(r.b.y), "+v"(r.b.z), "+v"(r.b.w));
    else
        asm volatile("" : "+v"(r.q.x), "+v"(r.q.y), "+v"(r.q.z), "+v"(r.q.w));
}

#ifndef AR_ACT_WAVES
#define AR_ACT_WAVES 1
#endif
// ---- lane-group kernels ----------------------------------------------------------------------------------------------
// A lane owns CH consecutive 16-byte chunks (8 * CH elements) of its group; the backward takes CH = 2 where the group size
// allows it, which halves the per-lane share of the group arithmetic (range, scale, routing, butterflies).
template <int ADT, int XR, bool BWD, int CPG, bool SYM, int CH>
__global__ __launch_bounds__(kTPB) __attribute__((amdgpu_waves_per_eu(AR_ACT_WAVES))) void k_int_act(const void* __restrict__ dXq, const void* __restrict__ X,
                                                  void* __restrict__ Out, void* __restrict__ scale_out, int64_t n_groups,
                                                  int bits, int s_dt, float thresh) {
    constexpr int cpg = CPG;        // lanes per group: compile-time, so the butterflies are fixed DPP controls
    constexpr int E = kEPT * CH;    // elements per lane
    constexpr int kNone = 0x7fffffff;
    const int64_t total_lanes = n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    const int64_t limit = (total_lanes + kWave - 1) / kWave * kWave;
    const float qlo = SYM ? -(float)(1 << (bits - 1)) : 0.f;
    const float qhi = SYM ? (float)(1 << (bits - 1)) - 1.f : (float)((1 << bits) - 1);
    for (int64_t c = (int64_t)blockIdx.x * kTPB + threadIdx.x; c < limit; c += stride) {
        const bool ok = c < total_lanes;
        const int cin = (int)(threadIdx.x & (cpg - 1));     // kTPB is a multiple of every cpg
        const int64_t at = (ok ? c : 0) * E;                // lanes past the end read lane 0's elements and are masked below
        float x[CH][8], g[CH][8];
        {   // every load in flight before the first use
            Raw8<ADT> rx[CH], rg[CH];
#pragma unroll
            for (int h = 0; h < CH; ++h) rx[h] = load8_raw<ADT>(X, at + h * kEPT);
            if (BWD) {
#pragma unroll
                for (int h = 0; h < CH; ++h) rg[h] = load8_raw<ADT>(dXq, at + h * kEPT);
#pragma unroll
                for (int h = 0; h < CH; ++h) { pin_raw(rx[h]); pin_raw(rg[h]); }
#pragma unroll
                for (int h = 0; h < CH; ++h) unpack8<ADT>(rg[h], g[h]);
            }
#pragma unroll
            for (int h = 0; h < CH; ++h) unpack8<ADT>(rx[h], x[h]);
        }
        // group range first (values only), then -- backward -- the first position that holds it
        float mn = x[0][0], mx = x[0][0];
#pragma unroll
        for (int h = 0; h < CH; ++h) {
#pragma unroll
            for (int k = (h == 0 ? 1 : 0); k < 8; ++k) { mn = __builtin_fminf(mn, x[h][k]); mx = __builtin_fmaxf(mx, x[h][k]); }
        }
        if (!ok) { mn = INFINITY; mx = -INFINITY; }
        mn = group_min<cpg>(mn);
        mx = group_max<cpg>(mx);
        ActQ q;
        act_scale<ADT, SYM>(mn, mx, bits, s_dt, thresh, q);
        float o[CH][8];
        if (!BWD) {
#pragma unroll
            for (int h = 0; h < CH; ++h) act_fwd8<ADT, XR, SYM>(x[h], q.s, q.zp, qlo, qhi, o[h]);
            if (ok) {
#pragma unroll
                for (int h = 0; h < CH; ++h) store8<ADT>(Out, at + h * kEPT, o[h]);
                if (scale_out && cin == 0) store1_rt(s_dt, scale_out, c / cpg, q.s);
            }
        } else {
            int kmn = E, kmx = E;
#pragma unroll
            for (int h = CH - 1; h >= 0; --h) {
#pragma unroll
                for (int k = 7; k >= 0; --k) {
                    kmn = (x[h][k] == mn) ? h * 8 + k : kmn;
                    kmx = (x[h][k] == mx) ? h * 8 + k : kmx;
                }
            }
            const int imn = group_imin<cpg>((ok && kmn < E) ? cin * E + kmn : kNone);
            const int imx = group_imin<cpg>((ok && kmx < E) ? cin * E + kmx : kNone);
            float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
            const bool fast = act_bwd_lane_fast<ADT, XR, SYM, CH>(g, x, q.s, q.zp, qlo, qhi, o, a1, a2, ae, ad);    // wave-uniform
            if (!fast) act_bwd_slow<ADT, XR, E>(dXq, X, Out, at, ok, q.s, q.zp, qlo, qhi, a1, a2, ae, ad);
            a1 = group_sum<cpg>(a1);
            a2 = group_sum<cpg>(a2);
            if (!SYM) { ae = group_sum<cpg>(ae); ad = group_sum<cpg>(ad); }
            float dmin, dmax;
            act_route<ADT, XR, SYM>(q, a1, a2, ae, ad, bits, s_dt, thresh, dmin, dmax);
            // the two scatter additions (autograd: direct gradient, + min scatter, + max scatter, each rounded to the activation
            // dtype).  o[] already holds activation-dtype values, so one addition followed by the rounding of the store is the
            // same thing unless one element is both the arg-min and the arg-max
            const int lmn = imn - cin * E, lmx = imx - cin * E;     // 0..E-1 on the owning lane
            // a constant group has both on one element (two roundings): that case, like the IEEE form, adds through memory
            const bool in_regs = fast && !__any(imn == imx);
            if (in_regs) {
#pragma unroll
                for (int h = 0; h < CH; ++h) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[h][k] += (h * 8 + k == lmn) ? dmin : ((h * 8 + k == lmx) ? dmax : 0.f);
                }
            }
            if (fast && ok) {
#pragma unroll
                for (int h = 0; h < CH; ++h) store8<ADT>(Out, at + h * kEPT, o[h]);
            }
            if (!in_regs) {     // the direct gradient is in memory: the owning lane adds to its own element
                if (ok && lmn >= 0 && lmn < E) store1<ADT>(Out, at + lmn, load1<ADT>(Out, at + lmn) + dmin);
                if (ok && lmx >= 0 && lmx < E) store1<ADT>(Out, at + lmx, load1<ADT>(Out, at + lmx) + dmax);
            }
        }
    }
}

// ---- a team of waves per group, the group parked in LDS ----------------------------------------------------------------------
// Per-token groups (the reference's INT8 / INT4 presets: act_group_size -1, i.e. the hidden size) need two passes over the group --
// its range, then the quantisation -- and with thousands of waves in flight the second read no longer hits L2 (measured: 6 and
// 8 B/element of HBM-side traffic against 4 and 6 algorithmic).  Here a team of WAVES waves (1, or the 4 of the workgroup) owns a
// group; every thread parks the 16-byte chunks it read (chunk = position in team + k * team size, coalesced) in LDS and takes them
// back for the later passes, so X is read from memory once.  A thread only ever re-reads its own pieces: LDS is a per-thread
// extension of the register file here and needs no barrier; the team reduces through DPP / ds_bpermute inside a wave and one
// small LDS exchange across waves.  (Keeping the pieces in registers instead unrolls every pass over them: 250+ VGPRs.)
#ifndef AR_ACT_ROW_WAVE_MAX
#define AR_ACT_ROW_WAVE_MAX 512     // chunks (of 8 elements) up to which one wave owns a group (4096 elements)
#endif
template <int WAVES> struct TeamLds { float f[2][6][4]; };
// all-reduce of N values across the waves of a team (each already reduced inside its wave): one LDS exchange, one barrier
template <int WAVES, int N, class F>
__device__ __forceinline__ void team_combine(float (&v)[N], float (*slot)[4], int wave, F op) {
    if constexpr (WAVES > 1) {
        if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
            for (int n = 0; n < N; ++n) slot[n][wave] = v[n];
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float r = slot[n][0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) r = op(n, r, slot[n][w]);
            v[n] = r;
        }
    }
}
template <int DT> __device__ __forceinline__ void park(uint4* slot, const Raw8<DT>& r) {
    if constexpr (DT == AR_DT_F32) {
        slot[0] = make_uint4(__float_as_uint(r.a.x), __float_as_uint(r.a.y), __float_as_uint(r.a.z), __float_as_uint(r.a.w));
        slot[1] = make_uint4(__float_as_uint(r.b.x), __float_as_uint(r.b.y), __float_as_uint(r.b.z), __float_as_uint(r.b.w));
    } else slot[0] = r.q;
}
template <int DT> __device__ __forceinline__ Raw8<DT> unpark(const uint4* slot) {
    Raw8<DT> r;
    if constexpr (DT == AR_DT_F32) {
        const uint4 a = slot[0], b = slot[1];
        r.a = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        r.b = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
    } else r.q = slot[0];
    return r;
}

template <int ADT, int XR, bool BWD, bool SYM, int WAVES>
__global__ __launch_bounds__(kTPB) void k_int_act_row(const void* __restrict__ dXq, const void* __restrict__ X,
                                                      void* __restrict__ Out, void* __restrict__ scale_out, int64_t n_groups,
                                                      int cpg, int bits, int s_dt, float thresh) {
    constexpr int T = kWave * WAVES;            // threads per team
    constexpr int P = (ADT == AR_DT_F32) ? 2 : 1;   // 16-byte pieces per chunk
    constexpr int kNone = 0x7fffffff;
    extern __shared__ uint4 row_lds[];          // [teams per workgroup][cpg * P]
    __shared__ TeamLds<WAVES> lds;
    const int tid = threadIdx.x & (T - 1);      // position in the team
    const int wave = (threadIdx.x / kWave) & (WAVES - 1);
    uint4* mine = row_lds + (size_t)(threadIdx.x / T) * cpg * P;
    const int64_t team0 = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / T;
    const int64_t n_teams = (int64_t)gridDim.x * (kTPB / T);
    const float qlo = SYM ? -(float)(1 << (bits - 1)) : 0.f;
    const float qhi = SYM ? (float)(1 << (bits - 1)) - 1.f : (float)((1 << bits) - 1);
    for (int64_t gi = team0; gi < n_groups; gi += n_teams) {
        if constexpr (WAVES > 1) __syncthreads();       // the previous group's exchange slots are free again
        const int64_t base = gi * cpg * kEPT;
        float mn = INFINITY, mx = -INFINITY;
        for (int c0 = tid; c0 < cpg; c0 += 2 * T) {     // two chunks per trip: both loads in flight
            const int c1 = c0 + T;
            const bool two = c1 < cpg;
            Raw8<ADT> r0 = load8_raw<ADT>(X, base + (int64_t)c0 * kEPT);
            Raw8<ADT> r1 = load8_raw<ADT>(X, base + (int64_t)(two ? c1 : c0) * kEPT);
            pin_raw(r0); pin_raw(r1);
            park<ADT>(mine + c0 * P, r0);
            if (two) park<ADT>(mine + c1 * P, r1);
            float x[8];
            unpack8<ADT>(r0, x);
#pragma unroll
            for (int k = 0; k < 8; ++k) { mn = __builtin_fminf(mn, x[k]); mx = __builtin_fmaxf(mx, x[k]); }
            unpack8<ADT>(r1, x);        // (the first chunk again when there is no second one)
#pragma unroll
            for (int k = 0; k < 8; ++k) { mn = __builtin_fminf(mn, x[k]); mx = __builtin_fmaxf(mx, x[k]); }
        }
        {
            float v[2] = {group_min<kWave>(mn), group_max<kWave>(mx)};
            team_combine<WAVES>(v, lds.f[0], wave, [](int n, float a, float b) { return n == 0 ? op_min(a, b) : op_max(a, b); });
            mn = v[0]; mx = v[1];
        }
        ActQ q;
        act_scale<ADT, SYM>(mn, mx, bits, s_dt, thresh, q);
        if (!BWD) {
            if (scale_out && tid == 0) store1_rt(s_dt, scale_out, gi, q.s);
            const float y = 1.0f / q.s;
            for (int ch = tid; ch < cpg; ch += T) {
                float x[8];
                unpack8<ADT>(unpark<ADT>(mine + ch * P), x);
                act_fwd_chunk<ADT, XR, SYM>(X, Out, base + (int64_t)ch * kEPT, x, q.s, y, q.zp, qlo, qhi);
            }
            continue;
        }
        // backward: direct gradient and partial sums; the first position of each range end on the way (a chunk that holds one is
        // rare, so only the equality tests are paid per element)
        int imn = kNone, imx = kNone;
        float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
        for (int c0 = tid; c0 < cpg; c0 += 2 * T) {
            const int c1 = c0 + T;
            const bool two = c1 < cpg;
            Raw8<ADT> g0 = load8_raw<ADT>(dXq, base + (int64_t)c0 * kEPT);
            Raw8<ADT> g1 = load8_raw<ADT>(dXq, base + (int64_t)(two ? c1 : c0) * kEPT);
            pin_raw(g0); pin_raw(g1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ch = u ? c1 : c0;
                if (u == 0 || two) {
                    float x[1][8], g[1][8];
                    unpack8<ADT>(u ? g1 : g0, g[0]);
                    unpack8<ADT>(unpark<ADT>(mine + ch * P), x[0]);
                    bool hmn = false, hmx = false;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { hmn = hmn || (x[0][k] == mn); hmx = hmx || (x[0][k] == mx); }
                    if (__any(hmn || hmx)) {
                        int kmn = 8, kmx = 8;
#pragma unroll
                        for (int k = 7; k >= 0; --k) { kmn = (x[0][k] == mn) ? k : kmn; kmx = (x[0][k] == mx) ? k : kmx; }
                        if (kmn < 8) imn = min(imn, ch * 8 + kmn);
                        if (kmx < 8) imx = min(imx, ch * 8 + kmx);
                    }
                    act_bwd_chunk<ADT, XR, SYM>(dXq, X, Out, base + (int64_t)ch * kEPT, g, x, q.s, q.zp, qlo, qhi, a1, a2, ae, ad);
                }
                __builtin_amdgcn_sched_barrier(0);      // one chunk's arithmetic at a time (registers)
            }
        }
        {
            float v[6] = {__int_as_float(group_imin<kWave>(imn)), __int_as_float(group_imin<kWave>(imx)), group_sum<kWave>(a1),
                          group_sum<kWave>(a2), SYM ? 0.f : group_sum<kWave>(ae), SYM ? 0.f : group_sum<kWave>(ad)};
            team_combine<WAVES>(v, lds.f[1], wave, [](int n, float a, float b) {
                return n < 2 ? __int_as_float(op_imin(__float_as_int(a), __float_as_int(b))) : a + b;
            });
            imn = __float_as_int(v[0]); imx = __float_as_int(v[1]); a1 = v[2]; a2 = v[3]; ae = v[4]; ad = v[5];
        }
        float dmin, dmax;
        act_route<ADT, XR, SYM>(q, a1, a2, ae, ad, bits, s_dt, thresh, dmin, dmax);
        // the two scatter additions: the thread that owns the element re-reads its own (already written) direct gradient
        if (imn != kNone && (imn >> 3) % T == tid) {
            const int64_t e = base + imn;
            store1<ADT>(Out, e, round_to<ADT>(load1<ADT>(Out, e) + dmin));
        }
        if (imx != kNone && (imx >> 3) % T == tid) {
            const int64_t e = base + imx;
            store1<ADT>(Out, e, round_to<ADT>(load1<ADT>(Out, e) + dmax));
        }
    }
}

// ---- one wave per group, two reads of the group (any gs % 8 == 0 that the register-resident form does not cover) -------
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
                float x[8];
                unpack8<ADT>(load8_raw<ADT>(X, (gi * cpg + ch) * kEPT), x);
                act_fwd_chunk<ADT, XR, SYM>(X, Out, (gi * cpg + ch) * kEPT, x, q.s, 1.0f / q.s, q.zp, qlo, qhi);
            }
            continue;
        }
        float a1 = 0.f, a2 = 0.f, ae = 0.f, ad = 0.f;
        for (int ch = lane; ch < cpg; ch += kWave) {
            float x[1][8], g[1][8];
            unpack8<ADT>(load8_raw<ADT>(X, (gi * cpg + ch) * kEPT), x[0]);
            unpack8<ADT>(load8_raw<ADT>(dXq, (gi * cpg + ch) * kEPT), g[0]);
            act_bwd_chunk<ADT, XR, SYM>(dXq, X, Out, (gi * cpg + ch) * kEPT, g, x, q.s, q.zp, qlo, qhi, a1, a2, ae, ad);
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
    // the backward gives a lane two chunks when the group is a power-of-two number of 32-byte pieces
    const bool two = BWD && gs % (2 * kEPT) == 0 && ilog2_pow2(gs / (2 * kEPT)) >= 0 && gs / (2 * kEPT) <= kWave;
    const int cpg = two ? gs / (2 * kEPT) : gs / kEPT;
    const bool lane_groups = cpg <= kWave && ilog2_pow2(cpg) >= 0;
    const bool same16 = a_dt == s_dt && a_dt != AR_DT_F32;      // fp16 activations / fp16 scale: the division stays fp16
    hipStream_t st = (hipStream_t)stream;
    // groups that are not a lane group: parked in the LDS of one wave or of the four waves of a workgroup, as long as a workgroup's
    // rows fit 64 KB of LDS; anything larger takes the two-read kernel
    int row_waves = 0;
    size_t row_lds_bytes = 0;
    if (!lane_groups) {
        const int c8 = gs / kEPT;
        const size_t row_bytes = (size_t)c8 * (a_dt == AR_DT_F32 ? 32 : 16);
        // measured at 4096-element groups: forward 5.7 TB/s with the workgroup as the team against 5.1 with a wave, backward 3.9
        // against 4.9 -- so the forward switches to the workgroup as soon as every thread has a chunk
        row_waves = (c8 <= (BWD ? AR_ACT_ROW_WAVE_MAX : 255)) ? 1 : 4;
        row_lds_bytes = row_bytes * (kTPB / (kWave * row_waves));
        if (row_lds_bytes + 512 > 65536 && row_waves == 1) { row_waves = 4; row_lds_bytes = row_bytes; }
        if (row_lds_bytes + 512 > 65536) row_waves = 0;
    }
    const int row_nc = row_waves;
    const int64_t teams_per_block = row_nc ? kTPB / (kWave * row_waves) : kTPB / kWave;
    const int64_t want = lane_groups ? (n_groups * cpg + kTPB - 1) / kTPB : (n_groups + teams_per_block - 1) / teams_per_block;
    const int grid = (int)(want < 1 ? 1 : (want > (1 << 22) ? (1 << 22) : want));
#define AR_ACT_ROW(ADT, XR, W)                                                                                               \
    hipLaunchKernelGGL((k_int_act_row<ADT, XR, BWD, SYM, W>), grid, kTPB, row_lds_bytes, st, dXq, X, Out, scale_out, n_groups, \
                       gs / kEPT, bits, s_dt, q_thresh)
#define AR_ACT_LG(ADT, XR, C)                                                                                                \
    do {                                                                                                                     \
        if constexpr (BWD) {                                                                                                 \
            if (two) hipLaunchKernelGGL((k_int_act<ADT, XR, BWD, C, SYM, 2>), grid, kTPB, 0, st, dXq, X, Out, scale_out,     \
                                        n_groups, bits, s_dt, q_thresh);                                                     \
            else hipLaunchKernelGGL((k_int_act<ADT, XR, BWD, C, SYM, 1>), grid, kTPB, 0, st, dXq, X, Out, scale_out,         \
                                    n_groups, bits, s_dt, q_thresh);                                                         \
        } else hipLaunchKernelGGL((k_int_act<ADT, XR, BWD, C, SYM, 1>), grid, kTPB, 0, st, dXq, X, Out, scale_out, n_groups, \
                                  bits, s_dt, q_thresh);                                                                     \
    } while (0)
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
        } else if (row_waves == 1) AR_ACT_ROW(ADT, XR, 1);                                                                   \
        else if (row_waves == 4) AR_ACT_ROW(ADT, XR, 4);                                                                     \
        else hipLaunchKernelGGL((k_int_act_wave<ADT, XR, BWD, SYM>), grid, kTPB, 0, st, dXq, X, Out, scale_out, n_groups,     \
                                  gs / kEPT, bits, s_dt, q_thresh);                                                          \
    } while (0)
    switch (a_dt) {
        case AR_DT_BF16: if (same16) AR_ACT(AR_DT_BF16, AR_DT_BF16); else AR_ACT(AR_DT_BF16, AR_DT_F32); break;
        case AR_DT_F16: if (same16) AR_ACT(AR_DT_F16, AR_DT_F16); else AR_ACT(AR_DT_F16, AR_DT_F32); break;
        case AR_DT_F32: AR_ACT(AR_DT_F32, AR_DT_F32); break;
        default: return AR_ERR_UNSUPPORTED;
    }
#undef AR_ACT
#undef AR_ACT_LG
#undef AR_ACT_ROW
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
