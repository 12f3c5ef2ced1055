/*
 * ar_oracle.c -- CPU restatement (plain C, scalar loops) of AutoRound's per-block tuning hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under auto_round_amd/ may import, link or call this file;
 * it exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the
 * HIP kernels.  The product path fails loudly when the HIP library is missing -- it never falls
 * back to this code.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (integers, packed words,
 * scales, dV) or sign-for-sign (min/max-scale gradients) against outputs of the reference itself,
 * imported from /root/reference in the build container; the generating script and the committed
 * vectors live in tests/golden/ (make_golden.py, *.npz).  The reference's own known-answer
 * vectors (cast_to_fp4 15-value vector, the E2M1 12-value vector, e8m0/e4m3 constants, nibble
 * packing answers) are part of those fixtures.
 *
 * Each function cites the reference file:line (relative to /root/reference) it restates.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, every fp32 op rounds
 * once, exactly like the reference's eager torch ops).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AR_DT_BF16 0
#define AR_DT_F16 1
#define AR_DT_F32 2

/* ------------------------------------------------------------------------------------------
 * scalar dtype helpers (round-to-nearest-even everywhere, like torch's .to())
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float bf16_bits_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }
static inline uint16_t f32_to_bf16_bits(float f) {
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        /* subnormal: value = m * 2^-24 */
        float v = (float)m * 5.9604644775390625e-08f;
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
static inline uint16_t f32_to_f16_bits(float f) {
    uint32_t u = f2u(f);
    uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);      /* NaN */
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);     /* >= 65520 -> inf (incl. inf) */
    if (a < 0x33000001u) return sign;                            /* <= 2^-25 -> 0 (tie to even) */
    if (a < 0x38800000u) {                                       /* subnormal result */
        /* value * 2^24 rounded to nearest even integer */
        float scaled = u2f(a) * 16777216.0f;                     /* exact */
        float r = nearbyintf(scaled);                            /* RNE under default mode */
        return (uint16_t)(sign | (uint16_t)r);
    }
    uint32_t mant = a & 0x7fffffu, exp = (a >> 23) - 112u;
    uint32_t half = (exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half += 1u; /* carries into exponent OK */
    return (uint16_t)(sign | half);
}

/* OCP e4m3fn (no inf; S.1111.111 = NaN; max 448), torch.float8_e4m3fn semantics: RNE, saturating
 * inputs are assumed pre-clamped to +-448 by the caller (the reference clamps before the cast). */
static inline float e4m3_bits_to_f32(uint8_t b) {
    uint32_t s = (b >> 7) & 1u, e = (b >> 3) & 0xfu, m = b & 7u;
    float v;
    if (e == 0) v = (float)m * 0.001953125f;                      /* m * 2^-9 */
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + (float)m / 8.0f, (int)e - 7);
    return s ? -v : v;
}
static inline uint8_t f32_to_e4m3_bits(float f) {
    uint8_t s = signbit(f) ? 0x80u : 0u;
    float a = fabsf(f);
    if (isnan(a)) return (uint8_t)(s | 0x7fu);
    if (a >= 464.0f) return (uint8_t)(s | 0x7fu);                 /* beyond max+half ulp -> NaN (torch) */
    if (a < 0.015625f) {                                          /* subnormal: multiples of 2^-9 */
        float r = nearbyintf(a * 512.0f);
        return (uint8_t)(s | (uint8_t)r);                         /* r==8 encodes as e=1,m=0: OK */
    }
    int ex; float fr = frexpf(a, &ex);                            /* a = fr*2^ex, fr in [0.5,1) */
    int e = ex - 1;                                               /* a = (2fr) * 2^e, 2fr in [1,2) */
    float m = nearbyintf((fr * 2.0f - 1.0f) * 8.0f);
    if (m == 8.0f) { m = 0.0f; e += 1; }
    int be = e + 7;
    if (be > 15 || (be == 15 && m == 7.0f)) return (uint8_t)(s | 0x7fu);
    return (uint8_t)(s | (uint8_t)(be << 3) | (uint8_t)m);
}

static inline float load_as_f32(const void* p, int64_t i, int dt) {
    if (dt == AR_DT_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
    if (dt == AR_DT_F16) return f16_bits_to_f32(((const uint16_t*)p)[i]);
    return ((const float*)p)[i];
}
static inline void store_from_f32(void* p, int64_t i, int dt, float v) {
    if (dt == AR_DT_BF16) ((uint16_t*)p)[i] = f32_to_bf16_bits(v);
    else if (dt == AR_DT_F16) ((uint16_t*)p)[i] = f32_to_f16_bits(v);
    else ((float*)p)[i] = v;
}
/* round an fp32 value to dtype dt and come back (what a torch op with that output dtype does) */
static inline float rnd(int dt, float v) {
    if (dt == AR_DT_BF16) return bf16_bits_to_f32(f32_to_bf16_bits(v));
    if (dt == AR_DT_F16) return f16_bits_to_f32(f32_to_f16_bits(v));
    return v;
}
/* torch type promotion of two floating dtypes */
static inline int promote(int a, int b) {
    if (a == b) return a;
    return AR_DT_F32; /* bf16 x f16 -> f32; anything x f32 -> f32 */
}
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* round_ste forward value: (x.round() - x).detach() + x  (data_type/utils.py:314-323).  Numerically
 * this is rint(x) exactly, except that a zero result is always +0 (x + (-x) = +0 under RNE), which is
 * visible in the sign bit of a baked zero weight -- so it is evaluated literally. */
static inline float round_ste_value(float y) { volatile float d = nearbyintf(y) - y; return d + y; }

/* ------------------------------------------------------------------------------------------
 * group min/max of the (group-reshaped) weight, clamped at 0
 * reference: auto_round/wrapper.py:154-164 (WrapperLinear._init_tuning_params_and_quant_func)
 * ---------------------------------------------------------------------------------------- */
void oracle_group_minmax(const void* W, int w_dt, int64_t G, int gs, void* wmin, void* wmax) {
    for (int64_t g = 0; g < G; ++g) {
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < gs; ++k) {
            float w = load_as_f32(W, g * gs + k, w_dt);
            if (w < lo) lo = w;
            if (w > hi) hi = w;
        }
        store_from_f32(wmin, g, w_dt, lo > 0.f ? 0.f : lo);
        store_from_f32(wmax, g, w_dt, hi < 0.f ? 0.f : hi);
    }
}

/* ------------------------------------------------------------------------------------------
 * per-group scale / zero-point
 * sym : auto_round/data_type/int.py:221-227 (quant_tensor_sym)
 * asym: auto_round/data_type/int.py:283-293 (quant_tensor_asym)
 * min/max scale bounds clamp: auto_round/wrapper.py:256-259 (WrapperLinear._qdq_weight)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float ms, Ms;        /* clamped min_scale / max_scale */
    float wmin, wmax;    /* float(wmin0), float(wmax0) */
    float a, b;          /* sym: -(wmin*ms), wmax*Ms ; asym: lo=wmin*ms, hi=wmax*Ms */
    int sgn;             /* sym only */
    float s_raw;         /* scale after the cast to scale dtype, before threshold clamp */
    float s;             /* scale after threshold clamp (value of the scale-dtype number) */
    float zp;            /* asym: rint(-lo/s) ; sym: maxq (reported only) */
} group_q_t;

static inline float thresh_in(int s_dt, float t) { return rnd(s_dt, t); }

/* `tensor / python_scalar` (the reference's `(wmax - wmin) / maxq`, data_type/int.py:283, and the `/ maxq` of its backward): torch on
 * the CPU divides; torch's CUDA / HIP kernel multiplies by fl(1 / scalar) ("may lose one bit of precision",
 * ATen/native/cuda/BinaryDivTrueKernel.cu).  Identical when the scalar is a power of two (every symmetric scheme); for the
 * asymmetric ones (maxq = 2^bits - 1) the last bit differs now and then.  mode 0 (default) = CPU semantics, what the CPU-generated
 * golden vectors carry; mode 1 = what the reference computes when it runs ON the GPU, the semantics the HIP kernels follow. */
static int g_scalar_div_mode = 0;
void oracle_set_scalar_div_mode(int gpu) { g_scalar_div_mode = gpu ? 1 : 0; }
int oracle_get_scalar_div_mode(void) { return g_scalar_div_mode; }
static inline float div_py_scalar(float x, float b) { return g_scalar_div_mode ? x * (1.0f / b) : x / b; }

/* d out / d t of the MX element rounding (quant_element, data_type/mxfp.py:49-85) applied to dx4 = d loss / d (rounded element), in
 * autograd's own fp32 arithmetic.  The chain through the rounding itself only multiplies / divides by 2^pe, 2 and +-1 (exact): it
 * hands dx4 through (0 for t == 0, where sign() kills it).  The chain through the private exponent pe = floor_ste(log2|t|).clip(min=0)
 * is live for |t| >= 1: the two `2.0 ** pe` nodes receive dx4 * (q / P) and -(dx4 P) ((t / P) / P); PowBackward multiplies each by
 * P * ln2 (two roundings, then their sum); Log2Backward divides by |t| * ln2; AbsBackward restores the sign.  Equals dx4 * q / t in
 * exact arithmetic (the closed form of rounds 1-2); these are the bits torch produces, on the CPU and on the GPU alike. */
static inline float mx_elem_grad(float dx4, float t, float q) {
    const float LN2F = 0.6931471805599453f;
    if (t == 0.f) return 0.f;
    const float at = fabsf(t);
    if (at < 1.0f) return dx4;
    const float a = (dx4 * q) * LN2F;
    const float b = (dx4 * t) * LN2F;
    const float darg = (a - b) / (at * LN2F);
    return dx4 + (t > 0.f ? darg : -darg);
}

static void group_scale_sym(group_q_t* q, int bits, int s_dt, float q_thresh) {
    const float maxq = (float)(1 << (bits - 1));
    q->a = -(q->wmin * q->ms);
    q->b = q->wmax * q->Ms;
    q->sgn = (q->b < q->a) ? 1 : -1;
    float m = (q->a > q->b) ? q->a : q->b;
    float max_v = (float)q->sgn * m;
    q->s_raw = rnd(s_dt, div_py_scalar(max_v, maxq));
    const float t = thresh_in(s_dt, q_thresh);
    if (q->s_raw < 0.f) q->s = (q->s_raw > -t) ? -t : q->s_raw;   /* clamp(max=-t) */
    else q->s = (q->s_raw < t) ? t : q->s_raw;                    /* clamp(min=t) */
    q->zp = maxq;
}
/* algorithm-extension sym path (sym == 2): the searched per-group init_scale travels in the `wmax` slot,
 * scale = s_dt(init_scale * max_scale), same signed threshold clamp (auto_round/data_type/int.py:201-216). */
static void group_scale_init(group_q_t* q, int bits, int s_dt, float q_thresh) {
    q->a = 0.f;
    q->b = q->wmax * q->Ms;
    q->sgn = 1;
    q->s_raw = rnd(s_dt, q->b);
    const float t = thresh_in(s_dt, q_thresh);
    if (q->s_raw < 0.f) q->s = (q->s_raw > -t) ? -t : q->s_raw;
    else q->s = (q->s_raw < t) ? t : q->s_raw;
    q->zp = (float)(1 << (bits - 1));
}
static void group_scale_asym(group_q_t* q, int bits, int s_dt, float q_thresh) {
    const float maxq = (float)((1 << bits) - 1);
    q->a = q->wmin * q->ms; /* lo */
    q->b = q->wmax * q->Ms; /* hi */
    q->sgn = 1;
    q->s_raw = rnd(s_dt, div_py_scalar(q->b - q->a, maxq));
    const float t = thresh_in(s_dt, q_thresh);
    q->s = (q->s_raw < t) ? t : q->s_raw;
    /* -lo is fp32, scale is s_dt: fp32 / s_dt -> fp32 */
    q->zp = nearbyintf((-q->a) / q->s);
}

/* ------------------------------------------------------------------------------------------
 * INT fake-quant forward (W2/W3/W4/W8, sym "full range" and asym)
 * reference: auto_round/data_type/int.py:165-238 and :241-298; dtype choreography SURVEY App. A.1/A.2
 *   W        [G*gs] in w_dt  (group-reshaped weight, row-major)
 *   V        [G*gs] fp32 rounding offsets (may be NULL == 0)
 *   wmin/wmax[G]    in w_dt  (precomputed, clamped at 0)
 *   min_s/max_s [G] fp32 (NULL == 1.0, no clamp)
 *   Wq       [G*gs] in w_dt  (out)
 *   scale    [G]    in s_dt  (out, may be NULL)
 *   zp       [G]    fp32     (out, may be NULL; sym writes maxq)
 * ---------------------------------------------------------------------------------------- */
void oracle_qdq_int_fwd(const void* W, const float* V, const void* wmin, const void* wmax,
                        const float* min_s, const float* max_s, int64_t G, int gs, int bits, int sym,
                        int w_dt, int s_dt, float q_thresh, float lo_bound, float hi_bound,
                        void* Wq, void* scale, float* zp) {
    const int x_dt = promote(w_dt, s_dt);
    const float maxq_sym = (float)(1 << (bits - 1));
    const float maxq_asym = (float)((1 << bits) - 1);
    for (int64_t g = 0; g < G; ++g) {
        group_q_t q;
        q.ms = min_s ? clampf(min_s[g], lo_bound, hi_bound) : 1.0f;
        q.Ms = max_s ? clampf(max_s[g], lo_bound, hi_bound) : 1.0f;
        q.wmin = load_as_f32(wmin, g, w_dt);
        q.wmax = load_as_f32(wmax, g, w_dt);
        if (sym == 2) group_scale_init(&q, bits, s_dt, q_thresh);
        else if (sym) group_scale_sym(&q, bits, s_dt, q_thresh);
        else group_scale_asym(&q, bits, s_dt, q_thresh);
        if (scale) store_from_f32(scale, g, s_dt, q.s);
        if (zp) zp[g] = q.zp;
        for (int k = 0; k < gs; ++k) {
            const int64_t i = g * gs + k;
            float w = load_as_f32(W, i, w_dt);
            float x = rnd(x_dt, w / q.s);
            float y = x + (V ? V[i] : 0.f);
            float r = round_ste_value(y);
            float out;
            if (sym) {
                float qq = clampf(r, -maxq_sym, maxq_sym - 1.f);
                out = q.s * qq;
            } else {
                float qq = clampf(r + q.zp, 0.f, maxq_asym);
                out = q.s * (qq - q.zp);
            }
            store_from_f32(Wq, i, w_dt, out);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * INT fake-quant backward, restating what torch autograd computes for the forward above, op by
 * op and dtype by dtype.
 * reference: autograd through auto_round/data_type/int.py:165-238 (quant_tensor_sym), :241-298 (quant_tensor_asym) and
 *            round_ste (auto_round/data_type/utils.py:314); the clamp bounds of WrapperLinear._qdq_weight (auto_round/wrapper.py:256-259):
 *   g        = float(dWq)                                     (.to(W.dtype) backward)
 *   e_k      = g_k * s                                        (MulBackward, other side)
 *   inside_k = lo_q <= r_k (+zp) <= hi_q (inclusive)          (ClampBackward)
 *   dV_k     = e_k * inside_k                                 (STE through round; AddBackward)
 *   c1 = s_dt( sum_k g_k * qq_k )                             (MulBackward self side; sum_to then cast)
 *   c2 = s_dt( sum_k -dx_k * ((W_k/s)/s) )                    (DivBackward other side)
 *   asym only: dzp = (-sum_k e_k) + (sum_k dV_k)  (fp32), c3 = s_dt( -dzp * ((-lo/s)/s) )
 *   ds_c = s_dt-add(c1, c2) [ then s_dt-add c3 ]              (grad accumulation in scale dtype)
 *   threshold clamp / where backward, cast to fp32, /maxq, sign, max() routing, * wmin/wmax.
 * Group sums are accumulated in double and rounded once to fp32 (torch's fp32 reduction order is
 * unspecified; this is the order-free value every order rounds towards).
 * Outputs: dV [G*gs] fp32, dmin [G] fp32, dmax [G] fp32 (gradients w.r.t. min_scale/max_scale).
 * ---------------------------------------------------------------------------------------- */
void oracle_qdq_int_bwd(const void* dWq, const void* W, const float* V, const void* wmin,
                        const void* wmax, const float* min_s, const float* max_s, int64_t G, int gs,
                        int bits, int sym, int w_dt, int s_dt, float q_thresh, float lo_bound,
                        float hi_bound, float* dV, float* dmin, float* dmax) {
    const int x_dt = promote(w_dt, s_dt);
    const float maxq = sym ? (float)(1 << (bits - 1)) : (float)((1 << bits) - 1);
    for (int64_t g = 0; g < G; ++g) {
        group_q_t q;
        q.ms = min_s ? clampf(min_s[g], lo_bound, hi_bound) : 1.0f;
        q.Ms = max_s ? clampf(max_s[g], lo_bound, hi_bound) : 1.0f;
        q.wmin = load_as_f32(wmin, g, w_dt);
        q.wmax = load_as_f32(wmax, g, w_dt);
        if (sym == 2) group_scale_init(&q, bits, s_dt, q_thresh);
        else if (sym) group_scale_sym(&q, bits, s_dt, q_thresh);
        else group_scale_asym(&q, bits, s_dt, q_thresh);
        double acc_c1 = 0.0, acc_c2 = 0.0, acc_e = 0.0, acc_dy = 0.0;
        for (int k = 0; k < gs; ++k) {
            const int64_t i = g * gs + k;
            float gk = load_as_f32(dWq, i, w_dt);
            float w = load_as_f32(W, i, w_dt);
            float x = rnd(x_dt, w / q.s);
            float y = x + (V ? V[i] : 0.f);
            float r = round_ste_value(y);
            float qq, inside;
            if (sym) {
                qq = clampf(r, -maxq, maxq - 1.f);
                inside = (r >= -maxq && r <= maxq - 1.f) ? 1.f : 0.f;
            } else {
                float t = r + q.zp;
                qq = clampf(t, 0.f, maxq) - q.zp;
                inside = (t >= 0.f && t <= maxq) ? 1.f : 0.f;
            }
            float e = gk * q.s;          /* grad * self(scale): fp32 */
            float dy = (inside != 0.f) ? e : 0.f;   /* ClampBackward is a where(mask, grad, 0): masked lanes are +0 */
            if (dV) dV[i] = dy;
            acc_c1 += (double)(gk * qq);
            /* DivBackward (other): -grad * ((self/other)/other), evaluated in x_dt */
            float dx = rnd(x_dt, dy);
            float t2 = rnd(x_dt, rnd(x_dt, w / q.s) / q.s);
            acc_c2 += (double)rnd(x_dt, (-dx) * t2);
            acc_e += (double)(-e);
            acc_dy += (double)dy;
        }
        float c1 = rnd(s_dt, (float)acc_c1);
        float c2 = rnd(s_dt, rnd(x_dt, (float)acc_c2));
        float ds_c = rnd(s_dt, c1 + c2);
        float dlo_from_zp = 0.f;
        if (!sym) {
            float dzp = (float)acc_e + (float)acc_dy;
            /* zp = round_ste(u), u = (-lo)/s  (fp32 / s_dt -> fp32) */
            float u_over_s = ((-q.a) / q.s) / q.s;
            float c3 = rnd(s_dt, (-dzp) * u_over_s);
            ds_c = rnd(s_dt, ds_c + c3);
            dlo_from_zp = -(dzp / q.s); /* d(-lo) = dzp / s  ->  dlo = -(...) */
        }
        /* threshold clamp backward */
        const float t = thresh_in(s_dt, q_thresh);
        float ds;
        if (sym) ds = (q.s_raw < 0.f) ? ((q.s_raw <= -t) ? ds_c : 0.f) : ((q.s_raw >= t) ? ds_c : 0.f);
        else ds = (q.s_raw >= t) ? ds_c : 0.f;
        float d32 = div_py_scalar(ds, maxq); /* .to(scale_dtype) backward = cast to fp32, then DivBackward by maxq */
        if (sym == 2) {
            /* scale = (init_scale * max_scale).to(s_dt): d max_scale = float(ds) * init_scale; min_scale is unused */
            if (dmin) dmin[g] = 0.f;
            if (dmax) dmax[g] = ds * q.wmax;
        } else if (sym) {
            float dm = d32 * (float)q.sgn;                       /* max_v = sgn * m */
            float da, db;                                        /* m = maximum(a, b) */
            if (q.a == q.b) { da = dm / 2.f; db = dm / 2.f; }
            else if (q.a > q.b) { da = dm; db = 0.f; }
            else { da = 0.f; db = dm; }
            /* a = -(wmin*ms) -> d ms = (-da) * wmin ; b = wmax*Ms -> d Ms = db * wmax */
            if (dmin) dmin[g] = (-da) * q.wmin;
            if (dmax) dmax[g] = db * q.wmax;
        } else {
            float dhi = d32;
            float dlo = (-d32) + dlo_from_zp;
            if (dmin) dmin[g] = dlo * q.wmin;
            if (dmax) dmax[g] = dhi * q.wmax;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * sign-SGD step: p += (-lr) * sign(g), sign(0)=0, fp32
 * reference: auto_round/algorithms/quantization/sign_round/sign_sgd.py:389 (_single_tensor_sgd)
 * ---------------------------------------------------------------------------------------- */
void oracle_sign_sgd(float* p, const float* g, int64_t n, float lr) {
    const float alpha = -lr;
    for (int64_t i = 0; i < n; ++i) {
        float s = (g[i] > 0.f) ? 1.f : ((g[i] < 0.f) ? -1.f : 0.f);
        p[i] = p[i] + alpha * s;
    }
}

/* ------------------------------------------------------------------------------------------
 * MSE loss forward/backward as the loop uses it:
 *   loss = mean((float(pred)-float(ref))^2)  ; (loss*1000).backward()
 *   dpred = act_dt( (alpha*(p-r)) * gout ), alpha = float(2/N), gout = 1000
 * reference: sign_round/quantizer.py:127-158 (_get_loss), :789-803 (_scale_loss_and_backward);
 *            ATen mse_loss_backward kernel: alpha * (a - b) * c.
 * loss is accumulated in double here (order-free value).
 * ---------------------------------------------------------------------------------------- */
void oracle_mse_fwd_bwd(const void* pred, const void* ref, int64_t n, int act_dt, float gout,
                        float* loss_out, void* dpred, const uint8_t* token_mask, int64_t row_len) {
    /* token_mask (optional): the valid-token mask of sign_round/quantizer.py:142-151 -- loss_func((pred*m).float(),
     * (ref*m).float()): masked rows add 0 to the n-normalised mean and receive a zero gradient. */
    double acc = 0.0;
    const float alpha = (float)(2.0 / (double)n);
    for (int64_t i = 0; i < n; ++i) {
        if (token_mask && !token_mask[i / row_len]) { if (dpred) store_from_f32(dpred, i, act_dt, 0.f); continue; }
        float p = load_as_f32(pred, i, act_dt), r = load_as_f32(ref, i, act_dt);
        float d = p - r;
        acc += (double)(d * d);
        if (dpred) store_from_f32(dpred, i, act_dt, (alpha * d) * gout);
    }
    if (loss_out) *loss_out = (float)(acc / (double)n);
}

/* ------------------------------------------------------------------------------------------
 * INT packers (GPTQ order), re-deriving the integers from the baked weight like the reference:
 *   intw = int32(rint(float(Wq[o,i]) / float(scale[o, i/gs]) + zp))        (fp32 division)
 *   bits in {2,4,8}: qweight[i/P, o] |= intw[o,i] << (bits*(i%P)), P = 32/bits
 *   bits == 3      : 32 consecutive input indices form a 96-bit little-endian stream -> 3 words
 *   scales_t[ig, o] = fp16(scale[o, ig])
 *   qzeros[ig, o/P] packs (zp[o,ig] - zp_off) the same way along o; zp_off = 1 for the
 *   "zp-1" convention of the *_zp packer (both tensor and python-int zeros), 0 for the plain
 *   packer qlinear_torch.py (which stores zp unchanged).
 * reference: auto_round_extension/torch/qlinear_torch_zp.py:93-149 (pack_248_bits), :151-263 (pack_3bits)
 *            auto_round_extension/torch/qlinear_torch.py:110-167, :170-281
 *   Wq [out,in] w_dt ; scale [out, n_groups] s_dt ; zp_tensor [out, n_groups] fp32 or NULL (then zp_scalar)
 *   qweight [in/32*bits, out] int32 ; qzeros [n_groups, out/32*bits] int32 ; scales_t [n_groups,out] fp16 bits
 * ---------------------------------------------------------------------------------------- */
/* Packs 32 consecutive integer values v[0..31] (read with stride vstride) into `bits` 32-bit words
 * written with stride wstride, with the reference's exact arithmetic:
 *   bits in {2,4,8}: word_w = SUM_j ( v[w*P + j] << (bits*j) )  -- a wrapping int32 SUM of UNMASKED values
 *                    (torch: `intweight << order_map` then torch.sum), so an out-of-range value (the
 *                    reference can produce 256 at W8 asym, or -1 for a zero-point of 0 under "zp-1")
 *                    carries/borrows into the neighbouring fields exactly as the reference does;
 *   bits == 3      : 10 values summed at shifts 0,3,..27 | v10<<30 ; (v10>>2)&1 | 10 values at 1,4,..28 |
 *                    v21<<31 ; (v21>>1)&3 | 10 values at 2,5,..29   (pack_3bits). */
static void pack32(const int32_t* v, int64_t vstride, uint32_t* words, int64_t wstride, int bits) {
    if (bits != 3) {
        const int P = 32 / bits;
        for (int w = 0; w < bits; ++w) {
            uint32_t acc = 0;
            for (int j = 0; j < P; ++j) acc += (uint32_t)v[(w * P + j) * vstride] << (bits * j);
            words[w * wstride] = acc;
        }
        return;
    }
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    for (int j = 0; j < 10; ++j) a0 += (uint32_t)v[j * vstride] << (3 * j);
    for (int j = 0; j < 10; ++j) a1 += (uint32_t)v[(11 + j) * vstride] << (3 * j + 1);
    for (int j = 0; j < 10; ++j) a2 += (uint32_t)v[(22 + j) * vstride] << (3 * j + 2);
    const uint32_t v10 = (uint32_t)v[10 * vstride], v21 = (uint32_t)v[21 * vstride];
    words[0] = a0 | (v10 << 30);
    words[wstride] = (((uint32_t)(v[10 * vstride] >> 2)) & 1u) | a1 | (v21 << 31);
    words[2 * wstride] = (((uint32_t)(v[21 * vstride] >> 1)) & 3u) | a2;
}
void oracle_pack_int(const void* Wq, const void* scale, const float* zp_tensor, float zp_scalar,
                     int64_t out_f, int64_t in_f, int gs, int bits, int w_dt, int s_dt, int zp_off,
                     int32_t* qweight, int32_t* qzeros, uint16_t* scales_t) {
    const int64_t n_groups = (in_f + gs - 1) / gs;
    const int64_t zcols = out_f / 32 * bits;
    int32_t* iw = (int32_t*)malloc(sizeof(int32_t) * in_f);
    for (int64_t o = 0; o < out_f; ++o) {
        for (int64_t i = 0; i < in_f; ++i) {
            int64_t ig = i / gs;
            float s = load_as_f32(scale, o * n_groups + ig, s_dt);
            float z = zp_tensor ? zp_tensor[o * n_groups + ig] : zp_scalar;
            float w = load_as_f32(Wq, o * in_f + i, w_dt);
            iw[i] = (int32_t)nearbyintf(w / s + z);
        }
        /* column o of qweight: 32 inputs -> `bits` consecutive rows */
        for (int64_t blk = 0; blk < in_f / 32; ++blk)
            pack32(iw + blk * 32, 1, (uint32_t*)qweight + (blk * bits) * out_f + o, out_f, bits);
        for (int64_t ig = 0; ig < n_groups; ++ig)
            scales_t[ig * out_f + o] = f32_to_f16_bits(load_as_f32(scale, o * n_groups + ig, s_dt));
    }
    free(iw);
    int32_t zv[32];
    for (int64_t ig = 0; ig < n_groups; ++ig) {
        for (int64_t blk = 0; blk < out_f / 32; ++blk) {
            for (int j = 0; j < 32; ++j) {
                int64_t o = blk * 32 + j;
                float z = zp_tensor ? zp_tensor[o * n_groups + ig] : zp_scalar;
                zv[j] = (int32_t)(z - (float)zp_off);   /* `zeros -= 1` happens in float, then .to(int32) */
            }
            pack32(zv, 1, (uint32_t*)qzeros + ig * zcols + blk * bits, 1, bits);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * E2M1 (fp4) helpers
 *  - e2m1_rne: nearest value in {0,.5,1,1.5,2,3,4,6} with ties to the even mantissa, saturating at 6.
 *    Equals mxfp.quant_element(ebits=2, mbits=3, max_norm=6) (data_type/mxfp.py:49-85) and
 *    nvfp.cast_to_fp4 (data_type/nvfp.py:26-39) on |x| <= 6.
 *  - fp4 nibble index: argmin |x| - table, first minimum wins; | signbit<<3
 *    (export/export_to_autoround/qlinear_fp.py:235-265 _pack_fp4_to_uint8)
 * ---------------------------------------------------------------------------------------- */
static inline float cast_to_fp4_ref(float x) {
    /* literal restatement of nvfp.cast_to_fp4 */
    float sign = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    float a = fabsf(x), out;
    if (a < 2.0f) out = nearbyintf(2.0f * a) / 2.0f;
    else if (a < 4.0f) out = nearbyintf(a);
    else out = 2.0f * nearbyintf(a / 2.0f);
    if (out > 6.f) out = 6.f;
    return out * sign;
}
static inline float mx_quant_element_fp4(float t) {
    /* literal restatement of mxfp.quant_element(ebits=2, mbits=3, max_norm=6.0, "even") */
    float a0 = fabsf(t);
    float pe = floorf(log2f(a0 + (t == 0.f ? 1.f : 0.f)));
    if (pe < 0.f) pe = 0.f;                       /* min_exp = -(2^(ebits-1)) + 2 = 0 */
    float x = t / exp2f(pe) * 2.0f;               /* 2^(mbits-2) = 2 */
    float a = fabsf(x);
    /* torch `%` is python-style remainder; it is zero iff fmodf is zero (a-0.5 >= -0.5 here) */
    float mask = (fmodf(a - 0.5f, 2.0f) == 0.f) ? 1.f : 0.f;
    float sign = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    float v = sign * (floorf(a + 0.5f) - mask);
    v = v / 2.0f * exp2f(pe);
    return clampf(v, -6.f, 6.f);
}
void oracle_cast_to_fp4(const float* x, int64_t n, float* out) { for (int64_t i = 0; i < n; ++i) out[i] = cast_to_fp4_ref(x[i]); }
void oracle_mx_quant_element_fp4(const float* x, int64_t n, float* out) { for (int64_t i = 0; i < n; ++i) out[i] = mx_quant_element_fp4(x[i]); }

/* ------------------------------------------------------------------------------------------
 * MXFP4 fake-quant forward (group 32, e8m0 shared exponent)
 * reference: auto_round/data_type/mxfp.py:233-291 (quant_mx), :49-85 (quant_element)
 *   all math in fp32; max_scale [G] fp32 or NULL (==1); init_scale scalar or per group
 *   exp_out [G] = shared_exp cast to w_dt (what the reference returns as "scale")
 * ---------------------------------------------------------------------------------------- */
void oracle_qdq_mxfp4_fwd(const void* W, const float* V, const float* max_s, float init_scale_scalar, const float* init_scale_arr,
                          int64_t G, int gs, int w_dt, float lo_bound, float hi_bound, void* Wq,
                          void* exp_out) {
    for (int64_t g = 0; g < G; ++g) {
        const float init_scale = init_scale_arr ? init_scale_arr[g] : init_scale_scalar;   /* searched per-group coefficient (alg. ext.) */
        float amax = 0.f;
        for (int k = 0; k < gs; ++k) { float a = fabsf(load_as_f32(W, g * gs + k, w_dt)); if (a > amax) amax = a; }
        float Ms = max_s ? clampf(max_s[g], lo_bound, hi_bound) : 1.0f;
        float mv = amax * (init_scale * Ms);
        float se = (mv == 0.f) ? 1.0f : log2f(mv);
        se = floorf(se);
        se = clampf(se - 2.0f, -127.f, 127.f);
        float sc = exp2f(se);
        if (exp_out) store_from_f32(exp_out, g, w_dt, se);
        for (int k = 0; k < gs; ++k) {
            const int64_t i = g * gs + k;
            float t = load_as_f32(W, i, w_dt) / sc + (V ? V[i] : 0.f);
            t = clampf(t, -6.f, 6.f);
            float e = mx_quant_element_fp4(t);
            store_from_f32(Wq, i, w_dt, e * sc);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * NVFP4 fake-quant forward (group 16, e4m3 block scale x fp32 global scale)
 * reference: auto_round/data_type/nvfp.py:67-98 (ref_nvfp4_quant, nv_fp4), :56-64 (calculate_gparam)
 *   get_reciprocal here is the nvfp-local one (nvfp.py:42-48): 0 -> 0 else 1/x
 *   scale_out [G] fp32 (holds e4m3 values)
 * ---------------------------------------------------------------------------------------- */
static inline float recip0(float x) { return x == 0.f ? 0.f : 1.0f / x; }
float oracle_nvfp4_global_scale(const void* W, int64_t n, int w_dt) {
    float amax = 0.f;
    for (int64_t i = 0; i < n; ++i) { float a = fabsf(load_as_f32(W, i, w_dt)); if (a > amax) amax = a; }
    return 448.0f * 6.0f * recip0(amax);
}
void oracle_qdq_nvfp4_fwd(const void* W, const float* V, const float* max_s, float init_scale_scalar, const float* init_scale_arr,
                          float global_scale, int64_t G, int gs, int w_dt, float lo_bound,
                          float hi_bound, void* Wq, float* scale_out) {
    const float r6 = (float)(1.0 / 6.0); /* get_reciprocal(FLOAT4_E2M1_MAX): python double -> fp32 scalar */
    for (int64_t g = 0; g < G; ++g) {
        const float init_scale = init_scale_arr ? init_scale_arr[g] : init_scale_scalar;   /* searched per-group coefficient (alg. ext.) */
        float amax = 0.f;
        for (int k = 0; k < gs; ++k) { float a = fabsf(load_as_f32(W, g * gs + k, w_dt)); if (a > amax) amax = a; }
        float Ms = max_s ? clampf(max_s[g], lo_bound, hi_bound) : 1.0f;
        float coeff = Ms * init_scale;
        float vm = rnd(w_dt, amax); /* torch.max(|x|) in w dtype, then .to(fp32) */
        vm = vm * coeff;
        float sc = global_scale * (vm * r6);
        sc = clampf(sc, -448.f, 448.f);
        sc = e4m3_bits_to_f32(f32_to_e4m3_bits(sc));
        float osc = recip0(sc * recip0(global_scale));
        float rosc = recip0(osc);
        if (scale_out) scale_out[g] = sc;
        for (int k = 0; k < gs; ++k) {
            const int64_t i = g * gs + k;
            float x = load_as_f32(W, i, w_dt) * osc + (V ? V[i] : 0.f);
            x = clampf(x, -6.f, 6.f);
            store_from_f32(Wq, i, w_dt, cast_to_fp4_ref(x) * rosc);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * FP4 nibble packer
 * reference: export/export_to_autoround/qlinear_fp.py:141-193 (pack), :235-265 (_pack_fp4_to_uint8)
 *   mode 0 (MXFP4): t = float(W) / 2^exp[g]   (exp given in w_dt as returned by quant_mx)
 *   mode 1 (NVFP4): t = cast_to_fp4(clamp(float(W) * recip(scale[g]*recip(global)), +-6))
 *   packed [out, in/2] uint8: idx[2k] | idx[2k+1] << 4 ; scale bytes: e8m0 = clamp(exp+127,0,255),
 *   e4m3 = torch .to(float8_e4m3fn)
 * ---------------------------------------------------------------------------------------- */
static const float E2M1_TAB[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static inline uint8_t fp4_nibble(float x, int tab_dt) {
    float a = fabsf(x); int best = 0; float bd = INFINITY;
    for (int j = 0; j < 8; ++j) { float d = rnd(tab_dt, fabsf(rnd(tab_dt, a - E2M1_TAB[j]))); if (d < bd) { bd = d; best = j; } }
    return (uint8_t)(best | (signbit(x) ? 8 : 0));
}
void oracle_pack_fp4(const void* W, const void* scale, float global_scale, int64_t out_f, int64_t in_f,
                     int gs, int mode, int w_dt, uint8_t* packed, uint8_t* scale_bytes) {
    const int64_t n_groups = in_f / gs;
    for (int64_t o = 0; o < out_f; ++o) {
        for (int64_t i = 0; i < in_f; i += 2) {
            uint8_t nib[2];
            for (int j = 0; j < 2; ++j) {
                int64_t gi = o * n_groups + (i + j) / gs;
                float w = load_as_f32(W, o * in_f + i + j, w_dt), t; int tdt;
                if (mode == 0) {
                    /* tensor(w_dt) / (2 ** scales(w_dt)) -> w_dt arithmetic */
                    float e = load_as_f32(scale, gi, w_dt);
                    float p = rnd(w_dt, exp2f(e));
                    t = rnd(w_dt, w / p); tdt = w_dt;
                } else {
                    float s = ((const float*)scale)[gi];
                    float r = recip0(s * recip0(global_scale));
                    t = clampf(w * r, -6.f, 6.f);
                    t = cast_to_fp4_ref(t); tdt = AR_DT_F32;
                }
                nib[j] = fp4_nibble(t, tdt);
            }
            packed[o * (in_f / 2) + i / 2] = (uint8_t)(nib[0] | (nib[1] << 4));
        }
        for (int64_t ig = 0; ig < n_groups; ++ig) {
            int64_t gi = o * n_groups + ig;
            if (mode == 0) {
                float e = load_as_f32(scale, gi, w_dt) + 127.f;
                e = rnd(w_dt, e);
                scale_bytes[gi] = (uint8_t)clampf(e, 0.f, 255.f);
            } else {
                scale_bytes[gi] = f32_to_e4m3_bits(((const float*)scale)[gi]);
            }
        }
    }
}

/* small utilities exported for the dtype-helper tests: torch's own .to(bfloat16/float16/float8_e4m3fn) casts, which every
 * reference function above goes through (known answers: the reference's test/unit/test_cpu/data_type/test_nvfp.py:52-64) */
uint16_t oracle_f32_to_bf16(float f) { return f32_to_bf16_bits(f); }
uint16_t oracle_f32_to_f16(float f) { return f32_to_f16_bits(f); }
float oracle_f16_to_f32(uint16_t h) { return f16_bits_to_f32(h); }
uint8_t oracle_f32_to_e4m3(float f) { return f32_to_e4m3_bits(f); }
float oracle_e4m3_to_f32(uint8_t b) { return e4m3_bits_to_f32(b); }

/* ------------------------------------------------------------------------------------------
 * FP4 fake-quant backward (what torch autograd computes for oracle_qdq_mxfp4_fwd / _nvfp4_fwd)
 * reference: autograd through auto_round/data_type/mxfp.py:233-291 (+ quant_element :49-85) and
 *            auto_round/data_type/nvfp.py:67-98 (+ cast_to_fp4 :26-39).
 * Closed forms (derivation in DESIGN.md section 3, "fp4 backward"):
 *   MXFP4  t_pre = W/sc + V, t = clamp(t_pre,+-6), q = e2m1(t)
 *          dq/dt = 0 if t == 0 ; 1 if |t| < 1 (private exponent clipped at 0, no grad through it) ;
 *                  q/t if |t| >= 1 (the STE through floor(log2|t|) cancels the direct path and leaves q/t)
 *          dV = g*sc * dq/dt * [|t_pre| <= 6]
 *          dsc = sum g*q - sum dV*((W/sc)/sc) ; sc = 2^(floor_ste(log2 m) - 2), m = amax*(init*Ms)
 *          dMs = ((dsc * (sc*ln2)) / (m*ln2)) * amax * init            (0 when m == 0)
 *   NVFP4  x_pre = W*osc + V, x = clamp(x_pre,+-6), q = cast_to_fp4(x), out = q*ro, ro = 1/osc, osc = 1/r,
 *          r = s/gs, s = e4m3_ste(clamp(gs*(vm/6), +-448)), vm = amax*(Ms*init)
 *          dq/dx = 0 if x == 0 else 1 ; dV = g*ro * dq/dx * [|x_pre| <= 6]
 *          dosc = sum dV*W - (sum g*q)*(ro*ro) ; dr = -dosc*(osc*osc) ; ds = dr*(1/gs) ;
 *          dMs = (((ds*gs) * (1/6)) * amax) * init , masked by the +-448 clamp and the zero guards
 * Group sums in double (order-free).  Only sign(dV) and sign(dMs) reach SignSGD.
 * ---------------------------------------------------------------------------------------- */
void oracle_qdq_fp4_bwd(const void* dXq, const void* W, const float* V, const float* max_s, float init_scale_scalar, const float* init_scale_arr,
                        float global_scale, int64_t G, int gs, int mode, int w_dt, float lo_bound,
                        float hi_bound, float* dV, float* dmax) {
    const float LN2 = 0.6931471805599453f;
    const float r6 = (float)(1.0 / 6.0);
    for (int64_t g = 0; g < G; ++g) {
        const float init_scale = init_scale_arr ? init_scale_arr[g] : init_scale_scalar;   /* searched per-group coefficient (alg. ext.) */
        float amax = 0.f;
        for (int k = 0; k < gs; ++k) { float a = fabsf(load_as_f32(W, g * gs + k, w_dt)); if (a > amax) amax = a; }
        const float Ms = max_s ? clampf(max_s[g], lo_bound, hi_bound) : 1.0f;
        double s_gq = 0.0, s_dvw = 0.0;
        if (mode == 0) {
            const float m = amax * (init_scale * Ms);
            float se = (m == 0.f) ? 1.0f : log2f(m);
            const float se_un = floorf(se) - 2.0f;
            se = clampf(se_un, -127.f, 127.f);
            const float sc = exp2f(se);
            for (int k = 0; k < gs; ++k) {
                const int64_t i = g * gs + k;
                const float gk = load_as_f32(dXq, i, w_dt), w = load_as_f32(W, i, w_dt);
                const float ws = w / sc;
                const float tp = ws + (V ? V[i] : 0.f);
                const float t = clampf(tp, -6.f, 6.f);
                const float q = mx_quant_element_fp4(t);
                const float inside = (tp >= -6.f && tp <= 6.f) ? 1.f : 0.f;
                const float dv = inside != 0.f ? mx_elem_grad(gk * sc, t, q) : 0.f;
                if (dV) dV[i] = dv;
                s_gq += (double)(gk * q);
                s_dvw += (double)(dv * (ws / sc));
            }
            if (dmax) {
                const float dsc = (float)s_gq - (float)s_dvw;
                const int pass = (se_un >= -127.f && se_un <= 127.f);
                dmax[g] = (m == 0.f || !pass) ? 0.f : ((dsc * (sc * LN2)) / (m * LN2)) * amax * init_scale;
            }
        } else {
            const float vm = rnd(w_dt, amax) * (Ms * init_scale);
            const float s_pre = global_scale * (vm * r6);
            const float s_c = clampf(s_pre, -448.f, 448.f);
            const float s = e4m3_bits_to_f32(f32_to_e4m3_bits(s_c));
            const float rg = recip0(global_scale);
            const float r = s * rg;
            const float osc = recip0(r), ro = recip0(osc);
            for (int k = 0; k < gs; ++k) {
                const int64_t i = g * gs + k;
                const float gk = load_as_f32(dXq, i, w_dt), w = load_as_f32(W, i, w_dt);
                const float xp = w * osc + (V ? V[i] : 0.f);
                const float x = clampf(xp, -6.f, 6.f);
                const float q = cast_to_fp4_ref(x);
                const float d = (x == 0.f) ? 0.f : 1.f;
                const float inside = (xp >= -6.f && xp <= 6.f) ? 1.f : 0.f;
                const float dv = (gk * ro) * d * inside;
                if (dV) dV[i] = dv;
                s_gq += (double)(gk * q);
                s_dvw += (double)(dv * w);
            }
            if (dmax) {
                float dosc = (osc == 0.f) ? 0.f : ((float)s_dvw - (float)s_gq * (ro * ro));
                float dr = (r == 0.f) ? 0.f : -dosc * (osc * osc);
                float ds = dr * rg;
                if (!(s_pre >= -448.f && s_pre <= 448.f)) ds = 0.f;
                dmax[g] = (((ds * global_scale) * r6) * rnd(w_dt, amax)) * init_scale;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Activation fp4 fake-quant backward w.r.t. the INPUT (dynamic per-group scale from the data):
 *   WrapperLinear._qdq_act (wrapper.py:295-321) -> quant_mx(x, v=0, max_scale=1) | nv_fp4_with_static_gs(x, tensor_max)
 * Two paths reach x: the direct one (through x/scale) and the one through the group max (the scale is a function
 * of max|x|; torch.max(dim) routes that gradient to the FIRST index attaining the max, times sign(x)).
 *   MXFP4 (all fp32 after x.float()):  dx = x_dt( dt_pre/sc + [k == k*] * dm * sign(x_k) )
 *   NVFP4 (max taken in the input dtype): dx = x_dt( x_dt(dxp*osc) + [k == k*] * x_dt(dvm) * sign(x_k) )
 * with dt_pre, dsc, dm, dxp, dosc, dvm as in oracle_qdq_fp4_bwd (amax -> max|x|, Ms = init = 1).
 * All-zero groups: the reference produces NaN at k* (0*inf through the unselected where branch); here 0.
 * ---------------------------------------------------------------------------------------- */
void oracle_fp4_act_bwd(const void* dXq, const void* X, float global_scale, int64_t G, int gs, int mode, int x_dt,
                        void* dX) {
    const float LN2 = 0.6931471805599453f;
    const float r6 = (float)(1.0 / 6.0);
    float* tmp = (float*)malloc(sizeof(float) * gs);
    for (int64_t g = 0; g < G; ++g) {
        float amax = -1.f; int kstar = 0;
        for (int k = 0; k < gs; ++k) { float a = fabsf(load_as_f32(X, g * gs + k, x_dt)); if (a > amax) { amax = a; kstar = k; } }
        double s_gq = 0.0, s_dvw = 0.0;
        float extra = 0.f;
        if (mode == 0) {
            const float m = amax;
            float se = (m == 0.f) ? 1.0f : log2f(m);
            const float se_un = floorf(se) - 2.0f;
            const float sc = exp2f(clampf(se_un, -127.f, 127.f));
            for (int k = 0; k < gs; ++k) {
                const int64_t i = g * gs + k;
                const float gk = load_as_f32(dXq, i, x_dt), x = load_as_f32(X, i, x_dt);
                const float ws = x / sc;
                const float t = clampf(ws, -6.f, 6.f);
                const float q = mx_quant_element_fp4(t);
                const float inside = (ws >= -6.f && ws <= 6.f) ? 1.f : 0.f;
                const float dtp = inside != 0.f ? mx_elem_grad(gk * sc, t, q) : 0.f;
                tmp[k] = dtp / sc;
                s_gq += (double)(gk * q);
                s_dvw += (double)(dtp * (ws / sc));
            }
            const float dsc = (float)s_gq - (float)s_dvw;
            const int pass = (se_un >= -127.f && se_un <= 127.f);
            extra = (m == 0.f || !pass) ? 0.f : (dsc * (sc * LN2)) / (m * LN2);
            for (int k = 0; k < gs; ++k) {
                const float x = load_as_f32(X, g * gs + k, x_dt);
                const float sg = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
                store_from_f32(dX, g * gs + k, x_dt, tmp[k] + ((k == kstar) ? extra * sg : 0.f));
            }
        } else {
            const float vm = amax;
            const float s_pre = global_scale * (vm * r6);
            const float s = e4m3_bits_to_f32(f32_to_e4m3_bits(clampf(s_pre, -448.f, 448.f)));
            const float rg = recip0(global_scale);
            const float r = s * rg;
            const float osc = recip0(r), ro = recip0(osc);
            for (int k = 0; k < gs; ++k) {
                const int64_t i = g * gs + k;
                const float gk = load_as_f32(dXq, i, x_dt), x = load_as_f32(X, i, x_dt);
                const float xp = x * osc;
                const float xc = clampf(xp, -6.f, 6.f);
                const float q = cast_to_fp4_ref(xc);
                const float inside = (xp >= -6.f && xp <= 6.f) ? 1.f : 0.f;
                const float dxp = (xc == 0.f) ? 0.f : (gk * ro) * inside;
                tmp[k] = dxp * osc;
                s_gq += (double)(gk * q);
                s_dvw += (double)(dxp * x);
            }
            float dosc = (osc == 0.f) ? 0.f : ((float)s_dvw - (float)s_gq * (ro * ro));
            float dr = (r == 0.f) ? 0.f : -dosc * (osc * osc);
            float ds = dr * rg;
            if (!(s_pre >= -448.f && s_pre <= 448.f)) ds = 0.f;
            extra = (ds * global_scale) * r6;
            for (int k = 0; k < gs; ++k) {
                const float x = load_as_f32(X, g * gs + k, x_dt);
                const float sg = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
                const float direct = rnd(x_dt, tmp[k]);
                const float viamax = (k == kstar) ? rnd(x_dt, extra) * sg : 0.f;
                store_from_f32(dX, g * gs + k, x_dt, direct + viamax);
            }
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------
 * AWQ "GEMM" packer (4-bit): qweight [in, out/8] with eight consecutive output channels per word at nibble positions
 * {0,4,1,5,2,6,3,7}; qzeros [in/gs, out/8] likewise (zero point stored unchanged); scales [in/gs, out] fp16.
 * reference: auto_round/export/export_to_awq/utils.py:196-274 (WQLinear_GEMM.from_linear): `intweight << new_order_map`
 * then torch.sum -> wrapping int32 SUM of unmasked values.
 * ---------------------------------------------------------------------------------------- */
void oracle_pack_awq(const void* Wq, const void* scale, const float* zp_tensor, float zp_scalar, int64_t out_f,
                     int64_t in_f, int gs, int w_dt, int s_dt, int32_t* qweight, int32_t* qzeros, uint16_t* scales_t) {
    static const int order[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    const int64_t n_groups = in_f / gs, words = out_f / 8;
    for (int64_t i = 0; i < in_f; ++i)
        for (int64_t w = 0; w < words; ++w) {
            uint32_t acc = 0;
            for (int j = 0; j < 8; ++j) {
                int64_t o = w * 8 + j, gi = o * n_groups + i / gs;
                float s = load_as_f32(scale, gi, s_dt);
                float z = zp_tensor ? zp_tensor[gi] : zp_scalar;
                int32_t v = (int32_t)nearbyintf(load_as_f32(Wq, o * in_f + i, w_dt) / s + z);
                acc += (uint32_t)v << (4 * order[j]);
            }
            qweight[i * words + w] = (int32_t)acc;
        }
    for (int64_t ig = 0; ig < n_groups; ++ig) {
        for (int64_t w = 0; w < words; ++w) {
            uint32_t acc = 0;
            for (int j = 0; j < 8; ++j) {
                int64_t o = w * 8 + j;
                float z = zp_tensor ? zp_tensor[o * n_groups + ig] : zp_scalar;
                acc += (uint32_t)(int32_t)z << (4 * order[j]);
            }
            qzeros[ig * words + w] = (int32_t)acc;
        }
        for (int64_t o = 0; o < out_f; ++o) scales_t[ig * out_f + o] = f32_to_f16_bits(load_as_f32(scale, o * n_groups + ig, s_dt));
    }
}

/* ------------------------------------------------------------------------------------------
 * fp4 init-scale search (algorithm extension)
 * reference: search_mx_scale (auto_round/data_type/mxfp.py:102-169) and search_nvfp4_scale
 *            (auto_round/data_type/nvfp.py:328-386).  For each group: loss_c = sum_k (qdq_c(x_k) - x_k)^2 * qw_k for the
 * candidates in order; the first candidate with the strictly smallest loss wins.  fp32 arithmetic, sequential fp32 sum
 * (torch's reduction order is unspecified; near-ties can resolve differently).  qw_row: [in_pad] per-input-channel
 * importance or NULL (== 1); element k of group g uses qw_row[(g % groups_per_row)*gs + k].
 * ---------------------------------------------------------------------------------------- */
void oracle_search_fp4_scale(const void* X, const float* qw_row, int64_t groups_per_row, float global_scale,
                             const float* cand, int n_cand, int64_t G, int gs, int mode, int x_dt, float* best_out) {
    const float r6 = (float)(1.0 / 6.0);
    for (int64_t g = 0; g < G; ++g) {
        float amax = 0.f;
        for (int k = 0; k < gs; ++k) { float a = fabsf(load_as_f32(X, g * gs + k, x_dt)); if (a > amax) amax = a; }
        float best = 0.f, best_c = 1.0f;
        for (int ci = 0; ci < n_cand; ++ci) {
            const float coeff = cand[ci];
            float sc = 1.f, rsc = 1.f;
            if (mode == 0) {
                const float mv = amax * coeff;
                float se = (mv == 0.f) ? 1.0f : log2f(mv);
                se = clampf(floorf(se) - 2.0f, -127.f, 127.f);
                sc = exp2f(se);
            } else {
                const float vm = amax * coeff;
                float s = clampf(global_scale * (vm * r6), -448.f, 448.f);
                s = e4m3_bits_to_f32(f32_to_e4m3_bits(s));
                sc = recip0(s * recip0(global_scale));
                rsc = recip0(sc);
            }
            float loss = 0.f;
            for (int k = 0; k < gs; ++k) {
                const float x = load_as_f32(X, g * gs + k, x_dt);
                float q;
                if (mode == 0) q = mx_quant_element_fp4(clampf(x / sc, -6.f, 6.f)) * sc;
                else q = cast_to_fp4_ref(clampf(x * sc, -6.f, 6.f)) * rsc;
                const float d = q - x;
                const float w = qw_row ? qw_row[(g % groups_per_row) * gs + k] : 1.0f;
                loss += (d * d) * w;
            }
            if (ci == 0 || loss < best) { best = loss; best_c = coeff; }
        }
        best_out[g] = best_c;
    }
}

/* ------------------------------------------------------------------------------------------
 * outlier-suppressed MSE (algorithm extension)
 * reference: SignRoundV2Quantizer._get_loss (sign_roundv2/quantizer.py:362-399): drop the topk largest |pred-ref|
 * (ranked on the activation-dtype difference), mean over ALL n of ((|p-r| in fp32) * token_mask * keep)^2.
 * Exactly topk elements are dropped: all above the k-th largest 16-bit magnitude, then the lowest-index ones tied
 * with it (torch.topk picks an unspecified subset of the ties).
 * ---------------------------------------------------------------------------------------- */
static int cmp_u16_desc(const void* a, const void* b) { return (int)(*(const uint16_t*)b) - (int)(*(const uint16_t*)a); }
void oracle_outlier_mse_fwd_bwd(const void* pred, const void* ref, int64_t n, int act_dt, float gout, int64_t topk,
                                const uint8_t* token_mask, int64_t row_len, float* loss_out, void* dpred, int64_t* n_dropped) {
    uint16_t* keys = (uint16_t*)malloc(sizeof(uint16_t) * n);
    uint16_t* sorted = (uint16_t*)malloc(sizeof(uint16_t) * n);
    for (int64_t i = 0; i < n; ++i) {
        float d = load_as_f32(pred, i, act_dt) - load_as_f32(ref, i, act_dt);
        uint16_t k = (act_dt == AR_DT_BF16 ? f32_to_bf16_bits(d) : f32_to_f16_bits(d)) & 0x7fffu;
        keys[i] = k; sorted[i] = k;
    }
    qsort(sorted, n, sizeof(uint16_t), cmp_u16_desc);
    const uint16_t thr = sorted[topk - 1];
    double acc = 0.0;
    const float alpha = (float)(2.0 / (double)n);
    int64_t dropped = 0, need = topk;
    for (int64_t i = 0; i < n; ++i) need -= keys[i] > thr;
    for (int64_t i = 0; i < n; ++i) {
        int keep = keys[i] <= thr;
        if (keys[i] == thr && need > 0) { keep = 0; --need; }
        if (!keep) ++dropped;
        if (token_mask && !token_mask[i / row_len]) keep = 0;
        float d = keep ? load_as_f32(pred, i, act_dt) - load_as_f32(ref, i, act_dt) : 0.f;
        acc += (double)(d * d);
        if (dpred) store_from_f32(dpred, i, act_dt, keep ? (alpha * d) * gout : 0.f);
    }
    if (loss_out) *loss_out = (float)(acc / (double)n);
    if (n_dropped) *n_dropped = dropped;
    free(keys); free(sorted);
}

/* ------------------------------------------------------------------------------------------
 * int-sym init-scale search of the algorithm extension
 * reference: search_scales (auto_round/data_type/int.py:24-86) + the threshold clamp of search_int
 *            (auto_round/data_type/utils.py:203-209).  Everything runs in the weight dtype (torch op by op):
 *   gmax   = the first element with the largest |x| (signed)
 *   isc_c  = wdt( (-c) * recip(gmax) ),  sc_c = recip(isc_c),  recip(t) = |t| >= eps ? wdt(1/t) : 0  (eps 1e-5 f16, else 1e-30)
 *   L      = clamp(rint(wdt(isc_c * x)), -nmax, nmax-1)
 *   loss_c = sum_k f32( wdt( wdt(sc_c * L) - x ) )^2 * qw_k      (sequential fp32 sum; torch's order is unspecified)
 * candidate 0 is c = nmax; a later candidate wins only with a strictly smaller loss.  out = clamped scale in wdt.
 * ---------------------------------------------------------------------------------------- */
static inline float recip_dt(float t, int dt) {
    const float eps = dt == AR_DT_F16 ? 1e-5f : 1e-30f;
    return fabsf(t) >= rnd(dt, eps) ? rnd(dt, 1.0f / t) : 0.f;
}
void oracle_search_int_scale(const void* X, const float* qw_row, int64_t groups_per_row, const float* cand, int n_cand,
                             int64_t G, int gs, int bits, int x_dt, float q_thresh, void* out_raw, void* out_init) {
    const float nmax = (float)(1 << (bits - 1));
    const float th = rnd(x_dt, q_thresh);
    for (int64_t g = 0; g < G; ++g) {
        float gmax = 0.f, amax = -1.f;
        for (int k = 0; k < gs; ++k) {
            const float x = load_as_f32(X, g * gs + k, x_dt);
            if (fabsf(x) > amax) { amax = fabsf(x); gmax = x; }
        }
        const float rg = recip_dt(gmax, x_dt);
        float best = 0.f, best_s = 0.f;
        for (int ci = 0; ci < n_cand; ++ci) {
            const float isc = rnd(x_dt, (-cand[ci]) * rg);
            const float sc = recip_dt(isc, x_dt);
            float loss = 0.f;
            for (int k = 0; k < gs; ++k) {
                const float x = load_as_f32(X, g * gs + k, x_dt);
                const float L = clampf(nearbyintf(rnd(x_dt, isc * x)), -nmax, nmax - 1.f);
                const float e = rnd(x_dt, rnd(x_dt, sc * L) - x);
                const float w = qw_row ? qw_row[(g % groups_per_row) * gs + k] : 1.0f;
                loss += (e * e) * w;
            }
            if (ci == 0 || loss < best) { best = loss; best_s = sc; }
        }
        if (out_raw) store_from_f32(out_raw, g, x_dt, best_s);
        const float cl = best_s < 0.f ? (best_s > -th ? -th : best_s) : (best_s < th ? th : best_s);
        if (out_init) store_from_f32(out_init, g, x_dt, cl);
    }
}

/* ------------------------------------------------------------------------------------------
 * dynamic symmetric INT activation fake-quant (W4A8-style schemes), forward and input gradient
 * reference: quant_tensor_sym (auto_round/data_type/int.py:165-238) as called by WrapperLinear._qdq_act
 *            (auto_round/wrapper.py:295-321) with v = 0, tensor_min/max = None and the wrapper's non-tunable 0-dim
 *            act_min_scale / act_max_scale (= 1.0).  A 0-dim fp32 tensor does not promote a 16-bit tensor, so -- unlike
 *            the weight path -- the range arithmetic (wmin, wmax, max_v, max_v / maxq) stays in the ACTIVATION dtype.
 * per group:  wmin = min(min x, 0), wmax = max(max x, 0); a = -wmin, b = wmax; sgn = b < a ? +1 : -1
 *             s = thresh_clamp( s_dt( a_dt( sgn*max(a,b) / maxq ) ) );  out = a_dt( s * clamp(round_ste(x/s), -maxq, maxq-1) )
 * backward (autograd mirrored):  dy_k = [inside] (g_k * s);  dx_k = a_dt( dy_k / s )
 *             ds = s_dt(c1 + c2) as for weights, thresh mask;  d = a_dt( a_dt(ds) / maxq ) * sgn, routed by max(a,b)
 *             (a == b: half each) to wmin (as -d, if min x <= 0) / wmax (if max x >= 0) and scattered to the FIRST
 *             arg-min / arg-max element of the group (an a_dt addition).
 * ---------------------------------------------------------------------------------------- */
typedef struct { float s, s_raw, a, b, sgn; int imin, imax; float xmin, xmax; } actq_t;
static void act_group(const void* X, int64_t base, int gs, int a_dt, int s_dt, int bits, float q_thresh, actq_t* q) {
    float mn = load_as_f32(X, base, a_dt), mx = mn;
    q->imin = 0; q->imax = 0;
    for (int k = 1; k < gs; ++k) {
        const float x = load_as_f32(X, base + k, a_dt);
        if (x < mn) { mn = x; q->imin = k; }
        if (x > mx) { mx = x; q->imax = k; }
    }
    q->xmin = mn; q->xmax = mx;
    const float wmin = mn < 0.f ? mn : 0.f, wmax = mx > 0.f ? mx : 0.f;
    const float maxq = (float)(1 << (bits - 1));
    q->a = -wmin; q->b = wmax;
    q->sgn = (q->b < q->a) ? 1.f : -1.f;
    const float m = (q->a > q->b) ? q->a : q->b;
    q->s_raw = rnd(s_dt, rnd(a_dt, (q->sgn * m) / maxq));
    const float t = rnd(s_dt, q_thresh);
    if (q->s_raw < 0.f) q->s = (q->s_raw > -t) ? -t : q->s_raw;
    else q->s = (q->s_raw < t) ? t : q->s_raw;
}
void oracle_int_act_fwd(const void* X, int64_t G, int gs, int bits, int a_dt, int s_dt, float q_thresh, void* Xq, void* scale) {
    const int x_dt = promote(a_dt, s_dt);
    const float maxq = (float)(1 << (bits - 1));
    for (int64_t g = 0; g < G; ++g) {
        actq_t q;
        act_group(X, g * gs, gs, a_dt, s_dt, bits, q_thresh, &q);
        if (scale) store_from_f32(scale, g, s_dt, q.s);
        for (int k = 0; k < gs; ++k) {
            const float x = load_as_f32(X, g * gs + k, a_dt);
            const float r = round_ste_value(rnd(x_dt, x / q.s) + 0.f);
            store_from_f32(Xq, g * gs + k, a_dt, q.s * clampf(r, -maxq, maxq - 1.f));
        }
    }
}
void oracle_int_act_bwd(const void* dXq, const void* X, int64_t G, int gs, int bits, int a_dt, int s_dt, float q_thresh,
                        void* dX) {
    const int x_dt = promote(a_dt, s_dt);
    const float maxq = (float)(1 << (bits - 1));
    for (int64_t g = 0; g < G; ++g) {
        actq_t q;
        act_group(X, g * gs, gs, a_dt, s_dt, bits, q_thresh, &q);
        double acc1 = 0.0, acc2 = 0.0;
        for (int k = 0; k < gs; ++k) {
            const int64_t i = g * gs + k;
            const float gk = load_as_f32(dXq, i, a_dt), x = load_as_f32(X, i, a_dt);
            const float xs = rnd(x_dt, x / q.s);
            const float r = round_ste_value(xs + 0.f);
            const float qq = clampf(r, -maxq, maxq - 1.f);
            const int inside = (r >= -maxq && r <= maxq - 1.f);
            const float e = rnd(x_dt, gk * q.s);                 /* grad wrt q, in the product dtype */
            const float dy = inside ? e : 0.f;
            /* DivBackward (self), the cast back to a_dt, then autograd's accumulation with the dense (zero-filled) min / max
             * scatter gradients: x + (+0) turns a -0 (masked +0 divided by a negative scale) into +0 */
            volatile float zero = 0.f;
            store_from_f32(dX, i, a_dt, rnd(a_dt, rnd(x_dt, dy / q.s)) + zero);
            acc1 += (double)rnd(x_dt, gk * qq);
            acc2 += (double)rnd(x_dt, (-dy) * rnd(x_dt, xs / q.s));
        }
        const float c1 = rnd(s_dt, rnd(x_dt, (float)acc1));
        const float c2 = rnd(s_dt, rnd(x_dt, (float)acc2));
        const float ds_c = rnd(s_dt, c1 + c2);
        const float t = rnd(s_dt, q_thresh);
        const float ds = (q.s_raw < 0.f) ? ((q.s_raw <= -t) ? ds_c : 0.f) : ((q.s_raw >= t) ? ds_c : 0.f);
        const float dm = rnd(a_dt, rnd(a_dt, ds) / maxq) * q.sgn;
        float da, db;
        if (q.a == q.b) { da = rnd(a_dt, dm / 2.f); db = da; }
        else if (q.a > q.b) { da = dm; db = 0.f; }
        else { da = 0.f; db = dm; }
        const float dmin = (q.xmin <= 0.f) ? -da : 0.f;
        const float dmax = (q.xmax >= 0.f) ? db : 0.f;
        const int64_t i0 = g * gs + q.imin, i1 = g * gs + q.imax;
        store_from_f32(dX, i0, a_dt, load_as_f32(dX, i0, a_dt) + dmin);
        store_from_f32(dX, i1, a_dt, load_as_f32(dX, i1, a_dt) + dmax);
    }
}

/* ------------------------------------------------------------------------------------------
 * dynamic ASYMMETRIC INT activation fake-quant, forward and input gradient
 * reference: quant_tensor_asym (auto_round/data_type/int.py:241-298) under WrapperLinear._qdq_act, v = 0, dynamic range,
 *            0-dim act_min_scale / act_max_scale.  Range arithmetic in the activation dtype, zero point in fp32:
 *   wmin = min(min x, 0), wmax = max(max x, 0);  s = max( s_dt( a_dt( a_dt(wmax - wmin) / maxq ) ), t );  zp = rint( (-wmin) / s )
 *   out = a_dt( s * (clamp(round_ste(x/s) + zp, 0, maxq) - zp) )
 * backward:  e_k = g_k*s, dy_k = [inside] e_k, dx_k = a_dt(dy_k / s);  dzp = sum(dy) - sum(e);
 *   ds = s_dt( s_dt(c1 + c2) + c3 ), c3 = s_dt( -dzp * ((-wmin/s)/s) ), thresh mask;  d = a_dt( a_dt(ds) / maxq )
 *   d wmax = d;  d wmin = a_dt( (-d) + (-a_dt(dzp / s)) );  clamp masks (min x <= 0, max x >= 0); scatter to first arg-min / arg-max.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float s, s_raw, wmin, wmax, zp, xmin, xmax; int imin, imax; } actqa_t;
static void act_group_asym(const void* X, int64_t base, int gs, int a_dt, int s_dt, int bits, float q_thresh, actqa_t* q) {
    float mn = load_as_f32(X, base, a_dt), mx = mn;
    q->imin = 0; q->imax = 0;
    for (int k = 1; k < gs; ++k) {
        const float x = load_as_f32(X, base + k, a_dt);
        if (x < mn) { mn = x; q->imin = k; }
        if (x > mx) { mx = x; q->imax = k; }
    }
    q->xmin = mn; q->xmax = mx;
    q->wmin = mn < 0.f ? mn : 0.f; q->wmax = mx > 0.f ? mx : 0.f;
    const float maxq = (float)((1 << bits) - 1);
    q->s_raw = rnd(s_dt, rnd(a_dt, rnd(a_dt, q->wmax - q->wmin) / maxq));
    const float t = rnd(s_dt, q_thresh);
    q->s = q->s_raw < t ? t : q->s_raw;
    q->zp = nearbyintf((-q->wmin) / q->s);
}
void oracle_int_act_asym_fwd(const void* X, int64_t G, int gs, int bits, int a_dt, int s_dt, float q_thresh, void* Xq, void* scale,
                             float* zp) {
    const int x_dt = promote(a_dt, s_dt);
    const float maxq = (float)((1 << bits) - 1);
    for (int64_t g = 0; g < G; ++g) {
        actqa_t q;
        act_group_asym(X, g * gs, gs, a_dt, s_dt, bits, q_thresh, &q);
        if (scale) store_from_f32(scale, g, s_dt, q.s);
        if (zp) zp[g] = q.zp;
        for (int k = 0; k < gs; ++k) {
            const float x = load_as_f32(X, g * gs + k, a_dt);
            const float r = round_ste_value(rnd(x_dt, x / q.s) + 0.f);
            store_from_f32(Xq, g * gs + k, a_dt, q.s * (clampf(r + q.zp, 0.f, maxq) - q.zp));
        }
    }
}
void oracle_int_act_asym_bwd(const void* dXq, const void* X, int64_t G, int gs, int bits, int a_dt, int s_dt, float q_thresh,
                             void* dX) {
    const int x_dt = promote(a_dt, s_dt);
    const float maxq = (float)((1 << bits) - 1);
    for (int64_t g = 0; g < G; ++g) {
        actqa_t q;
        act_group_asym(X, g * gs, gs, a_dt, s_dt, bits, q_thresh, &q);
        double acc1 = 0.0, acc2 = 0.0, acc_e = 0.0, acc_dy = 0.0;
        for (int k = 0; k < gs; ++k) {
            const int64_t i = g * gs + k;
            const float gk = load_as_f32(dXq, i, a_dt), x = load_as_f32(X, i, a_dt);
            const float xs = rnd(x_dt, x / q.s);
            const float r = round_ste_value(xs + 0.f);
            const float tq = r + q.zp;
            const float qq = clampf(tq, 0.f, maxq) - q.zp;
            const int inside = (tq >= 0.f && tq <= maxq);
            const float e = rnd(x_dt, gk * q.s);
            const float dy = inside ? e : 0.f;
            volatile float zero = 0.f;
            store_from_f32(dX, i, a_dt, rnd(a_dt, rnd(x_dt, dy / q.s)) + zero);
            acc1 += (double)rnd(x_dt, gk * qq);
            acc2 += (double)rnd(x_dt, (-dy) * rnd(x_dt, xs / q.s));
            acc_e += (double)(-e);
            acc_dy += (double)dy;
        }
        const float c1 = rnd(s_dt, rnd(x_dt, (float)acc1));
        const float c2 = rnd(s_dt, rnd(x_dt, (float)acc2));
        float ds_c = rnd(s_dt, c1 + c2);
        const float dzp = (float)acc_e + (float)acc_dy;
        const float u_over_s = ((-q.wmin) / q.s) / q.s;
        const float c3 = rnd(s_dt, (-dzp) * u_over_s);
        ds_c = rnd(s_dt, ds_c + c3);
        const float t = rnd(s_dt, q_thresh);
        const float ds = (q.s_raw >= t) ? ds_c : 0.f;
        const float d = rnd(a_dt, rnd(a_dt, ds) / maxq);
        const float dneg = rnd(a_dt, dzp / q.s);                 /* grad of (-wmin) from the zero-point path, cast to a_dt */
        const float dwmin = rnd(a_dt, (-d) + (-dneg));
        const float dmin = (q.xmin <= 0.f) ? dwmin : 0.f;
        const float dmax = (q.xmax >= 0.f) ? d : 0.f;
        const int64_t i0 = g * gs + q.imin, i1 = g * gs + q.imax;
        store_from_f32(dX, i0, a_dt, load_as_f32(dX, i0, a_dt) + dmin);
        store_from_f32(dX, i1, a_dt, load_as_f32(dX, i1, a_dt) + dmax);
    }
}
