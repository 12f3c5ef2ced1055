// ar_exact.hip -- the decoder block's elementwise / normalisation work with the ROUNDING POINTS AND THE SUMMATION ORDER OF EAGER
// TORCH ON THIS GPU, so that a block run through these kernels produces the same bits as the module code the reference runs
// (transformers/models/llama/modeling_llama.py under auto_round's block_forward, auto_round/compressors/utils.py:109-172) -- and
// with them the same sign-SGD trajectory and the same packed weights ("exact_rounding", auto_round_amd/exact_block.py).
//
// What "the same bits" takes, op by op (the eager ops are separate ATen kernels, each rounding its result to the tensor dtype):
//   LlamaRMSNorm.forward      x.float() ; pow(2) = x*x ; mean(-1) ; + eps ; rsqrt ; x * r ; .to(dt) ; w * (.)
//        mean(-1) is ATen's reduction kernel (ATen/native/cuda/Reduce.cuh): for a contiguous inner dimension of N >= 128 fp32
//        values every "thread" x of a 64-wide row of threads owns the float4 vectors x, x + 64, x + 128, ... of the row, keeps ONE
//        ACCUMULATOR PER VECTOR SLOT (4), adds them as ((a0 + a1) + a2) + a3, and the 64 threads combine neighbours first
//        (shuffle-down by 1, 2, 4, ... on ROCm); rows of more than 8128 values are first split over the 8 thread-rows of the block
//        (vectors x + 64 y + 512 j), reduced per thread-row as above and combined as ((y0+y4)+(y2+y6)) + ((y1+y5)+(y3+y7)).
//        `torch_rowsum` below reproduces exactly that association with 32 lanes per row (a lane owns two adjacent torch threads).
//        mean = sum * factor with factor = float(rows) / float(rows * N) (MeanOps::project).
//   its autograd backward     g = dt(dy * w) ; d_r = sum(g.float() * x, -1) (same reduction) ; rsqrt': (-0.5 * d_r) * ((r*r)*r) ;
//                             mean': * (1 / N) ; pow': * (2 * x) ; + g * r ; .to(dt) ; (+ the residual branch's gradient, in dt)
//   apply_rotary_pos_emb      dt(dt(x * cos) + dt(rotate_half(x) * sin)) and its backward dt(dt(g * cos) + -/+ dt(g' * sin'))
//   LlamaMLP act_fn(g) * u    dt(dt(silu(g)) * u), silu(x) = x / (1 + exp(-x)) in fp32 ; backward: d_silu = dt(da * u),
//                             du = dt(da * dt(silu(g))), dg = dt((d_silu * s) * (1 + g * (1 - s))), s = 1 / (1 + exp(-g)) --
//                             ATen's silu_backward kernel is ONE kernel compiled with HIP's default fp contraction, so its
//                             `1 + g * (1 - s)` is an fma there; `contract` selects that form (the caller verifies against torch).
// Everything else in this library is built with -ffp-contract=off; so is this file (explicit fmaf where torch's kernel has one).
#include "ar_common.hpp"

namespace ar {

constexpr int kRowLanes = 32;                    // lanes that share one row (two rows per wavefront)
constexpr int kRowsPerBlock = kTPB / kRowLanes;  // 8

// Sum over one row of `nch` chunks of 8 consecutive values in the association of ATen's reduction kernel (see the file header).
// chunk(c, v) fills v[0..8) with the values of elements 8c .. 8c+7.  Y = 1 | 8: thread-rows the row is split over.
template <class ChunkFn>
__device__ __forceinline__ float torch_rowsum(int l32, int nch, int Y, ChunkFn&& chunk) {
    float R[8];
#pragma unroll
    for (int y = 0; y < 8; ++y) R[y] = 0.f;
    for (int y = 0; y < Y; ++y) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll 2
        for (int c = l32 + kRowLanes * y; c < nch; c += kRowLanes * Y) {
            float v[8];
            chunk(c, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = acc[i] + v[i];        // (the first add is ATen's ident + v)
        }
        const float ta = ((acc[0] + acc[1]) + acc[2]) + acc[3];       // torch thread 2 l
        const float tb = ((acc[4] + acc[5]) + acc[6]) + acc[7];       // torch thread 2 l + 1
        const float s = group_sum<kRowLanes>(ta + tb);                 // shuffle-down 1, then 2, 4, ..., 32: neighbours first
        if (Y == 1) return s;
        R[y] = s;
    }
    return ((R[0] + R[4]) + (R[2] + R[6])) + ((R[1] + R[5]) + (R[3] + R[7]));
}

// which split ATen picks for `rows` rows of `hidden` fp32 values reduced over the (contiguous) inner dimension on a device with
// >= 100 CUs: 1 / 8 thread-rows per row, or -1 where another code path would run (short rows: no vectorised loads; fewer than 8
// rows: a wider thread-row; see setReduceConfig)
static inline int torch_reduce_split(int64_t rows, int hidden) {
    if (hidden < 128 || hidden % kEPT || rows < 8 || (int64_t)hidden * rows > 0x1fffffffLL) return -1;
    const int values_per_thread = (hidden + 63) / 64;
    if (values_per_thread < 128) return 1;
    if ((hidden + 511) / 512 >= 256) return -1;       // would also split over thread blocks
    return 8;
}

template <int DT, bool RES>
__global__ __launch_bounds__(kTPB) void k_x_rmsnorm_fwd(const void* __restrict__ x, const void* __restrict__ res, const void* __restrict__ w,
                                                         void* __restrict__ sum_out, void* __restrict__ y, float* __restrict__ rstd_out,
                                                         int64_t rows, int hidden, float eps, float mean_factor, int Y, int flags) {
    const int l32 = threadIdx.x & (kRowLanes - 1);
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x / kRowLanes);
    if (row >= rows) return;
    const int nch = hidden / kEPT;
    const int64_t base = row * hidden;
    auto load = [&](int c, float (&v)[8]) {
        unpack8<DT>(load8_raw<DT>(x, base + (int64_t)c * kEPT), v);
        if (RES) {      // the residual add is its own eager op: rounded to dt before anything else sees it
            float r[8];
            unpack8<DT>(load8_raw<DT>(res, base + (int64_t)c * kEPT), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = round_to<DT>(v[j] + r[j]);
        }
    };
    const float ss = torch_rowsum(l32, nch, Y, [&](int c, float (&p)[8]) {
        float v[8];
        load(c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = v[j] * v[j];
    });
    // torch.rsqrt on a float tensor: ATen calls `::rsqrt(a)`, and HIP's headers only declare rsqrt(double) (rsqrtf is the float
    // one) -- the argument is promoted, the result computed in double and rounded to float.  v_rsq_f32 (rsqrtf) differs from that
    // in the last bit for ~12 % of the inputs.  flags & 1: the float instruction instead (another stack might resolve it that way).
    const float var = ss * mean_factor + eps;
    const float r = (flags & 1) ? rsqrtf(var) : (float)rsqrt((double)var);
    if (l32 == 0 && rstd_out) rstd_out[row] = (flags & 2) ? ss : r;          // flags & 2 (probes): the raw row sum
    for (int c = l32; c < nch; c += kRowLanes) {
        float v[8], wv[8], o[8];
        load(c, v);
        if (RES) store8<DT>(sum_out, base + (int64_t)c * kEPT, v);
        unpack8<DT>(load8_raw<DT>(w, (int64_t)c * kEPT), wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = wv[j] * round_to<DT>(v[j] * r);
        store8<DT>(y, base + (int64_t)c * kEPT, o);
    }
}

template <int DT, bool RES>
__global__ __launch_bounds__(kTPB) void k_x_rmsnorm_bwd(const void* __restrict__ dy, const void* __restrict__ x, const void* __restrict__ w,
                                                         const float* __restrict__ rstd, const void* __restrict__ dres, void* __restrict__ dx,
                                                         int64_t rows, int hidden, int Y) {
    const int l32 = threadIdx.x & (kRowLanes - 1);
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x / kRowLanes);
    if (row >= rows) return;
    const int nch = hidden / kEPT;
    const int64_t base = row * hidden;
    const float r = rstd[row];
    auto grad_in = [&](int c, float (&g)[8], float (&xv)[8]) {          // g = dt(dy * w) (MulBackward of `weight * h.to(dt)`), x
        float d[8], wv[8];
        unpack8<DT>(load8_raw<DT>(dy, base + (int64_t)c * kEPT), d);
        unpack8<DT>(load8_raw<DT>(w, (int64_t)c * kEPT), wv);
        unpack8<DT>(load8_raw<DT>(x, base + (int64_t)c * kEPT), xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = round_to<DT>(d[j] * wv[j]);
    };
    const float d_r = torch_rowsum(l32, nch, Y, [&](int c, float (&p)[8]) {
        float g[8], xv[8];
        grad_in(c, g, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = g[j] * xv[j];
    });
    const float d_var = (-0.5f * d_r) * ((r * r) * r);                   // rsqrt_backward: -0.5 * grad * result.pow(3)
    const float d_sq = d_var * (1.0f / (float)hidden);                   // mean_backward: grad.expand(...) / N  (GPU: * (1 / N))
    for (int c = l32; c < nch; c += kRowLanes) {
        float g[8], xv[8], o[8];
        grad_in(c, g, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = round_to<DT>(g[j] * r + d_sq * (2.0f * xv[j]));     // (contraction is off: three roundings)
        if (RES) {
            float rv[8];
            unpack8<DT>(load8_raw<DT>(dres, base + (int64_t)c * kEPT), rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = o[j] + rv[j];
        }
        store8<DT>(dx, base + (int64_t)c * kEPT, o);
    }
}

// ---- rotary embedding, q and k as separate (possibly strided) projections, no head repeat ------------------------------------
template <int DT>
__global__ __launch_bounds__(kTPB) void k_x_rope_fwd(const void* __restrict__ q, int64_t ldq, const void* __restrict__ k, int64_t ldk,
                                                      const void* __restrict__ cs, const void* __restrict__ sn, int64_t cs_bstride,
                                                      void* __restrict__ qo, void* __restrict__ ko, int64_t tokens, int64_t seq, int hq, int hkv,
                                                      int d) {
    const int ppl = d / (2 * kEPT);
    const int heads = hq + hkv;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= tokens * heads * ppl) return;
    const int p = (int)(idx % ppl);
    const int head = (int)((idx / ppl) % heads);
    const int64_t t = idx / ((int64_t)ppl * heads);
    const bool isq = head < hq;
    const void* src = isq ? q : k;
    void* dst = isq ? qo : ko;
    const int h = isq ? head : head - hq;
    const int64_t so = t * (isq ? ldq : ldk) + (int64_t)h * d + p * kEPT;
    const int64_t dofs = t * (int64_t)(isq ? hq : hkv) * d + (int64_t)h * d + p * kEPT;
    float lo[8], hi[8], olo[8], ohi[8], cl[8], ch[8], sl[8], sh[8];
    unpack8<DT>(load8_raw<DT>(src, so), lo);
    unpack8<DT>(load8_raw<DT>(src, so + d / 2), hi);
    const int64_t b = t / seq, s = t - b * seq;
    const int64_t co = b * cs_bstride + s * d + p * kEPT;
    unpack8<DT>(load8_raw<DT>(cs, co), cl);
    unpack8<DT>(load8_raw<DT>(cs, co + d / 2), ch);
    unpack8<DT>(load8_raw<DT>(sn, co), sl);
    unpack8<DT>(load8_raw<DT>(sn, co + d / 2), sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        olo[j] = round_to<DT>(lo[j] * cl[j]) + round_to<DT>(-hi[j] * sl[j]);       // rotate_half: the first half takes -x[d/2:]
        ohi[j] = round_to<DT>(hi[j] * ch[j]) + round_to<DT>(lo[j] * sh[j]);
    }
    store8<DT>(dst, dofs, olo);
    store8<DT>(dst, dofs + d / 2, ohi);
}

// gradients of the rotated q / k ([B, S, h, d] addressed through element strides sb / ss / sh: whatever layout the attention
// backward left them in) -> gradients of the projections, token-major contiguous
template <int DT>
__global__ __launch_bounds__(kTPB) void k_x_rope_bwd(const void* __restrict__ gq, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                                                      const void* __restrict__ gk, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                                                      const void* __restrict__ cs, const void* __restrict__ sn, int64_t cs_bstride,
                                                      void* __restrict__ dq, int64_t lddq, void* __restrict__ dk, int64_t lddk, int64_t tokens,
                                                      int64_t seq, int hq, int hkv, int d) {
    const int ppl = d / (2 * kEPT);
    const int heads = hq + hkv;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= tokens * heads * ppl) return;
    const int p = (int)(idx % ppl);
    const int head = (int)((idx / ppl) % heads);
    const int64_t t = idx / ((int64_t)ppl * heads);
    const bool isq = head < hq;
    const int h = isq ? head : head - hq;
    const int64_t b = t / seq, s = t - b * seq;
    const void* src = isq ? gq : gk;
    const int64_t so = isq ? (b * q_sb + s * q_ss + (int64_t)h * q_sh) : (b * k_sb + s * k_ss + (int64_t)h * k_sh);
    float lo[8], hi[8], olo[8], ohi[8], cl[8], ch[8], sl[8], sh[8];
    unpack8<DT>(load8_raw<DT>(src, so + p * kEPT), lo);
    unpack8<DT>(load8_raw<DT>(src, so + p * kEPT + d / 2), hi);
    const int64_t co = b * cs_bstride + s * d + p * kEPT;
    unpack8<DT>(load8_raw<DT>(cs, co), cl);
    unpack8<DT>(load8_raw<DT>(cs, co + d / 2), ch);
    unpack8<DT>(load8_raw<DT>(sn, co), sl);
    unpack8<DT>(load8_raw<DT>(sn, co + d / 2), sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // MulBackward: dt(g * cos), dt(g * sin); Cat / Neg / Slice backward route the second product to the other half
        olo[j] = round_to<DT>(lo[j] * cl[j]) + round_to<DT>(hi[j] * sh[j]);
        ohi[j] = round_to<DT>(hi[j] * ch[j]) + (-round_to<DT>(lo[j] * sl[j]));
    }
    void* dst = isq ? dq : dk;
    const int64_t dofs = t * (isq ? lddq : lddk) + (int64_t)h * d + p * kEPT;
    store8<DT>(dst, dofs, olo);
    store8<DT>(dst, dofs + d / 2, ohi);
}

// ---- SwiGLU with gate and up as separate (possibly strided) projections ---------------------------------------------------------
__device__ __forceinline__ float x_silu(float x) { return x / (1.0f + expf(-x)); }

template <int DT>
__global__ __launch_bounds__(kTPB) void k_x_swiglu_fwd(const void* __restrict__ g, int64_t ldg, const void* __restrict__ u, int64_t ldu,
                                                        void* __restrict__ a, int64_t rows, int64_t F) {
    const int64_t cpr = F / kEPT;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= rows * cpr) return;
    const int64_t r = idx / cpr, c = idx - r * cpr;
    float gv[8], uv[8], o[8];
    unpack8<DT>(load8_raw<DT>(g, r * ldg + c * kEPT), gv);
    unpack8<DT>(load8_raw<DT>(u, r * ldu + c * kEPT), uv);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = round_to<DT>(x_silu(gv[j])) * uv[j];
    store8<DT>(a, r * F + c * kEPT, o);
}

template <int DT, bool CONTRACT>
__global__ __launch_bounds__(kTPB) void k_x_swiglu_bwd(const void* __restrict__ da, const void* __restrict__ g, int64_t ldg,
                                                        const void* __restrict__ u, int64_t ldu, void* __restrict__ dg, int64_t lddg,
                                                        void* __restrict__ du, int64_t lddu, int64_t rows, int64_t F) {
    const int64_t cpr = F / kEPT;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= rows * cpr) return;
    const int64_t r = idx / cpr, c = idx - r * cpr;
    float gv[8], uv[8], d[8], og[8], ou[8];
    unpack8<DT>(load8_raw<DT>(g, r * ldg + c * kEPT), gv);
    unpack8<DT>(load8_raw<DT>(u, r * ldu + c * kEPT), uv);
    unpack8<DT>(load8_raw<DT>(da, r * F + c * kEPT), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ou[j] = d[j] * round_to<DT>(x_silu(gv[j]));                         // MulBackward: da * silu(g) (the saved dt tensor)
        const float ds = round_to<DT>(d[j] * uv[j]);                         // MulBackward: da * u, rounded to dt
        const float s = 1.0f / (1.0f + expf(-gv[j]));                        // silu_backward (one ATen kernel)
        const float t = CONTRACT ? __builtin_fmaf(gv[j], 1.0f - s, 1.0f) : 1.0f + gv[j] * (1.0f - s);
        og[j] = (ds * s) * t;
    }
    store8<DT>(dg, r * lddg + c * kEPT, og);
    store8<DT>(du, r * lddu + c * kEPT, ou);
}

static inline int xgrid1d(int64_t n) { return (int)((n + kTPB - 1) / kTPB); }

}  // namespace ar

using namespace ar;

#define AR_DT_SWITCH2(dt, CALL)                    \
    switch (dt) {                                  \
        case AR_DT_BF16: CALL(AR_DT_BF16); break;  \
        case AR_DT_F16: CALL(AR_DT_F16); break;    \
        default: return AR_ERR_UNSUPPORTED;        \
    }

extern "C" int ar_rmsnorm_fwd_exact(const void* x, const void* res, const void* w, void* sum_out, void* y, float* rstd_out, int64_t rows,
                                    int hidden, float eps, float mean_factor, int flags, int dt, ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    const int Y = torch_reduce_split(rows, hidden);
    if (Y < 0 || !x || !w || !y || (res && !sum_out)) return AR_ERR_UNSUPPORTED;
    if (mean_factor <= 0.f) mean_factor = (float)rows / (float)(rows * (int64_t)hidden);
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
#define AR_CALL(DT)                                                                                                                     \
    if (res) AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_rmsnorm_fwd<DT, true>), grid, kTPB, 0, st, x, res, w, sum_out, y, rstd_out, rows, hidden, eps, mean_factor, Y, flags); \
    else AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_rmsnorm_fwd<DT, false>), grid, kTPB, 0, st, x, res, w, sum_out, y, rstd_out, rows, hidden, eps, mean_factor, Y, flags)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_rmsnorm_bwd_exact(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx, int64_t rows,
                                    int hidden, int dt, ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    const int Y = torch_reduce_split(rows, hidden);
    if (Y < 0 || !dy || !x || !w || !rstd || !dx) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
#define AR_CALL(DT)                                                                                                           \
    if (dres) AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_rmsnorm_bwd<DT, true>), grid, kTPB, 0, st, dy, x, w, rstd, dres, dx, rows, hidden, Y); \
    else AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_x_rmsnorm_bwd<DT, false>), grid, kTPB, 0, st, dy, x, w, rstd, dres, dx, rows, hidden, Y)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_rope_fwd_exact(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* cos, const void* sin,
                                 int64_t cs_batch_stride, void* q_out, void* k_out, int64_t tokens, int64_t seq, int hq, int hkv, int d, int dt,
                                 ar_stream_t stream) {
    if (tokens <= 0) return AR_OK;
    if (d % (2 * kEPT) || hq <= 0 || hkv <= 0 || ldq % kEPT || ldk % kEPT || seq <= 0 || tokens % seq) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = xgrid1d(tokens * (hq + hkv) * (d / (2 * kEPT)));
#define AR_CALL(DT) AR_LAUNCH_PROF(AR_PROF_ROPE, tokens, (k_x_rope_fwd<DT>), grid, kTPB, 0, st, q, ldq, k, ldk, cos, sin, cs_batch_stride, q_out, k_out, tokens, seq, hq, hkv, d)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_rope_bwd_exact(const void* gq, int64_t q_sb, int64_t q_ss, int64_t q_sh, const void* gk, int64_t k_sb, int64_t k_ss,
                                 int64_t k_sh, const void* cos, const void* sin, int64_t cs_batch_stride, void* dq, int64_t lddq, void* dk,
                                 int64_t lddk, int64_t tokens, int64_t seq, int hq, int hkv, int d, int dt, ar_stream_t stream) {
    if (tokens <= 0) return AR_OK;
    if (d % (2 * kEPT) || hq <= 0 || hkv <= 0 || seq <= 0 || tokens % seq) return AR_ERR_UNSUPPORTED;
    if ((q_sb | q_ss | q_sh | k_sb | k_ss | k_sh | lddq | lddk) % kEPT) return AR_ERR_UNSUPPORTED;      // 16-byte accesses
    if (lddq < (int64_t)hq * d || lddk < (int64_t)hkv * d) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = xgrid1d(tokens * (hq + hkv) * (d / (2 * kEPT)));
#define AR_CALL(DT) AR_LAUNCH_PROF(AR_PROF_ROPE, tokens, (k_x_rope_bwd<DT>), grid, kTPB, 0, st, gq, q_sb, q_ss, q_sh, gk, k_sb, k_ss, k_sh, cos, sin, cs_batch_stride, dq, lddq, dk, lddk, tokens, seq, hq, hkv, d)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_swiglu_fwd_exact(const void* g, int64_t ldg, const void* u, int64_t ldu, void* a, int64_t rows, int64_t F, int dt,
                                   ar_stream_t stream) {
    if (rows <= 0 || F <= 0) return AR_OK;
    if (F % kEPT || ldg % kEPT || ldu % kEPT || ldg < F || ldu < F) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = xgrid1d(rows * (F / kEPT));
#define AR_CALL(DT) AR_LAUNCH_PROF(AR_PROF_SWIGLU, rows, (k_x_swiglu_fwd<DT>), grid, kTPB, 0, st, g, ldg, u, ldu, a, rows, F)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_swiglu_bwd_exact(const void* da, const void* g, int64_t ldg, const void* u, int64_t ldu, void* dg, int64_t lddg, void* du,
                                   int64_t lddu, int64_t rows, int64_t F, int contract, int dt, ar_stream_t stream) {
    if (rows <= 0 || F <= 0) return AR_OK;
    if (F % kEPT || ldg % kEPT || ldu % kEPT || ldg < F || ldu < F || lddg % kEPT || lddu % kEPT || lddg < F || lddu < F) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = xgrid1d(rows * (F / kEPT));
#define AR_CALL(DT)                                                                                                                 \
    if (contract) AR_LAUNCH_PROF(AR_PROF_SWIGLU, rows, (k_x_swiglu_bwd<DT, true>), grid, kTPB, 0, st, da, g, ldg, u, ldu, dg, lddg, du, lddu, rows, F); \
    else AR_LAUNCH_PROF(AR_PROF_SWIGLU, rows, (k_x_swiglu_bwd<DT, false>), grid, kTPB, 0, st, da, g, ldg, u, ldu, dg, lddg, du, lddu, rows, F)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}
