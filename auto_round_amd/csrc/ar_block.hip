// ar_block.hip -- the elementwise / normalisation work of a Llama-family decoder block as fused HBM-bound kernels.
//
// replaces (for the duration of the tuning loop): what the reference gets from torch.compile(block_forward)
// (auto_round/utils/device.py:112-122, compressors/base.py:1177-1179, "about 20 %") on the block code transformers runs eagerly:
//   LlamaRMSNorm.forward, apply_rotary_pos_emb + repeat_kv, act_fn(gate) * up and their autograd backwards
//   (transformers/models/llama/modeling_llama.py).  On MI355X these were ~90 launches and 20 % of a Llama-3-8B tuning iteration
//   (profiles/archive/r01_llama8b_block_kernel_stats.csv); here they are six streaming kernels, 16-byte accesses, one pass each.
//
// Forward kernels round where the eager module code rounds (so the fused forward tracks transformers' bf16 forward closely);
// backward kernels evaluate the exact fp32 derivative and round once.  All tensors are token-major: [tokens, features].
#include "ar_common.hpp"

namespace ar {

// ---- RMSNorm ------------------------------------------------------------------------------------------------------------
// y = w * dt(x * rsqrt(mean(x^2) + eps))     (LlamaRMSNorm.forward: fp32 statistics, one rounding to the activation dtype,
// then the product with the weight rounded again).  One wave per row, the row stays in registers (MAXC chunks of 8 per lane).
template <int DT, int MAXC>
__global__ __launch_bounds__(kTPB) void k_rmsnorm_fwd(const void* __restrict__ x, const void* __restrict__ w, void* __restrict__ y,
                                                       float* __restrict__ rstd_out, int64_t rows, int hidden, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kTPB / kWave) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = hidden / kEPT;
    Raw8<DT> rx[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            rx[c] = load8_raw<DT>(x, row * hidden + (int64_t)ch * kEPT);
            float v[8];
            unpack8<DT>(rx[c], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
        }
    }
    ss = lanes_sum(ss, kWave);
    const float rstd = 1.0f / __builtin_sqrtf(ss / (float)hidden + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            float v[8], wv[8], o[8];
            unpack8<DT>(rx[c], v);
            unpack8<DT>(load8_raw<DT>(w, (int64_t)ch * kEPT), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = wv[j] * round_to<DT>(v[j] * rstd);
            store8<DT>(y, row * hidden + (int64_t)ch * kEPT, o);
        }
    }
}

// dx = rstd * (g - xhat * mean(g * xhat)) [+ dres],  g = dy * w,  xhat = x * rstd        (exact derivative, fp32, one rounding)
template <int DT, int MAXC>
__global__ __launch_bounds__(kTPB) void k_rmsnorm_bwd(const void* __restrict__ dy, const void* __restrict__ x, const void* __restrict__ w,
                                                       const float* __restrict__ rstd_in, const void* __restrict__ dres,
                                                       void* __restrict__ dx, int64_t rows, int hidden) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kTPB / kWave) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = hidden / kEPT;
    const float rstd = rstd_in[row];
    Raw8<DT> rx[MAXC], rg[MAXC];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            rx[c] = load8_raw<DT>(x, row * hidden + (int64_t)ch * kEPT);
            rg[c] = load8_raw<DT>(dy, row * hidden + (int64_t)ch * kEPT);
            float xv[8], gv[8], wv[8];
            unpack8<DT>(rx[c], xv);
            unpack8<DT>(rg[c], gv);
            unpack8<DT>(load8_raw<DT>(w, (int64_t)ch * kEPT), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += (gv[j] * wv[j]) * (xv[j] * rstd);
        }
    }
    dot = lanes_sum(dot, kWave) / (float)hidden;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            float xv[8], gv[8], wv[8], o[8];
            unpack8<DT>(rx[c], xv);
            unpack8<DT>(rg[c], gv);
            unpack8<DT>(load8_raw<DT>(w, (int64_t)ch * kEPT), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (gv[j] * wv[j] - (xv[j] * rstd) * dot);
            if (dres) {
                float rv[8];
                unpack8<DT>(load8_raw<DT>(dres, row * hidden + (int64_t)ch * kEPT), rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += rv[j];
            }
            store8<DT>(dx, row * hidden + (int64_t)ch * kEPT, o);
        }
    }
}

// ---- per-head RMSNorm of q and k (Qwen3-style q_norm / k_norm) -----------------------------------------------------------------
// qkv [tokens, ld]: heads 0 .. hq-1 are queries, hq .. hq+hkv-1 keys, the rest values; every query / key head (d elements) is
// normalised on its own with the weight wq / wk [d] (Qwen3RMSNorm = LlamaRMSNorm on the last dimension), values are copied.
// L = d / 8 lanes own one head (8 elements each), 64 / L heads per wave; the sum of squares is a DPP reduction inside the lane group.
template <int DT, int L>
__global__ __launch_bounds__(kTPB) void k_headnorm_fwd(const void* __restrict__ qkv, const void* __restrict__ wq, const void* __restrict__ wk,
                                                        void* __restrict__ out, float* __restrict__ rstd_out, int64_t tokens, int64_t ld, int hq,
                                                        int hkv, float eps) {
    constexpr int D = L * kEPT;
    const int heads = hq + 2 * hkv, nh = hq + hkv;
    const int64_t item = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / L;       // (token, head)
    const int sub = threadIdx.x % L;
    if (item >= tokens * heads) return;
    const int64_t t = item / heads;
    const int head = (int)(item % heads);
    const int64_t at = t * ld + (int64_t)head * D + sub * kEPT;
    const Raw8<DT> rx = load8_raw<DT>(qkv, at);
    if (head >= nh) {                                                           // value head: copied (whole lane groups take this branch)
        *reinterpret_cast<uint4*>(static_cast<uint16_t*>(out) + at) = rx.q;
        return;
    }
    float v[8], wv[8], o[8];
    unpack8<DT>(rx, v);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
    ss = group_sum<L>(ss);
    const float rstd = 1.0f / __builtin_sqrtf(ss / (float)D + eps);
    if (sub == 0 && rstd_out) rstd_out[t * nh + head] = rstd;
    unpack8<DT>(load8_raw<DT>(head < hq ? wq : wk, (int64_t)sub * kEPT), wv);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = wv[j] * round_to<DT>(v[j] * rstd);
    store8<DT>(out, at, o);
}

// in place on the gradient: dqkv <- d(raw qkv) for the query / key heads (exact derivative of the norm, fp32, one rounding); value
// heads are left as they are
template <int DT, int L>
__global__ __launch_bounds__(kTPB) void k_headnorm_bwd(void* __restrict__ dqkv, const void* __restrict__ qkv, const void* __restrict__ wq,
                                                        const void* __restrict__ wk, const float* __restrict__ rstd_in, int64_t tokens,
                                                        int64_t ld, int hq, int hkv) {
    constexpr int D = L * kEPT;
    const int nh = hq + hkv;
    const int64_t item = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / L;       // (token, query-or-key head)
    const int sub = threadIdx.x % L;
    if (item >= tokens * nh) return;
    const int64_t t = item / nh;
    const int head = (int)(item % nh);
    const int64_t at = t * ld + (int64_t)head * D + sub * kEPT;
    const float rstd = rstd_in[item];
    float xv[8], gv[8], wv[8], o[8];
    unpack8<DT>(load8_raw<DT>(qkv, at), xv);
    unpack8<DT>(load8_raw<DT>(dqkv, at), gv);
    unpack8<DT>(load8_raw<DT>(head < hq ? wq : wk, (int64_t)sub * kEPT), wv);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += (gv[j] * wv[j]) * (xv[j] * rstd);
    dot = group_sum<L>(dot) / (float)D;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rstd * (gv[j] * wv[j] - (xv[j] * rstd) * dot);
    store8<DT>(dqkv, at, o);
}

// ---- LayerNorm (OPT / GPT-style blocks) ---------------------------------------------------------------------------------------
// y = dt(((x - mean) * rstd) * w + b)       (nn.LayerNorm under autocast runs in fp32 on the upcast input and the next Linear rounds
// its input once: transformers/models/opt/modeling_opt.py OPTDecoderLayer.self_attn_layer_norm / final_layer_norm).  Two passes over
// the row in registers (mean, then the centred sum of squares); one wave per row.
template <int DT, int MAXC>
__global__ __launch_bounds__(kTPB) void k_layernorm_fwd(const void* __restrict__ x, const void* __restrict__ w, const void* __restrict__ b,
                                                         void* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         int64_t rows, int hidden, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kTPB / kWave) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = hidden / kEPT;
    Raw8<DT> rx[MAXC];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            rx[c] = load8_raw<DT>(x, row * hidden + (int64_t)ch * kEPT);
            float v[8];
            unpack8<DT>(rx[c], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[j];
        }
    }
    const float mean = lanes_sum(sum, kWave) / (float)hidden;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (lane + c * kWave < nch) {
            float v[8];
            unpack8<DT>(rx[c], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += (v[j] - mean) * (v[j] - mean);
        }
    }
    const float rstd = 1.0f / __builtin_sqrtf(lanes_sum(ss, kWave) / (float)hidden + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            float v[8], wv[8], bv[8], o[8];
            unpack8<DT>(rx[c], v);
            unpack8<DT>(load8_raw<DT>(w, (int64_t)ch * kEPT), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = 0.f;
            if (b) unpack8<DT>(load8_raw<DT>(b, (int64_t)ch * kEPT), bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = ((v[j] - mean) * rstd) * wv[j] + bv[j];
            store8<DT>(y, row * hidden + (int64_t)ch * kEPT, o);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres],  g = dy * w,  xhat = (x - mean) * rstd     (exact derivative, fp32)
template <int DT, int MAXC>
__global__ __launch_bounds__(kTPB) void k_layernorm_bwd(const void* __restrict__ dy, const void* __restrict__ x, const void* __restrict__ w,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         const void* __restrict__ dres, void* __restrict__ dx, int64_t rows, int hidden) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kTPB / kWave) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = hidden / kEPT;
    const float mean = mean_in[row], rstd = rstd_in[row];
    Raw8<DT> rx[MAXC], rg[MAXC];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            rx[c] = load8_raw<DT>(x, row * hidden + (int64_t)ch * kEPT);
            rg[c] = load8_raw<DT>(dy, row * hidden + (int64_t)ch * kEPT);
            float xv[8], gv[8], wv[8];
            unpack8<DT>(rx[c], xv);
            unpack8<DT>(rg[c], gv);
            unpack8<DT>(load8_raw<DT>(w, (int64_t)ch * kEPT), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = gv[j] * wv[j];
                sg += g;
                sgx += g * ((xv[j] - mean) * rstd);
            }
        }
    }
    sg = lanes_sum(sg, kWave) / (float)hidden;
    sgx = lanes_sum(sgx, kWave) / (float)hidden;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * kWave;
        if (ch < nch) {
            float xv[8], gv[8], wv[8], o[8];
            unpack8<DT>(rx[c], xv);
            unpack8<DT>(rg[c], gv);
            unpack8<DT>(load8_raw<DT>(w, (int64_t)ch * kEPT), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (gv[j] * wv[j] - sg - ((xv[j] - mean) * rstd) * sgx);
            if (dres) {
                float rv[8];
                unpack8<DT>(load8_raw<DT>(dres, row * hidden + (int64_t)ch * kEPT), rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += rv[j];
            }
            store8<DT>(dx, row * hidden + (int64_t)ch * kEPT, o);
        }
    }
}

// ---- SwiGLU -------------------------------------------------------------------------------------------------------------
// a = dt(dt(silu(g)) * u), g = gu[:, :F], u = gu[:, F:2F]       (LlamaMLP.forward: act_fn(gate_proj(x)) * up_proj(x))
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

template <int DT>
__global__ __launch_bounds__(kTPB) void k_swiglu_fwd(const void* __restrict__ gu, int64_t ld, void* __restrict__ a, int64_t rows, int64_t F) {
    const int64_t cpr = F / kEPT;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= rows * cpr) return;
    const int64_t r = idx / cpr, c = idx - r * cpr;
    float g[8], u[8], o[8];
    unpack8<DT>(load8_raw<DT>(gu, r * ld + c * kEPT), g);
    unpack8<DT>(load8_raw<DT>(gu, r * ld + F + c * kEPT), u);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = round_to<DT>(silu_f(g[j])) * u[j];
    store8<DT>(a, r * F + c * kEPT, o);
}

// in place on gu: (g, u) <- (dg, du) with dg = da * u * silu'(g), du = da * silu(g); silu'(g) = s * (1 + g * (1 - s)), s = sigmoid(g)
template <int DT>
__global__ __launch_bounds__(kTPB) void k_swiglu_bwd(const void* __restrict__ da, void* __restrict__ gu, int64_t ld, int64_t rows, int64_t F) {
    const int64_t cpr = F / kEPT;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= rows * cpr) return;
    const int64_t r = idx / cpr, c = idx - r * cpr;
    float g[8], u[8], d[8], dg[8], du[8];
    unpack8<DT>(load8_raw<DT>(gu, r * ld + c * kEPT), g);
    unpack8<DT>(load8_raw<DT>(gu, r * ld + F + c * kEPT), u);
    unpack8<DT>(load8_raw<DT>(da, r * F + c * kEPT), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float s = 1.0f / (1.0f + expf(-g[j]));
        dg[j] = d[j] * u[j] * (s * (1.0f + g[j] * (1.0f - s)));
        du[j] = d[j] * (g[j] * s);
    }
    store8<DT>(gu, r * ld + c * kEPT, dg);
    store8<DT>(gu, r * ld + F + c * kEPT, du);
}

// ---- sparse-MoE routing glue (sorted-token expert pass) -----------------------------------------------------------------------
// The expert pass of a MoE block runs over rows sorted by expert: row p of the sorted buffers belongs to token tok[p]; token t's
// K routed copies sit at rows pos[t * K + k].  Three streaming kernels replace the per-expert index / scale / index_add_ ops of
// MixtralExperts.forward (transformers/models/mixtral/modeling_mixtral.py) and their autograd mirrors; none uses float atomics.
//   expand : out[p, :] = dt(scale[p] * src[tok[p], :])                       (scale NULL: plain row gather)
//   combine: out[t, :] = dt(res[t, :] + sum_k w[t K + k] * D[pos[t K + k], :])   (res / w NULL: 0 / 1), fp32 sum in slot order
//   rowdot : out[p]    = sum_j A[tok[p], j] * B[p, j]                          (fp32; one wave per row)
template <int DT>
__global__ __launch_bounds__(kTPB) void k_moe_expand(const void* __restrict__ src, const int64_t* __restrict__ tok, const float* __restrict__ scale,
                                                      void* __restrict__ out, int64_t rows, int64_t H) {
    const int64_t cpr = H / kEPT;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= rows * cpr) return;
    const int64_t p = idx / cpr, c = idx - p * cpr;
    float v[8];
    unpack8<DT>(load8_raw<DT>(src, tok[p] * H + c * kEPT), v);
    if (scale) {
        const float sc = scale[p];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= sc;
    }
    store8<DT>(out, p * H + c * kEPT, v);
}

template <int DT>
__global__ __launch_bounds__(kTPB) void k_moe_combine(const void* __restrict__ D, const int64_t* __restrict__ pos, const float* __restrict__ w,
                                                       const void* __restrict__ res, void* __restrict__ out, int64_t T, int64_t H, int K) {
    const int64_t cpr = H / kEPT;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= T * cpr) return;
    const int64_t t = idx / cpr, c = idx - t * cpr;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (res) unpack8<DT>(load8_raw<DT>(res, t * H + c * kEPT), acc);
    for (int k = 0; k < K; ++k) {
        float v[8];
        unpack8<DT>(load8_raw<DT>(D, pos[t * K + k] * H + c * kEPT), v);
        const float wk = w ? w[t * K + k] : 1.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wk * v[j];
    }
    store8<DT>(out, t * H + c * kEPT, acc);
}

template <int DT>
__global__ __launch_bounds__(kTPB) void k_moe_rowdot(const void* __restrict__ A, const int64_t* __restrict__ tok, const void* __restrict__ B,
                                                      float* __restrict__ out, int64_t rows, int64_t H) {
    const int64_t p = (int64_t)blockIdx.x * (kTPB / kWave) + (threadIdx.x >> 6);
    if (p >= rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t a0 = tok[p] * H, b0 = p * H;
    float s = 0.f;
    for (int64_t c = lane; c < H / kEPT; c += kWave) {
        float x[8], y[8];
        unpack8<DT>(load8_raw<DT>(A, a0 + c * kEPT), x);
        unpack8<DT>(load8_raw<DT>(B, b0 + c * kEPT), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j] * y[j];
    }
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, kWave);
    if (lane == 0) out[p] = s;
}

// ---- rotary embedding + grouped-query head repeat -------------------------------------------------------------------------
// qkv [T, (hq + 2 hkv) * d] (ld) -> q [T, hq*d], k [T, hq*d], v [T, hq*d] with every kv head written hq/hkv times (what repeat_kv
// materialises for the SDPA kernels).  x_embed = dt(dt(x*cos) + dt(rotate_half(x)*sin))   (apply_rotary_pos_emb, each op rounded).
// One lane handles the chunk pair (i, i + d/2) of one head of one token.
template <int DT>
__global__ __launch_bounds__(kTPB) void k_rope_fwd(const void* __restrict__ qkv, int64_t ld, const void* __restrict__ cs, const void* __restrict__ sn,
                                                    int64_t cs_bstride, void* __restrict__ q, void* __restrict__ k, void* __restrict__ v,
                                                    int64_t tokens, int64_t seq, int hq, int hkv, int d) {
    const int ppl = d / (2 * kEPT);                         // chunk pairs per head
    const int heads = hq + 2 * hkv;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= tokens * heads * ppl) return;
    const int p = (int)(idx % ppl);
    const int head = (int)((idx / ppl) % heads);
    const int64_t t = idx / ((int64_t)ppl * heads);
    const int64_t src = t * ld + (int64_t)head * d + p * kEPT;
    float lo[8], hi[8], olo[8], ohi[8];
    unpack8<DT>(load8_raw<DT>(qkv, src), lo);
    unpack8<DT>(load8_raw<DT>(qkv, src + d / 2), hi);
    const int rep = hq / hkv;
    if (head < hq + hkv) {
        const int64_t b = t / seq, s = t - b * seq;
        const int64_t co = b * cs_bstride + s * d + p * kEPT;
        float cl[8], ch[8], sl[8], sh[8];
        unpack8<DT>(load8_raw<DT>(cs, co), cl);
        unpack8<DT>(load8_raw<DT>(cs, co + d / 2), ch);
        unpack8<DT>(load8_raw<DT>(sn, co), sl);
        unpack8<DT>(load8_raw<DT>(sn, co + d / 2), sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            olo[j] = round_to<DT>(lo[j] * cl[j]) + round_to<DT>(-hi[j] * sl[j]);       // rotate_half: first half takes -x[d/2:]
            ohi[j] = round_to<DT>(hi[j] * ch[j]) + round_to<DT>(lo[j] * sh[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { olo[j] = lo[j]; ohi[j] = hi[j]; }
    }
    const int64_t orow = t * (int64_t)hq * d;
    if (head < hq) {
        store8<DT>(q, orow + (int64_t)head * d + p * kEPT, olo);
        store8<DT>(q, orow + (int64_t)head * d + p * kEPT + d / 2, ohi);
    } else {
        void* dst = head < hq + hkv ? k : v;
        const int kvh = head < hq + hkv ? head - hq : head - hq - hkv;
        for (int r = 0; r < rep; ++r) {
            const int64_t o = orow + (int64_t)(kvh * rep + r) * d + p * kEPT;
            store8<DT>(dst, o, olo);
            store8<DT>(dst, o + d / 2, ohi);
        }
    }
}

// backward: dq, dk, dv [T, hq*d] (gradients of the repeated heads) -> dqkv [T, (hq + 2 hkv) * d] (ld): sums over the hq/hkv copies,
// then the transpose of the rotation: dx_lo = dy_lo*cos_lo + dy_hi*sin_hi ; dx_hi = dy_hi*cos_hi - dy_lo*sin_lo.
template <int DT>
__global__ __launch_bounds__(kTPB) void k_rope_bwd(const void* __restrict__ dq, const void* __restrict__ dk, const void* __restrict__ dv,
                                                    const void* __restrict__ cs, const void* __restrict__ sn, int64_t cs_bstride,
                                                    void* __restrict__ dqkv, int64_t ld, int64_t tokens, int64_t seq, int hq, int hkv, int d) {
    const int ppl = d / (2 * kEPT);
    const int heads = hq + 2 * hkv;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= tokens * heads * ppl) return;
    const int p = (int)(idx % ppl);
    const int head = (int)((idx / ppl) % heads);
    const int64_t t = idx / ((int64_t)ppl * heads);
    const int rep = hq / hkv;
    const int64_t irow = t * (int64_t)hq * d;
    float lo[8], hi[8];
    if (head < hq) {
        unpack8<DT>(load8_raw<DT>(dq, irow + (int64_t)head * d + p * kEPT), lo);
        unpack8<DT>(load8_raw<DT>(dq, irow + (int64_t)head * d + p * kEPT + d / 2), hi);
    } else {
        const void* srcp = head < hq + hkv ? dk : dv;
        const int kvh = head < hq + hkv ? head - hq : head - hq - hkv;
#pragma unroll
        for (int j = 0; j < 8; ++j) { lo[j] = 0.f; hi[j] = 0.f; }
        for (int r = 0; r < rep; ++r) {
            float a[8], b[8];
            const int64_t o = irow + (int64_t)(kvh * rep + r) * d + p * kEPT;
            unpack8<DT>(load8_raw<DT>(srcp, o), a);
            unpack8<DT>(load8_raw<DT>(srcp, o + d / 2), b);
#pragma unroll
            for (int j = 0; j < 8; ++j) { lo[j] += a[j]; hi[j] += b[j]; }
        }
    }
    float olo[8], ohi[8];
    if (head < hq + hkv) {
        const int64_t b = t / seq, s = t - b * seq;
        const int64_t co = b * cs_bstride + s * d + p * kEPT;
        float cl[8], ch[8], sl[8], sh[8];
        unpack8<DT>(load8_raw<DT>(cs, co), cl);
        unpack8<DT>(load8_raw<DT>(cs, co + d / 2), ch);
        unpack8<DT>(load8_raw<DT>(sn, co), sl);
        unpack8<DT>(load8_raw<DT>(sn, co + d / 2), sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            olo[j] = lo[j] * cl[j] + hi[j] * sh[j];
            ohi[j] = hi[j] * ch[j] - lo[j] * sl[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { olo[j] = lo[j]; ohi[j] = hi[j]; }
    }
    const int64_t dst = t * ld + (int64_t)head * d + p * kEPT;
    store8<DT>(dqkv, dst, olo);
    store8<DT>(dqkv, dst + d / 2, ohi);
}

// ------------------------------------------------------------------------------------------------------------------
// 2-byte matrix transpose dst[c, r] = src[r, c] (rows, cols multiples of 64): a 64 x 64 tile through LDS.  Loads are 16 B per lane
// (eight lanes cover 128 B of a source row), stores likewise (eight lanes cover 128 B of a destination row); the LDS row pitch of
// 33 words keeps the column gather at two lanes per bank.  The fused block keeps W^T next to W so that the input-gradient GEMM
// dX = dY W runs in the library's fast layout (both operands contiguous along the reduction).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kTrTile = 64, kTrPitch = 66;      // pitch in 2-byte elements

__global__ __launch_bounds__(kTPB) void k_transpose16(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int64_t rows, int64_t cols,
                                                       int tiles_c) {
    __shared__ uint32_t lds[kTrTile * kTrPitch / 2];
    const int64_t tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int64_t r0 = tr * kTrTile, c0 = tc * kTrTile;
    const uint16_t* l16 = reinterpret_cast<const uint16_t*>(lds);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = threadIdx.x + p * kTPB, r = idx >> 3, cc = idx & 7;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + (r0 + r) * cols + c0 + cc * 8);
        uint32_t* w = lds + (r * kTrPitch + cc * 8) / 2;
        w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = threadIdx.x + p * kTPB, oc = idx >> 3, rr = idx & 7;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = l16[(rr * 8 + 2 * j) * kTrPitch + oc], hi = l16[(rr * 8 + 2 * j + 1) * kTrPitch + oc];
            o[j] = lo | (hi << 16);
        }
        const u32x4_t v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<u32x4_t*>(dst + (c0 + oc) * rows + r0 + rr * 8) = v;
    }
}

static inline int grid1d(int64_t n) { return (int)((n + kTPB - 1) / kTPB); }

}  // namespace ar

using namespace ar;

#define AR_DT_SWITCH2(dt, CALL)                    \
    switch (dt) {                                  \
        case AR_DT_BF16: CALL(AR_DT_BF16); break;  \
        case AR_DT_F16: CALL(AR_DT_F16); break;    \
        default: return AR_ERR_UNSUPPORTED;        \
    }

extern "C" int ar_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd_out, int64_t rows, int hidden, float eps, int dt,
                              ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    if (hidden <= 0 || hidden % kEPT || hidden > 16 * kWave * kEPT) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((rows + 3) / 4);
    const bool small = hidden <= 8 * kWave * kEPT;
#define AR_CALL(DT)                                                                                                              \
    if (small) AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_rmsnorm_fwd<DT, 8>), grid, kTPB, 0, st, x, w, y, rstd_out, rows, hidden, eps); \
    else AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_rmsnorm_fwd<DT, 16>), grid, kTPB, 0, st, x, w, y, rstd_out, rows, hidden, eps)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                              int64_t rows, int hidden, int dt, ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    if (hidden <= 0 || hidden % kEPT || hidden > 16 * kWave * kEPT || !rstd) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((rows + 3) / 4);
    const bool small = hidden <= 8 * kWave * kEPT;
#define AR_CALL(DT)                                                                                                              \
    if (small) AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_rmsnorm_bwd<DT, 8>), grid, kTPB, 0, st, dy, x, w, rstd, dres, dx, rows, hidden); \
    else AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_rmsnorm_bwd<DT, 16>), grid, kTPB, 0, st, dy, x, w, rstd, dres, dx, rows, hidden)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_headnorm_fwd(const void* qkv, const void* wq, const void* wk, void* out, float* rstd_out, int64_t tokens, int64_t ld,
                               int hq, int hkv, int d, float eps, int dt, ar_stream_t stream) {
    if (tokens <= 0) return AR_OK;
    const int L = d / kEPT;
    if (d % kEPT || !(L == 8 || L == 16 || L == 32 || L == 64) || hq <= 0 || hkv <= 0 || ld % kEPT || ld < (int64_t)(hq + 2 * hkv) * d || !wq || !wk)
        return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(tokens * (hq + 2 * hkv) * L);
#define AR_CALL_L(DT, LL) hipLaunchKernelGGL((k_headnorm_fwd<DT, LL>), grid, kTPB, 0, st, qkv, wq, wk, out, rstd_out, tokens, ld, hq, hkv, eps)
#define AR_CALL(DT) switch (L) { case 8: AR_CALL_L(DT, 8); break; case 16: AR_CALL_L(DT, 16); break; case 32: AR_CALL_L(DT, 32); break; default: AR_CALL_L(DT, 64); }
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
#undef AR_CALL_L
    return launch_status();
}

extern "C" int ar_headnorm_bwd(void* dqkv, const void* qkv, const void* wq, const void* wk, const float* rstd, int64_t tokens, int64_t ld,
                               int hq, int hkv, int d, int dt, ar_stream_t stream) {
    if (tokens <= 0) return AR_OK;
    const int L = d / kEPT;
    if (d % kEPT || !(L == 8 || L == 16 || L == 32 || L == 64) || hq <= 0 || hkv <= 0 || ld % kEPT || ld < (int64_t)(hq + 2 * hkv) * d || !wq || !wk || !rstd)
        return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(tokens * (hq + hkv) * L);
#define AR_CALL_L(DT, LL) hipLaunchKernelGGL((k_headnorm_bwd<DT, LL>), grid, kTPB, 0, st, dqkv, qkv, wq, wk, rstd, tokens, ld, hq, hkv)
#define AR_CALL(DT) switch (L) { case 8: AR_CALL_L(DT, 8); break; case 16: AR_CALL_L(DT, 16); break; case 32: AR_CALL_L(DT, 32); break; default: AR_CALL_L(DT, 64); }
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
#undef AR_CALL_L
    return launch_status();
}

extern "C" int ar_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean_out, float* rstd_out, int64_t rows,
                                int hidden, float eps, int dt, ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    if (hidden <= 0 || hidden % kEPT || hidden > 16 * kWave * kEPT || !w) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((rows + 3) / 4);
    // chunks of 8 per lane the row needs (2, 4, 8 or 16): the smallest instantiation keeps the register count, hence the number of
    // rows in flight per CU, where a short row (OPT-125M: 768) needs it -- the kernel is a chain of dependent latencies
    const int maxc = hidden <= 2 * kWave * kEPT ? 2 : hidden <= 4 * kWave * kEPT ? 4 : hidden <= 8 * kWave * kEPT ? 8 : 16;
#define AR_CALL_C(DT, C) AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_layernorm_fwd<DT, C>), grid, kTPB, 0, st, x, w, b, y, mean_out, rstd_out, rows, hidden, eps)
#define AR_CALL(DT) switch (maxc) { case 2: AR_CALL_C(DT, 2); break; case 4: AR_CALL_C(DT, 4); break; case 8: AR_CALL_C(DT, 8); break; default: AR_CALL_C(DT, 16); }
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL_C
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, const void* dres,
                                void* dx, int64_t rows, int hidden, int dt, ar_stream_t stream) {
    if (rows <= 0) return AR_OK;
    if (hidden <= 0 || hidden % kEPT || hidden > 16 * kWave * kEPT || !mean || !rstd || !w) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((rows + 3) / 4);
    const int maxc = hidden <= 2 * kWave * kEPT ? 2 : hidden <= 4 * kWave * kEPT ? 4 : hidden <= 8 * kWave * kEPT ? 8 : 16;
#define AR_CALL_C(DT, C) AR_LAUNCH_PROF(AR_PROF_NORM, rows, (k_layernorm_bwd<DT, C>), grid, kTPB, 0, st, dy, x, w, mean, rstd, dres, dx, rows, hidden)
#define AR_CALL(DT) switch (maxc) { case 2: AR_CALL_C(DT, 2); break; case 4: AR_CALL_C(DT, 4); break; case 8: AR_CALL_C(DT, 8); break; default: AR_CALL_C(DT, 16); }
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL_C
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_transpose16(const void* src, void* dst, int64_t rows, int64_t cols, ar_stream_t stream) {
    if (rows <= 0 || cols <= 0) return AR_OK;
    if (rows % kTrTile || cols % kTrTile || !src || !dst || (rows / kTrTile) * (cols / kTrTile) > 0x7fffffffLL) return AR_ERR_UNSUPPORTED;
    const int tiles_c = (int)(cols / kTrTile);
    hipLaunchKernelGGL(k_transpose16, (unsigned)((rows / kTrTile) * tiles_c), kTPB, 0, (hipStream_t)stream, (const uint16_t*)src,
                       (uint16_t*)dst, rows, cols, tiles_c);
    return launch_status();
}

extern "C" int ar_swiglu_fwd(const void* gu, int64_t ld, void* a, int64_t rows, int64_t F, int dt, ar_stream_t stream) {
    if (rows <= 0 || F <= 0) return AR_OK;
    if (F % kEPT || ld % kEPT || ld < 2 * F) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(rows * (F / kEPT));
#define AR_CALL(DT) AR_LAUNCH_PROF(AR_PROF_SWIGLU, rows, (k_swiglu_fwd<DT>), grid, kTPB, 0, st, gu, ld, a, rows, F)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_swiglu_bwd(const void* da, void* gu, int64_t ld, int64_t rows, int64_t F, int dt, ar_stream_t stream) {
    if (rows <= 0 || F <= 0) return AR_OK;
    if (F % kEPT || ld % kEPT || ld < 2 * F) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(rows * (F / kEPT));
#define AR_CALL(DT) AR_LAUNCH_PROF(AR_PROF_SWIGLU, rows, (k_swiglu_bwd<DT>), grid, kTPB, 0, st, da, gu, ld, rows, F)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_rope_fwd(const void* qkv, int64_t ld, const void* cos, const void* sin, int64_t cs_batch_stride, void* q, void* k,
                           void* v, int64_t tokens, int64_t seq, int hq, int hkv, int d, int dt, ar_stream_t stream) {
    if (tokens <= 0) return AR_OK;
    if (d % (2 * kEPT) || hkv <= 0 || hq % hkv || ld % kEPT || seq <= 0 || tokens % seq) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(tokens * (hq + 2 * hkv) * (d / (2 * kEPT)));
#define AR_CALL(DT) AR_LAUNCH_PROF(AR_PROF_ROPE, tokens, (k_rope_fwd<DT>), grid, kTPB, 0, st, qkv, ld, cos, sin, cs_batch_stride, q, k, v, tokens, seq, hq, hkv, d)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_rope_bwd(const void* dq, const void* dk, const void* dv, const void* cos, const void* sin, int64_t cs_batch_stride,
                           void* dqkv, int64_t ld, int64_t tokens, int64_t seq, int hq, int hkv, int d, int dt, ar_stream_t stream) {
    if (tokens <= 0) return AR_OK;
    if (d % (2 * kEPT) || hkv <= 0 || hq % hkv || ld % kEPT || seq <= 0 || tokens % seq) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(tokens * (hq + 2 * hkv) * (d / (2 * kEPT)));
#define AR_CALL(DT) AR_LAUNCH_PROF(AR_PROF_ROPE, tokens, (k_rope_bwd<DT>), grid, kTPB, 0, st, dq, dk, dv, cos, sin, cs_batch_stride, dqkv, ld, tokens, seq, hq, hkv, d)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_moe_expand(const void* src, const int64_t* tok, const float* scale, void* out, int64_t rows, int64_t H, int dt,
                             ar_stream_t stream) {
    if (rows <= 0 || H <= 0) return AR_OK;
    if (H % kEPT || !tok) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(rows * (H / kEPT));
#define AR_CALL(DT) hipLaunchKernelGGL((k_moe_expand<DT>), grid, kTPB, 0, st, src, tok, scale, out, rows, H)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_moe_combine(const void* D, const int64_t* pos, const float* w, const void* res, void* out, int64_t T, int64_t H, int K,
                              int dt, ar_stream_t stream) {
    if (T <= 0 || H <= 0) return AR_OK;
    if (H % kEPT || K <= 0 || !pos) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid1d(T * (H / kEPT));
#define AR_CALL(DT) hipLaunchKernelGGL((k_moe_combine<DT>), grid, kTPB, 0, st, D, pos, w, res, out, T, H, K)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}

extern "C" int ar_moe_rowdot(const void* A, const int64_t* tok, const void* B, float* out, int64_t rows, int64_t H, int dt, ar_stream_t stream) {
    if (rows <= 0 || H <= 0) return AR_OK;
    if (H % kEPT || !tok) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)((rows + (kTPB / kWave) - 1) / (kTPB / kWave));
#define AR_CALL(DT) hipLaunchKernelGGL((k_moe_rowdot<DT>), grid, kTPB, 0, st, A, tok, B, out, rows, H)
    AR_DT_SWITCH2(dt, AR_CALL)
#undef AR_CALL
    return launch_status();
}
