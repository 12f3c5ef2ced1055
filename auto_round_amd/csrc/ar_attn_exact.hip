// ar_attn_exact.hip -- flash-attention forward for gfx950 with THE LIBRARY'S BITS: the output and log-sum-exp rows that torch's
// scaled_dot_product_attention(q, k, v, attn_mask=<additive bias>) returns on this stack (torch 2.10.0+rocm7.0 -> AOTriton 0.11.1
// `attn_fwd`, the "efficient attention with bias" path), value for value.
//
// replaces: the attention forward of the decoder block on the bit-identical (`exact_rounding`) paths -- transformers'
//           sdpa_attention_forward (transformers/integrations/sdpa_attention.py) -> F.scaled_dot_product_attention with the
//           calibration flow's 0 / 1 additive mask (auto_round/calibration/llm.py:360-402 + inputs.py:100-107), reached through the
//           reference's block_forward (auto_round/compressors/utils.py:109-172).  The library kernel takes 1.61 ms per call at
//           Llama-3-8B's minibatch (8 x 32 x 2048 x 128) and 0.40 ms at OPT-125M's (8 x 12 x 2048 x 64): 64 x 32 / 128 x 64 tiles on
//           two / four waves, the score tile converted through LDS twice per step (profiles/r06_aotriton_configs.txt).
//
// What decides the library's bits was read off the gfx950 code objects torch ships (torch/lib/aotriton.images/amd-gfx950/flash/
// attn_fwd/*.aks2: LZMA archives of one code object per tuned configuration; the configuration torch picks for the two problems was
// identified from the launch's grid / workgroup / scratch sizes, tools/aotriton_images.py + tools/gpu/r06_aotriton_configs.py):
//   * S^T = K Q^T per key block with v_mfma_f32_32x32x16_bf16, accumulator from ZERO, d ascending in steps of 16; a-operand = key
//     rows, b-operand = the query row of the lane -- the roles (and the k-slot <-> d mapping: 8 consecutive d per lane half) of
//     csrc/ar_attn.hip's product, which is why this kernel is that kernel with other arithmetic around the MFMAs;
//   * x = fl(s * qk_scale) + bias2, two roundings; qk_scale = fl(sm_scale * fl32(log2 e)) computed in fp32;
//     bias2 = bf16(bias * bf16(log2 e)) -- the bias is scaled in ITS OWN type, bf16 (log2 e -> 1.4453125);
//   * online softmax per KEY BLOCK of BN keys -- 32 at head size 64 (BLOCK_M 64, BLOCK_N 32, 2 waves), 64 at head size 128
//     (BLOCK_M 128, BLOCK_N 64, 4 waves): m' = max(m, max x); p = exp2(x - m') as v_sub + v_exp_f32; alpha = exp2(m - m');
//     acc = acc * alpha; l = fma(l, alpha, l_blk);
//   * l_blk, the block's row sum, is NOT summed in the MFMA accumulator layout: the library converts p to the bias tile's load layout
//     (8 consecutive keys per lane, BN / 8 neighbouring lanes per query row) and reduces there -- 8 keys in ascending order within
//     a lane, then the lanes pairwise over lane-xor 4, 2, 1 (head size 64: 2, 1):
//         s_g = ((((((p[8g] + p[8g+1]) + p[8g+2]) + p[8g+3]) + p[8g+4]) + p[8g+5]) + p[8g+6]) + p[8g+7]
//         BN = 32:  l_blk = (s0 + s2) + (s1 + s3)        BN = 64:  l_blk = ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7))
//     Here p stays in the accumulator layout (lane half h of a query holds keys 8 j + 4 h + i): lane half 0 sums its four keys of
//     every group, hands the partial to lane half 1 (v_permlane32_swap), which continues the chain with its own four -- the same
//     additions in the same order, one cross-lane move per 8 keys;
//   * O^T += V^T P^T with p rounded to bf16 (RNE), the accumulator registers adopted as the k order of the product (the library's
//     own shortcut, and ar_attn.hip's); keys ascending in steps of 16;
//   * out = acc * (1 / l) with the correctly rounded reciprocal, lse = (m + v_log_f32(l)) * fl32(ln 2).
// The additive mask is the calibration flow's STRUCTURED one (ar_attn.hip: bias_in where `key <= query and key < valid_len`,
// bias_out elsewhere, both finite), so no [S, S] operand is read.  K / V may be grouped (kv_rep query heads per key head):
// repeat_kv only copies, the product reads the un-repeated rows.
// Built with -ffp-contract=off like the rest of the library: every fma below is written out.
#include "ar_common.hpp"
#include <cstring>
#include <type_traits>

namespace ar {

typedef short xs16x4_t __attribute__((ext_vector_type(4)));
typedef short xs16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x8_t __attribute__((ext_vector_type(8)));
typedef float xf32x16_t __attribute__((ext_vector_type(16)));

constexpr int XK = 64;               // keys per staged tile

template <int D>
__device__ __forceinline__ int xattn_swz(int r) {          // csrc/ar_attn.hip attn_swz
    if constexpr (D == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    else return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
}

// v_permlane32_swap of a value with itself: lo <- {x[0:31], x[0:31]}, hi <- {x[32:63], x[32:63]}
__device__ __forceinline__ void xhalves(float x, float& lo, float& hi) {
    lo = x; hi = x;
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(lo), "+v"(hi));
}

struct XAttnArgs {
    const uint16_t* Q; const uint16_t* K; const uint16_t* V;
    uint16_t* O;                                                   // [B, S, H, D] token-major, contiguous
    float* LSE;                                                    // [B, H, S]
    int B, S, H, kv_rep;
    int64_t q_bs, q_hs, q_ts;                                      // element strides of Q: batch, head, token
    int64_t k_bs, k_hs, k_ts, v_bs, v_hs, v_ts;                    // ... of K and V (head = query head / kv_rep)
    float qk_scale;                                                // fl(sm_scale * fl32(log2 e))
    float bias_in2, bias_out2;                                     // bf16(bias * bf16(log2 e)) inside / outside the kept region
    int valid_len;
};

template <int WAVES, int AD, int BN>
__global__ __launch_bounds__(64 * WAVES, 2) void k_xattn_fwd(XAttnArgs a) {
    constexpr int AROW = AD * 2;
    constexpr int ATILE = XK * AROW;
    constexpr int ABUF = 2 * ATILE;
    constexpr int NKS = AD / 16;
    constexpr int ND = AD / 32;
    constexpr int AQ = 32 * WAVES;
    constexpr int RPW = XK / WAVES;
    constexpr int RPI = 1024 / AROW;
    constexpr int CPR = AROW / 16;
    constexpr int NP = RPW / RPI;
    static_assert(NP >= 1, "a wave stages at least one DMA instruction per operand");
    static_assert(BN == 32 || BN == 64, "key block of the online softmax");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lq = lane & 31;
    const int n_qt = a.S / AQ;
    const int n_bh = a.B * a.H;
    int qt, bh;
    if ((n_bh & 7) == 0) {               // whole heads per XCD (ar_attn.hip)
        const int j = blockIdx.x >> 3, per = n_bh >> 3;
        bh = (j % per) * 8 + (int)(blockIdx.x & 7);
        qt = j / per;
    } else {
        qt = (int)(blockIdx.x % n_qt);
        bh = blockIdx.x / n_qt;
    }
    const int b = bh / a.H, head = bh % a.H, kvh = head / a.kv_rep;
    const int q0 = qt * AQ;
    const uint16_t* Qb = a.Q + (int64_t)b * a.q_bs + (int64_t)head * a.q_hs;
    const uint16_t* Kb = a.K + (int64_t)b * a.k_bs + (int64_t)kvh * a.k_hs;
    const uint16_t* Vb = a.V + (int64_t)b * a.v_bs + (int64_t)kvh * a.v_hs;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;

    const int myq = q0 + 32 * wave + lq;
    xbf16x8_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const uint4 r = *reinterpret_cast<const uint4*>(Qb + (int64_t)myq * a.q_ts + 16 * ks + 8 * h);
        qf[ks] = __builtin_bit_cast(xbf16x8_t, r);
    }

    const int drow = lane / CPR, pchunk = lane % CPR;
    uint32_t koff[NP], voff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = RPW * wave + RPI * p + drow;
        koff[p] = (uint32_t)(r * a.k_ts + (pchunk ^ xattn_swz<AD>(r)) * 8);
        voff[p] = (uint32_t)(r * a.v_ts + (pchunk ^ xattn_swz<AD>(r)) * 8);
    }
    auto issue_tile = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) {
            const int p = j % NP;
            const uint16_t* T = j < NP ? Kb + (int64_t)kt * XK * a.k_ts + koff[p] : Vb + (int64_t)kt * XK * a.v_ts + voff[p];
            const uint32_t dst = lds0 + buf * ABUF + (j < NP ? 0 : ATILE) + (RPW * wave + RPI * p) * AROW;
            __builtin_amdgcn_global_load_lds((const void*)T, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
        }
    };

    uint32_t kA[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kA[ks] = lds0 + lq * AROW + ((uint32_t)((2 * ks + h) ^ xattn_swz<AD>(lq)) << 4);
    const int gi = lane & 15, gg = lane >> 4;
    const int vrow = 4 * h + (gi >> 2);
    const int vcol0 = 16 * (gg & 1) + 4 * (gi & 3);
    uint32_t vAlo[ND], vAhi[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        const int col = 32 * dt + vcol0;
        vAlo[dt] = lds0 + vrow * AROW + ((uint32_t)((col >> 3) ^ xattn_swz<AD>(vrow)) << 4) + (col & 7) * 2;
        vAhi[dt] = lds0 + (vrow + 8) * AROW + ((uint32_t)((col >> 3) ^ xattn_swz<AD>(vrow + 8)) << 4) + (col & 7) * 2;
    }

    xf32x16_t o[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;     // (the library starts at -3.4e38 / 1.0: alpha of the first block is 0 either way)

#define XA_PIN() __builtin_amdgcn_sched_barrier(0)
#define XA_KREAD(KS, T) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(kf[(KS) & 3][T]) : "v"(kA[KS]), "n"(BUF * ABUF + (T) * 32 * AROW) : "memory")
#define XA_VREAD(ST)                                                                                                     \
    _Pragma("unroll") for (int dt = 0; dt < ND; ++dt)                                                                    \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"                        \
                     : "=&v"(vlo[(ST) & 1][dt]), "=&v"(vhi[(ST) & 1][dt])                                                \
                     : "v"(vAlo[dt]), "v"(vAhi[dt]), "n"(BUF * ABUF + ATILE + (ST) * 16 * AROW) : "memory");
    auto tile = [&](auto bufc, int kt) {
        constexpr int BUF = decltype(bufc)::value;
        const int k0 = kt * XK;
        xf32x16_t s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        u32x4_t kf[4][2];
        XA_KREAD(0, 0); XA_KREAD(0, 1); XA_KREAD(1, 0); XA_KREAD(1, 1); XA_KREAD(2, 0); XA_KREAD(2, 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 3 < NKS) { XA_KREAD(ks + 3, 0); XA_KREAD(ks + 3, 1); }
            const int ahead = (ks + 3 < NKS ? ks + 3 : NKS - 1) - ks;
            if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, kf[ks & 3][0]), qf[ks], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, kf[ks & 3][1]), qf[ks], s[1], 0, 0, 0);
            XA_PIN();
        }
        xs16x4_t vlo[2][ND], vhi[2][ND];
        XA_VREAD(0)
        XA_VREAD(1)
        XA_PIN();
        const bool plain = k0 + XK - 1 <= q0 + 32 * wave && k0 + XK <= a.valid_len;      // wave-uniform: every pair of the tile is kept
        // x = fl(s * qk_scale) + bias2 (two roundings)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                const float bias = (plain || (key <= myq && key < a.valid_len)) ? a.bias_in2 : a.bias_out2;
                const float x = s[t][r] * a.qk_scale;
                s[t][r] = x + bias;
            }
        // one online-softmax step over the sub-tiles [T0, T1)
        // one online-softmax step over the sub-tiles [T0, T1) (compile-time constants after inlining)
        auto step = [&](const int T0, const int T1) __attribute__((always_inline)) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = T0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
            float mlo, mhi;
            xhalves(mx, mlo, mhi);
            const float m_new = fmaxf(m_run, fmaxf(mlo, mhi));
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int t = T0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m_new);
            // block row sum in the library's order (header): group g = 8 consecutive keys = (sub-tile t, j) of both lane halves
            float sg[8];
#pragma unroll
            for (int g = 0; g < 4 * (T1 - T0); ++g) {
                const int t = T0 + (g >> 2), j = g & 3;
                const float part = ((s[t][4 * j] + s[t][4 * j + 1]) + s[t][4 * j + 2]) + s[t][4 * j + 3];
                float lo, hi;
                xhalves(part, lo, hi);                                     // lo = lane half 0's partial, on both halves
                sg[g] = (((lo + s[t][4 * j]) + s[t][4 * j + 1]) + s[t][4 * j + 2]) + s[t][4 * j + 3];       // meaningful on lane half 1
            }
            float l_blk;
            if (T1 - T0 == 1) l_blk = (sg[0] + sg[2]) + (sg[1] + sg[3]);
            else l_blk = ((sg[0] + sg[4]) + (sg[2] + sg[6])) + ((sg[1] + sg[5]) + (sg[3] + sg[7]));
            l_run = __builtin_fmaf(l_run, alpha, l_blk);                   // (lane half 1 carries the row's l)
            m_run = m_new;
            if (__any(alpha != 1.0f)) {
                typedef float xf32x2 __attribute__((ext_vector_type(2)));
                const xf32x2 a2 = {alpha, alpha};
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const xf32x2 v = xf32x2{o[dt][r], o[dt][r + 1]} * a2;
                        o[dt][r] = v.x; o[dt][r + 1] = v.y;
                    }
            }
        };
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st == 0) { step(0, BN == 64 ? 2 : 1); XA_PIN(); }
            if (BN == 32 && st == 2) { step(1, 2); XA_PIN(); }
            const int t = st >> 1, s2 = st & 1;
            xs16x8_t pb;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const uint32_t w = pack_bf16x2(s[t][8 * s2 + e], s[t][8 * s2 + e + 1]);
                pb[e] = (short)(w & 0xffffu);
                pb[e + 1] = (short)(w >> 16);
            }
            if constexpr (ND == 4) {
                if (st < 3) {
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vlo[st & 1][0]), "+v"(vhi[st & 1][0]), "+v"(vlo[st & 1][1]), "+v"(vhi[st & 1][1]),
                                 "+v"(vlo[st & 1][2]), "+v"(vhi[st & 1][2]), "+v"(vlo[st & 1][3]), "+v"(vhi[st & 1][3])::"memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[st & 1][0]), "+v"(vhi[st & 1][0]), "+v"(vlo[st & 1][1]), "+v"(vhi[st & 1][1]),
                                 "+v"(vlo[st & 1][2]), "+v"(vhi[st & 1][2]), "+v"(vlo[st & 1][3]), "+v"(vhi[st & 1][3])::"memory");
                }
            } else {
                if (st < 3) {
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vlo[st & 1][0]), "+v"(vhi[st & 1][0]), "+v"(vlo[st & 1][1]), "+v"(vhi[st & 1][1])::"memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[st & 1][0]), "+v"(vhi[st & 1][0]), "+v"(vlo[st & 1][1]), "+v"(vhi[st & 1][1])::"memory");
                }
            }
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const xs16x8_t va = __builtin_shufflevector(vlo[st & 1][dt], vhi[st & 1][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, va), __builtin_bit_cast(xbf16x8_t, pb), o[dt], 0, 0, 0);
            }
            XA_PIN();
            if (st + 2 < 4) { XA_VREAD(st + 2) }
        }
    };

    const int n_kt = a.S / XK;
    issue_tile(0, 0);
    for (int kt = 0; kt < n_kt; kt += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_tile(kt + 1, 1);
        tile(std::integral_constant<int, 0>{}, kt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < n_kt) issue_tile(kt + 2, 0);
        tile(std::integral_constant<int, 1>{}, kt + 1);
    }
#undef XA_KREAD
#undef XA_VREAD
#undef XA_PIN
    float llo, l_tot;
    xhalves(l_run, llo, l_tot);                                            // lane half 1's l, on both halves
    const float inv = 1.0f / l_tot;
    uint16_t* orow = a.O + ((int64_t)(b * a.S + myq) * a.H + head) * AD;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 w;
            w.x = pack_bf16x2(o[dt][4 * j + 0] * inv, o[dt][4 * j + 1] * inv);
            w.y = pack_bf16x2(o[dt][4 * j + 2] * inv, o[dt][4 * j + 3] * inv);
            *reinterpret_cast<uint2*>(orow + 32 * dt + 8 * j + 4 * h) = w;
        }
    if (h == 0) a.LSE[((int64_t)b * a.H + head) * a.S + myq] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.693147182464599609375f;
}

// bf16(bias * bf16(log2 e)), as a float: the library scales the additive mask in its own type
static float bias_log2e_bf16(float bias) {
    const float prod = bias * 1.4453125f;                // exact for a bf16-representable bias (8 x 8 significant bits)
    uint32_t u; memcpy(&u, &prod, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float r; memcpy(&r, &u, 4);
    return r;
}

}  // namespace ar

using namespace ar;

extern "C" int ar_attn_fwd_exact(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H, int64_t D,
                                 int64_t kv_rep, float scale, float bias_in, float bias_out, int64_t valid_len, int64_t q_bs, int64_t q_hs,
                                 int64_t q_ts, int64_t k_bs, int64_t k_hs, int64_t k_ts, int64_t v_bs, int64_t v_hs, int64_t v_ts,
                                 ar_stream_t stream) {
    const int64_t q_strides[3] = {q_bs, q_hs, q_ts}, k_strides[3] = {k_bs, k_hs, k_ts}, v_strides[3] = {v_bs, v_hs, v_ts};
    if ((D != 128 && D != 64) || S % 128 || B <= 0 || H <= 0 || S <= 0 || kv_rep < 1 || H % kv_rep) return AR_ERR_UNSUPPORTED;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) return AR_ERR_UNSUPPORTED;
    if (!(bias_in == bias_in) || !(bias_out == bias_out) || fabsf(bias_in) > 1e4f || fabsf(bias_out) > 1e4f || valid_len < 1 || valid_len > S)
        return AR_ERR_UNSUPPORTED;
    {   // the mask values must be bf16 numbers (the library reads a bf16 mask)
        uint32_t u0, u1; memcpy(&u0, &bias_in, 4); memcpy(&u1, &bias_out, 4);
        if ((u0 | u1) & 0xffffu) return AR_ERR_UNSUPPORTED;
    }
    const int64_t* st[3] = {q_strides, k_strides, v_strides};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            if (st[i][j] < 0 || (st[i][j] % 8)) return AR_ERR_UNSUPPORTED;
    if (64 * k_strides[2] > 0x7fffffffLL || 64 * v_strides[2] > 0x7fffffffLL) return AR_ERR_UNSUPPORTED;
    XAttnArgs a;
    a.Q = (const uint16_t*)Q; a.K = (const uint16_t*)K; a.V = (const uint16_t*)V; a.O = (uint16_t*)O; a.LSE = LSE;
    a.B = (int)B; a.S = (int)S; a.H = (int)H; a.kv_rep = (int)kv_rep;
    a.q_bs = q_strides[0]; a.q_hs = q_strides[1]; a.q_ts = q_strides[2];
    a.k_bs = k_strides[0]; a.k_hs = k_strides[1]; a.k_ts = k_strides[2];
    a.v_bs = v_strides[0]; a.v_hs = v_strides[1]; a.v_ts = v_strides[2];
    a.qk_scale = scale * 1.44269502162933349609375f;        // fp32 product with fl32(log2 e) = 0x3fb8aa3b, as the library computes it
    a.bias_in2 = bias_log2e_bf16(bias_in); a.bias_out2 = bias_log2e_bf16(bias_out); a.valid_len = (int)valid_len;
    constexpr int LDS128 = 4 * XK * 128 * 2, LDS64 = 4 * XK * 64 * 2;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<4, 128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<8, 128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
    }
    hipStream_t s = (hipStream_t)stream;
    if (D == 128) {
        if (S % 256 == 0) hipLaunchKernelGGL((k_xattn_fwd<8, 128, 64>), (int)(B * H * (S / 256)), 512, LDS128, s, a);
        else hipLaunchKernelGGL((k_xattn_fwd<4, 128, 64>), (int)(B * H * (S / 128)), 256, LDS128, s, a);
    } else {
        if (S % 256 == 0) hipLaunchKernelGGL((k_xattn_fwd<8, 64, 32>), (int)(B * H * (S / 256)), 512, LDS64, s, a);
        else hipLaunchKernelGGL((k_xattn_fwd<4, 64, 32>), (int)(B * H * (S / 128)), 256, LDS64, s, a);
    }
    return launch_status();
}
