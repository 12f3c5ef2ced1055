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
//   * online softmax per KEY BLOCK of BN keys -- at the tuning minibatch (S = 2048) 32 at head size 64 (BLOCK_M 64, BLOCK_N 32, 2 waves)
//     and 64 at head size 128 (BLOCK_M 128, BLOCK_N 64, 4 waves); the library picks the configuration by head size and sequence length,
//     and BN is the only number of it that reaches the bits, so it is a launch parameter (`key_block`: 16 / 32 / 64; measured table:
//     ops.attn_key_block_guess, profiles/r06_attn_exact_keyblock_probe.json) -- m' = max(m, max x); p = exp2(x - m') as v_sub +
//     v_exp_f32; alpha = exp2(m - m'); acc = acc * alpha; l = fma(l, alpha, l_blk);
//   * l_blk, the block's row sum, is NOT summed in the MFMA accumulator layout: the library converts p to the bias tile's load layout
//     (8 consecutive keys per lane, BN / 8 neighbouring lanes per query row) and reduces there -- 8 keys in ascending order within
//     a lane, then the lanes pairwise over lane-xor 4, 2, 1 (head size 64: 2, 1):
//         s_g = ((((((p[8g] + p[8g+1]) + p[8g+2]) + p[8g+3]) + p[8g+4]) + p[8g+5]) + p[8g+6]) + p[8g+7]
//         BN = 16:  l_blk = s0 + s1      BN = 32:  l_blk = (s0 + s2) + (s1 + s3)      BN = 64:  l_blk = ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7))
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
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef float xf32x4_t __attribute__((ext_vector_type(4)));

constexpr int XK = 64;               // keys per staged tile

template <int D>
__device__ __forceinline__ int xattn_swz(int r) {          // csrc/ar_attn.hip attn_swz
    if constexpr (D == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    else return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
}

// v_permlane32_swap of a value with itself: lo <- {x[0:31], x[0:31]}, hi <- {x[32:63], x[32:63]}
__device__ __forceinline__ void xhalves(float x, float& lo, float& hi) {
    lo = x; hi = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}

// blockIdx -> (block of own rows, batch * heads + head).  Workgroups go to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its
// own 4 MB L2: the workgroups that stream the SAME rows -- forward / dQ: the kv_rep query heads x all row blocks of one key head
// (K, V: 1 MB at S = 2048, head size 128); dK / dV: all row blocks of one query head (Q, dO) -- are made neighbours in time on ONE
// XCD, so a stream crosses the fabric once per group instead of once per workgroup.  Measured (profiles/r06_attn_xcd_map_ab.json):
// the backward 5.07 -> 4.67 ms at 64 query heads over 8 key heads (Llama-3-70B's block), 2.51 -> 2.33 ms at 32 heads without
// grouping, the forward 0.81 -> 0.73 there; neutral at Llama-3-8B's 32 / 8 and at head size 64, whose streams already fit the
// 256 MB Infinity Cache behind the L2s -- the kernels are not bound by this traffic.
__device__ __forceinline__ void xattn_map(int bid, int n_ob, int B, int H, int kv_rep, bool by_kv, bool legacy, int& ob, int& bh) {
    const int members = by_kv ? kv_rep * n_ob : n_ob;
    const int groups = by_kv ? B * (H / kv_rep) : B * H;
    if (!legacy && (groups & 7) == 0) {
        const int j = bid >> 3, g = (j / members) * 8 + (bid & 7), m = j % members;
        if (by_kv) {
            const int hk = H / kv_rep;
            bh = (g / hk) * H + (g % hk) * kv_rep + m % kv_rep;
            ob = m / kv_rep;
        } else {
            bh = g;
            ob = m;
        }
    } else if (((B * H) & 7) == 0) {
        const int j = bid >> 3, per = (B * H) >> 3;
        bh = (j % per) * 8 + (bid & 7);
        ob = j / per;
    } else {
        ob = bid % n_ob;
        bh = bid / n_ob;
    }
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
    int map_legacy;                                                // the blockIdx mapping before xattn_map (config bit 32; A/B)
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
    static_assert(BN == 16 || BN == 32 || BN == 64, "key block of the online softmax");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lq = lane & 31;
    const int n_qt = a.S / AQ;
    int qt, bh;
    xattn_map((int)blockIdx.x, n_qt, a.B, a.H, a.kv_rep, true, a.map_legacy != 0, qt, bh);
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
        XA_VREAD(0)                                   // (the second step's fragments are read behind the softmax: registers)
        XA_PIN();
        // wave-uniform shortcuts: every (query, key) pair of the tile inside the kept region / every pair outside it (a tile above the
        // diagonal or behind the valid keys) -- only the tiles the region's edge crosses pay for a per-element choice
        const bool plain = k0 + XK - 1 <= q0 + 32 * wave && k0 + XK <= a.valid_len;
        const bool all_out = k0 > q0 + 32 * wave + 31 || k0 >= a.valid_len;
        // x = fl(s * qk_scale) + bias2 (two roundings)
        if (plain || all_out) {
            const float bias = plain ? a.bias_in2 : a.bias_out2;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const xf32x2 x = xf32x2{s[t][r], s[t][r + 1]} * xf32x2{a.qk_scale, a.qk_scale};
                    const xf32x2 y = x + xf32x2{bias, bias};
                    s[t][r] = y.x; s[t][r + 1] = y.y;
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                    const float bias = (key <= myq && key < a.valid_len) ? a.bias_in2 : a.bias_out2;
                    const float x = s[t][r] * a.qk_scale;
                    s[t][r] = x + bias;
                }
        }
        // one online-softmax step over the sub-tiles [T0, T1)
        // one online-softmax step over the 8-key groups [G0, G1) of the tile (group g = sub-tile g >> 2, accumulator rows 4 (g & 3) ..
        // + 3 of both lane halves; compile-time constants after inlining)
        auto step = [&](const int G0, const int G1) __attribute__((always_inline)) {
            float mx = -INFINITY;
#pragma unroll
            for (int g = G0; g < G1; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[g >> 2][4 * (g & 3) + i]);
            float mlo, mhi;
            xhalves(mx, mlo, mhi);
            const float m_new = fmaxf(m_run, fmaxf(mlo, mhi));
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int g = G0; g < G1; ++g)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const int t = g >> 2, r = 4 * (g & 3) + i;
                    const xf32x2 d = xf32x2{s[t][r], s[t][r + 1]} - xf32x2{m_new, m_new};
                    s[t][r] = __builtin_amdgcn_exp2f(d.x);
                    s[t][r + 1] = __builtin_amdgcn_exp2f(d.y);
                }
            // block row sum in the library's order (header): group g = 8 consecutive keys = (sub-tile t, j) of both lane halves
            float sg[8];
#pragma unroll
            for (int g = G0; g < G1; g += 2) {                             // groups g, g + 1 as the two halves of packed adds
                const int t = g >> 2, j = g & 3;
                xf32x2 part = xf32x2{s[t][4 * j], s[t][4 * j + 4]} + xf32x2{s[t][4 * j + 1], s[t][4 * j + 5]};
                part = part + xf32x2{s[t][4 * j + 2], s[t][4 * j + 6]};
                part = part + xf32x2{s[t][4 * j + 3], s[t][4 * j + 7]};
                float lo0, hi0, lo1, hi1;
                xhalves(part.x, lo0, hi0);                                 // lo = lane half 0's partial, on both halves
                xhalves(part.y, lo1, hi1);
                xf32x2 c = xf32x2{lo0, lo1} + xf32x2{s[t][4 * j], s[t][4 * j + 4]};       // meaningful on lane half 1
                c = c + xf32x2{s[t][4 * j + 1], s[t][4 * j + 5]};
                c = c + xf32x2{s[t][4 * j + 2], s[t][4 * j + 6]};
                c = c + xf32x2{s[t][4 * j + 3], s[t][4 * j + 7]};
                sg[g - G0] = c.x; sg[g - G0 + 1] = c.y;
            }
            float l_blk;
            if (G1 - G0 == 2) l_blk = sg[0] + sg[1];
            else if (G1 - G0 == 4) l_blk = (sg[0] + sg[2]) + (sg[1] + sg[3]);
            else l_blk = ((sg[0] + sg[4]) + (sg[2] + sg[6])) + ((sg[1] + sg[5]) + (sg[3] + sg[7]));
            l_run = __builtin_fmaf(l_run, alpha, l_blk);                   // (lane half 1 carries the row's l)
            m_run = m_new;
            if (__any(alpha != 1.0f)) {
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
            if ((16 * st) % BN == 0) { step(2 * st, 2 * st + BN / 8); XA_PIN(); }      // the key block of the softmax starts here
            if (st == 0) { XA_VREAD(1) XA_PIN(); }
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

// launch forms (equal results; measured A/B, profiles/r06_attn_exact_waves_ab.json): bits 0-1 forward, bits 2-3 backward -- 0 default
// (forward: 8 waves at head size 128, 4 at 64; backward: head size 128 the fused key-side kernel on 4 waves + the 8-wave query-side
// kernel, head size 64 4 waves), 1 = workgroups of 4 waves (128 own rows), 2 = workgroups of 8 waves (256 own rows; head size 128: the
// key side as two kernels), 3 (backward) = the fused key-side kernel; bit 4 (16): the fused key-side kernel WITH the hand pipeline (k_xattn_bwd_kv; off by default); bit 5 (32): the blockIdx mapping before xattn_map; bit 6 (64): head size 64's key side on the pipelined kernel (slower: A/B)
static int g_xattn_cfg = 0;
extern "C" int ar_attn_exact_config(int cfg) {
    const int old = g_xattn_cfg;
    if (cfg >= 0) g_xattn_cfg = cfg & 127;
    return old;
}

extern "C" int ar_attn_fwd_exact(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H, int64_t D,
                                 int64_t kv_rep, float scale, float bias_in, float bias_out, int64_t valid_len, int64_t q_bs, int64_t q_hs,
                                 int64_t q_ts, int64_t k_bs, int64_t k_hs, int64_t k_ts, int64_t v_bs, int64_t v_hs, int64_t v_ts,
                                 int64_t key_block, ar_stream_t stream) {
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
    a.map_legacy = (g_xattn_cfg & 32) ? 1 : 0;
    constexpr int LDS128 = 4 * XK * 128 * 2, LDS64 = 4 * XK * 64 * 2;
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<4, 128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<8, 128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<4, 128, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<8, 128, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<4, 128, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        (void)hipFuncSetAttribute((const void*)k_xattn_fwd<8, 128, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
    }
    hipStream_t s = (hipStream_t)stream;
    const int form = g_xattn_cfg & 3;
    const bool eight = S % 256 == 0 && (form == 2 || (form == 0 && D == 128));      // measured (profiles/r06_attn_exact_waves_ab.json)
    if (key_block == 0) key_block = D == 128 ? 64 : 32;       // the library's choice at the tuning minibatch's shape
    if (key_block != 16 && key_block != 32 && key_block != 64) return AR_ERR_UNSUPPORTED;
    const int g8 = (int)(B * H * (S / 256)), g4 = (int)(B * H * (S / 128));
#define XA_LAUNCH(DD, BNN)                                                                                          \
    do {                                                                                                            \
        if (eight) hipLaunchKernelGGL((k_xattn_fwd<8, DD, BNN>), g8, 512, DD == 128 ? LDS128 : LDS64, s, a);        \
        else hipLaunchKernelGGL((k_xattn_fwd<4, DD, BNN>), g4, 256, DD == 128 ? LDS128 : LDS64, s, a);              \
    } while (0)
    if (D == 128) {
        if (key_block == 64) XA_LAUNCH(128, 64);
        else if (key_block == 32) XA_LAUNCH(128, 32);
        else XA_LAUNCH(128, 16);
    } else {
        if (key_block == 64) XA_LAUNCH(64, 64);
        else if (key_block == 32) XA_LAUNCH(64, 32);
        else XA_LAUNCH(64, 16);
    }
#undef XA_LAUNCH
    return launch_status();
}

namespace ar {

// ================================================================================================================================
// BACKWARD with the library's bits: what aten::_scaled_dot_product_efficient_attention_backward returns on this stack (AOTriton
// 0.11.1 `bwd_preprocess` + `bwd_kernel_dk_dv` + `bwd_kernel_dq` with an additive bias), value for value.
//
// replaces: autograd of the attention call above on the bit-identical paths (library: 3.64 + 2.15 + 0.05 ms per call at Llama-3-8B's
//           minibatch, 0.68 + 0.40 + 0.01 ms at OPT-125M's: 64 x 32 / 64 x 64 tiles on two waves, P and dS through LDS).
// Read off the shipped code objects (bwd_kernel_dk_dv 64_32 wave1|2 warp2, bwd_kernel_dq 64_64 wave1|2 warp2, bwd_preprocess 128):
//   * delta[row] = sum_d o * do over BLOCKED lanes: 8 consecutive d per lane, t = o1 do1 (rounded), t = fma(o0, do0, t),
//     t = fma(o_e, do_e, t) for e = 2 .. 7; the D / 8 lanes of a row pairwise over lane-xor D/16 .. 1; + 0.0;
//   * l2[row] = fl(lse * fl32(log2 e));   qk_scale = fl(sm_scale * fl32(log2 e));   bias_scale = 1 / sm_scale (IEEE);
//   * scores: MFMA chain over d ascending (16 per step) whose accumulator STARTS at fl(bias * bias_scale) (fp32);
//   * p = exp2(fma(qk_scale, s, -l2)) -- except, in bwd_kernel_dk_dv only, accumulator register 15 of the key block (keys 27 and 31
//     of every 32): exp2(fl(qk_scale * s) - l2), two roundings (the compiler folded that element's multiply into a packed multiply
//     with the l2 product);
//   * dp = MFMA chain from zero; ds = p * (dp - delta), two roundings; p and ds rounded to bf16 (RNE) for the accumulating products;
//   * dV^T += dO^T P and dK^T += Q^T dS over query blocks ascending, dQ^T += K^T dS over key blocks ascending, 16 rows per MFMA with
//     lane half h holding rows 4 h + i and 8 + 4 h + i of the 16 (i = 0 .. 3) -- the accumulator registers adopted as the k order in
//     bwd_kernel_dq, a k-width-4 dot-operand layout read back from LDS in bwd_kernel_dk_dv: the same k-slot sets either way (what
//     matters to the hardware is WHICH 8 of the 16 k-slots sit in which lane half, not their order inside it:
//     tools/gpu/r06_mfma_kslot_probe.hip);
//   * dq = sm_scale * acc, dk = sm_scale * acc, dv = acc, rounded to bf16.
// The kernels are csrc/ar_attn_bwd.hip's (own row on the lane, two resident b-operands, two streamed tensors through LDS-DMA) with
// that arithmetic; dK / dV are written per QUERY head (the caller sums the kv_rep heads of a group as autograd's expand does).

struct XBwdArgs {
    const uint16_t* Q; const uint16_t* K; const uint16_t* V; const uint16_t* dO;
    const float* L2; const float* Dv;                              // [B, H, S] fp32: fl(lse * log2 e), delta
    uint16_t* dQ; uint16_t* dK; uint16_t* dV;                      // token-major [B, S, H, D] with token strides below
    int B, S, H, kv_rep;
    int64_t q_bs, q_hs, q_ts, k_bs, k_hs, k_ts, v_bs, v_hs, v_ts, o_bs, o_hs, o_ts;      // element strides (batch, head, token); o = dO
    int64_t lddq, lddk, lddv;
    float sm_scale, qk_scale;
    float bias_in_s, bias_out_s;                                   // fl(bias * bias_scale)
    int valid_len;
    int map_legacy;
};

// delta and l2 per (batch, head, token) row (the library's bwd_preprocess order; l2 as bwd_kernel_* compute it)
__global__ __launch_bounds__(kTPB) void k_xattn_bwd_prep(const uint16_t* __restrict__ dO, int64_t do_bs, int64_t do_hs, int64_t do_ts,
                                                          const uint16_t* __restrict__ O, int64_t o_bs, int64_t o_hs, int64_t o_ts,
                                                          const float* __restrict__ lse, float* __restrict__ Dv, float* __restrict__ L2,
                                                          int B, int S, int H, int AD) {
    const int lpr = AD / 8;
    const int64_t row = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / lpr;
    const int part = threadIdx.x % lpr;
    const int64_t rows = (int64_t)B * S * H;
    float s = 0.f;
    int64_t b = 0, sq = 0; int h = 0;
    if (row < rows) {
        const int64_t tok = row / H;
        h = (int)(row % H);
        b = tok / S; sq = tok % S;
        float a[8], c[8];
        unpack8<AR_DT_BF16>(load8_raw<AR_DT_BF16>(dO, b * do_bs + h * do_hs + sq * do_ts + part * 8), a);
        unpack8<AR_DT_BF16>(load8_raw<AR_DT_BF16>(O, b * o_bs + h * o_hs + sq * o_ts + part * 8), c);
        s = a[1] * c[1];
        s = __builtin_fmaf(a[0], c[0], s);
#pragma unroll
        for (int e = 2; e < 8; ++e) s = __builtin_fmaf(a[e], c[e], s);
    }
    for (int m = lpr >> 1; m >= 1; m >>= 1) s = s + __shfl_xor(s, m, kWave);
    s = s + 0.0f;
    if (row < rows && part == 0) {
        const int64_t o = (b * H + h) * S + sq;
        Dv[o] = s;
        L2[o] = lse[o] * 1.44269502162933349609375f;
    }
}

// MODE 0: dQ (own rows = queries; K / V tiles stream).  MODE 1: dK / dV (own rows = keys; Q / dO tiles stream); OUT: 3 = both,
// 1 = dV only, 2 = dK only (head size 128 runs the key side as two kernels: registers, csrc/ar_attn_bwd.hip).
template <int MODE, int WAVES, int AD, int OUT = 3, int OCC = 2>
__global__ __launch_bounds__(64 * WAVES, OCC) void k_xattn_bwd(XBwdArgs a) {
    constexpr bool NEED_S1 = MODE == 0 || (OUT & 2);
    constexpr bool NEED_A0 = MODE == 0 || (OUT & 2);
    constexpr bool NEED_A1 = MODE == 1 && (OUT & 1);
    constexpr int KB = AD == 128 ? 4 : AD / 16;
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
    static_assert(NP >= 1, "a wave stages at least one DMA instruction per tensor");
    static_assert(AD == 64 || (AD == 128 && (MODE == 0 || OUT != 3 || OCC == 1)),
                  "head size 128: the key side as two kernels (register budget), or one wave per SIMD with the whole register file");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lq = lane & 31;
    const int n_ob = a.S / AQ;
    int ob, bh;
    xattn_map((int)blockIdx.x, n_ob, a.B, a.H, a.kv_rep, MODE == 0, a.map_legacy != 0, ob, bh);
    const int b = bh / a.H, head = bh % a.H, kvh = head / a.kv_rep;
    const int o0 = ob * AQ;
    const int myrow = o0 + 32 * wave + lq;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    constexpr int VEC_OFF = 2 * ABUF;

    const uint16_t* Qh = a.Q + (int64_t)b * a.q_bs + (int64_t)head * a.q_hs;
    const uint16_t* Kh = a.K + (int64_t)b * a.k_bs + (int64_t)kvh * a.k_hs;
    const uint16_t* Vh = a.V + (int64_t)b * a.v_bs + (int64_t)kvh * a.v_hs;
    const uint16_t* Oh = a.dO + (int64_t)b * a.o_bs + (int64_t)head * a.o_hs;
    const int64_t ld0 = MODE == 0 ? a.k_ts : a.q_ts, ld1 = MODE == 0 ? a.v_ts : a.o_ts;
    const uint16_t* R0b = MODE == 0 ? Kh : Qh;
    const uint16_t* R1b = MODE == 0 ? Vh : Oh;
    const int64_t lb0 = MODE == 0 ? a.q_ts : a.k_ts, lb1 = MODE == 0 ? a.o_ts : a.v_ts;
    const uint16_t* B0b = MODE == 0 ? Qh : Kh;
    const uint16_t* B1b = MODE == 0 ? Oh : Vh;
    xbf16x8_t bf0[NKS], bf1[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        bf0[ks] = __builtin_bit_cast(xbf16x8_t, *reinterpret_cast<const uint4*>(B0b + (int64_t)myrow * lb0 + 16 * ks + 8 * h));
        if (NEED_S1) bf1[ks] = __builtin_bit_cast(xbf16x8_t, *reinterpret_cast<const uint4*>(B1b + (int64_t)myrow * lb1 + 16 * ks + 8 * h));
    }
    const float* L2row = a.L2 + ((int64_t)b * a.H + head) * a.S;
    const float* Dvrow = a.Dv + ((int64_t)b * a.H + head) * a.S;
    float myL2 = 0.f, myD = 0.f;
    if (MODE == 0) { myL2 = L2row[myrow]; myD = Dvrow[myrow]; }
    if (MODE == 1) {
        float* vL = reinterpret_cast<float*>(lds + VEC_OFF);
        float* vD = vL + a.S;
        for (int i = tid; i < a.S; i += 64 * WAVES) { vL[i] = L2row[i]; vD[i] = Dvrow[i]; }
    }
    // MODE 1: the library's non-fused element -- accumulator register 15 of its key block = keys 27 and 31 of every 32
    const bool quirk = MODE == 1 && (lq == 27 || lq == 31);
    const float qmul = quirk ? 1.0f : a.qk_scale;

    const int drow = lane / CPR, pchunk = lane % CPR;
    uint32_t doff0[NP], doff1[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = RPW * wave + RPI * p + drow;
        doff0[p] = (uint32_t)(r * ld0 + (pchunk ^ xattn_swz<AD>(r)) * 8);
        doff1[p] = (uint32_t)(r * ld1 + (pchunk ^ xattn_swz<AD>(r)) * 8);
    }
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) {
            const int p = j % NP;
            const uint16_t* T = (j < NP ? R0b + (int64_t)t * XK * ld0 + doff0[p] : R1b + (int64_t)t * XK * ld1 + doff1[p]);
            const uint32_t dst = lds0 + buf * ABUF + (j < NP ? 0 : ATILE) + (RPW * wave + RPI * p) * AROW;
            __builtin_amdgcn_global_load_lds((const void*)T, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
        }
    };

    uint32_t rA[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) rA[ks] = lds0 + lq * AROW + ((uint32_t)((2 * ks + h) ^ xattn_swz<AD>(lq)) << 4);
    const int gi = lane & 15, gg = lane >> 4;
    // transposed fragments in the adopted k order (rows 4 h + .. and 8 + 4 h + ..)
    const int vrow = 4 * h + (gi >> 2);
    const int vrow_hi = vrow + 8;
    const int vcol0 = 16 * (gg & 1) + 4 * (gi & 3);
    uint32_t tAlo[ND], tAhi[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        const int col = 32 * dt + vcol0;
        tAlo[dt] = lds0 + vrow * AROW + ((uint32_t)((col >> 3) ^ xattn_swz<AD>(vrow)) << 4) + (col & 7) * 2;
        tAhi[dt] = lds0 + vrow_hi * AROW + ((uint32_t)((col >> 3) ^ xattn_swz<AD>(vrow_hi)) << 4) + (col & 7) * 2;
    }

    xf32x16_t acc0[ND], acc1[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[dt][r] = 0.f; acc1[dt][r] = 0.f; }

#define XB_PIN() __builtin_amdgcn_sched_barrier(0)
#define XB_RREAD(DST, KS, T, REG) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(rA[KS]), "n"(BUF * ABUF + (REG) * ATILE + (T) * 32 * AROW) : "memory")
#define XB_TREAD(LO, HI, DT, ST, REG)                                                                                          \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"                                  \
                 : "=&v"(LO), "=&v"(HI) : "v"(tAlo[DT]), "v"(tAhi[DT]), "n"(BUF * ABUF + (REG) * ATILE + (ST) * 16 * AROW) : "memory")

    auto tile = [&](auto bufc, int t0) {
        constexpr int BUF = decltype(bufc)::value;
        // `plain`: every (query, key) pair between the wave's own rows and this tile is inside the kept region
        const bool plain = MODE == 0 ? (t0 + XK - 1 <= o0 + 32 * wave && t0 + XK <= a.valid_len)
                                     : (t0 >= o0 + 32 * wave + 31 && o0 + 32 * wave + 31 < a.valid_len);
        // ... or every pair outside it (keys after the queries, or invalid keys only)
        const bool all_out = MODE == 0 ? (t0 > o0 + 32 * wave + 31 || t0 >= a.valid_len)
                                       : (o0 + 32 * wave > t0 + XK - 1 || o0 + 32 * wave >= a.valid_len);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            xf32x16_t s0, s1;
            u32x4_t f0[KB], f1[KB];
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                XB_RREAD(f0[ks], ks, t, 0);
                if (NEED_S1) XB_RREAD(f1[ks], ks, t, 1);
            }
            // the score accumulator starts at fl(bias * bias_scale); dp at zero
            if (plain || all_out) {
                const float bias = plain ? a.bias_in_s : a.bias_out_s;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] = bias; s1[r] = 0.f; }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int srow = t0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                    const int query = MODE == 0 ? myrow : srow, key = MODE == 0 ? srow : myrow;
                    s0[r] = (key <= query && key < a.valid_len) ? a.bias_in_s : a.bias_out_s;
                    s1[r] = 0.f;
                }
            }
#pragma unroll
            for (int kb = 0; kb < NKS; kb += KB) {
                if (kb > 0) {
#pragma unroll
                    for (int ks = 0; ks < KB; ++ks) {
                        XB_RREAD(f0[ks], kb + ks, t, 0);
                        if (NEED_S1) XB_RREAD(f1[ks], kb + ks, t, 1);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                XB_PIN();
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, f0[ks]), bf0[kb + ks], s0, 0, 0, 0);
                    if (NEED_S1) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, f1[ks]), bf1[kb + ks], s1, 0, 0, 0);
                }
                if (kb + KB < NKS) XB_PIN();
            }
            constexpr bool LATE_Q = AD == 128;
            xs16x4_t q0lo[2][ND], q0hi[2][ND], q1lo[2][ND], q1hi[2][ND];
#pragma unroll
            for (int s2 = 0; s2 < (LATE_Q ? 1 : 2); ++s2)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    if (NEED_A0) XB_TREAD(q0lo[s2][dt], q0hi[s2][dt], dt, 2 * t + s2, 0);
                    if (NEED_A1) XB_TREAD(q1lo[s2][dt], q1hi[s2][dt], dt, 2 * t + s2, 1);
                }
            XB_PIN();
            // ---- p = exp2(fma(qk_scale, s, -l2)) (the quirk lanes: exp2(fl(qk_scale * s) - l2)); ds = p * (dp - delta)
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(a.qk_scale, s0[r], -myL2));
                    s0[r] = p;
                    const float dd = s1[r] - myD;
                    s1[r] = p * dd;
                }
            } else {
                const float* vL = reinterpret_cast<const float*>(lds + VEC_OFF);
                const float* vD = vL + a.S;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 l4 = *reinterpret_cast<const float4*>(vL + t0 + 32 * t + 4 * h + 8 * j);
                    float4 d4 = {0.f, 0.f, 0.f, 0.f};
                    if (NEED_S1) d4 = *reinterpret_cast<const float4*>(vD + t0 + 32 * t + 4 * h + 8 * j);
                    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * j + i;
                        const float prod = a.qk_scale * s0[r];
                        const float tv = quirk ? prod : s0[r];
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(qmul, tv, -lv[i]));
                        s0[r] = p;
                        if (NEED_S1) {
                            const float dd = s1[r] - dv[i];
                            s1[r] = p * dd;
                        }
                    }
                }
            }
            XB_PIN();
            // ---- bf16 operands of the accumulating products
            uint32_t pk0[8], pk1[8];                   // pk[m] = rows (r = 2 m, 2 m + 1) of P (pk0) and dS (pk1)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (NEED_A1) pk0[m] = pack_bf16x2(s0[2 * m], s0[2 * m + 1]);
                if (NEED_A0) pk1[m] = pack_bf16x2(s1[2 * m], s1[2 * m + 1]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4_t pe, pp;
                if (NEED_A0) { pe.x = pk1[4 * s2]; pe.y = pk1[4 * s2 + 1]; pe.z = pk1[4 * s2 + 2]; pe.w = pk1[4 * s2 + 3]; }
                if (NEED_A1) { pp.x = pk0[4 * s2]; pp.y = pk0[4 * s2 + 1]; pp.z = pk0[4 * s2 + 2]; pp.w = pk0[4 * s2 + 3]; }
                constexpr int INFL = ((NEED_A0 ? 2 : 0) + (NEED_A1 ? 2 : 0)) * ND;
                if (LATE_Q && s2 == 0) {
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        if (NEED_A0) XB_TREAD(q0lo[1][dt], q0hi[1][dt], dt, 2 * t + 1, 0);
                        if (NEED_A1) XB_TREAD(q1lo[1][dt], q1hi[1][dt], dt, 2 * t + 1, 1);
                    }
                }
                if (s2 == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(INFL > 15 ? 15 : INFL) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                XB_PIN();
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    if (NEED_A0) {
                        const xs16x8_t a0 = __builtin_shufflevector(q0lo[s2][dt], q0hi[s2][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc0[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, a0), __builtin_bit_cast(xbf16x8_t, pe), acc0[dt], 0, 0, 0);
                    }
                    if (NEED_A1) {
                        const xs16x8_t a1 = __builtin_shufflevector(q1lo[s2][dt], q1hi[s2][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, a1), __builtin_bit_cast(xbf16x8_t, pp), acc1[dt], 0, 0, 0);
                    }
                }
                XB_PIN();
            }
        }
    };

    const int t_end = a.S / XK;
    __syncthreads();
    issue_tile(0, 0);
    for (int t = 0; t < t_end; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_tile(t + 1, 1);
        tile(std::integral_constant<int, 0>{}, t * XK);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < t_end) issue_tile(t + 2, 0);
        tile(std::integral_constant<int, 1>{}, (t + 1) * XK);
    }
#undef XB_RREAD
#undef XB_TREAD
#undef XB_PIN
    auto store = [&](uint16_t* base, int64_t ld, const xf32x16_t (&acc)[ND], float mul, bool scaled) {
        uint16_t* orow = base + ((int64_t)b * a.S + myrow) * ld + head * AD;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 w;
                if (scaled) {
                    w.x = pack_bf16x2(mul * acc[dt][4 * j + 0], mul * acc[dt][4 * j + 1]);
                    w.y = pack_bf16x2(mul * acc[dt][4 * j + 2], mul * acc[dt][4 * j + 3]);
                } else {
                    w.x = pack_bf16x2(acc[dt][4 * j + 0], acc[dt][4 * j + 1]);
                    w.y = pack_bf16x2(acc[dt][4 * j + 2], acc[dt][4 * j + 3]);
                }
                *reinterpret_cast<uint2*>(orow + 32 * dt + 8 * j + 4 * h) = w;
            }
    };
    if (MODE == 0) store(a.dQ, a.lddq, acc0, a.sm_scale, true);
    else {
        if (NEED_A0) store(a.dK, a.lddk, acc0, a.sm_scale, true);
        if (NEED_A1) store(a.dV, a.lddv, acc1, 1.0f, false);
    }
}


// The key side (dK and dV) as ONE SOFTWARE-PIPELINED wave per SIMD.  k_xattn_bwd<1, 4, 128, 3, 1> above runs its phases one after the
// other -- fragment reads, score MFMAs, softmax VALU, accumulating MFMAs -- and with a single wave on the SIMD nothing fills the gaps:
// 4050 cycles per 32-query step against 1024 of MFMA, and the four waves of the workgroup ask the LDS for the same 32 KB at the same
// time (128 B / clk = as long as the MFMAs take).  Here the two 32-query steps of a staged tile are interleaved by hand: the softmax of
// step 0 is issued BETWEEN the score MFMAs of step 1, the softmax of step 1 between the accumulating MFMAs of step 0, every LDS read is
// issued one stage before its use (two fragment buffers, two transposed-fragment sets), and the tile barrier sits in the middle of
// the last stage with the next tile's first reads behind it.  Every value is computed by the same instructions in the same order per
// accumulator as above (the chains S, dP, dK^T, dV^T are untouched), so the results are equal bit for bit (test_gpu_attn_exact.py).
template <int AD>
__global__ __launch_bounds__(256, 1) void k_xattn_bwd_kv(XBwdArgs a) {
    constexpr int WAVES = 4;
    constexpr int KB = AD / 32;                   // k-steps per fragment chunk (two chunks per score product)
    constexpr int ND = AD / 32;
    constexpr int NKS = AD / 16;
    constexpr int EPM = 8 / (2 * KB);             // softmax elements issued behind each MFMA of a stage
    constexpr int AROW = AD * 2;
    constexpr int ATILE = XK * AROW;
    constexpr int ABUF = 2 * ATILE;
    constexpr int AQ = 32 * WAVES;
    constexpr int RPW = XK / WAVES;
    constexpr int RPI = 1024 / AROW;
    constexpr int CPR = AROW / 16;
    constexpr int NP = RPW / RPI;
    constexpr int VEC_OFF = 2 * ABUF;
    static_assert(EPM >= 1 && NP >= 1, "head sizes 64 and 128");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lq = lane & 31;
    const int n_ob = a.S / AQ;
    int ob, bh;
    xattn_map((int)blockIdx.x, n_ob, a.B, a.H, a.kv_rep, false, a.map_legacy != 0, ob, bh);
    const int b = bh / a.H, head = bh % a.H, kvh = head / a.kv_rep;
    const int o0 = ob * AQ;
    const int myrow = o0 + 32 * wave + lq;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;

    const uint16_t* Qh = a.Q + (int64_t)b * a.q_bs + (int64_t)head * a.q_hs;
    const uint16_t* Kh = a.K + (int64_t)b * a.k_bs + (int64_t)kvh * a.k_hs;
    const uint16_t* Vh = a.V + (int64_t)b * a.v_bs + (int64_t)kvh * a.v_hs;
    const uint16_t* Oh = a.dO + (int64_t)b * a.o_bs + (int64_t)head * a.o_hs;
    xbf16x8_t bf0[NKS], bf1[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        bf0[ks] = __builtin_bit_cast(xbf16x8_t, *reinterpret_cast<const uint4*>(Kh + (int64_t)myrow * a.k_ts + 16 * ks + 8 * h));
        bf1[ks] = __builtin_bit_cast(xbf16x8_t, *reinterpret_cast<const uint4*>(Vh + (int64_t)myrow * a.v_ts + 16 * ks + 8 * h));
    }
    {
        const float* L2row = a.L2 + ((int64_t)b * a.H + head) * a.S;
        const float* Dvrow = a.Dv + ((int64_t)b * a.H + head) * a.S;
        float* vL = reinterpret_cast<float*>(lds + VEC_OFF);
        float* vD = vL + a.S;
        for (int i = tid; i < a.S; i += 64 * WAVES) { vL[i] = L2row[i]; vD[i] = Dvrow[i]; }
    }
    const bool quirk = lq == 27 || lq == 31;      // the library's non-fused element (k_xattn_bwd)
    const float qmul = quirk ? 1.0f : a.qk_scale;

    const int drow = lane / CPR, pchunk = lane % CPR;
    uint32_t doff0[NP], doff1[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = RPW * wave + RPI * p + drow;
        doff0[p] = (uint32_t)(r * a.q_ts + (pchunk ^ xattn_swz<AD>(r)) * 8);
        doff1[p] = (uint32_t)(r * a.o_ts + (pchunk ^ xattn_swz<AD>(r)) * 8);
    }
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) {
            const int p = j % NP;
            const uint16_t* T = (j < NP ? Qh + (int64_t)t * XK * a.q_ts + doff0[p] : Oh + (int64_t)t * XK * a.o_ts + doff1[p]);
            const uint32_t dst = lds0 + buf * ABUF + (j < NP ? 0 : ATILE) + (RPW * wave + RPI * p) * AROW;
            __builtin_amdgcn_global_load_lds((const void*)T, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
        }
    };
    uint32_t rA[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) rA[ks] = lds0 + lq * AROW + ((uint32_t)((2 * ks + h) ^ xattn_swz<AD>(lq)) << 4);
    const int gi = lane & 15, gg = lane >> 4;
    const int vrow = 4 * h + (gi >> 2);
    const int vrow_hi = vrow + 8;
    const int vcol0 = 16 * (gg & 1) + 4 * (gi & 3);
    uint32_t tAlo[ND], tAhi[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        const int col = 32 * dt + vcol0;
        tAlo[dt] = lds0 + vrow * AROW + ((uint32_t)((col >> 3) ^ xattn_swz<AD>(vrow)) << 4) + (col & 7) * 2;
        tAhi[dt] = lds0 + vrow_hi * AROW + ((uint32_t)((col >> 3) ^ xattn_swz<AD>(vrow_hi)) << 4) + (col & 7) * 2;
    }
    const uint32_t vLa = lds0 + VEC_OFF + 16 * h;             // + 4 * (tile row + 32 t + 8 j): the four l2 of accumulator registers 4 j ..
    const uint32_t vDa = vLa + 4 * a.S;

    xf32x16_t acc0[ND], acc1[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[dt][r] = 0.f; acc1[dt][r] = 0.f; }

    u32x4_t F[2][2][KB];                          // [chunk parity][Q | dO][k-step]: row fragments of the score products
    xs16x4_t TL[2][2][ND], TH[2][2][ND];          // [set][Q | dO][d tile]: transposed fragments of the accumulating products
    xf32x4_t LV[2][2], DV[2][2];                  // [half of the 16 elements][j]: l2 and delta of the streamed rows

#define XK_PIN() __builtin_amdgcn_sched_barrier(0)
#define XK_WAIT(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((N) > 15 ? 15 : (N)) : "memory")
#define XK_RREAD(DST, KS, T, REG, BUFX) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(rA[KS]), "n"((BUFX) * ABUF + (REG) * ATILE + (T) * 32 * AROW) : "memory")
#define XK_TREAD(LO, HI, DT, ST, REG, BUFX)                                                                                    \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"                                  \
                 : "=&v"(LO), "=&v"(HI) : "v"(tAlo[DT]), "v"(tAhi[DT]), "n"((BUFX) * ABUF + (REG) * ATILE + (ST) * 16 * AROW) : "memory")
#define XK_VREAD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(ADDR), "n"(OFF) : "memory")

    // R(t, c): the row fragments of chunk c of step t into F[c]
#define XK_ISSUE_R(BUFX, T, C)                                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < KB; ++ks) {                                                                        \
        XK_RREAD(F[C][0][ks], (C) * KB + ks, T, 0, BUFX);                                                                      \
        XK_RREAD(F[C][1][ks], (C) * KB + ks, T, 1, BUFX);                                                                      \
    }
    // T(t, s2): the transposed fragments of 16-row step 2 t + s2 into set s2
#define XK_ISSUE_T(BUFX, T, S2)                                                                                                \
    _Pragma("unroll") for (int dt = 0; dt < ND; ++dt) {                                                                        \
        XK_TREAD(TL[S2][0][dt], TH[S2][0][dt], dt, 2 * (T) + (S2), 0, BUFX);                                                   \
        XK_TREAD(TL[S2][1][dt], TH[S2][1][dt], dt, 2 * (T) + (S2), 1, BUFX);                                                   \
    }
    // L(t, half): l2 / delta of streamed rows 32 t + 4 h + 8 j + i, j = 2 half, 2 half + 1; VA = vLa + 4 * (first row of the tile)
#define XK_ISSUE_L(VA, VD, T, HF, X)                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                            \
        XK_VREAD(LV[HF][j], VA, 4 * ((X) + 32 * (T) + 8 * (2 * (HF) + j)));                                                    \
        XK_VREAD(DV[HF][j], VD, 4 * ((X) + 32 * (T) + 8 * (2 * (HF) + j)));                                                    \
    }
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    auto tile = [&](auto bufc, int tn, bool has_next, bool dma) {
        constexpr int BUF = decltype(bufc)::value;
        const int t0 = tn * XK;
        const uint32_t va = vLa + 4 * t0, vd = vDa + 4 * t0;
        const bool plain = t0 >= o0 + 32 * wave + 31 && o0 + 32 * wave + 31 < a.valid_len;
        const bool all_out = o0 + 32 * wave > t0 + XK - 1 || o0 + 32 * wave >= a.valid_len;
        xf32x16_t s0[2], s1[2];
        uint32_t pkP[2][8], pkS[2][8];
        auto init = [&](int t) __attribute__((always_inline)) {
            if (plain || all_out) {
                const float bias = plain ? a.bias_in_s : a.bias_out_s;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[t][r] = bias; s1[t][r] = 0.f; }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int query = t0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                    s0[t][r] = (myrow <= query && myrow < a.valid_len) ? a.bias_in_s : a.bias_out_s;
                    s1[t][r] = 0.f;
                }
            }
        };
        // MFMA i of score chunk c of step t:  i even -> S (K Q^T), i odd -> dP (V dO^T), k-step c KB + i / 2
        auto score_mfma = [&](int t, int c, int i) __attribute__((always_inline)) {
            const int ks = i >> 1;
            if ((i & 1) == 0) s0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, F[c][0][ks]), bf0[c * KB + ks], s0[t], 0, 0, 0);
            else s1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, F[c][1][ks]), bf1[c * KB + ks], s1[t], 0, 0, 0);
        };
        // MFMA i of the accumulating products of step t, 16-row half s2:  i even -> dK^T += Q^T dS, i odd -> dV^T += dO^T P, d tile i / 2
        auto acc_mfma = [&](int t, int s2, int i) __attribute__((always_inline)) {
            const int dt = i >> 1;
            if ((i & 1) == 0) {
                u32x4_t pe; pe.x = pkS[t][4 * s2]; pe.y = pkS[t][4 * s2 + 1]; pe.z = pkS[t][4 * s2 + 2]; pe.w = pkS[t][4 * s2 + 3];
                const xs16x8_t a0 = __builtin_shufflevector(TL[s2][0][dt], TH[s2][0][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                acc0[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, a0), __builtin_bit_cast(xbf16x8_t, pe), acc0[dt], 0, 0, 0);
            } else {
                u32x4_t pp; pp.x = pkP[t][4 * s2]; pp.y = pkP[t][4 * s2 + 1]; pp.z = pkP[t][4 * s2 + 2]; pp.w = pkP[t][4 * s2 + 3];
                const xs16x8_t a1 = __builtin_shufflevector(TL[s2][1][dt], TH[s2][1][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                acc1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8_t, a1), __builtin_bit_cast(xbf16x8_t, pp), acc1[dt], 0, 0, 0);
            }
        };
        // softmax element r of step t:  p = exp2(fma(qk_scale, s, -l2)) (quirk lanes: two roundings), ds = p * (dp - delta); pairs packed
        auto element = [&](int t, int r) __attribute__((always_inline)) {
            const int hf = r >> 3, j = (r >> 2) & 1, i = r & 3;
            const float prod = a.qk_scale * s0[t][r];
            const float tv = quirk ? prod : s0[t][r];
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(qmul, tv, -LV[hf][j][i]));
            const float dd = s1[t][r] - DV[hf][j][i];
            s0[t][r] = p;
            s1[t][r] = p * dd;
            if (r & 1) {
                pkP[t][r >> 1] = pack_bf16x2(s0[t][r - 1], s0[t][r]);
                pkS[t][r >> 1] = pack_bf16x2(s1[t][r - 1], s1[t][r]);
            }
        };

        // ---- stage 0   in flight: L(0) R(0,0) R(0,1)
        init(0);
        XK_WAIT(2 * KB);
        XK_PIN();
#pragma unroll
        for (int i = 0; i < 2 * KB; ++i) score_mfma(0, 0, i);
        XK_PIN();
        // ---- stage 1
        XK_WAIT(0);
        XK_ISSUE_R(BUF, 1, 0);
        XK_PIN();
#pragma unroll
        for (int i = 0; i < 2 * KB; ++i) score_mfma(0, 1, i);
        XK_PIN();
        // ---- stage 2   score MFMAs of step 1 (chunk 0) | softmax of step 0, elements 0 .. 7
        init(1);
        XK_WAIT(0);
        XK_ISSUE_T(BUF, 0, 0);
        XK_ISSUE_R(BUF, 1, 1);
        XK_PIN();
#pragma unroll
        for (int i = 0; i < 2 * KB; ++i) {
            score_mfma(1, 0, i);
#pragma unroll
            for (int e = 0; e < EPM; ++e) element(0, EPM * i + e);
            XK_PIN();
        }
        // ---- stage 3   score MFMAs of step 1 (chunk 1) | softmax of step 0, elements 8 .. 15
        XK_WAIT(0);
        XK_ISSUE_T(BUF, 0, 1);
        XK_ISSUE_L(va, vd, 1, 0, 0);
        XK_PIN();
#pragma unroll
        for (int i = 0; i < 2 * KB; ++i) {
            score_mfma(1, 1, i);
#pragma unroll
            for (int e = 0; e < EPM; ++e) element(0, 8 + EPM * i + e);
            XK_PIN();
        }
        // ---- stage 4   accumulating MFMAs of step 0 (rows 0 .. 15) | softmax of step 1, elements 0 .. 7
        XK_WAIT(0);
        XK_ISSUE_L(va, vd, 1, 1, 0);
        XK_PIN();
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) {
            acc_mfma(0, 0, i);
#pragma unroll
            for (int e = 0; e < EPM; ++e) element(1, EPM * i + e);
            XK_PIN();
        }
        XK_ISSUE_T(BUF, 1, 0);
        // ---- stage 5   accumulating MFMAs of step 0 (rows 16 .. 31) | softmax of step 1, elements 8 .. 15
        XK_WAIT(4 * ND);
        XK_PIN();
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) {
            acc_mfma(0, 1, i);
#pragma unroll
            for (int e = 0; e < EPM; ++e) element(1, 8 + EPM * i + e);
            XK_PIN();
        }
        XK_ISSUE_T(BUF, 1, 1);
        // ---- stage 6   accumulating MFMAs of step 1 (rows 0 .. 15); then the tile boundary
        XK_WAIT(4 * ND);
        XK_PIN();
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) acc_mfma(1, 0, i);
        XK_PIN();
        XK_WAIT(0);
        if (has_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (dma) issue_tile(tn + 2, BUF);
            XK_ISSUE_L(va, vd, 0, 0, XK);
            XK_ISSUE_L(va, vd, 0, 1, XK);
            XK_ISSUE_R((BUF ^ 1), 0, 0);
            XK_ISSUE_R((BUF ^ 1), 0, 1);
        }
        XK_PIN();
        // ---- stage 7   accumulating MFMAs of step 1 (rows 16 .. 31), over the next tile's first reads
#pragma unroll
        for (int i = 0; i < 2 * ND; ++i) acc_mfma(1, 1, i);
        XK_PIN();
    };

    const int t_end = a.S / XK;
    __syncthreads();
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_tile(1, 1);
    XK_ISSUE_L(vLa, vDa, 0, 0, 0);
    XK_ISSUE_L(vLa, vDa, 0, 1, 0);
    XK_ISSUE_R(0, 0, 0);
    XK_ISSUE_R(0, 0, 1);
    for (int t = 0; t < t_end; t += 2) {
        tile(I0{}, t, true, t + 2 < t_end);
        tile(I1{}, t + 1, t + 2 < t_end, t + 3 < t_end);
    }
#undef XK_PIN
#undef XK_WAIT
#undef XK_RREAD
#undef XK_TREAD
#undef XK_VREAD
#undef XK_ISSUE_R
#undef XK_ISSUE_T
#undef XK_ISSUE_L
    auto store = [&](uint16_t* base, int64_t ld, const xf32x16_t (&acc)[ND], float mul, bool scaled) {
        uint16_t* orow = base + ((int64_t)b * a.S + myrow) * ld + head * AD;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 w;
                if (scaled) {
                    w.x = pack_bf16x2(mul * acc[dt][4 * j + 0], mul * acc[dt][4 * j + 1]);
                    w.y = pack_bf16x2(mul * acc[dt][4 * j + 2], mul * acc[dt][4 * j + 3]);
                } else {
                    w.x = pack_bf16x2(acc[dt][4 * j + 0], acc[dt][4 * j + 1]);
                    w.y = pack_bf16x2(acc[dt][4 * j + 2], acc[dt][4 * j + 3]);
                }
                *reinterpret_cast<uint2*>(orow + 32 * dt + 8 * j + 4 * h) = w;
            }
    };
    store(a.dK, a.lddk, acc0, a.sm_scale, true);
    store(a.dV, a.lddv, acc1, 1.0f, false);
}

}  // namespace ar

extern "C" int64_t ar_attn_bwd_exact_workspace_bytes(int64_t B, int64_t S, int64_t H) { return 2 * B * S * H * (int64_t)sizeof(float); }

extern "C" int ar_attn_bwd_exact(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                                 void* dV, int64_t B, int64_t S, int64_t H, int64_t D, int64_t kv_rep, float scale, float bias_in,
                                 float bias_out, int64_t valid_len, int64_t q_bs, int64_t q_hs, int64_t q_ts, int64_t k_bs, int64_t k_hs,
                                 int64_t k_ts, int64_t v_bs, int64_t v_hs, int64_t v_ts, int64_t o_bs, int64_t o_hs, int64_t o_ts,
                                 int64_t do_bs, int64_t do_hs, int64_t do_ts, int64_t lddq, int64_t lddk, int64_t lddv, void* workspace,
                                 int64_t workspace_bytes, ar_stream_t stream) {
    if ((D != 64 && D != 128) || S % 128 || S > 4096 || B <= 0 || H <= 0 || kv_rep < 1 || H % kv_rep) return AR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ar_attn_bwd_exact_workspace_bytes(B, S, H)) return AR_ERR_UNSUPPORTED;
    if (!(bias_in == bias_in) || !(bias_out == bias_out) || fabsf(bias_in) > 1e4f || fabsf(bias_out) > 1e4f || valid_len < 1 || valid_len > S)
        return AR_ERR_UNSUPPORTED;
    {
        uint32_t u0, u1; memcpy(&u0, &bias_in, 4); memcpy(&u1, &bias_out, 4);
        if ((u0 | u1) & 0xffffu) return AR_ERR_UNSUPPORTED;
    }
    const int64_t hd = H * D;
    if (lddq <= 0) lddq = hd;
    if (lddk <= 0) lddk = hd;
    if (lddv <= 0) lddv = hd;
    const int64_t st[18] = {q_bs, q_hs, q_ts, k_bs, k_hs, k_ts, v_bs, v_hs, v_ts, o_bs, o_hs, o_ts, do_bs, do_hs, do_ts, lddq, lddk, lddv};
    for (int i = 0; i < 18; ++i)
        if (st[i] < 0 || (st[i] % 8)) return AR_ERR_UNSUPPORTED;
    if (lddq < hd || lddk < hd || lddv < hd) return AR_ERR_UNSUPPORTED;
    if (64 * q_ts > 0x7fffffffLL || 64 * k_ts > 0x7fffffffLL || 64 * v_ts > 0x7fffffffLL || 64 * do_ts > 0x7fffffffLL) return AR_ERR_UNSUPPORTED;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15)
        return AR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float* Dv = (float*)workspace;
    float* L2 = Dv + B * S * H;
    {
        const int64_t rows = B * S * H, lanes = rows * (D / 8);
        hipLaunchKernelGGL(k_xattn_bwd_prep, (int)((lanes + kTPB - 1) / kTPB), kTPB, 0, s, (const uint16_t*)dO, do_bs, do_hs, do_ts,
                           (const uint16_t*)O, o_bs, o_hs, o_ts, LSE, Dv, L2, (int)B, (int)S, (int)H, (int)D);
    }
    XBwdArgs a;
    a.Q = (const uint16_t*)Q; a.K = (const uint16_t*)K; a.V = (const uint16_t*)V; a.dO = (const uint16_t*)dO;
    a.L2 = L2; a.Dv = Dv;
    a.dQ = (uint16_t*)dQ; a.dK = (uint16_t*)dK; a.dV = (uint16_t*)dV;
    a.B = (int)B; a.S = (int)S; a.H = (int)H; a.kv_rep = (int)kv_rep;
    a.q_bs = q_bs; a.q_hs = q_hs; a.q_ts = q_ts; a.k_bs = k_bs; a.k_hs = k_hs; a.k_ts = k_ts; a.v_bs = v_bs; a.v_hs = v_hs; a.v_ts = v_ts;
    a.o_bs = do_bs; a.o_hs = do_hs; a.o_ts = do_ts;
    a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.sm_scale = scale;
    a.qk_scale = scale * 1.44269502162933349609375f;
    const float bias_scale = 1.0f / scale;                 // correctly rounded (the library: v_div_scale / v_div_fmas / v_div_fixup)
    a.bias_in_s = bias_in * bias_scale; a.bias_out_s = bias_out * bias_scale; a.valid_len = (int)valid_len;
    a.map_legacy = (g_xattn_cfg & 32) ? 1 : 0;
    constexpr int LDS_T = 4 * XK * 64 * 2, LDS_T128 = 4 * XK * 128 * 2;
    const int vec = (int)(2 * S * sizeof(float));
    static PerDeviceOnce attr;
    if (attr.first()) {
        constexpr int VEC_MAX = 2 * 4096 * (int)sizeof(float);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<0, 8, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<1, 8, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<0, 8, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<1, 8, 128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<1, 8, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<0, 4, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<1, 4, 128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<1, 4, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_xattn_bwd<1, 4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T + VEC_MAX);
    }
    const int bform = (g_xattn_cfg >> 2) & 3;
    if ((bform == 3 || bform == 0) && D == 128) {         // dK and dV in ONE kernel (7 GEMM passes instead of 8): 4 waves, one wave per SIMD with the
                                                          // whole register file (accumulators in the AGPRs) -- 3 % faster than the two key-side kernels
        static PerDeviceOnce attr1;
        if (attr1.first())
            (void)hipFuncSetAttribute((const void*)k_xattn_bwd<1, 4, 128, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + 2 * 4096 * (int)sizeof(float));
        const int grid4 = (int)(B * H * (S / 128));
        if (!(g_xattn_cfg & 16)) hipLaunchKernelGGL((k_xattn_bwd<1, 4, 128, 3, 1>), grid4, 256, LDS_T128 + vec, s, a);
        else {                                            // on request (config bit 16): the same kernel software-pipelined by hand -- 6 % faster, equal bits in
                                                          // every direct comparison (1 650 stress calls, Llama-3-8B 33 full runs), but 4 of ~80 Mixtral module-path
                                                          // runs with it parted inside the loop against 0 of ~70 with the phase kernel: not the default
            static PerDeviceOnce attr2;
            if (attr2.first())
                (void)hipFuncSetAttribute((const void*)k_xattn_bwd_kv<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + 2 * 4096 * (int)sizeof(float));
            hipLaunchKernelGGL((k_xattn_bwd_kv<128>), grid4, 256, LDS_T128 + vec, s, a);
        }
        if (S % 256) hipLaunchKernelGGL((k_xattn_bwd<0, 4, 128>), grid4, 256, LDS_T128, s, a);
        else hipLaunchKernelGGL((k_xattn_bwd<0, 8, 128>), (int)(B * H * (S / 256)), 512, LDS_T128, s, a);
        return launch_status();
    }
    if (bform == 1 || (bform == 0 && D == 64) || S % 256) {         // workgroups of 4 waves (the default at head size 64: measured; S % 256 != 0)
        const int grid4 = (int)(B * H * (S / 128));
        if (D == 128) {
            hipLaunchKernelGGL((k_xattn_bwd<1, 4, 128, 1>), grid4, 256, LDS_T128 + vec, s, a);
            hipLaunchKernelGGL((k_xattn_bwd<1, 4, 128, 2>), grid4, 256, LDS_T128 + vec, s, a);
            hipLaunchKernelGGL((k_xattn_bwd<0, 4, 128>), grid4, 256, LDS_T128, s, a);
        } else {
            if (g_xattn_cfg & 64) {                       // A/B only: the hand-pipelined key-side kernel at head size 64 (one wave per SIMD) is
                                                          // SLOWER than two phase-kernel workgroups per CU: 0.56 against 0.45 ms per backward call
                static PerDeviceOnce attr64;
                if (attr64.first())
                    (void)hipFuncSetAttribute((const void*)k_xattn_bwd_kv<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T + 2 * 4096 * (int)sizeof(float));
                hipLaunchKernelGGL((k_xattn_bwd_kv<64>), grid4, 256, LDS_T + vec, s, a);
            } else
            hipLaunchKernelGGL((k_xattn_bwd<1, 4, 64>), grid4, 256, LDS_T + vec, s, a);
            hipLaunchKernelGGL((k_xattn_bwd<0, 4, 64>), grid4, 256, LDS_T, s, a);
        }
        return launch_status();
    }
    const int grid = (int)(B * H * (S / 256));
    if (D == 128) {
        hipLaunchKernelGGL((k_xattn_bwd<1, 8, 128, 1>), grid, 512, LDS_T128 + vec, s, a);
        hipLaunchKernelGGL((k_xattn_bwd<1, 8, 128, 2>), grid, 512, LDS_T128 + vec, s, a);
        hipLaunchKernelGGL((k_xattn_bwd<0, 8, 128>), grid, 512, LDS_T128, s, a);
    } else {
        hipLaunchKernelGGL((k_xattn_bwd<1, 8, 64>), grid, 512, LDS_T + vec, s, a);
        hipLaunchKernelGGL((k_xattn_bwd<0, 8, 64>), grid, 512, LDS_T, s, a);
    }
    return launch_status();
}
