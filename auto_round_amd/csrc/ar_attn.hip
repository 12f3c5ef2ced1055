// ar_attn.hip -- causal flash-attention forward for gfx950 (head dimension 128 or 64, bf16), token-major operands.
//
// replaces: the attention forward of the decoder block inside the tuning loop -- transformers' sdpa_attention_forward
//           (transformers/integrations/sdpa_attention.py) -> torch scaled_dot_product_attention, which on ROCm 7.2 / torch 2.10 is
//           AOTriton's `attn_fwd`: 0.84 ms per call at the tuning minibatch (8 x 32 heads x 2048 x 128), 14 % MFMA utilisation
//           (profiles/archive/r02_llama8b_fused_pmc_MfmaUtil.csv).  The backward stays the library's (aiter fmha_bwd, 50 %): this kernel
//           returns the output and the natural-log row sums in the layout aten::_scaled_dot_product_efficient_attention_backward
//           takes (out [B,S,H,D] token-major, logsumexp [B,H,S] fp32).
//
// Design (CDNA4-first):
//   * one 256-thread workgroup per (batch, head, 128 queries); wave w owns 32 queries.  Keys / values come in tiles of 64 through
//     LDS-DMA (global_load_lds, 16 B per lane), double buffered, XOR-swizzled on the 16-byte chunk (attn_swz) so that the
//     fragment reads below are bank-conflict free.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: a-operand = K fragment (ds_read_b128, keys on lanes), b-operand = Q^T fragment held
//     in registers for the whole kernel.  The accumulator layout puts ONE query on each lane (column), 16 keys in its registers, so
//     the softmax row statistics are in-lane reductions plus one exchange between the two lane halves.
//   * O^T += V^T P^T: the a-operand is the V^T fragment, read with ds_read_b64_tr_b16 (hardware transposing LDS read: d on lanes,
//     keys in registers); the b-operand is P^T -- the S^T accumulators themselves, converted to bf16 in place: the accumulator's
//     key order (lane half h holds keys 4h..4h+3 and 8+4h..8+4h+3 of each 16) is simply adopted as the k order of the second MFMA,
//     and the V^T reads pick their rows to match.  O^T again has one query per lane: the online-softmax rescale is a per-lane scalar.
//   * exp2 with the scale folded in (scores * scale * log2 e); running max / sum per query in registers; causal mask only on the two
//     diagonal key tiles; waves whose queries lie entirely before a key tile skip its MFMAs.
#include "ar_common.hpp"
#include <type_traits>

namespace ar {

typedef short as16x4_t __attribute__((ext_vector_type(4)));
typedef short as16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x8_t __attribute__((ext_vector_type(8)));
typedef float af32x16_t __attribute__((ext_vector_type(16)));

constexpr int AK = 64;               // keys per tile

// LDS swizzle of a staged row r: the 16-byte chunk index is XORed with bits of the row number so that the 16 rows of a
// ds_read_b128 group and the 4 consecutive rows of a transposing read both land on disjoint banks (256 B of banks per pass).
//   256-byte rows (D = 128): every row starts on bank 0 -> the row's low four bits, bit pairs swapped (distinct r & 15 for the
//     b128 group; distinct r & 3 in the top two bits = distinct 64-byte blocks for the transposing read);
//   128-byte rows (D = 64): rows alternate between the two halves of the banks, eight chunks per row -> three bits of r >> 1,
//     rotated so that r and r + 2 (same half) differ in the top bit (the transposing read covers four chunks of each row).
template <int D>
__device__ __forceinline__ int attn_swz(int r) {
    if constexpr (D == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    else return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
}

// value of the other lane half (lane ^ 32) combined with this one, without touching LDS: v_permlane32_swap hands every lane both
// halves' values.  (A ds_bpermute here would make the compiler drain the LDS-DMA of the next tile -- vmcnt(0) -- in front of it.)
// (inline asm: the two-result builtin returned the first result twice with this compiler -- probed on the GPU)
__device__ __forceinline__ float halves_max(float x) {
    float lo = x, hi = x;
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(lo), "+v"(hi));      // lo <- {x[0:31], x[0:31]}, hi <- {x[32:63], x[32:63]}
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float halves_sum(float x) {
    float lo = x, hi = x;
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(lo), "+v"(hi));
    return lo + hi;
}

struct AttnArgs {
    const uint16_t* Q; const uint16_t* K; const uint16_t* V;    // [B, S, H, D] token-major
    uint16_t* O;                                                   // [B, S, H, D]
    float* LSE;                                                    // [B, H, S]
    int B, S, H;
    int64_t ldq, ldkv;                                             // elements between consecutive tokens of Q and of K / V (>= H * D:
                                                                   // the operands may be column slices of one merged projection output)
    float scale_log2e;                                             // softmax scale * log2(e)
    float bias_in_l2, bias_out_l2;                                 // MASKED: additive bias * log2(e) inside / outside the kept region
    int valid_len;                                                 // MASKED: keys >= valid_len are outside (key padding at the end)
};

// WAVES waves of 32 queries per workgroup: 8 (256 queries, 512 threads) halves the K / V staging per wave and per query against 4 --
// the LDS-DMA issue is the largest non-MFMA cost of the kernel (ablations in DESIGN.md) -- and is used whenever S % 256 == 0.
// MASKED = false: causal attention (keys after the query are skipped).  MASKED = true: the structured additive mask of a calibration
// flow -- bias(q, k) = bias_in where `k <= q and k < valid_len`, bias_out elsewhere, BOTH finite (transformers >= 5 hands the block a
// boolean `causal & key-is-valid` mask and the reference's input cache casts it to a 0 / 1 bias: auto_round/calibration/llm.py:360-402,
// inputs.py:100-107) -- so every query attends to every key and no tile is skipped; the bias is two registers, not an [S, S] operand.
template <int WAVES, int AD, bool MASKED = false>
__global__ __launch_bounds__(64 * WAVES, 2) void k_attn_fwd(AttnArgs a) {
    constexpr int AROW = AD * 2;         // bytes per staged key / value row
    constexpr int ATILE = AK * AROW;     // one operand's tile (16 KB at D = 128)
    constexpr int ABUF = 2 * ATILE;      // K tile + V tile; two of them (double buffering) are the kernel's LDS
    constexpr int NKS = AD / 16;         // k-steps of the S^T product
    constexpr int ND = AD / 32;          // d-tiles of O^T
    constexpr int AQ = 32 * WAVES;       // queries per workgroup
    constexpr int RPW = AK / WAVES;      // tile rows staged per wave (16 or 8)
    constexpr int RPI = 1024 / AROW;     // rows per DMA instruction (64 lanes x 16 B): 4 or 8
    constexpr int CPR = AROW / 16;       // 16-byte chunks per row
    constexpr int NP = RPW / RPI;
    static_assert(NP >= 1, "a wave stages at least one DMA instruction per operand");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lq = lane & 31;
    // Workgroup -> (batch * head, query tile).  Workgroups go to the 8 XCDs round-robin, so the (batch, head) index takes the low
    // three bits of blockIdx: every XCD then owns whole heads -- all their query tiles, whose K / V it keeps in its own L2 -- and
    // the same mix of long and short (causal) query tiles; within that, the long tiles are started first.
    const int n_qt = a.S / AQ;
    const int n_bh = a.B * a.H;
    int qt, bh;
    if ((n_bh & 7) == 0) {
        const int j = blockIdx.x >> 3, per = n_bh >> 3;
        bh = (j % per) * 8 + (int)(blockIdx.x & 7);
        qt = n_qt - 1 - j / per;
    } else {
        qt = n_qt - 1 - (int)(blockIdx.x % n_qt);
        bh = blockIdx.x / n_qt;
    }
    const int b = bh / a.H, head = bh % a.H;
    const int q0 = qt * AQ;
    const int64_t row_stride = a.ldkv;                             // elements between consecutive tokens of K / V
    const uint16_t* Qb = a.Q + (int64_t)b * a.S * a.ldq + head * AD;
    const uint16_t* Kb = a.K + (int64_t)b * a.S * a.ldkv + head * AD;
    const uint16_t* Vb = a.V + (int64_t)b * a.S * a.ldkv + head * AD;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;

    // ---- Q^T fragments (b-operand: n = query = lane & 31, k = d = 16 ks + 8 h .. +7), kept for the whole kernel
    const int myq = q0 + 32 * wave + lq;
    abf16x8_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const uint4 r = *reinterpret_cast<const uint4*>(Qb + (int64_t)myq * a.ldq + 16 * ks + 8 * h);
        qf[ks] = __builtin_bit_cast(abf16x8_t, r);
    }

    // ---- DMA: a tile is 64 rows of AROW bytes; one instruction moves RPI rows (lane -> row RPI p + lane / CPR, physical chunk
    // lane % CPR).  wave w moves rows RPW w .. RPW w + RPW - 1 of the K tile and of the V tile.  Per-lane element offsets
    // are fixed; the tile base pointers are uniform and advance by 64 rows per tile.
    const int drow = lane / CPR, pchunk = lane % CPR;
    uint32_t doff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = RPW * wave + RPI * p + drow;                 // row inside the tile
        doff[p] = (uint32_t)(r * row_stride + (pchunk ^ attn_swz<AD>(r)) * 8);
    }
    // piece j of 2 NP: K rows (j < NP) or V rows of this wave
    auto issue_piece = [&](int kt, int buf, int j) {
        const int p = j % NP;
        const uint16_t* T = (j < NP ? Kb : Vb) + (int64_t)kt * AK * row_stride;      // uniform
        const uint32_t dst = lds0 + buf * ABUF + (j < NP ? 0 : ATILE) + (RPW * wave + RPI * p) * AROW;   // wave-uniform
        __builtin_amdgcn_global_load_lds((const void*)(T + doff[p]), (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
    };
    auto issue_tile = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) issue_piece(kt, buf, j);
    };

    // ---- fragment read addresses: one register per k-step (K) / per d-tile (V^T, low and high key rows); buffer, key sub-tile and
    // 16-key step are immediate offsets of the read instructions (the swizzle of a row does not depend on them)
    // K (a-operand of S^T): m = key = 32 t + lq, k = d = 16 ks + 8 h .. +7 -> 16-byte chunk 2 ks + h of row key
    uint32_t kA[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kA[ks] = lds0 + lq * AROW + ((uint32_t)((2 * ks + h) ^ attn_swz<AD>(lq)) << 4);
    // V^T (a-operand of O^T): m = d = 32 dt + (lane & 31), k = keys; lane group g = lane >> 4 supplies rows 16 st + 4 h + (i >> 2)
    // (+ 8 for the second half) and the 8-byte piece (i & 3) of the 16 columns 32 dt + 16 (g & 1) ..
    const int gi = lane & 15, gg = lane >> 4;
    const int vrow = 4 * h + (gi >> 2);
    const int vcol0 = 16 * (gg & 1) + 4 * (gi & 3);
    uint32_t vAlo[ND], vAhi[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        const int col = 32 * dt + vcol0;
        vAlo[dt] = lds0 + vrow * AROW + ((uint32_t)((col >> 3) ^ attn_swz<AD>(vrow)) << 4) + (col & 7) * 2;
        vAhi[dt] = lds0 + (vrow + 8) * AROW + ((uint32_t)((col >> 3) ^ attn_swz<AD>(vrow + 8)) << 4) + (col & 7) * 2;
    }

    af32x16_t o[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

#ifndef AR_ATTN_ABL
#define AR_ATTN_ABL 0      // timing experiments only (wrong results): 1 no K/V staging after the first tile, 2 no softmax arithmetic
#endif
#define AR_PINA() __builtin_amdgcn_sched_barrier(0)
#define AR_KREAD(KS, T) if (!(AR_ATTN_ABL & 4)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(kf[(KS) & 3][T]) : "v"(kA[KS]), "n"(BUF * ABUF + (T) * 32 * AROW) : "memory")
#define AR_VREAD(ST)                                                                                                     \
    _Pragma("unroll") for (int dt = 0; dt < ND; ++dt) if (!(AR_ATTN_ABL & 4))                                            \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"                        \
                     : "=&v"(vlo[(ST) & 1][dt]), "=&v"(vhi[(ST) & 1][dt])                                                \
                     : "v"(vAlo[dt]), "v"(vAhi[dt]), "n"(BUF * ABUF + ATILE + (ST) * 16 * AROW) : "memory");
    // one key tile from LDS buffer BUF (compile time).  (Issuing the next tile's DMA pieces between the MFMAs of the S^T product
    // instead of up front was tried: their 64-bit addresses stay live across the loop and the kernel spills.)
    auto tile = [&](auto bufc, int kt) {
        constexpr int BUF = decltype(bufc)::value;
        const int k0 = kt * AK;
        // ---- S^T = K Q^T: 2 key sub-tiles x NKS k-steps; a ring of 4 k-steps of fragments, 6 reads in flight ahead of the MFMAs.
        // A wait is tied to the registers it releases ("+v"): nothing else would keep the MFMAs behind it.
        af32x16_t s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        u32x4_t kf[4][2];
        AR_KREAD(0, 0); AR_KREAD(0, 1); AR_KREAD(1, 0); AR_KREAD(1, 1); AR_KREAD(2, 0); AR_KREAD(2, 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 3 < NKS) { AR_KREAD(ks + 3, 0); AR_KREAD(ks + 3, 1); }
            const int ahead = (ks + 3 < NKS ? ks + 3 : NKS - 1) - ks;         // k-steps of reads issued behind this one's
            if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[ks & 3][0]), "+v"(kf[ks & 3][1])::"memory");
            if (!(AR_ATTN_ABL & 8)) {
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8_t, kf[ks & 3][0]), qf[ks], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8_t, kf[ks & 3][1]), qf[ks], s[1], 0, 0, 0);
            }
            AR_PINA();
        }
        // the V^T fragments of the first 32 keys do not depend on the softmax: their reads fly under it
        as16x4_t vlo[2][ND], vhi[2][ND];
        AR_VREAD(0)
        AR_VREAD(1)
        AR_PINA();
        // ---- online softmax on the lane's query (column)
        const bool diag = k0 + AK - 1 > q0 + 32 * wave;                   // wave-uniform: the tile reaches past some query
        if (!(AR_ATTN_ABL & 2)) {
        // row maximum on the raw scores (the scale is positive), then p = exp2(s * scale - m) as one fma + v_exp_f32
        float mx = -INFINITY;
        if (MASKED) {
            // scores in log2 units with the bias added: s * scale * log2 e + (kept ? bias_in : bias_out) * log2 e
            const bool plain = !diag && k0 + AK <= a.valid_len;           // wave-uniform: every (query, key) of the tile is kept
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                    const float bias = (plain || (key <= myq && key < a.valid_len)) ? a.bias_in_l2 : a.bias_out_l2;
                    s[t][r] = __builtin_fmaf(s[t][r], a.scale_log2e, bias);
                }
        } else if (diag) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                    s[t][r] = key > myq ? -INFINITY : s[t][r];
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = halves_max(mx);
        const float m_new = fmaxf(m_run, MASKED ? mx : mx * a.scale_log2e);     // finite: key 0 is visible to every query
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);        // raw v_exp_f32: arguments <= 0
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(MASKED ? s[t][r] - m_new : __builtin_fmaf(s[t][r], a.scale_log2e, -m_new));
                s[t][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;                                     // per lane half; the halves are added at the end
        m_run = m_new;
        if (__any(alpha != 1.0f)) {                                       // the running maximum moved for some query of the wave
            typedef float af32x2 __attribute__((ext_vector_type(2)));
            const af32x2 a2 = {alpha, alpha};
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const af32x2 v = af32x2{o[dt][r], o[dt][r + 1]} * a2;
                    o[dt][r] = v.x; o[dt][r + 1] = v.y;
                }
        }
        }
        AR_PINA();
        // ---- O^T += V^T P^T: 4 k-steps of 16 keys (sub-tile t, half s2) x ND d-tiles; two steps of fragments in flight
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int t = st >> 1, s2 = st & 1;
            as16x8_t pb;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const uint32_t w = pack_bf16x2(s[t][8 * s2 + e], s[t][8 * s2 + e + 1]);
                pb[e] = (short)(w & 0xffffu);
                pb[e + 1] = (short)(w >> 16);
            }
            // (the next step's 2 ND reads may stay in flight, except behind the last step)
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
                const as16x8_t va = __builtin_shufflevector(vlo[st & 1][dt], vhi[st & 1][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                if (!(AR_ATTN_ABL & 8)) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8_t, va), __builtin_bit_cast(abf16x8_t, pb),
                                                                o[dt], 0, 0, 0);
            }
            AR_PINA();
            if (st + 2 < 4) { AR_VREAD(st + 2) }
        }
    };

    const int n_kt = MASKED ? a.S / AK : (q0 + AQ) / AK;                  // causal: key tiles up to the diagonal; always an even number
    issue_tile(0, 0);
    for (int kt = 0; kt < n_kt; kt += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // tile kt landed for everybody; buffer 1 is free
        if (!(AR_ATTN_ABL & 1)) issue_tile(kt + 1, 1);
        if (MASKED || kt * AK <= q0 + 32 * wave + 31) tile(std::integral_constant<int, 0>{}, kt);     // else every query precedes the tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < n_kt && !(AR_ATTN_ABL & 1)) issue_tile(kt + 2, 0);
        if (MASKED || (kt + 1) * AK <= q0 + 32 * wave + 31) tile(std::integral_constant<int, 1>{}, kt + 1);
    }
#undef AR_KREAD
#undef AR_VREAD
#undef AR_PINA
    // ---- normalise and store: lane = query, registers = d (runs of 4 consecutive d -> 8-byte stores)
    const float l_tot = halves_sum(l_run);
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
    if (h == 0) a.LSE[((int64_t)b * a.H + head) * a.S + myq] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
}

}  // namespace ar

using namespace ar;

static int attn_fwd_impl(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H, int64_t D,
                         float scale, int causal, int64_t ldq, int64_t ldkv, bool masked, float bias_in, float bias_out, int64_t valid_len,
                         ar_stream_t stream);

extern "C" int ar_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H,
                           int64_t D, float scale, int causal, int64_t ldq, int64_t ldkv, ar_stream_t stream) {
    return attn_fwd_impl(Q, K, V, O, LSE, B, S, H, D, scale, causal, ldq, ldkv, false, 0.f, 0.f, S, stream);
}

extern "C" int ar_attn_fwd_masked(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H,
                                  int64_t D, float scale, float bias_in, float bias_out, int64_t valid_len, int64_t ldq, int64_t ldkv,
                                  ar_stream_t stream) {
    if (!(bias_in == bias_in) || !(bias_out == bias_out) || fabsf(bias_in) > 1e4f || fabsf(bias_out) > 1e4f || valid_len < 1 || valid_len > S)
        return AR_ERR_UNSUPPORTED;          // hard (-inf) masks are another kernel's job: every key must keep a finite score
    return attn_fwd_impl(Q, K, V, O, LSE, B, S, H, D, scale, 1, ldq, ldkv, true, bias_in, bias_out, valid_len, stream);
}

static int attn_fwd_impl(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t S, int64_t H, int64_t D,
                         float scale, int causal, int64_t ldq, int64_t ldkv, bool masked, float bias_in, float bias_out, int64_t valid_len,
                         ar_stream_t stream) {
    if ((D != 128 && D != 64) || !causal || S % 128 || B <= 0 || H <= 0 || S <= 0) return AR_ERR_UNSUPPORTED;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) return AR_ERR_UNSUPPORTED;
    if (ldq <= 0) ldq = H * D;
    if (ldkv <= 0) ldkv = H * D;
    if (ldq < H * D || ldkv < H * D || (ldq % 8) || (ldkv % 8) || 64 * ldkv > 0x7fffffffLL) return AR_ERR_UNSUPPORTED;
    AttnArgs a;
    a.ldq = ldq; a.ldkv = ldkv;
    a.Q = (const uint16_t*)Q; a.K = (const uint16_t*)K; a.V = (const uint16_t*)V; a.O = (uint16_t*)O; a.LSE = LSE;
    a.B = (int)B; a.S = (int)S; a.H = (int)H;
    a.scale_log2e = scale * 1.4426950408889634f;
    a.bias_in_l2 = bias_in * 1.4426950408889634f; a.bias_out_l2 = bias_out * 1.4426950408889634f; a.valid_len = (int)valid_len;
    constexpr int LDS128 = 4 * AK * 128 * 2, LDS64 = 4 * AK * 64 * 2;      // 2 buffers x (K tile + V tile)
    static PerDeviceOnce attr;
    if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)k_attn_fwd<4, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        (void)hipFuncSetAttribute((const void*)k_attn_fwd<8, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
    }
    hipStream_t st = (hipStream_t)stream;
    if (masked) {
        static PerDeviceOnce attr_m;
        if (attr_m.first()) {
            (void)hipFuncSetAttribute((const void*)k_attn_fwd<4, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
            (void)hipFuncSetAttribute((const void*)k_attn_fwd<8, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128);
        }
        if (D == 128) {
            if (S % 256 == 0) hipLaunchKernelGGL((k_attn_fwd<8, 128, true>), (int)(B * H * (S / 256)), 512, LDS128, st, a);
            else hipLaunchKernelGGL((k_attn_fwd<4, 128, true>), (int)(B * H * (S / 128)), 256, LDS128, st, a);
        } else {
            if (S % 256 == 0) hipLaunchKernelGGL((k_attn_fwd<8, 64, true>), (int)(B * H * (S / 256)), 512, LDS64, st, a);
            else hipLaunchKernelGGL((k_attn_fwd<4, 64, true>), (int)(B * H * (S / 128)), 256, LDS64, st, a);
        }
        return launch_status();
    }
    if (D == 128) {
        if (S % 256 == 0) hipLaunchKernelGGL((k_attn_fwd<8, 128>), (int)(B * H * (S / 256)), 512, LDS128, st, a);
        else hipLaunchKernelGGL((k_attn_fwd<4, 128>), (int)(B * H * (S / 128)), 256, LDS128, st, a);
    } else {
        if (S % 256 == 0) hipLaunchKernelGGL((k_attn_fwd<8, 64>), (int)(B * H * (S / 256)), 512, LDS64, st, a);
        else hipLaunchKernelGGL((k_attn_fwd<4, 64>), (int)(B * H * (S / 128)), 256, LDS64, st, a);
    }
    return launch_status();
}
