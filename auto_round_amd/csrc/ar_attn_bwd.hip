// ar_attn_bwd.hip -- flash-attention BACKWARD for gfx950, bf16, token-major operands, deterministic: causal at head dimension 64;
// under the calibration flow's structured mask at head dimensions 64 and 128.
//
// replaces: the attention backward of the decoder block inside the tuning loop -- autograd of transformers' sdpa_attention_forward
//           (transformers/integrations/sdpa_attention.py -> torch scaled_dot_product_attention(..., is_causal=True)), which on
//           ROCm 7.2 / torch 2.10 is aiter's `fmha_bwd_hd64_bf16_causal_a32_rtne_pssk` (+ pre / post-process kernels): 0.42 ms per
//           call at OPT-125M's tuning minibatch (8 x 12 heads x 2048 x 64), 0.30 PFLOP/s, fp32 atomics for dQ
//           (profiles/r03_opt125m_graph_kernel_stats.csv: 26 % of the iteration, the largest kernel of BASELINE configs[0]).
//
// Design: the forward kernel's skeleton (csrc/ar_attn.hip) used twice, no float atomics, every sum in a fixed order.
//   Both kernels keep two operands RESIDENT in registers as MFMA b-operands (one "own" row per lane), stream two tensors through
//   LDS in tiles of 64 rows (LDS-DMA, double buffered, the forward's XOR swizzle), compute two score tiles
//        sc0 = R0 * B0^T,  sc1 = R1 * B1^T          (a-operand: tile rows by ds_read_b128; accumulators: own row on the lane, 16
//                                                     streamed rows in registers)
//   (their accumulators initialised to -lse / scale and -D of the query row, so the softmax shift and the "- D" of dS ride in the
//   MFMA), turn them into P = exp2(sc0 * c) and E = P * sc1 in registers, and feed those -- converted to bf16 in
//   place, exactly like P^T in the forward -- as b-operands of the accumulating products, whose a-operands are the SAME tiles read
//   through the transposing LDS read (ds_read_b64_tr_b16):
//     MODE 0 (dQ):      own rows = 256 queries;  R0 = K, R1 = V tiles up to the diagonal;  B0 = Q^T, B1 = dO^T;  L2 / D per lane;
//                       dQ^T += K^T E                      -> dQ = scale * acc
//     MODE 1 (dK, dV):  own rows = 256 keys;     R0 = Q, R1 = dO tiles from the diagonal on; B0 = K^T, B1 = V^T; L2 / D per
//                       streamed row, read from an LDS copy of the (batch, head)'s whole L2 / D rows;
//                       dV^T += dO^T P,  dK^T += Q^T E     -> dK = scale * acc
//   7 MFMA passes instead of the 5 an atomics-based backward needs -- the price of determinism; at head size 64 the library runs
//   at 0.30 PFLOP/s, so the forward kernel's 0.59 PFLOP/s leaves room (at head size 128 the library's 0.71 does not: not built).
//   k_attn_bwd_prep computes -D = -rowsum(dO * O) and -lse / scale once.
// Outputs are written token-major with a caller-given row stride: dq / dk / dv can be column slices of ONE [tokens, 3 H D] buffer --
// the gradient of a merged q/k/v projection -- without a gather pass.
#include "ar_common.hpp"
#include <type_traits>

namespace ar {

typedef short bs16x4_t __attribute__((ext_vector_type(4)));
typedef short bs16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bbf16x8_t __attribute__((ext_vector_type(8)));
typedef float bf32x16_t __attribute__((ext_vector_type(16)));

constexpr int BK = 64;               // streamed rows per tile

template <int D>
__device__ __forceinline__ int battn_swz(int r) {      // the forward's LDS swizzle (csrc/ar_attn.hip attn_swz)
    if constexpr (D == 128) return ((r & 3) << 2) | ((r >> 2) & 3);
    else return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);
}

struct AttnBwdArgs {
    const uint16_t* Q; const uint16_t* K; const uint16_t* V; const uint16_t* dO;     // token-major, row strides below
    const float* L2; const float* Dv;                                               // [B, H, S] fp32: -lse / scale, -rowsum(dO * O)
    uint16_t* dQ; uint16_t* dK; uint16_t* dV;
    int B, S, H;
    int64_t ldq, ldk, ldv, ldo;              // elements between consecutive tokens of Q, K, V, dO
    int64_t lddq, lddk, lddv;                // ... of the outputs
    float scale, scale_log2e;
    float bias_in_l2, bias_out_l2;           // MASKED: additive bias * log2(e) inside / outside the kept region (ar_attn_fwd_masked)
    int valid_len;                           // MASKED: keys >= valid_len are outside
};

// Dv[b, h, s] = -sum_d dO * O (fp32), L2 = -lse / scale.  One 8-lane group per (token, head) row of 64 values.
__global__ __launch_bounds__(kTPB) void k_attn_bwd_prep(const uint16_t* __restrict__ dO, int64_t ldo, const uint16_t* __restrict__ O, int64_t ldO,
                                                         const float* __restrict__ lse, float* __restrict__ Dv, float* __restrict__ L2,
                                                         int B, int S, int H, int AD, float inv_scale) {
    const int lpr = AD / 8;                                          // lanes per row
    const int64_t row = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / lpr;
    const int part = threadIdx.x % lpr;
    const int64_t rows = (int64_t)B * S * H;
    float s = 0.f;
    if (row < rows) {
        const int64_t tok = row / H;
        const int h = (int)(row % H);
        float a[8], b[8];
        unpack8<AR_DT_BF16>(load8_raw<AR_DT_BF16>(dO, tok * ldo + (int64_t)h * AD + part * 8), a);
        unpack8<AR_DT_BF16>(load8_raw<AR_DT_BF16>(O, tok * ldO + (int64_t)h * AD + part * 8), b);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k] * b[k];
    }
    for (int m = 1; m < lpr; m <<= 1) s += __shfl_xor(s, m, kWave);
    if (row < rows && part == 0) {
        const int64_t tok = row / H;
        const int h = (int)(row % H);
        const int64_t b = tok / S, sq = tok % S;
        const int64_t o = (b * H + h) * S + sq;
        Dv[o] = -s;                          // both rows negated: they are the INITIAL VALUES of the score accumulators (below)
        L2[o] = -(lse[o] * inv_scale);
    }
}

// MASKED (round 4): the calibration flow's structured additive mask instead of causality -- bias(q, k) = bias_in where `k <= q and
// k < valid_len`, bias_out elsewhere, both finite (csrc/ar_attn.hip k_attn_fwd<.., MASKED>): every (query, key) pair contributes, no
// tile is skipped, P = exp2((q.k - lse / scale) * c + bias * log2 e); dS = P (dP - D) as before (the bias is a constant).
// (Head size 128, masked form: the two-kernel skeleton -- 16 resident b-operand registers + 8 accumulator tiles per lane -- does not fit
//  the 256 registers a wave has at two waves per SIMD (MODE 1 spilled 157); see OUT below.)
// OUT (MODE 1): 3 = dK and dV (head size 64), 1 = dV only, 2 = dK only -- head size 128 runs the key side as TWO kernels, each
// with one resident operand set less and half the accumulators (8 GEMM passes instead of 7), which is what fits 256 registers.
template <int MODE, int WAVES, int AD, bool MASKED = false, int OUT = 3>
__global__ __launch_bounds__(64 * WAVES, 2) void k_attn_bwd(AttnBwdArgs a) {
    constexpr bool NEED_S1 = MODE == 0 || (OUT & 2);      // the dP = dO . V product (feeds dQ and dK)
    constexpr bool NEED_A0 = MODE == 0 || (OUT & 2);      // acc0: dQ (MODE 0) / dK
    constexpr bool NEED_A1 = MODE == 1 && (OUT & 1);      // acc1: dV
    constexpr int KB = AD == 128 ? 4 : AD / 16;      // k-steps of a score product whose row fragments are in flight together
    constexpr int AROW = AD * 2;         // bytes per staged row
    constexpr int ATILE = BK * AROW;     // one tensor's tile
    constexpr int ABUF = 2 * ATILE;      // R0 tile + R1 tile; two buffers
    constexpr int NKS = AD / 16;         // k-steps of a score product
    constexpr int ND = AD / 32;          // d-tiles of an accumulator
    constexpr int AQ = 32 * WAVES;       // own rows per workgroup
    constexpr int RPW = BK / WAVES;      // tile rows staged per wave
    constexpr int RPI = 1024 / AROW;     // rows per DMA instruction
    constexpr int CPR = AROW / 16;
    constexpr int NP = RPW / RPI;
    static_assert(NP >= 1, "a wave stages at least one DMA instruction per tensor");
    static_assert(AD == 64 || (AD == 128 && (MODE == 0 || OUT != 3)), "head size 128: the key side as two kernels (register budget)");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, lq = lane & 31;
    // workgroup -> (batch * head, block of own rows): whole heads per XCD, long (causal) blocks first -- MODE 0: late query blocks,
    // MODE 1: early key blocks
    const int n_ob = a.S / AQ;
    const int n_bh = a.B * a.H;
    int ob, bh;
    if ((n_bh & 7) == 0) {
        const int j = blockIdx.x >> 3, per = n_bh >> 3;
        bh = (j % per) * 8 + (int)(blockIdx.x & 7);
        ob = j / per;
    } else {
        ob = (int)(blockIdx.x % n_ob);
        bh = blockIdx.x / n_ob;
    }
    if (MODE == 0) ob = n_ob - 1 - ob;
    const int b = bh / a.H, head = bh % a.H;
    const int o0 = ob * AQ;                                           // first own row of the workgroup
    const int myrow = o0 + 32 * wave + lq;                            // this lane's own row (query in MODE 0, key in MODE 1)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    constexpr int VEC_OFF = 2 * ABUF;                                 // MODE 1: L2 row [S] then D row [S] behind the tile buffers

    // ---- streamed tensors and resident operands
    const int64_t ld0 = MODE == 0 ? a.ldk : a.ldq, ld1 = MODE == 0 ? a.ldv : a.ldo;
    const uint16_t* R0b = (MODE == 0 ? a.K : a.Q) + (int64_t)b * a.S * ld0 + head * AD;
    const uint16_t* R1b = (MODE == 0 ? a.V : a.dO) + (int64_t)b * a.S * ld1 + head * AD;
    const int64_t lb0 = MODE == 0 ? a.ldq : a.ldk, lb1 = MODE == 0 ? a.ldo : a.ldv;
    const uint16_t* B0b = (MODE == 0 ? a.Q : a.K) + (int64_t)b * a.S * lb0 + head * AD;
    const uint16_t* B1b = (MODE == 0 ? a.dO : a.V) + (int64_t)b * a.S * lb1 + head * AD;
    bbf16x8_t bf0[NKS], bf1[NKS];        // b-operands: n = own row = lane & 31, k = d = 16 ks + 8 h .. +7
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        bf0[ks] = __builtin_bit_cast(bbf16x8_t, *reinterpret_cast<const uint4*>(B0b + (int64_t)myrow * lb0 + 16 * ks + 8 * h));
        if (NEED_S1) bf1[ks] = __builtin_bit_cast(bbf16x8_t, *reinterpret_cast<const uint4*>(B1b + (int64_t)myrow * lb1 + 16 * ks + 8 * h));
    }
    const float* L2row = a.L2 + ((int64_t)b * a.H + head) * a.S;
    const float* Dvrow = a.Dv + ((int64_t)b * a.H + head) * a.S;
    float myL2 = 0.f, myD = 0.f;
    if (MODE == 0) { myL2 = L2row[myrow]; myD = Dvrow[myrow]; }
    if (MODE == 1) {      // the head's whole L2 / D rows into LDS (S <= 4096: 32 KB), read per streamed row below
        float* vL = reinterpret_cast<float*>(lds + VEC_OFF);
        float* vD = vL + a.S;
        for (int i = tid; i < a.S; i += 64 * WAVES) { vL[i] = L2row[i]; vD[i] = Dvrow[i]; }
    }

    // ---- DMA: lane -> row RPI p + lane / CPR of the wave's RPW rows, physical chunk lane % CPR (swizzled source column)
    const int drow = lane / CPR, pchunk = lane % CPR;
    uint32_t doff0[NP], doff1[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = RPW * wave + RPI * p + drow;
        doff0[p] = (uint32_t)(r * ld0 + (pchunk ^ battn_swz<AD>(r)) * 8);
        doff1[p] = (uint32_t)(r * ld1 + (pchunk ^ battn_swz<AD>(r)) * 8);
    }
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) {
            const int p = j % NP;
            const uint16_t* T = (j < NP ? R0b + (int64_t)t * BK * ld0 + doff0[p] : R1b + (int64_t)t * BK * ld1 + doff1[p]);
            const uint32_t dst = lds0 + buf * ABUF + (j < NP ? 0 : ATILE) + (RPW * wave + RPI * p) * AROW;     // wave-uniform
            __builtin_amdgcn_global_load_lds((const void*)T, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
        }
    };

    // ---- fragment read addresses (the forward's): row fragments (a-operand of a score product: m = streamed row = 32 t + lq,
    // k = d = 16 ks + 8 h .. +7) and transposed fragments (a-operand of an accumulating product: m = d, k = streamed rows)
    uint32_t rA[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) rA[ks] = lds0 + lq * AROW + ((uint32_t)((2 * ks + h) ^ battn_swz<AD>(lq)) << 4);
    const int gi = lane & 15, gg = lane >> 4;
    const int vrow = 4 * h + (gi >> 2);
    const int vcol0 = 16 * (gg & 1) + 4 * (gi & 3);
    uint32_t tAlo[ND], tAhi[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        const int col = 32 * dt + vcol0;
        tAlo[dt] = lds0 + vrow * AROW + ((uint32_t)((col >> 3) ^ battn_swz<AD>(vrow)) << 4) + (col & 7) * 2;
        tAhi[dt] = lds0 + (vrow + 8) * AROW + ((uint32_t)((col >> 3) ^ battn_swz<AD>(vrow + 8)) << 4) + (col & 7) * 2;
    }

    bf32x16_t acc0[ND], acc1[ND];        // MODE 0: acc0 = dQ^T;  MODE 1: acc0 = dK^T, acc1 = dV^T
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[dt][r] = 0.f; acc1[dt][r] = 0.f; }

#define BW_PIN() __builtin_amdgcn_sched_barrier(0)
#define BW_RREAD(DST, KS, T, REG) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(rA[KS]), "n"(BUF * ABUF + (REG) * ATILE + (T) * 32 * AROW) : "memory")
#define BW_TREAD(LO, HI, DT, ST, REG)                                                                                          \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"                                  \
                 : "=&v"(LO), "=&v"(HI) : "v"(tAlo[DT]), "v"(tAhi[DT]), "n"(BUF * ABUF + (REG) * ATILE + (ST) * 16 * AROW) : "memory")

    // one streamed tile from LDS buffer BUF (compile time); t0 = its first row
    auto tile = [&](auto bufc, int t0) {
        constexpr int BUF = decltype(bufc)::value;
        // MODE 0: streamed rows are keys, own row the query -> masked where key > query.  MODE 1: streamed rows are queries, own row
        // the key -> masked where key > query as well.
        // MASKED: `diag` = the tile is not entirely inside the kept region for this wave (some pair has key > query or key >= valid_len)
        const bool diag = MASKED ? (MODE == 0 ? (t0 + BK - 1 > o0 + 32 * wave || t0 + BK > a.valid_len)
                                              : (t0 < o0 + 32 * wave + 31 || o0 + 32 * wave + 31 >= a.valid_len))
                                 : (MODE == 0 ? (t0 + BK - 1 > o0 + 32 * wave) : (t0 < o0 + 32 * wave + 31));
#pragma unroll
        for (int t = 0; t < 2; ++t) {                                 // the two 32-row halves of the tile, one after the other
            // The score accumulators START at -lse / scale and -D of their query row, so the products come out as
            // (q.k - lse / scale) and (dO.v - D): the softmax shift and the "- D" of dS cost no VALU instruction (at head size 64
            // the elementwise step, not the MFMA pipe, bounds this kernel: exp2 issues at quarter rate).
            bf32x16_t s0, s1;
            u32x4_t f0[KB], f1[KB];                                   // row fragments of KB k-steps (head size 128: two batches -- registers)
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                BW_RREAD(f0[ks], ks, t, 0);
                if (NEED_S1) BW_RREAD(f1[ks], ks, t, 1);
            }
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] = myL2; s1[r] = myD; }
            } else {                                                  // L2 / D of this half's 16 streamed rows: LDS -> accumulator
                const float* vL = reinterpret_cast<const float*>(lds + VEC_OFF);
                const float* vD = vL + a.S;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 x = *reinterpret_cast<const float4*>(vL + t0 + 32 * t + 4 * h + 8 * j);
                    s0[4 * j] = x.x; s0[4 * j + 1] = x.y; s0[4 * j + 2] = x.z; s0[4 * j + 3] = x.w;
                    if (NEED_S1) {
                        const float4 y = *reinterpret_cast<const float4*>(vD + t0 + 32 * t + 4 * h + 8 * j);
                        s1[4 * j] = y.x; s1[4 * j + 1] = y.y; s1[4 * j + 2] = y.z; s1[4 * j + 3] = y.w;
                    }
                }
            }
#pragma unroll
            for (int kb = 0; kb < NKS; kb += KB) {
                if (kb > 0) {
#pragma unroll
                    for (int ks = 0; ks < KB; ++ks) {
                        BW_RREAD(f0[ks], kb + ks, t, 0);
                        if (NEED_S1) BW_RREAD(f1[ks], kb + ks, t, 1);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                BW_PIN();
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bbf16x8_t, f0[ks]), bf0[kb + ks], s0, 0, 0, 0);
                    if (NEED_S1) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bbf16x8_t, f1[ks]), bf1[kb + ks], s1, 0, 0, 0);
                }
                if (kb + KB < NKS) BW_PIN();
            }
            // transposed fragments of this half: 2 sixteen-row steps x ND d-tiles x (R0 and, in MODE 1, R1): in flight under the
            // elementwise step
            // (head size 128: the second step's fragments are read under the first step's MFMAs instead -- registers)
            constexpr bool LATE_Q = AD == 128;
            bs16x4_t q0lo[2][ND], q0hi[2][ND], q1lo[2][ND], q1hi[2][ND];
#pragma unroll
            for (int s2 = 0; s2 < (LATE_Q ? 1 : 2); ++s2)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    if (NEED_A0) BW_TREAD(q0lo[s2][dt], q0hi[s2][dt], dt, 2 * t + s2, 0);
                    if (NEED_A1) BW_TREAD(q1lo[s2][dt], q1hi[s2][dt], dt, 2 * t + s2, 1);
                }
            BW_PIN();
            // ---- P = exp2(s0 * c) (0 where key > query), E = P * s1; register r <-> streamed row 32 t + 4 h + (r & 3) + 8 (r >> 2)
            if (MASKED) {
                if (diag) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int srow = t0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                        const int query = MODE == 0 ? myrow : srow, key = MODE == 0 ? srow : myrow;
                        const float bias = (key <= query && key < a.valid_len) ? a.bias_in_l2 : a.bias_out_l2;
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], a.scale_log2e, bias));
                        s0[r] = p;
                        if (NEED_S1) s1[r] = p * s1[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], a.scale_log2e, a.bias_in_l2));
                        s0[r] = p;
                        if (NEED_S1) s1[r] = p * s1[r];
                    }
                }
            } else if (diag) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int srow = t0 + 32 * t + 4 * h + (r & 3) + 8 * (r >> 2);
                    const bool masked = MODE == 0 ? (srow > myrow) : (myrow > srow);
                    const float p = masked ? 0.f : __builtin_amdgcn_exp2f(s0[r] * a.scale_log2e);
                    s0[r] = p;
                    if (NEED_S1) s1[r] = p * s1[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s0[r] * a.scale_log2e);
                    s0[r] = p;
                    if (NEED_S1) s1[r] = p * s1[r];
                }
            }
            BW_PIN();
            // ---- accumulate: k = this half's 32 streamed rows in two steps of 16 (the accumulator's row order is the k order)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bs16x8_t pe, pp;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    if (NEED_A0) {
                        const uint32_t we = pack_bf16x2(s1[8 * s2 + e], s1[8 * s2 + e + 1]);
                        pe[e] = (short)(we & 0xffffu); pe[e + 1] = (short)(we >> 16);
                    }
                    if (NEED_A1) {
                        const uint32_t wp = pack_bf16x2(s0[8 * s2 + e], s0[8 * s2 + e + 1]);
                        pp[e] = (short)(wp & 0xffffu); pp[e + 1] = (short)(wp >> 16);
                    }
                }
                // (the second step's reads may stay in flight; the counter field holds at most 15)
                constexpr int INFL = ((NEED_A0 ? 2 : 0) + (NEED_A1 ? 2 : 0)) * ND;
                if (LATE_Q && s2 == 0) {
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        if (NEED_A0) BW_TREAD(q0lo[1][dt], q0hi[1][dt], dt, 2 * t + 1, 0);
                        if (NEED_A1) BW_TREAD(q1lo[1][dt], q1hi[1][dt], dt, 2 * t + 1, 1);
                    }
                }
                if (s2 == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(INFL > 15 ? 15 : INFL) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                BW_PIN();
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    if (NEED_A0) {
                        const bs16x8_t a0 = __builtin_shufflevector(q0lo[s2][dt], q0hi[s2][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc0[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bbf16x8_t, a0), __builtin_bit_cast(bbf16x8_t, pe), acc0[dt], 0, 0, 0);
                    }
                    if (NEED_A1) {
                        const bs16x8_t a1 = __builtin_shufflevector(q1lo[s2][dt], q1hi[s2][dt], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bbf16x8_t, a1), __builtin_bit_cast(bbf16x8_t, pp), acc1[dt], 0, 0, 0);
                    }
                }
                BW_PIN();
            }
        }
    };

    // ---- tile loop.  MODE 0: key tiles 0 .. diagonal;  MODE 1: query tiles from the diagonal to the end.  Always an even count.
    const int t_first = (MODE == 0 || MASKED) ? 0 : o0 / BK;
    const int t_end = (MODE == 0 && !MASKED) ? (o0 + AQ) / BK : a.S / BK;
    __syncthreads();                                                  // (MODE 1: the L2 / D rows are in LDS)
    issue_tile(t_first, 0);
    for (int t = t_first; t < t_end; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // tile t landed for everybody; buffer 1 is free
        issue_tile(t + 1, 1);
        {   // wave-uniform skip: no (key <= query) pair between the wave's own rows and this tile
            const int t0 = t * BK;
            const bool live = MASKED || (MODE == 0 ? (t0 <= o0 + 32 * wave + 31) : (t0 + BK - 1 >= o0 + 32 * wave));
            if (live) tile(std::integral_constant<int, 0>{}, t0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < t_end) issue_tile(t + 2, 0);
        {
            const int t0 = (t + 1) * BK;
            const bool live = MASKED || (MODE == 0 ? (t0 <= o0 + 32 * wave + 31) : (t0 + BK - 1 >= o0 + 32 * wave));
            if (live) tile(std::integral_constant<int, 1>{}, t0);
        }
    }
#undef BW_RREAD
#undef BW_TREAD
#undef BW_PIN
    // ---- store: lane = own row, registers = d (runs of 4 consecutive d -> 8-byte stores)
    auto store = [&](uint16_t* base, int64_t ld, const bf32x16_t (&acc)[ND], float mul) {
        uint16_t* orow = base + ((int64_t)b * a.S + myrow) * ld + head * AD;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 w;
                w.x = pack_bf16x2(acc[dt][4 * j + 0] * mul, acc[dt][4 * j + 1] * mul);
                w.y = pack_bf16x2(acc[dt][4 * j + 2] * mul, acc[dt][4 * j + 3] * mul);
                *reinterpret_cast<uint2*>(orow + 32 * dt + 8 * j + 4 * h) = w;
            }
    };
    if (MODE == 0) store(a.dQ, a.lddq, acc0, a.scale);
    else {
        if (NEED_A0) store(a.dK, a.lddk, acc0, a.scale);
        if (NEED_A1) store(a.dV, a.lddv, acc1, 1.0f);
    }
}

}  // namespace ar

using namespace ar;

extern "C" int64_t ar_attn_bwd_workspace_bytes(int64_t B, int64_t S, int64_t H) { return 2 * B * S * H * (int64_t)sizeof(float); }

static int attn_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                         void* dV, int64_t B, int64_t S, int64_t H, int64_t D, float scale, int causal, int64_t ldq, int64_t ldk,
                         int64_t ldv, int64_t ldo_, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, void* workspace,
                         int64_t workspace_bytes, bool masked, float bias_in, float bias_out, int64_t valid_len, ar_stream_t stream);

extern "C" int ar_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                           void* dV, int64_t B, int64_t S, int64_t H, int64_t D, float scale, int causal, int64_t ldq, int64_t ldk,
                           int64_t ldv, int64_t ldo_, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, void* workspace,
                           int64_t workspace_bytes, ar_stream_t stream) {
    if (D != 64) return AR_ERR_UNSUPPORTED;
    return attn_bwd_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, B, S, H, D, scale, causal, ldq, ldk, ldv, ldo_, lddo, lddq, lddk, lddv, workspace,
                         workspace_bytes, false, 0.f, 0.f, S, stream);
}

extern "C" int ar_attn_bwd_masked(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                                  void* dK, void* dV, int64_t B, int64_t S, int64_t H, int64_t D, float scale, float bias_in, float bias_out,
                                  int64_t valid_len, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo_, int64_t lddo, int64_t lddq,
                                  int64_t lddk, int64_t lddv, void* workspace, int64_t workspace_bytes, ar_stream_t stream) {
    if (!(bias_in == bias_in) || !(bias_out == bias_out) || fabsf(bias_in) > 1e4f || fabsf(bias_out) > 1e4f || valid_len < 1 || valid_len > S)
        return AR_ERR_UNSUPPORTED;
    return attn_bwd_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, B, S, H, D, scale, 1, ldq, ldk, ldv, ldo_, lddo, lddq, lddk, lddv, workspace,
                         workspace_bytes, true, bias_in, bias_out, valid_len, stream);
}

static int attn_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ, void* dK,
                         void* dV, int64_t B, int64_t S, int64_t H, int64_t D, float scale, int causal, int64_t ldq, int64_t ldk,
                         int64_t ldv, int64_t ldo_, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, void* workspace,
                         int64_t workspace_bytes, bool masked, float bias_in, float bias_out, int64_t valid_len, ar_stream_t stream) {
    if ((D != 64 && !(D == 128 && masked)) || !causal || S % 256 || S > 4096 || B <= 0 || H <= 0) return AR_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ar_attn_bwd_workspace_bytes(B, S, H)) return AR_ERR_UNSUPPORTED;
    const int64_t hd = H * D;
    if (ldq <= 0) ldq = hd;
    if (ldk <= 0) ldk = hd;
    if (ldv <= 0) ldv = hd;
    if (ldo_ <= 0) ldo_ = hd;
    if (lddo <= 0) lddo = hd;
    if (lddq <= 0) lddq = hd;
    if (lddk <= 0) lddk = hd;
    if (lddv <= 0) lddv = hd;
    const int64_t lds_[8] = {ldq, ldk, ldv, ldo_, lddo, lddq, lddk, lddv};
    for (int i = 0; i < 8; ++i)
        if (lds_[i] < hd || (lds_[i] % 8) || 64 * lds_[i] > 0x7fffffffLL) return AR_ERR_UNSUPPORTED;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15)
        return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    float* Dv = (float*)workspace;
    float* L2 = Dv + B * S * H;
    {
        const int64_t rows = B * S * H, lanes = rows * (D / 8);
        hipLaunchKernelGGL(k_attn_bwd_prep, (int)((lanes + kTPB - 1) / kTPB), kTPB, 0, st, (const uint16_t*)dO, lddo, (const uint16_t*)O, ldo_,
                           LSE, Dv, L2, (int)B, (int)S, (int)H, (int)D, 1.0f / scale);
    }
    AttnBwdArgs a;
    a.Q = (const uint16_t*)Q; a.K = (const uint16_t*)K; a.V = (const uint16_t*)V; a.dO = (const uint16_t*)dO;
    a.L2 = L2; a.Dv = Dv;
    a.dQ = (uint16_t*)dQ; a.dK = (uint16_t*)dK; a.dV = (uint16_t*)dV;
    a.B = (int)B; a.S = (int)S; a.H = (int)H;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = lddo;
    a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    a.bias_in_l2 = bias_in * 1.4426950408889634f; a.bias_out_l2 = bias_out * 1.4426950408889634f; a.valid_len = (int)valid_len;
    constexpr int LDS_T = 4 * BK * 64 * 2;                            // 2 buffers x (R0 tile + R1 tile) at head size 64
    constexpr int LDS_T128 = 4 * BK * 128 * 2;
    const int vec = (int)(2 * S * sizeof(float));
    static PerDeviceOnce attr;
    if (attr.first()) {
        constexpr int VEC_MAX = 2 * 4096 * (int)sizeof(float);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd<0, 8, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd<1, 8, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd<0, 8, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd<1, 8, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd<0, 8, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd<1, 8, 128, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + VEC_MAX);
        (void)hipFuncSetAttribute((const void*)k_attn_bwd<1, 8, 128, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_T128 + VEC_MAX);
    }
    const int grid = (int)(B * H * (S / 256));
    if (D == 128) {                                                    // (masked only) the key side as two kernels, then the query side
        hipLaunchKernelGGL((k_attn_bwd<1, 8, 128, true, 1>), grid, 512, LDS_T128 + vec, st, a);
        hipLaunchKernelGGL((k_attn_bwd<1, 8, 128, true, 2>), grid, 512, LDS_T128 + vec, st, a);
        hipLaunchKernelGGL((k_attn_bwd<0, 8, 128, true>), grid, 512, LDS_T128, st, a);
    } else if (masked) {
        hipLaunchKernelGGL((k_attn_bwd<1, 8, 64, true>), grid, 512, LDS_T + vec, st, a);
        hipLaunchKernelGGL((k_attn_bwd<0, 8, 64, true>), grid, 512, LDS_T, st, a);
    } else {
        hipLaunchKernelGGL((k_attn_bwd<1, 8, 64>), grid, 512, LDS_T + vec, st, a);
        hipLaunchKernelGGL((k_attn_bwd<0, 8, 64>), grid, 512, LDS_T, st, a);
    }
    return launch_status();
}
