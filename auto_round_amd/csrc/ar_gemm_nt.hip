// ar_gemm_nt.hip -- hand-written MFMA "NT" GEMM for gfx950:  C[M,N] = A[M,K] B[N,K]^T, both operands K-contiguous bf16, fp32
// accumulation, one rounding to bf16.  SURVEY section 8 row f1, forward side (round 5).
//
// replaces: the forward of F.linear(x, weight_q) inside WrapperLinear.forward (auto_round/wrapper.py:528-556: x [tokens, in] against
//           the fake-quant weight [out, in]) and, through a transposed weight copy, its input gradient dX = dY Wq; grouped over the
//           experts of a sparse-MoE block it replaces the Python loop of per-expert GEMMs of the reference's "linear loop" experts
//           (auto_round/modeling/fused_moe/moe_experts_interface.py:173-289): one launch per projection, row ranges read from device
//           memory (no host synchronisation, deterministic).
//
// Not the weight-gradient kernel (csrc/ar_gemm.hip) fed transposed operands -- round 3's A/B -- but a kernel designed for K-contiguous
// rows (CDNA4-first):
//   * 256 x 256 output tile per 512-thread workgroup, 8 waves as 2 (m) x 4 (n), wave tile 128 x 64 = acc[8][4] of
//     v_mfma_f32_16x16x32_bf16 in the default form (M16; acc[4][2] of v_mfma_f32_32x32x16_bf16 in the round-5 first cut, still
//     selectable: identical bits) -- a-operand = B rows, b-operand = A rows: a lane then owns ONE output row and runs of 4 consecutive n.
//   * K is staged 64 deep: a stage is four HALF-tiles of [128 rows][128 bytes] (A rows 0-127 / 128-255, B rows likewise), 16 KB each,
//     two stages in 128 KB of LDS.  Every row of a half-tile is one full 128-byte line of HBM -- LDS-DMA (global_load_lds, 16 bytes
//     per lane, no staging VGPRs) moves 8 rows per wave-instruction.
//   * the fragments are plain ds_read_b128 (8 consecutive k of one row = one lane's MFMA operand; no transposing read).  Rows are 128
//     bytes apart, so the 16-byte chunk index is XOR-swizzled with bits 1..3 of the row: chunk ^= (row >> 1) & 7, applied to the
//     per-lane SOURCE address of the DMA (the chunks of a row are permuted inside their own 128-byte line: coalescing is untouched)
//     and to the read address; every 16-lane group of a read then covers all 16 bank slots (checked in tools/gemm_nt_index_model.py).
//   * a phase is K = 32: 12 fragment reads (L part), a barrier, 16 MFMAs with the wave's 4 DMA pieces issued in between (M part), a
//     barrier.  The two wave groups (m halves) run half a period apart, so one wave of a SIMD is in its MFMA cluster while the other
//     reads: the same skeleton as the weight-gradient kernel.
//   * who stages what is chosen so that no wave ever writes a buffer another group may still read, with two stages only: group g
//     stages A-half g (read by group g alone) and B-half g (read by everybody).  On odd phases (2t+1) a wave issues its "X" pieces
//     of stage t+2 (group 0: A, group 1: B), on even phases (2t) its "Y" pieces of stage t+1 (group 0: B, group 1: A); counted
//     s_waitcnt vmcnt(4) at the end of the L and M parts of odd phases -- never 0 in the loop.  The hazard table is in DESIGN.md.
//   * XCD-aware tile order: each XCD (own L2) walks 4 x 8-tile patches, so its 32 resident workgroups share 4 A panels and 8 B panels.
//   * rows past the end of a (ragged) row range are clamped on the load side and masked on the store side, so M need not be a
//     multiple of 256 and the grouped form needs no padding between experts.
#include "ar_common.hpp"

namespace ar {

typedef __bf16 nt_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float nt_f32x16_t __attribute__((ext_vector_type(16)));
typedef float nt_f32x4_t __attribute__((ext_vector_type(4)));

constexpr int NT_B = 256;                      // tile edge
constexpr int NT_K = 64;                       // k per stage
constexpr int NT_HALF = 128 * 128;             // bytes of a half-tile
constexpr int NT_LDS = 8 * NT_HALF;            // 131072
constexpr int NT_THREADS = 512;

struct NtArgs {
    const uint16_t* A;        // [M, K] (lda)
    const uint16_t* B;        // [N, K] (ldb); grouped: group e's matrix starts at B + b_off[e]
    uint16_t* C;              // [M, N] (ldc)
    int64_t lda, ldb, ldc;
    int M, N, K;              // grouped: M = total rows of all groups
    int tiles_m, tiles_n, order;
    const int32_t* row_off;   // grouped: [E + 1] prefix sums of the groups' row counts (device memory)
    const int64_t* b_off;     // grouped: [E] element offsets (device memory)
    int E;
    unsigned long long* trace;   // TRACE builds: [workgroup][wave][8] cycle sums (s_memtime): L part, barrier after L, M part, barrier after M,
                                 // whole loop, phases, the whole loop in 100 MHz ticks (s_memrealtime), 0
};

template <int N>
__device__ __forceinline__ void nt_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// DMAL = false: the protocol of the header (4 pieces per M part, streams X / Y).  DMAL = true (see the note at k_gemm_dw4): the M part
// is 16 bare MFMAs and ALL 8 pieces of stage tau+1 (the wave's B pieces, then its A pieces) are issued at the end of the L part of the
// even phase 2 tau, after the fragment reads returned: the buffer is the one stage tau-1 lived in, whose last reads (L(2 tau - 1) of both
// groups) are behind the barrier this wave passed to enter L(2 tau); `vmcnt(4)` at the end of L(2 tau + 1) retires the B pieces (read by
// both groups from L(2 tau + 2) on), `vmcnt(0)` at the end of M(2 tau + 1) the A pieces (read by the issuing group alone).
// TRACE (ar_gemm_nt_trace, tools only): every wave sums, over all its phases, the shader cycles (s_memtime) it spends in the four segments
// of a phase -- fragment reads until they have returned; parked at the barrier that ends the L part; the 16 MFMAs (+ DMA issues) until
// the last one is issued; parked at the barrier that ends the M part -- the per-phase cycle table of DESIGN.md section 3.
// M16 (round 5): the same kernel on v_mfma_f32_16x16x32_bf16 -- the shape this chip sustains at a higher clock on random operands
// (tools/mfma_power.hip; the note at k_gemm_dw6 in ar_gemm.hip).  Same LDS image, same DMA protocol; a phase is still K = 32 and 12
// fragment reads, now one ds_read_b128 per 16-row fragment (lane = row lane % 16, k chunk lane / 16: the operand's k = 8 (lane / 16) + j
// layout; 8 A + 4 B fragments), then 32 MFMAs into acc[8][4] of 16 x 16; a lane owns output row m = lane % 16 of a fragment and the
// 4 consecutive n = 4 (lane / 16) .. +3.
template <bool GROUPED, bool DMAL, bool TRACE = false, bool M16 = false>
__global__ __launch_bounds__(NT_THREADS, 2) void k_gemm_nt(NtArgs a) {
    static_assert(!(M16 && DMAL), "the 16x16x32 form keeps the DMA pieces in the MFMA part");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- which tile
    int64_t m0;
    int n0, rows_valid;
    const uint16_t* Bp = a.B;
    if (GROUPED) {
        const int nwg = gridDim.x;
        int id = blockIdx.x;
        if ((nwg & 7) == 0) id = (id & 7) * (nwg >> 3) + (id >> 3);            // a contiguous run of tiles per XCD
        // bands of 8 column tiles, row tiles inside a band, columns fastest: 32 consecutive ids = 4 row tiles x 8 column tiles, i.e. the
        // 32 workgroups an XCD runs at a time share 4 A panels and 8 B panels (walking a whole row of 112 column tiles first -- the first
        // cut -- made them fetch 1 + 32 panels per stage: 0.96 PF on Mixtral's merged gate|up against 1.11 on the down projection)
        const int bw = (a.tiles_n % 8 == 0) ? 8 : a.tiles_n;
        const int per_band = a.tiles_m * bw;
        const int band = id / per_band, rem = id - band * per_band;
        const int gmt = rem / bw, tn = band * bw + (rem - gmt * bw);
        if (tn >= a.tiles_n) return;
        int e = -1, base = 0, r0 = 0, r1 = 0;
        for (int i = 0; i < a.E; ++i) {
            const int s0 = a.row_off[i], s1 = a.row_off[i + 1];
            const int mt = (s1 - s0 + NT_B - 1) / NT_B;
            if (e < 0 && gmt < base + mt) { e = i; r0 = s0 + (gmt - base) * NT_B; r1 = s1; }
            base += mt;
        }
        if (e < 0) return;                                                     // past the last tile of the last group (whole workgroup)
        m0 = r0;
        rows_valid = (r1 - r0) < NT_B ? (r1 - r0) : NT_B;
        n0 = tn * NT_B;
        Bp = a.B + a.b_off[e];
    } else {
        const int nwg = a.tiles_m * a.tiles_n;
        const int bid = blockIdx.x;
        int tm, tn;
        if (a.order == 2 && (a.tiles_m % 4 == 0) && (a.tiles_n % 8 == 0) && (nwg % 256 == 0)) {
            const int x = bid & 7, s = bid >> 3, pi = s >> 5, w = s & 31;
            const int P = pi * 8 + x, pn = a.tiles_n >> 3;
            tm = (P / pn) * 4 + (w >> 3);
            tn = (P % pn) * 8 + (w & 7);
        } else {
            int id = bid;
            if (a.order >= 1 && nwg % 8 == 0) id = (bid & 7) * (nwg >> 3) + (bid >> 3);
            tm = id / a.tiles_n;
            tn = id % a.tiles_n;
        }
        m0 = (int64_t)tm * NT_B;
        n0 = tn * NT_B;
        const int64_t left = (int64_t)a.M - m0;
        rows_valid = left < NT_B ? (int)left : NT_B;
    }
    const int T = a.K / NT_K;                       // stages; even (K % 128 == 0, checked by the host)

    // ---- DMA: this wave's four pieces of a half-tile are rows wc*32 + 8j + (lane >> 3), j = 0..3; lane -> physical chunk lane & 7
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    uint32_t voffA[4], voffB[4];
    {
        const int rr = wc * 32 + (lane >> 3);
        const int pc = lane & 7;
        const int64_t last = (int64_t)a.M - 1 - m0;             // last row of A that exists, relative to the tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lc = pc ^ ((4 * j + (lane >> 4)) & 7);
            int64_t ra = wr * 128 + rr + 8 * j;
            if (ra > last) ra = last;                           // (rows past the end: any existing row -- their results are never stored)
            voffA[j] = (uint32_t)(ra * a.lda * 2 + lc * 16);
            voffB[j] = (uint32_t)((int64_t)(wr * 128 + rr + 8 * j) * a.ldb * 2 + lc * 16);
        }
    }
    const uint8_t* gA = reinterpret_cast<const uint8_t*>(a.A + m0 * a.lda);
    const uint8_t* gB = reinterpret_cast<const uint8_t*>(Bp + (int64_t)n0 * a.ldb);
    // the two piece streams of this wave: X on odd phases, Y on even phases (group 0: X = A, Y = B; group 1: X = B, Y = A)
    const uint8_t* gX = wr ? gB : gA;
    const uint8_t* gY = wr ? gA : gB;
    uint32_t voffX[4], voffY[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        voffX[j] = wr ? voffB[j] : voffA[j];
        voffY[j] = wr ? voffA[j] : voffB[j];
    }
    const uint32_t dstA = lds0 + wr * 32768 + wc * 32 * 128;
    const uint32_t dstB = dstA + 65536;
    const uint32_t dstX = wr ? dstB : dstA, dstY = wr ? dstA : dstB;
    int sx = 0, sy = 0;                              // next stage of each stream
    const uint8_t* gA2 = gA;                         // DMAL: running pointers of the two operands (stage sx)
    const uint8_t* gB2 = gB;

    // ---- fragment read addresses: X[a][u] = row * 128 + 16 * ((4a + 2u + h) ^ s)
    uint32_t adA[2][2], adB[2][2];
    {
        const int l31 = lane & 31, h = lane >> 5, s = (l31 >> 1) & 7;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t X = l31 * 128 + 16 * ((4 * ah + 2 * u + h) ^ s);
                adA[ah][u] = lds0 + wr * 32768 + X;
                adB[ah][u] = lds0 + 65536 + (wc >> 1) * 32768 + (wc & 1) * 8192 + X;
            }
    }

    uint32_t adA16[2], adB16[2];      // M16: row = 16 f + lane % 16 (fragment f by immediate offset), chunk (4 AH + lane / 16) ^ ((lane % 16) >> 1)
    {
        const int l15 = lane & 15, kg = lane >> 4, s = l15 >> 1;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
            const uint32_t X = l15 * 128 + 16 * ((4 * ah + kg) ^ s);
            adA16[ah] = lds0 + wr * 32768 + X;
            adB16[ah] = lds0 + 65536 + (wc >> 1) * 32768 + (wc & 1) * 8192 + X;
        }
    }

    nt_f32x16_t acc[M16 ? 1 : 4][2];
    nt_f32x4_t acc16[M16 ? 8 : 1][4];
#pragma unroll
    for (int mi = 0; mi < (M16 ? 1 : 4); ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
    for (int mi = 0; mi < (M16 ? 8 : 1); ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[mi][ni][r] = 0.f;

    // one piece (1 KB: 8 rows x 128 B) of a stream; SO = byte offset of the stage buffer (0 / 16384), J = piece
#define NT_ISSUE_X(SO, J)                                                                                               \
    __builtin_amdgcn_global_load_lds((const void*)(gX + voffX[J]),                                                      \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(dstX + (SO) + (J) * 1024), 16, 0, 0)
#define NT_ISSUE_Y(SO, J)                                                                                               \
    __builtin_amdgcn_global_load_lds((const void*)(gY + voffY[J]),                                                      \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(dstY + (SO) + (J) * 1024), 16, 0, 0)
#define NT_ISSUE_A(SO, J)                                                                                               \
    __builtin_amdgcn_global_load_lds((const void*)(gA2 + voffA[J]),                                                     \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(dstA + (SO) + (J) * 1024), 16, 0, 0)
#define NT_ISSUE_B(SO, J)                                                                                               \
    __builtin_amdgcn_global_load_lds((const void*)(gB2 + voffB[J]),                                                     \
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(dstB + (SO) + (J) * 1024), 16, 0, 0)
#define NT_ADV_AB() do { ++sx; const int st_ = (sx < T) ? 128 : 0; gA2 += st_; gB2 += st_; } while (0)
    // past the last stage the pointers stay on it (staged again into a buffer nobody reads any more)
#define NT_ADV_X() do { ++sx; gX += (sx < T) ? 128 : 0; } while (0)
#define NT_ADV_Y() do { ++sy; gY += (sy < T) ? 128 : 0; } while (0)
    // Fragment reads are inline asm on purpose (as in ar_gemm.hip): to hipcc an LDS-DMA in flight is a pending store to "some LDS" and
    // it would put s_waitcnt vmcnt(0) in front of the first LDS read it can see.  Ordering against the DMA is the counted vmcnt +
    // barrier protocol; the MFMAs wait for their operands with the explicit lgkmcnt(0) of the L part.
#define NT_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define NT_PIN() __builtin_amdgcn_sched_barrier(0)
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    u32x4_t fa[2][4], fb[2][2];
    u32x4_t fa16[8], fb16[4];
    unsigned long long tr_acc[4] = {0, 0, 0, 0}, tr_last = 0, tr_begin = 0, tr_wall = 0;
#define NT_TRACE(I)                                                                                                     \
    do {                                                                                                                \
        if (TRACE) {                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            const unsigned long long now_ = __builtin_readcyclecounter();                                               \
            tr_acc[I] += now_ - tr_last;                                                                                \
            tr_last = now_;                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        }                                                                                                               \
    } while (0)
#define NT_MMA(U, MI, NI)                                                                                               \
    acc[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nt_bf16x8_t, fb[U][NI]),                   \
                                                          __builtin_bit_cast(nt_bf16x8_t, fa[U][MI]), acc[MI][NI], 0, 0, 0)
#define NT_MMA16(MI, NI)                                                                                                \
    acc16[MI][NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(nt_bf16x8_t, fb16[NI]),                  \
                                                            __builtin_bit_cast(nt_bf16x8_t, fa16[MI]), acc16[MI][NI], 0, 0, 0)
#define NT_MMA16R(MI) NT_MMA16(MI, 0); NT_MMA16(MI, 1); NT_MMA16(MI, 2); NT_MMA16(MI, 3)
#define NT_PHASE16(S)                                                                                                   \
    do {                                                                                                                \
        constexpr int SG = (S) >> 1, AH = (S) & 1, O = SG * 16384;                                                      \
        constexpr int OI = AH ? O : (16384 - O);                                                                        \
        NT_RD(fb16[0], adB16[AH], O);         NT_RD(fb16[1], adB16[AH], O + 2048);                                      \
        NT_RD(fb16[2], adB16[AH], O + 4096);  NT_RD(fb16[3], adB16[AH], O + 6144);                                      \
        NT_RD(fa16[0], adA16[AH], O);         NT_RD(fa16[1], adA16[AH], O + 2048);                                      \
        NT_RD(fa16[2], adA16[AH], O + 4096);  NT_RD(fa16[3], adA16[AH], O + 6144);                                      \
        NT_RD(fa16[4], adA16[AH], O + 8192);  NT_RD(fa16[5], adA16[AH], O + 10240);                                     \
        NT_RD(fa16[6], adA16[AH], O + 12288); NT_RD(fa16[7], adA16[AH], O + 14336);                                     \
        if (AH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        NT_TRACE(0);                                                                                                    \
        bar();                                                                                                          \
        NT_TRACE(1);                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        NT_MMA16R(0); NT_PIN(); if (AH) NT_ISSUE_X(OI, 0); else NT_ISSUE_Y(OI, 0); NT_PIN();                            \
        NT_MMA16R(1); NT_MMA16R(2); NT_PIN(); if (AH) NT_ISSUE_X(OI, 1); else NT_ISSUE_Y(OI, 1); NT_PIN();              \
        NT_MMA16R(3); NT_MMA16R(4); NT_PIN(); if (AH) NT_ISSUE_X(OI, 2); else NT_ISSUE_Y(OI, 2); NT_PIN();              \
        NT_MMA16R(5); NT_MMA16R(6); NT_PIN();                                                                           \
        if (AH) { NT_ISSUE_X(OI, 3); NT_ADV_X(); } else { NT_ISSUE_Y(OI, 3); NT_ADV_Y(); } NT_PIN();                    \
        NT_MMA16R(7);                                                                                                   \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        if (AH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                        \
        NT_TRACE(2);                                                                                                    \
        bar();                                                                                                          \
        NT_TRACE(3);                                                                                                    \
    } while (0)
    // phase S of the loop body (0..3): stage parity SG = S >> 1, K half AH = S & 1
#define NT_PHASE(S)                                                                                                     \
    do {                                                                                                                \
        constexpr int SG = (S) >> 1, AH = (S) & 1, O = SG * 16384;                                                      \
        constexpr int OI = AH ? O : (16384 - O);      /* odd phase: stage t+2 -> this parity; even: stage t+1 -> the other */ \
        NT_RD(fb[0][0], adB[AH][0], O);        NT_RD(fb[0][1], adB[AH][0], O + 4096);                                   \
        NT_RD(fa[0][0], adA[AH][0], O);        NT_RD(fa[0][1], adA[AH][0], O + 4096);                                   \
        NT_RD(fa[0][2], adA[AH][0], O + 8192); NT_RD(fa[0][3], adA[AH][0], O + 12288);                                  \
        NT_RD(fb[1][0], adB[AH][1], O);        NT_RD(fb[1][1], adB[AH][1], O + 4096);                                   \
        NT_RD(fa[1][0], adA[AH][1], O);        NT_RD(fa[1][1], adA[AH][1], O + 4096);                                   \
        NT_RD(fa[1][2], adA[AH][1], O + 8192); NT_RD(fa[1][3], adA[AH][1], O + 12288);                                  \
        if (AH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        if (DMAL && !AH) {      /* stage tau + 1 into the other parity: the wave's B pieces first, then its A pieces */         \
            NT_PIN();                                                                                                   \
            NT_ISSUE_B(16384 - O, 0); NT_ISSUE_B(16384 - O, 1); NT_ISSUE_B(16384 - O, 2); NT_ISSUE_B(16384 - O, 3);       \
            NT_ISSUE_A(16384 - O, 0); NT_ISSUE_A(16384 - O, 1); NT_ISSUE_A(16384 - O, 2); NT_ISSUE_A(16384 - O, 3);       \
            NT_ADV_AB();                                                                                                \
        }                                                                                                               \
        NT_TRACE(0);                                                                                                    \
        bar();                                                                                                          \
        NT_TRACE(1);                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        NT_MMA(0, 0, 0); NT_MMA(0, 0, 1); NT_PIN(); if (!DMAL) { if (AH) NT_ISSUE_X(OI, 0); else NT_ISSUE_Y(OI, 0); } NT_PIN(); \
        NT_MMA(0, 1, 0); NT_MMA(0, 1, 1); NT_MMA(0, 2, 0); NT_MMA(0, 2, 1); NT_PIN();                                   \
        if (!DMAL) { if (AH) NT_ISSUE_X(OI, 1); else NT_ISSUE_Y(OI, 1); } NT_PIN();                                      \
        NT_MMA(0, 3, 0); NT_MMA(0, 3, 1); NT_MMA(1, 0, 0); NT_MMA(1, 0, 1); NT_PIN();                                   \
        if (!DMAL) { if (AH) NT_ISSUE_X(OI, 2); else NT_ISSUE_Y(OI, 2); } NT_PIN();                                      \
        NT_MMA(1, 1, 0); NT_MMA(1, 1, 1); NT_MMA(1, 2, 0); NT_MMA(1, 2, 1); NT_PIN();                                   \
        if (!DMAL) { if (AH) { NT_ISSUE_X(OI, 3); NT_ADV_X(); } else { NT_ISSUE_Y(OI, 3); NT_ADV_Y(); } } NT_PIN();        \
        NT_MMA(1, 3, 0); NT_MMA(1, 3, 1);                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        if (AH) { if (DMAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
        NT_TRACE(2);                                                                                                    \
        bar();                                                                                                          \
        NT_TRACE(3);                                                                                                    \
    } while (0)

    if (DMAL) {
        // ---- prologue: stage 0 (the first even phase issues stage 1)
        NT_ISSUE_B(0, 0); NT_ISSUE_B(0, 1); NT_ISSUE_B(0, 2); NT_ISSUE_B(0, 3);
        NT_ISSUE_A(0, 0); NT_ISSUE_A(0, 1); NT_ISSUE_A(0, 2); NT_ISSUE_A(0, 3);
        NT_ADV_AB();
        nt_wait_vm<0>();
    } else {
        // ---- prologue: X(0), Y(0) into parity 0, X(1) into parity 1; stage 0 landed for everybody (Y(1) is the first even phase's issue)
        NT_ISSUE_X(0, 0); NT_ISSUE_X(0, 1); NT_ISSUE_X(0, 2); NT_ISSUE_X(0, 3); NT_ADV_X();
        NT_ISSUE_Y(0, 0); NT_ISSUE_Y(0, 1); NT_ISSUE_Y(0, 2); NT_ISSUE_Y(0, 3); NT_ADV_Y();
        NT_ISSUE_X(16384, 0); NT_ISSUE_X(16384, 1); NT_ISSUE_X(16384, 2); NT_ISSUE_X(16384, 3); NT_ADV_X();
        nt_wait_vm<4>();
    }
    bar();
    if (wr == 1) bar();
    if (TRACE) { tr_wall = __builtin_amdgcn_s_memrealtime(); tr_begin = tr_last = __builtin_readcyclecounter(); }
    for (int t = 0; t < T; t += 2) {
        if constexpr (M16) {
            NT_PHASE16(0);
            NT_PHASE16(1);
            NT_PHASE16(2);
            NT_PHASE16(3);
        } else {
            NT_PHASE(0);
            NT_PHASE(1);
            NT_PHASE(2);
            NT_PHASE(3);
        }
    }
    if (TRACE && a.trace && lane == 0) {
        unsigned long long* o = a.trace + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = tr_acc[0]; o[1] = tr_acc[1]; o[2] = tr_acc[2]; o[3] = tr_acc[3];
        o[4] = tr_last - tr_begin; o[5] = (unsigned long long)(2 * T);
        o[6] = __builtin_amdgcn_s_memrealtime() - tr_wall;        // the same loop on the constant 100 MHz counter: cycles / this = the shader clock
        o[7] = 0;
    }
    if (wr == 0) bar();
    nt_wait_vm<0>();          // no LDS-DMA may outlive the workgroup's LDS allocation
#undef NT_PHASE
#undef NT_PHASE16
#undef NT_MMA16R
#undef NT_MMA16
#undef NT_MMA
#undef NT_RD
#undef NT_PIN
#undef NT_ISSUE_X
#undef NT_ISSUE_Y
#undef NT_ISSUE_A
#undef NT_ISSUE_B
#undef NT_ADV_AB
#undef NT_TRACE
#undef NT_ADV_X
#undef NT_ADV_Y

    if constexpr (M16) {      // lane owns row m = m0 + wr*128 + mi*16 + lane % 16; n = n0 + wc*64 + ni*16 + 4 (lane / 16) + r
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int ml = wr * 128 + mi * 16 + (lane & 15);
            if (ml < rows_valid) {
                uint16_t* rowp = a.C + (m0 + ml) * a.ldc + n0 + wc * 64 + 4 * (lane >> 4);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    uint2 o;
                    o.x = pack_bf16x2(acc16[mi][ni][0], acc16[mi][ni][1]);
                    o.y = pack_bf16x2(acc16[mi][ni][2], acc16[mi][ni][3]);
                    *reinterpret_cast<uint2*>(rowp + ni * 16) = o;
                }
            }
        }
        return;
    }
    // ---- epilogue: lane owns row m = m0 + wr*128 + mi*32 + (lane & 31); n = n0 + wc*64 + ni*32 + 8*t + 4*(lane >> 5) + r
    const int h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < (M16 ? 1 : 4); ++mi) {
        const int ml = wr * 128 + mi * 32 + (lane & 31);
        if (ml < rows_valid) {
            uint16_t* rowp = a.C + (m0 + ml) * a.ldc + n0 + wc * 64 + 4 * h;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint2 o;
                    o.x = pack_bf16x2(acc[mi][ni][4 * t + 0], acc[mi][ni][4 * t + 1]);
                    o.y = pack_bf16x2(acc[mi][ni][4 * t + 2], acc[mi][ni][4 * t + 3]);
                    *reinterpret_cast<uint2*>(rowp + ni * 32 + 8 * t) = o;
                }
        }
    }
}


// ---- nt2: one wave per SIMD, 128 x 128 per wave (round 5) -------------------------------------------------------------------
// The per-phase cycle table of k_gemm_nt (profiles/r05_gemm_nt_phase_cycles.json) says the eight-wave form issues MFMAs 85 % of its
// cycles and the chip clocks it at ~1.55 GHz: it is bound by POWER, not by issue slots, and the library's forward kernel (1.5-1.6 PF)
// must be spending less energy per MFMA.  The obvious candidate is operand traffic: a 128 x 64 wave tile reads 6 KB of LDS per 8 MFMAs,
// a 128 x 128 tile 8 KB per 16.  nt2 is that shape on the same LDS image, DMA mapping and swizzle as k_gemm_nt:
//   * 4 waves (2 x 2), acc[4][4] = 256 accumulator registers, one wave per SIMD (512-register budget);
//   * no wave has a partner to hide its fragment reads: the reads of k16 unit u+1 and the LDS-DMA pieces are interleaved BETWEEN the
//     16 MFMAs of unit u (two fragment register sets; an in-order wave hides ~5 single-issue instructions per 32-cycle MFMA);
//   * one barrier per 64-deep stage, placed BEFORE the last unit's MFMAs: at that point every wave has waited for its own pieces of
//     stage t+1 (vmcnt(0): nothing of stage t+2 has been issued yet) and has finished reading stage t (its unit-3 fragments are in
//     registers), so behind the barrier stage t+1 may be read and stage t's buffer may be refilled with stage t+2.  Pieces of stage
//     t+2 are issued in unit 3 of stage t (6) and units 0 / 1 of stage t+1 (5 + 5); unit 2 is their landing slack.
template <bool GROUPED>
__global__ __launch_bounds__(256, 1) void k_gemm_nt2(NtArgs a) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    int64_t m0;
    int n0, rows_valid;
    const uint16_t* Bp = a.B;
    if (GROUPED) {
        const int nwg = gridDim.x;
        int id = blockIdx.x;
        if ((nwg & 7) == 0) id = (id & 7) * (nwg >> 3) + (id >> 3);
        const int bw = (a.tiles_n % 8 == 0) ? 8 : a.tiles_n;
        const int per_band = a.tiles_m * bw;
        const int band = id / per_band, rem = id - band * per_band;
        const int gmt = rem / bw, tn = band * bw + (rem - gmt * bw);
        if (tn >= a.tiles_n) return;
        int e = -1, base = 0, r0 = 0, r1 = 0;
        for (int i = 0; i < a.E; ++i) {
            const int s0 = a.row_off[i], s1 = a.row_off[i + 1];
            const int mt = (s1 - s0 + NT_B - 1) / NT_B;
            if (e < 0 && gmt < base + mt) { e = i; r0 = s0 + (gmt - base) * NT_B; r1 = s1; }
            base += mt;
        }
        if (e < 0) return;
        m0 = r0;
        rows_valid = (r1 - r0) < NT_B ? (r1 - r0) : NT_B;
        n0 = tn * NT_B;
        Bp = a.B + a.b_off[e];
    } else {
        const int nwg = a.tiles_m * a.tiles_n;
        const int bid = blockIdx.x;
        int tm, tn;
        if (a.order == 2 && (a.tiles_m % 4 == 0) && (a.tiles_n % 8 == 0) && (nwg % 256 == 0)) {
            const int x = bid & 7, s = bid >> 3, pi = s >> 5, w = s & 31;
            const int P = pi * 8 + x, pn = a.tiles_n >> 3;
            tm = (P / pn) * 4 + (w >> 3);
            tn = (P % pn) * 8 + (w & 7);
        } else {
            int id = bid;
            if (a.order >= 1 && nwg % 8 == 0) id = (bid & 7) * (nwg >> 3) + (bid >> 3);
            tm = id / a.tiles_n;
            tn = id % a.tiles_n;
        }
        m0 = (int64_t)tm * NT_B;
        n0 = tn * NT_B;
        const int64_t left = (int64_t)a.M - m0;
        rows_valid = left < NT_B ? (int)left : NT_B;
    }
    const int T = a.K / NT_K;

    // ---- DMA: wave w stages rows [64 w, 64 w + 64) of the A tile and of the B tile: 8 pieces each, piece j = rows 64 w + 8 j + (lane >> 3)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    uint32_t voffA[8], voffB[8];
    {
        const int pc = lane & 7;
        const int64_t last = (int64_t)a.M - 1 - m0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lc = pc ^ ((4 * j + (lane >> 4)) & 7);
            const int row = wave * 64 + 8 * j + (lane >> 3);
            int64_t ra = row;
            if (ra > last) ra = last;
            voffA[j] = (uint32_t)(ra * a.lda * 2 + lc * 16);
            voffB[j] = (uint32_t)((int64_t)row * a.ldb * 2 + lc * 16);
        }
    }
    const uint8_t* gA = reinterpret_cast<const uint8_t*>(a.A + m0 * a.lda);
    const uint8_t* gB = reinterpret_cast<const uint8_t*>(Bp + (int64_t)n0 * a.ldb);
    // rows 64 w .. 64 w + 63 live in half w >> 1 at within-half row (w & 1) * 64
    const uint32_t dstA = lds0 + (wave >> 1) * 32768 + (wave & 1) * 64 * 128;
    const uint32_t dstB = dstA + 65536;
    int sdma = 0;                                    // stage the DMA pointers stand on

    // ---- fragment read addresses: unit u of a stage = logical chunks 2u, 2u + 1
    uint32_t adA[4], adB[4];
    {
        const int l31 = lane & 31, h = lane >> 5, s = (l31 >> 1) & 7;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t X = l31 * 128 + 16 * ((2 * u + h) ^ s);
            adA[u] = lds0 + wr * 32768 + X;
            adB[u] = lds0 + 65536 + wc * 32768 + X;
        }
    }

    nt_f32x16_t acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // piece P of the stage the pointers stand on (0..7: A pieces, 8..15: B pieces) into the buffer at byte offset SO (0 / 16384)
#define N2_ISSUE(SO, P)                                                                                                 \
    do {                                                                                                                \
        if ((P) < 8)                                                                                                    \
            __builtin_amdgcn_global_load_lds((const void*)(gA + voffA[(P) & 7]),                                        \
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(dstA + (SO) + ((P) & 7) * 1024), 16, 0, 0); \
        else                                                                                                            \
            __builtin_amdgcn_global_load_lds((const void*)(gB + voffB[(P) & 7]),                                        \
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(dstB + (SO) + ((P) & 7) * 1024), 16, 0, 0); \
    } while (0)
#define N2_ADV() do { ++sdma; const int st_ = (sdma < T) ? 128 : 0; gA += st_; gB += st_; } while (0)
#define N2_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define N2_PIN() __builtin_amdgcn_sched_barrier(0)
    u32x4_t fa[2][4], fb[2][4];
#define N2_MMA(FB, MI, NI)                                                                                              \
    acc[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nt_bf16x8_t, fb[FB][NI]),                  \
                                                          __builtin_bit_cast(nt_bf16x8_t, fa[FB][MI]), acc[MI][NI], 0, 0, 0)
    // the 8 fragment reads of unit RU (0..3) of the stage in the buffer at RO into register set NB, one per slot i = 0..7
#define N2_READ(NB, RU, RO, I)                                                                                          \
    do {                                                                                                                \
        if ((I) < 4) N2_RD(fa[NB][(I) & 3], adA[RU], (RO) + ((I) & 3) * 4096);                                          \
        else N2_RD(fb[NB][(I) & 3], adB[RU], (RO) + ((I) & 3) * 4096);                                                  \
    } while (0)
    // one k16 unit: 16 MFMAs from register set FB; between them the 8 reads of the next unit (into FB ^ 1) and up to 6 DMA pieces
    // (pieces P0 .. P0 + NP - 1 into the buffer at DO); every slot is pinned: MFMA, then at most one read or one piece
#define N2_UNIT(FB, RU, RO, DO, P0, NP)                                                                                 \
    do {                                                                                                                \
        N2_MMA(FB, 0, 0); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 0); N2_PIN();                                             \
        N2_MMA(FB, 0, 1); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 4); N2_PIN();                                             \
        N2_MMA(FB, 0, 2); N2_PIN(); if ((NP) > 0) N2_ISSUE(DO, (P0) + 0); N2_PIN();                                     \
        N2_MMA(FB, 0, 3); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 1); N2_PIN();                                             \
        N2_MMA(FB, 1, 0); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 5); N2_PIN();                                             \
        N2_MMA(FB, 1, 1); N2_PIN(); if ((NP) > 1) N2_ISSUE(DO, (P0) + 1); N2_PIN();                                     \
        N2_MMA(FB, 1, 2); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 2); N2_PIN();                                             \
        N2_MMA(FB, 1, 3); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 6); N2_PIN();                                             \
        N2_MMA(FB, 2, 0); N2_PIN(); if ((NP) > 2) N2_ISSUE(DO, (P0) + 2); N2_PIN();                                     \
        N2_MMA(FB, 2, 1); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 3); N2_PIN();                                             \
        N2_MMA(FB, 2, 2); N2_PIN(); N2_READ((FB) ^ 1, RU, RO, 7); N2_PIN();                                             \
        N2_MMA(FB, 2, 3); N2_PIN(); if ((NP) > 3) N2_ISSUE(DO, (P0) + 3); N2_PIN();                                     \
        N2_MMA(FB, 3, 0); N2_PIN(); if ((NP) > 4) N2_ISSUE(DO, (P0) + 4); N2_PIN();                                     \
        N2_MMA(FB, 3, 1); N2_PIN(); if ((NP) > 5) N2_ISSUE(DO, (P0) + 5); N2_PIN();                                     \
        N2_MMA(FB, 3, 2); N2_MMA(FB, 3, 3); N2_PIN();                                                                   \
    } while (0)
    // one stage in the buffer at O (0 / 16384); OO = the other buffer
#define N2_STAGE(O, OO)                                                                                                 \
    do {                                                                                                                \
        /* unit 0: reads unit 1; pieces 6..10 of stage t+1 (pointers stand on it) into the other buffer */              \
        N2_UNIT(0, 1, O, OO, 6, 5);                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); N2_PIN();                                                    \
        N2_UNIT(1, 2, O, OO, 11, 5); N2_ADV();                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); N2_PIN();                                                    \
        N2_UNIT(0, 3, O, OO, 0, 0);                                                                                     \
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                        \
        N2_PIN(); __builtin_amdgcn_s_barrier(); N2_PIN();                                                               \
        /* unit 3: reads unit 0 of stage t+1 from the other buffer; pieces 0..5 of stage t+2 into THIS buffer */        \
        N2_UNIT(1, 0, OO, O, 0, 6);                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); N2_PIN();                                                    \
    } while (0)

    // ---- prologue: stage 0 whole, stage 1's pieces 0..5 (what "unit 3 of stage -1" would have issued)
#pragma unroll
    for (int p = 0; p < 16; ++p) N2_ISSUE(0, p);
    N2_ADV();
#pragma unroll
    for (int p = 0; p < 6; ++p) N2_ISSUE(16384, p);
    nt_wait_vm<6>();
    N2_PIN(); __builtin_amdgcn_s_barrier(); N2_PIN();
    N2_READ(0, 0, 0, 0); N2_READ(0, 0, 0, 1); N2_READ(0, 0, 0, 2); N2_READ(0, 0, 0, 3);
    N2_READ(0, 0, 0, 4); N2_READ(0, 0, 0, 5); N2_READ(0, 0, 0, 6); N2_READ(0, 0, 0, 7);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); N2_PIN();
    for (int t = 0; t < T; t += 2) {
        N2_STAGE(0, 16384);
        N2_STAGE(16384, 0);
    }
    nt_wait_vm<0>();
#undef N2_STAGE
#undef N2_UNIT
#undef N2_READ
#undef N2_MMA
#undef N2_PIN
#undef N2_RD
#undef N2_ADV
#undef N2_ISSUE

    const int h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int ml = wr * 128 + mi * 32 + (lane & 31);
        if (ml < rows_valid) {
            uint16_t* rowp = a.C + (m0 + ml) * a.ldc + n0 + wc * 128 + 4 * h;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint2 o;
                    o.x = pack_bf16x2(acc[mi][ni][4 * t + 0], acc[mi][ni][4 * t + 1]);
                    o.y = pack_bf16x2(acc[mi][ni][4 * t + 2], acc[mi][ni][4 * t + 3]);
                    *reinterpret_cast<uint2*>(rowp + ni * 32 + 8 * t) = o;
                }
        }
    }
}

}  // namespace ar

using namespace ar;

static int g_nt_dmal = 3;        // 3: the 16x16x32 form (default since round 5); 0: 32x32x16; 1: 32x32x16 with the DMA pieces at the end of the L part; 2: nt2
// experiment knob (binding hygiene, tools/gpu/r05_gemm_nt_probe.py): variant 0 / 1 selects where the LDS-DMA pieces are issued; -1 keeps.
extern "C" int ar_gemm_nt_config(int variant) {
    if (variant >= 0 && variant <= 3) g_nt_dmal = variant;        // 2: nt2 (one wave per SIMD, 128 x 128 per wave); 3: 16x16x32
    return g_nt_dmal;
}

typedef void (*nt_fn)(NtArgs);
template <bool GROUPED>
static nt_fn nt_kernel() {
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)k_gemm_nt<GROUPED, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_nt<GROUPED, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_nt2<GROUPED>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_nt<GROUPED, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
    }
    if (g_nt_dmal == 3) return k_gemm_nt<GROUPED, false, false, true>;
    if (g_nt_dmal == 2) return k_gemm_nt2<GROUPED>;
    return g_nt_dmal ? k_gemm_nt<GROUPED, true> : k_gemm_nt<GROUPED, false>;
}
static int nt_threads() { return g_nt_dmal == 2 ? 256 : NT_THREADS; }

static int nt_check(const void* A, const void* B, void* C, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc) {
    if (N % NT_B || K < 128 || K % 128 || (lda % 8) || (ldb % 8) || (ldc % 4)) return AR_ERR_UNSUPPORTED;
    if ((((uintptr_t)A | (uintptr_t)B) & 15) || ((uintptr_t)C & 7)) return AR_ERR_UNSUPPORTED;
    // per-lane byte offsets inside a tile are 32-bit: 256 rows x the leading dimension
    if (lda * 2 * 256 >= (int64_t)1 << 32 || ldb * 2 * 256 >= (int64_t)1 << 32) return AR_ERR_UNSUPPORTED;
    return AR_OK;
}

// C[M, N] = A[M, K] B[N, K]^T.  M any, N % 256 == 0, K % 128 == 0.
extern "C" int ar_gemm_nt(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                          ar_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return AR_OK;
    const int rc = nt_check(A, B, C, N, K, lda, ldb, ldc);
    if (rc != AR_OK) return rc;
    NtArgs a;
    a.A = (const uint16_t*)A; a.B = (const uint16_t*)B; a.C = (uint16_t*)C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.tiles_m = (int)((M + NT_B - 1) / NT_B); a.tiles_n = (int)(N / NT_B); a.order = 2;
    a.row_off = nullptr; a.b_off = nullptr; a.E = 0; a.trace = nullptr;
    AR_LAUNCH_PROF(AR_PROF_GEMM_NT, M * N, (nt_kernel<false>()), a.tiles_m * a.tiles_n, nt_threads(), NT_LDS, (hipStream_t)stream, a);
    return launch_status();
}

// Grouped form: group e owns rows [row_off[e], row_off[e+1]) of A and of C and multiplies them with its own matrix B + b_off[e]
// ([N, K], ldb).  row_off ([n_groups + 1] int32) and b_off ([n_groups] int64, in elements) live on the DEVICE: the launch needs no
// host knowledge of the row counts -- the grid covers the worst case, M / 256 + n_groups row tiles, and workgroups past the last
// real tile exit at once.  M = row_off[n_groups] = rows of A.  Deterministic (no atomics, fixed summation order).
extern "C" int ar_gemm_nt_grouped(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                  int64_t ldc, const int32_t* row_off, const int64_t* b_off, int n_groups, ar_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || n_groups <= 0) return AR_OK;
    if (!row_off || !b_off || n_groups > 4096) return AR_ERR_UNSUPPORTED;
    const int rc = nt_check(A, B, C, N, K, lda, ldb, ldc);
    if (rc != AR_OK) return rc;
    NtArgs a;
    a.A = (const uint16_t*)A; a.B = (const uint16_t*)B; a.C = (uint16_t*)C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.tiles_n = (int)(N / NT_B); a.order = 1;
    a.tiles_m = (int)(M / NT_B) + n_groups;             // worst case: every group ends with a partial tile
    a.row_off = row_off; a.b_off = b_off; a.E = n_groups; a.trace = nullptr;
    int grid = a.tiles_m * a.tiles_n;
    grid = (grid + 7) / 8 * 8;                          // (a multiple of 8: the per-XCD runs of the tile order)
    AR_LAUNCH_PROF(AR_PROF_GEMM_NT, M * N, (nt_kernel<true>()), grid, nt_threads(), NT_LDS, (hipStream_t)stream, a);
    return launch_status();
}

// The dense kernel with s_memtime bookkeeping per phase (tools/gpu/r05_gemm_nt_trace.py -> DESIGN.md's per-phase cycle table): trace gets
// [tiles][8 waves][8] uint64 -- cycles in the fragment-read part, parked at the barrier after it, in the MFMA part, parked at the barrier
// after it, the whole K loop, the number of phases, the whole K loop in ticks of the constant 100 MHz counter, 0.  variant as ar_gemm_nt_config.  Costs ~10 % of the kernel's speed; C is still written.
extern "C" int ar_gemm_nt_trace(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                                unsigned long long* trace, int variant, ar_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !trace) return AR_ERR_UNSUPPORTED;
    const int rc = nt_check(A, B, C, N, K, lda, ldb, ldc);
    if (rc != AR_OK) return rc;
    NtArgs a;
    a.A = (const uint16_t*)A; a.B = (const uint16_t*)B; a.C = (uint16_t*)C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.tiles_m = (int)((M + NT_B - 1) / NT_B); a.tiles_n = (int)(N / NT_B); a.order = 2;
    a.row_off = nullptr; a.b_off = nullptr; a.E = 0; a.trace = trace;
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)k_gemm_nt<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_nt<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_nt<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS);
    }
    if (variant == 3) hipLaunchKernelGGL((k_gemm_nt<false, false, true, true>), a.tiles_m * a.tiles_n, NT_THREADS, NT_LDS, (hipStream_t)stream, a);
    else if (variant) hipLaunchKernelGGL((k_gemm_nt<false, true, true>), a.tiles_m * a.tiles_n, NT_THREADS, NT_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_gemm_nt<false, false, true>), a.tiles_m * a.tiles_n, NT_THREADS, NT_LDS, (hipStream_t)stream, a);
    return launch_status();
}
