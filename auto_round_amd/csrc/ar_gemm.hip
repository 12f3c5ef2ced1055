// ar_gemm.hip -- hand-written MFMA weight-gradient GEMM for gfx950:  dW[M,N] = dY^T X  with dY [K,M], X [K,N] row-major bf16.
//
// replaces: the autograd backward of F.linear(x, weight_q) w.r.t. weight_q inside WrapperLinear.forward
//           (auto_round/wrapper.py:528-556) -- torch.mm(dY.t(), X), a "TN" GEMM whose two operands are both strided in the
//           reduction dimension (tokens).  hipBLASLt runs it at ~1.0 PFLOP/s on MI355X (MT256x256x32, 47 % MFMA utilisation);
//           it is 27 % of a Llama-3-8B tuning iteration (profiles/archive/r01_llama8b_block_kernel_stats.csv).
//
// Design (CDNA4-first):
//   * 256x256 output tile per 512-thread workgroup (8 waves = 2 per SIMD), wave tile 64(m) x 128(n), v_mfma_f32_32x32x16_bf16
//     (k_gemm_dw4, rounds 2-4) or v_mfma_f32_16x16x32_bf16 (k_gemm_dw6, round 5: the default -- identical bits, 6-10 % faster)
//     computing the TRANSPOSED tile (a-operand = X fragment, b-operand = dY fragment): every lane then owns, for ONE output
//     row m, runs of 4 consecutive n -- 8-byte stores, and a quant group (128 consecutive n of one row) lives in one lane
//     pair, which is what the fused backward + sign-SGD epilogue wants.
//   * both operands are K-strided in memory, so tiles are staged [k][256] exactly as they lie in HBM (each k-row is one
//     coalesced 512-byte piece) by LDS-DMA (global_load_lds, 16 B per lane, no staging VGPRs) and the MFMA fragments are read
//     with ds_read_b64_tr_b16 -- the hardware 4x4 transposing LDS read -- instead of transposing in registers.
//   * the LDS image is XOR-swizzled on the 16-byte chunk index, chunk ^= (k & 3) << 2 (applied to the per-lane SOURCE address of
//     the DMA and to the read address; the DMA destination stays lane-linear), so the four k-rows a transposing read touches
//     fall on disjoint bank ranges.
//   * K is consumed in units of 16 (one MFMA K-step; 16 KB of LDS per unit for both operands) through a ring of 8 units with
//     6 units of DMA in flight; counted s_waitcnt vmcnt(8) -- never 0 -- and raw s_barrier keep the DMA in flight across
//     barriers.  The two waves of a SIMD run half a period apart (waves 4-7 take one extra barrier up front): while one is in
//     its 8-MFMA cluster the other issues its 12 fragment reads and 2 DMA pieces, so the matrix pipe always has a taker.
//   * XCD-aware tile order: each XCD (own L2) works on 2x8-tile patches, two patches of one column band at a time, so the 32
//     concurrent workgroups of an XCD share 8 X panels and 4 dY panels per K-unit instead of fetching 64.
#include "ar_common.hpp"

namespace ar {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int GB = 256;                       // block tile edge (M and N)
constexpr int GU = 16;                        // k per unit
constexpr int GR = 8;                         // ring size in units
constexpr int GD = 6;                         // units of DMA in flight
constexpr int ROWB = GB * 2;                  // bytes per staged k-row
constexpr int PIECE = GU * ROWB;              // 8192: one operand's piece of a unit
constexpr int UNIT = 2 * PIECE;               // 16384
constexpr int GEMM_LDS = GR * UNIT;           // 131072
constexpr int GTHREADS = 512;

struct GemmArgs {
    const uint16_t* Y;   // dY [K, M]  (ldy)
    const uint16_t* X;   // X  [K, N]  (ldx)
    uint16_t* W;         // dW [M, N]  (ldw)
    int M, N, K;
    int64_t ldy, ldx, ldw;
    int accumulate;      // dW += (torch addmm_ semantics: fp32 sum of the old bf16 value and the fp32 accumulator, rounded once)
    int tiles_m, tiles_n, order;
    float* ws;           // split-K: fp32 partial tiles [nsplit][M][N] (caller-owned scratch), reduced in a fixed order afterwards
    int nsplit;
    int tile0;           // > 0: split-K over the LAST tiles of the launch order only (tile0 = first of them); partial tiles are
                         // then stored compactly, [tile - tile0][nsplit][256][256]
    const int* kcut;     // ar_gemm_dw_sk: per row-major tile the k-row where its two parts meet (0 = one part); ws = [tile][256 * 256] fp32
    const int32_t* goff; // ar_gemm_dw_grouped: [groups + 1] k-row ranges of the groups (device memory)
    const int64_t* woff; // ar_gemm_dw_grouped: [groups] element offsets of the groups' outputs relative to W (device memory)
};

__device__ __forceinline__ void tile_of_block(const GemmArgs& a, int bid, int& tm, int& tn) {
    const int nwg = a.tiles_m * a.tiles_n;
    if (a.order == 2 && (a.tiles_m % 2 == 0) && (a.tiles_n % 8 == 0) && ((nwg / 16) % 8 == 0)) {
        const int x = bid & 7, s = bid >> 3, t = s >> 4, j = s & 15;
        const int pn = a.tiles_n >> 3;
        const int p = x + 8 * t;
        tm = (p / pn) * 2 + (j >> 3);
        tn = (p % pn) * 8 + (j & 7);
        return;
    }
    int id = bid;
    if (a.order >= 1 && nwg % 8 == 0) id = (bid & 7) * (nwg >> 3) + (bid >> 3);      // contiguous chunk per XCD
    tm = id / a.tiles_n;
    tn = id % a.tiles_n;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The earlier generations (v0, v1, the timing ablations of v1, v4) are kept for tools/gemm_dw_probe.py only and are compiled only
// with -DAR_GEMM_EXPERIMENTS (make EXTRA=-DAR_GEMM_EXPERIMENTS): the shipped library holds v3 alone, so nothing in a process can
// switch the tuner's weight-gradient GEMM to another kernel.
#ifdef AR_GEMM_EXPERIMENTS
// SEM selects which (k-row, 4-column piece) a lane of a 16-lane group hands to the transposing read: 1 = row i>>2, piece i&3 --
// the hardware's rule, pinned on the GPU by tools/mfma_probe (profiles/archive/r02_mfma_probe.json: within a 16-lane group, lane L
// receives element L%4 of the 8-byte pieces supplied by lanes L/4, L/4+4, L/4+8, L/4+12); 2 = row i&3, piece i>>2 (kept as the
// negative control of tools/gemm_dw_probe.py).
template <int SEM>
__global__ __launch_bounds__(GTHREADS, 2) void k_gemm_dw(GemmArgs a) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3, wn = wave >> 2;
    int tm, tn;
    tile_of_block(a, blockIdx.x, tm, tn);
    const int64_t m0 = (int64_t)tm * GB, n0 = (int64_t)tn * GB;
    const int U = a.K / GU;

    // ---- DMA source pointers: this wave moves k-rows {2*wave, 2*wave+1} of each operand's piece; lane -> (row, physical chunk)
    const int drow = 2 * wave + (lane >> 5);
    const int pchunk = lane & 31;
    const int lchunk = pchunk ^ ((drow & 3) << 2);
    const uint16_t* srcP = a.X + (int64_t)drow * a.ldx + n0 + lchunk * 8;
    const uint16_t* srcQ = a.Y + (int64_t)drow * a.ldy + m0 + lchunk * 8;
    const int64_t stepP = (int64_t)GU * a.ldx, stepQ = (int64_t)GU * a.ldy;
    const int dstoff = 2 * wave * ROWB;       // wave-uniform byte offset inside a piece

    // ---- fragment read offsets (bytes inside a unit)
    const int q = lane >> 4, i = lane & 15, g = q >> 1;
    const int rowsel = (SEM == 2) ? (i & 3) : (i >> 2);
    const int piece = (SEM == 2) ? (i >> 2) : (i & 3);
    const int rowoff = (8 * g + rowsel) * ROWB + (piece & 1) * 8;
    int offP[4], offQ[2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int chunk = wn * 16 + ni * 4 + (q & 1) * 2 + (piece >> 1);
        offP[ni] = rowoff + ((chunk ^ (rowsel << 2)) << 4);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int chunk = wm * 8 + mi * 4 + (q & 1) * 2 + (piece >> 1);
        offQ[mi] = PIECE + rowoff + ((chunk ^ (rowsel << 2)) << 4);
    }

    f32x16_t acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto issue = [&](int v) {      // DMA of unit v (clamped: the last D issues of the loop re-stage the final unit, never read)
        const int vv = v < U ? v : U - 1;
        uint8_t* dst = lds + (v & (GR - 1)) * UNIT + dstoff;
        __builtin_amdgcn_global_load_lds((const void*)(srcP + vv * stepP), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(srcQ + vv * stepQ), (__attribute__((address_space(3))) void*)(dst + PIECE), 16, 0, 0);
    };
    // Fragment reads are issued as inline asm on purpose: to hipcc an LDS-DMA in flight is a pending store to "some LDS", so
    // it puts s_waitcnt vmcnt(0) in front of the first LDS read it can see after a global_load_lds -- which would drain the
    // 6-unit DMA pipeline every unit.  The asm reads are invisible to that pass; ordering against the DMA is provided by the
    // counted vmcnt + barrier protocol above, and the MFMAs wait for their operands with an explicit counted lgkmcnt (LDS
    // reads return in order: lgkmcnt(12) leaves exactly the 12 reads of the NEXT unit's prefetch outstanding).
    auto rd = [&](uint32_t addr, s16x4_t& lo, s16x4_t& hi) {
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                     : "=&v"(lo), "=&v"(hi)
                     : "v"(addr), "n"(4 * ROWB)
                     : "memory");
    };
    struct Frags { s16x4_t plo[4], phi[4], qlo[2], qhi[2]; };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    auto read_frags = [&](int v, Frags& f) {
        const uint32_t base = lds0 + (uint32_t)((v & (GR - 1)) * UNIT);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) rd(base + offQ[mi], f.qlo[mi], f.qhi[mi]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) rd(base + offP[ni], f.plo[ni], f.phi[ni]);
    };
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto cat = [](const s16x4_t& lo, const s16x4_t& hi) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](const Frags& f) {
        asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(f.plo[ni], f.phi[ni]), cat(f.qlo[mi], f.qhi[mi]), acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue
#pragma unroll
    for (int v = 0; v < GD; ++v) issue(v);
    wait_vm<(GD - 2) * 2>();
    bar();
    Frags fa, fb;
    read_frags(0, fa);
    if (grp == 1) bar();

    // ---- main loop: two units per trip (static register sets)
    for (int u = 0; u < U; u += 2) {
        read_frags(u + 1, fb);
        issue(u + GD);
        wait_vm<(GD - 2) * 2>();
        bar();
        mma(fa);
        bar();
        read_frags(u + 2, fa);
        issue(u + 1 + GD);
        wait_vm<(GD - 2) * 2>();
        bar();
        mma(fb);
        bar();
    }
    if (grp == 0) bar();
    wait_vm<0>();          // no LDS-DMA may outlive the workgroup's LDS allocation

    // ---- epilogue: lane owns row m = m0 + wm*64 + mi*32 + (lane&31); n = n0 + wn*128 + ni*32 + 8*t + 4*(lane>>5) + r
    const int h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + (lane & 31);
        uint16_t* rowp = a.W + m * a.ldw + n0 + wn * 128 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint16_t* p = rowp + ni * 32 + 8 * t;
                float v0 = acc[mi][ni][4 * t + 0], v1 = acc[mi][ni][4 * t + 1], v2 = acc[mi][ni][4 * t + 2], v3 = acc[mi][ni][4 * t + 3];
                if (a.accumulate) {
                    const uint2 old = *reinterpret_cast<const uint2*>(p);
                    v0 += bf16_lo(old.x); v1 += bf16_hi(old.x); v2 += bf16_lo(old.y); v3 += bf16_hi(old.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(p) = o;
            }
        }
    }
}


// ---- v1: 32-deep phases, immediate LDS offsets, running DMA pointers ----------------------------------------------------
// Same tile, staging and swizzle as k_gemm_dw.  Differences, all aimed at making a wave's non-MFMA work per phase shorter than
// its partner's MFMA cluster (in v0 ~50 issue slots of address arithmetic, reads and DMA stood beside only 8 MFMAs = 256 cycles):
//   * a phase is a PAIR of k16 units (32 KB of LDS): 24 fragment reads + 4 DMA pieces per wave against 16 MFMAs (512 cycles);
//     half the barriers per MFMA;
//   * the K loop is unrolled by the ring period (4 pairs), so every LDS address is base register + immediate (two base sets,
//     64 KB apart: the ds offset field is 16 bit) and every DMA destination is wave base + constant; the DMA source pointers
//     are running 64-bit registers advanced by one row-step per unit;
//   * fragments are read in the L part of their own phase and the wave waits for them (lgkmcnt(0)) BEFORE the barrier that ends
//     the L part -- it would only idle at that barrier anyway while the partner group finishes its MFMAs.  All reads of pair
//     p are therefore complete at the barrier that ends group 1's L(p), so the DMA of pair p+3 (same LDS slot as pair p-1) may be
//     issued in L(p): 3 pairs = 96 KB = ~2 periods of DMA in flight, counted s_waitcnt vmcnt(8) as before.
template <bool STAGGER>
__global__ __launch_bounds__(GTHREADS, 2) void k_gemm_dw2(GemmArgs a) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3, wn = wave >> 2;
    int tm, tn;
    tile_of_block(a, blockIdx.x, tm, tn);
    const int64_t m0 = (int64_t)tm * GB, n0 = (int64_t)tn * GB;
    const int U = a.K / GU;                        // k16 units; K % 128 == 0 (checked by the host)

    const int drow = 2 * wave + (lane >> 5);
    const int lchunk = (lane & 31) ^ ((drow & 3) << 2);
    const uint16_t* srcP = a.X + (int64_t)drow * a.ldx + n0 + lchunk * 8;
    const uint16_t* srcQ = a.Y + (int64_t)drow * a.ldy + m0 + lchunk * 8;
    const int64_t stepP = (int64_t)GU * a.ldx, stepQ = (int64_t)GU * a.ldy;     // elements per unit
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t dmabase = lds0 + 2 * wave * ROWB;                            // wave-uniform

    const int q = lane >> 4, i = lane & 15, g = q >> 1;
    const int rowsel = i >> 2, piece = i & 3;                                     // hardware rule of ds_read_b64_tr_b16
    const int rowoff = (8 * g + rowsel) * ROWB + (piece & 1) * 8;
    uint32_t aP[2][4], aQ[2][2];                                                  // [64 KB half][tile]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int chunk = wn * 16 + ni * 4 + (q & 1) * 2 + (piece >> 1);
        aP[0][ni] = lds0 + rowoff + ((chunk ^ (rowsel << 2)) << 4);
        aP[1][ni] = aP[0][ni] + 65536;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int chunk = wm * 8 + mi * 4 + (q & 1) * 2 + (piece >> 1);
        aQ[0][mi] = lds0 + PIECE + rowoff + ((chunk ^ (rowsel << 2)) << 4);
        aQ[1][mi] = aQ[0][mi] + 65536;
    }

    f32x16_t acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    int vnext = 0;                                 // next unit to stage
    auto issue_unit = [&](int slot_unit) {         // slot_unit: 0..7 compile-time after unrolling
        __builtin_amdgcn_global_load_lds((const void*)srcP, (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)srcQ, (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT + PIECE), 16, 0, 0);
        ++vnext;
        const bool more = vnext < U;               // past the end the pointers stay on the last unit (staged again, never read)
        srcP += more ? stepP : 0;
        srcQ += more ? stepQ : 0;
    };
    struct Pair { s16x4_t plo[2][4], phi[2][4], qlo[2][2], qhi[2][2]; };
#define AR_RD(LO, HI, ADDR, OFF)                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                           \
                 : "=&v"(LO), "=&v"(HI)                                                                                 \
                 : "v"(ADDR), "n"(OFF), "n"((OFF) + 4 * ROWB)                                                            \
                 : "memory")
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto cat = [](const s16x4_t& lo, const s16x4_t& hi) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    Pair f;
    // one phase on pair slot S (0..3): L part then M part
#define AR_PHASE(S)                                                                                                     \
    do {                                                                                                                \
        constexpr int H = (S) >> 1;                      /* which 64 KB half */                                          \
        constexpr int O = ((S) & 1) * 2 * UNIT;          /* offset inside the half */                                    \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                                 \
            if (k == 0) {                                                                                               \
                AR_RD(f.qlo[0][0], f.qhi[0][0], aQ[H][0], O); AR_RD(f.qlo[0][1], f.qhi[0][1], aQ[H][1], O);             \
                AR_RD(f.plo[0][0], f.phi[0][0], aP[H][0], O); AR_RD(f.plo[0][1], f.phi[0][1], aP[H][1], O);             \
                AR_RD(f.plo[0][2], f.phi[0][2], aP[H][2], O); AR_RD(f.plo[0][3], f.phi[0][3], aP[H][3], O);             \
            } else {                                                                                                    \
                AR_RD(f.qlo[1][0], f.qhi[1][0], aQ[H][0], O + UNIT); AR_RD(f.qlo[1][1], f.qhi[1][1], aQ[H][1], O + UNIT); \
                AR_RD(f.plo[1][0], f.phi[1][0], aP[H][0], O + UNIT); AR_RD(f.plo[1][1], f.phi[1][1], aP[H][1], O + UNIT); \
                AR_RD(f.plo[1][2], f.phi[1][2], aP[H][2], O + UNIT); AR_RD(f.plo[1][3], f.phi[1][3], aP[H][3], O + UNIT); \
            }                                                                                                           \
        }                                                                                                               \
        issue_unit((((S) + 3) & 3) * 2);                                                                                \
        issue_unit((((S) + 3) & 3) * 2 + 1);                                                                            \
        asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        bar();                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                                   \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                            \
                _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                        \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(f.plo[k][ni], f.phi[k][ni]),              \
                                                                          cat(f.qlo[k][mi], f.qhi[k][mi]), acc[mi][ni], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        bar();                                                                                                          \
    } while (0)

    // ---- prologue: pairs 0, 1, 2 in flight; pair 0 landed for everybody
#pragma unroll
    for (int v = 0; v < 6; ++v) issue_unit(v);
    wait_vm<8>();
    bar();
    if (STAGGER && grp == 1) bar();
    for (int u = 0; u < U; u += 8) {
        AR_PHASE(0);
        AR_PHASE(1);
        AR_PHASE(2);
        AR_PHASE(3);
    }
    if (STAGGER && grp == 0) bar();
    wait_vm<0>();
#undef AR_PHASE
#undef AR_RD

    const int h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + (lane & 31);
        uint16_t* rowp = a.W + m * a.ldw + n0 + wn * 128 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint16_t* p = rowp + ni * 32 + 8 * t;
                float v0 = acc[mi][ni][4 * t + 0], v1 = acc[mi][ni][4 * t + 1], v2 = acc[mi][ni][4 * t + 2], v3 = acc[mi][ni][4 * t + 3];
                if (a.accumulate) {
                    const uint2 old = *reinterpret_cast<const uint2*>(p);
                    v0 += bf16_lo(old.x); v1 += bf16_hi(old.x); v2 += bf16_lo(old.y); v3 += bf16_hi(old.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(p) = o;
            }
        }
    }
}


template <int ABL>
__global__ __launch_bounds__(GTHREADS, 2) void k_gemm_dw2_abl(GemmArgs a) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3, wn = wave >> 2;
    int tm, tn;
    tile_of_block(a, blockIdx.x, tm, tn);
    const int64_t m0 = (int64_t)tm * GB, n0 = (int64_t)tn * GB;
    const int U = a.K / GU;                        // k16 units; K % 128 == 0 (checked by the host)

    const int drow = 2 * wave + (lane >> 5);
    const int lchunk = (lane & 31) ^ ((drow & 3) << 2);
    const uint16_t* srcP = a.X + (int64_t)drow * a.ldx + n0 + lchunk * 8;
    const uint16_t* srcQ = a.Y + (int64_t)drow * a.ldy + m0 + lchunk * 8;
    const int64_t stepP = (int64_t)GU * a.ldx, stepQ = (int64_t)GU * a.ldy;     // elements per unit
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t dmabase = lds0 + 2 * wave * ROWB;                            // wave-uniform

    const int q = lane >> 4, i = lane & 15, g = q >> 1;
    const int rowsel = i >> 2, piece = i & 3;                                     // hardware rule of ds_read_b64_tr_b16
    const int rowoff = (8 * g + rowsel) * ROWB + (piece & 1) * 8;
    uint32_t aP[2][4], aQ[2][2];                                                  // [64 KB half][tile]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int chunk = wn * 16 + ni * 4 + (q & 1) * 2 + (piece >> 1);
        aP[0][ni] = lds0 + rowoff + ((chunk ^ (rowsel << 2)) << 4);
        aP[1][ni] = aP[0][ni] + 65536;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int chunk = wm * 8 + mi * 4 + (q & 1) * 2 + (piece >> 1);
        aQ[0][mi] = lds0 + PIECE + rowoff + ((chunk ^ (rowsel << 2)) << 4);
        aQ[1][mi] = aQ[0][mi] + 65536;
    }

    f32x16_t acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    int vnext = 0;                                 // next unit to stage
    auto issue_unit = [&](int slot_unit) {         // slot_unit: 0..7 compile-time after unrolling
        __builtin_amdgcn_global_load_lds((const void*)srcP, (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)srcQ, (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT + PIECE), 16, 0, 0);
        ++vnext;
        const bool more = vnext < U;               // past the end the pointers stay on the last unit (staged again, never read)
        srcP += more ? stepP : 0;
        srcQ += more ? stepQ : 0;
    };
    struct Pair { s16x4_t plo[2][4], phi[2][4], qlo[2][2], qhi[2][2]; };
#define AR_RD(LO, HI, ADDR, OFF)                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                           \
                 : "=&v"(LO), "=&v"(HI)                                                                                 \
                 : "v"(ADDR), "n"(OFF), "n"((OFF) + 4 * ROWB)                                                            \
                 : "memory")
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto cat = [](const s16x4_t& lo, const s16x4_t& hi) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    Pair f;
    // one phase on pair slot S (0..3): L part then M part
#define AR_PHASE(S)                                                                                                     \
    do {                                                                                                                \
        constexpr int H = (S) >> 1;                      /* which 64 KB half */                                          \
        constexpr int O = ((S) & 1) * 2 * UNIT;          /* offset inside the half */                                    \
        _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                                 \
            if (ABL >= 2) { if (u == 0 && (S) == 0) { AR_RD(f.qlo[k][0], f.qhi[k][0], aQ[0][0], 0); AR_RD(f.qlo[k][1], f.qhi[k][1], aQ[0][1], 0); \
                AR_RD(f.plo[k][0], f.phi[k][0], aP[0][0], 0); AR_RD(f.plo[k][1], f.phi[k][1], aP[0][1], 0);              \
                AR_RD(f.plo[k][2], f.phi[k][2], aP[0][2], 0); AR_RD(f.plo[k][3], f.phi[k][3], aP[0][3], 0); } }          \
            else if (k == 0) {                                                                                               \
                AR_RD(f.qlo[0][0], f.qhi[0][0], aQ[H][0], O); AR_RD(f.qlo[0][1], f.qhi[0][1], aQ[H][1], O);             \
                AR_RD(f.plo[0][0], f.phi[0][0], aP[H][0], O); AR_RD(f.plo[0][1], f.phi[0][1], aP[H][1], O);             \
                AR_RD(f.plo[0][2], f.phi[0][2], aP[H][2], O); AR_RD(f.plo[0][3], f.phi[0][3], aP[H][3], O);             \
            } else {                                                                                                    \
                AR_RD(f.qlo[1][0], f.qhi[1][0], aQ[H][0], O + UNIT); AR_RD(f.qlo[1][1], f.qhi[1][1], aQ[H][1], O + UNIT); \
                AR_RD(f.plo[1][0], f.phi[1][0], aP[H][0], O + UNIT); AR_RD(f.plo[1][1], f.phi[1][1], aP[H][1], O + UNIT); \
                AR_RD(f.plo[1][2], f.phi[1][2], aP[H][2], O + UNIT); AR_RD(f.plo[1][3], f.phi[1][3], aP[H][3], O + UNIT); \
            }                                                                                                           \
        }                                                                                                               \
        if (ABL == 0) { issue_unit((((S) + 3) & 3) * 2); issue_unit((((S) + 3) & 3) * 2 + 1);                           \
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        if (ABL < 3) bar();                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                                   \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                            \
                _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                        \
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(f.plo[k][ni], f.phi[k][ni]),              \
                                                                          cat(f.qlo[k][mi], f.qhi[k][mi]), acc[mi][ni], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        if (ABL < 3) bar();                                                                                             \
    } while (0)

    // ---- prologue: pairs 0, 1, 2 in flight; pair 0 landed for everybody
#pragma unroll
    for (int v = 0; v < 6; ++v) issue_unit(v);
    wait_vm<8>();
    bar();
    if ((ABL < 3) && grp == 1) bar();
    for (int u = 0; u < U; u += 8) {
        AR_PHASE(0);
        AR_PHASE(1);
        AR_PHASE(2);
        AR_PHASE(3);
    }
    if ((ABL < 3) && grp == 0) bar();
    wait_vm<0>();
#undef AR_PHASE
#undef AR_RD

    const int h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + (lane & 31);
        uint16_t* rowp = a.W + m * a.ldw + n0 + wn * 128 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint16_t* p = rowp + ni * 32 + 8 * t;
                float v0 = acc[mi][ni][4 * t + 0], v1 = acc[mi][ni][4 * t + 1], v2 = acc[mi][ni][4 * t + 2], v3 = acc[mi][ni][4 * t + 3];
                if (a.accumulate) {
                    const uint2 old = *reinterpret_cast<const uint2*>(p);
                    v0 += bf16_lo(old.x); v1 += bf16_hi(old.x); v2 += bf16_lo(old.y); v3 += bf16_hi(old.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(p) = o;
            }
        }
    }
}


#endif  // AR_GEMM_EXPERIMENTS

// (v2 -- v1 with half of the fragment reads moved into the MFMA cluster so that both waves of a SIMD feed the LDS pipe all the
//  time -- measured equal to v1 within noise on every shape, profiles/archive/r02_gemm_dw_v2_split_reads_no_gain.jsonl, and was removed.)

// ---- v3: v1 with the DMA pieces issued from inside the MFMA cluster (copy of the v1 setup) --------------------------------
// SPLITK: few output tiles but a deep K (OPT-125M's 768x768 weight against 16384 tokens is 9 tiles): the grid is tiles x nsplit,
// every workgroup reduces its own K slice (a whole number of 128-row chunks) into an fp32 partial tile in caller-owned scratch,
// and k_splitk_reduce sums the slices in slice order (deterministic: no float atomics) into the bf16 result.
// TAIL: K is not a multiple of 128 (an expert's share of the tokens in a MoE block): the last 128-row chunk is completed with
// zeros -- every lane of the LDS-DMA supplies its own source address, so rows past K simply read a 512-byte row of zeros.
__device__ __attribute__((aligned(512))) uint16_t g_zero_row[256];
// CUT (ar_gemm_dw_sk): a tile may be the sum of TWO accumulations, k-rows [0, cut) and [cut, K) -- the structure of a stream-K
// kernel that hands the tile's K range to two workgroups.  One workgroup still walks the whole K in order (same operand traffic
// and L2 sharing as the plain kernel): at the cut it parks its accumulators in the workspace (fp32, lane-major), restarts from zero,
// and adds the parked part back before the one rounding to bf16.
// GROUPED (ar_gemm_dw_grouped, round 5): the launch holds one full tile grid per group (an expert of a sparse-MoE block); a workgroup's
// K range is its group's row range [goff[g], goff[g+1]) read from device memory -- completed with zero rows like TAIL (which it implies)
// -- and its output is that group's own matrix W + woff[g].  A group without rows writes zeros.
// DMAL (round 5): the four LDS-DMA pieces of a phase are issued at the END of the L part -- after the wave's fragment reads have
// returned (lgkmcnt(0)), before the barrier -- instead of between the MFMAs of the M part.  Why: the two waves of a SIMD alternate
// (one in its 16-MFMA cluster, the other in its L part), every barrier interval is as long as the LONGER of the two, and a DMA piece
// issued among the MFMAs costs the issuing wave ~60 cycles during which the matrix pipe has no taker -- 4 pieces stretch a 512-cycle
// cluster to ~750 (MfmaUtil 68 %, profiles/r04_pmc_mfma_kernels.json).  The L part idles at its barrier for most of that time; v1
// (round 2) had the pieces there too, but issued them while the fragment reads were still in flight (100-185 cycles each, the L part
// outgrew the cluster).  Same ring, same counted vmcnt(4), same arithmetic: the slot of pair S+3 is the slot of pair S-1, whose reads
// both groups finished before this wave entered L(S); the data is waited for in L(S+2) and read in L(S+3).
template <bool STAGGER, bool SPLITK = false, bool TAIL = false, bool CUT = false, bool GROUPED = false, bool DMAL = false>
__global__ __launch_bounds__(GTHREADS, 2) void k_gemm_dw4(GemmArgs a) {
    static_assert(!GROUPED || (TAIL && !SPLITK && !CUT), "the grouped form is the ragged-K form per group");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3, wn = wave >> 2;
    int tm, tn;
    int sp = 0;
    int64_t krow0 = 0, kend = a.K;
    int U = (a.K + 127) / 128 * 8;                 // k16 units, whole 128-row chunks (without TAIL the host guarantees K % 128 == 0)
    if (SPLITK) {
        sp = blockIdx.x % a.nsplit;
        const int tile = blockIdx.x / a.nsplit;
        if (a.tile0 > 0) {
            tile_of_block(a, a.tile0 + tile, tm, tn);      // the tail of the full launch's own tile order
        } else {
            tm = tile / a.tiles_n;
            tn = tile % a.tiles_n;
        }
        const int chunks = (a.K + 127) / 128;
        const int c0 = (int)((int64_t)chunks * sp / a.nsplit), c1 = (int)((int64_t)chunks * (sp + 1) / a.nsplit);
        krow0 = (int64_t)c0 * 128;
        U = (c1 - c0) * 8;
    } else if (GROUPED) {
        const int tiles = a.tiles_m * a.tiles_n;
        const int g = blockIdx.x / tiles;
        tile_of_block(a, blockIdx.x - g * tiles, tm, tn);
        krow0 = a.goff[g];
        kend = a.goff[g + 1];
        U = (int)((kend - krow0 + 127) / 128) * 8;
        a.W += a.woff[g];
    } else {
        tile_of_block(a, blockIdx.x, tm, tn);
    }
    const int64_t m0 = (int64_t)tm * GB, n0 = (int64_t)tn * GB;
    const int tile_id = tm * a.tiles_n + tn;
    int cut_u = 0;                                                              // the cut in k16 units; 0: none
    if (CUT) {
        const int cut = a.kcut[tile_id];
        if (cut > 0 && cut < a.K && (cut & 31) == 0) cut_u = cut >> 4;          // (anything else in the table: one pass)
    }
    float* park = CUT ? a.ws + (int64_t)tile_id * (GB * GB) + tid * 4 : nullptr;

    const int drow = 2 * wave + (lane >> 5);
    const int lchunk = (lane & 31) ^ ((drow & 3) << 2);
    const uint16_t* srcP = a.X + (krow0 + drow) * a.ldx + n0 + lchunk * 8;
    const uint16_t* srcQ = a.Y + (krow0 + drow) * a.ldy + m0 + lchunk * 8;
    const int64_t stepP = (int64_t)GU * a.ldx, stepQ = (int64_t)GU * a.ldy;     // elements per unit
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t dmabase = lds0 + 2 * wave * ROWB;                            // wave-uniform
    const uint16_t* zsrc = g_zero_row + (lane & 31) * 8;                        // TAIL: this lane's 16 bytes of a zero row
    int64_t vrow = krow0 + drow;                                                // TAIL: absolute k-row of the next unit's piece

    const int q = lane >> 4, i = lane & 15, g = q >> 1;
    const int rowsel = i >> 2, piece = i & 3;                                     // hardware rule of ds_read_b64_tr_b16
    const int rowoff = (8 * g + rowsel) * ROWB + (piece & 1) * 8;
    uint32_t aP[2][4], aQ[2][2];                                                  // [64 KB half][tile]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int chunk = wn * 16 + ni * 4 + (q & 1) * 2 + (piece >> 1);
        aP[0][ni] = lds0 + rowoff + ((chunk ^ (rowsel << 2)) << 4);
        aP[1][ni] = aP[0][ni] + 65536;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int chunk = wm * 8 + mi * 4 + (q & 1) * 2 + (piece >> 1);
        aQ[0][mi] = lds0 + PIECE + rowoff + ((chunk ^ (rowsel << 2)) << 4);
        aQ[1][mi] = aQ[0][mi] + 65536;
    }

    f32x16_t acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    int vnext = 0;                                 // next unit to stage
    auto issue_unit = [&](int slot_unit) {         // slot_unit: 0..7 compile-time after unrolling
        const bool real = !TAIL || vrow < kend;
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcP : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcQ : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT + PIECE), 16, 0, 0);
        ++vnext;
        vrow += GU;
        const bool more = vnext < U;               // past the end the pointers stay on the last unit (staged again, never read)
        srcP += more ? stepP : 0;
        srcQ += more ? stepQ : 0;
    };
    struct Pair { s16x4_t plo[2][4], phi[2][4], qlo[2][2], qhi[2][2]; };
#define AR_RD(LO, HI, ADDR, OFF)                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                           \
                 : "=&v"(LO), "=&v"(HI)                                                                                 \
                 : "v"(ADDR), "n"(OFF), "n"((OFF) + 4 * ROWB)                                                            \
                 : "memory")
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto cat = [](const s16x4_t& lo, const s16x4_t& hi) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    Pair f;
    // one phase on pair slot S (0..3): L part then M part
    auto issue_p = [&](int slot_unit) {
        const bool real = !TAIL || vrow < kend;
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcP : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT), 16, 0, 0);
    };
    auto issue_q = [&](int slot_unit) {
        const bool real = !TAIL || vrow < kend;
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcQ : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT + PIECE), 16, 0, 0);
        ++vnext;
        vrow += GU;
        const bool more = vnext < U;
        srcP += more ? stepP : 0;
        srcQ += more ? stepQ : 0;
    };
#define AR_MMA(K, MI, NI) acc[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(f.plo[K][NI], f.phi[K][NI]), cat(f.qlo[K][MI], f.qhi[K][MI]), acc[MI][NI], 0, 0, 0)
#define AR_PIN() __builtin_amdgcn_sched_barrier(0)
    // L part: the 24 fragment reads of this pair, then wait for them and for the wave's own DMA of the NEXT pair; M part: 16 MFMAs
    // with the 4 DMA pieces of pair S+3 issued in between (an LDS-DMA piece costs ~60 issue cycles among bare MFMAs but
    // 100-185 inside a phase that also carries the fragment reads -- in v1 the four of them made the L part longer than the
    // partner's MFMA cluster: the no-DMA ablation of v1 ran 18-23 % faster, profiles/archive/r02_gemm_dw_ablation.jsonl)
#define AR_PHASE(S)                                                                                                     \
    do {                                                                                                                \
        constexpr int H = (S) >> 1;                                                                                     \
        constexpr int O = ((S) & 1) * 2 * UNIT;                                                                         \
        constexpr int DU = (((S) + 3) & 3) * 2;                                                                         \
        AR_RD(f.qlo[0][0], f.qhi[0][0], aQ[H][0], O); AR_RD(f.qlo[0][1], f.qhi[0][1], aQ[H][1], O);                     \
        AR_RD(f.plo[0][0], f.phi[0][0], aP[H][0], O); AR_RD(f.plo[0][1], f.phi[0][1], aP[H][1], O);                     \
        AR_RD(f.plo[0][2], f.phi[0][2], aP[H][2], O); AR_RD(f.plo[0][3], f.phi[0][3], aP[H][3], O);                     \
        AR_RD(f.qlo[1][0], f.qhi[1][0], aQ[H][0], O + UNIT); AR_RD(f.qlo[1][1], f.qhi[1][1], aQ[H][1], O + UNIT);       \
        AR_RD(f.plo[1][0], f.phi[1][0], aP[H][0], O + UNIT); AR_RD(f.plo[1][1], f.phi[1][1], aP[H][1], O + UNIT);       \
        AR_RD(f.plo[1][2], f.phi[1][2], aP[H][2], O + UNIT); AR_RD(f.plo[1][3], f.phi[1][3], aP[H][3], O + UNIT);       \
        asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        if (DMAL) { AR_PIN(); issue_p(DU); issue_q(DU); issue_p(DU + 1); issue_q(DU + 1); }                              \
        bar();                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        AR_MMA(0, 0, 0); AR_MMA(0, 0, 1); AR_PIN(); if (!DMAL) issue_p(DU); AR_PIN();                                   \
        AR_MMA(0, 0, 2); AR_MMA(0, 0, 3); AR_MMA(0, 1, 0); AR_MMA(0, 1, 1); AR_PIN(); if (!DMAL) issue_q(DU); AR_PIN(); \
        AR_MMA(0, 1, 2); AR_MMA(0, 1, 3); AR_MMA(1, 0, 0); AR_MMA(1, 0, 1); AR_PIN(); if (!DMAL) issue_p(DU + 1); AR_PIN(); \
        AR_MMA(1, 0, 2); AR_MMA(1, 0, 3); AR_MMA(1, 1, 0); AR_MMA(1, 1, 1); AR_PIN(); if (!DMAL) issue_q(DU + 1); AR_PIN(); \
        AR_MMA(1, 1, 2); AR_MMA(1, 1, 3);                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        bar();                                                                                                          \
    } while (0)

    // ---- prologue: pairs 0, 1, 2 in flight; pair 0 landed for everybody
#pragma unroll
    for (int v = 0; v < 6; ++v) issue_unit(v);
    wait_vm<8>();
    bar();
    if (STAGGER && grp == 1) bar();
    auto park_acc = [&]() {                          // the cut: accumulators out (32 x 16 bytes per lane, coalesced), restart from zero
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    st16f(park + ((mi * 4 + ni) * 4 + t) * (GTHREADS * 4), make_float4(acc[mi][ni][4 * t + 0], acc[mi][ni][4 * t + 1],
                                                                                      acc[mi][ni][4 * t + 2], acc[mi][ni][4 * t + 3]));
                    acc[mi][ni][4 * t + 0] = 0.f; acc[mi][ni][4 * t + 1] = 0.f; acc[mi][ni][4 * t + 2] = 0.f; acc[mi][ni][4 * t + 3] = 0.f;
                }
    };
    for (int u = 0; u < U; u += 8) {
        AR_PHASE(0);
        if (CUT && u + 2 == cut_u) park_acc();
        AR_PHASE(1);
        if (CUT && u + 4 == cut_u) park_acc();
        AR_PHASE(2);
        if (CUT && u + 6 == cut_u) park_acc();
        AR_PHASE(3);
        if (CUT && u + 8 == cut_u) park_acc();
    }
    if (STAGGER && grp == 0) bar();
    wait_vm<0>();
    if (CUT && cut_u > 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 p = *reinterpret_cast<const float4*>(park + ((mi * 4 + ni) * 4 + t) * (GTHREADS * 4));
                    acc[mi][ni][4 * t + 0] += p.x; acc[mi][ni][4 * t + 1] += p.y; acc[mi][ni][4 * t + 2] += p.z; acc[mi][ni][4 * t + 3] += p.w;
                }
    }
#undef AR_PHASE
#undef AR_RD
#undef AR_MMA
#undef AR_PIN

    const int h = lane >> 5;
    if (SPLITK) {
        const bool compact = a.tile0 > 0;
        float* wsp = compact ? a.ws + ((int64_t)(blockIdx.x / a.nsplit) * a.nsplit + sp) * (GB * GB) : a.ws + (int64_t)sp * a.M * a.N;
        const int64_t wld = compact ? GB : a.N;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int64_t m = (compact ? 0 : m0) + wm * 64 + mi * 32 + (lane & 31);
            float* rowp = wsp + m * wld + (compact ? 0 : n0) + wn * 128 + 4 * h;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    st16f(rowp + ni * 32 + 8 * t, make_float4(acc[mi][ni][4 * t + 0], acc[mi][ni][4 * t + 1], acc[mi][ni][4 * t + 2],
                                                              acc[mi][ni][4 * t + 3]));
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 32 + (lane & 31);
        uint16_t* rowp = a.W + m * a.ldw + n0 + wn * 128 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint16_t* p = rowp + ni * 32 + 8 * t;
                float v0 = acc[mi][ni][4 * t + 0], v1 = acc[mi][ni][4 * t + 1], v2 = acc[mi][ni][4 * t + 2], v3 = acc[mi][ni][4 * t + 3];
                if (a.accumulate) {
                    const uint2 old = *reinterpret_cast<const uint2*>(p);
                    v0 += bf16_lo(old.x); v1 += bf16_hi(old.x); v2 += bf16_lo(old.y); v3 += bf16_hi(old.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(p) = o;
            }
        }
    }
}

// ---- v6 (round 5): the same kernel on v_mfma_f32_16x16x32_bf16 -------------------------------------------------------------------
// Why (tools/mfma_power.hip, profiles/r05_mfma_power.jsonl): on random operands the chip is POWER-bound, and the 16x16x32 shape draws
// less per flop than 32x32x16 -- bare streams sustain 2.05 vs 1.85 PFLOP/s (2.08 vs 1.77 GHz), with the fragment reads of a 128 x 64
// wave tile 1.87 vs 1.63, with the LDS-DMA of a 256 x 256 x 64 tile step on top 1.55 vs 1.29: the two ends of that last pair are
// hipBLASLt's MI16x16x1 kernels (1.5-1.6 PF) and v3 (1.25-1.37).  A 32 x 32 accumulator tile is read and written once per 32768
// flops (8 KB), a 16 x 16 one once per 16384 (2 KB): a quarter of the accumulator traffic per flop for twice the operand traffic
// (2 KB per MFMA either way), 0.25 vs 0.31 register-file bytes per flop.
// Same ring, same DMA protocol, same barriers, same 64(m) x 128(n) wave tile as v3; a phase is ONE K-step of 32: 24 transposing
// reads (4 dY + 8 X fragments, k-rows 8q .. 8q+7 for the 16-lane group q: the operand's k = 8 (lane / 16) + j layout), then
// 32 MFMAs with the 4 DMA pieces among them.  The 4 groups of a wave now read the SAME 16 columns at k-rows 8 apart, so the
// swizzle gains bit 3 of the k-row:  chunk ^= ((k & 3) << 2) | (((k >> 3) & 1) << 1)  -- the eight 32-byte pieces a half-wave
// touches (2 groups x 4 k-rows) fall on 256 disjoint bytes.  Transposed tile as in v3 (a-operand = X fragment): a lane owns output
// row m = lane % 16 of a 16 x 16 tile and 4 consecutive n = 4 (lane / 16) .. +3.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool SPLITK = false, bool TAIL = false, bool CUT = false, bool GROUPED = false>
__global__ __launch_bounds__(GTHREADS, 2) void k_gemm_dw6(GemmArgs a) {
    static_assert(!GROUPED || (TAIL && !SPLITK && !CUT), "the grouped form is the ragged-K form per group");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3, wn = wave >> 2;
    int tm, tn;
    int sp = 0;
    int64_t krow0 = 0, kend = a.K;
    int U = (a.K + 127) / 128 * 8;
    if (SPLITK) {
        sp = blockIdx.x % a.nsplit;
        const int tile = blockIdx.x / a.nsplit;
        if (a.tile0 > 0) {
            tile_of_block(a, a.tile0 + tile, tm, tn);
        } else {
            tm = tile / a.tiles_n;
            tn = tile % a.tiles_n;
        }
        const int chunks = (a.K + 127) / 128;
        const int c0 = (int)((int64_t)chunks * sp / a.nsplit), c1 = (int)((int64_t)chunks * (sp + 1) / a.nsplit);
        krow0 = (int64_t)c0 * 128;
        U = (c1 - c0) * 8;
    } else if (GROUPED) {
        const int tiles = a.tiles_m * a.tiles_n;
        const int g = blockIdx.x / tiles;
        tile_of_block(a, blockIdx.x - g * tiles, tm, tn);
        krow0 = a.goff[g];
        kend = a.goff[g + 1];
        U = (int)((kend - krow0 + 127) / 128) * 8;
        a.W += a.woff[g];
    } else {
        tile_of_block(a, blockIdx.x, tm, tn);
    }
    const int64_t m0 = (int64_t)tm * GB, n0 = (int64_t)tn * GB;
    const int tile_id = tm * a.tiles_n + tn;
    int cut_u = 0;
    if (CUT) {
        const int cut = a.kcut[tile_id];
        if (cut > 0 && cut < a.K && (cut & 31) == 0) cut_u = cut >> 4;
    }
    // the parked accumulators of a lane are 512 contiguous bytes: ONE address register and immediate offsets (32 separate 64-bit
    // addresses do not fit beside the accumulators; spilled ones would come back through scratch loads, which share the vmcnt queue
    // the DMA protocol counts on)
    float* park = CUT ? a.ws + (int64_t)tile_id * (GB * GB) + tid * 128 : nullptr;

    const int drow = 2 * wave + (lane >> 5);                                     // k-row (0..15) of a unit's piece this lane stages
    const int lchunk = (lane & 31) ^ (((drow & 3) << 2) | (((drow >> 3) & 1) << 1));
    const uint16_t* srcP = a.X + (krow0 + drow) * a.ldx + n0 + lchunk * 8;
    const uint16_t* srcQ = a.Y + (krow0 + drow) * a.ldy + m0 + lchunk * 8;
    const int64_t stepP = (int64_t)GU * a.ldx, stepQ = (int64_t)GU * a.ldy;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t dmabase = lds0 + 2 * wave * ROWB;
    const uint16_t* zsrc = g_zero_row + (lane & 31) * 8;
    int64_t vrow = krow0 + drow;

    // fragment read addresses: group q of the wave reads k-rows 8q .. 8q+7 of the pair's 32 (unit q >> 1, rows 8 (q & 1) ..), lane i of
    // the group supplies k-row i >> 2 (+4 for the second read) and the 8-byte piece i & 3 of the fragment's 16 columns
    const int q = lane >> 4, i = lane & 15;
    const int rowsel = i >> 2, piece = i & 3;
    const int swz = (rowsel << 2) | ((q & 1) << 1);
    const int rowoff = (q >> 1) * UNIT + (8 * (q & 1) + rowsel) * ROWB + (piece & 1) * 8;
    uint32_t aP[2][8], aQ[2][4];                                                  // [64 KB half][fragment]
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
        const int chunk = wn * 16 + ni * 2 + (piece >> 1);
        aP[0][ni] = lds0 + rowoff + ((chunk ^ swz) << 4);
        aP[1][ni] = aP[0][ni] + 65536;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int chunk = wm * 8 + mi * 2 + (piece >> 1);
        aQ[0][mi] = lds0 + PIECE + rowoff + ((chunk ^ swz) << 4);
        aQ[1][mi] = aQ[0][mi] + 65536;
    }

    f32x4_t acc[4][8];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;

    int vnext = 0;
    auto issue_unit = [&](int slot_unit) {
        const bool real = !TAIL || vrow < kend;
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcP : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcQ : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT + PIECE), 16, 0, 0);
        ++vnext;
        vrow += GU;
        const bool more = vnext < U;
        srcP += more ? stepP : 0;
        srcQ += more ? stepQ : 0;
    };
    auto issue_p = [&](int slot_unit) {
        const bool real = !TAIL || vrow < kend;
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcP : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT), 16, 0, 0);
    };
    auto issue_q = [&](int slot_unit) {
        const bool real = !TAIL || vrow < kend;
        __builtin_amdgcn_global_load_lds((const void*)(real ? srcQ : zsrc), (__attribute__((address_space(3))) void*)(uintptr_t)(dmabase + slot_unit * UNIT + PIECE), 16, 0, 0);
        ++vnext;
        vrow += GU;
        const bool more = vnext < U;
        srcP += more ? stepP : 0;
        srcQ += more ? stepQ : 0;
    };
    struct Frags { s16x4_t plo[8], phi[8], qlo[4], qhi[4]; };
#define AR_RD(LO, HI, ADDR, OFF)                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                           \
                 : "=&v"(LO), "=&v"(HI)                                                                                 \
                 : "v"(ADDR), "n"(OFF), "n"((OFF) + 4 * ROWB)                                                            \
                 : "memory")
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto cat = [](const s16x4_t& lo, const s16x4_t& hi) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    Frags f;
#define AR_MMA(MI, NI) acc[MI][NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat(f.plo[NI], f.phi[NI]), cat(f.qlo[MI], f.qhi[MI]), acc[MI][NI], 0, 0, 0)
#define AR_MMA4(MI, N0) AR_MMA(MI, N0); AR_MMA(MI, N0 + 1); AR_MMA(MI, N0 + 2); AR_MMA(MI, N0 + 3)
#define AR_PIN() __builtin_amdgcn_sched_barrier(0)
#define AR_PHASE(S)                                                                                                     \
    do {                                                                                                                \
        constexpr int H = (S) >> 1;                                                                                     \
        constexpr int O = ((S) & 1) * 2 * UNIT;                                                                         \
        constexpr int DU = (((S) + 3) & 3) * 2;                                                                         \
        AR_RD(f.qlo[0], f.qhi[0], aQ[H][0], O); AR_RD(f.qlo[1], f.qhi[1], aQ[H][1], O);                                 \
        AR_RD(f.plo[0], f.phi[0], aP[H][0], O); AR_RD(f.plo[1], f.phi[1], aP[H][1], O);                                 \
        AR_RD(f.plo[2], f.phi[2], aP[H][2], O); AR_RD(f.plo[3], f.phi[3], aP[H][3], O);                                 \
        AR_RD(f.plo[4], f.phi[4], aP[H][4], O); AR_RD(f.plo[5], f.phi[5], aP[H][5], O);                                 \
        AR_RD(f.plo[6], f.phi[6], aP[H][6], O); AR_RD(f.plo[7], f.phi[7], aP[H][7], O);                                 \
        AR_RD(f.qlo[2], f.qhi[2], aQ[H][2], O); AR_RD(f.qlo[3], f.qhi[3], aQ[H][3], O);                                 \
        asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        bar();                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        AR_MMA4(0, 0); AR_PIN(); issue_p(DU); AR_PIN();                                                                 \
        AR_MMA4(0, 4); AR_MMA4(1, 0); AR_PIN(); issue_q(DU); AR_PIN();                                                  \
        AR_MMA4(1, 4); AR_MMA4(2, 0); AR_PIN(); issue_p(DU + 1); AR_PIN();                                              \
        AR_MMA4(2, 4); AR_MMA4(3, 0); AR_PIN(); issue_q(DU + 1); AR_PIN();                                              \
        AR_MMA4(3, 4);                                                                                                  \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        bar();                                                                                                          \
    } while (0)

#pragma unroll
    for (int v = 0; v < 6; ++v) issue_unit(v);
    wait_vm<8>();
    bar();
    if (grp == 1) bar();
    auto park_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                st16f(park + (mi * 8 + ni) * 4, make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]));
                acc[mi][ni][0] = 0.f; acc[mi][ni][1] = 0.f; acc[mi][ni][2] = 0.f; acc[mi][ni][3] = 0.f;
            }
    };
    for (int u = 0; u < U; u += 8) {
        AR_PHASE(0);
        if (CUT && u + 2 == cut_u) park_acc();
        AR_PHASE(1);
        if (CUT && u + 4 == cut_u) park_acc();
        AR_PHASE(2);
        if (CUT && u + 6 == cut_u) park_acc();
        AR_PHASE(3);
        if (CUT && u + 8 == cut_u) park_acc();
    }
    if (grp == 0) bar();
    wait_vm<0>();
    if (CUT && cut_u > 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const float4 p = *reinterpret_cast<const float4*>(park + (mi * 8 + ni) * 4);
                acc[mi][ni][0] += p.x; acc[mi][ni][1] += p.y; acc[mi][ni][2] += p.z; acc[mi][ni][3] += p.w;
            }
    }
#undef AR_PHASE
#undef AR_RD
#undef AR_MMA
#undef AR_MMA4
#undef AR_PIN

    const int mrow = lane & 15, ncol = 4 * (lane >> 4);
    if (SPLITK) {
        const bool compact = a.tile0 > 0;
        float* wsp = compact ? a.ws + ((int64_t)(blockIdx.x / a.nsplit) * a.nsplit + sp) * (GB * GB) : a.ws + (int64_t)sp * a.M * a.N;
        const int64_t wld = compact ? GB : a.N;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int64_t m = (compact ? 0 : m0) + wm * 64 + mi * 16 + mrow;
            float* rowp = wsp + m * wld + (compact ? 0 : n0) + wn * 128 + ncol;
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
                st16f(rowp + ni * 16, make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]));
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 16 + mrow;
        uint16_t* rowp = a.W + m * a.ldw + n0 + wn * 128 + ncol;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            uint16_t* p = rowp + ni * 16;
            float v0 = acc[mi][ni][0], v1 = acc[mi][ni][1], v2 = acc[mi][ni][2], v3 = acc[mi][ni][3];
            if (a.accumulate) {
                const uint2 old = *reinterpret_cast<const uint2*>(p);
                v0 += bf16_lo(old.x); v1 += bf16_hi(old.x); v2 += bf16_lo(old.y); v3 += bf16_hi(old.y);
            }
            uint2 o;
            o.x = pack_bf16x2(v0, v1);
            o.y = pack_bf16x2(v2, v3);
            *reinterpret_cast<uint2*>(p) = o;
        }
    }
}

#ifdef AR_GEMM_EXPERIMENTS
// ---- v4: four waves, one per SIMD, 128 x 128 per wave; operands staged through registers -------------------------------------
// What v3 taught (profiles/archive/r02_gemm_dw_ablation.jsonl, r02_gemm_dw_pmc.json): with eight waves the fragment reads move 6 KB of LDS
// per wave and K-step of 16 for 64x128 of output, and every LDS-DMA piece issued among the MFMAs stalls its wave for 60-185 issue
// cycles -- the no-DMA ablation ran 15-20 % faster.  hipBLASLt's NT kernels (forward GEMMs, 1.5 PFLOP/s here) use the other
// classical shape: 4 waves x (128 x 128), i.e. 8 KB of fragments per K-step for twice the output, 256 accumulator registers per
// lane (the unified 512-entry file makes that one wave per SIMD), and plain global loads into registers followed by ds_write.
// v4 is that shape for the TN problem: same [k][256] swizzled LDS image and transposing fragment reads as v0-v3; a K-unit of 16 per
// phase = 16 MFMAs (512 matrix-pipe cycles) that shadow the 16 fragment reads of the next unit, the 4 ds_write_b128 of unit +3
// and the 4 global_load_dwordx4 of unit +3+L (L units = 16*L registers of loads in flight); one s_barrier every second phase.
// Result (profiles/archive/r02_gemm_dw_v4_vs_v3.jsonl): bit-identical output, 1.19-1.33 PFLOP/s against v3's 1.23-1.36 on the same shapes,
// so v3 stays the default and v4 is reachable only through ar_gemm_dw_config(18 | 19) for tools/gemm_dw_probe.py.  Its timing
// ablations (r02_gemm_dw_v4_ablation.json) are the useful part: MFMAs + barriers alone 1.6-2.0 PFLOP/s (the matrix pipe at the
// clock the chip sustains), without the LDS stores 1.45-1.55, without the loads 1.22-1.36, without the fragment reads 1.25-1.7 --
// with one wave per SIMD every cycle another instruction holds the issue port beyond an MFMA's 32 is lost, and the VGPR->LDS
// store path (13 cycles per ds_write_b128, shared by a SIMD pair) is the largest such item; v3's second wave per SIMD hides it.
template <int L, bool SPLITK>
__global__ __launch_bounds__(256, 1) void k_gemm_dw5(GemmArgs a) {
    static_assert(8 % L == 0, "the load ring must divide the unroll of 8 phases");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    int tm, tn, sp = 0;
    int64_t krow0 = 0;
    int U = a.K / GU;                              // K % 128 == 0 (host)
    if (SPLITK) {
        sp = blockIdx.x % a.nsplit;
        const int tile = blockIdx.x / a.nsplit;
        tm = tile / a.tiles_n;
        tn = tile % a.tiles_n;
        const int chunks = a.K / 128;
        const int c0 = (int)((int64_t)chunks * sp / a.nsplit), c1 = (int)((int64_t)chunks * (sp + 1) / a.nsplit);
        krow0 = (int64_t)c0 * 128;
        U = (c1 - c0) * 8;
    } else {
        tile_of_block(a, blockIdx.x, tm, tn);
    }
    const int64_t m0 = (int64_t)tm * GB, n0 = (int64_t)tn * GB;

    // ---- staging: this wave moves k-rows 4w .. 4w+3 of each operand's piece, two rows per 16-byte-per-lane instruction
    const uint16_t* srcP[2];
    const uint16_t* srcQ[2];
    int dstoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 4 * wave + 2 * j + (lane >> 5);
        const int pchunk = lane & 31;
        const int lchunk = pchunk ^ ((r & 3) << 2);
        srcP[j] = a.X + (krow0 + r) * a.ldx + n0 + lchunk * 8;
        srcQ[j] = a.Y + (krow0 + r) * a.ldy + m0 + lchunk * 8;
        dstoff[j] = r * ROWB + pchunk * 16;
    }
    const int64_t stepP = (int64_t)GU * a.ldx, stepQ = (int64_t)GU * a.ldy;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;

    // ---- fragment read addresses (ds_read_b64_tr_b16: lane i of a 16-lane group supplies k-row i>>2, 8-byte piece i&3)
    const int q = lane >> 4, i = lane & 15, g = q >> 1;
    const int rowsel = i >> 2, piece = i & 3;
    const int rowoff = (8 * g + rowsel) * ROWB + (piece & 1) * 8;
    uint32_t aP[2][4], aQ[2][4];                   // [64 KB half of the ring][tile]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int cp = wn * 16 + t * 4 + (q & 1) * 2 + (piece >> 1);
        const int cq = wm * 16 + t * 4 + (q & 1) * 2 + (piece >> 1);
        aP[0][t] = lds0 + rowoff + ((cp ^ (rowsel << 2)) << 4);
        aQ[0][t] = lds0 + PIECE + rowoff + ((cq ^ (rowsel << 2)) << 4);
        aP[1][t] = aP[0][t] + 65536;
        aQ[1][t] = aQ[0][t] + 65536;
    }

    f32x16_t acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    struct Frag { s16x4_t plo[4], phi[4], qlo[4], qhi[4]; };
    Frag F[2];
    u32x4_t G[L][4];                                 // loads in flight: [unit % L][P rows j=0,1 | Q rows j=0,1]; constant indices only
    int vnext = 0;                                 // next unit to load
#define AR_LOAD_UNIT(GS)                                                                                                \
    do {                                                                                                                \
        G[GS][0] = *reinterpret_cast<const u32x4_t*>(srcP[0]);                                                            \
        G[GS][1] = *reinterpret_cast<const u32x4_t*>(srcP[1]);                                                            \
        G[GS][2] = *reinterpret_cast<const u32x4_t*>(srcQ[0]);                                                            \
        G[GS][3] = *reinterpret_cast<const u32x4_t*>(srcQ[1]);                                                            \
        ++vnext;                                                                                                        \
        const bool more = vnext < U;       /* past the end the pointers stay on the last unit (staged again, never read) */ \
        srcP[0] += more ? stepP : 0; srcP[1] += more ? stepP : 0;                                                       \
        srcQ[0] += more ? stepQ : 0; srcQ[1] += more ? stepQ : 0;                                                       \
    } while (0)
#define AR_STORE_UNIT(SL, GS)                                                                                           \
    do {                                                                                                                \
        uint8_t* base_ = lds + (SL) * UNIT;                                                                             \
        *reinterpret_cast<u32x4_t*>(base_ + dstoff[0]) = G[GS][0];                                                        \
        *reinterpret_cast<u32x4_t*>(base_ + dstoff[1]) = G[GS][1];                                                        \
        *reinterpret_cast<u32x4_t*>(base_ + PIECE + dstoff[0]) = G[GS][2];                                                \
        *reinterpret_cast<u32x4_t*>(base_ + PIECE + dstoff[1]) = G[GS][3];                                                \
    } while (0)
#define AR_RD5(LO, HI, ADDR, OFF)                                                                                       \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                           \
                 : "=&v"(LO), "=&v"(HI)                                                                                 \
                 : "v"(ADDR), "n"(OFF), "n"((OFF) + 4 * ROWB)                                                            \
                 : "memory")
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto cat = [](const s16x4_t& lo, const s16x4_t& hi) -> bf16x8_t {
        return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#define AR_PIN5() __builtin_amdgcn_sched_barrier(0)
#define AR_MMA5(FB, MI, NI)                                                                                             \
    acc[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(F[FB].plo[NI], F[FB].phi[NI]), cat(F[FB].qlo[MI], F[FB].qhi[MI]), \
                                                          acc[MI][NI], 0, 0, 0)
    // fragment reads of the unit in ring slot SL into F[NB], tile T of P / of Q
#define AR_RDP(NB, SL, T) AR_RD5(F[NB].plo[T], F[NB].phi[T], aP[(SL) >> 2][T], ((SL) & 3) * UNIT)
#define AR_RDQ(NB, SL, T) AR_RD5(F[NB].qlo[T], F[NB].qhi[T], aQ[(SL) >> 2][T], ((SL) & 3) * UNIT)
    // one phase: S = unit index mod 8 (compile time).  MFMAs of unit S from F[S&1]; reads of unit S+1; stores of unit S+3; loads of
    // unit S+3+L.  The stores sit late in the phase: their loads were issued L phases ago.
#define AR_ST1(SL, GS, J) *reinterpret_cast<u32x4_t*>(lds + (SL) * UNIT + ((J) >> 1) * PIECE + dstoff[(J) & 1]) = G[GS][J]
#define AR_LD1P(GS, J) G[GS][J] = *reinterpret_cast<const u32x4_t*>(srcP[J])
#define AR_LD1Q(GS, J) G[GS][2 + (J)] = *reinterpret_cast<const u32x4_t*>(srcQ[J])
#define AR_ADVANCE()                                                                                                    \
    do {                                                                                                                \
        ++vnext;                                                                                                        \
        const bool more = vnext < U;                                                                                    \
        srcP[0] += more ? stepP : 0; srcP[1] += more ? stepP : 0;                                                       \
        srcQ[0] += more ? stepQ : 0; srcQ[1] += more ? stepQ : 0;                                                       \
    } while (0)
    // every MFMA is followed by at most ~25 issue cycles of other work (the matrix pipe is busy 32): three fragment reads, or two
    // and a store, or one load -- a single wave per SIMD has nobody to cover a longer gap
    // The stores sit in the second half of the phase, between the loads: measured best of the placements tried (stores first:
    // -12 %; the two waves of a store-path half in opposite halves of the phase: the duplicated loop body halved the rate).
#define AR_PHASE5(S, EARLY)                                                                                             \
    do {                                                                                                                \
        constexpr int FB = (S) & 1, NB = FB ^ 1, SR = ((S) + 1) & 7, SW = ((S) + 3) & 7, GS = ((S) + 3) % L;             \
        AR_MMA5(FB, 0, 0); AR_PIN5(); AR_RDQ(NB, SR, 0); if (EARLY) AR_ST1(SW, GS, 0); AR_PIN5();                        \
        AR_MMA5(FB, 0, 1); AR_PIN5(); AR_RDP(NB, SR, 0); AR_PIN5();                                                      \
        AR_MMA5(FB, 0, 2); AR_PIN5(); AR_RDP(NB, SR, 1); if (EARLY) AR_ST1(SW, GS, 1); AR_PIN5();                        \
        AR_MMA5(FB, 0, 3); AR_PIN5(); AR_RDQ(NB, SR, 1); AR_PIN5();                                                      \
        AR_MMA5(FB, 1, 0); AR_PIN5(); AR_RDP(NB, SR, 2); if (EARLY) AR_ST1(SW, GS, 2); AR_PIN5();                        \
        AR_MMA5(FB, 1, 1); AR_PIN5(); AR_RDP(NB, SR, 3); AR_PIN5();                                                      \
        AR_MMA5(FB, 1, 2); AR_PIN5(); AR_RDQ(NB, SR, 2); if (EARLY) AR_ST1(SW, GS, 3); AR_PIN5();                        \
        AR_MMA5(FB, 1, 3); AR_PIN5(); AR_RDQ(NB, SR, 3); AR_PIN5();                                                      \
        AR_MMA5(FB, 2, 0); AR_PIN5(); if (!(EARLY)) AR_ST1(SW, GS, 0); AR_PIN5();                                        \
        AR_MMA5(FB, 2, 1); AR_PIN5(); AR_LD1P(GS, 0); AR_PIN5();                                                         \
        AR_MMA5(FB, 2, 2); AR_PIN5(); if (!(EARLY)) AR_ST1(SW, GS, 1); AR_PIN5();                                        \
        AR_MMA5(FB, 2, 3); AR_PIN5(); AR_LD1P(GS, 1); AR_PIN5();                                                         \
        AR_MMA5(FB, 3, 0); AR_PIN5(); if (!(EARLY)) AR_ST1(SW, GS, 2); AR_PIN5();                                        \
        AR_MMA5(FB, 3, 1); AR_PIN5(); AR_LD1Q(GS, 0); AR_PIN5();                                                         \
        AR_MMA5(FB, 3, 2); AR_PIN5(); if (!(EARLY)) AR_ST1(SW, GS, 3); AR_PIN5();                                        \
        AR_MMA5(FB, 3, 3); AR_PIN5(); AR_LD1Q(GS, 1); AR_ADVANCE(); AR_PIN5();                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
        if (((S) & 1) == 1) { AR_PIN5(); __builtin_amdgcn_s_barrier(); AR_PIN5(); }                                      \
    } while (0)

    // ---- prologue: units 0..2 into LDS, units 3..3+L-1 in flight, fragments of unit 0 in F[0]
    AR_LOAD_UNIT(0); AR_STORE_UNIT(0, 0);
    AR_LOAD_UNIT(1); AR_STORE_UNIT(1, 1);
    AR_LOAD_UNIT(2); AR_STORE_UNIT(2, 2);
    AR_LOAD_UNIT(3 % L); AR_LOAD_UNIT(4 % L); AR_LOAD_UNIT(5 % L); AR_LOAD_UNIT(6 % L);
    if constexpr (L == 8) { AR_LOAD_UNIT(7); AR_LOAD_UNIT(0); AR_LOAD_UNIT(1); AR_LOAD_UNIT(2); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    AR_PIN5(); __builtin_amdgcn_s_barrier(); AR_PIN5();
    AR_RDQ(0, 0, 0); AR_RDQ(0, 0, 1); AR_RDQ(0, 0, 2); AR_RDQ(0, 0, 3);
    AR_RDP(0, 0, 0); AR_RDP(0, 0, 1); AR_RDP(0, 0, 2); AR_RDP(0, 0, 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int u = 0; u < U; u += 8) {
        AR_PHASE5(0, false); AR_PHASE5(1, false); AR_PHASE5(2, false); AR_PHASE5(3, false);
        AR_PHASE5(4, false); AR_PHASE5(5, false); AR_PHASE5(6, false); AR_PHASE5(7, false);
    }
#undef AR_PHASE5
#undef AR_RDP
#undef AR_RDQ
#undef AR_MMA5
#undef AR_PIN5
#undef AR_RD5
#undef AR_LOAD_UNIT
#undef AR_STORE_UNIT
#undef AR_ST1
#undef AR_LD1P
#undef AR_LD1Q
#undef AR_ADVANCE

    const int h = lane >> 5;
    if (SPLITK) {
        float* wsp = a.ws + (int64_t)sp * a.M * a.N;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int64_t m = m0 + wm * 128 + mi * 32 + (lane & 31);
            float* rowp = wsp + m * a.N + n0 + wn * 128 + 4 * h;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    st16f(rowp + ni * 32 + 8 * t, make_float4(acc[mi][ni][4 * t + 0], acc[mi][ni][4 * t + 1], acc[mi][ni][4 * t + 2],
                                                              acc[mi][ni][4 * t + 3]));
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 128 + mi * 32 + (lane & 31);
        uint16_t* rowp = a.W + m * a.ldw + n0 + wn * 128 + 4 * h;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint16_t* p = rowp + ni * 32 + 8 * t;
                float v0 = acc[mi][ni][4 * t + 0], v1 = acc[mi][ni][4 * t + 1], v2 = acc[mi][ni][4 * t + 2], v3 = acc[mi][ni][4 * t + 3];
                if (a.accumulate) {
                    const uint2 old = *reinterpret_cast<const uint2*>(p);
                    v0 += bf16_lo(old.x); v1 += bf16_hi(old.x); v2 += bf16_lo(old.y); v3 += bf16_hi(old.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(p) = o;
            }
        }
    }
}

#endif  // AR_GEMM_EXPERIMENTS

// sum of the split-K slices in slice order (+ the previous bf16 value when accumulating), one rounding to bf16
__global__ __launch_bounds__(kTPB) void k_splitk_reduce(const float* __restrict__ ws, int nsplit, int64_t M, int64_t N, uint16_t* __restrict__ W,
                                                         int64_t ldw, int accumulate) {
    const int64_t cpr = N / kEPT;
    const int64_t idx = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    if (idx >= M * cpr) return;
    const int64_t m = idx / cpr, c = idx - m * cpr;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        float p[8];
        unpack_f8(load8_f32(ws, ((int64_t)s * M + m) * N + c * kEPT), p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += p[j];
    }
    if (accumulate) {
        float o[8];
        unpack8<AR_DT_BF16>(load8_raw<AR_DT_BF16>(W, m * ldw + c * kEPT), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += o[j];
    }
    store8<AR_DT_BF16>(W, m * ldw + c * kEPT, v);
}

// the same for a hybrid launch: only the `ntail` last tiles of the launch order were split; their partial tiles lie compactly
// [tile][nsplit][256][256].  32 workgroups per tile, one 16-byte piece of the result per lane.
__global__ __launch_bounds__(kTPB) void k_splitk_reduce_tiles(GemmArgs a) {
    const int t = blockIdx.x >> 5;
    const int idx = (blockIdx.x & 31) * kTPB + threadIdx.x;          // 0 .. 8191: 256 rows x 32 pieces
    const int ml = idx >> 5, c = idx & 31;
    int tm, tn;
    tile_of_block(a, a.tile0 + t, tm, tn);
    const float* base = a.ws + (int64_t)t * a.nsplit * (GB * GB) + ml * GB + c * kEPT;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int s = 0; s < a.nsplit; ++s) {
        float p[8];
        unpack_f8(load8_f32(base, (int64_t)s * (GB * GB)), p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += p[j];
    }
    const int64_t off = ((int64_t)tm * GB + ml) * a.ldw + (int64_t)tn * GB + c * kEPT;
    if (a.accumulate) {
        float o[8];
        unpack8<AR_DT_BF16>(load8_raw<AR_DT_BF16>(a.W, off), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += o[j];
    }
    store8<AR_DT_BF16>(a.W, off, v);
}

static int g_gemm_kernel = 7;     // 0: v0  1: v1 staggered  2: v1 lockstep  3: v2 (split reads)  4-6: timing ablations  7: v3 (DMA in the MFMA cluster)
static int g_gemm_tail = 1;       // hybrid split of the last partial round (ar_gemm_dw_config(20 | 21) switches it for the A/B)
static int g_gemm_sem = 1, g_gemm_order = 2;    // rule 1 is what the hardware does (profiles/archive/r02_mfma_probe.json)
static int g_gemm_dmal = 2;       // 2: v6 (16x16x32, the default since round 5);  0: v3 (32x32x16);  1: v3 with the DMA pieces at the end of the L part   (ar_gemm_dw_config(32 | 30 | 31))

}  // namespace ar

using namespace ar;

extern "C" int ar_gemm_dw_config(int sem, int order) {      // experiment knobs (tools/gemm_dw_probe.py); -1 keeps a value
#ifdef AR_GEMM_EXPERIMENTS
    if (sem == 1 || sem == 2) g_gemm_sem = sem;                 // v0 only: lane->piece rule (2 = negative control)
    if (sem >= 10 && sem <= 19) g_gemm_kernel = sem - 10;       // 10: v0, 11: v1 staggered, 12: v1 lockstep, 14-16: ablations, 17: v3, 18 / 19: v4
#endif
    if (sem == 20 || sem == 21) g_gemm_tail = sem - 20;
    if (sem >= 30 && sem <= 32) g_gemm_dmal = sem - 30;
    if (order >= 0 && order <= 2) g_gemm_order = order;
    return g_gemm_kernel * 100 + g_gemm_sem * 10 + g_gemm_order;
}

// kernel of a given form, by the A/B knob ar_gemm_dw_config(30 | 31 | 32): v3 with the DMA pieces in the M part (rounds 2-4), v3 with
// them at the end of the L part (round 5, slower), v6 = the 16x16x32 form (round 5)
typedef void (*dw4_fn)(GemmArgs);
template <bool SPLITK, bool TAIL, bool CUT, bool GROUPED>
static dw4_fn dw4_kernel() {
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, SPLITK, TAIL, CUT, GROUPED, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, SPLITK, TAIL, CUT, GROUPED, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_dw6<SPLITK, TAIL, CUT, GROUPED>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    }
    if (g_gemm_dmal == 2) return k_gemm_dw6<SPLITK, TAIL, CUT, GROUPED>;
    return g_gemm_dmal ? k_gemm_dw4<true, SPLITK, TAIL, CUT, GROUPED, true> : k_gemm_dw4<true, SPLITK, TAIL, CUT, GROUPED, false>;
}

// One 512-thread workgroup holds a CU (128 KB of LDS): 256 run at a time, and a launch of 256 r + t workgroups costs r + 1 rounds
// however small t is.  The plan therefore never EXCEEDS a round: tiles x nsplit <= 256 (round 2 rounded the quotient up -- 288
// workgroups for OPT-125M's 3072 x 768 weight, i.e. a second round for 32 of them).
constexpr int kCUs = 256;
static int splitk_plan(int64_t M, int64_t N, int64_t K) {
    if (M % GB || N % GB) return 1;
    const int64_t tiles = (M / GB) * (N / GB);
    if (tiles > kCUs / 2 || K < 1024) return 1;
    int64_t ns = kCUs / tiles;
    if (ns > K / 512) ns = K / 512;
    return ns < 2 ? 1 : (int)ns;
}
// Hybrid plan for many tiles: when the last round of the launch holds <= 128 tiles they are split along K so that the round is
// full and 1 / nsplit as long (Llama-3-8B: the merged q/k/v weight is 384 tiles = 1.5 rounds, down_proj 896 = 3.5 rounds).
// -> number of tail tiles (0: no hybrid), *ns = their split
static int tail_plan(int64_t M, int64_t N, int64_t K, int* ns) {
    *ns = 1;
    if (M % GB || N % GB || K < 2048) return 0;
    const int64_t tiles = (M / GB) * (N / GB);
    const int r = (int)(tiles % kCUs);
    if (tiles <= kCUs || r == 0 || r > kCUs / 2) return 0;
    int64_t n = kCUs / r;
    if (n > K / 1024) n = K / 1024;
    if (n < 2) return 0;
    *ns = (int)n;
    return r;
}

extern "C" int64_t ar_gemm_dw_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int ns = splitk_plan(M, N, K);
    if (ns > 1) return (int64_t)ns * M * N * 4;
    int tns;
    const int r = tail_plan(M, N, K, &tns);
    return r ? (int64_t)r * tns * GB * GB * 4 : 0;
}

static int gemm_dw_impl(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx, int64_t ldw,
                        int accumulate, void* workspace, int64_t workspace_bytes, int force_ns, ar_stream_t stream);

extern "C" int ar_gemm_dw(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx,
                          int64_t ldw, int accumulate, void* workspace, int64_t workspace_bytes, ar_stream_t stream) {
    return gemm_dw_impl(dY, X, dW, M, N, K, ldy, ldx, ldw, accumulate, workspace, workspace_bytes, 0, stream);
}

// nsplit >= 1: every output tile is computed as `nsplit` contiguous K slices (whole 128-row chunks, slice s = chunks s/nsplit ..
// (s+1)/nsplit) summed in slice order -- the caller chooses the summation structure instead of the launch-shape heuristics
// (1: one pass over K).  Needs nsplit * M * N * 4 bytes of workspace for nsplit > 1.
extern "C" int ar_gemm_dw_ex(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx,
                             int64_t ldw, int accumulate, void* workspace, int64_t workspace_bytes, int nsplit, ar_stream_t stream) {
    if (nsplit < 1 || nsplit > 64) return AR_ERR_UNSUPPORTED;
    return gemm_dw_impl(dY, X, dW, M, N, K, ldy, ldx, ldw, accumulate, workspace, workspace_bytes, nsplit, stream);
}

// The summation structure of a stream-K GEMM, given by the caller: tile t (row-major 256 x 256 tile id) is summed in one pass over
// K when kcut[t] == 0, else in two parts, k-rows [0, kcut[t]) and [kcut[t], K) (a multiple of 32, 0 < kcut < K), each from a zero
// accumulator, added in fp32 and rounded once.  This is how the exact path reproduces, bit for bit, a library kernel that streams
// its last tiles over a fixed grid (auto_round_amd/streamk.py).  kcut lives on the device.  Workspace: tiles * 256 * 256 * 4 bytes
// (only the two-part tiles' slots are touched).
extern "C" int ar_gemm_dw_sk(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx,
                             int64_t ldw, void* workspace, int64_t workspace_bytes, const int32_t* kcut, ar_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return AR_OK;
    if (M % GB || N % GB || K < 128 || (ldy % 8) || (ldx % 8) || (ldw % 4) || !kcut) return AR_ERR_UNSUPPORTED;
    if ((((uintptr_t)dY | (uintptr_t)X) & 15) || ((uintptr_t)dW & 7) || ((uintptr_t)workspace & 15)) return AR_ERR_UNSUPPORTED;
    GemmArgs a;
    a.nsplit = 1; a.tile0 = 0; a.kcut = kcut; a.ws = (float*)workspace; a.goff = nullptr; a.woff = nullptr;
    a.Y = (const uint16_t*)dY; a.X = (const uint16_t*)X; a.W = (uint16_t*)dW;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.ldy = ldy; a.ldx = ldx; a.ldw = ldw; a.accumulate = 0;
    a.tiles_m = (int)(M / GB); a.tiles_n = (int)(N / GB); a.order = g_gemm_order;
    const int grid = a.tiles_m * a.tiles_n;
    if (!workspace || workspace_bytes < (int64_t)grid * GB * GB * 4) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    static PerDeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    }
    if (K % 128) AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (dw4_kernel<false, true, true, false>()), grid, GTHREADS, GEMM_LDS, st, a);
    else AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (dw4_kernel<false, false, true, false>()), grid, GTHREADS, GEMM_LDS, st, a);
    return launch_status();
}

// One launch for all the experts of a sparse-MoE projection: n_groups tile grids, every workgroup reads its group's k-row range from
// device memory (see the GROUPED note at k_gemm_dw4).
extern "C" int ar_gemm_dw_grouped(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t ldy, int64_t ldx, int64_t ldw,
                                  const int32_t* row_off, const int64_t* w_off, int n_groups, ar_stream_t stream) {
    if (M <= 0 || N <= 0 || n_groups <= 0) return AR_OK;
    if (M % GB || N % GB || (ldy % 8) || (ldx % 8) || (ldw % 4) || !row_off || !w_off) return AR_ERR_UNSUPPORTED;
    if ((((uintptr_t)dY | (uintptr_t)X) & 15) || ((uintptr_t)dW & 7)) return AR_ERR_UNSUPPORTED;
    GemmArgs a;
    a.ws = nullptr; a.nsplit = 1; a.tile0 = 0; a.kcut = nullptr;
    a.Y = (const uint16_t*)dY; a.X = (const uint16_t*)X; a.W = (uint16_t*)dW;
    a.M = (int)M; a.N = (int)N; a.K = 0; a.ldy = ldy; a.ldx = ldx; a.ldw = ldw; a.accumulate = 0;
    a.tiles_m = (int)(M / GB); a.tiles_n = (int)(N / GB); a.order = g_gemm_order;
    a.goff = row_off; a.woff = w_off;
    const int64_t grid = (int64_t)a.tiles_m * a.tiles_n * n_groups;
    if (grid > (1 << 30)) return AR_ERR_UNSUPPORTED;
    static PerDeviceOnce once;
    if (once.first())
        (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N * n_groups, (dw4_kernel<false, true, false, true>()), (int)grid, GTHREADS, GEMM_LDS,
                   (hipStream_t)stream, a);
    return launch_status();
}

static int gemm_dw_impl(const void* dY, const void* X, void* dW, int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx, int64_t ldw,
                        int accumulate, void* workspace, int64_t workspace_bytes, int force_ns, ar_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0) return AR_OK;
    if (M % GB || N % GB || K < GD * GU || (ldy % 8) || (ldx % 8) || (ldw % 4)) return AR_ERR_UNSUPPORTED;
    if (K % 128 && g_gemm_kernel != 7) return AR_ERR_UNSUPPORTED;       // only the default kernel completes a ragged K with zeros
    if (((uintptr_t)dY | (uintptr_t)X) & 15 || ((uintptr_t)dW & 7)) return AR_ERR_UNSUPPORTED;
    GemmArgs a;
    a.ws = nullptr; a.nsplit = 1; a.tile0 = 0; a.kcut = nullptr; a.goff = nullptr; a.woff = nullptr;
    a.Y = (const uint16_t*)dY; a.X = (const uint16_t*)X; a.W = (uint16_t*)dW;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.ldy = ldy; a.ldx = ldx; a.ldw = ldw; a.accumulate = accumulate;
    a.tiles_m = (int)(M / GB); a.tiles_n = (int)(N / GB); a.order = g_gemm_order;
    const int grid = a.tiles_m * a.tiles_n;
    hipStream_t st = (hipStream_t)stream;
#ifdef AR_GEMM_EXPERIMENTS
    static PerDeviceOnce attr_done;
    if (attr_done.first()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_dw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_dw<2>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_dw2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_dw2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e != hipSuccess) return (int)e;
    }
    if (g_gemm_kernel >= 8 && K % 128 == 0 && K >= 128) {
        static PerDeviceOnce a5;
        if (a5.first()) {
            (void)hipFuncSetAttribute((const void*)k_gemm_dw5<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            (void)hipFuncSetAttribute((const void*)k_gemm_dw5<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        }
        if (g_gemm_kernel == 8) AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (k_gemm_dw5<4, false>), grid, 256, GEMM_LDS, st, a);
        else AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (k_gemm_dw5<8, false>), grid, 256, GEMM_LDS, st, a);
        return launch_status();
    }
#endif  // AR_GEMM_EXPERIMENTS
    if (g_gemm_kernel >= 1 && (K % 128 == 0 || g_gemm_kernel == 7) && K >= 128) {
        if (g_gemm_kernel == 7) {
            static PerDeviceOnce a4;
            if (a4.first()) {
                (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
                (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
                (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
                (void)hipFuncSetAttribute((const void*)k_gemm_dw4<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            }
            const int ns = force_ns > 0 ? force_ns : splitk_plan(M, N, K);
            if (force_ns > 1 && !(workspace && workspace_bytes >= (int64_t)ns * M * N * 4 && (ldw % 8) == 0 && !((uintptr_t)dW & 15) && K / 128 >= ns))
                return AR_ERR_UNSUPPORTED;          // a forced structure is either delivered or refused, never silently replaced
            if (ns > 1 && workspace && workspace_bytes >= (int64_t)ns * M * N * 4 && (ldw % 8) == 0 && !((uintptr_t)dW & 15)) {
                a.ws = (float*)workspace; a.nsplit = ns;
                if (K % 128) AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (dw4_kernel<true, true, false, false>()), grid * ns, GTHREADS, GEMM_LDS, st, a);
                else AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (dw4_kernel<true, false, false, false>()), grid * ns, GTHREADS, GEMM_LDS, st, a);
                const int rgrid = (int)((M * (N / kEPT) + kTPB - 1) / kTPB);
                hipLaunchKernelGGL(k_splitk_reduce, rgrid, kTPB, 0, st, a.ws, ns, M, N, a.W, ldw, accumulate);
                return launch_status();
            }
            int tns = 1;
            const int rtail = (K % 128 == 0 && g_gemm_tail && force_ns == 0) ? tail_plan(M, N, K, &tns) : 0;
            if (rtail && workspace && workspace_bytes >= (int64_t)rtail * tns * GB * GB * 4 && (ldw % 8) == 0 && !((uintptr_t)dW & 15)) {
                // full rounds with the whole K, then the last (partial) round split along K, then its slices summed in order
                AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (dw4_kernel<false, false, false, false>()), grid - rtail, GTHREADS, GEMM_LDS, st, a);
                a.ws = (float*)workspace; a.nsplit = tns; a.tile0 = grid - rtail;
                hipLaunchKernelGGL((dw4_kernel<true, false, false, false>()), rtail * tns, GTHREADS, GEMM_LDS, st, a);
                hipLaunchKernelGGL(k_splitk_reduce_tiles, rtail * 32, kTPB, 0, st, a);
                return launch_status();
            }
            if (K % 128) AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (dw4_kernel<false, true, false, false>()), grid, GTHREADS, GEMM_LDS, st, a);
            else AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (dw4_kernel<false, false, false, false>()), grid, GTHREADS, GEMM_LDS, st, a);
            return launch_status();
        }
#ifdef AR_GEMM_EXPERIMENTS
        if (g_gemm_kernel >= 4) {      // timing ablations (tools/gemm_dw_probe.py --ablate); outputs are not a product
            (void)hipFuncSetAttribute((const void*)k_gemm_dw2_abl<1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            (void)hipFuncSetAttribute((const void*)k_gemm_dw2_abl<2>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            (void)hipFuncSetAttribute((const void*)k_gemm_dw2_abl<3>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
            if (g_gemm_kernel == 4) hipLaunchKernelGGL((k_gemm_dw2_abl<1>), grid, GTHREADS, GEMM_LDS, st, a);
            else if (g_gemm_kernel == 5) hipLaunchKernelGGL((k_gemm_dw2_abl<2>), grid, GTHREADS, GEMM_LDS, st, a);
            else hipLaunchKernelGGL((k_gemm_dw2_abl<3>), grid, GTHREADS, GEMM_LDS, st, a);
            return launch_status();
        }
        if (g_gemm_kernel == 1) AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (k_gemm_dw2<true>), grid, GTHREADS, GEMM_LDS, st, a);
        else AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (k_gemm_dw2<false>), grid, GTHREADS, GEMM_LDS, st, a);
        return launch_status();
    }
    if (g_gemm_sem == 1) AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (k_gemm_dw<1>), grid, GTHREADS, GEMM_LDS, st, a);
    else AR_LAUNCH_PROF(AR_PROF_GEMM_DW, M * N, (k_gemm_dw<2>), grid, GTHREADS, GEMM_LDS, st, a);
    return launch_status();
#else
    }
    return AR_ERR_UNSUPPORTED;      // K < 128
#endif
}
