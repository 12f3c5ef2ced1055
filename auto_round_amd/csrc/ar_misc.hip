// ar_misc.hip -- the small kernels around the quant kernels: MSE loss fwd+bwd, best-loss bookkeeping, calibration
// activation gather, INT packer.  All HBM-bound byte/integer work; nothing here is GEMM shaped.
#include <float.h>

#include "ar_common.hpp"

namespace ar {

// ------------------------------------------------------------------------------------------------------------------
// MSE loss forward + backward in one pass over (pred, ref): 2 reads + 1 write of the activation dtype.
// Stage 1: per-workgroup partial sums (fp32 per lane over a strided slice, fp64 across lanes) into the workspace.
// Stage 2: a single workgroup adds the partials in a fixed order -> deterministic loss (the loss feeds the
//          best-iterate decision, so it must not depend on atomics ordering).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMseMaxBlocks = 1024;

template <int ADT>
__global__ __launch_bounds__(kTPB) void k_mse_stage1(const void* __restrict__ pred, const void* __restrict__ ref,
                                                     void* __restrict__ dpred, double* __restrict__ partials, int64_t n,
                                                     float alpha, float gout, const uint8_t* __restrict__ mask,
                                                     int64_t row_len) {
    __shared__ double red[kTPB / kWave];
    const int64_t n_chunks = n / kEPT;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    float acc = 0.f;
    for (int64_t c = (int64_t)blockIdx.x * kTPB + threadIdx.x; c < n_chunks; c += stride) {
        float p[8], r[8], d[8];
        // row_len % 8 == 0 (checked on the host): a chunk never straddles two tokens
        const bool valid = mask == nullptr || mask[(c * kEPT) / row_len] != 0;
        if (valid) {
            unpack8<ADT>(load8_raw<ADT>(pred, c * kEPT), p);
            unpack8<ADT>(load8_raw<ADT>(ref, c * kEPT), r);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float df = p[k] - r[k];
                acc += df * df;
                d[k] = (alpha * df) * gout;   // ATen mse_loss_backward: alpha * (a - b) * grad_output
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = 0.f;
        }
        if (dpred) store8<ADT>(dpred, c * kEPT, d);
    }
    // tail (n not a multiple of 8)
    for (int64_t i = n_chunks * kEPT + (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n; i += stride) {
        const bool valid = mask == nullptr || mask[i / row_len] != 0;
        const float df = valid ? load1<ADT>(pred, i) - load1<ADT>(ref, i) : 0.f;
        acc += df * df;
        if (dpred) store1<ADT>(dpred, i, valid ? (alpha * df) * gout : 0.f);
    }
    double dacc = (double)acc;
    for (int m = kWave >> 1; m > 0; m >>= 1) dacc += __shfl_xor(dacc, m, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = dacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kTPB / kWave; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kTPB) void k_mse_stage2(const double* __restrict__ partials, int n_partials, int64_t n,
                                                     float* __restrict__ loss_out, float* __restrict__ loss_accum,
                                                     float accum_scale) {
    __shared__ double red[kTPB];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += kTPB) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = kTPB / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(red[0] / (double)n);
        if (loss_out) *loss_out = mean;
        if (loss_accum) *loss_accum += mean * accum_scale;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// outlier-suppressed loss: two-level radix select of the k-th largest |pred-ref| (15-bit magnitude key of the 16-bit
// difference: 8 high bits, then 7 low bits inside the selected bin), then the masked MSE pass.  Exactly `topk` elements
// are dropped: every key above the k-th one, plus the lowest-index elements tied with it.  Every workgroup owns one
// contiguous range of chunks so the tie rank is an ordered prefix: per-workgroup level-2 histograms give the number of
// ties in the ranges before it, a block scan orders the ties inside it.
// workspace layout: double partials[kMseMaxBlocks] | u32 hist1[256] | u32 hist2[128] | u32 block_hist2[kMseMaxBlocks][128]
// ------------------------------------------------------------------------------------------------------------------
template <int ADT> __device__ __forceinline__ uint32_t diff_key15(float p, float r) {
    // |pred - ref| evaluated in the activation dtype, magnitude bits only
    if constexpr (ADT == AR_DT_BF16) return f32_to_bf16(p - r) & 0x7fffu;
    else return f32_to_f16(p - r) & 0x7fffu;
}
__device__ __forceinline__ void select_from_top(const uint32_t* hist, int nbins, uint64_t want, int& bin, uint64_t& above) {
    // the bin b with count(bins > b) < want <= count(bins >= b); `above` = count(bins > b)
    uint64_t cum = 0;
    for (int b = nbins - 1; b > 0; --b) {
        const uint64_t c = hist[b];
        if (cum + c >= want) { bin = b; above = cum; return; }
        cum += c;
    }
    bin = 0; above = cum;
}

__global__ void k_zero_u32(uint32_t* __restrict__ p) { p[threadIdx.x] = 0u; }

// The 16-bit differences of a batch fall into a handful of exponent bins, so a single LDS histogram serialises the 64 lanes
// of a wave on a few addresses (measured 95 us for 67 M elements = one atomic per cycle per CU).  kHistCopies privatised
// copies (lane & 15 picks one) cut the conflicts 16-fold; they are summed once at the end.
constexpr int kHistCopies = 16;
template <int ADT, int LEVEL>
__global__ __launch_bounds__(kTPB) void k_outlier_hist(const void* __restrict__ pred, const void* __restrict__ ref,
                                                       int64_t n_chunks, int64_t chunks_per_block,
                                                       uint32_t* __restrict__ hist1, uint32_t* __restrict__ hist2,
                                                       uint32_t* __restrict__ block_hist2, uint64_t topk) {
    constexpr int NB = LEVEL == 1 ? 256 : 128;
    __shared__ uint32_t lh[kHistCopies][NB + 1];          // +1: the copies start on different banks
    __shared__ int sbin;
    for (int i = threadIdx.x; i < kHistCopies * (NB + 1); i += kTPB) (&lh[0][0])[i] = 0;
    if (LEVEL == 2 && threadIdx.x == 0) { int b; uint64_t ab; select_from_top(hist1, 256, topk, b, ab); sbin = b; }
    __syncthreads();
    const int b1 = LEVEL == 2 ? sbin : 0;
    uint32_t* mine = lh[threadIdx.x & (kHistCopies - 1)];
    const int64_t c_lo = (int64_t)blockIdx.x * chunks_per_block;
    const int64_t c_hi = c_lo + chunks_per_block < n_chunks ? c_lo + chunks_per_block : n_chunks;
    for (int64_t c = c_lo + threadIdx.x; c < c_hi; c += kTPB) {
        float p[8], r[8];
        unpack8<ADT>(load8_raw<ADT>(pred, c * kEPT), p);
        unpack8<ADT>(load8_raw<ADT>(ref, c * kEPT), r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t key = diff_key15<ADT>(p[k], r[k]);
            if (LEVEL == 1) atomicAdd(&mine[key >> 7], 1u);
            else if ((int)(key >> 7) == b1) atomicAdd(&mine[key & 127u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += kTPB) {
        uint32_t t = 0;
#pragma unroll
        for (int c = 0; c < kHistCopies; ++c) t += lh[c][i];
        if (LEVEL == 1) {
            if (t) atomicAdd(&hist1[i], t);
        } else {
            block_hist2[(int64_t)blockIdx.x * 128 + i] = t;
            if (t) atomicAdd(&hist2[i], t);
        }
    }
}

template <int ADT>
__global__ __launch_bounds__(kTPB) void k_outlier_mse(const void* __restrict__ pred, const void* __restrict__ ref,
                                                      void* __restrict__ dpred, double* __restrict__ partials,
                                                      int64_t n_chunks, int64_t chunks_per_block, float alpha, float gout,
                                                      const uint8_t* __restrict__ mask, int64_t row_len,
                                                      const uint32_t* __restrict__ hist1, const uint32_t* __restrict__ hist2,
                                                      const uint32_t* __restrict__ block_hist2, uint64_t topk) {
    __shared__ double red[kTPB / kWave];
    __shared__ uint32_t sthr, sneed, sprefix;
    __shared__ uint32_t wtot[2][kTPB / kWave];
    if (threadIdx.x == 0) {
        int b1, b2; uint64_t ab1, ab2;
        select_from_top(hist1, 256, topk, b1, ab1);
        select_from_top(hist2, 128, topk - ab1, b2, ab2);
        sthr = ((uint32_t)b1 << 7) | (uint32_t)b2;        // the k-th largest key
        sneed = (uint32_t)(topk - ab1 - ab2);               // how many elements tied with it are dropped too (>= 1)
        sprefix = 0;
    }
    __syncthreads();
    const uint32_t thr = sthr, need = sneed;
    {   // ties that live in the ranges of the workgroups before this one
        uint32_t t = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += kTPB) t += block_hist2[(int64_t)b * 128 + (thr & 127u)];
        for (int m = kWave >> 1; m > 0; m >>= 1) t += __shfl_xor(t, m, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0 && t) atomicAdd(&sprefix, t);
    }
    __syncthreads();
    uint32_t running = sprefix;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t c_lo = (int64_t)blockIdx.x * chunks_per_block;
    const int64_t c_hi = c_lo + chunks_per_block < n_chunks ? c_lo + chunks_per_block : n_chunks;
    const int64_t iters = (chunks_per_block + kTPB - 1) / kTPB;
    float acc = 0.f;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t c = c_lo + it * kTPB + threadIdx.x;
        const bool live = c < c_hi;
        float p[8], r[8], d[8];
        uint32_t key[8];
        uint32_t tc = 0;
        if (live) {
            unpack8<ADT>(load8_raw<ADT>(pred, c * kEPT), p);
            unpack8<ADT>(load8_raw<ADT>(ref, c * kEPT), r);
#pragma unroll
            for (int k = 0; k < 8; ++k) { key[k] = diff_key15<ADT>(p[k], r[k]); tc += key[k] == thr; }
        }
        uint32_t incl = tc;                                  // ordered (index-order) scan of the tie counts
        for (int m = 1; m < kWave; m <<= 1) { const uint32_t o = __shfl_up(incl, m, kWave); if (lane >= m) incl += o; }
        if (lane == kWave - 1) wtot[it & 1][wave] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kTPB / kWave; ++w) { const uint32_t t = wtot[it & 1][w]; total += t; woff += w < wave ? t : 0u; }
        uint32_t rank = running + woff + incl - tc;
        running += total;
        if (live) {
            const bool valid = mask == nullptr || mask[(c * kEPT) / row_len] != 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                bool drop = key[k] > thr;
                if (key[k] == thr) { drop = rank < need; ++rank; }
                const bool keep = valid && !drop;
                const float df = keep ? p[k] - r[k] : 0.f;
                acc += df * df;
                d[k] = keep ? (alpha * df) * gout : 0.f;
            }
            if (dpred) store8<ADT>(dpred, c * kEPT, d);
        }
    }
    double dacc = (double)acc;
    for (int m = kWave >> 1; m > 0; m >>= 1) dacc += __shfl_xor(dacc, m, kWave);
    if (lane == 0) red[wave] = dacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kTPB / kWave; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// best-loss bookkeeping (one lane)
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_best_loss_update(float* total_loss, float* state, int32_t* istate, int32_t iter, int32_t* iter_dev, float* loss_hist) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (iter_dev) iter = *iter_dev;               // hipGraph replays: the iteration number lives on the device
    const float tl = *total_loss;
    if (iter == 0) state[1] = tl;
    state[2] = tl;
    if (tl < state[0]) {
        state[0] = tl;
        istate[0] = 1;
        istate[1] = iter;
        istate[2] += 1;
    } else {
        istate[0] = 0;
    }
    if (loss_hist) loss_hist[iter] = tl;
    if (iter_dev) *iter_dev = iter + 1;
    *total_loss = 0.f;
}

// start of a tuning iteration whose host-side values come from device tables (one captured hipGraph replayed `iters` times):
// this iteration's minibatch indices and learning rates are copied to the fixed addresses the captured kernels read
__global__ void k_iter_begin(const int32_t* iter_dev, const int64_t* sched, int batch, int64_t* cur_idx, const float* lr_table,
                             int n_lr, int iters, float* lr_out) {
    const int it = *iter_dev;
    if (it < 0 || it >= iters) return;
    for (int j = threadIdx.x; j < batch; j += kWave) cur_idx[j] = sched[(int64_t)it * batch + j];
    for (int j = threadIdx.x; j < n_lr; j += kWave) lr_out[j] = lr_table[(int64_t)j * iters + it];
}

// ------------------------------------------------------------------------------------------------------------------
// row gather: dst[j,:] = src[idx[j],:], 16 bytes per lane per step
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTPB) void k_gather_rows(const uint4* __restrict__ src, const int64_t* __restrict__ idx,
                                                      uint4* __restrict__ dst, int64_t row_vecs) {
    const int64_t j = blockIdx.y;
    const int64_t s = idx[j];
    const uint4* sp = src + s * row_vecs;
    uint4* dp = dst + j * row_vecs;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    for (int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x; i < row_vecs; i += stride) dp[i] = sp[i];
}

// ------------------------------------------------------------------------------------------------------------------
// INT packer.  One lane owns one (32-input block, output column) pair: it re-derives 32 integers from the baked
// weight exactly as the reference (fp32 division, rint, + zp) and emits `bits` words with the reference's additive
// arithmetic (see oracle/ar_oracle.c pack32 for the rationale).  Lanes are mapped so that stores to
// qweight[row, o] are coalesced along o; the 64-byte reads of Wq[o, 32 inputs] are full sectors.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack32_words(const int32_t (&v)[32], int bits, uint32_t (&w)[8]) {
    if (bits != 3) {
        const int P = 32 / bits;
        for (int k = 0; k < bits; ++k) {
            uint32_t acc = 0;
            for (int j = 0; j < P; ++j) acc += (uint32_t)v[k * P + j] << (bits * j);
            w[k] = acc;
        }
        return;
    }
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    for (int j = 0; j < 10; ++j) a0 += (uint32_t)v[j] << (3 * j);
    for (int j = 0; j < 10; ++j) a1 += (uint32_t)v[11 + j] << (3 * j + 1);
    for (int j = 0; j < 10; ++j) a2 += (uint32_t)v[22 + j] << (3 * j + 2);
    w[0] = a0 | ((uint32_t)v[10] << 30);
    w[1] = (((uint32_t)(v[10] >> 2)) & 1u) | a1 | ((uint32_t)v[21] << 31);
    w[2] = (((uint32_t)(v[21] >> 1)) & 3u) | a2;
}

template <int WDT>
__global__ __launch_bounds__(kTPB) void k_pack_qweight(const void* __restrict__ Wq, const void* __restrict__ scale,
                                                       const float* __restrict__ zp_tensor, float zp_scalar,
                                                       int64_t out_f, int64_t in_f, int gs, int bits, int s_dt,
                                                       int32_t* __restrict__ qweight) {
    const int64_t o = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    const int64_t blk = blockIdx.y;
    if (o >= out_f) return;
    const int64_t n_groups = (in_f + gs - 1) / gs;
    int32_t v[32];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float w[8];
        unpack8<WDT>(load8_raw<WDT>(Wq, o * in_f + blk * 32 + c * 8), w);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = blk * 32 + c * 8 + k;
            const int64_t gi = o * n_groups + i / gs;
            const float s = load1_rt(s_dt, scale, gi);
            const float z = zp_tensor ? zp_tensor[gi] : zp_scalar;
            v[c * 8 + k] = (int32_t)__builtin_rintf(w[k] / s + z);
        }
    }
    uint32_t words[8];
    pack32_words(v, bits, words);
    for (int k = 0; k < bits; ++k) qweight[(blk * bits + k) * out_f + o] = (int32_t)words[k];
}

__global__ __launch_bounds__(kTPB) void k_pack_qzeros_scales(const void* __restrict__ scale,
                                                             const float* __restrict__ zp_tensor, float zp_scalar,
                                                             int64_t out_f, int64_t n_groups, int bits, int s_dt,
                                                             int zp_off, int32_t* __restrict__ qzeros,
                                                             uint16_t* __restrict__ scales_t) {
    const int64_t n_blk = out_f / 32;
    const int64_t zcols = n_blk * bits;
    const int64_t total_z = n_groups * n_blk;
    const int64_t total_s = n_groups * out_f;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    for (int64_t t = (int64_t)blockIdx.x * kTPB + threadIdx.x; t < total_z; t += stride) {
        const int64_t ig = t / n_blk, blk = t % n_blk;
        int32_t v[32];
        for (int j = 0; j < 32; ++j) {
            const int64_t o = blk * 32 + j;
            const float z = zp_tensor ? zp_tensor[o * n_groups + ig] : zp_scalar;
            v[j] = (int32_t)(z - (float)zp_off);   // `zeros -= 1` in float, then .to(int32)
        }
        uint32_t words[8];
        pack32_words(v, bits, words);
        for (int k = 0; k < bits; ++k) qzeros[ig * zcols + blk * bits + k] = (int32_t)words[k];
    }
    for (int64_t t = (int64_t)blockIdx.x * kTPB + threadIdx.x; t < total_s; t += stride) {
        const int64_t ig = t / out_f, o = t % out_f;
        scales_t[t] = (uint16_t)f32_to_f16(load1_rt(s_dt, scale, o * n_groups + ig));
    }
}

// AWQ GEMM packer: one lane owns (input row i, word w): the eight output channels 8w..8w+7 of input i.
// Lanes run along i, so the eight strided reads Wq[(8w+j)*in + i] are coalesced across the wave.
template <int WDT>
__global__ __launch_bounds__(kTPB) void k_pack_awq(const void* __restrict__ Wq, const void* __restrict__ scale,
                                                   const float* __restrict__ zp_tensor, float zp_scalar, int64_t out_f,
                                                   int64_t in_f, int gs, int s_dt, int32_t* __restrict__ qweight) {
    const int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x;
    const int64_t w = blockIdx.y;
    if (i >= in_f) return;
    const int64_t n_groups = in_f / gs;
    const int64_t ig = i / gs;
    const int order[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t o = w * 8 + j;
        const float s = load1_rt(s_dt, scale, o * n_groups + ig);
        const float z = zp_tensor ? zp_tensor[o * n_groups + ig] : zp_scalar;
        const int32_t v = (int32_t)__builtin_rintf(load1<WDT>(Wq, o * in_f + i) / s + z);
        acc += (uint32_t)v << (4 * order[j]);
    }
    qweight[i * (out_f / 8) + w] = (int32_t)acc;
}

__global__ __launch_bounds__(kTPB) void k_pack_awq_zeros_scales(const void* __restrict__ scale,
                                                                const float* __restrict__ zp_tensor, float zp_scalar,
                                                                int64_t out_f, int64_t n_groups, int s_dt,
                                                                int32_t* __restrict__ qzeros, uint16_t* __restrict__ scales_t) {
    const int64_t words = out_f / 8;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    const int order[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    for (int64_t t = (int64_t)blockIdx.x * kTPB + threadIdx.x; t < n_groups * words; t += stride) {
        const int64_t ig = t / words, w = t % words;
        uint32_t acc = 0;
        for (int j = 0; j < 8; ++j) {
            const int64_t o = w * 8 + j;
            const float z = zp_tensor ? zp_tensor[o * n_groups + ig] : zp_scalar;
            acc += (uint32_t)(int32_t)z << (4 * order[j]);
        }
        qzeros[t] = (int32_t)acc;
    }
    for (int64_t t = (int64_t)blockIdx.x * kTPB + threadIdx.x; t < n_groups * out_f; t += stride) {
        const int64_t ig = t / out_f, o = t % out_f;
        scales_t[t] = (uint16_t)f32_to_f16(load1_rt(s_dt, scale, o * n_groups + ig));
    }
}

}  // namespace ar

using namespace ar;

extern "C" int ar_pack_awq(const void* Wq, const void* scale, const float* zp_tensor, float zp_scalar, int64_t out_f,
                           int64_t in_f, int gs, int w_dt, int s_dt, int32_t* qweight, int32_t* qzeros, uint16_t* scales_t,
                           ar_stream_t stream) {
    if (gs <= 0 || in_f % gs || out_f % 8 || out_f <= 0 || in_f <= 0) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((in_f + kTPB - 1) / kTPB), (unsigned)(out_f / 8));
    switch (w_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_pack_awq<AR_DT_BF16>, grid, kTPB, 0, st, Wq, scale, zp_tensor, zp_scalar, out_f, in_f, gs, s_dt, qweight); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_pack_awq<AR_DT_F16>, grid, kTPB, 0, st, Wq, scale, zp_tensor, zp_scalar, out_f, in_f, gs, s_dt, qweight); break;
        case AR_DT_F32: hipLaunchKernelGGL(k_pack_awq<AR_DT_F32>, grid, kTPB, 0, st, Wq, scale, zp_tensor, zp_scalar, out_f, in_f, gs, s_dt, qweight); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    int rc = launch_status();
    if (rc) return rc;
    const int64_t n_groups = in_f / gs;
    int g2 = (int)((n_groups * out_f + kTPB - 1) / kTPB);
    if (g2 > 2048) g2 = 2048;
    hipLaunchKernelGGL(k_pack_awq_zeros_scales, g2, kTPB, 0, st, scale, zp_tensor, zp_scalar, out_f, n_groups, s_dt, qzeros, scales_t);
    return launch_status();
}

extern "C" int ar_abi_version(void) { return AR_ABI_VERSION; }

extern "C" const char* ar_error_string(int code) {
    if (code == AR_OK) return "ok";
    if (code == AR_ERR_UNSUPPORTED) return "argument combination not supported by this build";
    return hipGetErrorString((hipError_t)code);
}

extern "C" int64_t ar_mse_workspace_bytes(void) { return (int64_t)kMseMaxBlocks * sizeof(double); }

extern "C" int ar_mse_loss_fwd_bwd(const void* pred, const void* ref, void* dpred, float* loss_out, float* loss_accum,
                                   float accum_scale, int64_t n, int act_dt, float grad_scale, const uint8_t* token_mask,
                                   int64_t row_len, void* workspace, ar_stream_t stream) {
    if (n <= 0 || !workspace) return AR_ERR_UNSUPPORTED;
    if (token_mask && (row_len <= 0 || row_len % kEPT || n % row_len)) return AR_ERR_UNSUPPORTED;
    if (!token_mask) row_len = 1;
    hipStream_t st = (hipStream_t)stream;
    int64_t want = (n / kEPT + kTPB - 1) / kTPB;
    const int grid = (int)(want < 1 ? 1 : (want > kMseMaxBlocks ? kMseMaxBlocks : want));
    const float alpha = (float)(2.0 / (double)n);
    double* partials = (double*)workspace;
    switch (act_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_mse_stage1<AR_DT_BF16>, grid, kTPB, 0, st, pred, ref, dpred, partials, n, alpha, grad_scale, token_mask, row_len); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_mse_stage1<AR_DT_F16>, grid, kTPB, 0, st, pred, ref, dpred, partials, n, alpha, grad_scale, token_mask, row_len); break;
        case AR_DT_F32: hipLaunchKernelGGL(k_mse_stage1<AR_DT_F32>, grid, kTPB, 0, st, pred, ref, dpred, partials, n, alpha, grad_scale, token_mask, row_len); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_mse_stage2, 1, kTPB, 0, st, partials, grid, n, loss_out, loss_accum, accum_scale);
    return launch_status();
}

extern "C" int64_t ar_outlier_loss_workspace_bytes(void) {
    return (int64_t)kMseMaxBlocks * sizeof(double) + (256 + 128 + (int64_t)kMseMaxBlocks * 128) * sizeof(uint32_t);
}

extern "C" int ar_outlier_mse_loss_fwd_bwd(const void* pred, const void* ref, void* dpred, float* loss_out, float* loss_accum,
                                           float accum_scale, int64_t n, int act_dt, float grad_scale,
                                           const uint8_t* token_mask, int64_t row_len, int64_t topk, void* workspace,
                                           ar_stream_t stream) {
    if (n <= 0 || n % kEPT || !workspace || topk < 1 || topk > n) return AR_ERR_UNSUPPORTED;
    if (act_dt != AR_DT_BF16 && act_dt != AR_DT_F16) return AR_ERR_UNSUPPORTED;
    if (token_mask && (row_len <= 0 || row_len % kEPT || n % row_len)) return AR_ERR_UNSUPPORTED;
    if (!token_mask) row_len = 1;
    hipStream_t st = (hipStream_t)stream;
    double* partials = (double*)workspace;
    uint32_t* hist1 = (uint32_t*)(partials + kMseMaxBlocks);
    uint32_t* hist2 = hist1 + 256;
    uint32_t* bh2 = hist2 + 128;
    // a kernel rather than hipMemsetAsync: memset nodes of a captured hipGraph did not re-execute on replay with ROCm 7.2
    // (tests/test_gpu_kernels.py::test_kernels_are_hip_graph_capturable...), a kernel node always does
    hipLaunchKernelGGL(k_zero_u32, 1, 256 + 128, 0, st, hist1);
    int rc = 0;
    const int64_t n_chunks = n / kEPT;
    int64_t want = (n_chunks + kTPB - 1) / kTPB;
    int grid = (int)(want < 1 ? 1 : (want > kMseMaxBlocks ? kMseMaxBlocks : want));
    const int64_t cpb = (n_chunks + grid - 1) / grid;
    grid = (int)((n_chunks + cpb - 1) / cpb);
    const float alpha = (float)(2.0 / (double)n);
    const uint64_t k = (uint64_t)topk;
    if (act_dt == AR_DT_BF16) {
        hipLaunchKernelGGL((k_outlier_hist<AR_DT_BF16, 1>), grid, kTPB, 0, st, pred, ref, n_chunks, cpb, hist1, hist2, bh2, k);
        hipLaunchKernelGGL((k_outlier_hist<AR_DT_BF16, 2>), grid, kTPB, 0, st, pred, ref, n_chunks, cpb, hist1, hist2, bh2, k);
        hipLaunchKernelGGL(k_outlier_mse<AR_DT_BF16>, grid, kTPB, 0, st, pred, ref, dpred, partials, n_chunks, cpb, alpha,
                           grad_scale, token_mask, row_len, hist1, hist2, bh2, k);
    } else {
        hipLaunchKernelGGL((k_outlier_hist<AR_DT_F16, 1>), grid, kTPB, 0, st, pred, ref, n_chunks, cpb, hist1, hist2, bh2, k);
        hipLaunchKernelGGL((k_outlier_hist<AR_DT_F16, 2>), grid, kTPB, 0, st, pred, ref, n_chunks, cpb, hist1, hist2, bh2, k);
        hipLaunchKernelGGL(k_outlier_mse<AR_DT_F16>, grid, kTPB, 0, st, pred, ref, dpred, partials, n_chunks, cpb, alpha,
                           grad_scale, token_mask, row_len, hist1, hist2, bh2, k);
    }
    rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_mse_stage2, 1, kTPB, 0, st, partials, grid, n, loss_out, loss_accum, accum_scale);
    return launch_status();
}

extern "C" int ar_best_loss_update(float* total_loss, float* state, int32_t* istate, int32_t iter, int32_t* iter_dev, float* loss_hist,
                                   ar_stream_t stream) {
    hipLaunchKernelGGL(k_best_loss_update, 1, kWave, 0, (hipStream_t)stream, total_loss, state, istate, iter, iter_dev, loss_hist);
    return launch_status();
}

extern "C" int ar_iter_begin(const int32_t* iter_dev, const int64_t* sched, int batch, int64_t* cur_idx, const float* lr_table, int n_lr,
                             int iters, float* lr_out, ar_stream_t stream) {
    if (!iter_dev || batch < 0 || n_lr < 0 || iters <= 0 || (batch > 0 && (!sched || !cur_idx)) || (n_lr > 0 && (!lr_table || !lr_out)))
        return AR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_iter_begin, 1, kWave, 0, (hipStream_t)stream, iter_dev, sched, batch, cur_idx, lr_table, n_lr, iters, lr_out);
    return launch_status();
}

extern "C" int ar_gather_rows(const void* src, const int64_t* idx_dev, void* dst, int64_t n_idx, int64_t row_bytes,
                              ar_stream_t stream) {
    if (row_bytes % 16 || n_idx < 0 || n_idx > 65535) return AR_ERR_UNSUPPORTED;
    if (n_idx == 0 || row_bytes == 0) return AR_OK;
    const int64_t row_vecs = row_bytes / 16;
    int64_t gx = (row_vecs + kTPB * 4 - 1) / (kTPB * 4);
    if (gx > 512) gx = 512;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)gx, (unsigned)n_idx), kTPB, 0, (hipStream_t)stream,
                       (const uint4*)src, idx_dev, (uint4*)dst, row_vecs);
    return launch_status();
}

extern "C" int ar_pack_int(const void* Wq, const void* scale, const float* zp_tensor, float zp_scalar, int64_t out_f,
                           int64_t in_f, int gs, int bits, int w_dt, int s_dt, int zp_off, int32_t* qweight,
                           int32_t* qzeros, uint16_t* scales_t, ar_stream_t stream) {
    if (!(bits == 2 || bits == 3 || bits == 4 || bits == 8) || in_f % 32 || gs <= 0 || out_f <= 0) return AR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_groups = (in_f + gs - 1) / gs;
    dim3 grid((unsigned)((out_f + kTPB - 1) / kTPB), (unsigned)(in_f / 32));
    switch (w_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_pack_qweight<AR_DT_BF16>, grid, kTPB, 0, st, Wq, scale, zp_tensor, zp_scalar, out_f, in_f, gs, bits, s_dt, qweight); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_pack_qweight<AR_DT_F16>, grid, kTPB, 0, st, Wq, scale, zp_tensor, zp_scalar, out_f, in_f, gs, bits, s_dt, qweight); break;
        case AR_DT_F32: hipLaunchKernelGGL(k_pack_qweight<AR_DT_F32>, grid, kTPB, 0, st, Wq, scale, zp_tensor, zp_scalar, out_f, in_f, gs, bits, s_dt, qweight); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    int rc = launch_status();
    if (rc) return rc;
    int64_t work = n_groups * out_f;
    int g2 = (int)((work + kTPB - 1) / kTPB);
    if (g2 > 2048) g2 = 2048;
    hipLaunchKernelGGL(k_pack_qzeros_scales, g2, kTPB, 0, st, scale, zp_tensor, zp_scalar, out_f, n_groups, bits, s_dt,
                       zp_off, qzeros, scales_t);
    return launch_status();
}
