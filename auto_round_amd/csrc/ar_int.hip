// ar_int.hip -- INT (W2/W3/W4/W8, sym + asym) fake-quant kernels for gfx950:
//   k_group_minmax : per-group min/max (wave shuffle reductions)
//   k_int_fwd      : K1  fake-quant forward                    8 B/elem + 12 B/group   (HBM bound)
//   k_int_bwd      : K2(+K3) backward [+ sign-SGD + best-param snapshot + next forward], 12 B/elem + 8 B/group
//
// Two implementations of the hot pair are kept and selectable at build time (AR_INT_FLAT):
//  * flat (default, faster): every lane issues its 16-byte streaming loads up front, derives its group's (scale, zp)
//    itself from four broadcast-loaded group parameters, and the gs/8 lanes of a group meet only in the shuffle
//    butterfly; no LDS, no barrier (k_int_fwd_flat / k_int_bwd_flat).
//  * LDS-tiled (the first design): the weight is a flat array of groups; a workgroup (256 lanes = 4 waves) owns a
//    tile of up to 8192 consecutive elements (whole groups).  Per tile:
//   (A) up to 256 "group lanes" load the 4 per-group parameters and                      [12 B/group]
//   (B) every lane issues all of its 16-byte streaming loads (W, V[, dWq]) up front,     [the HBM stream]
//   (C) the group lanes turn the parameters into (scale, zp) -- fp16 cast, threshold clamp, true IEEE division,
//       all the data-dependent branching -- and stage them in LDS while the streaming loads are in flight,
//   (D) one barrier, then every lane consumes its registers against the LDS-staged scales and stores 16 B.
// The backward additionally reduces 2-4 per-group sums with a DPP/bpermute butterfly over the gs/8 lanes that share
// a group, hands them to the group lane through LDS, and the group lane applies the sign step to min/max scale.
// There is no inter-workgroup reuse, so no XCD-aware remap is needed: consecutive tiles simply round-robin over the
// 8 XCDs and stream straight from HBM.
#include "ar_common.hpp"

namespace ar {

// ------------------------------------------------------------------------------------------------------------------
// group min/max
// ------------------------------------------------------------------------------------------------------------------
template <int WDT>
__global__ __launch_bounds__(kTPB) void k_group_minmax(const void* __restrict__ W, void* __restrict__ wmin,
                                                       void* __restrict__ wmax, float* __restrict__ absmax,
                                                       float* __restrict__ tensor_absmax, int64_t n_groups, int cpg) {
    // one group per `cpg` lanes when cpg <= 64, else one group per wave looping over the group
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave_global = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / kWave;
    const int64_t n_waves = (int64_t)gridDim.x * kTPB / kWave;
    float tmax = 0.f;
    const bool lane_groups = cpg > 0;   // host passes -cpg to select the wave-per-group form (long or non-pow2 groups)
    if (!lane_groups) cpg = -cpg;
    if (lane_groups) {
        const int gpw = kWave / cpg;  // groups per wave pass
        for (int64_t gb = wave_global * gpw; gb < n_groups; gb += n_waves * gpw) {
            const int64_t g = gb + lane / cpg;
            float lo = INFINITY, hi = -INFINITY;
            if (g < n_groups) {
                float v[8];
                unpack8<WDT>(load8_raw<WDT>(W, (g * cpg + (lane % cpg)) * kEPT), v);
#pragma unroll
                for (int k = 0; k < 8; ++k) { lo = fminf(lo, v[k]); hi = fmaxf(hi, v[k]); }
            }
            lo = lanes_min(lo, cpg);
            hi = lanes_max(hi, cpg);
            if (g < n_groups && (lane % cpg) == 0) {
                if (wmin) store1<WDT>(wmin, g, lo > 0.f ? 0.f : lo);   // torch.clamp(max=0): a -0.0 extremum stays -0.0
                if (wmax) store1<WDT>(wmax, g, hi < 0.f ? 0.f : hi);
                const float am = fmaxf(-lo, hi);
                if (absmax) absmax[g] = am;
                tmax = fmaxf(tmax, am);
            }
        }
    } else {
        for (int64_t g = wave_global; g < n_groups; g += n_waves) {
            float lo = INFINITY, hi = -INFINITY;
            for (int c = lane; c < cpg; c += kWave) {
                float v[8];
                unpack8<WDT>(load8_raw<WDT>(W, (g * cpg + c) * kEPT), v);
#pragma unroll
                for (int k = 0; k < 8; ++k) { lo = fminf(lo, v[k]); hi = fmaxf(hi, v[k]); }
            }
            lo = lanes_min(lo, kWave);
            hi = lanes_max(hi, kWave);
            if (lane == 0) {
                if (wmin) store1<WDT>(wmin, g, lo > 0.f ? 0.f : lo);   // torch.clamp(max=0): a -0.0 extremum stays -0.0
                if (wmax) store1<WDT>(wmax, g, hi < 0.f ? 0.f : hi);
                const float am = fmaxf(-lo, hi);
                if (absmax) absmax[g] = am;
                tmax = fmaxf(tmax, am);
            }
        }
    }
    if (tensor_absmax) {    // wave -> workgroup -> ONE atomic per workgroup (the grid is capped for this case on the host)
        __shared__ float wmaxs[kTPB / kWave];
        tmax = lanes_max(tmax, kWave);
        if (lane == 0) wmaxs[threadIdx.x / kWave] = tmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = wmaxs[0];
            for (int w = 1; w < kTPB / kWave; ++w) m = fmaxf(m, wmaxs[w]);
            // non-negative floats order like their bit patterns
            if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(tensor_absmax), __float_as_uint(m));
        }
    }
}

// Lane-group form with the group width as a template parameter (DPP butterflies) and U chunks per lane in flight: U
// independent loads issued before the first use -- the one-load-per-wave form above is latency-bound (3.5 TB/s on 2 B/element).
template <int WDT, int CPG, int U>
__global__ __launch_bounds__(kTPB) void k_group_minmax_lg(const void* __restrict__ W, void* __restrict__ wmin,
                                                          void* __restrict__ wmax, float* __restrict__ absmax,
                                                          float* __restrict__ tensor_absmax, int64_t n_groups) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t total = n_groups * CPG;                       // chunks
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    float tmax = 0.f;
    for (int64_t c0 = (int64_t)blockIdx.x * kTPB + threadIdx.x; c0 - threadIdx.x < total; c0 += stride * U) {
        Raw8<WDT> r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + u * stride;
            r[u] = load8_raw<WDT>(W, (c < total ? c : 0) * kEPT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + u * stride;
            float v[8];
            unpack8<WDT>(r[u], v);
            float lo = v[0], hi = v[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) { lo = __builtin_fminf(lo, v[k]); hi = __builtin_fmaxf(hi, v[k]); }
            if (c >= total) { lo = INFINITY; hi = -INFINITY; }
            lo = group_min<CPG>(lo);
            hi = group_max<CPG>(hi);
            if (c < total && (lane & (CPG - 1)) == 0) {
                const int64_t g = c / CPG;
                if (wmin) store1<WDT>(wmin, g, lo > 0.f ? 0.f : lo);   // torch.clamp(max=0): a -0.0 extremum stays -0.0
                if (wmax) store1<WDT>(wmax, g, hi < 0.f ? 0.f : hi);
                const float am = fmaxf(-lo, hi);
                if (absmax) absmax[g] = am;
                tmax = fmaxf(tmax, am);
            }
        }
    }
    if (tensor_absmax) {    // wave -> workgroup -> ONE atomic per workgroup (the grid is capped for this case on the host)
        __shared__ float wmaxs[kTPB / kWave];
        tmax = lanes_max(tmax, kWave);
        if (lane == 0) wmaxs[threadIdx.x / kWave] = tmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = wmaxs[0];
            for (int w = 1; w < kTPB / kWave; ++w) m = fmaxf(m, wmaxs[w]);
            if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(tensor_absmax), __float_as_uint(m));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
struct FwdArgs {
    const void* W; const float* V; const void* wmin; const void* wmax; const float* min_s; const float* max_s;
    void* Wq; void* scale_out; float* zp_out;
    int64_t n_groups;
    int cpg, cpg_shift;   // chunks (of 8 elements) per group; shift = log2(cpg) or -1
    int x_dt;             // dtype in which W/scale is evaluated (torch promotion of w_dt and s_dt)
    float qlo, qhi;       // clamp bounds of the integer grid (sym: -maxq..maxq-1 with zp 0; asym: 0..maxq with zp)
    IntCfg cfg;
};

__device__ __forceinline__ int group_of_chunk(int cl, int cpg, int shift) { return shift >= 0 ? (cl >> shift) : (cl / cpg); }

// one chunk of the forward: o = s * (clamp(rint(w/s + v) + zp, qlo, qhi) - zp)
template <int XR, bool FAST>
__device__ __forceinline__ void qdq8_impl(const float (&w)[8], const float (&v)[8], float s, float y, float zp, float qlo,
                                          float qhi, float (&o)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // correctly rounded w/s: exact 3-instruction form (ar_common.hpp) or the IEEE division; never rcp*mul
        float x = round_to<XR>(FAST ? div_fast(w[k], s, y) : w[k] / s);
        float r = round_ste_value(x + v[k]);
        float qq = clamp3(r + zp, qlo, qhi) - zp;
        o[k] = s * qq;
    }
}
template <int XR>
__device__ __forceinline__ void qdq8(const float (&w)[8], const float (&v)[8], float s, float zp, float qlo, float qhi,
                                     float (&o)[8]) {
#if AR_FASTDIV
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) ok = ok && div_fast_ok(w[k]);
    if (__all(ok)) {                       // wave-uniform
        qdq8_impl<XR, true>(w, v, s, 1.0f / s, zp, qlo, qhi, o);
        return;
    }
#endif
    qdq8_impl<XR, false>(w, v, s, 0.f, zp, qlo, qhi, o);
}

#ifndef AR_FWD_MINW
#define AR_FWD_MINW 1
#endif
#ifndef AR_BWD_MINW
#define AR_BWD_MINW 1
#endif
template <int WDT, int XR, int UNROLL>
__global__ __launch_bounds__(kTPB, AR_FWD_MINW) void k_int_fwd(const FwdArgs a) {
    __shared__ float2 sg[2][kTPB];
    const int tid = threadIdx.x;
    const int cpg = a.cpg, shift = a.cpg_shift;
    const int u_eff = cpg < UNROLL ? cpg : UNROLL;
    const int tile_chunks = kTPB * u_eff;
    const int tile_groups = tile_chunks / cpg > 0 ? tile_chunks / cpg : 1;
    const int64_t total_chunks = a.n_groups * cpg;
    const int64_t n_tiles = (a.n_groups + tile_groups - 1) / tile_groups;
    int par = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
        const int64_t c0 = tile * (int64_t)tile_groups * cpg;
        // (A) group parameters
        const int64_t g = tile * tile_groups + tid;
        const bool has_g = tid < tile_groups && g < a.n_groups;
        float wmn = 0.f, wmx = 0.f, ms = 1.f, Ms = 1.f;
        if (has_g) {
            wmn = load1<WDT>(a.wmin, g);
            wmx = load1<WDT>(a.wmax, g);
            if (a.min_s) ms = a.min_s[g];
            if (a.max_s) Ms = a.max_s[g];
        }
        // (B) streaming loads, all issued before anything waits on them
        Raw8<WDT> wr[UNROLL];
        F8 vr[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t c = c0 + u * kTPB + tid;
            ok[u] = (u < u_eff) && (u * kTPB + tid < tile_groups * cpg) && (c < total_chunks);
            if (ok[u]) {
                wr[u] = load8_raw<WDT>(a.W, c * kEPT);
                if (a.V) vr[u] = load8_f32(a.V, c * kEPT);
            }
        }
        // (C) scales -> LDS
        if (has_g) {
            GroupQ q;
            group_scale(a.cfg, wmn, wmx, ms, Ms, q);
            sg[par][tid] = make_float2(q.s, a.cfg.sym ? 0.f : q.zp);
            if (a.scale_out) store1_rt(a.cfg.s_dt, a.scale_out, g, q.s);
            if (a.zp_out) a.zp_out[g] = q.zp;
        }
        __syncthreads();
        // (D) consume
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (!ok[u]) continue;
            const float2 sz = sg[par][group_of_chunk(u * kTPB + tid, cpg, shift)];
            float w[8], v[8], o[8];
            unpack8<WDT>(wr[u], w);
            if (a.V) unpack_f8(vr[u], v);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            qdq8<XR>(w, v, sz.x, sz.y, a.qlo, a.qhi, o);
            store8<WDT>(a.Wq, (c0 + u * kTPB + tid) * kEPT, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward (+ sign-SGD, + snapshot, + next forward)
// ------------------------------------------------------------------------------------------------------------------
struct BwdArgs {
    const void* dWq; const void* W; float* V; const void* wmin; const void* wmax; float* min_s; float* max_s;
    float* dV; float* dmin; float* dmax;            // unfused outputs (optional)
    const float* lr_v; const float* lr_mm;          // device learning rates (NULL => no update)
    const int32_t* snap; float* best_V; float* best_min; float* best_max;
    void* Wq_next;
    int64_t n_groups;
    int cpg, cpg_shift, x_dt, tune_minmax;
    float qlo, qhi;
    IntCfg cfg;
};

struct Sums { float c1, c2, e, dy; };

// The four group sums are accumulated in the association torch's reduction kernel (ATen/native/cuda/Reduce.cuh) uses for a
// contiguous row of fp32 values, so that d min_scale / d max_scale carry the bits autograd's `sum_to_size` produces on this GPU
// (probed inside the reference's own loop at real shapes: tests/t3_baseline_shapes.py -> profiles/r03_t3_baseline_shapes.json):
//   rows of 128 and more  -- vectorised input: every "thread" owns four consecutive elements, added left to right, then the
//                            threads are combined neighbours first; a lane's eight elements are two such runs (TREE = false);
//   rows of 32 / 64       -- one element per thread, combined neighbours first: a pure pairwise tree (TREE = true).
// Either way the lanes of a group finish with lanes_sum_torch (neighbours first).
template <int XR, bool FAST, bool TREE>
__device__ __forceinline__ void bwd8_impl(const float (&g)[8], const float (&w)[8], const float (&v)[8], float s, float y,
                                          float zp, float qlo, float qhi, float (&dy)[8], Sums& acc) {
    float t1[8], t2[8], te[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = round_to<XR>(FAST ? div_fast(w[k], s, y) : w[k] / s);
        const float r = round_ste_value(x + v[k]);
        const float t = r + zp;
        const bool inside = (t >= qlo) && (t <= qhi);
        const float qq = clamp3(t, qlo, qhi) - zp;
        const float e = g[k] * s;                         // MulBackward, `other` side
        dy[k] = inside ? e : 0.f;                          // ClampBackward (where), STE through round
        t1[k] = g[k] * qq;                                 // MulBackward, scale side (summed over the group)
        const float dx = round_to<XR>(dy[k]);
        const float tq = round_to<XR>(FAST ? div_fast(x, s, y) : x / s);
        t2[k] = round_to<XR>((-dx) * tq);                  // DivBackward, scale side
        te[k] = -e;                                        // SubBackward -> zp (asym)
    }
    acc.c1 += sum8_torch<TREE>(t1);
    acc.c2 += sum8_torch<TREE>(t2);
    acc.e += sum8_torch<TREE>(te);
    acc.dy += sum8_torch<TREE>(dy);                        // AddBackward -> zp (asym)
}
template <int XR>
__device__ __forceinline__ void bwd8(const float (&g)[8], const float (&w)[8], const float (&v)[8], float s, float zp,
                                     float qlo, float qhi, float (&dy)[8], Sums& acc, bool tree = false) {
#if AR_FASTDIV
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) ok = ok && div_fast_ok(w[k]);
    if (__all(ok)) {                       // wave-uniform; x = w/s then stays inside the verified window as well
        if (tree) bwd8_impl<XR, true, true>(g, w, v, s, 1.0f / s, zp, qlo, qhi, dy, acc);
        else bwd8_impl<XR, true, false>(g, w, v, s, 1.0f / s, zp, qlo, qhi, dy, acc);
        return;
    }
#endif
    if (tree) bwd8_impl<XR, false, true>(g, w, v, s, 0.f, zp, qlo, qhi, dy, acc);
    else bwd8_impl<XR, false, false>(g, w, v, s, 0.f, zp, qlo, qhi, dy, acc);
}

// Gradients of min_scale / max_scale from the reduced group sums: autograd's scale path mirrored op by op
// (oracle/ar_oracle.c oracle_qdq_int_bwd): partial sums rounded to the scale dtype, accumulated in it, threshold-clamp
// mask, /maxq, sign, maximum() routing (sym) or the zero-point path (asym).
template <int XR>
__device__ __forceinline__ void minmax_grads(const IntCfg& cfg, const GroupQ& q, float4 sm, float& gmin, float& gmax) {
    const bool sym = cfg.sym != 0;
    const int s_dt = cfg.s_dt;
    const float maxq = sym ? (float)(1 << (cfg.bits - 1)) : (float)((1 << cfg.bits) - 1);
    const float c1 = round_to_rt(s_dt, sm.x);
    const float c2 = round_to_rt(s_dt, round_to<XR>(sm.y));
    float ds_c = round_to_rt(s_dt, c1 + c2);
    float dlo_zp = 0.f;
    if (!sym) {
        const float dzp = sm.z + sm.w;
        const float u_over_s = ((-q.a) / q.s) / q.s;
        const float c3 = round_to_rt(s_dt, (-dzp) * u_over_s);
        ds_c = round_to_rt(s_dt, ds_c + c3);
        dlo_zp = -(dzp / q.s);
    }
    const float t = round_to_rt(s_dt, cfg.thresh);
    float ds;
    if (sym) ds = (q.s_raw < 0.f) ? ((q.s_raw <= -t) ? ds_c : 0.f) : ((q.s_raw >= t) ? ds_c : 0.f);
    else ds = (q.s_raw >= t) ? ds_c : 0.f;
    const float d32 = div_py_scalar(ds, maxq);
    if (cfg.sym == 2) {      // scale = (init_scale * max_scale).to(s_dt); min_scale is not part of the graph
        gmin = 0.f;
        gmax = ds * q.wmax;
    } else if (sym) {
        const float dm = d32 * q.sgn;
        float da, db;
        if (q.a == q.b) { da = dm / 2.f; db = dm / 2.f; }
        else if (q.a > q.b) { da = dm; db = 0.f; }
        else { da = 0.f; db = dm; }
        gmin = (-da) * q.wmin;
        gmax = db * q.wmax;
    } else {
        gmin = ((-d32) + dlo_zp) * q.wmin;
        gmax = d32 * q.wmax;
    }
}

template <int WDT, int XR, int UNROLL>
__global__ __launch_bounds__(kTPB, AR_BWD_MINW) void k_int_bwd(const BwdArgs a) {
    __shared__ float2 sg[kTPB];        // (scale, zp) of the tile's groups
    __shared__ float4 ssum[kTPB];      // per-group reduced sums, written by the first lane of each lane-group
    __shared__ float2 sg2[kTPB];       // updated (scale, zp) for the fused next forward
    const int tid = threadIdx.x;
    const int cpg = a.cpg, shift = a.cpg_shift;      // cpg is a power of two <= 64 here
    const int u_eff = cpg < UNROLL ? cpg : UNROLL;
    const int tile_chunks = kTPB * u_eff;
    const int tile_groups = tile_chunks >> shift;    // <= kTPB
    const int64_t total_chunks = a.n_groups << shift;
    const int64_t n_tiles = (a.n_groups + tile_groups - 1) / tile_groups;
    const bool sym = a.cfg.sym != 0;
    const bool do_snap = a.snap != nullptr && *a.snap != 0;
    const float alpha_v = a.lr_v ? -(*a.lr_v) : 0.f;
    const float alpha_mm = a.lr_mm ? -(*a.lr_mm) : 0.f;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t c0 = tile * (int64_t)tile_chunks;
        const int64_t g = tile * tile_groups + tid;
        const bool has_g = tid < tile_groups && g < a.n_groups;
        float wmn = 0.f, wmx = 0.f, ms = 1.f, Ms = 1.f;
        if (has_g) {
            wmn = load1<WDT>(a.wmin, g);
            wmx = load1<WDT>(a.wmax, g);
            if (a.min_s) ms = a.min_s[g];
            if (a.max_s) Ms = a.max_s[g];
        }
        Raw8<WDT> gr[UNROLL], wr[UNROLL];
        F8 vr[UNROLL];
        bool ok[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t c = c0 + u * kTPB + tid;
            ok[u] = (u < u_eff) && (c < total_chunks);
            if (ok[u]) {
                gr[u] = load8_raw<WDT>(a.dWq, c * kEPT);
                wr[u] = load8_raw<WDT>(a.W, c * kEPT);
                if (a.V) vr[u] = load8_f32(a.V, c * kEPT);
            }
        }
        GroupQ q;
        if (has_g) {
            group_scale(a.cfg, wmn, wmx, ms, Ms, q);
            sg[tid] = make_float2(q.s, sym ? 0.f : q.zp);
        }
        __syncthreads();

        float vnew[UNROLL][8];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            // every lane of a lane-group must take part in the butterfly, so no early `continue` here
            const int gl = (u * kTPB + tid) >> shift;
            const float2 sz = sg[gl < tile_groups ? gl : 0];
            float gg[8], w[8], v[8], dy[8];
            Sums acc = {0.f, 0.f, 0.f, 0.f};
            if (ok[u]) {
                unpack8<WDT>(gr[u], gg);
                unpack8<WDT>(wr[u], w);
                if (a.V) unpack_f8(vr[u], v);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = 0.f;
                }
                bwd8<XR>(gg, w, v, sz.x, sz.y, a.qlo, a.qhi, dy, acc, a.cpg <= 8);
                const int64_t e0 = (c0 + u * kTPB + tid) * kEPT;
                if (a.dV) store8_f32(a.dV, e0, dy);
                if (a.lr_v) {
                    if (do_snap && a.best_V) store8_f32(a.best_V, e0, v);   // pre-update V == the best iterate
#pragma unroll
                    for (int k = 0; k < 8; ++k) vnew[u][k] = v[k] + alpha_v * sgnf(dy[k]);
                    store8_f32(a.V, e0, vnew[u]);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) vnew[u][k] = v[k];
                }
            }
            if (u < u_eff) {   // wave-uniform
                acc.c1 = lanes_sum_torch(acc.c1, cpg);
                acc.c2 = lanes_sum_torch(acc.c2, cpg);
                if (!sym) { acc.e = lanes_sum_torch(acc.e, cpg); acc.dy = lanes_sum_torch(acc.dy, cpg); }
                if ((tid & (cpg - 1)) == 0 && gl < tile_groups) ssum[gl] = make_float4(acc.c1, acc.c2, acc.e, acc.dy);
            }
        }
        __syncthreads();

        if (has_g) {
            float gmin, gmax;
            minmax_grads<XR>(a.cfg, q, ssum[tid], gmin, gmax);
            if (a.dmin) a.dmin[g] = gmin;
            if (a.dmax) a.dmax[g] = gmax;
            float ms_new = ms, Ms_new = Ms;
            if (a.lr_mm && a.tune_minmax) {
                if (do_snap) {
                    if (a.best_min) a.best_min[g] = q.ms;
                    if (a.best_max) a.best_max[g] = q.Ms;
                }
                ms_new = q.ms + alpha_mm * sgnf(gmin);   // the clamp happened "in place" at the forward
                Ms_new = q.Ms + alpha_mm * sgnf(gmax);
                a.min_s[g] = ms_new;
                a.max_s[g] = Ms_new;
            }
            if (a.Wq_next) {
                GroupQ q2;
                group_scale(a.cfg, wmn, wmx, ms_new, Ms_new, q2);
                sg2[tid] = make_float2(q2.s, sym ? 0.f : q2.zp);
            }
        }
        if (a.Wq_next) {   // kernel-uniform
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (!ok[u]) continue;
                const float2 sz = sg2[(u * kTPB + tid) >> shift];
                float w[8], o[8];
                unpack8<WDT>(wr[u], w);
                qdq8<XR>(w, vnew[u], sz.x, sz.y, a.qlo, a.qhi, o);
                store8<WDT>(a.Wq_next, (c0 + u * kTPB + tid) * kEPT, o);
            }
        }
        __syncthreads();   // sg / ssum / sg2 are reused by the next tile
    }
}


// ------------------------------------------------------------------------------------------------------------------
// generic group sizes (any multiple of 8: per-channel rows of 11008 / 28672, non power-of-two groups, gs > 512):
// one wave owns one group and strides over its chunks; same arithmetic, same LDS-free wave reductions.  Slower per
// byte than the tiled kernels (one group's parameters per wave), used only where those do not apply.
// ------------------------------------------------------------------------------------------------------------------
template <int WDT, int XR>
__global__ __launch_bounds__(kTPB) void k_int_fwd_generic(const FwdArgs a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / kWave;
    const int64_t n_waves = (int64_t)gridDim.x * (kTPB / kWave);
    for (int64_t g = wave0; g < a.n_groups; g += n_waves) {
        GroupQ q;
        group_scale(a.cfg, load1<WDT>(a.wmin, g), load1<WDT>(a.wmax, g), a.min_s ? a.min_s[g] : 1.f,
                    a.max_s ? a.max_s[g] : 1.f, q);
        if (lane == 0) {
            if (a.scale_out) store1_rt(a.cfg.s_dt, a.scale_out, g, q.s);
            if (a.zp_out) a.zp_out[g] = q.zp;
        }
        const float zp = a.cfg.sym ? 0.f : q.zp;
        for (int c = lane; c < a.cpg; c += kWave) {
            const int64_t e0 = (g * a.cpg + c) * kEPT;
            float w[8], v[8], o[8];
            unpack8<WDT>(load8_raw<WDT>(a.W, e0), w);
            if (a.V) unpack_f8(load8_f32(a.V, e0), v);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            qdq8<XR>(w, v, q.s, zp, a.qlo, a.qhi, o);
            store8<WDT>(a.Wq, e0, o);
        }
    }
}

template <int WDT, int XR>
__global__ __launch_bounds__(kTPB) void k_int_bwd_generic(const BwdArgs a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / kWave;
    const int64_t n_waves = (int64_t)gridDim.x * (kTPB / kWave);
    const bool sym = a.cfg.sym != 0;
    const bool do_snap = a.snap != nullptr && *a.snap != 0;
    const float alpha_v = a.lr_v ? -(*a.lr_v) : 0.f;
    const float alpha_mm = a.lr_mm ? -(*a.lr_mm) : 0.f;
    for (int64_t g = wave0; g < a.n_groups; g += n_waves) {
        const float wmn = load1<WDT>(a.wmin, g), wmx = load1<WDT>(a.wmax, g);
        const float ms = a.min_s ? a.min_s[g] : 1.f, Ms = a.max_s ? a.max_s[g] : 1.f;
        GroupQ q;
        group_scale(a.cfg, wmn, wmx, ms, Ms, q);
        const float zp = sym ? 0.f : q.zp;
        Sums acc = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane; c < a.cpg; c += kWave) {
            const int64_t e0 = (g * a.cpg + c) * kEPT;
            float gg[8], w[8], v[8], dy[8];
            unpack8<WDT>(load8_raw<WDT>(a.dWq, e0), gg);
            unpack8<WDT>(load8_raw<WDT>(a.W, e0), w);
            if (a.V) unpack_f8(load8_f32(a.V, e0), v);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            bwd8<XR>(gg, w, v, q.s, zp, a.qlo, a.qhi, dy, acc);
            if (a.dV) store8_f32(a.dV, e0, dy);
            if (a.lr_v) {
                if (do_snap && a.best_V) store8_f32(a.best_V, e0, v);
                float vn[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) vn[k] = v[k] + alpha_v * sgnf(dy[k]);
                store8_f32(a.V, e0, vn);
            }
        }
        acc.c1 = lanes_sum(acc.c1, kWave);
        acc.c2 = lanes_sum(acc.c2, kWave);
        if (!sym) { acc.e = lanes_sum(acc.e, kWave); acc.dy = lanes_sum(acc.dy, kWave); }
        if (lane == 0) {
            float gmin, gmax;
            minmax_grads<XR>(a.cfg, q, make_float4(acc.c1, acc.c2, acc.e, acc.dy), gmin, gmax);
            if (a.dmin) a.dmin[g] = gmin;
            if (a.dmax) a.dmax[g] = gmax;
            if (a.lr_mm && a.tune_minmax) {
                if (do_snap) {
                    if (a.best_min) a.best_min[g] = q.ms;
                    if (a.best_max) a.best_max[g] = q.Ms;
                }
                a.min_s[g] = q.ms + alpha_mm * sgnf(gmin);
                a.max_s[g] = q.Ms + alpha_mm * sgnf(gmax);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// "flat" variants of the two hot kernels (AR_INT_FLAT=1): no LDS tile, no barrier.  Every lane derives the (scale, zp)
// of its chunk's group itself from the four group parameters (broadcast loads; gs/8 lanes repeat the same ~40 VALU
// operations), the gs/8 lanes of a group meet only in the shuffle butterfly, and the first lane of the group applies
// the min/max step.  A/B against the LDS-staged kernels is recorded in DESIGN.md section 3.
// ------------------------------------------------------------------------------------------------------------------
// SPEC != 0: instantiation for the configurations the BASELINE runs use -- symmetric, fp16 scales, SPEC lanes per group
// (16 = group 128, 4 = group 32): the scheme branches, the runtime scale-dtype rounding and the butterfly width fold at
// compile time (same arithmetic; the generic instantiation serves everything else).
template <int WDT, int XR, int U, int SPEC = 0>
__global__ __launch_bounds__(kTPB) void k_int_fwd_flat(const FwdArgs a0) {
    FwdArgs a = a0;
    if (SPEC) { a.cfg.sym = 1; a.cfg.s_dt = AR_DT_F16; a.cpg = SPEC; a.cpg_shift = SPEC == 4 ? 2 : 4; }
    const int shift = a.cpg_shift;
    const int64_t total_chunks = a.n_groups << shift;
    const int64_t stride = (int64_t)gridDim.x * kTPB * U;
    for (int64_t c0 = (int64_t)blockIdx.x * kTPB * U + threadIdx.x; c0 < total_chunks; c0 += stride) {
        Raw8<WDT> wr[U];
        F8 vr[U];
        float wmn[U], wmx[U], ms[U], Ms[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + (int64_t)u * kTPB;
            ok[u] = c < total_chunks;
            ms[u] = 1.f; Ms[u] = 1.f; wmn[u] = 0.f; wmx[u] = 0.f;
            if (ok[u]) {
                wr[u] = load8_raw<WDT>(a.W, c * kEPT);
                if (a.V) vr[u] = load8_f32(a.V, c * kEPT);
                const int64_t g = c >> shift;
                wmn[u] = load1<WDT>(a.wmin, g);
                wmx[u] = load1<WDT>(a.wmax, g);
                if (a.min_s) ms[u] = a.min_s[g];
                if (a.max_s) Ms[u] = a.max_s[g];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int64_t c = c0 + (int64_t)u * kTPB;
            GroupQ q;
            group_scale(a.cfg, wmn[u], wmx[u], ms[u], Ms[u], q);
            if ((c & ((1 << shift) - 1)) == 0) {
                const int64_t g = c >> shift;
                if (a.scale_out) store1_rt(a.cfg.s_dt, a.scale_out, g, q.s);
                if (a.zp_out) a.zp_out[g] = q.zp;
            }
            float w[8], v[8], o[8];
            unpack8<WDT>(wr[u], w);
            if (a.V) unpack_f8(vr[u], v);
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
            qdq8<XR>(w, v, q.s, a.cfg.sym ? 0.f : q.zp, a.qlo, a.qhi, o);
            store8<WDT>(a.Wq, c * kEPT, o);
        }
    }
}

template <int WDT, int XR, int U, int SPEC = 0>
__global__ __launch_bounds__(kTPB) void k_int_bwd_flat(const BwdArgs a0) {
    BwdArgs a = a0;
    if (SPEC) { a.cfg.sym = 1; a.cfg.s_dt = AR_DT_F16; a.cpg = SPEC; a.cpg_shift = SPEC == 4 ? 2 : 4; }
    const int shift = a.cpg_shift, cpg = a.cpg;
    const int64_t total_chunks = a.n_groups << shift;
    const int64_t stride = (int64_t)gridDim.x * kTPB * U;
    const bool sym = a.cfg.sym != 0;
    const bool do_snap = a.snap != nullptr && *a.snap != 0;
    const float alpha_v = a.lr_v ? -(*a.lr_v) : 0.f;
    const float alpha_mm = a.lr_mm ? -(*a.lr_mm) : 0.f;
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    for (int64_t c0 = (int64_t)blockIdx.x * kTPB * U + threadIdx.x; c0 < limit; c0 += stride) {
        Raw8<WDT> gr[U], wr[U];
        F8 vr[U];
        float wmn[U], wmx[U], ms[U], Ms[U];
        bool okk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + (int64_t)u * kTPB;
            okk[u] = c < total_chunks;
            ms[u] = 1.f; Ms[u] = 1.f; wmn[u] = 0.f; wmx[u] = 0.f;
            if (okk[u]) {
                gr[u] = load8_raw<WDT>(a.dWq, c * kEPT);
                wr[u] = load8_raw<WDT>(a.W, c * kEPT);
                if (a.V) vr[u] = load8_f32(a.V, c * kEPT);
                const int64_t g = c >> shift;
                wmn[u] = load1<WDT>(a.wmin, g);
                wmx[u] = load1<WDT>(a.wmax, g);
                if (a.min_s) ms[u] = a.min_s[g];
                if (a.max_s) Ms[u] = a.max_s[g];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t c = c0 + (int64_t)u * kTPB;
            if (c >= limit) break;              // wave-uniform
            const bool ok = okk[u];
            GroupQ q;
            group_scale(a.cfg, wmn[u], wmx[u], ms[u], Ms[u], q);
            const float zp = sym ? 0.f : q.zp;
            Sums acc = {0.f, 0.f, 0.f, 0.f};
            float w[8], vnew[8];
            if (ok) {
                float gg[8], v[8], dy[8];
                unpack8<WDT>(gr[u], gg);
                unpack8<WDT>(wr[u], w);
                if (a.V) unpack_f8(vr[u], v);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = 0.f;
                }
                bwd8<XR>(gg, w, v, q.s, zp, a.qlo, a.qhi, dy, acc, cpg <= 8);
                if (a.dV) store8_f32(a.dV, c * kEPT, dy);
                if (a.lr_v) {
                    if (do_snap && a.best_V) store8_f32(a.best_V, c * kEPT, v);
#pragma unroll
                    for (int k = 0; k < 8; ++k) vnew[k] = v[k] + alpha_v * sgnf(dy[k]);
                    store8_f32(a.V, c * kEPT, vnew);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) vnew[k] = v[k];
                }
            }
            acc.c1 = lanes_sum_torch(acc.c1, cpg);
            acc.c2 = lanes_sum_torch(acc.c2, cpg);
            if (!sym) { acc.e = lanes_sum_torch(acc.e, cpg); acc.dy = lanes_sum_torch(acc.dy, cpg); }
            if (!ok) continue;
            float gmin, gmax;
            minmax_grads<XR>(a.cfg, q, make_float4(acc.c1, acc.c2, acc.e, acc.dy), gmin, gmax);
            float ms_new = ms[u], Ms_new = Ms[u];
            const bool upd = a.lr_mm && a.tune_minmax;
            if (upd) { ms_new = q.ms + alpha_mm * sgnf(gmin); Ms_new = q.Ms + alpha_mm * sgnf(gmax); }
            if ((c & (cpg - 1)) == 0) {
                const int64_t g = c >> shift;
                if (a.dmin) a.dmin[g] = gmin;
                if (a.dmax) a.dmax[g] = gmax;
                if (upd) {
                    if (do_snap) {
                        if (a.best_min) a.best_min[g] = q.ms;
                        if (a.best_max) a.best_max[g] = q.Ms;
                    }
                    a.min_s[g] = ms_new;
                    a.max_s[g] = Ms_new;
                }
            }
            if (a.Wq_next) {   // every lane already knows its group's new scales: no hand-off needed
                GroupQ q2;
                group_scale(a.cfg, wmn[u], wmx[u], ms_new, Ms_new, q2);
                float o[8];
                qdq8<XR>(w, vnew, q2.s, sym ? 0.f : q2.zp, a.qlo, a.qhi, o);
                store8<WDT>(a.Wq_next, c * kEPT, o);
            }
        }
    }
}

// unfused sign-SGD
__global__ __launch_bounds__(kTPB) void k_sign_sgd(float* __restrict__ p, const float* __restrict__ g, int64_t n,
                                                   const float* __restrict__ lr) {
    const float alpha = -(*lr);
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    for (int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        pv.x += alpha * sgnf(gv.x); pv.y += alpha * sgnf(gv.y); pv.z += alpha * sgnf(gv.z); pv.w += alpha * sgnf(gv.w);
        reinterpret_cast<float4*>(p)[i] = pv;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n; i += stride) p[i] += alpha * sgnf(g[i]);
}

// ------------------------------------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------------------------------------
static inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}
static inline int promote_dt(int a, int b) { return a == b ? a : AR_DT_F32; }
static inline int grid_for_tiles(int64_t n_tiles) {
#ifndef AR_GRID_CAP
#define AR_GRID_CAP (1 << 24)      // effectively one tile per workgroup (measured: >= the capped grid-stride form)
#endif
    const int64_t cap = AR_GRID_CAP;
    return (int)(n_tiles < cap ? (n_tiles > 0 ? n_tiles : 1) : cap);
}
static inline bool fill_cfg(IntCfg& c, float& qlo, float& qhi, int bits, int sym, int w_dt, int s_dt, float th, float lo,
                            float hi) {
    if (bits < 2 || bits > 8) return false;
    if (w_dt < 0 || w_dt > 2 || s_dt < 0 || s_dt > 2) return false;
    c.bits = bits; c.sym = sym == 2 ? 2 : (sym ? 1 : 0); c.s_dt = s_dt; c.w_dt = w_dt; c.thresh = th; c.lo = lo; c.hi = hi;
    if (sym) { const float m = (float)(1 << (bits - 1)); qlo = -m; qhi = m - 1.f; }
    else { qlo = 0.f; qhi = (float)((1 << bits) - 1); }
    return true;
}

}  // namespace ar

using namespace ar;

// lane groups (power-of-two chunks per group up to a wave): the templated kernel, four loads per lane in flight
template <int WDT>
static bool launch_minmax_lg(const void* W, void* wmin, void* wmax, float* absmax, float* tensor_absmax, int64_t n_groups, int cpg,
                             hipStream_t st) {
    constexpr int U = 4;
    const int64_t chunks = n_groups * cpg;
    int64_t blocks = (chunks + (int64_t)kTPB * U - 1) / ((int64_t)kTPB * U);
    if (blocks < 1) blocks = 1;
    if (tensor_absmax && blocks > 256 * 8) blocks = 256 * 8;    // grid-stride: at most 2048 atomics on the global max
    if (blocks > (1 << 22)) blocks = 1 << 22;
    const int grid = (int)blocks;
#define AR_MM(C) hipLaunchKernelGGL((k_group_minmax_lg<WDT, C, U>), grid, kTPB, 0, st, W, wmin, wmax, absmax, tensor_absmax, n_groups)
    switch (cpg) {
        case 1: AR_MM(1); break;
        case 2: AR_MM(2); break;
        case 4: AR_MM(4); break;
        case 8: AR_MM(8); break;
        case 16: AR_MM(16); break;
        case 32: AR_MM(32); break;
        case 64: AR_MM(64); break;
        default: return false;
    }
#undef AR_MM
    return true;
}
static bool launch_minmax_lg(const void* W, void* wmin, void* wmax, float* absmax, float* tensor_absmax, int64_t n_groups, int cpg,
                             int w_dt, hipStream_t st) {
    switch (w_dt) {
        case AR_DT_BF16: return launch_minmax_lg<AR_DT_BF16>(W, wmin, wmax, absmax, tensor_absmax, n_groups, cpg, st);
        case AR_DT_F16: return launch_minmax_lg<AR_DT_F16>(W, wmin, wmax, absmax, tensor_absmax, n_groups, cpg, st);
        case AR_DT_F32: return launch_minmax_lg<AR_DT_F32>(W, wmin, wmax, absmax, tensor_absmax, n_groups, cpg, st);
        default: return false;
    }
}

extern "C" int ar_group_minmax(const void* W, void* wmin, void* wmax, int64_t n_groups, int gs, int w_dt,
                               ar_stream_t stream) {
    if (gs <= 0 || gs % kEPT || n_groups < 0) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    const int cpg = gs / kEPT;
    const bool lane_groups = cpg <= kWave && ilog2_exact(cpg) >= 0;
    if (lane_groups && launch_minmax_lg(W, wmin, wmax, nullptr, nullptr, n_groups, cpg, w_dt, (hipStream_t)stream)) return launch_status();
    const int64_t waves = lane_groups ? (n_groups + (kWave / cpg) - 1) / (kWave / cpg) : n_groups;
    const int grid = grid_for_tiles((waves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    switch (w_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_group_minmax<AR_DT_BF16>, grid, kTPB, 0, st, W, wmin, wmax, nullptr, nullptr, n_groups, lane_groups ? cpg : -cpg); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_group_minmax<AR_DT_F16>, grid, kTPB, 0, st, W, wmin, wmax, nullptr, nullptr, n_groups, lane_groups ? cpg : -cpg); break;
        case AR_DT_F32: hipLaunchKernelGGL(k_group_minmax<AR_DT_F32>, grid, kTPB, 0, st, W, wmin, wmax, nullptr, nullptr, n_groups, lane_groups ? cpg : -cpg); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    return launch_status();
}

extern "C" int ar_group_absmax(const void* W, float* absmax, float* tensor_absmax, int64_t n_groups, int gs, int w_dt,
                               ar_stream_t stream) {
    if (gs <= 0 || gs % kEPT || n_groups < 0) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    const int cpg = gs / kEPT;
    const bool lane_groups = cpg <= kWave && ilog2_exact(cpg) >= 0;
    if (lane_groups && launch_minmax_lg(W, nullptr, nullptr, absmax, tensor_absmax, n_groups, cpg, w_dt, (hipStream_t)stream)) return launch_status();
    const int64_t waves = lane_groups ? (n_groups + (kWave / cpg) - 1) / (kWave / cpg) : n_groups;
    int grid = grid_for_tiles((waves + 3) / 4);
    if (tensor_absmax && grid > 256 * 8) grid = 256 * 8;   // grid-stride: at most 2048 atomics on the global max
    hipStream_t st = (hipStream_t)stream;
    switch (w_dt) {
        case AR_DT_BF16: hipLaunchKernelGGL(k_group_minmax<AR_DT_BF16>, grid, kTPB, 0, st, W, nullptr, nullptr, absmax, tensor_absmax, n_groups, lane_groups ? cpg : -cpg); break;
        case AR_DT_F16: hipLaunchKernelGGL(k_group_minmax<AR_DT_F16>, grid, kTPB, 0, st, W, nullptr, nullptr, absmax, tensor_absmax, n_groups, lane_groups ? cpg : -cpg); break;
        case AR_DT_F32: hipLaunchKernelGGL(k_group_minmax<AR_DT_F32>, grid, kTPB, 0, st, W, nullptr, nullptr, absmax, tensor_absmax, n_groups, lane_groups ? cpg : -cpg); break;
        default: return AR_ERR_UNSUPPORTED;
    }
    return launch_status();
}

// AR_INT_FLAT=1 (default): the flat kernels; 0: the LDS-tiled ones.  A/B on MI355X, Llama-3-8B block, same session:
//   forward 5.82 vs 5.44 TB/s, fused backward+sign-SGD 5.02 vs 4.70 TB/s, backward+next-forward 5.12 vs 4.83 TB/s.
#ifndef AR_INT_FLAT
#define AR_INT_FLAT 1
#endif
#ifndef AR_FLAT_FWD_UNROLL
#define AR_FLAT_FWD_UNROLL 1
#endif
#ifndef AR_FLAT_BWD_UNROLL
#define AR_FLAT_BWD_UNROLL 2
#endif
#ifndef AR_INT_SPEC
#define AR_INT_SPEC 1
#endif
#ifndef AR_FWD_UNROLL
#define AR_FWD_UNROLL 4
#endif
#ifndef AR_BWD_UNROLL
#define AR_BWD_UNROLL 2
#endif

extern "C" int ar_qdq_int_fwd(const void* W, const float* V, const void* wmin, const void* wmax, const float* min_s,
                              const float* max_s, void* Wq, void* scale_out, float* zp_out, int64_t n_groups, int gs,
                              int bits, int sym, int w_dt, int s_dt, float q_thresh, float lo_bound, float hi_bound,
                              ar_stream_t stream) {
    FwdArgs a;
    if (!fill_cfg(a.cfg, a.qlo, a.qhi, bits, sym, w_dt, s_dt, q_thresh, lo_bound, hi_bound)) return AR_ERR_UNSUPPORTED;
    if (gs <= 0 || gs % kEPT || n_groups < 0) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    a.W = W; a.V = V; a.wmin = wmin; a.wmax = wmax; a.min_s = min_s; a.max_s = max_s; a.Wq = Wq;
    a.scale_out = scale_out; a.zp_out = zp_out; a.n_groups = n_groups;
    a.cpg = gs / kEPT; a.cpg_shift = ilog2_exact(a.cpg); a.x_dt = promote_dt(w_dt, s_dt);
    const int u_eff = a.cpg < AR_FWD_UNROLL ? a.cpg : AR_FWD_UNROLL;
    int tile_groups = kTPB * u_eff / a.cpg;
    if (tile_groups < 1) tile_groups = 1;
    const int grid = grid_for_tiles((n_groups + tile_groups - 1) / tile_groups);
    hipStream_t st = (hipStream_t)stream;
    // W/scale is evaluated in torch's promoted dtype: the 16-bit type when weight and scale share it, else fp32
    const bool same16 = (a.x_dt == w_dt) && (w_dt != AR_DT_F32);
    if (a.cpg > kTPB * AR_FWD_UNROLL) {   // very long groups (per-channel rows): one wave per group
        const int ggrid = grid_for_tiles((n_groups + 3) / 4);
        switch (w_dt) {
            case AR_DT_BF16:
                if (same16) hipLaunchKernelGGL((k_int_fwd_generic<AR_DT_BF16, AR_DT_BF16>), ggrid, kTPB, 0, st, a);
                else hipLaunchKernelGGL((k_int_fwd_generic<AR_DT_BF16, AR_DT_F32>), ggrid, kTPB, 0, st, a);
                break;
            case AR_DT_F16:
                if (same16) hipLaunchKernelGGL((k_int_fwd_generic<AR_DT_F16, AR_DT_F16>), ggrid, kTPB, 0, st, a);
                else hipLaunchKernelGGL((k_int_fwd_generic<AR_DT_F16, AR_DT_F32>), ggrid, kTPB, 0, st, a);
                break;
            default: hipLaunchKernelGGL((k_int_fwd_generic<AR_DT_F32, AR_DT_F32>), ggrid, kTPB, 0, st, a); break;
        }
        return launch_status();
    }
#if AR_INT_FLAT
    if (a.cpg_shift >= 0 && a.cpg <= kWave) {
        const int fgrid = grid_for_tiles((n_groups * a.cpg + kTPB * AR_FLAT_FWD_UNROLL - 1) / (kTPB * AR_FLAT_FWD_UNROLL));
#if AR_INT_SPEC
        // measured (tools/kbench.py): the forward is HBM-bound either way (group 128: equal, group 32: the generic one is
        // 1.7 % faster), the fused backward gains 4-9 % -- so only group 128 takes the specialised forward
        if (w_dt == AR_DT_BF16 && !same16 && a.cfg.sym == 1 && s_dt == AR_DT_F16 && a.cpg == 16) {
            AR_LAUNCH_PROF(AR_PROF_INT_FWD, a.n_groups, (k_int_fwd_flat<AR_DT_BF16, AR_DT_F32, AR_FLAT_FWD_UNROLL, 16>), fgrid, kTPB, 0, st, a);
            return launch_status();
        }
#endif
        switch (w_dt) {
            case AR_DT_BF16:
                if (same16) AR_LAUNCH_PROF(AR_PROF_INT_FWD, a.n_groups, (k_int_fwd_flat<AR_DT_BF16, AR_DT_BF16, AR_FLAT_FWD_UNROLL>), fgrid, kTPB, 0, st, a);
                else AR_LAUNCH_PROF(AR_PROF_INT_FWD, a.n_groups, (k_int_fwd_flat<AR_DT_BF16, AR_DT_F32, AR_FLAT_FWD_UNROLL>), fgrid, kTPB, 0, st, a);
                break;
            case AR_DT_F16:
                if (same16) AR_LAUNCH_PROF(AR_PROF_INT_FWD, a.n_groups, (k_int_fwd_flat<AR_DT_F16, AR_DT_F16, AR_FLAT_FWD_UNROLL>), fgrid, kTPB, 0, st, a);
                else AR_LAUNCH_PROF(AR_PROF_INT_FWD, a.n_groups, (k_int_fwd_flat<AR_DT_F16, AR_DT_F32, AR_FLAT_FWD_UNROLL>), fgrid, kTPB, 0, st, a);
                break;
            default: AR_LAUNCH_PROF(AR_PROF_INT_FWD, a.n_groups, (k_int_fwd_flat<AR_DT_F32, AR_DT_F32, AR_FLAT_FWD_UNROLL>), fgrid, kTPB, 0, st, a); break;
        }
        return launch_status();
    }
#endif
    switch (w_dt) {
        case AR_DT_BF16:
            if (same16) hipLaunchKernelGGL((k_int_fwd<AR_DT_BF16, AR_DT_BF16, AR_FWD_UNROLL>), grid, kTPB, 0, st, a);
            else hipLaunchKernelGGL((k_int_fwd<AR_DT_BF16, AR_DT_F32, AR_FWD_UNROLL>), grid, kTPB, 0, st, a);
            break;
        case AR_DT_F16:
            if (same16) hipLaunchKernelGGL((k_int_fwd<AR_DT_F16, AR_DT_F16, AR_FWD_UNROLL>), grid, kTPB, 0, st, a);
            else hipLaunchKernelGGL((k_int_fwd<AR_DT_F16, AR_DT_F32, AR_FWD_UNROLL>), grid, kTPB, 0, st, a);
            break;
        default: hipLaunchKernelGGL((k_int_fwd<AR_DT_F32, AR_DT_F32, AR_FWD_UNROLL>), grid, kTPB, 0, st, a); break;
    }
    return launch_status();
}

static int launch_int_bwd(BwdArgs& a, int gs, int bits, int sym, int w_dt, int s_dt, float q_thresh, float lo, float hi,
                          ar_stream_t stream) {
    if (!fill_cfg(a.cfg, a.qlo, a.qhi, bits, sym, w_dt, s_dt, q_thresh, lo, hi)) return AR_ERR_UNSUPPORTED;
    if (gs <= 0 || gs % kEPT || a.n_groups < 0) return AR_ERR_UNSUPPORTED;
    a.cpg = gs / kEPT; a.cpg_shift = ilog2_exact(a.cpg); a.x_dt = promote_dt(w_dt, s_dt);
    if (a.n_groups == 0) return AR_OK;
    if (a.cpg_shift < 0 || a.cpg > kWave) {   // lane-group butterfly needs gs in {8,16,...,512}: otherwise one wave per group
        if (a.Wq_next) return AR_ERR_UNSUPPORTED;      // the fused next forward exists in the tiled kernel only
        const bool same16g = (a.x_dt == w_dt) && (w_dt != AR_DT_F32);
        const int ggrid = grid_for_tiles((a.n_groups + 3) / 4);
        hipStream_t gst = (hipStream_t)stream;
        switch (w_dt) {
            case AR_DT_BF16:
                if (same16g) hipLaunchKernelGGL((k_int_bwd_generic<AR_DT_BF16, AR_DT_BF16>), ggrid, kTPB, 0, gst, a);
                else hipLaunchKernelGGL((k_int_bwd_generic<AR_DT_BF16, AR_DT_F32>), ggrid, kTPB, 0, gst, a);
                break;
            case AR_DT_F16:
                if (same16g) hipLaunchKernelGGL((k_int_bwd_generic<AR_DT_F16, AR_DT_F16>), ggrid, kTPB, 0, gst, a);
                else hipLaunchKernelGGL((k_int_bwd_generic<AR_DT_F16, AR_DT_F32>), ggrid, kTPB, 0, gst, a);
                break;
            default: hipLaunchKernelGGL((k_int_bwd_generic<AR_DT_F32, AR_DT_F32>), ggrid, kTPB, 0, gst, a); break;
        }
        return launch_status();
    }
    const int u_eff = a.cpg < AR_BWD_UNROLL ? a.cpg : AR_BWD_UNROLL;
    const int tile_groups = kTPB * u_eff / a.cpg;
    const int grid = grid_for_tiles((a.n_groups + tile_groups - 1) / tile_groups);
    hipStream_t st = (hipStream_t)stream;
    const bool same16 = (a.x_dt == w_dt) && (w_dt != AR_DT_F32);
#if AR_INT_FLAT
    {
        const int fgrid = grid_for_tiles((a.n_groups * a.cpg + kTPB * AR_FLAT_BWD_UNROLL - 1) / (kTPB * AR_FLAT_BWD_UNROLL));
#if AR_INT_SPEC
        if (w_dt == AR_DT_BF16 && !same16 && a.cfg.sym == 1 && s_dt == AR_DT_F16 && (a.cpg == 16 || a.cpg == 4)) {
            // (one chunk per lane for small blocks -- OPT-125M's 85 MB stream -- was measured in round 3: 22 vs 21 us, no gain)
            if (a.cpg == 16) AR_LAUNCH_PROF(AR_PROF_INT_BWD, a.n_groups, (k_int_bwd_flat<AR_DT_BF16, AR_DT_F32, AR_FLAT_BWD_UNROLL, 16>), fgrid, kTPB, 0, st, a);
            else AR_LAUNCH_PROF(AR_PROF_INT_BWD, a.n_groups, (k_int_bwd_flat<AR_DT_BF16, AR_DT_F32, AR_FLAT_BWD_UNROLL, 4>), fgrid, kTPB, 0, st, a);
            return launch_status();
        }
#endif
        switch (w_dt) {
            case AR_DT_BF16:
                if (same16) AR_LAUNCH_PROF(AR_PROF_INT_BWD, a.n_groups, (k_int_bwd_flat<AR_DT_BF16, AR_DT_BF16, AR_FLAT_BWD_UNROLL>), fgrid, kTPB, 0, st, a);
                else AR_LAUNCH_PROF(AR_PROF_INT_BWD, a.n_groups, (k_int_bwd_flat<AR_DT_BF16, AR_DT_F32, AR_FLAT_BWD_UNROLL>), fgrid, kTPB, 0, st, a);
                break;
            case AR_DT_F16:
                if (same16) AR_LAUNCH_PROF(AR_PROF_INT_BWD, a.n_groups, (k_int_bwd_flat<AR_DT_F16, AR_DT_F16, AR_FLAT_BWD_UNROLL>), fgrid, kTPB, 0, st, a);
                else AR_LAUNCH_PROF(AR_PROF_INT_BWD, a.n_groups, (k_int_bwd_flat<AR_DT_F16, AR_DT_F32, AR_FLAT_BWD_UNROLL>), fgrid, kTPB, 0, st, a);
                break;
            default: AR_LAUNCH_PROF(AR_PROF_INT_BWD, a.n_groups, (k_int_bwd_flat<AR_DT_F32, AR_DT_F32, AR_FLAT_BWD_UNROLL>), fgrid, kTPB, 0, st, a); break;
        }
        return launch_status();
    }
#endif
    switch (w_dt) {
        case AR_DT_BF16:
            if (same16) hipLaunchKernelGGL((k_int_bwd<AR_DT_BF16, AR_DT_BF16, AR_BWD_UNROLL>), grid, kTPB, 0, st, a);
            else hipLaunchKernelGGL((k_int_bwd<AR_DT_BF16, AR_DT_F32, AR_BWD_UNROLL>), grid, kTPB, 0, st, a);
            break;
        case AR_DT_F16:
            if (same16) hipLaunchKernelGGL((k_int_bwd<AR_DT_F16, AR_DT_F16, AR_BWD_UNROLL>), grid, kTPB, 0, st, a);
            else hipLaunchKernelGGL((k_int_bwd<AR_DT_F16, AR_DT_F32, AR_BWD_UNROLL>), grid, kTPB, 0, st, a);
            break;
        default: hipLaunchKernelGGL((k_int_bwd<AR_DT_F32, AR_DT_F32, AR_BWD_UNROLL>), grid, kTPB, 0, st, a); break;
    }
    return launch_status();
}

extern "C" int ar_qdq_int_bwd(const void* dWq, const void* W, const float* V, const void* wmin, const void* wmax,
                              const float* min_s, const float* max_s, float* dV, float* dmin, float* dmax,
                              int64_t n_groups, int gs, int bits, int sym, int w_dt, int s_dt, float q_thresh,
                              float lo_bound, float hi_bound, ar_stream_t stream) {
    BwdArgs a = {};
    a.dWq = dWq; a.W = W; a.V = const_cast<float*>(V); a.wmin = wmin; a.wmax = wmax;
    a.min_s = const_cast<float*>(min_s); a.max_s = const_cast<float*>(max_s);
    a.dV = dV; a.dmin = dmin; a.dmax = dmax; a.n_groups = n_groups; a.tune_minmax = 0;
    return launch_int_bwd(a, gs, bits, sym, w_dt, s_dt, q_thresh, lo_bound, hi_bound, stream);
}

extern "C" int ar_qdq_int_bwd_sgd(const void* dWq, const void* W, float* V, const void* wmin, const void* wmax,
                                  float* min_s, float* max_s, int64_t n_groups, int gs, int bits, int sym, int w_dt,
                                  int s_dt, float q_thresh, float lo_bound, float hi_bound, const float* lr_v_dev,
                                  const float* lr_mm_dev, int tune_minmax, const int32_t* snapshot_flag, float* best_V,
                                  float* best_min, float* best_max, void* Wq_next, ar_stream_t stream) {
    if (n_groups == 0) return AR_OK;                 // empty block: nothing to launch (pointers may be NULL)
    if (!V || !lr_v_dev) return AR_ERR_UNSUPPORTED;
    BwdArgs a = {};
    a.dWq = dWq; a.W = W; a.V = V; a.wmin = wmin; a.wmax = wmax; a.min_s = min_s; a.max_s = max_s;
    a.lr_v = lr_v_dev; a.lr_mm = (tune_minmax && min_s && max_s) ? lr_mm_dev : nullptr;
    a.tune_minmax = (tune_minmax && min_s && max_s && lr_mm_dev) ? 1 : 0;
    a.snap = snapshot_flag; a.best_V = best_V; a.best_min = best_min; a.best_max = best_max;
    a.Wq_next = Wq_next; a.n_groups = n_groups;
    return launch_int_bwd(a, gs, bits, sym, w_dt, s_dt, q_thresh, lo_bound, hi_bound, stream);
}

extern "C" int ar_sign_sgd(float* p, const float* g, int64_t n, const float* lr_dev, ar_stream_t stream) {
    if (n <= 0) return AR_OK;
    const int grid = grid_for_tiles((n / 4 + kTPB - 1) / kTPB);
    hipLaunchKernelGGL(k_sign_sgd, grid, kTPB, 0, (hipStream_t)stream, p, g, n, lr_dev);
    return launch_status();
}


// ------------------------------------------------------------------------------------------------------------------
// int-sym init-scale search of the algorithm extension (search_scales, auto_round/data_type/int.py:24-86, + the
// threshold clamp of search_int, data_type/utils.py:203-209).  Runs once per layer before tuning.  A lane-group owns a
// group and keeps its 8 elements per lane in registers across the ~400 candidates; groups that do not fit a lane-group
// take a whole wave and re-read their chunks from L2 per candidate.  All arithmetic is rounded to the weight dtype
// where torch rounds it.
// ------------------------------------------------------------------------------------------------------------------
namespace ar {
template <int XDT> __device__ __forceinline__ float recip_dt(float t) {
    const float eps = round_to<XDT>(XDT == AR_DT_F16 ? 1e-5f : 1e-30f);
    return fabsf(t) >= eps ? round_to<XDT>(1.0f / t) : 0.f;
}
// A lane's eight importance-weighted squared errors are added in the association torch's reduction kernel gives `torch.sum(loss,
// dim=-1)` on this GPU (TREE: rows shorter than 128 values, one value per torch thread -> a pairwise tree; otherwise two float4 runs;
// see sum8_torch): with the lanes combined neighbours first (lanes_sum_torch) a group's loss then carries the bits the reference's
// `loss < best_loss` compares when it runs on the GPU -- a near-tie between two candidates picks the same scale (round 5; before,
// eight in a row + a far-stride butterfly: the same value, another rounding, ~0.5 % other scales).
template <int XDT, bool TREE> __device__ __forceinline__ float search_loss8(const float (&x)[8], const float (&qw)[8], float isc, float sc,
                                                                            float nmax) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float L = clamp3(__builtin_rintf(round_to<XDT>(isc * x[k])), -nmax, nmax - 1.f);
        const float e = round_to<XDT>(round_to<XDT>(sc * L) - x[k]);
        t[k] = (e * e) * qw[k];
    }
    return sum8_torch<TREE>(t);
}
// first-index arg-max of |x| over `width` lanes: (abs, flat index, signed value)
__device__ __forceinline__ void lanes_argmax(float& a, int& idx, float& v, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) {
        const float oa = __shfl_xor(a, m, kWave);
        const int oi = __shfl_xor(idx, m, kWave);
        const float ov = __shfl_xor(v, m, kWave);
        if (oa > a || (oa == a && oi < idx)) { a = oa; idx = oi; v = ov; }
    }
}

template <int XDT>
__global__ __launch_bounds__(kTPB) void k_search_int_scale(const void* __restrict__ X, const float* __restrict__ qw_row,
                                                           int64_t groups_per_row, const float* __restrict__ cand,
                                                           int n_cand, void* __restrict__ out_raw,
                                                           void* __restrict__ out_init, int64_t n_groups, int cpg,
                                                           float nmax, float thresh) {
    const int64_t total_chunks = n_groups * cpg;
    const int64_t stride = (int64_t)gridDim.x * kTPB;
    const int64_t limit = (total_chunks + kWave - 1) / kWave * kWave;
    const float th = round_to<XDT>(thresh);
    for (int64_t c = (int64_t)blockIdx.x * kTPB + threadIdx.x; c < limit; c += stride) {
        const bool ok = c < total_chunks;
        const int64_t g = ok ? c / cpg : 0;
        const int cin = ok ? (int)(c % cpg) : 0;
        float x[8], qw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = 0.f; qw[k] = ok ? 1.f : 0.f; }
        if (ok) {
            unpack8<XDT>(load8_raw<XDT>(X, c * kEPT), x);
            if (qw_row) unpack_f8(load8_f32(qw_row, ((g % groups_per_row) * cpg + cin) * kEPT), qw);
        }
        float am = -1.f, gv = 0.f;
        int ai = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (fabsf(x[k]) > am) { am = fabsf(x[k]); gv = x[k]; ai = cin * 8 + k; }
        if (!ok) am = -2.f;
        lanes_argmax(am, ai, gv, cpg);
        const float rg = recip_dt<XDT>(gv);
        float best = 0.f, best_s = 0.f;
        for (int ci = 0; ci < n_cand; ++ci) {
            const float isc = round_to<XDT>((-cand[ci]) * rg);
            const float sc = recip_dt<XDT>(isc);
            const float loss = lanes_sum_torch(cpg < 16 ? search_loss8<XDT, true>(x, qw, isc, sc, nmax)
                                                        : search_loss8<XDT, false>(x, qw, isc, sc, nmax), cpg);
            if (ci == 0 || loss < best) { best = loss; best_s = sc; }
        }
        if (ok && cin == 0) {
            if (out_raw) store1<XDT>(out_raw, g, best_s);
            const float cl = best_s < 0.f ? (best_s > -th ? -th : best_s) : (best_s < th ? th : best_s);
            if (out_init) store1<XDT>(out_init, g, cl);
        }
    }
}

template <int XDT>
__global__ __launch_bounds__(kTPB) void k_search_int_scale_wave(const void* __restrict__ X, const float* __restrict__ qw_row,
                                                                int64_t groups_per_row, const float* __restrict__ cand,
                                                                int n_cand, void* __restrict__ out_raw,
                                                                void* __restrict__ out_init, int64_t n_groups, int cpg,
                                                                float nmax, float thresh) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave0 = ((int64_t)blockIdx.x * kTPB + threadIdx.x) / kWave;
    const int64_t n_waves = (int64_t)gridDim.x * (kTPB / kWave);
    const float th = round_to<XDT>(thresh);
    for (int64_t g = wave0; g < n_groups; g += n_waves) {
        float am = -1.f, gv = 0.f;
        int ai = 0;
        for (int ch = lane; ch < cpg; ch += kWave) {
            float x[8];
            unpack8<XDT>(load8_raw<XDT>(X, (g * cpg + ch) * kEPT), x);
#pragma unroll
            for (int k = 0; k < 8; ++k) if (fabsf(x[k]) > am) { am = fabsf(x[k]); gv = x[k]; ai = ch * 8 + k; }
        }
        lanes_argmax(am, ai, gv, kWave);
        const float rg = recip_dt<XDT>(gv);
        float best = 0.f, best_s = 0.f;
        for (int ci = 0; ci < n_cand; ++ci) {
            const float isc = round_to<XDT>((-cand[ci]) * rg);
            const float sc = recip_dt<XDT>(isc);
            float part = 0.f;
            for (int ch = lane; ch < cpg; ch += kWave) {
                float x[8], qw[8];
                unpack8<XDT>(load8_raw<XDT>(X, (g * cpg + ch) * kEPT), x);
#pragma unroll
                for (int k = 0; k < 8; ++k) qw[k] = 1.f;
                if (qw_row) unpack_f8(load8_f32(qw_row, ((g % groups_per_row) * cpg + ch) * kEPT), qw);
                part += search_loss8<XDT, false>(x, qw, isc, sc, nmax);
            }
            const float loss = lanes_sum(part, kWave);
            if (ci == 0 || loss < best) { best = loss; best_s = sc; }
        }
        if (lane == 0) {
            if (out_raw) store1<XDT>(out_raw, g, best_s);
            const float cl = best_s < 0.f ? (best_s > -th ? -th : best_s) : (best_s < th ? th : best_s);
            if (out_init) store1<XDT>(out_init, g, cl);
        }
    }
}
}  // namespace ar

extern "C" int ar_search_int_scale(const void* X, const float* qw_row, int64_t groups_per_row, const float* candidates_dev,
                                   int n_candidates, void* out_raw, void* out_init, int64_t n_groups, int gs, int bits,
                                   int x_dt, float q_thresh, ar_stream_t stream) {
    if (gs <= 0 || gs % kEPT || n_groups < 0 || n_candidates <= 0 || !candidates_dev || bits < 2 || bits > 8) return AR_ERR_UNSUPPORTED;
    if (qw_row && groups_per_row <= 0) return AR_ERR_UNSUPPORTED;
    if (n_groups == 0) return AR_OK;
    const int cpg = gs / kEPT;
    const bool lane_groups = cpg <= kWave && ilog2_exact(cpg) >= 0;
    const float nmax = (float)(1 << (bits - 1));
    hipStream_t st = (hipStream_t)stream;
    const int64_t want = lane_groups ? (n_groups * cpg + kTPB - 1) / kTPB : (n_groups + kTPB / kWave - 1) / (kTPB / kWave);
    const int grid = (int)(want < 1 ? 1 : (want > (1 << 20) ? (1 << 20) : want));
#define AR_LAUNCH_SEARCH(DT)                                                                                              \
    if (lane_groups) hipLaunchKernelGGL(k_search_int_scale<DT>, grid, kTPB, 0, st, X, qw_row, groups_per_row, candidates_dev, \
                                        n_candidates, out_raw, out_init, n_groups, cpg, nmax, q_thresh);                  \
    else hipLaunchKernelGGL(k_search_int_scale_wave<DT>, grid, kTPB, 0, st, X, qw_row, groups_per_row, candidates_dev,    \
                            n_candidates, out_raw, out_init, n_groups, cpg, nmax, q_thresh)
    switch (x_dt) {
        case AR_DT_BF16: AR_LAUNCH_SEARCH(AR_DT_BF16); break;
        case AR_DT_F16: AR_LAUNCH_SEARCH(AR_DT_F16); break;
        case AR_DT_F32: AR_LAUNCH_SEARCH(AR_DT_F32); break;
        default: return AR_ERR_UNSUPPORTED;
    }
#undef AR_LAUNCH_SEARCH
    return launch_status();
}
