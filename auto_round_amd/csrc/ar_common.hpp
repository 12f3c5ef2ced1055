// ar_common.hpp -- device helpers shared by the gfx950 kernels (dtype conversion, 16-byte vector access,
// per-group scale arithmetic).  CDNA4 only: 64-wide wavefronts are assumed everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "ar_mi355x.h"

namespace ar {

// true exactly once per (call site, HIP device): function attributes such as the dynamic-LDS limit are per device, and a process
// may drive more than one (SignRoundQuantizer(device="cuda:1") after work on cuda:0)
struct PerDeviceOnce {
    bool done[64] = {};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

constexpr int kWave = 64;   // CDNA wavefront
constexpr int kTPB = 256;   // threads per workgroup (4 waves, one per SIMD)
constexpr int kEPT = 8;     // elements per lane per chunk: 16 B of bf16/f16, 32 B of fp32

// ---- scalar dtype conversions (RNE, identical to torch .to()) ------------------------------------------------
__device__ __forceinline__ float bf16_lo(uint32_t pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t pair) { return __uint_as_float(pair & 0xffff0000u); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float f16_to_f32(uint32_t h) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}
__device__ __forceinline__ uint32_t f32_to_f16(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);   // v_cvt_f16_f32, RNE
}
// two fp32 -> packed bf16x2 with the gfx950 conversion instruction (v_cvt_pk_bf16_f32: RNE, NaN quieted); verified
// against the integer RNE formula over all 2^32 inputs by tools/exactcheck (profiles/archive/r01_exactcheck.json).
#ifndef AR_HW_BF16
#define AR_HW_BF16 1
#endif
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if AR_HW_BF16
    f32x2_t v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
#else
    return f32_to_bf16(lo) | (f32_to_bf16(hi) << 16);
#endif
}

template <int DT> __device__ __forceinline__ float round_to(float v) {
    if constexpr (DT == AR_DT_BF16) return __uint_as_float(pack_bf16x2(v, v) & 0xffff0000u);      // 2 instructions (software RNE: 7)
    else if constexpr (DT == AR_DT_F16) return f16_to_f32(f32_to_f16(v));
    else return v;
}
__device__ __forceinline__ float round_to_rt(int dt, float v) {
    if (dt == AR_DT_BF16) return round_to<AR_DT_BF16>(v);
    if (dt == AR_DT_F16) return round_to<AR_DT_F16>(v);
    return v;
}
template <int DT> __device__ __forceinline__ float load1(const void* p, int64_t i) {
    if constexpr (DT == AR_DT_BF16) return __uint_as_float((uint32_t)((const uint16_t*)p)[i] << 16);
    else if constexpr (DT == AR_DT_F16) return f16_to_f32(((const uint16_t*)p)[i]);
    else return ((const float*)p)[i];
}
template <int DT> __device__ __forceinline__ void store1(void* p, int64_t i, float v) {
    if constexpr (DT == AR_DT_BF16) ((uint16_t*)p)[i] = (uint16_t)f32_to_bf16(v);
    else if constexpr (DT == AR_DT_F16) ((uint16_t*)p)[i] = (uint16_t)f32_to_f16(v);
    else ((float*)p)[i] = v;
}
__device__ __forceinline__ float load1_rt(int dt, const void* p, int64_t i) {
    if (dt == AR_DT_BF16) return load1<AR_DT_BF16>(p, i);
    if (dt == AR_DT_F16) return load1<AR_DT_F16>(p, i);
    return load1<AR_DT_F32>(p, i);
}
__device__ __forceinline__ void store1_rt(int dt, void* p, int64_t i, float v) {
    if (dt == AR_DT_BF16) store1<AR_DT_BF16>(p, i, v);
    else if (dt == AR_DT_F16) store1<AR_DT_F16>(p, i, v);
    else store1<AR_DT_F32>(p, i, v);
}

// ---- 8-element (one chunk) register tiles with 16-byte global accesses ------------------------------------------
template <int DT> struct Raw8;                         // what one lane holds for one chunk, still packed
template <> struct Raw8<AR_DT_BF16> { uint4 q; };
template <> struct Raw8<AR_DT_F16> { uint4 q; };
template <> struct Raw8<AR_DT_F32> { float4 a, b; };

// 16-byte global accesses.  AR_NT=1 would mark the streaming arrays non-temporal.  Measured on MI355X (round 1, A/B in one
// session, Llama-3-8B block): no gain for these kernels (fwd 5.0-5.2 TB/s either way), and with ROCm 7.2 the
// __builtin_nontemporal_store path lost the sign of a -0 in the high fp16 half of the 4th dword (caught by the f16
// golden test), so it stays off.
#ifndef AR_NT
#define AR_NT 0
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(const void* p) {
#if AR_NT
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}
__device__ __forceinline__ float4 ld16f(const void* p) {
#if AR_NT
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st16(void* p, uint4 q) {
#if AR_NT
    u32x4_t v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
#else
    *reinterpret_cast<uint4*>(p) = q;
#endif
}
__device__ __forceinline__ void st16f(void* p, float4 q) {
#if AR_NT
    f32x4_t v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(p));
#else
    *reinterpret_cast<float4*>(p) = q;
#endif
}
template <int DT> __device__ __forceinline__ Raw8<DT> load8_raw(const void* base, int64_t elem) {
    Raw8<DT> r;
    if constexpr (DT == AR_DT_F32) {
        const float* p = reinterpret_cast<const float*>(base) + elem;
        r.a = ld16f(p); r.b = ld16f(p + 4);
    } else {
        r.q = ld16(reinterpret_cast<const uint16_t*>(base) + elem);
    }
    return r;
}
template <int DT> __device__ __forceinline__ void unpack8(const Raw8<DT>& r, float (&o)[8]) {
    if constexpr (DT == AR_DT_F32) {
        o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
    } else if constexpr (DT == AR_DT_BF16) {
        o[0] = bf16_lo(r.q.x); o[1] = bf16_hi(r.q.x); o[2] = bf16_lo(r.q.y); o[3] = bf16_hi(r.q.y);
        o[4] = bf16_lo(r.q.z); o[5] = bf16_hi(r.q.z); o[6] = bf16_lo(r.q.w); o[7] = bf16_hi(r.q.w);
    } else {
        o[0] = f16_to_f32(r.q.x & 0xffffu); o[1] = f16_to_f32(r.q.x >> 16); o[2] = f16_to_f32(r.q.y & 0xffffu);
        o[3] = f16_to_f32(r.q.y >> 16); o[4] = f16_to_f32(r.q.z & 0xffffu); o[5] = f16_to_f32(r.q.z >> 16);
        o[6] = f16_to_f32(r.q.w & 0xffffu); o[7] = f16_to_f32(r.q.w >> 16);
    }
}
// ---- exact division by a per-group scale in 3 instructions --------------------------------------------------------
// y = 1/s is the correctly rounded reciprocal (one IEEE division per lane per chunk, shared by its 8 elements); then
//     q0 = w*y ; r = fma(-q0, s, w) (exact residual) ; q = fma(r, y, q0)
// is the correctly rounded quotient w/s (Markstein's theorem) as long as nothing under/overflows.  tools/exactcheck
// enumerates ALL bf16/fp16 weights x ALL admissible fp16 scales (2^32 pairs each, both for w/s and for (w/s)/s) and
// finds bit-identical results to the IEEE division whenever |w| is in [2^-64, 2^64] or w == 0; outside that window the
// kernels take the plain IEEE division (wave-uniform branch, never taken for real checkpoints).
#ifndef AR_FASTDIV
#define AR_FASTDIV 1
#endif
__device__ __forceinline__ bool div_fast_ok(float w) {
    const uint32_t e = (__float_as_uint(w) >> 23) & 0xffu;
    return (e - 63u <= 128u) || (w == 0.f);
}
__device__ __forceinline__ float div_fast(float w, float s, float y) {
    const float q0 = w * y;
    const float r = __builtin_fmaf(-q0, s, w);
    const float q1 = __builtin_fmaf(r, y, q0);
    return (w == 0.f) ? q0 : q1;
}

template <int DT> __device__ __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    if constexpr (DT == AR_DT_F32) {
        float* p = reinterpret_cast<float*>(base) + elem;
        st16f(p, make_float4(v[0], v[1], v[2], v[3]));
        st16f(p + 4, make_float4(v[4], v[5], v[6], v[7]));
    } else {
        uint4 q;
        if constexpr (DT == AR_DT_BF16) {
            q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]);
            q.z = pack_bf16x2(v[4], v[5]); q.w = pack_bf16x2(v[6], v[7]);
        } else {
            q.x = f32_to_f16(v[0]) | (f32_to_f16(v[1]) << 16); q.y = f32_to_f16(v[2]) | (f32_to_f16(v[3]) << 16);
            q.z = f32_to_f16(v[4]) | (f32_to_f16(v[5]) << 16); q.w = f32_to_f16(v[6]) | (f32_to_f16(v[7]) << 16);
        }
        st16(reinterpret_cast<uint16_t*>(base) + elem, q);
    }
}
struct F8 { float4 a, b; };
__device__ __forceinline__ F8 load8_f32(const float* base, int64_t elem) {
    F8 r; r.a = ld16f(base + elem); r.b = ld16f(base + elem + 4); return r;
}
__device__ __forceinline__ void unpack_f8(const F8& r, float (&o)[8]) {
    o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
}
__device__ __forceinline__ void store8_f32(float* base, int64_t elem, const float (&v)[8]) {
    st16f(base + elem, make_float4(v[0], v[1], v[2], v[3]));
    st16f(base + elem + 4, make_float4(v[4], v[5], v[6], v[7]));
}

// ---- small math -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }
// forward value of the reference's round_ste: (x.round() - x).detach() + x  == rint(x) with a zero result always +0
__device__ __forceinline__ float round_ste_value(float y) {
    float d = __builtin_rintf(y) - y;
    return d + y;
}
// butterfly all-reduce (sum) over `width` consecutive lanes (width = power of two <= 64)
__device__ __forceinline__ float lanes_sum(float v, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
// the same sum in the association torch's reduction kernel uses for a contiguous inner dimension (ATen/native/cuda/Reduce.cuh,
// block_x_reduce: `for (offset = 1; offset < dim_x; offset <<= 1) value += shfl_down(value, offset)`): neighbours first.  With the
// in-lane order of bwd8 (two runs of four, like two of torch's float4-vectorised threads) the group sums of the fake-quant
// backward then round exactly where autograd's `sum_to_size` rounds on this GPU.
__device__ __forceinline__ float lanes_sum_torch(float v, int width) {
    for (int m = 1; m < width; m <<= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
// a lane's eight consecutive elements in the same association: TREE = one element per torch thread (rows of 16 / 32 / 64: a pure
// pairwise tree), otherwise two float4-vectorised threads (rows of 128 and more)
template <bool TREE>
__device__ __forceinline__ float sum8_torch(const float (&t)[8]) {
    if (TREE) return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    return (((t[0] + t[1]) + t[2]) + t[3]) + (((t[4] + t[5]) + t[6]) + t[7]);
}
__device__ __forceinline__ float lanes_max(float v, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}
__device__ __forceinline__ float lanes_min(float v, int width) {
    for (int m = width >> 1; m > 0; m >>= 1) v = fminf(v, __shfl_xor(v, m, kWave));
    return v;
}

// ---- lane-group butterflies on DPP (no LDS crossbar): after the two quad permutes every quad is uniform, row_half_mirror then
// swaps the two quads of an 8-lane half and row_mirror the two halves of a 16-lane row; 32 / 64 lanes finish through ds_bpermute.
// One VOP2 instruction with the DPP operand per step; written as inline asm because the compiler's own form of it is
// mov + mov_dpp + canonicalise + op.  `s_nop 1` covers the two wait states between a VALU write of the source and its DPP read.
#define AR_DPP_STEP(OPC, CTRL, V)                                                                                  \
    asm volatile("s_nop 1\n\t" OPC "_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(V) : "v"(V))
#define AR_GROUP_REDUCE(T, NAME, OPC, OP)                                                                         \
    template <int W> __device__ __forceinline__ T NAME(T v) {                                                     \
        if constexpr (W >= 2) AR_DPP_STEP(OPC, "quad_perm:[1,0,3,2]", v);                                         \
        if constexpr (W >= 4) AR_DPP_STEP(OPC, "quad_perm:[2,3,0,1]", v);                                         \
        if constexpr (W >= 8) AR_DPP_STEP(OPC, "row_half_mirror", v);                                             \
        if constexpr (W >= 16) AR_DPP_STEP(OPC, "row_mirror", v);                                                 \
        if constexpr (W >= 32) { const T o = __shfl_xor(v, 16, kWave); v = OP(v, o); }                            \
        if constexpr (W >= 64) { const T o = __shfl_xor(v, 32, kWave); v = OP(v, o); }                            \
        return v;                                                                                                 \
    }
__device__ __forceinline__ float op_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float op_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float op_add(float a, float b) { return a + b; }
__device__ __forceinline__ int op_imin(int a, int b) { return a < b ? a : b; }
AR_GROUP_REDUCE(float, group_min, "v_min_f32", op_min)
AR_GROUP_REDUCE(float, group_max, "v_max_f32", op_max)
AR_GROUP_REDUCE(float, group_sum, "v_add_f32", op_add)
AR_GROUP_REDUCE(int, group_imin, "v_min_i32", op_imin)
// sum over 2 or 4 lanes in the order of the shuffle butterfly it replaces (stride 2, then 1): bit-identical sums
template <int W> __device__ __forceinline__ float group_sum_desc(float v) {
    static_assert(W == 1 || W == 2 || W == 4, "quad permutes only");
    if constexpr (W >= 4) AR_DPP_STEP("v_add_f32", "quad_perm:[2,3,0,1]", v);
    if constexpr (W >= 2) AR_DPP_STEP("v_add_f32", "quad_perm:[1,0,3,2]", v);
    return v;
}
// the same over 2 or 4 lanes neighbours first (stride 1, then 2): the association of torch's reduction kernel (lanes_sum_torch)
template <int W> __device__ __forceinline__ float group_sum_asc(float v) {
    static_assert(W == 1 || W == 2 || W == 4, "quad permutes only");
    if constexpr (W >= 2) AR_DPP_STEP("v_add_f32", "quad_perm:[1,0,3,2]", v);
    if constexpr (W >= 4) AR_DPP_STEP("v_add_f32", "quad_perm:[2,3,0,1]", v);
    return v;
}
#undef AR_GROUP_REDUCE
#undef AR_DPP_STEP

// ---- per-group INT scale / zero-point (runs once per group, off the streaming path) -----------------------------
// Mirrors auto_round/data_type/int.py:221-227 (sym) and :283-293 (asym) with the dtype choreography of
// SURVEY App. A.1/A.2; min/max scale are clamped to [lo,hi] as WrapperLinear._qdq_weight does (wrapper.py:257-259).
// `tensor / python_scalar` on the GPU is not a division: ATen's CUDA / HIP kernel multiplies by the reciprocal computed once in the
// op-math type (ATen/native/cuda/BinaryDivTrueKernel.cu: "compute a * reciprocal(b); this may lose one bit of precision").  The
// reference's `(wmax - wmin) / maxq` and the `/ maxq` of its backward are such divisions; for the asymmetric schemes maxq = 2^bits - 1
// is not a power of two, so x * fl(1 / maxq) and x / maxq differ in the last bit now and then -- enough to move an fp16 scale that
// sits on a rounding tie (1e-5 of the groups) and the low bit of 30 % of the scale gradients (tests/asym_grad_probe.py).  The kernels
// follow the GPU semantics (AR_TORCH_GPU_SCALAR_DIV = 1: what the reference computes when it runs on this device); 0 = the CPU
// semantics (true division) the CPU-generated golden vectors carry.  Symmetric schemes (maxq a power of two) are the same either way.
#ifndef AR_TORCH_GPU_SCALAR_DIV
#define AR_TORCH_GPU_SCALAR_DIV 1
#endif
__device__ __forceinline__ float div_py_scalar(float x, float b) {
#if AR_TORCH_GPU_SCALAR_DIV
    return x * (1.0f / b);
#else
    return x / b;
#endif
}

struct IntCfg {
    int bits, sym, s_dt, w_dt;   // sym: 0 asym, 1 sym ("full range"), 2 sym with a searched init scale in the wmax slot
    float thresh;      // q_scale_thresh
    float lo, hi;      // min/max-scale bounds
};
struct GroupQ {
    float ms, Ms, wmin, wmax;   // clamped scales, float(wmin0/wmax0)
    float a, b;                 // sym: -(wmin*ms), wmax*Ms ; asym: lo, hi
    float sgn;                  // sym only (+1 / -1)
    float s_raw, s, zp;         // scale before / after the threshold clamp ; zero point (sym: maxq, informational)
};
__device__ __forceinline__ void group_scale(const IntCfg& c, float wmin, float wmax, float ms_raw, float Ms_raw, GroupQ& q) {
    q.ms = clamp3(ms_raw, c.lo, c.hi);
    q.Ms = clamp3(Ms_raw, c.lo, c.hi);
    q.wmin = wmin; q.wmax = wmax;
    const float t = round_to_rt(c.s_dt, c.thresh);
    if (c.sym == 2) {
        // algorithm-extension sym path: the searched init_scale rides in the wmax slot (data_type/int.py:201-216)
        q.a = 0.f;
        q.b = wmax * q.Ms;
        q.sgn = 1.f;
        q.s_raw = round_to_rt(c.s_dt, q.b);
        q.s = (q.s_raw < 0.f) ? ((q.s_raw > -t) ? -t : q.s_raw) : ((q.s_raw < t) ? t : q.s_raw);
        q.zp = (float)(1 << (c.bits - 1));
    } else if (c.sym) {
        const float maxq = (float)(1 << (c.bits - 1));
        q.a = -(wmin * q.ms);
        q.b = wmax * q.Ms;
        q.sgn = (q.b < q.a) ? 1.f : -1.f;
        const float m = (q.a > q.b) ? q.a : q.b;
        q.s_raw = round_to_rt(c.s_dt, div_py_scalar(q.sgn * m, maxq));
        q.s = (q.s_raw < 0.f) ? ((q.s_raw > -t) ? -t : q.s_raw) : ((q.s_raw < t) ? t : q.s_raw);
        q.zp = maxq;
    } else {
        const float maxq = (float)((1 << c.bits) - 1);
        q.a = wmin * q.ms;
        q.b = wmax * q.Ms;
        q.sgn = 1.f;
        q.s_raw = round_to_rt(c.s_dt, div_py_scalar(q.b - q.a, maxq));
        q.s = (q.s_raw < t) ? t : q.s_raw;
        q.zp = __builtin_rintf((-q.a) / q.s);
    }
}

// error plumbing for the C ABI
inline int launch_status() { return (int)hipGetLastError(); }

// ---- optional per-launch device timing of the hot kernels (ar_profile_*, csrc/ar_prof.hip) ---------------------------
// When profiling is on, a hot launch goes through hipExtLaunchKernelGGL with a start/stop event pair: the events carry the
// dispatch's OWN begin/end timestamps (what rocprofv3 --kernel-trace reports), unlike an event bracket recorded around
// the launch, whose marker packets cost more than an 11 us kernel.  Off (the default): the plain launch, zero overhead.
bool prof_on();
void prof_events(int kernel_id, int64_t units, hipEvent_t* start, hipEvent_t* stop);
#define AR_LAUNCH_PROF(KID, UNITS, KERNEL, GRID, BLOCK, LDS, ST, ...)                                                    \
    do {                                                                                                               \
        if (ar::prof_on()) {                                                                                           \
            hipEvent_t e0_, e1_;                                                                                       \
            ar::prof_events((KID), (int64_t)(UNITS), &e0_, &e1_);                                                      \
            hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), (LDS), (ST), e0_, e1_, 0, __VA_ARGS__);             \
        } else {                                                                                                       \
            hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), (LDS), (ST), __VA_ARGS__);                             \
        }                                                                                                              \
    } while (0)

}  // namespace ar
