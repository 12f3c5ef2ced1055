"""ctypes/numpy front-end of the CPU oracle (oracle/ar_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by anything under auto_round_amd/ (the product path has no CPU fallback).

Tensors cross this boundary as numpy arrays; 16-bit floats travel as their raw uint16 bit patterns
(`to_bits` / `from_bits` convert torch tensors).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libar_oracle.so")

DT_BF16, DT_F16, DT_F32 = 0, 1, 2
_DT_NAMES = {"bf16": DT_BF16, "bfloat16": DT_BF16, "f16": DT_F16, "float16": DT_F16, "fp16": DT_F16,
             "f32": DT_F32, "float32": DT_F32, "fp32": DT_F32}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds). Returns the path of the shared object."""
    src = os.path.join(_HERE, "ar_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_nvfp4_global_scale.restype = ctypes.c_float
        _lib.oracle_f16_to_f32.restype = ctypes.c_float
        _lib.oracle_e4m3_to_f32.restype = ctypes.c_float
        _lib.oracle_f32_to_bf16.restype = ctypes.c_uint16
        _lib.oracle_f32_to_f16.restype = ctypes.c_uint16
        _lib.oracle_f32_to_e4m3.restype = ctypes.c_uint8
    return _lib


def dt_code(dt) -> int:
    if isinstance(dt, int):
        return dt
    name = str(dt).replace("torch.", "")
    return _DT_NAMES[name]


def np_dtype(code: int):
    return np.float32 if code == DT_F32 else np.uint16


def to_bits(t):
    """torch tensor (bf16/f16/f32) -> contiguous numpy array (uint16 bit patterns for 16-bit floats)."""
    import torch

    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.to(torch.float32).numpy().copy()


def from_bits(a: np.ndarray, dt):
    """numpy array produced by the oracle -> torch tensor of dtype dt."""
    import torch

    code = dt_code(dt)
    if code == DT_F32:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    tdt = torch.bfloat16 if code == DT_BF16 else torch.float16
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(tdt)


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(x):
    return ctypes.c_float(float(x))


def set_scalar_div_mode(gpu: bool) -> None:
    """False (default): `tensor / python_scalar` is a division, as torch evaluates it on the CPU (the golden vectors' semantics);
    True: a multiplication by fl(1 / scalar), as torch's GPU kernels evaluate it -- what the reference computes when it runs on
    the GPU and what the HIP kernels follow.  Only the asymmetric INT schemes (maxq = 2^bits - 1) can tell the two apart."""
    lib().oracle_set_scalar_div_mode(int(bool(gpu)))


def group_minmax(W: np.ndarray, w_dt: int, G: int, gs: int):
    wmin = np.empty(G, dtype=np_dtype(w_dt))
    wmax = np.empty(G, dtype=np_dtype(w_dt))
    lib().oracle_group_minmax(_p(W), w_dt, ctypes.c_int64(G), gs, _p(wmin), _p(wmax))
    return wmin, wmax


def qdq_int_fwd(W, V, wmin, wmax, min_s, max_s, G, gs, bits, sym, w_dt=DT_BF16, s_dt=DT_F16,
                q_thresh=1e-5, bounds=(0.0, 1.0)):
    """-> (Wq bits [G*gs], scale bits [G], zp f32 [G])"""
    Wq = np.empty(G * gs, dtype=np_dtype(w_dt))
    scale = np.empty(G, dtype=np_dtype(s_dt))
    zp = np.empty(G, dtype=np.float32)
    lib().oracle_qdq_int_fwd(_p(W), _p(V), _p(wmin), _p(wmax), _p(min_s), _p(max_s), ctypes.c_int64(G),
                             gs, bits, int(sym), w_dt, s_dt, _f(q_thresh), _f(bounds[0]), _f(bounds[1]),
                             _p(Wq), _p(scale), _p(zp))
    return Wq, scale, zp


def qdq_int_bwd(dWq, W, V, wmin, wmax, min_s, max_s, G, gs, bits, sym, w_dt=DT_BF16, s_dt=DT_F16,
                q_thresh=1e-5, bounds=(0.0, 1.0)):
    """-> (dV f32 [G*gs], dmin f32 [G], dmax f32 [G])"""
    dV = np.empty(G * gs, dtype=np.float32)
    dmin = np.empty(G, dtype=np.float32)
    dmax = np.empty(G, dtype=np.float32)
    lib().oracle_qdq_int_bwd(_p(dWq), _p(W), _p(V), _p(wmin), _p(wmax), _p(min_s), _p(max_s),
                             ctypes.c_int64(G), gs, bits, int(sym), w_dt, s_dt, _f(q_thresh), _f(bounds[0]),
                             _f(bounds[1]), _p(dV), _p(dmin), _p(dmax))
    return dV, dmin, dmax


def sign_sgd(p: np.ndarray, g: np.ndarray, lr: float) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.float32).copy()
    g = np.ascontiguousarray(g, dtype=np.float32)
    lib().oracle_sign_sgd(_p(p), _p(g), ctypes.c_int64(p.size), _f(lr))
    return p


def mse_fwd_bwd(pred, ref, act_dt=DT_BF16, gout=1000.0, token_mask=None, row_len=0):
    n = pred.size
    loss = np.zeros(1, dtype=np.float32)
    dpred = np.empty(n, dtype=np_dtype(act_dt))
    tm = None if token_mask is None else np.ascontiguousarray(token_mask, dtype=np.uint8)
    lib().oracle_mse_fwd_bwd(_p(pred), _p(ref), ctypes.c_int64(n), act_dt, _f(gout), _p(loss), _p(dpred), _p(tm),
                             ctypes.c_int64(row_len))
    return float(loss[0]), dpred


def pack_int(Wq, scale, zp, out_f, in_f, gs, bits, w_dt=DT_BF16, s_dt=DT_F16, zp_off=1):
    """zp: python number (sym) or f32 array [out, n_groups]. -> (qweight, qzeros, scales_t bits)"""
    n_groups = (in_f + gs - 1) // gs
    qweight = np.empty((in_f // 32 * bits, out_f), dtype=np.int32)
    qzeros = np.empty((n_groups, out_f // 32 * bits), dtype=np.int32)
    scales_t = np.empty((n_groups, out_f), dtype=np.uint16)
    zt, zs = (None, float(zp)) if not isinstance(zp, np.ndarray) else (np.ascontiguousarray(zp, np.float32), 0.0)
    lib().oracle_pack_int(_p(Wq), _p(scale), _p(zt), _f(zs), ctypes.c_int64(out_f), ctypes.c_int64(in_f), gs,
                          bits, w_dt, s_dt, int(zp_off), _p(qweight), _p(qzeros), _p(scales_t))
    return qweight, qzeros, scales_t


def cast_to_fp4(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_cast_to_fp4(_p(x), ctypes.c_int64(x.size), _p(out))
    return out


def mx_quant_element_fp4(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_mx_quant_element_fp4(_p(x), ctypes.c_int64(x.size), _p(out))
    return out


def _init(init_scale):
    """scalar or per-group init_scale -> (c_float scalar, fp32 array or None)"""
    if isinstance(init_scale, np.ndarray):
        return _f(1.0), np.ascontiguousarray(init_scale.reshape(-1), dtype=np.float32)
    return _f(init_scale), None


def qdq_mxfp4_fwd(W, V, max_s, G, gs=32, w_dt=DT_BF16, init_scale=1.0, bounds=(0.0, 1.0)):
    Wq = np.empty(G * gs, dtype=np_dtype(w_dt))
    e = np.empty(G, dtype=np_dtype(w_dt))
    i_s, i_a = _init(init_scale)
    lib().oracle_qdq_mxfp4_fwd(_p(W), _p(V), _p(max_s), i_s, _p(i_a), ctypes.c_int64(G), gs, w_dt,
                               _f(bounds[0]), _f(bounds[1]), _p(Wq), _p(e))
    return Wq, e


def nvfp4_global_scale(W, w_dt=DT_BF16) -> float:
    return float(lib().oracle_nvfp4_global_scale(_p(W), ctypes.c_int64(W.size), w_dt))


def qdq_nvfp4_fwd(W, V, max_s, global_scale, G, gs=16, w_dt=DT_BF16, init_scale=1.0, bounds=(0.0, 1.0)):
    Wq = np.empty(G * gs, dtype=np_dtype(w_dt))
    sc = np.empty(G, dtype=np.float32)
    i_s, i_a = _init(init_scale)
    lib().oracle_qdq_nvfp4_fwd(_p(W), _p(V), _p(max_s), i_s, _p(i_a), _f(global_scale), ctypes.c_int64(G), gs,
                               w_dt, _f(bounds[0]), _f(bounds[1]), _p(Wq), _p(sc))
    return Wq, sc


def pack_fp4(W, scale, out_f, in_f, gs, mode, w_dt=DT_BF16, global_scale=1.0):
    """mode 0 = MXFP4 (scale = exponents in w_dt bits), 1 = NVFP4 (scale = f32 e4m3 values)."""
    packed = np.empty((out_f, in_f // 2), dtype=np.uint8)
    sb = np.empty((out_f, in_f // gs), dtype=np.uint8)
    lib().oracle_pack_fp4(_p(W), _p(scale), _f(global_scale), ctypes.c_int64(out_f), ctypes.c_int64(in_f), gs,
                          mode, w_dt, _p(packed), _p(sb))
    return packed, sb


def qdq_fp4_bwd(dXq, W, V, max_s, G, gs, mode, w_dt=DT_BF16, init_scale=1.0, global_scale=1.0, bounds=(0.0, 1.0)):
    """-> (dV f32 [G*gs], dmax f32 [G]) : autograd-equivalent gradients of the fp4 fake-quant."""
    dV = np.empty(G * gs, dtype=np.float32)
    dmax = np.empty(G, dtype=np.float32)
    i_s, i_a = _init(init_scale)
    lib().oracle_qdq_fp4_bwd(_p(dXq), _p(W), _p(V), _p(max_s), i_s, _p(i_a), _f(global_scale), ctypes.c_int64(G), gs,
                             mode, w_dt, _f(bounds[0]), _f(bounds[1]), _p(dV), _p(dmax))
    return dV, dmax


def fp4_act_bwd(dXq, X, G, gs, mode, x_dt=DT_BF16, global_scale=1.0):
    """-> dX bits [G*gs]: gradient of the dynamic activation fake-quant w.r.t. its input."""
    dX = np.empty(G * gs, dtype=np_dtype(x_dt))
    lib().oracle_fp4_act_bwd(_p(dXq), _p(X), _f(global_scale), ctypes.c_int64(G), gs, mode, x_dt, _p(dX))
    return dX


def pack_awq(Wq, scale, zp, out_f, in_f, gs, w_dt=DT_BF16, s_dt=DT_F16):
    qweight = np.empty((in_f, out_f // 8), dtype=np.int32)
    qzeros = np.empty((in_f // gs, out_f // 8), dtype=np.int32)
    scales_t = np.empty((in_f // gs, out_f), dtype=np.uint16)
    zt, zs = (None, float(zp)) if not isinstance(zp, np.ndarray) else (np.ascontiguousarray(zp, np.float32), 0.0)
    lib().oracle_pack_awq(_p(Wq), _p(scale), _p(zt), _f(zs), ctypes.c_int64(out_f), ctypes.c_int64(in_f), gs, w_dt, s_dt,
                          _p(qweight), _p(qzeros), _p(scales_t))
    return qweight, qzeros, scales_t


def fp4_candidates(mode: int) -> np.ndarray:
    """Candidate coefficients in the reference's evaluation order (mxfp.py:147 ; nvfp.py:358-362)."""
    if mode == 0:
        return np.array([1.0, 0.5, 2.0], dtype=np.float32)
    return np.array([1.0] + [v / 100.0 for v in range(50, 152) if v != 100], dtype=np.float32)


def search_fp4_scale(X, G, gs, mode, x_dt=DT_BF16, qw_row=None, groups_per_row=0, global_scale=1.0):
    cand = fp4_candidates(mode)
    best = np.empty(G, dtype=np.float32)
    qw = None if qw_row is None else np.ascontiguousarray(qw_row, dtype=np.float32)
    lib().oracle_search_fp4_scale(_p(X), _p(qw), ctypes.c_int64(groups_per_row), _f(global_scale), _p(cand), len(cand),
                                  ctypes.c_int64(G), gs, mode, x_dt, _p(best))
    return best


def outlier_mse_fwd_bwd(pred, ref, act_dt=DT_BF16, gout=1000.0, topk=None, token_mask=None, row_len=0):
    n = pred.size
    topk = max(1, n // 1000) if topk is None else topk
    loss = np.zeros(1, dtype=np.float32)
    dropped = np.zeros(1, dtype=np.int64)
    dpred = np.empty(n, dtype=np_dtype(act_dt))
    tm = None if token_mask is None else np.ascontiguousarray(token_mask, dtype=np.uint8)
    lib().oracle_outlier_mse_fwd_bwd(_p(pred), _p(ref), ctypes.c_int64(n), act_dt, _f(gout), ctypes.c_int64(topk), _p(tm),
                                     ctypes.c_int64(row_len), _p(loss), _p(dpred), _p(dropped))
    return float(loss[0]), dpred, int(dropped[0])


def int_search_candidates(bits: int, ratio: float = 0.75) -> np.ndarray:
    """The candidate numerators of search_scales (auto_round/data_type/int.py:49-64), the initial nmax first."""
    nmax = int(2.0 ** (bits - 1))
    if bits == 2:
        search_min, step = 18 * 5, 0.01
    else:
        grid = 200
        search_min = nmax * ratio
        step = search_min / grid * 2
        search_min = int(search_min / step)
    c = [float(nmax)] + [nmax - step * i for i in range(-search_min, search_min + 1) if i != 0]
    return np.asarray(c, dtype=np.float32)


def search_int_scale(X, G, gs, bits, x_dt=DT_BF16, qw_row=None, groups_per_row=0, q_thresh=1e-5):
    cand = int_search_candidates(bits)
    raw = np.empty(G, dtype=np_dtype(x_dt))
    init = np.empty(G, dtype=np_dtype(x_dt))
    qw = None if qw_row is None else np.ascontiguousarray(qw_row, dtype=np.float32)
    lib().oracle_search_int_scale(_p(X), _p(qw), ctypes.c_int64(groups_per_row), _p(cand), len(cand), ctypes.c_int64(G), gs,
                                  bits, x_dt, _f(q_thresh), _p(raw), _p(init))
    return raw, init


def int_act_fwd(X, G, gs, bits, a_dt=DT_BF16, s_dt=DT_F16, q_thresh=1e-5):
    Xq = np.empty(G * gs, dtype=np_dtype(a_dt))
    scale = np.empty(G, dtype=np_dtype(s_dt))
    lib().oracle_int_act_fwd(_p(X), ctypes.c_int64(G), gs, bits, a_dt, s_dt, _f(q_thresh), _p(Xq), _p(scale))
    return Xq, scale


def int_act_bwd(dXq, X, G, gs, bits, a_dt=DT_BF16, s_dt=DT_F16, q_thresh=1e-5):
    dX = np.empty(G * gs, dtype=np_dtype(a_dt))
    lib().oracle_int_act_bwd(_p(dXq), _p(X), ctypes.c_int64(G), gs, bits, a_dt, s_dt, _f(q_thresh), _p(dX))
    return dX


def int_act_asym_fwd(X, G, gs, bits, a_dt=DT_BF16, s_dt=DT_F16, q_thresh=1e-5):
    Xq = np.empty(G * gs, dtype=np_dtype(a_dt))
    scale = np.empty(G, dtype=np_dtype(s_dt))
    zp = np.empty(G, dtype=np.float32)
    lib().oracle_int_act_asym_fwd(_p(X), ctypes.c_int64(G), gs, bits, a_dt, s_dt, _f(q_thresh), _p(Xq), _p(scale), _p(zp))
    return Xq, scale, zp


def int_act_asym_bwd(dXq, X, G, gs, bits, a_dt=DT_BF16, s_dt=DT_F16, q_thresh=1e-5):
    dX = np.empty(G * gs, dtype=np_dtype(a_dt))
    lib().oracle_int_act_asym_bwd(_p(dXq), _p(X), ctypes.c_int64(G), gs, bits, a_dt, s_dt, _f(q_thresh), _p(dX))
    return dX
