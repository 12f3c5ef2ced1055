"""Thin torch-tensor front end of the C ABI: every function here validates its tensors, hands raw device pointers
and the CURRENT torch HIP stream to `libar_mi355x.so`, and returns.  No arithmetic happens in Python.

PyTorch is plumbing only (device memory, streams); the kernels are the product.

Which reference interface each wrapper stands for (paths relative to the auto-round tree; the C entry point of the same name
in include/ar_mi355x.h carries the same citation):
  group_minmax / group_absmax      weight_min/max of WrapperLinear._init_tuning_params_and_quant_func   auto_round/wrapper.py:154-164
  qdq_int_fwd / qdq_int_bwd        quant_tensor_sym / quant_tensor_asym and their autograd              auto_round/data_type/int.py:165-298
  sign_sgd_ / qdq_int_bwd_sgd_     SignSGD._single_tensor_sgd (+ collect_best_params)                   sign_round/sign_sgd.py:356-389, compressors/utils.py:205-217
  mse_loss_fwd_bwd                 _get_loss + (loss * 1000).backward()                                 sign_round/quantizer.py:127-158,789-803
  outlier_mse_loss_fwd_bwd         SignRoundV2Quantizer._get_loss                                       sign_roundv2/quantizer.py:362-399
  best_loss_update                 best-loss bookkeeping                                                sign_round/quantizer.py:508-521
  gather_rows                      minibatch gather of the cached activations                           algorithms/block_runner.py:368-422
  pack_int / pack_awq / pack_fp4   QuantLinear.pack, WQLinear_GEMM.from_linear, qlinear_fp pack         auto_round_extension/torch/qlinear_torch[_zp].py, export/export_to_awq/utils.py:139-274, export/export_to_autoround/qlinear_fp.py:141-265
  qdq_fp4_fwd / qdq_fp4_bwd_sgd_   quant_mx / nv_fp4 (+ autograd + SignSGD)                             data_type/mxfp.py:233-291, data_type/nvfp.py:56-98
  fp4_act_bwd, qdq_int_act_fwd, int_act_bwd   WrapperLinear._qdq_act and its autograd                   auto_round/wrapper.py:295-321
  search_int_scale / search_fp4_scale          search_scales, search_mx_scale, search_nvfp4_scale       data_type/int.py:24-86, mxfp.py:103-170, nvfp.py:329-386
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import AR_DT_BF16, AR_DT_F16, AR_DT_F32, check, load

_DT = {torch.bfloat16: AR_DT_BF16, torch.float16: AR_DT_F16, torch.float32: AR_DT_F32}


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {dtype}; expected bf16/f16/f32") from None


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.Mi355xLibraryError(
            f"{name} is on {t.device}: the MI355X path only runs on a HIP device and has no CPU fallback")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


import contextlib as _contextlib

_NULLCTX = _contextlib.nullcontext()


class _Ptr(int):
    """A device address that remembers which HIP device it lives on (ctypes takes it as the integer it is): `_launch` derives the
    launch device from its own arguments -- no module-level bookkeeping that an exception between `_p()` and `_launch()` could
    leave stale, nothing shared between threads."""
    dev: int = -1


def _p(t: Optional[torch.Tensor], name: str = "tensor"):
    if t is None:
        return None
    p = _Ptr(_dev(t, name).data_ptr())
    p.dev = t.device.index
    return p


def _launch(entry: str, *args):
    """One C-ABI call on the stream of the device that OWNS the tensors (not whatever device happens to be current):
    every pointer of a launch must live on one HIP device; if that device is not the current one the call runs under a
    device guard, so `SignRoundQuantizer(device="cuda:1")` works while cuda:0 is current."""
    devs = {a.dev for a in args if isinstance(a, _Ptr)}
    if len(devs) != 1:
        raise _lib.Mi355xLibraryError(f"{entry}: tensors live on HIP devices {sorted(devs)}; one launch needs one device")
    (dev,) = devs
    fn = getattr(load(), entry)
    if dev == torch.cuda.current_device():
        return check(fn(*args, torch.cuda.current_stream().cuda_stream), entry)
    with torch.cuda.device(dev):
        return check(fn(*args, torch.cuda.current_stream(dev).cuda_stream), entry)


def group_minmax(W: torch.Tensor, gs: int):
    """-> (wmin, wmax) in W.dtype, [numel/gs].  (wrapper.py:154-164)"""
    G = W.numel() // gs
    wmin = torch.empty(G, dtype=W.dtype, device=W.device)
    wmax = torch.empty(G, dtype=W.dtype, device=W.device)
    _launch("ar_group_minmax", _p(W, "W"), _p(wmin), _p(wmax), G, gs, dt_code(W.dtype))
    return wmin, wmax


def group_absmax(W: torch.Tensor, gs: int, want_tensor_max: bool = False, want_groups: bool = True):
    G = W.numel() // gs
    am = torch.empty(G, dtype=torch.float32, device=W.device) if want_groups else None
    tm = torch.zeros(1, dtype=torch.float32, device=W.device) if want_tensor_max else None
    _launch("ar_group_absmax", _p(W, "W"), _p(am), _p(tm), G, gs, dt_code(W.dtype))
    return am, tm


def qdq_int_fwd(W, V, wmin, wmax, min_s, max_s, *, gs, bits, sym, scale_dtype=torch.float16, q_thresh=1e-5,
                bounds=(0.0, 1.0), out=None, want_scale=False):
    """Fake-quant forward over a flat group array. -> Wq [, scale, zp]"""
    G = W.numel() // gs
    Wq = out if out is not None else torch.empty_like(W)
    scale = torch.empty(G, dtype=scale_dtype, device=W.device) if want_scale else None
    zp = torch.empty(G, dtype=torch.float32, device=W.device) if want_scale else None
    _launch("ar_qdq_int_fwd", _p(W, "W"), _p(V, "V"), _p(wmin, "wmin"), _p(wmax, "wmax"), _p(min_s, "min_scale"),
                                _p(max_s, "max_scale"), _p(Wq, "Wq"), _p(scale), _p(zp), G, gs, bits, int(sym),
                                dt_code(W.dtype), dt_code(scale_dtype), q_thresh, bounds[0], bounds[1])
    return (Wq, scale, zp) if want_scale else Wq


def qdq_int_bwd(dWq, W, V, wmin, wmax, min_s, max_s, *, gs, bits, sym, scale_dtype=torch.float16, q_thresh=1e-5,
                bounds=(0.0, 1.0)):
    """Unfused backward -> (dV, dmin, dmax) fp32."""
    G = W.numel() // gs
    dV = torch.empty(W.numel(), dtype=torch.float32, device=W.device)
    dmin = torch.empty(G, dtype=torch.float32, device=W.device)
    dmax = torch.empty(G, dtype=torch.float32, device=W.device)
    _launch("ar_qdq_int_bwd", _p(dWq, "dWq"), _p(W, "W"), _p(V, "V"), _p(wmin), _p(wmax), _p(min_s), _p(max_s), _p(dV),
                                _p(dmin), _p(dmax), G, gs, bits, int(sym), dt_code(W.dtype), dt_code(scale_dtype),
                                q_thresh, bounds[0], bounds[1])
    return dV, dmin, dmax


def sign_sgd_(p: torch.Tensor, g: torch.Tensor, lr_dev: torch.Tensor):
    _launch("ar_sign_sgd", _p(p, "param"), _p(g, "grad"), p.numel(), _p(lr_dev, "lr"))
    return p


def qdq_int_bwd_sgd_(dWq, W, V, wmin, wmax, min_s, max_s, *, gs, bits, sym, lr_v, lr_mm, tune_minmax=True,
                     scale_dtype=torch.float16, q_thresh=1e-5, bounds=(0.0, 1.0), snapshot_flag=None, best_V=None,
                     best_min=None, best_max=None, Wq_next=None):
    """Fused backward + sign-SGD (in place on V, min_s, max_s) [+ snapshot] [+ next forward]."""
    G = W.numel() // gs
    _launch("ar_qdq_int_bwd_sgd", _p(dWq, "dWq"), _p(W, "W"), _p(V, "V"), _p(wmin), _p(wmax), _p(min_s), _p(max_s), G,
                                    gs, bits, int(sym), dt_code(W.dtype), dt_code(scale_dtype), q_thresh, bounds[0],
                                    bounds[1], _p(lr_v, "lr_v"), _p(lr_mm, "lr_mm"), int(tune_minmax),
                                    _p(snapshot_flag), _p(best_V), _p(best_min), _p(best_max), _p(Wq_next))


_mse_ws = {}


def mse_workspace(device) -> torch.Tensor:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    ws = _mse_ws.get(key)
    if ws is None:
        ws = torch.empty(load().ar_mse_workspace_bytes(), dtype=torch.uint8, device=f"cuda:{key}")
        _mse_ws[key] = ws
    return ws


def mse_loss_fwd_bwd(pred, ref, *, dpred=None, loss_out=None, loss_accum=None, accum_scale=1.0, grad_scale=1000.0,
                     token_mask=None):
    """loss = mean((pred-ref)^2) ; dpred = d(loss*grad_scale)/dpred in pred.dtype. -> (loss_out [1] f32, dpred).
    token_mask: optional uint8 [tokens] valid-token mask (rows of pred.shape[-1] elements)."""
    if dpred is None:
        dpred = torch.empty_like(pred)
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=pred.device)
    _launch("ar_mse_loss_fwd_bwd", _p(pred, "pred"), _p(ref, "ref"), _p(dpred), _p(loss_out), _p(loss_accum),
                                     accum_scale, pred.numel(), dt_code(pred.dtype), grad_scale, _p(token_mask, "token_mask"),
                                     pred.shape[-1] if token_mask is not None else 0, _p(mse_workspace(pred.device)))
    return loss_out, dpred


_ol_ws = {}


def outlier_mse_loss_fwd_bwd(pred, ref, *, topk=None, dpred=None, loss_out=None, loss_accum=None, accum_scale=1.0,
                             grad_scale=1000.0, token_mask=None):
    """Outlier-suppressed MSE of the algorithm extension: the topk = max(1, n//1000) largest |pred-ref| are dropped."""
    n = pred.numel()
    if topk is None:
        topk = max(1, n // 1000)
    if dpred is None:
        dpred = torch.empty_like(pred)
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=pred.device)
    key = pred.device.index if pred.device.index is not None else torch.cuda.current_device()
    ws = _ol_ws.get(key)
    if ws is None:
        ws = torch.empty(load().ar_outlier_loss_workspace_bytes(), dtype=torch.uint8, device=pred.device)
        _ol_ws[key] = ws
    _launch("ar_outlier_mse_loss_fwd_bwd", _p(pred, "pred"), _p(ref, "ref"), _p(dpred), _p(loss_out), _p(loss_accum),
                                             accum_scale, n, dt_code(pred.dtype), grad_scale, _p(token_mask),
                                             pred.shape[-1] if token_mask is not None else 0, topk, _p(ws))
    return loss_out, dpred


def best_loss_update(total_loss, state, istate, it: int, iter_dev=None, loss_hist=None):
    """iter_dev (int32 [1]): take the iteration number from the device and advance it (captured iterations);
    loss_hist (fp32 [iters]): record this iteration's loss"""
    _launch("ar_best_loss_update", _p(total_loss), _p(state), _p(istate), it, _p(iter_dev), _p(loss_hist))


def iter_begin(iter_dev, sched, cur_idx, lr_table, lr_out, iters: int):
    """cur_idx <- sched[*iter_dev], lr_out[k] <- lr_table[k, *iter_dev]: the host side of an iteration, from device tables"""
    batch = cur_idx.numel()
    n_lr = lr_out.numel()
    if sched.dtype != torch.int64 or cur_idx.dtype != torch.int64 or iter_dev.dtype != torch.int32:
        raise TypeError("iter_begin: sched / cur_idx int64, iter_dev int32")
    if sched.numel() != iters * batch or lr_table.numel() != n_lr * iters or lr_table.dtype != torch.float32 or lr_out.dtype != torch.float32:
        raise ValueError("iter_begin: table shapes do not match (iters, batch, n_lr)")
    _launch("ar_iter_begin", _p(iter_dev, "iter_dev"), _p(sched, "sched"), batch, _p(cur_idx), _p(lr_table), n_lr, iters, _p(lr_out))


def gather_rows(src: torch.Tensor, idx_dev: torch.Tensor, out: Optional[torch.Tensor] = None):
    """out[j] = src[idx[j]] along dim 0 (rows must be a multiple of 16 bytes)."""
    row_bytes = src[0].numel() * src.element_size()
    n = idx_dev.numel()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if idx_dev.dtype != torch.int64:
        raise TypeError("idx must be int64")
    _launch("ar_gather_rows", _p(src, "src"), _p(idx_dev, "idx"), _p(out, "out"), n, row_bytes)
    return out


def pack_int(Wq2d: torch.Tensor, scale2d: torch.Tensor, zp, *, gs, bits, zp_off=1):
    """-> (qweight int32 [in/32*bits, out], qzeros int32 [in/gs, out/32*bits], scales fp16 [in/gs, out])"""
    out_f, in_f = Wq2d.shape
    ng = (in_f + gs - 1) // gs
    dev = Wq2d.device
    qweight = torch.empty((in_f // 32 * bits, out_f), dtype=torch.int32, device=dev)
    qzeros = torch.empty((ng, out_f // 32 * bits), dtype=torch.int32, device=dev)
    scales_t = torch.empty((ng, out_f), dtype=torch.float16, device=dev)
    if isinstance(zp, torch.Tensor):
        zt, zs = zp.to(device=dev, dtype=torch.float32).contiguous(), 0.0
    else:
        zt, zs = None, float(zp)
    _launch("ar_pack_int", _p(Wq2d, "Wq"), _p(scale2d, "scale"), _p(zt), zs, out_f, in_f, gs, bits, dt_code(Wq2d.dtype),
                             dt_code(scale2d.dtype), zp_off, _p(qweight), _p(qzeros), _p(scales_t))
    return qweight, qzeros, scales_t


def pack_awq(Wq2d: torch.Tensor, scale2d: torch.Tensor, zp, *, gs):
    """AWQ GEMM container (4-bit). -> (qweight int32 [in, out/8], qzeros int32 [in/gs, out/8], scales fp16 [in/gs, out])"""
    out_f, in_f = Wq2d.shape
    dev = Wq2d.device
    qweight = torch.empty((in_f, out_f // 8), dtype=torch.int32, device=dev)
    qzeros = torch.empty((in_f // gs, out_f // 8), dtype=torch.int32, device=dev)
    scales_t = torch.empty((in_f // gs, out_f), dtype=torch.float16, device=dev)
    if isinstance(zp, torch.Tensor):
        zt, zs = zp.to(device=dev, dtype=torch.float32).contiguous(), 0.0
    else:
        zt, zs = None, float(zp)
    _launch("ar_pack_awq", _p(Wq2d, "Wq"), _p(scale2d, "scale"), _p(zt), zs, out_f, in_f, gs, dt_code(Wq2d.dtype),
                             dt_code(scale2d.dtype), _p(qweight), _p(qzeros), _p(scales_t))
    return qweight, qzeros, scales_t


def qdq_fp4_fwd(X, V, absmax, max_s, *, mode, gs, init_scale=1.0, global_scale=None, bounds=(0.0, 1.0), out=None,
                want_scale=False, init_scale_dev=None):
    G = X.numel() // gs
    Xq = out if out is not None else torch.empty_like(X)
    scale = None
    if want_scale:
        scale = torch.empty(G, dtype=X.dtype if mode == 0 else torch.float32, device=X.device)
    _launch("ar_qdq_fp4_fwd", _p(X, "X"), _p(V), _p(absmax), _p(max_s), init_scale, _p(init_scale_dev), _p(global_scale), _p(Xq), _p(scale),
                                G, gs, mode, dt_code(X.dtype), bounds[0], bounds[1])
    return (Xq, scale) if want_scale else Xq


def qdq_fp4_bwd_sgd_(dXq, X, V, absmax, max_s, *, mode, gs, init_scale=1.0, global_scale=None, bounds=(0.0, 1.0), lr_v=None,
                     lr_mm=None, tune_minmax=True, snapshot_flag=None, best_V=None, best_max=None, want_grads=False,
                     init_scale_dev=None):
    """fp4 backward: with lr_v/lr_mm fused sign-SGD in place on V / max_s; want_grads also returns (dV, dmax)."""
    G = X.numel() // gs
    dV = torch.empty(X.numel(), dtype=torch.float32, device=X.device) if want_grads else None
    dmax = torch.empty(G, dtype=torch.float32, device=X.device) if want_grads else None
    _launch("ar_qdq_fp4_bwd_sgd", _p(dXq, "dXq"), _p(X, "X"), _p(V), _p(absmax, "absmax"), _p(max_s), init_scale,
                                    _p(init_scale_dev), _p(global_scale), G, gs, mode, dt_code(X.dtype), bounds[0], bounds[1], _p(lr_v),
                                    _p(lr_mm), int(tune_minmax), _p(snapshot_flag), _p(best_V), _p(best_max), _p(dV),
                                    _p(dmax))
    return (dV, dmax) if want_grads else None


def fp4_search_candidates(mode: int):
    """Candidate coefficients in the reference's evaluation order: MXFP4 1, 0.5, 2 (data_type/mxfp.py:147);
    NVFP4 1.0 first, then 0.50 ... 1.51 in steps of 0.01 (data_type/nvfp.py:358-362)."""
    if mode == 0:
        return [1.0, 0.5, 2.0]
    return [1.0] + [v / 100.0 for v in range(50, 152) if v != 100]


def search_fp4_scale(X, absmax, candidates, *, mode, gs, qw_row=None, groups_per_row=0, global_scale=None):
    """Per-group init-scale search over `candidates` (fp32 device tensor, evaluated in order). -> fp32 [numel/gs]"""
    G = X.numel() // gs
    best = torch.empty(G, dtype=torch.float32, device=X.device)
    _launch("ar_search_fp4_scale", _p(X, "X"), _p(absmax, "absmax"), _p(qw_row), groups_per_row, _p(global_scale),
                                     _p(candidates, "candidates"), candidates.numel(), _p(best), G, gs, mode,
                                     dt_code(X.dtype))
    return best


def int_search_candidates(bits: int, ratio: float = 0.75):
    """Candidate numerators of the reference's search_scales grid (auto_round/data_type/int.py:49-64): the plain
    nmax first (its initial best), then nmax - step*i for i in [-search_min, search_min] \\ {0}, as fp32."""
    nmax = int(2.0 ** (bits - 1))
    if bits == 2:
        search_min, step = 18 * 5, 0.01
    else:
        grid = 200
        search_min = nmax * ratio
        step = search_min / grid * 2
        search_min = int(search_min / step)
    return [float(nmax)] + [nmax - step * i for i in range(-search_min, search_min + 1) if i != 0]


_int_cand = {}


def search_int_scale(X, *, gs, bits, qw_row=None, groups_per_row=0, q_thresh=1e-5, want_raw=False):
    """Per-group searched init scale for the sym int path of the algorithm extension -> init_scale [G] in X.dtype."""
    G = X.numel() // gs
    key = (bits, X.device.index)
    cand = _int_cand.get(key)
    if cand is None:
        cand = torch.tensor(int_search_candidates(bits), dtype=torch.float32, device=X.device)
        _int_cand[key] = cand
    init = torch.empty(G, dtype=X.dtype, device=X.device)
    raw = torch.empty(G, dtype=X.dtype, device=X.device) if want_raw else None
    _launch("ar_search_int_scale", _p(X, "X"), _p(qw_row), groups_per_row, _p(cand), cand.numel(), _p(raw), _p(init), G, gs,
                                     bits, dt_code(X.dtype), q_thresh)
    return (raw, init) if want_raw else init


def qdq_int_act_fwd(X, *, gs, bits, sym=True, scale_dtype=torch.float16, q_thresh=1e-5, out=None, want_scale=False):
    """Dynamic INT fake-quant of an activation tensor (symmetric or asymmetric), groups of `gs` along the last dimension."""
    G = X.numel() // gs
    Xq = out if out is not None else torch.empty_like(X)
    scale = torch.empty(G, dtype=scale_dtype, device=X.device) if want_scale else None
    _launch("ar_qdq_int_act_fwd", _p(X, "X"), _p(Xq), _p(scale), G, gs, bits, int(bool(sym)), dt_code(X.dtype), dt_code(scale_dtype), q_thresh)
    return (Xq, scale) if want_scale else Xq


def int_act_bwd(dXq, X, *, gs, bits, sym=True, scale_dtype=torch.float16, q_thresh=1e-5, out=None):
    G = X.numel() // gs
    dX = out if out is not None else torch.empty_like(X)
    _launch("ar_int_act_bwd", _p(dXq, "dXq"), _p(X, "X"), _p(dX), G, gs, bits, int(bool(sym)), dt_code(X.dtype), dt_code(scale_dtype), q_thresh)
    return dX


def fp4_act_bwd(dXq, X, *, mode, gs, global_scale=None, out=None):
    """Gradient of the dynamic activation fake-quant w.r.t. its input (same dtype/shape as X)."""
    dX = out if out is not None else torch.empty_like(X)
    _launch("ar_fp4_act_bwd", _p(dXq, "dXq"), _p(X, "X"), _p(dX), _p(global_scale), X.numel() // gs, gs, mode,
                                dt_code(X.dtype))
    return dX


def pack_fp4(Wq2d, scale, *, mode, gs, global_scale=None):
    out_f, in_f = Wq2d.shape
    packed = torch.empty((out_f, in_f // 2), dtype=torch.uint8, device=Wq2d.device)
    sb = torch.empty((out_f, in_f // gs), dtype=torch.uint8, device=Wq2d.device)
    _launch("ar_pack_fp4", _p(Wq2d, "Wq"), _p(scale, "scale"), _p(global_scale), out_f, in_f, gs, mode,
                             dt_code(Wq2d.dtype), _p(packed), _p(sb))
    return packed, sb


def rmsnorm_fwd(x2d, weight, eps, want_rstd=True):
    """-> (y [rows, H], rstd [rows] fp32 or None)"""
    rows, H = x2d.shape
    y = torch.empty_like(x2d)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device) if want_rstd else None
    _launch("ar_rmsnorm_fwd", _p(x2d, "x"), _p(weight, "weight"), _p(y), _p(rstd), rows, H, float(eps), dt_code(x2d.dtype))
    return y, rstd


def rmsnorm_bwd(dy2d, x2d, weight, rstd, dres=None, out=None):
    """dx = d rmsnorm / dx applied to dy (+ dres) -> [rows, H]"""
    rows, H = x2d.shape
    dx = out if out is not None else torch.empty_like(x2d)
    _launch("ar_rmsnorm_bwd", _p(dy2d, "dy"), _p(x2d, "x"), _p(weight, "weight"), _p(rstd, "rstd"), _p(dres), _p(dx), rows, H,
            dt_code(x2d.dtype))
    return dx


def layernorm_fwd(x2d, weight, bias, eps, want_stats=True):
    """-> (y [rows, H], mean [rows] fp32, rstd [rows] fp32) (stats None unless wanted)"""
    rows, H = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device) if want_stats else None
    _launch("ar_layernorm_fwd", _p(x2d, "x"), _p(weight, "weight"), _p(bias), _p(y), _p(mean), _p(rstd), rows, H, float(eps),
            dt_code(x2d.dtype))
    return y, mean, rstd


def layernorm_bwd(dy2d, x2d, weight, mean, rstd, dres=None, out=None):
    """dx = d layernorm / dx applied to dy (+ dres) -> [rows, H]"""
    rows, H = x2d.shape
    dx = out if out is not None else torch.empty_like(x2d)
    _launch("ar_layernorm_bwd", _p(dy2d, "dy"), _p(x2d, "x"), _p(weight, "weight"), _p(mean, "mean"), _p(rstd, "rstd"), _p(dres), _p(dx),
            rows, H, dt_code(x2d.dtype))
    return dx


def headnorm_fwd(qkv2d, wq, wk, hq, hkv, d, eps):
    """q / k heads of qkv2d [tokens, (hq + 2 hkv) d] normalised per head (Qwen3 q_norm / k_norm), v copied -> (out, rstd [tokens, hq + hkv])"""
    tokens = qkv2d.shape[0]
    out = torch.empty_like(qkv2d)
    rstd = torch.empty((tokens, hq + hkv), dtype=torch.float32, device=qkv2d.device)
    _launch("ar_headnorm_fwd", _p(qkv2d, "qkv"), _p(wq, "wq"), _p(wk, "wk"), _p(out), _p(rstd), tokens, qkv2d.stride(0), hq, hkv, d, float(eps),
            dt_code(qkv2d.dtype))
    return out, rstd


def headnorm_bwd_(dqkv2d, qkv2d, wq, wk, rstd, hq, hkv, d):
    """in place: gradient w.r.t. the normalised q / k heads -> gradient w.r.t. the raw projection output"""
    _launch("ar_headnorm_bwd", _p(dqkv2d, "dqkv"), _p(qkv2d, "qkv"), _p(wq, "wq"), _p(wk, "wk"), _p(rstd, "rstd"), dqkv2d.shape[0],
            dqkv2d.stride(0), hq, hkv, d, dt_code(dqkv2d.dtype))
    return dqkv2d


def transpose16(src2d, out=None):
    """[R, C] bf16 / fp16 (contiguous, R and C multiples of 64) -> [C, R]; None when the shape is not covered"""
    R, C = src2d.shape
    if R % 64 or C % 64 or src2d.element_size() != 2 or not src2d.is_contiguous():
        return None
    dst = out if out is not None else torch.empty((C, R), dtype=src2d.dtype, device=src2d.device)
    _launch("ar_transpose16", _p(src2d, "src"), _p(dst, "dst"), R, C)
    return dst


def swiglu_fwd(gu2d, F_):
    """gu2d [rows, >= 2F] (row stride = its stride(0)) -> a [rows, F] = silu(gu[:, :F]) * gu[:, F:2F]"""
    rows = gu2d.shape[0]
    a = torch.empty((rows, F_), dtype=gu2d.dtype, device=gu2d.device)
    _launch("ar_swiglu_fwd", _p(gu2d, "gu"), gu2d.stride(0), _p(a), rows, F_, dt_code(gu2d.dtype))
    return a


def swiglu_bwd_(da2d, gu2d, F_):
    """in place: gu2d <- (d gate, d up)"""
    _launch("ar_swiglu_bwd", _p(da2d, "da"), _p(gu2d, "gu"), gu2d.stride(0), gu2d.shape[0], F_, dt_code(gu2d.dtype))
    return gu2d


def _chk(t, name, dtype, shape=None):
    """the kernels trust raw pointers: a wrong index / weight dtype would read out of bounds instead of failing"""
    if t is None:
        return
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def moe_expand(src2d, tok, scale=None, out=None):
    """out[p] = scale[p] * src2d[tok[p]] (scale None: row gather) -> [len(tok), H]"""
    R, H = tok.numel(), src2d.shape[1]
    _chk(tok, "tok", torch.int64)
    _chk(scale, "scale", torch.float32, (R,))
    _chk(out, "out", src2d.dtype, (R, H))
    if out is None:
        out = torch.empty((R, H), dtype=src2d.dtype, device=src2d.device)
    _launch("ar_moe_expand", _p(src2d, "src"), _p(tok, "tok"), _p(scale), _p(out), R, H, dt_code(src2d.dtype))
    return out


def moe_combine(D2d, pos, w=None, res=None, out=None):
    """out[t] = res[t] + sum_k w[t, k] * D2d[pos[t, k]]  (pos int64 [T, K]; w fp32 [T, K] or None; res [T, H] or None)"""
    T, K = pos.shape
    H = D2d.shape[1]
    _chk(pos, "pos", torch.int64)
    _chk(w, "w", torch.float32, (T, K))
    _chk(res, "res", D2d.dtype, (T, H))
    if D2d.shape[0] < T * K:
        raise ValueError(f"moe_combine: D has {D2d.shape[0]} rows, the {T} x {K} positions address {T * K}")
    if out is None:
        out = torch.empty((T, H), dtype=D2d.dtype, device=D2d.device)
    _launch("ar_moe_combine", _p(D2d, "D"), _p(pos, "pos"), _p(w), _p(res), _p(out), T, H, K, dt_code(D2d.dtype))
    return out


def moe_rowdot(A2d, tok, B2d):
    """out[p] = <A2d[tok[p]], B2d[p]> in fp32 -> [len(tok)]"""
    R, H = tok.numel(), B2d.shape[1]
    _chk(tok, "tok", torch.int64)
    if B2d.shape[0] != R or A2d.shape[1] != H or A2d.dtype != B2d.dtype:
        raise ValueError("moe_rowdot: B must have one row per entry of tok and A the same row length / dtype")
    out = torch.empty(R, dtype=torch.float32, device=B2d.device)
    _launch("ar_moe_rowdot", _p(A2d, "A"), _p(tok, "tok"), _p(B2d, "B"), _p(out), R, H, dt_code(B2d.dtype))
    return out


def rope_fwd(qkv2d, cos, sin, batch, seq, hq, hkv, d):
    """qkv2d [tokens, (hq+2hkv)*d] -> q, k, v [tokens, hq*d] (rotary on q/k, kv heads repeated hq/hkv times)"""
    tokens = qkv2d.shape[0]
    q = torch.empty((tokens, hq * d), dtype=qkv2d.dtype, device=qkv2d.device)
    k = torch.empty_like(q)
    v = torch.empty_like(q)
    bstride = 0 if cos.shape[0] == 1 else cos.stride(0)
    _launch("ar_rope_fwd", _p(qkv2d, "qkv"), qkv2d.stride(0), _p(cos, "cos"), _p(sin, "sin"), bstride, _p(q), _p(k), _p(v), tokens, seq,
            hq, hkv, d, dt_code(qkv2d.dtype))
    return q, k, v


def rope_bwd(dq2d, dk2d, dv2d, cos, sin, batch, seq, hq, hkv, d, out=None):
    tokens = dq2d.shape[0]
    dqkv = out if out is not None else torch.empty((tokens, (hq + 2 * hkv) * d), dtype=dq2d.dtype, device=dq2d.device)
    bstride = 0 if cos.shape[0] == 1 else cos.stride(0)
    _launch("ar_rope_bwd", _p(dq2d, "dq"), _p(dk2d, "dk"), _p(dv2d, "dv"), _p(cos, "cos"), _p(sin, "sin"), bstride, _p(dqkv), dqkv.stride(0),
            tokens, seq, hq, hkv, d, dt_code(dq2d.dtype))
    return dqkv


# ---- "exact_rounding": eager torch's rounding points and summation order (csrc/ar_exact.hip) --------------------------------------
def _strided2d(t, name):
    """a [rows, cols] matrix with unit inner stride, 16-byte aligned rows (a column slice of a wider matrix is fine)"""
    if not t.is_cuda:
        raise _lib.Mi355xLibraryError(f"{name} is on {t.device}: the MI355X path only runs on a HIP device and has no CPU fallback")
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16:
        raise ValueError(f"{name}: expected a 2-D matrix with unit inner stride and 16-byte aligned rows")
    p = _Ptr(t.data_ptr())
    p.dev = t.device.index
    return p


def exact_mean_factor(rows: int, hidden: int) -> float:
    """ATen's mean kernel multiplies the row sum by float(num_outputs) / float(numel) (ReduceMomentKernel.cu mean_kernel_impl)"""
    import numpy as np

    return float(np.float32(rows) / np.float32(rows * hidden))


def rmsnorm_fwd_exact(x2d, weight, eps, res=None, rsqrt_f32=False, raw_sum=False):
    """LlamaRMSNorm.forward with eager torch's bits.  res given: s = x + res (rounded, its own eager op) is normalised and returned.
    -> (y, rstd [rows] fp32, s or x2d); None when ATen would take a code path the kernel does not mirror (rows < 8, hidden < 256).
    rsqrt_f32: the float rsqrt instruction instead of torch-on-ROCm's double evaluation; raw_sum (probes): `rstd` holds the row sums."""
    rows, H = x2d.shape
    if rows < 8 or H < 128 or H % 8 or weight.dtype != x2d.dtype or rows * H > 0x1fffffff:
        return None
    y = torch.empty_like(x2d)
    s = torch.empty_like(x2d) if res is not None else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    _launch("ar_rmsnorm_fwd_exact", _p(x2d, "x"), _p(res), _p(weight, "weight"), _p(s), _p(y), _p(rstd), rows, H, float(eps),
            exact_mean_factor(rows, H), int(bool(rsqrt_f32)) | (2 if raw_sum else 0), dt_code(x2d.dtype))
    return y, rstd, (s if res is not None else x2d)


def rmsnorm_bwd_exact(dy2d, x2d, weight, rstd, dres=None, out=None):
    """torch autograd's backward of LlamaRMSNorm.forward w.r.t. its input, op by op (+ dres: the residual branch's gradient)"""
    rows, H = x2d.shape
    dx = out if out is not None else torch.empty_like(x2d)
    _launch("ar_rmsnorm_bwd_exact", _p(dy2d, "dy"), _p(x2d, "x"), _p(weight, "weight"), _p(rstd, "rstd"), _p(dres), _p(dx), rows, H,
            dt_code(x2d.dtype))
    return dx


def layernorm_fwd_exact(x2d, weight, bias, eps, flags=0, want_stats=True):
    """nn.LayerNorm as the module path computes it under autocast (fp32 `native_layer_norm` on x.float(), result cast to the activation
    dtype) with the bits of torch's own kernel -> (y, mean [rows] fp32, rstd [rows] fp32) (stats None unless wanted); None where ATen
    would take a code path csrc/ar_exact_ln.hip does not restate (hidden % 4, unaligned rows)."""
    rows, H = x2d.shape
    if H < 4 or H % 4 or weight is None or bias is None or weight.dtype != x2d.dtype or bias.dtype != x2d.dtype or not x2d.is_contiguous() \
            or x2d.dtype not in (torch.bfloat16, torch.float16) or (x2d.data_ptr() | weight.data_ptr() | bias.data_ptr()) % 8:
        return None
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device) if want_stats else None
    _launch("ar_layernorm_fwd_exact", _p(x2d, "x"), _p(weight, "weight"), _p(bias, "bias"), _p(y), _p(mean), _p(rstd), rows, H, float(eps),
            int(flags), dt_code(x2d.dtype))
    return y, mean, rstd


def layernorm_bwd_exact(dy2d, x2d, weight, mean, rstd, dres=None, flags=0, out=None):
    """torch autograd's input gradient of that LayerNorm (fp32 `layer_norm_grad_input_kernel`, cast to the activation dtype) + dres,
    the residual branch's gradient, added in the activation dtype.  Raises Mi355xLibraryError for rows >= 32768 (ATen's ROCm build
    switches kernels there): callers check `layernorm_bwd_exact_ok` first."""
    rows, H = x2d.shape
    dx = out if out is not None else torch.empty_like(x2d)
    _launch("ar_layernorm_bwd_exact", _p(dy2d, "dy"), _p(x2d, "x"), _p(weight, "weight"), _p(mean, "mean"), _p(rstd, "rstd"), _p(dres), _p(dx),
            rows, H, dt_code(x2d.dtype))
    return dx


def layernorm_bwd_exact_ok(rows: int, H: int) -> bool:
    return H >= 4 and H % 4 == 0 and rows < 32768


def rope_fwd_exact(q2d, k2d, cos, sin, seq, hq, hkv, d):
    """apply_rotary_pos_emb on separate q [tokens, hq*d] / k [tokens, hkv*d] projections (column slices allowed) -> contiguous q, k"""
    tokens = q2d.shape[0]
    qo = torch.empty((tokens, hq * d), dtype=q2d.dtype, device=q2d.device)
    ko = torch.empty((tokens, hkv * d), dtype=q2d.dtype, device=q2d.device)
    bstride = 0 if cos.shape[0] == 1 else cos.stride(0)
    _launch("ar_rope_fwd_exact", _strided2d(q2d, "q"), q2d.stride(0), _strided2d(k2d, "k"), k2d.stride(0), _p(cos, "cos"), _p(sin, "sin"), bstride,
            _p(qo), _p(ko), tokens, seq, hq, hkv, d, dt_code(q2d.dtype))
    return qo, ko


def rope_bwd_exact(gq4, gk4, cos, sin, seq, d, out=None):
    """gq4 [B, hq, S, d] / gk4 [B, hkv, S, d] (any strides with a contiguous head dimension): gradients of the rotated q / k as the
    attention backward left them -> (dq [tokens, hq*d], dk [tokens, hkv*d]) gradients of the projections; `out` = (dq, dk) may be
    column slices of one merged [tokens, (hq + 2 hkv) d] gradient buffer"""
    def fix(t):
        if t.stride(3) != 1 or any(st % 8 for st in t.stride()[:3]) or t.data_ptr() % 16:
            t = t.contiguous()
        return t

    gq4, gk4 = fix(gq4), fix(gk4)
    B, hq, S, _ = gq4.shape
    hkv = gk4.shape[1]
    tokens = B * S
    if out is None:
        dq = torch.empty((tokens, hq * d), dtype=gq4.dtype, device=gq4.device)
        dk = torch.empty((tokens, hkv * d), dtype=gq4.dtype, device=gq4.device)
    else:
        dq, dk = out
    bstride = 0 if cos.shape[0] == 1 else cos.stride(0)
    pq, pk = _Ptr(gq4.data_ptr()), _Ptr(gk4.data_ptr())
    pq.dev = pk.dev = gq4.device.index
    _launch("ar_rope_bwd_exact", pq, gq4.stride(0), gq4.stride(2), gq4.stride(1), pk, gk4.stride(0), gk4.stride(2), gk4.stride(1),
            _p(cos, "cos"), _p(sin, "sin"), bstride, _strided2d(dq, "dq"), dq.stride(0), _strided2d(dk, "dk"), dk.stride(0), tokens, seq,
            hq, hkv, d, dt_code(gq4.dtype))
    return dq, dk


def swiglu_fwd_exact(g2d, u2d):
    rows, F_ = g2d.shape
    a = torch.empty((rows, F_), dtype=g2d.dtype, device=g2d.device)
    _launch("ar_swiglu_fwd_exact", _strided2d(g2d, "gate"), g2d.stride(0), _strided2d(u2d, "up"), u2d.stride(0), _p(a), rows, F_, dt_code(g2d.dtype))
    return a


def swiglu_bwd_exact(da2d, g2d, u2d, contract: bool, out=None):
    """-> (d gate, d up) [rows, F]; contract: `1 + g * (1 - s)` of silu_backward as an fma (what ATen's kernel compiles to); `out` =
    (dg, du) may be the two halves of one merged [rows, 2F] gradient buffer"""
    rows, F_ = g2d.shape
    if out is None:
        dg = torch.empty((rows, F_), dtype=g2d.dtype, device=g2d.device)
        du = torch.empty_like(dg)
    else:
        dg, du = out
    _launch("ar_swiglu_bwd_exact", _p(da2d, "da"), _strided2d(g2d, "gate"), g2d.stride(0), _strided2d(u2d, "up"), u2d.stride(0),
            _strided2d(dg, "dg"), dg.stride(0), _strided2d(du, "du"), du.stride(0), rows, F_, int(bool(contract)), dt_code(g2d.dtype))
    return dg, du


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, seq: int, heads: int, head_dim: int, scale=None, mask_struct=None):
    """mask_struct = (bias_in, bias_out, valid_len): the structured additive mask of a calibration flow instead of causality
    (ar_attn_fwd_masked: bias_in where `k <= q and k < valid_len`, bias_out elsewhere, both finite; see `mask_structure`).
    Causal attention forward on token-major operands q / k / v [batch * seq, heads * head_dim] (bf16, K / V already repeated to
    `heads`; unit inner stride -- the three may be column slices of one merged projection output, k and v with the same row stride):
    -> (out [batch * seq, heads * head_dim], lse [batch, heads, seq] fp32), or None when the kernel does not take the shape (head
    size other than 128 / 64, seq not a multiple of 128) -- the caller then keeps torch's SDPA."""
    if q.dtype != torch.bfloat16 or head_dim not in (128, 64) or seq % 128:
        return None
    if any(t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16 for t in (q, k, v)) or k.stride(0) != v.stride(0):
        return None
    out = torch.empty((q.shape[0], heads * head_dim), dtype=q.dtype, device=q.device)
    lse = torch.empty((batch, heads, seq), dtype=torch.float32, device=q.device)
    sc = float(scale) if scale is not None else head_dim ** -0.5
    devs = set()
    for t in (q, k, v):
        if not t.is_cuda:
            raise _lib.Mi355xLibraryError("attn_fwd: the MI355X path only runs on a HIP device and has no CPU fallback")
        devs.add(t.device.index)
    if len(devs) != 1:
        raise _lib.Mi355xLibraryError("attn_fwd: tensors live on different HIP devices")
    (dev,) = devs
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        if mask_struct is not None:
            b_in, b_out, valid = mask_struct
            rc = load().ar_attn_fwd_masked(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), batch, seq, heads, head_dim,
                                           sc, float(b_in), float(b_out), int(valid), q.stride(0), k.stride(0),
                                           torch.cuda.current_stream(dev).cuda_stream)
        else:
            rc = load().ar_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), batch, seq, heads, head_dim,
                                    sc, 1, q.stride(0), k.stride(0), torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return None
    check(rc, "ar_attn_fwd")
    return out, lse


def attn_key_block_guess(head_dim: int, seq: int) -> int:
    """The key block (keys per online-softmax step) of the configuration the library's attention forward uses for a problem, as
    measured on torch 2.10.0+rocm7.0 / AOTriton 0.11.1 (profiles/r06_attn_exact_keyblock_probe.json: it depends on the head size and
    the sequence length only) -- the first candidate of a caller's proof, never trusted by itself."""
    if head_dim == 128:
        return 32 if seq <= 512 else 64
    return 64 if (seq <= 512 or seq > 2048) else 32


def attn_fwd_exact(q4: torch.Tensor, k4: torch.Tensor, v4: torch.Tensor, mask_struct, scale: float, key_block: int = 0):
    """The attention forward with the LIBRARY'S bits (ar_attn_fwd_exact, csrc/ar_attn_exact.hip): what
    `F.scaled_dot_product_attention(q4, repeat_kv(k4), repeat_kv(v4), attn_mask=<the structured 0 / 1 mask>, scale=scale)` returns on
    this stack, value for value -- callers PROVE that per call signature before relying on it (exact_block.plan_against_module).
    q4 [B, H, S, D], k4 / v4 [B, H / kv_rep, S, D]: bf16 views with unit stride along D (token-major or head-major).
    mask_struct = (bias_in, bias_out, valid_len) from `mask_structure`.  key_block: keys per online-softmax step (0 = the tuning
    minibatch's; 16 / 32 / 64: the library's other configurations, see the header).  -> (out [B, S, H, D] contiguous, lse [B, H, S] fp32), or
    None when the kernel does not take the call (the caller keeps torch's SDPA)."""
    if mask_struct is None or q4.dim() != 4 or q4.dtype != torch.bfloat16 or k4.dtype != q4.dtype or v4.dtype != q4.dtype:
        return None
    B, H, S, D = (int(x) for x in q4.shape)
    if D not in (64, 128) or S % 128 or k4.shape != v4.shape or k4.shape[0] != B or k4.shape[2] != S or k4.shape[3] != D:
        return None
    hk = int(k4.shape[1])
    if hk < 1 or H % hk:
        return None
    for t in (q4, k4, v4):
        if t.stride(3) != 1 or any(s % 8 for s in t.stride()[:3]) or t.data_ptr() % 16:
            return None
        if not t.is_cuda:
            raise _lib.Mi355xLibraryError("attn_fwd_exact: the MI355X path only runs on a HIP device and has no CPU fallback")
    dev = q4.device.index
    if k4.device.index != dev or v4.device.index != dev:
        raise _lib.Mi355xLibraryError("attn_fwd_exact: tensors live on different HIP devices")
    out = torch.empty((B, S, H, D), dtype=q4.dtype, device=q4.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q4.device)
    b_in, b_out, valid = mask_struct
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        rc = load().ar_attn_fwd_exact(q4.data_ptr(), k4.data_ptr(), v4.data_ptr(), out.data_ptr(), lse.data_ptr(), B, S, H, D, H // hk,
                                      float(scale), float(b_in), float(b_out), int(valid), q4.stride(0), q4.stride(1), q4.stride(2),
                                      k4.stride(0), k4.stride(1), k4.stride(2), v4.stride(0), v4.stride(1), v4.stride(2),
                                      int(key_block), torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return None
    check(rc, "ar_attn_fwd_exact")
    return out, lse


_xattn_ws: dict = {}


def attn_bwd_exact(q4, k4, v4, out, lse, dout, mask_struct, scale: float, dq=None, dk=None, dv=None):
    """The attention backward with the LIBRARY'S bits (ar_attn_bwd_exact): autograd of the call `attn_fwd_exact` replaces.
    q4 [B, H, S, D], k4 / v4 [B, H / kv_rep, S, D] (the forward's operands), out [B, S, H, D] and lse [B, H, S] (the forward's
    results), dout [B, S, H, D]-shaped gradient of out (any strides with unit stride along D).
    -> (dq, dk_exp, dv_exp), each [B, S, H, D] (token-major; dk_exp / dv_exp per QUERY head: sum the kv_rep heads of a group as
    autograd's expand backward does), or None when the kernel does not take the call.  dq / dk / dv: optional [B * S, >= H * D]
    destinations (column slices of a merged buffer)."""
    if mask_struct is None or q4.dim() != 4 or q4.dtype != torch.bfloat16:
        return None
    B, H, S, D = (int(x) for x in q4.shape)
    if D not in (64, 128) or S % 128 or S > 4096 or k4.shape != v4.shape or tuple(out.shape) != (B, S, H, D) or tuple(dout.shape) != (B, S, H, D):
        return None
    hk = int(k4.shape[1])
    if hk < 1 or H % hk or k4.shape[0] != B or k4.shape[2] != S or k4.shape[3] != D:
        return None
    for t in (q4, k4, v4, out, dout):
        if t.dtype != torch.bfloat16 or t.stride(3) != 1 or any(s % 8 for s in t.stride()[:3]) or t.data_ptr() % 16:
            return None
    _chk(lse, "lse", torch.float32, (B, H, S))
    dev = q4.device.index
    if any((not t.is_cuda) or t.device.index != dev for t in (q4, k4, v4, out, dout, lse)):
        raise _lib.Mi355xLibraryError("attn_bwd_exact: every tensor must live on one HIP device (no CPU fallback)")
    outs = []
    for t in (dq, dk, dv):
        if t is None:
            t = torch.empty((B * S, H * D), dtype=q4.dtype, device=q4.device)
        if t.dim() != 2 or tuple(t.shape) != (B * S, H * D) or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16 or t.dtype != q4.dtype:
            return None
        outs.append(t)
    need = load().ar_attn_bwd_exact_workspace_bytes(B, S, H)
    ws = _xattn_ws.get(dev)
    if ws is None or ws.numel() < need:
        ws = _xattn_ws[dev] = torch.empty(need, dtype=torch.uint8, device=q4.device)
    b_in, b_out, valid = mask_struct
    # out / dout are [B, S, H, D]-shaped: (batch, head, token) strides = (0, 2, 1)
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        rc = load().ar_attn_bwd_exact(q4.data_ptr(), k4.data_ptr(), v4.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                      outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), B, S, H, D, H // hk, float(scale),
                                      float(b_in), float(b_out), int(valid),
                                      q4.stride(0), q4.stride(1), q4.stride(2), k4.stride(0), k4.stride(1), k4.stride(2),
                                      v4.stride(0), v4.stride(1), v4.stride(2), out.stride(0), out.stride(2), out.stride(1),
                                      dout.stride(0), dout.stride(2), dout.stride(1), outs[0].stride(0), outs[1].stride(0), outs[2].stride(0),
                                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return None
    check(rc, "ar_attn_bwd_exact")
    return tuple(t.view(B, S, H, D) for t in outs)      # (a column slice of a merged buffer views as [B, S, H, D] with its row stride)


_mask_struct_cache: dict = {}


def mask_structure(mask: torch.Tensor, seq: int):
    """(bias_in, bias_out, valid_len) when `mask` -- a [1 | B, 1, S, S] additive attention mask with the same rows for every batch
    entry -- is the calibration flow's structured mask: one finite value where `k <= q and k < valid_len`, another finite value
    elsewhere (auto_round/calibration/llm.py:360-402 + inputs.py:100-107: the boolean `causal & key-is-valid` mask cast to 0 / 1).
    None for anything else (hard -inf masks, per-sample padding, arbitrary biases): the caller keeps torch's SDPA.  One device
    comparison and one host read per distinct mask tensor (cached per tensor object and version: the mask is the same object every
    iteration)."""
    if mask is None or mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[-1] != seq or mask.shape[-2] != seq or not mask.is_floating_point():
        return None
    import weakref

    key = id(mask)          # (the tensor OBJECT, held weakly: a freed mask's address -- and its id -- can be handed out again)
    hit = _mask_struct_cache.get(key)
    if hit is not None and hit[0]() is mask and hit[1] == mask._version:
        return hit[2]
    with torch.no_grad():
        m = mask[0, 0].float()
        ok_batch = bool((mask == mask[:1]).all()) if mask.shape[0] > 1 else True
        b_in, b_out = float(m[0, 0]), float(m[0, seq - 1]) if seq > 1 else float(m[0, 0])
        last = m[seq - 1]                                   # the last query sees every causal key: its row shows the key padding
        invalid = (last != b_in)
        n_inv = int(invalid.sum())
        valid = seq - n_inv
        res = None
        finite = all(abs(v) <= 1e4 for v in (b_in, b_out))
        if ok_batch and finite and valid >= 1 and b_in != b_out and (n_inv == 0 or bool(invalid[valid:].all())):
            idx = torch.arange(seq, device=mask.device)
            keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < valid)
            want = torch.where(keep, torch.full_like(m, b_in), torch.full_like(m, b_out))
            if bool((m == want).all()):
                res = (b_in, b_out, valid)
    if len(_mask_struct_cache) > 64:
        _mask_struct_cache.clear()
    _mask_struct_cache[key] = (weakref.ref(mask), mask._version, res)
    return res


_attn_ws: dict = {}


def attn_bwd(q, k, v, out, lse, dout, batch: int, seq: int, heads: int, head_dim: int, scale=None, dq=None, dk=None, dv=None, mask_struct=None):
    """mask_struct = (bias_in, bias_out, valid_len): the calibration flow's structured additive mask instead of causality
    (ar_attn_bwd_masked; `out` / `lse` then come from `attn_fwd(..., mask_struct=...)`).
    (head size 64 or 128).
    Causal attention backward (head size 64, deterministic) on token-major operands [batch * seq, heads * 64] with unit inner
    stride (column slices of merged buffers are fine; so are dq / dk / dv given as such slices): -> (dq, dk, dv), or None when the
    kernel does not take the shape -- the caller then keeps the library backward."""
    if head_dim not in (64, 128) or (head_dim == 128 and mask_struct is None) or seq % 256 or seq > 4096 or q.dtype != torch.bfloat16:
        return None          # (causal at head size 128: the library's flash backward is the faster one)
    ts = [q, k, v, out, dout]
    if any(t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16 or t.dtype != torch.bfloat16 for t in ts):
        return None
    T, HD = batch * seq, heads * head_dim
    _chk(lse, "lse", torch.float32, (batch, heads, seq))
    if any(tuple(t.shape) != (T, HD) for t in ts):
        raise ValueError(f"attn_bwd: q / k / v / out / dout must be [{T}, {HD}]")
    outs = []
    for t in (dq, dk, dv):
        if t is None:
            t = torch.empty((T, HD), dtype=q.dtype, device=q.device)
        if t.dim() != 2 or tuple(t.shape) != (T, HD) or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16:
            return None
        outs.append(t)
    dev = q.device.index
    if any((not t.is_cuda) or t.device.index != dev for t in ts + outs + [lse]):
        raise _lib.Mi355xLibraryError("attn_bwd: every tensor must live on one HIP device (no CPU fallback)")
    need = load().ar_attn_bwd_workspace_bytes(batch, seq, heads)
    ws = _attn_ws.get(dev)
    if ws is None or ws.numel() < need:
        ws = _attn_ws[dev] = torch.empty(need, dtype=torch.uint8, device=q.device)
    sc = float(scale) if scale is not None else head_dim ** -0.5
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        if mask_struct is not None:
            b_in, b_out, valid = mask_struct
            rc = load().ar_attn_bwd_masked(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                           outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), batch, seq, heads, head_dim, sc,
                                           float(b_in), float(b_out), int(valid), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                           dout.stride(0), outs[0].stride(0), outs[1].stride(0), outs[2].stride(0), ws.data_ptr(), ws.numel(),
                                           torch.cuda.current_stream(dev).cuda_stream)
        else:
            rc = load().ar_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                    outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), batch, seq, heads, head_dim, sc, 1,
                                    q.stride(0), k.stride(0), v.stride(0), out.stride(0), dout.stride(0), outs[0].stride(0), outs[1].stride(0),
                                    outs[2].stride(0), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return None
    check(rc, "ar_attn_bwd")
    return tuple(outs)


_gemm_ws: dict = {}


def gemm_dw(dY2d: torch.Tensor, X2d: torch.Tensor, out: torch.Tensor, accumulate: bool = False, split: bool = True) -> bool:
    """out[M,N] (+)= dY2d[K,M]^T @ X2d[K,N] through the hand-written MFMA kernel (bf16, fp32 accumulate).  Operands may be
    column slices of wider row-major buffers (unit inner stride).  -> False when the shape / alignment is outside what the
    kernel takes (the caller then keeps the library GEMM); raises on a failed launch.  split: True = the kernel's own launch-shape
    plans (split-K for few tiles, a split last round); False or 1 = every output element is ONE pass over K in token order; an int
    n >= 2 = exactly n contiguous K slices summed in slice order (ar_gemm_dw_ex) -- the summation structures the exact_rounding plan
    compares with the library's."""
    if dY2d.dtype != torch.bfloat16 or X2d.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        return False
    if dY2d.dim() != 2 or X2d.dim() != 2 or out.dim() != 2 or dY2d.stride(1) != 1 or X2d.stride(1) != 1 or out.stride(1) != 1:
        return False
    K, M = dY2d.shape
    N = X2d.shape[1]
    if X2d.shape[0] != K or tuple(out.shape) != (M, N):
        raise ValueError("gemm_dw: shape mismatch")
    devs = set()
    for t in (dY2d, X2d, out):
        if not t.is_cuda:
            raise _lib.Mi355xLibraryError("gemm_dw: the MI355X path only runs on a HIP device and has no CPU fallback")
        devs.add(t.device.index)
    if len(devs) != 1:
        raise _lib.Mi355xLibraryError("gemm_dw: tensors live on different HIP devices")
    (dev,) = devs
    forced = 0 if split is True else (1 if split is False else int(split))
    ws_bytes = load().ar_gemm_dw_workspace_bytes(M, N, K) if forced == 0 else (forced * M * N * 4 if forced > 1 else 0)
    ws = None
    if ws_bytes > 0:                # split-K partial tiles: one growing scratch buffer per device, reused by every call
        ws = _gemm_ws.get(dev)
        if ws is None or ws.numel() < ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=f"cuda:{dev}")
            _gemm_ws[dev] = ws
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        if forced:
            rc = load().ar_gemm_dw_ex(dY2d.data_ptr(), X2d.data_ptr(), out.data_ptr(), M, N, K, dY2d.stride(0), X2d.stride(0), out.stride(0),
                                      int(bool(accumulate)), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), forced,
                                      torch.cuda.current_stream(dev).cuda_stream)
        else:
            rc = load().ar_gemm_dw(dY2d.data_ptr(), X2d.data_ptr(), out.data_ptr(), M, N, K, dY2d.stride(0), X2d.stride(0), out.stride(0),
                                   int(bool(accumulate)), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(),
                                   torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return False
    check(rc, "ar_gemm_dw")
    return True


def gemm_dw_sk(dY2d: torch.Tensor, X2d: torch.Tensor, out: torch.Tensor, kcut: torch.Tensor) -> bool:
    """out[M,N] = dY2d[K,M]^T @ X2d[K,N] with a stream-K summation structure (ar_gemm_dw_sk): the 256 x 256 output tile t (row-major)
    in one pass over K when kcut[t] == 0, else as k-rows [0, kcut[t]) + [kcut[t], K).  kcut: int32 [tiles] on the device, multiples
    of 32.  -> False when the shape is outside what the kernel takes."""
    if dY2d.dtype != torch.bfloat16 or X2d.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        return False
    if dY2d.dim() != 2 or X2d.dim() != 2 or out.dim() != 2 or dY2d.stride(1) != 1 or X2d.stride(1) != 1 or out.stride(1) != 1:
        return False
    K, M = dY2d.shape
    N = X2d.shape[1]
    if X2d.shape[0] != K or tuple(out.shape) != (M, N):
        raise ValueError("gemm_dw_sk: shape mismatch")
    if M % 256 or N % 256:
        return False
    tiles = (M // 256) * (N // 256)
    if kcut.dtype != torch.int32 or kcut.dim() != 1 or not kcut.is_contiguous() or kcut.numel() != tiles:
        raise ValueError("gemm_dw_sk: kcut must be a contiguous int32 table with one entry per 256 x 256 output tile")
    devs = set()
    for t in (dY2d, X2d, out, kcut):
        if not t.is_cuda:
            raise _lib.Mi355xLibraryError("gemm_dw_sk: the MI355X path only runs on a HIP device and has no CPU fallback")
        devs.add(t.device.index)
    if len(devs) != 1:
        raise _lib.Mi355xLibraryError("gemm_dw_sk: tensors live on different HIP devices")
    (dev,) = devs
    ws_bytes = tiles * 256 * 256 * 4
    ws = _gemm_ws.get(dev)
    if ws is None or ws.numel() < ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=f"cuda:{dev}")
        _gemm_ws[dev] = ws
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        rc = load().ar_gemm_dw_sk(dY2d.data_ptr(), X2d.data_ptr(), out.data_ptr(), M, N, K, dY2d.stride(0), X2d.stride(0), out.stride(0),
                                  ws.data_ptr(), ws.numel(), kcut.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return False
    check(rc, "ar_gemm_dw_sk")
    return True


def _same_device(name, *ts):
    devs = set()
    for t in ts:
        if not t.is_cuda:
            raise _lib.Mi355xLibraryError(f"{name}: the MI355X path only runs on a HIP device and has no CPU fallback")
        devs.add(t.device.index)
    if len(devs) != 1:
        raise _lib.Mi355xLibraryError(f"{name}: tensors live on different HIP devices")
    return devs.pop()


def _nt_operands_ok(A, B, out) -> bool:
    if A.dtype != torch.bfloat16 or B.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
        return False
    return A.dim() == 2 and B.dim() == 2 and out.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1


def gemm_nt(A: torch.Tensor, B: torch.Tensor, out: torch.Tensor) -> bool:
    """out[M,N] = A[M,K] @ B[N,K]^T through the hand-written MFMA NT kernel (ar_gemm_nt: bf16, fp32 accumulate over K in ascending
    order, one rounding) -- the forward of F.linear(x, weight_q) (auto_round/wrapper.py:528-556), or with B a transposed weight copy
    its input gradient.  Operands may be row slices / column slices of wider row-major buffers (unit inner stride).  -> False when the
    shape / alignment is outside what the kernel takes (N % 256, K % 128: the caller keeps the library GEMM); raises on a failed launch."""
    if not _nt_operands_ok(A, B, out):
        return False
    M, K = A.shape
    N = B.shape[0]
    if B.shape[1] != K or tuple(out.shape) != (M, N):
        raise ValueError("gemm_nt: shape mismatch")
    dev = _same_device("gemm_nt", A, B, out)
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        rc = load().ar_gemm_nt(A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, A.stride(0), B.stride(0), out.stride(0),
                               torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return False
    check(rc, "ar_gemm_nt")
    return True


def gemm_nt_grouped(A: torch.Tensor, Bbase: torch.Tensor, out: torch.Tensor, row_off: torch.Tensor, b_off: torch.Tensor, N: int, ldb: int) -> bool:
    """Grouped NT GEMM over the experts of a sparse-MoE projection (ar_gemm_nt_grouped): out[r] = A[r] @ B_e^T for the rows r of group e,
    row_off int32 [E + 1] (prefix sums of the groups' row counts, ON THE DEVICE: no host read), B_e = the [N, K] matrix (leading
    dimension ldb) that starts b_off[e] ELEMENTS into `Bbase` (int64 [E], device).  One launch, deterministic."""
    if not Bbase.is_contiguous() or not _nt_operands_ok(A, Bbase.view(-1, 1), out):      # (Bbase: any contiguous buffer holding the matrices)
        return False
    M, K = A.shape
    if tuple(out.shape) != (M, N):
        raise ValueError("gemm_nt_grouped: shape mismatch")
    E = int(b_off.numel())
    if row_off.dtype != torch.int32 or b_off.dtype != torch.int64 or row_off.numel() != E + 1 or not row_off.is_contiguous() or not b_off.is_contiguous():
        raise ValueError("gemm_nt_grouped: row_off must be int32 [E + 1] and b_off int64 [E], contiguous")
    dev = _same_device("gemm_nt_grouped", A, Bbase, out, row_off, b_off)
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        rc = load().ar_gemm_nt_grouped(A.data_ptr(), Bbase.data_ptr(), out.data_ptr(), M, N, K, A.stride(0), ldb, out.stride(0),
                                       row_off.data_ptr(), b_off.data_ptr(), E, torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return False
    check(rc, "ar_gemm_nt_grouped")
    return True


def gemm_dw_grouped(dY2d: torch.Tensor, X2d: torch.Tensor, Wbase: torch.Tensor, row_off: torch.Tensor, w_off: torch.Tensor, ldw: int) -> bool:
    """Grouped weight-gradient GEMM (ar_gemm_dw_grouped): dW_e[M,N] = dY2d[rows of group e]^T @ X2d[rows of group e], written to the
    [M, N] matrix (leading dimension ldw) that starts w_off[e] ELEMENTS into `Wbase`; row_off int32 [E + 1] and w_off int64 [E] live on
    the device.  A group without rows gets zeros.  One launch, deterministic."""
    if dY2d.dtype != torch.bfloat16 or X2d.dtype != torch.bfloat16 or Wbase.dtype != torch.bfloat16:
        return False
    if dY2d.dim() != 2 or X2d.dim() != 2 or dY2d.stride(1) != 1 or X2d.stride(1) != 1 or dY2d.shape[0] != X2d.shape[0]:
        return False
    M, N = int(dY2d.shape[1]), int(X2d.shape[1])
    E = int(w_off.numel())
    if row_off.dtype != torch.int32 or w_off.dtype != torch.int64 or row_off.numel() != E + 1 or not row_off.is_contiguous() or not w_off.is_contiguous():
        raise ValueError("gemm_dw_grouped: row_off must be int32 [E + 1] and w_off int64 [E], contiguous")
    dev = _same_device("gemm_dw_grouped", dY2d, X2d, Wbase, row_off, w_off)
    with (torch.cuda.device(dev) if dev != torch.cuda.current_device() else _NULLCTX):
        rc = load().ar_gemm_dw_grouped(dY2d.data_ptr(), X2d.data_ptr(), Wbase.data_ptr(), M, N, dY2d.stride(0), X2d.stride(0), ldw,
                                       row_off.data_ptr(), w_off.data_ptr(), E, torch.cuda.current_stream(dev).cuda_stream)
    if rc == _lib.AR_ERR_UNSUPPORTED:
        return False
    check(rc, "ar_gemm_dw_grouped")
    return True


# ---- optional device-side timing of the hot kernels (ar_profile_*; bench.py and tools only) ----------------------------
PROF_INT_FWD, PROF_INT_BWD, PROF_FP4_FWD, PROF_FP4_BWD, PROF_GEMM_DW, PROF_NORM, PROF_SWIGLU, PROF_ROPE, PROF_GEMM_NT = range(9)


def profile_enable(on: bool = True) -> bool:
    """Attach start/stop events to every hot-kernel dispatch from now on (off by default).  -> previous state"""
    return bool(load().ar_profile_enable(int(bool(on))))


def profile_reset() -> None:
    check(load().ar_profile_reset(), "ar_profile_reset")


def profile_read(kernel_id: int, min_units: int = 0):
    """-> (total_ms, min_ms, launches) over the recorded launches of `kernel_id` that processed >= min_units units
    (synchronises on the recorded events)."""
    import ctypes

    tot, mn, cnt = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int64(0)
    check(load().ar_profile_read(kernel_id, int(min_units), ctypes.byref(tot), ctypes.byref(mn), ctypes.byref(cnt)),
          "ar_profile_read")
    return tot.value, mn.value, cnt.value
