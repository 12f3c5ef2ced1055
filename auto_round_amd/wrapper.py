"""Host-side mirror of the reference's fake-quant tuning wrappers (auto_round/wrapper.py), MI355X-first.

Same names and argument meaning as the reference -- `WrapperLinear`, `wrapper_block`, `unwrapper_block` -- so that
tests and callers read like the reference's, but the layout underneath is different by design:

* All quantised linears of a transformer block live in ONE `BlockArena`: block-wide flat HBM buffers
  (W, Wq, dWq in the weight dtype; V, best_V fp32; min/max scale, wmin/wmax per group).  Each layer's weight,
  `value`, `min_scale`, `max_scale` are views into those buffers.
* The per-layer `_qdq_weight` of the reference (wrapper.py:244-293) becomes one grouped `ar_qdq_int_fwd` launch per
  iteration for the whole block (the fake-quant weight only depends on parameters, never on activations), and the
  autograd backward + `SignSGD.step()` + `collect_best_params` become one fused `ar_qdq_int_bwd_sgd` launch.
* `WrapperLinear.forward` is a plain MFMA GEMM (hipBLASLt through F.linear) against the layer's Wq view; its
  backward writes dWq = dY^T X straight into the layer's slice of the arena's dWq buffer.

There is no CPU path: every tensor must be on a HIP device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops

try:  # Conv1D (GPT-2 style) is handled like the reference does: weight is stored transposed
    from transformers.pytorch_utils import Conv1D
except Exception:  # pragma: no cover
    Conv1D = ()

SUPPORTED_INT = ("int", "int_sym", "int_asym")


def is_int_dtype(data_type: str) -> bool:
    return data_type in SUPPORTED_INT or data_type.startswith("int")


def is_mx_fp(data_type: str) -> bool:
    return data_type.startswith("mx_fp")


def is_nv_fp(data_type: str) -> bool:
    return data_type.startswith("nv_fp")


def check_to_quantized(layer) -> bool:
    """reference: auto_round/utils check_to_quantized -- a layer is tuned iff bits < 16 (or act_bits < 16)."""
    return int(getattr(layer, "bits", 16)) < 16


def get_scale_shape(weight: torch.Tensor, group_size: int) -> int:
    """Number of groups of a [out, in] weight (reference: wrapper.py get_scale_shape, int group sizes)."""
    out_f, in_f = weight.shape
    if group_size == -1 or in_f < group_size:
        return out_f
    if group_size == 0:
        return 1
    return out_f * ((in_f + group_size - 1) // group_size)


class _QLinearFn(torch.autograd.Function):
    """y = x @ Wq^T (+ b) with the weight gradient written into a preallocated arena slice.

    `token` is a dummy requires-grad scalar that keeps autograd interested even when x itself has no grad (the
    first linear of a block sees the cached calibration activations)."""

    @staticmethod
    def forward(ctx, x, token, wq, bias, dwq_out, accumulate, post_dw=None, mfma_dw=False):
        ctx.post_dw = post_dw
        ctx.mfma_dw = mfma_dw
        ctx.x_dtype = x.dtype
        xc = x if x.dtype == wq.dtype else x.to(wq.dtype)
        ctx.save_for_backward(xc, wq)
        ctx.dwq_out = dwq_out
        ctx.accumulate = accumulate
        ctx.x_needs_grad = x.requires_grad
        return F.linear(xc, wq, None if bias is None else bias.to(wq.dtype))

    @staticmethod
    def backward(ctx, dy):
        xc, wq = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != wq.dtype:
            dy2 = dy2.to(wq.dtype)
        x2 = xc.reshape(-1, xc.shape[-1])
        # dWq[out,in] = dY^T X  -- MFMA GEMM straight into the arena (no autograd accumulation buffers)
        done = False
        if ctx.mfma_dw and ctx.dwq_out.is_contiguous():      # hand-written MFMA kernel where its shape constraints hold and it wins
            from .fused_block import mfma_dw_pays

            if mfma_dw_pays(ctx.dwq_out.shape[0], ctx.dwq_out.shape[1], x2.shape[0]):
                done = ops.gemm_dw(dy2 if dy2.is_contiguous() else dy2.contiguous(), x2 if x2.is_contiguous() else x2.contiguous(),
                                   ctx.dwq_out, accumulate=ctx.accumulate[0])
        if done:
            ctx.accumulate[0] = True
        elif ctx.accumulate[0]:
            ctx.dwq_out.addmm_(dy2.t(), x2)
        else:
            torch.mm(dy2.t(), x2, out=ctx.dwq_out)
            ctx.accumulate[0] = True
        if ctx.post_dw is not None:       # data-parallel tuning: start this layer's gradient all-reduce now, so that it overlaps
            ctx.post_dw()                 # with the backward pass of the layers upstream (sharding.enable_overlapped_sync)
        dx = None
        if ctx.x_needs_grad:
            dx = torch.mm(dy2, wq).reshape(xc.shape)
            if dx.dtype != ctx.x_dtype:
                dx = dx.to(ctx.x_dtype)
        return dx, None, None, None, None, None, None, None


class _ActQdqFn(torch.autograd.Function):
    """Dynamic fp4 fake-quant of an activation tensor (grouped along the last dim) with the reference's autograd
    semantics: `ar_qdq_fp4_fwd` (V = absmax = max_scale = NULL) forward, `ar_fp4_act_bwd` backward.
    reference: WrapperLinear._qdq_act (wrapper.py:295-321) -> quant_mx / nv_fp4_with_static_gs."""

    @staticmethod
    def forward(ctx, x, mode, gs, gscale):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        ctx.mode, ctx.gs, ctx.gscale = mode, gs, gscale
        return ops.qdq_fp4_fwd(xc.view(-1), None, None, None, mode=mode, gs=gs, global_scale=gscale).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        dyc = dy.contiguous()
        if dyc.dtype != xc.dtype:
            dyc = dyc.to(xc.dtype)
        dx = ops.fp4_act_bwd(dyc.view(-1), xc.view(-1), mode=ctx.mode, gs=ctx.gs, global_scale=ctx.gscale)
        return dx.view(xc.shape), None, None, None


class _IntActQdqFn(torch.autograd.Function):
    """Dynamic symmetric INT fake-quant of an activation tensor (W4A8-style schemes): `ar_qdq_int_act_fwd` forward,
    `ar_int_act_bwd` backward (direct path + the gradient through the dynamic scale, routed to the group's arg-min /
    arg-max).  reference: WrapperLinear._qdq_act -> quant_tensor_sym (wrapper.py:295-321, data_type/int.py:165-238)."""

    @staticmethod
    def forward(ctx, x, bits, gs, scale_dtype, thresh, sym=True):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        ctx.cfg = (bits, gs, scale_dtype, thresh, sym)
        return ops.qdq_int_act_fwd(xc.view(-1), gs=gs, bits=bits, sym=sym, scale_dtype=scale_dtype, q_thresh=thresh).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        bits, gs, scale_dtype, thresh, sym = ctx.cfg
        dyc = dy.contiguous()
        if dyc.dtype != xc.dtype:
            dyc = dyc.to(xc.dtype)
        dx = ops.int_act_bwd(dyc.view(-1), xc.view(-1), gs=gs, bits=bits, sym=sym, scale_dtype=scale_dtype, q_thresh=thresh)
        return dx.view(xc.shape), None, None, None, None, None


def act_quant_plan(layer, hidden: int):
    """The activation fake-quant configured on `layer` (act_bits / act_data_type / act_group_size / act_sym) for inputs of last
    dimension `hidden`, as a hashable tuple: ("int", bits, group, scale_dtype, thresh, sym) | ("mx",) | ("nv",), or None when the
    layer's activations stay in 16 bit.  reference: WrapperLinear._qdq_act (auto_round/wrapper.py:295-321)."""
    if int(getattr(layer, "act_bits", 16) or 16) > 8:
        return None
    adt = str(getattr(layer, "act_data_type", ""))
    gs = getattr(layer, "act_group_size", 0)
    gs = int(gs) if gs is not None else 0
    if is_int_dtype(adt):
        if getattr(layer, "act_dynamic", True) is False:
            raise NotImplementedError("int activation fake-quant implements the dynamic case (static act_max scales: NVFP4 only)")
        asym = getattr(layer, "act_sym", True) is False
        g = hidden if (gs in (-1, 0) or hidden < gs) else gs          # data_type/utils.py:47-48
        if g % 8 or hidden % g:
            raise NotImplementedError(f"act_group_size={gs} with hidden size {hidden}: groups must be a multiple of 8 that divides it")
        sdt = getattr(layer, "scale_dtype", torch.float16) or torch.float16
        return ("int", int(layer.act_bits), g, sdt, 1e-8 if sdt == torch.float32 else 1e-5, not asym)
    if int(getattr(layer, "act_bits", 16)) != 4 or hidden % max(gs, 1):
        raise NotImplementedError("activation fake-quant implements MXFP4 (gs 32) / NVFP4 (gs 16) and dynamic symmetric int")
    if is_mx_fp(adt) and gs == 32:
        return ("mx",)
    if is_nv_fp(adt) and gs == 16:
        return ("nv",)
    raise NotImplementedError(f"act_data_type={adt} with act_group_size={gs}")


_nv_gs_dev: dict = {}


def nv_static_plan(layer, device):
    """("nv", global_scale) for a statically activation-quantised NVFP4 layer whose `act_max` was calibrated: the per-layer scale
    448 * 6 / act_max that `act_fake_quant` derives at every call (nv_fp4_with_static_gs, data_type/nvfp.py:102-125), evaluated
    once with the same fp32 arithmetic and carried in the plan as a number, so that layers sharing an input can be compared (q / k / v,
    gate / up).  None when the layer has no calibrated maximum (the dynamic fall-back needs a reduction per call)."""
    act_max = getattr(layer, "act_max", None)
    if act_max is None:
        return None
    tmax = torch.as_tensor(act_max, dtype=torch.float32, device=device).abs().max().reshape(1)
    gscale = torch.where(tmax == 0, torch.zeros_like(tmax), (448.0 * 6.0) * (1.0 / tmax))
    return ("nv", float(gscale.item()))


def _nv_gscale_tensor(value: float, device) -> torch.Tensor:
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:        # a bare "cuda": the CURRENT device, which set_device may have changed
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, value)
    t = _nv_gs_dev.get(key)
    if t is None:
        if len(_nv_gs_dev) > 4096:
            _nv_gs_dev.clear()
        t = _nv_gs_dev[key] = torch.tensor([value], dtype=torch.float32, device=dev)
    return t


def act_quant_fwd_raw(x: torch.Tensor, plan, out=None):
    """Kernel-level forward of a plan ("int" / "mx" dynamic, ("nv", global_scale) static; no autograd): contiguous x -> fake-quantised x."""
    o = None if out is None else out.view(-1)
    if plan[0] == "int":
        _, bits, g, sdt, thresh, sym = plan
        return ops.qdq_int_act_fwd(x.view(-1), gs=g, bits=bits, sym=sym, scale_dtype=sdt, q_thresh=thresh, out=o).view(x.shape)
    if plan[0] == "nv":
        return ops.qdq_fp4_fwd(x.view(-1), None, None, None, mode=1, gs=16, global_scale=_nv_gscale_tensor(plan[1], x.device), out=o).view(x.shape)
    assert plan[0] == "mx"
    return ops.qdq_fp4_fwd(x.view(-1), None, None, None, mode=0, gs=32, global_scale=None, out=o).view(x.shape)


def act_quant_bwd_raw(dy: torch.Tensor, x: torch.Tensor, plan, out=None):
    """Kernel-level input gradient of a plan: (gradient w.r.t. the quantised activation, the activation) -> gradient."""
    if plan[0] == "int":
        _, bits, g, sdt, thresh, sym = plan
        return ops.int_act_bwd(dy.view(-1), x.view(-1), gs=g, bits=bits, sym=sym, scale_dtype=sdt, q_thresh=thresh,
                               out=None if out is None else out.view(-1)).view(x.shape)
    if plan[0] == "nv":
        return ops.fp4_act_bwd(dy.view(-1), x.view(-1), mode=1, gs=16, global_scale=_nv_gscale_tensor(plan[1], x.device),
                               out=None if out is None else out.view(-1)).view(x.shape)
    assert plan[0] == "mx"
    return ops.fp4_act_bwd(dy.view(-1), x.view(-1), mode=0, gs=32, global_scale=None, out=None if out is None else out.view(-1)).view(x.shape)


def act_fake_quant(x: torch.Tensor, layer) -> torch.Tensor:
    """Activation fake-quant as configured on `layer` (act_bits / act_data_type / act_group_size / act_max); reference:
    WrapperLinear._qdq_act (auto_round/wrapper.py:295-321) and WrapperWALayer.forward (:568-612)."""
    plan = act_quant_plan(layer, x.shape[-1])
    if plan is None:
        raise NotImplementedError("activation fake-quant implements MXFP4 (gs 32) / NVFP4 (gs 16) and dynamic symmetric int")
    if plan[0] == "int":
        _, bits, g, sdt, thresh, sym = plan
        return _IntActQdqFn.apply(x, bits, g, sdt, thresh, sym)
    if plan[0] == "mx":
        return _ActQdqFn.apply(x, 0, 32, None)
    act_max = getattr(layer, "act_max", None)
    if act_max is None:     # nv_fp4_with_static_gs falls back to the tensor's own max (nvfp.py:107-108)
        _, tmax = ops.group_absmax(x.detach().contiguous().view(-1), 16, want_tensor_max=True, want_groups=False)
    else:
        tmax = torch.as_tensor(act_max, dtype=torch.float32, device=x.device).abs().max().reshape(1)
    gscale = torch.where(tmax == 0, torch.zeros_like(tmax), (448.0 * 6.0) * (1.0 / tmax))
    return _ActQdqFn.apply(x, 1, 16, gscale.contiguous())


class WrapperWALayer(torch.nn.Module):
    """What the reference leaves in the model after unwrapping an activation-quantised layer (wrapper.py:568-638):
    the baked weight plus the activation fake-quant applied at every forward."""

    def __init__(self, orig_layer):
        super().__init__()
        self.orig_layer = orig_layer

    def forward(self, x):
        x = act_fake_quant(x, self.orig_layer)
        return F.linear(x.to(self.orig_layer.weight.dtype), self.orig_layer.weight, self.orig_layer.bias)


class BlockArena:
    """Block-wide flat HBM buffers for all layers that share (bits, group_size, sym, data_type, dtypes): the storage behind
    the per-layer `value` / `min_scale` / `max_scale` Parameters and `weight_min` / `weight_max` buffers the reference allocates
    one WrapperLinear at a time (auto_round/wrapper.py:139-242)."""

    def __init__(self, key, device):
        (self.data_type, self.bits, self.gs, self.sym, self.w_dtype, self.scale_dtype, self.bounds, self.optimized,
         self.shared) = key        # shared: every layer has ONE (min_scale, max_scale) pair -- per-tensor groups
        self.device = device
        self.layers: List["WrapperLinear"] = []
        self.n = 0
        self.G = 0
        self.built = False
        self.q_thresh = 1e-8 if self.scale_dtype == torch.float32 else 1e-5
        self.tune_minmax = True
        self.wq_fresh = False       # Wq corresponds to the current parameters
        self.kind = "mx" if is_mx_fp(self.data_type) else ("nv" if is_nv_fp(self.data_type) else "int")
        cpg = self.gs // 8
        self.tiled = cpg <= 64 and (cpg & (cpg - 1)) == 0     # lane-group kernels; otherwise one wave per group
        self.mode = {"int": -1, "mx": 0, "nv": 1}[self.kind]
        # `sym` as the kernels take it: 0 asym, 1 sym, 2 (AR_SYM_INIT) sym with the searched init scale in the wmax slot
        self.sym_code = 2 if (self.optimized and self.kind == "int") else int(bool(self.sym))
        self.init = None            # algorithm extension: per-group searched init scale (int: weight dtype, fp4: fp32)

    # -- construction ---------------------------------------------------------------------------------------------
    def add(self, layer: "WrapperLinear") -> Tuple[int, int]:
        off, goff = self.n, self.G
        self.layers.append(layer)
        self.n += layer.numel
        self.G += layer.n_groups
        return off, goff

    def build(self, tune_minmax: bool):
        dev, wd = self.device, self.w_dtype
        self.tune_minmax = tune_minmax
        self.W = torch.empty(self.n, dtype=wd, device=dev)
        self.Wq = torch.empty(self.n, dtype=wd, device=dev)
        self.dWq = torch.zeros(self.n, dtype=wd, device=dev)
        self.V = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.min_scale = torch.ones(self.G, dtype=torch.float32, device=dev)
        self.max_scale = torch.ones(self.G, dtype=torch.float32, device=dev)
        self.best_V = None
        self.best_min = None
        self.best_max = None
        if self.shared:
            if self.kind != "int" or self.optimized:
                raise NotImplementedError("per-tensor groups (group_size=0) are implemented for the INT schemes")
            P = len(self.layers)
            self.min_c = torch.ones(P, dtype=torch.float32, device=dev)        # the tunable parameters proper, one pair per layer
            self.max_c = torch.ones(P, dtype=torch.float32, device=dev)
            self.best_min_c = self.best_max_c = None
            self.share_index = torch.cat([torch.full((l.n_groups,), i, dtype=torch.int64, device=dev)
                                          for i, l in enumerate(self.layers)])
        for li, lyr in enumerate(self.layers):
            lyr._share_slot = li
            lyr._bind(self)
        if self.kind == "int" and self.shared:
            gmin, gmax = ops.group_minmax(self.W, self.gs)
            self.wmin, self.wmax = torch.empty_like(gmin), torch.empty_like(gmax)
            for l in self.layers:       # the tensor's own min / max (wrapper.py:154-164 on the [1, numel] view), on every virtual group
                gl = slice(l._goff, l._goff + l.n_groups)
                self.wmin[gl] = gmin[gl].min()
                self.wmax[gl] = gmax[gl].max()
        elif self.kind == "int":
            self.wmin, self.wmax = ops.group_minmax(self.W, self.gs)
        else:   # fp4: per-group absmax is constant during tuning; NVFP4 also needs the per-layer global scale
            self.wmin = self.wmax = None
            self.absmax, _ = ops.group_absmax(self.W, self.gs)
            if self.kind == "nv":
                for lyr in self.layers:
                    lyr._init_global_scale()
        if self.optimized:
            self._search_init_scales()
        self.token = torch.zeros((), dtype=torch.float32, device=dev, requires_grad=True)
        self.built = True

    def _search_init_scales(self):
        """Algorithm extension: seed every group with the searched init scale (one search launch per layer, since the
        importance matrix is per layer).  reference: SignRoundOptimizedWrapperLinear._init_tuning_params_and_quant_func
        (sign_roundv2/quantizer.py:104-126) -> search_optimized_init_scale (data_type/utils.py:224-254)."""
        dev = self.device
        if self.kind == "int":
            self.init = torch.empty(self.G, dtype=self.w_dtype, device=dev)
        else:
            self.init = torch.empty(self.G, dtype=torch.float32, device=dev)
            cand = torch.tensor(ops.fp4_search_candidates(self.mode), dtype=torch.float32, device=dev)
        for l in self.layers:
            sl, gl = slice(l._off, l._off + l.numel), slice(l._goff, l._goff + l.n_groups)
            qw = l._imatrix_row()
            gpr = l.in_pad // self.gs
            if self.kind == "int":
                self.init[gl] = ops.search_int_scale(self.W[sl], gs=self.gs, bits=self.bits, qw_row=qw, groups_per_row=gpr,
                                                     q_thresh=self.q_thresh)
            else:
                own_gs = None
                if self.kind == "nv":   # the search always uses the tensor's OWN 448*6/amax (nvfp.py:332-338), even when
                    amax = self.absmax[gl].max()                   # the layer tunes with a q/k/v- or gate/up-unified one
                    own_gs = torch.where(amax == 0, torch.zeros_like(amax), (448.0 * 6.0) * (1.0 / amax)).reshape(1)
                self.init[gl] = ops.search_fp4_scale(self.W[sl], self.absmax[gl], cand, mode=self.mode, gs=self.gs, qw_row=qw,
                                                     groups_per_row=gpr, global_scale=own_gs)
        if self.kind == "int":
            self.wmin = self.wmax = self.init

    def alloc_best(self):
        if self.best_V is None:
            self.best_V = self.V.clone()
            self.best_min = self.min_scale.clone()
            self.best_max = self.max_scale.clone()
            if self.shared:
                self.best_min_c, self.best_max_c = self.min_c.clone(), self.max_c.clone()

    def _expand_shared(self):
        """Shared form: in-place [lo, hi] clamp of the per-layer parameters (wrapper.py:257-259), then onto the virtual groups."""
        self.min_c.clamp_(self.bounds[0], self.bounds[1])
        self.max_c.clamp_(self.bounds[0], self.bounds[1])
        torch.index_select(self.min_c, 0, self.share_index, out=self.min_scale)
        torch.index_select(self.max_c, 0, self.share_index, out=self.max_scale)

    def best_params_of(self, lyr) -> dict:
        """The best-so-far parameters of one layer in the reference's shapes (views into the arena's snapshot buffers)."""
        bp = {}
        if "value" in lyr.params:
            bp["value"] = self.best_V[lyr._off:lyr._off + lyr.numel].view(lyr.value.shape)
        if "min_scale" in lyr.params:
            if self.shared:
                i = lyr._share_slot
                bp["min_scale"], bp["max_scale"] = self.best_min_c[i:i + 1], self.best_max_c[i:i + 1]
            else:
                bp["min_scale"] = self.best_min[lyr._goff:lyr._goff + lyr.n_groups]
                bp["max_scale"] = self.best_max[lyr._goff:lyr._goff + lyr.n_groups]
        return bp

    # -- the two grouped launches of an iteration ---------------------------------------------------------------------
    def qdq_forward(self, V=None, min_s=None, max_s=None, want_scale=False):
        """K1 for every layer of the block in one launch."""
        V = self.V if V is None else V
        if self.shared and min_s is None and max_s is None:
            self._expand_shared()
        mn = self.min_scale if min_s is None else min_s
        mx = self.max_scale if max_s is None else max_s
        if self.kind != "int":
            self.wq_fresh = V is self.V and mx is self.max_scale
            if self.kind == "mx":
                return ops.qdq_fp4_fwd(self.W, V, self.absmax, mx, mode=0, gs=self.gs, bounds=self.bounds, out=self.Wq,
                                       init_scale_dev=self.init)
            for l in self.layers:   # NVFP4: one launch per layer (each layer has its own global scale)
                sl, gl = slice(l._off, l._off + l.numel), slice(l._goff, l._goff + l.n_groups)
                ops.qdq_fp4_fwd(self.W[sl], V[sl], self.absmax[gl], mx[gl], mode=1, gs=self.gs,
                                global_scale=l.weight_global_scale_dev, bounds=self.bounds, out=self.Wq[sl],
                                init_scale_dev=None if self.init is None else self.init[gl])
            return self.Wq
        res = ops.qdq_int_fwd(self.W, V, self.wmin, self.wmax, mn, mx, gs=self.gs, bits=self.bits, sym=self.sym_code,
                              scale_dtype=self.scale_dtype, q_thresh=self.q_thresh, bounds=self.bounds, out=self.Wq,
                              want_scale=want_scale)
        self.wq_fresh = V is self.V and mn is self.min_scale and mx is self.max_scale
        return res

    def _momentum_step(self, lr_v, lr_mm, snapshot_flag, momentum):
        """SignSGD with momentum (sign_sgd.py:356-389; off by default): the sign is taken of a running buffer, so the gradients
        have to exist as tensors -- unfused backward, buffers buf <- momentum * buf + grad (first step: buf = grad), sign step."""
        if self.shared:
            raise NotImplementedError("momentum with per-tensor groups")
        mom = self.__dict__.setdefault("_mom", {})

        def step(p, g, lr, key):
            buf = mom.get(key)
            if buf is None:
                buf = mom[key] = g.clone()
            else:
                buf.mul_(momentum).add_(g)
            ops.sign_sgd_(p, buf, lr)

        lo, hi = self.bounds
        # the forward's in-place clamp of the scale parameters (wrapper.py:257-259) comes first: collect_best_params, which runs
        # after the forward, sees the clamped values
        if self.kind == "int":
            self.min_scale.clamp_(lo, hi)
        self.max_scale.clamp_(lo, hi)
        if snapshot_flag is not None:       # collect_best_params happens before optimizer.step()
            take = snapshot_flag.to(torch.bool)
            self.best_V.copy_(torch.where(take, self.V, self.best_V))
            self.best_max.copy_(torch.where(take, self.max_scale, self.best_max))
            self.best_min.copy_(torch.where(take, self.min_scale, self.best_min))
        # one layer at a time: a layer whose forward did not run this iteration (a MoE expert without tokens) has grad None in
        # the reference, and SignSGD skips such parameters -- buffer untouched, no step (sign_sgd.py:356-389)
        for li, l in enumerate(self.layers):
            if not l._dw_accum[0]:
                continue
            sl, gl = slice(l._off, l._off + l.numel), slice(l._goff, l._goff + l.n_groups)
            if self.kind == "int":
                dV, dmin, dmax = ops.qdq_int_bwd(self.dWq[sl], self.W[sl], self.V[sl], self.wmin[gl], self.wmax[gl], self.min_scale[gl],
                                                 self.max_scale[gl], gs=self.gs, bits=self.bits, sym=self.sym_code,
                                                 scale_dtype=self.scale_dtype, q_thresh=self.q_thresh, bounds=self.bounds)
                step(self.V[sl], dV, lr_v, ("v", li))
                if self.tune_minmax:
                    if self.sym_code != 2:
                        step(self.min_scale[gl], dmin, lr_mm, ("min", li))
                    step(self.max_scale[gl], dmax, lr_mm, ("max", li))
            else:
                dV, dmax = ops.qdq_fp4_bwd_sgd_(self.dWq[sl], self.W[sl], self.V[sl], self.absmax[gl], self.max_scale[gl], mode=self.mode,
                                                gs=self.gs, bounds=self.bounds, init_scale_dev=None if self.init is None else self.init[gl],
                                                global_scale=l.weight_global_scale_dev if self.kind == "nv" else None, want_grads=True)
                step(self.V[sl], dV, lr_v, ("v", li))
                if self.tune_minmax:
                    step(self.max_scale[gl], dmax, lr_mm, ("max", li))
        self.wq_fresh = False
        for lyr in self.layers:
            lyr._dw_accum[0] = False

    def backward_step(self, lr_v: torch.Tensor, lr_mm: torch.Tensor, snapshot_flag=None, fuse_next_fwd=True, momentum=0.0):
        """K2+K3 (+ snapshot + next K1) for every layer of the block in one launch; consumes self.dWq."""
        if snapshot_flag is not None:
            self.alloc_best()
        if momentum:
            return self._momentum_step(lr_v, lr_mm, snapshot_flag, float(momentum))
        # a layer whose forward did not run this iteration (e.g. a MoE expert that received no token) has no gradient:
        # the reference's SignSGD skips parameters with grad None; here a zero dWq slice makes every sign step 0
        for lyr in self.layers:
            if not lyr._dw_accum[0]:
                lyr.weight_grad.zero_()
        if self.kind != "int":
            layers = [None] if self.kind == "mx" else self.layers
            for l in layers:
                sl = slice(None) if l is None else slice(l._off, l._off + l.numel)
                gl = slice(None) if l is None else slice(l._goff, l._goff + l.n_groups)
                ops.qdq_fp4_bwd_sgd_(self.dWq[sl], self.W[sl], self.V[sl], self.absmax[gl], self.max_scale[gl],
                                     mode=self.mode, gs=self.gs, bounds=self.bounds,
                                     init_scale_dev=None if self.init is None else self.init[gl],
                                     global_scale=None if l is None else l.weight_global_scale_dev, lr_v=lr_v, lr_mm=lr_mm,
                                     tune_minmax=self.tune_minmax, snapshot_flag=snapshot_flag,
                                     best_V=None if self.best_V is None else self.best_V[sl],
                                     best_max=None if self.best_max is None else self.best_max[gl])
            self.wq_fresh = False
            for lyr in self.layers:
                lyr._dw_accum[0] = False
            return
        if self.shared:
            dV, dmin, dmax = ops.qdq_int_bwd(self.dWq, self.W, self.V, self.wmin, self.wmax, self.min_scale, self.max_scale,
                                             gs=self.gs, bits=self.bits, sym=self.sym_code, scale_dtype=self.scale_dtype,
                                             q_thresh=self.q_thresh, bounds=self.bounds)
            if snapshot_flag is not None:       # collect_best_params happens before optimizer.step() (sign_round/quantizer.py:508-523)
                take = snapshot_flag.to(torch.bool)
                self.best_V.copy_(torch.where(take, self.V, self.best_V))
                self.best_min_c.copy_(torch.where(take, self.min_c, self.best_min_c))
                self.best_max_c.copy_(torch.where(take, self.max_c, self.best_max_c))
            ops.sign_sgd_(self.V, dV, lr_v)
            if self.tune_minmax:
                P = self.min_c.numel()
                gmin = torch.zeros(P, dtype=torch.float32, device=self.device).index_add_(0, self.share_index, dmin)
                gmax = torch.zeros(P, dtype=torch.float32, device=self.device).index_add_(0, self.share_index, dmax)
                ops.sign_sgd_(self.min_c, gmin, lr_mm)
                ops.sign_sgd_(self.max_c, gmax, lr_mm)
            self.wq_fresh = False
            for lyr in self.layers:
                lyr._dw_accum[0] = False
            return
        ops.qdq_int_bwd_sgd_(self.dWq, self.W, self.V, self.wmin, self.wmax, self.min_scale, self.max_scale, gs=self.gs,
                             bits=self.bits, sym=self.sym_code, lr_v=lr_v, lr_mm=lr_mm, tune_minmax=self.tune_minmax,
                             scale_dtype=self.scale_dtype, q_thresh=self.q_thresh, bounds=self.bounds,
                             snapshot_flag=snapshot_flag, best_V=self.best_V, best_min=self.best_min,
                             best_max=self.best_max, Wq_next=self.Wq if (fuse_next_fwd and self.tiled) else None)
        self.wq_fresh = bool(fuse_next_fwd and self.tiled)
        for lyr in self.layers:
            lyr._dw_accum[0] = False

    def param_grads(self):
        """Unfused backward (ar_qdq_int_bwd): materialises dV / d min_scale / d max_scale like autograd would."""
        if self.kind != "int":
            raise NotImplementedError("unfused gradients are exposed for the INT path only")
        return ops.qdq_int_bwd(self.dWq, self.W, self.V, self.wmin, self.wmax, self.min_scale, self.max_scale, gs=self.gs,
                               bits=self.bits, sym=self.sym_code, scale_dtype=self.scale_dtype, q_thresh=self.q_thresh,
                               bounds=self.bounds)


class WrapperLinear(torch.nn.Module):
    """Mirror of auto_round.wrapper.WrapperLinear (wrapper.py:62-565): INT (W2/W3/W4/W8 sym + asym), MXFP4 and NVFP4 weights,
    optional activation fake-quant (MXFP4 / NVFP4 / dynamic INT) in front of the GEMM.

    The wrapped layer must carry the scheme attributes the reference's `apply_plan_to_model` sets: `bits`,
    `group_size`, `sym`, `data_type`, `scale_dtype`, `act_bits` (compressors/layer_config/resolver.py:482-497).
    Tunable parameters (fp32, same shapes as the reference): `value` [G, gs], `min_scale` [G], `max_scale` [G].
    """

    minmax_scale_bound = (0.0, 1.0)
    optimized = False       # True in SignRoundOptimizedWrapperLinear (algorithm extension)

    def __init__(self, orig_layer, enable_minmax_tuning=True, enable_norm_bias_tuning=False, device="cuda",
                 enable_round_tuning=True, enable_torch_compile=False, disable_opt_rtn=True, **kwargs):
        super().__init__()
        if enable_norm_bias_tuning:
            raise NotImplementedError("norm/bias tuning is outside the MI355X hot path (SURVEY 8a: off by default)")
        self.orig_layer = orig_layer
        self.orig_layer.iters = kwargs.pop("iters", 200)
        self.device = torch.device(getattr(orig_layer, "tuning_device", device))
        if self.device.type != "cuda":
            raise RuntimeError(f"WrapperLinear needs a HIP device, got {self.device}; there is no CPU fallback")
        self.output_device = self.device
        self.enable_minmax_tuning = enable_minmax_tuning
        self.enable_round_tuning = enable_round_tuning
        self.enable_act_quant = int(getattr(orig_layer, "act_bits", 16)) <= 8
        self.is_conv1d = bool(Conv1D) and isinstance(orig_layer, Conv1D)
        w = orig_layer.weight.data
        self.out_features, self.in_features = (w.shape[1], w.shape[0]) if self.is_conv1d else tuple(w.shape)
        if isinstance(orig_layer.group_size, (tuple, list)):
            # 2-D block groups (data_type/utils.py:49-56) exist for the FP8 block schemes; through WrapperLinear the reference's
            # own INT / fp4 quant functions cannot take them (weight_min is reduced over both block dims, the quant function
            # reduces over the last one only and the shapes no longer broadcast -- tests/test_torch_ref.py pins that RuntimeError)
            raise NotImplementedError(f"group_size={orig_layer.group_size}: 2-D block groups are an FP8 block-scheme feature; the "
                                      "reference's W2/W3/W4/W8 and fp4 quant functions do not accept them either")
        gs = int(orig_layer.group_size)
        # per-tensor (group_size == 0, data_type/utils.py:57-59): ONE group = the whole weight.  The kernels keep their
        # lane-group form on "virtual" groups of <= 128 consecutive weights that all carry the tensor's (wmin, wmax,
        # min_scale, max_scale); the arena sums their scale gradients before the sign step (BlockArena, shared form)
        self.per_tensor = gs == 0
        if self.per_tensor:
            gs = next((g for g in (128, 64, 32, 16, 8) if self.in_features % g == 0), 0)
        if gs == -1 or self.in_features < gs:
            gs = self.in_features            # per-output-channel groups (data_type/utils.py:47-48)
        if gs <= 0 or gs % 8:
            raise NotImplementedError(f"group_size={orig_layer.group_size} with in_features={self.in_features}: group "
                                      "sizes must be a multiple of 8")
        self.gs = gs
        # rows are zero-padded to a multiple of the group size exactly like reshape_pad_tensor_by_group_size
        # (data_type/utils.py:52-56); the pad columns never receive a gradient, so their V stays 0
        self.in_pad = (self.in_features + gs - 1) // gs * gs
        self.padded = self.in_pad != self.in_features
        self.numel = self.out_features * self.in_pad
        self.n_groups = self.numel // gs
        self.bits = int(orig_layer.bits)
        self.sym = bool(orig_layer.sym)
        self.data_type = str(getattr(orig_layer, "data_type", "int"))
        if is_mx_fp(self.data_type):
            if self.bits != 4 or self.gs != 32:
                raise NotImplementedError("MX path implements mx_fp4 with group_size 32 (MXFP4)")
        elif is_nv_fp(self.data_type):
            if self.bits != 4 or self.gs != 16:
                raise NotImplementedError("NV path implements nv_fp4 with group_size 16 (NVFP4)")
        elif not is_int_dtype(self.data_type):
            raise NotImplementedError(f"data_type {self.data_type}: implemented are int (sym/asym), mx_fp4, nv_fp4")
        self.weight_global_scale = getattr(orig_layer, "weight_global_scale", None)
        self.weight_global_scale_dev = None
        self.scale_dtype = getattr(orig_layer, "scale_dtype", torch.float16) or torch.float16
        self.q_scale_thresh = 1e-8 if self.scale_dtype == torch.float32 else 1e-5
        self.params: Dict[str, torch.nn.Parameter] = {}
        self._dw_accum = [False]
        self.arena: Optional[BlockArena] = None

    # arenas are keyed by everything the grouped kernels treat as launch-uniform
    def arena_key(self):
        return (self.data_type, self.bits, self.gs, self.sym, self.orig_layer.weight.dtype, self.scale_dtype,
                tuple(self.minmax_scale_bound), bool(self.optimized), bool(self.per_tensor))

    def _imatrix_row(self):
        """The layer's importance matrix as one padded fp32 row [in_pad] (pad = 1e-5), or None for uniform importance.
        reference: reshape_imatrix_for_weight (data_type/utils.py:269-282); the attribute is consumed like the reference
        does (sign_roundv2/quantizer.py:123-124)."""
        im = getattr(self.orig_layer, "imatrix", None)
        if hasattr(self.orig_layer, "imatrix"):
            del self.orig_layer.imatrix
        if not isinstance(im, torch.Tensor):
            return None
        row = torch.full((self.in_pad,), 1e-5, dtype=torch.float32, device=self.device)
        row[:self.in_features] = im.reshape(-1).to(device=self.device, dtype=torch.float32)
        return row

    def _bind(self, arena: BlockArena):
        off, goff = self._off, self._goff
        self.arena = arena
        n, G = self.numel, self.n_groups
        w = self.orig_layer.weight.data
        w2d = w.t() if self.is_conv1d else w
        Wv = arena.W[off:off + n].view(self.out_features, self.in_pad)
        if self.padded:
            Wv.zero_()
        Wv[:, :self.in_features].copy_(w2d)
        if not self.is_conv1d and not self.padded:  # the layer's weight now lives in the arena (no second copy in HBM)
            self.orig_layer.weight.data = Wv
        self.weight_q = arena.Wq[off:off + n].view(self.out_features, self.in_pad)[:, :self.in_features]
        self.weight_grad = arena.dWq[off:off + n].view(self.out_features, self.in_pad)[:, :self.in_features]
        tunable_v = self.enable_round_tuning and self.bits < 16
        tunable_mm = self.enable_minmax_tuning and self.bits < 16
        if self.per_tensor:       # the reference's shapes: value [1, numel], min_scale / max_scale [1]
            i = self._share_slot
            self.value = torch.nn.Parameter(arena.V[off:off + n].view(1, n), requires_grad=tunable_v)
            self.min_scale = torch.nn.Parameter(arena.min_c[i:i + 1], requires_grad=tunable_mm)
            self.max_scale = torch.nn.Parameter(arena.max_c[i:i + 1], requires_grad=tunable_mm)
        else:
            self.value = torch.nn.Parameter(arena.V[off:off + n].view(G, self.gs), requires_grad=tunable_v)
            self.min_scale = torch.nn.Parameter(arena.min_scale[goff:goff + G], requires_grad=tunable_mm)
            self.max_scale = torch.nn.Parameter(arena.max_scale[goff:goff + G], requires_grad=tunable_mm)
        if tunable_v:
            self.params["value"] = self.value
        if tunable_mm:
            self.params["min_scale"] = self.min_scale
            self.params["max_scale"] = self.max_scale

    def _init_global_scale(self):
        """NVFP4 per-tensor global scale 448*6/amax(W) (reference: calculate_gparam, data_type/nvfp.py:56-64) unless the
        caller already attached a (possibly q/k/v- or gate/up-unified) `weight_global_scale` to the layer."""
        a = self.arena
        if self.weight_global_scale is None:
            amax = a.absmax[self._goff:self._goff + self.n_groups].max()
            self.weight_global_scale = torch.where(amax == 0, torch.zeros_like(amax), (448.0 * 6.0) * (1.0 / amax))
        self.weight_global_scale_dev = self.weight_global_scale.to(device=self.device, dtype=torch.float32).reshape(1).contiguous()

    @property
    def weight(self):
        return self.orig_layer.weight

    @property
    def bias(self):
        return self.orig_layer.bias

    @property
    def weight_min(self):
        return self.arena.wmin[self._goff:self._goff + self.n_groups]

    @property
    def weight_max(self):
        return self.arena.wmax[self._goff:self._goff + self.n_groups]

    def _qdq_weight(self, value=None, min_scale=None, max_scale=None):
        """Per-layer fake-quant (reference signature, wrapper.py:244-293) -> (weight_q, scale, zp).
        The tuning loop does not call this (it uses the arena's grouped launch); unwrapper and tests do."""
        a = self.arena
        sl = slice(self._off, self._off + self.numel)
        gl = slice(self._goff, self._goff + self.n_groups)

        def flat(t, default, n):
            if t is None:
                return default
            t = t.to(device=self.device, dtype=torch.float32)
            if t.numel() == 1:      # reference passes tensor(0.0)/tensor(1.0) when a parameter is not tunable
                return t.reshape(1).expand(n).contiguous()
            return t.reshape(-1).contiguous()

        V = flat(value, a.V[sl], self.numel)
        mn = flat(min_scale, a.min_scale[gl], self.n_groups)
        mx = flat(max_scale, a.max_scale[gl], self.n_groups)
        if a.kind != "int":
            Wq, scale = ops.qdq_fp4_fwd(a.W[sl], V, a.absmax[gl], mx, mode=a.mode, gs=self.gs, bounds=self.minmax_scale_bound,
                                        global_scale=self.weight_global_scale_dev, want_scale=True,
                                        init_scale_dev=None if a.init is None else a.init[gl])
            wq2d = Wq.view(self.out_features, self.in_pad)[:, :self.in_features]
            return (wq2d.t() if self.is_conv1d else wq2d), scale.view(self.n_groups, 1), None
        Wq, scale, zp = ops.qdq_int_fwd(a.W[sl], V, a.wmin[gl], a.wmax[gl], mn, mx, gs=self.gs, bits=self.bits,
                                        sym=a.sym_code, scale_dtype=self.scale_dtype, q_thresh=self.q_scale_thresh,
                                        bounds=self.minmax_scale_bound, want_scale=True)
        wq2d = Wq.view(self.out_features, self.in_pad)[:, :self.in_features]
        if self.is_conv1d:
            wq2d = wq2d.t()
        if self.sym:
            zp = int(2 ** (self.bits - 1))
        if self.per_tensor:       # every virtual group carries the tensor's scale / zero point: hand back the one value, [1, 1]
            return wq2d, scale[:1].view(1, 1), zp if self.sym else zp[:1].view(1, 1)
        return wq2d, scale.view(self.n_groups, 1), zp if self.sym else zp.view(self.n_groups, 1)

    def forward(self, x):
        a = self.arena
        if not a.wq_fresh:
            a.qdq_forward()
        if self.enable_act_quant:
            x = act_fake_quant(x, self.orig_layer)
        return _QLinearFn.apply(x, a.token, self.weight_q, self.orig_layer.bias, self.weight_grad, self._dw_accum,
                                getattr(self, "_post_dw", None), getattr(self, "_mfma_dw", False))

    def unwrapper(self, best_params):
        """Bake the best parameters into the layer (reference: wrapper.py:345-468): weight <- qdq weight,
        attributes `scale` [out, in/gs] (CPU, scale dtype), `zp` (int for sym, [out, in/gs] CPU tensor for asym)."""
        best_params = best_params or {}
        v = best_params.get("value", torch.tensor(0.0))
        mn = best_params.get("min_scale", torch.tensor(1.0))
        mx = best_params.get("max_scale", torch.tensor(1.0))
        wq, scale, zp = self._qdq_weight(v, mn, mx)
        self.orig_layer.weight.data.copy_(wq)
        self.orig_layer.weight.grad = None
        if scale.numel() > 1:           # wrapper.py:393-398: a single (per-tensor) scale is stored flat
            self.orig_layer.scale = scale.reshape(self.out_features, -1).to("cpu")
        else:
            self.orig_layer.scale = scale.view(-1).to("cpu")
        if isinstance(zp, torch.Tensor):
            self.orig_layer.zp = (zp.reshape(self.out_features, -1) if zp.numel() > 1 else zp.view(-1)).to("cpu")
        else:
            self.orig_layer.zp = zp
        if self.weight_global_scale_dev is not None:
            self.orig_layer.weight_global_scale = self.weight_global_scale_dev.to("cpu")
        if self.enable_act_quant:
            return WrapperWALayer(self.orig_layer)
        return self.orig_layer


class SignRoundOptimizedWrapperLinear(WrapperLinear):
    """Algorithm-extension wrapper (reference: sign_roundv2/quantizer.py:101-161): every group starts from a searched
    `init_scale` (importance-weighted grid search, one kernel launch per layer at wrap time) instead of the min/max
    range, `max_scale` tunes a coefficient on top of it within (0, 2), `min_scale` is not part of the graph.
    Symmetric int / mx_fp4 / nv_fp4 only, like the reference."""

    minmax_scale_bound = (0.0, 2.0)
    optimized = True

    def __init__(self, orig_layer, *args, **kwargs):
        super().__init__(orig_layer, *args, **kwargs)
        if is_int_dtype(self.data_type) and (not self.sym or "asym" in self.data_type or self.data_type.endswith("dq")):
            raise ValueError(f"SignRound optimized path does not support data_type={self.data_type!r} (sym={self.sym}); "
                             "expected a symmetric int / mx / nv type.")

    @property
    def init_scale(self):
        return self.arena.init[self._goff:self._goff + self.n_groups].view(self.n_groups, 1)


def update_block_global_scale_if_needed(block) -> None:
    """NVFP4: give every nv_fp layer a `weight_global_scale` = 448*6/amax(W) and unify it (minimum) across q/k/v, across
    gate/up and across a MoE expert's w1/w3, as the reference does before tuning a block so that fused inference kernels
    can share one scale (data_type/utils.py:433-530; AR_NVFP4_FUSED_LAYER_GLOBAL_SCALE=0 disables the unification)."""
    import os

    nv = [m for m in block.modules() if _quantizable(m) and check_to_quantized(m) and is_nv_fp(str(getattr(m, "data_type", "")))]
    if not nv:
        return
    for m in nv:
        if not hasattr(m, "weight_global_scale"):
            amax = m.weight.detach().to(torch.float32).abs().max()
            m.weight_global_scale = torch.where(amax == 0, torch.zeros_like(amax), (448.0 * 6.0) * (1.0 / amax))
    if os.environ.get("AR_NVFP4_FUSED_LAYER_GLOBAL_SCALE", "1") in ("0", "false", "False"):
        return
    for mod in block.modules():
        for names in (("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"), ("w1", "w3")):
            if all(hasattr(mod, n) for n in names):
                members = [getattr(mod, n) for n in names if hasattr(getattr(mod, n), "weight_global_scale")]
                if members:
                    g = torch.min(torch.stack([x.weight_global_scale.reshape(1).to(members[0].weight.device) for x in members]), dim=0).values
                    for x in members:
                        x.weight_global_scale = g.clone()
                break


def _quantizable(m) -> bool:
    return isinstance(m, torch.nn.Linear) or (bool(Conv1D) and isinstance(m, Conv1D))


def _set_module(root, name, new):
    parts = name.split(".")
    parent = root
    for p in parts[:-1]:
        parent = getattr(parent, p)
    setattr(parent, parts[-1], new)


def wrapper_block(block, enable_minmax_tuning, enable_norm_bias_tuning, enable_torch_compile=False, device="cuda",
                  wrapper_cls=WrapperLinear, **kwargs):
    """Swap every quantisable linear of `block` for a tuning wrapper and build the block arenas.
    reference: auto_round/wrapper.py:774-828.  Returns (quantized_layer_names, unquantized_layer_names)."""
    quantized, unquantized = [], []
    wrappers = []
    for n, m in list(block.named_modules()):
        if not _quantizable(m):
            continue
        if not check_to_quantized(m):
            unquantized.append(n)
            continue
        w = wrapper_cls(m, enable_minmax_tuning=enable_minmax_tuning, enable_norm_bias_tuning=enable_norm_bias_tuning,
                        device=device, enable_torch_compile=enable_torch_compile, **kwargs)
        _set_module(block, n, w)
        wrappers.append(w)
        quantized.append(n)
    arenas: Dict[tuple, BlockArena] = {}
    for w in wrappers:
        key = w.arena_key()
        if key not in arenas:
            arenas[key] = BlockArena(key, w.device)
        w._off, w._goff = arenas[key].add(w)
    for a in arenas.values():
        a.build(enable_minmax_tuning)
    block._ar_arenas = list(arenas.values())
    return quantized, unquantized


@torch.no_grad()
def unwrapper_block(block, best_params):
    """reference: auto_round/wrapper.py:861-878 -- restores the original layers with the best parameters baked in."""
    for n, m in list(block.named_modules()):
        if hasattr(m, "orig_layer") and hasattr(m, "unwrapper"):
            bp = best_params.get(n) if best_params else None
            _set_module(block, n, m.unwrapper(bp))
    if hasattr(block, "_ar_arenas"):
        del block._ar_arenas
