"""Fused tuning-time forward / backward of a Llama-family decoder block (RMSNorm -> q/k/v [-> per-head q / k RMSNorm, Qwen3] ->
rotary -> SDPA -> o -> residual -> RMSNorm -> SwiGLU MLP -> residual) and of an OPT-family one (LayerNorm -> q/k/v + bias -> SDPA -> out_proj -> residual -> LayerNorm
-> fc1 -> ReLU -> fc2 -> residual; `FusedOPTBlock`) on MI355X.

The reference speeds the same code up with `torch.compile(block_forward)` (auto_round/utils/device.py:112-122,
compressors/base.py:1177-1179: "about 20 %"); here the block is written once, by hand, for the one thing the tuning loop does with
it -- forward on a cached minibatch, backward to the fake-quant weights only:

  * the elementwise / normalisation work (20 % of a Llama-3-8B iteration as ~90 eager launches) is six HIP kernels
    (csrc/ar_block.hip: RMSNorm fwd/bwd, rotary + GQA head repeat fwd/bwd, SwiGLU fwd/bwd), token-major, no transposes;
  * q/k/v run as ONE GEMM against the arena's contiguous [q;k;v] slice of the fake-quant weights, gate/up likewise; the two
    residual adds ride in the GEMM epilogue (addmm);
  * the block input needs no gradient, so nothing upstream of q/k/v is differentiated; weight gradients are written straight
    into the arena's dWq slices (merged for q/k/v and gate/up) by the hand-written MFMA kernel (csrc/ar_gemm.hip) where its
    shape constraints hold and it wins, by hipBLASLt otherwise;
  * the causal attention forward is csrc/ar_attn.hip at head size 128 (PyTorch's SDPA otherwise); the backward is the library's.

Same arithmetic per op as the module code, but different bf16 rounding points (fused residual epilogue, fp32 backward of the
elementwise ops, merged GEMMs): parity with the generic path is trajectory-level, exactly like the reference's compiled path,
which is why `SignRoundConfig.fused_block` is opt-in.  Blocks this file does not recognise keep the generic path silently.
"""
from __future__ import annotations

from typing import Optional

import os

import torch
import torch.nn.functional as F

from . import ops


def _is_rmsnorm(m) -> bool:
    return (type(m).__name__.endswith("RMSNorm") and hasattr(m, "weight") and hasattr(m, "variance_epsilon")
            and getattr(m, "bias", None) is None)


def _qk_norm(attn, hd):
    """(wq, wk, eps) of Qwen3-style per-head q_norm / k_norm, None when the block has none, False when it has something else"""
    qn, kn = getattr(attn, "q_norm", None), getattr(attn, "k_norm", None)
    absent = [m is None or isinstance(m, torch.nn.Identity) for m in (qn, kn)]
    if all(absent):
        return None
    if any(absent) or not (_is_rmsnorm(qn) and _is_rmsnorm(kn)) or hd not in (64, 128, 256, 512):
        return False
    if tuple(qn.weight.shape) != (hd,) or tuple(kn.weight.shape) != (hd,) or qn.variance_epsilon != kn.variance_epsilon:
        return False
    return qn.weight, kn.weight, float(qn.variance_epsilon)


def _is_silu(act) -> bool:
    return "silu" in type(act).__name__.lower() or act is F.silu


# Decoder-layer classes whose forward IS the computation written out below.  Structural recognition alone is not enough: other
# families share every attribute name and differ in the arithmetic (Granite: residual_multiplier; SmolLM3: layers without rotary
# embedding; HunYuan: q/k norms AFTER the rotation; Ernie 4.5: interleaved rotate_half) -- those keep the generic module path.
LLAMA_FAMILY = ("LlamaDecoderLayer", "MistralDecoderLayer", "Qwen2DecoderLayer", "Qwen3DecoderLayer")
OPT_FAMILY = ("OPTDecoderLayer",)
MOE_FAMILY = ("MixtralDecoderLayer",)


def _class_in(block, names) -> bool:
    return type(block).__name__ in names


def _act_plans(proj):
    """One activation fake-quant plan per projection (None: 16-bit input).  NVFP4's static per-layer scale becomes ("nv", value) from
    the layer's calibrated `act_max`; raises NotImplementedError for anything the raw kernels do not cover."""
    from .wrapper import act_quant_plan, nv_static_plan

    plans = []
    for p in proj:
        pl = act_quant_plan(p.orig_layer, p.in_features) if p.enable_act_quant else None
        if pl is not None and pl[0] == "nv":
            pl = nv_static_plan(p.orig_layer, p.device)
            if pl is None:
                raise NotImplementedError("NVFP4 activations without a calibrated act_max (dynamic per-call maximum)")
        plans.append(pl)
    return plans


def _rotary_ok(pe, hd) -> bool:
    """(cos, sin) of the full head size: partial-rotary models (rotary_dim < head_dim) keep the module path"""
    if not (isinstance(pe, (tuple, list)) and len(pe) == 2):
        return False
    return all(isinstance(t, torch.Tensor) and t.dim() in (2, 3) and t.shape[-1] == hd for t in pe)


def disagreement(y_fused: torch.Tensor, y_module: torch.Tensor, x: torch.Tensor) -> float:
    """|| y_fused - y_module || / || y_module - x ||: how far the two paths are apart in units of what the block ADDS to its input
    (the residual stream dominates both outputs, so a distance relative to || y || would hide a wrong branch); one host read"""
    a, b = y_fused.detach().float(), y_module.detach().float()
    if a.shape != b.shape:
        return float("inf")
    ref = b - x.detach().float().reshape(b.shape) if x.numel() == b.numel() else b
    num, den = float((a - b).norm()), float(ref.norm())
    if num != num or den != den:
        return float("inf")
    if den <= 0.0:
        return 0.0 if num <= 1e-6 else float("inf")
    return max(num - 1e-6, 0.0) / den


def outputs_agree(y_fused: torch.Tensor, y_module: torch.Tensor, x: torch.Tensor, tol: float) -> bool:
    return disagreement(y_fused, y_module, x) <= tol


class FusedLlamaBlock:
    """Built per block after `wrapper_block`; `forward(x)` returns the block output connected to autograd through the arena's
    dummy token, its backward fills the arena's dWq buffer."""

    def __init__(self):
        self.use_mfma_dw = True
        self._tn = None
        self.qk_norm = None

    # -- recognition ----------------------------------------------------------------------------------------------------
    @classmethod
    def try_build(cls, block, arenas, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None, use_mfma_dw=True,
                  tn_dx_gemm=True) -> Optional["FusedLlamaBlock"]:
        from .wrapper import WrapperLinear

        try:
            n1, n2, attn, mlp = block.input_layernorm, block.post_attention_layernorm, block.self_attn, block.mlp
            proj = [attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj]
        except AttributeError:
            return None
        if not _class_in(block, LLAMA_FAMILY):
            return None
        if not arenas or not (_is_rmsnorm(n1) and _is_rmsnorm(n2)) or not _is_silu(getattr(mlp, "act_fn", None)):
            return None
        # (a sliding window reaches the SDPA call only through `attention_mask` -- transformers' sdpa_attention_forward ignores the
        # module's `sliding_window` -- and the mask is honoured below, so such blocks need nothing special)
        if not all(isinstance(p, WrapperLinear) for p in proj):
            return None
        q, k, v, o, g, u, d = proj
        # layers of one scheme and group size share an arena; the merged projections need theirs contiguous in ONE arena each
        # (per-row presets such as INT8 put down_proj, with its own row length, in a second arena)
        if any(p.padded or p.is_conv1d or not any(p.arena is a for a in arenas) for p in proj):
            return None
        if not (q.arena is k.arena is v.arena and g.arena is u.arena):
            return None
        if len({a.w_dtype for a in arenas}) != 1:
            return None
        arena = q.arena
        # dynamic activation fake-quant (INT8 / INT4 / W4A8 presets, MXFP4): one plan per GEMM input; the merged projections must
        # agree on theirs.  NVFP4's per-layer static activation scale keeps the module path.
        try:
            plans = _act_plans(proj)
        except NotImplementedError:
            return None
        if not (plans[0] == plans[1] == plans[2]) or plans[4] != plans[5]:
            return None
        if arena.w_dtype not in (torch.bfloat16, torch.float16) or arena.w_dtype != amp_dtype:
            return None
        if not (k._off == q._off + q.numel and v._off == k._off + k.numel and u._off == g._off + g.numel):
            return None
        if not (q.in_features == k.in_features == v.in_features and g.in_features == u.in_features
                and g.out_features == u.out_features and k.out_features == v.out_features):
            return None
        others = dict(input_others or {})
        pe = others.get("position_embeddings")
        if not (isinstance(pe, (tuple, list)) and len(pe) == 2) or others.get("past_key_values") is not None:
            return None
        hd = int(getattr(attn, "head_dim", 0))
        if hd <= 0 or hd % 16 or q.out_features % hd or k.out_features % hd or not _rotary_ok(pe, hd):
            return None
        hq, hkv = q.out_features // hd, k.out_features // hd
        if hq % hkv or o.in_features != hq * hd or (g.out_features % 8) or (q.in_features % 8):
            return None
        qk_norm = _qk_norm(attn, hd)
        if qk_norm is False:
            return None
        qkv_bias = [p.orig_layer.bias for p in (q, k, v)]
        if any(b is not None for b in qkv_bias) and not all(b is not None for b in qkv_bias):
            return None
        gu_bias = [p.orig_layer.bias for p in (g, u)]
        if any(b is not None for b in gu_bias) and not all(b is not None for b in gu_bias):
            return None

        self = cls()
        self.block, self.arena, self.arenas, self.attn = block, arena, list(arenas), attn
        self.layers = dict(q=q, k=k, v=v, o=o, g=g, u=u, d=d)
        self.w1, self.eps1 = n1.weight, float(n1.variance_epsilon)
        self.w2, self.eps2 = n2.weight, float(n2.variance_epsilon)
        self.hq, self.hkv, self.hd = hq, hkv, hd
        self.qk_norm = qk_norm
        self.H, self.Fdim = q.in_features, g.out_features
        self.scaling = getattr(attn, "scaling", None)
        self.dtype = arena.w_dtype
        self.sdpa_ctx = sdpa_ctx
        self.use_mfma_dw = bool(use_mfma_dw)
        self.aq = dict(qkv=plans[0], o=plans[3], gu=plans[4], d=plans[6])

        def view(first, last_numel_sum, rows, cols, buf):
            return buf[first._off:first._off + last_numel_sum].view(rows, cols)

        nqkv = q.numel + k.numel + v.numel
        self.Wqkv = view(q, nqkv, (hq + 2 * hkv) * hd, self.H, arena.Wq)
        self.dWqkv = view(q, nqkv, (hq + 2 * hkv) * hd, self.H, arena.dWq)
        self.Wo, self.dWo = o.weight_q, o.weight_grad
        self.Wgu = view(g, g.numel + u.numel, 2 * self.Fdim, self.H, g.arena.Wq)
        self.dWgu = view(g, g.numel + u.numel, 2 * self.Fdim, self.H, g.arena.dWq)
        self.Wd, self.dWd = d.weight_q, d.weight_grad
        dt = self.dtype
        self.b_qkv = torch.cat([b.to(dt) for b in qkv_bias]) if qkv_bias[0] is not None else None
        self.b_gu = torch.cat([b.to(dt) for b in gu_bias]) if gu_bias[0] is not None else None
        self.b_o = None if o.orig_layer.bias is None else o.orig_layer.bias.to(dt)
        self.b_d = None if d.orig_layer.bias is None else d.orig_layer.bias.to(dt)
        self._tn = None
        self.set_tn_dx(tn_dx_gemm)
        return self

    def set_tn_dx(self, on: bool):
        """Keep W^T next to the o / gate-up / down weights (refreshed by one transpose kernel each per iteration) so that the three
        input-gradient GEMMs dX = dY W run with both operands contiguous along the reduction -- the layout the library's tuned
        kernel covers (1.52-1.58 PFLOP/s against 1.02-1.35 for the K-strided form on the Llama-3-8B shapes,
        profiles/archive/r02_dx_gemm_layout_probe.json).  Costs one extra copy of those weights in HBM."""
        self._tn = None
        if not on or self.arena is None:
            return
        ws = self._dx_weights()
        if any(w.shape[0] % 64 or w.shape[1] % 64 or not w.is_contiguous() for w in ws) or min(min(w.shape) for w in ws) < self._tn_min_dim:
            return
        self._tn = [torch.empty((w.shape[1], w.shape[0]), dtype=w.dtype, device=w.device) for w in ws]

    _tn_min_dim = 1024

    def _dx_weights(self):
        return (self.Wo, self.Wgu, self.Wd)

    def _refresh_tn(self):
        if self._tn is not None:
            for w, wt in zip(self._dx_weights(), self._tn):
                ops.transpose16(w, out=wt)

    def _dx(self, dY2d, W, slot):
        """dY @ W through the transposed copy when there is one"""
        if self._tn is not None:
            return torch.mm(dY2d, self._tn[slot].t())
        return torch.mm(dY2d, W)

    @classmethod
    def try_build_plain(cls, block, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None) -> Optional["FusedLlamaBlock"]:
        """The no-grad form for an UNWRAPPED block (plain nn.Linear layers): the reference forward that produces the block's
        targets and the quantised-output forward that feeds the next block (composer.py steps 3 and 6) run through the same
        fused kernels as the tuning-time forward.  The q/k/v and gate/up weights are concatenated once per call (a copy of the
        block's weights against 128 samples of forward work).  Refused (None) when any projection carries forward hooks -- the
        calibration hooks (act_max, imatrix) must see the module calls -- or an activation-quant shell."""
        try:
            n1, n2, attn, mlp = block.input_layernorm, block.post_attention_layernorm, block.self_attn, block.mlp
            proj = [attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj]
        except AttributeError:
            return None
        if not _class_in(block, LLAMA_FAMILY):
            return None
        if not (_is_rmsnorm(n1) and _is_rmsnorm(n2)) or not _is_silu(getattr(mlp, "act_fn", None)):
            return None
        if not all(type(p) is torch.nn.Linear for p in proj):
            return None
        if any(p._forward_hooks or p._forward_pre_hooks for p in proj + [attn, mlp, n1, n2, block]):
            return None
        q, k, v, o, g, u, d = proj
        if any(p.weight.dtype != amp_dtype or p.weight.device.type != "cuda" for p in proj) or amp_dtype not in (torch.bfloat16, torch.float16):
            return None
        others = dict(input_others or {})
        pe = others.get("position_embeddings")
        if not (isinstance(pe, (tuple, list)) and len(pe) == 2) or others.get("past_key_values") is not None:
            return None
        hd = int(getattr(attn, "head_dim", 0))
        if hd <= 0 or hd % 16 or q.out_features % hd or k.out_features % hd or k.out_features != v.out_features or not _rotary_ok(pe, hd):
            return None
        hq, hkv = q.out_features // hd, k.out_features // hd
        if hq % hkv or o.in_features != hq * hd or (g.out_features % 8) or (q.in_features % 8) or g.out_features != u.out_features:
            return None
        qk_norm = _qk_norm(attn, hd)
        if qk_norm is False or any(m is not None and (m._forward_hooks or m._forward_pre_hooks) for m in (getattr(attn, "q_norm", None), getattr(attn, "k_norm", None))):
            return None
        for grp in ((q, k, v), (g, u)):
            b = [p.bias for p in grp]
            if any(x is not None for x in b) and not all(x is not None for x in b):
                return None
        self = cls()
        self.block, self.arena, self.attn, self.layers = block, None, attn, {}
        self._tn = None
        self.aq = dict(qkv=None, o=None, gu=None, d=None)
        self.w1, self.eps1 = n1.weight, float(n1.variance_epsilon)
        self.w2, self.eps2 = n2.weight, float(n2.variance_epsilon)
        self.hq, self.hkv, self.hd = hq, hkv, hd
        self.qk_norm = qk_norm
        self.H, self.Fdim = q.in_features, g.out_features
        self.scaling = getattr(attn, "scaling", None)
        self.dtype = amp_dtype
        self.sdpa_ctx = sdpa_ctx
        self.Wqkv = torch.cat([q.weight, k.weight, v.weight], dim=0)
        self.Wgu = torch.cat([g.weight, u.weight], dim=0)
        self.Wo, self.Wd = o.weight, d.weight
        self.b_qkv = None if q.bias is None else torch.cat([q.bias, k.bias, v.bias]).to(amp_dtype)
        self.b_gu = None if g.bias is None else torch.cat([g.bias, u.bias]).to(amp_dtype)
        self.b_o = None if o.bias is None else o.bias.to(amp_dtype)
        self.b_d = None if d.bias is None else d.bias.to(amp_dtype)
        return self

    # -- pieces ---------------------------------------------------------------------------------------------------------
    def _cos_sin(self, others, B, S):
        cos, sin = others["position_embeddings"]
        cos, sin = cos.to(self.dtype), sin.to(self.dtype)
        if cos.dim() == 2:
            cos, sin = cos.unsqueeze(0), sin.unsqueeze(0)
        if cos.shape[0] not in (1, B) or cos.shape[1] != S or cos.shape[2] != self.hd:
            raise ValueError(f"position_embeddings of shape {tuple(cos.shape)} do not fit a [{B}, {S}] batch with head_dim {self.hd}")
        return cos.contiguous(), sin.contiguous()

    def _attention(self, q2d, k2d, v2d, mask, B, S, grad: bool):
        hq, hd = self.hq, self.hd

        def heads(t):
            return t.view(B, S, hq, hd).transpose(1, 2)

        import contextlib

        from .attention import efficient_backward_ok

        if mask is None and S > 1 and getattr(self, "flash_fwd", True) and efficient_backward_ok(S):
            # hand-written causal flash-attention forward (csrc/ar_attn.hip; head size 128, S % 128 == 0); it returns the rows'
            # log-sum-exp in the form the library's attention backward consumes, so the backward stays torch's
            res = ops.attn_fwd(q2d, k2d, v2d, B, S, hq, hd, scale=self.scaling)
            if res is not None:
                out2d, lse = res
                return heads(out2d), (("flash", q2d, k2d, v2d, out2d, lse) if grad else None)
        if mask is not None and S > 1 and getattr(self, "flash_fwd", True) and efficient_backward_ok(S):
            # the calibration flow's structured mask (0 / 1 additive bias over `causal & key-is-valid`): the same kernel with the bias
            # in two registers; the library's backward gets the real bias tensor and this forward's log-sum-exp rows
            st = ops.mask_structure(mask, S)
            if st is not None:
                res = ops.attn_fwd(q2d, k2d, v2d, B, S, hq, hd, scale=self.scaling, mask_struct=st)
                if res is not None:
                    out2d, lse = res
                    return heads(out2d), (("flash", q2d, k2d, v2d, out2d, lse, mask) if grad else None)
        ctx = self.sdpa_ctx(S) if self.sdpa_ctx is not None else contextlib.nullcontext()
        if mask is not None and mask.dim() == 4:
            mask = mask[:, :, :, :S]
        with ctx:
            if grad:
                with torch.enable_grad():
                    ql, kl, vl = (heads(t).detach().requires_grad_(True) for t in (q2d, k2d, v2d))
                    out = F.scaled_dot_product_attention(ql, kl, vl, attn_mask=mask, dropout_p=0.0, scale=self.scaling,
                                                         is_causal=mask is None and S > 1)
                return out, (ql, kl, vl)
            out = F.scaled_dot_product_attention(heads(q2d), heads(k2d), heads(v2d), attn_mask=mask, dropout_p=0.0,
                                                 scale=self.scaling, is_causal=mask is None and S > 1)
            return out, None

    @staticmethod
    def _linear_residual(res2d, a2d, W, bias, inplace=False):
        """res + a @ W^T (+ bias): the residual add rides in the GEMM epilogue (one rounding).  torch's out-of-place addmm first
        copies `res` into the result; inplace=True accumulates into `res` itself."""
        if bias is None:
            return res2d.addmm_(a2d, W.t()) if inplace else torch.addmm(res2d, a2d, W.t())
        return (res2d.add_(bias) if inplace else res2d + bias).addmm_(a2d, W.t())

    def _dw(self, dY2d, X2d, out2d, layers):
        """out (+)= dY^T X into the arena; `layers` are the wrapped layers whose slices `out2d` covers."""
        acc = layers[0]._dw_accum[0]
        done = False
        if self.use_mfma_dw and out2d.is_contiguous() and mfma_dw_pays(out2d.shape[0], out2d.shape[1], dY2d.shape[0]):
            done = ops.gemm_dw(dY2d, X2d, out2d, accumulate=acc)
        if not done:
            if acc:
                out2d.addmm_(dY2d.t(), X2d)
            else:
                torch.mm(dY2d.t(), X2d, out=out2d)
        for lyr in layers:
            lyr._dw_accum[0] = True
            post = getattr(lyr, "_post_dw", None)
            if post is not None:
                post()

    # -- the two directions -----------------------------------------------------------------------------------------------
    def forward(self, x, input_others, donate_input=False):
        """donate_input: the caller's `x` is scratch (the tuning loop's gathered minibatch) and may be overwritten -- the first
        residual GEMM then accumulates into it in place instead of into a copy of it."""
        for a in self.arenas:
            if not a.wq_fresh:
                a.qdq_forward()
        self._donated = bool(donate_input)
        try:
            return _FusedBlockFn.apply(x, self.arena.token, self, input_others)
        finally:
            self._donated = False

    @torch.no_grad()
    def forward_nograd(self, x, input_others):
        return self._forward_impl(x, input_others, None)

    def forward_direct(self, x, input_others, donate_input=False):
        """The tuning-time forward without an autograd node around it -> (y, ctx) for `backward_direct`: the same kernels in the same
        order as `forward` + `.backward()`, as two plain calls (what a captured hipGraph of the iteration records)."""
        import types

        for a in self.arenas:
            if not a.wq_fresh:
                a.qdq_forward()
        self._donated = bool(donate_input)
        ctx = types.SimpleNamespace(saved=None)
        try:
            with torch.no_grad():
                y = self._forward_impl(x, input_others, ctx)
        finally:
            self._donated = False
        return y, ctx

    def backward_direct(self, ctx, dy):
        with torch.no_grad():
            self._backward_impl(ctx, dy)

    def agrees_with_module(self, module_forward, x, input_others) -> bool:
        """One minibatch through the fused kernels and through the block's own module code (`module_forward(x, others)`): the fused
        path is only used when both compute the same function (bf16 rounding apart; 4-bit activation grids amplify it).  Guards
        against look-alike blocks the class whitelist does not know about (a subclass overriding forward, a patched attention)."""
        act_quant = any(p is not None for p in self.aq.values())
        with torch.no_grad():
            if self.arena is not None:
                for a in self.arenas:
                    if not a.wq_fresh:
                        a.qdq_forward()
            try:
                y_f = self._forward_impl(x, input_others, None)
            except Exception as e:  # noqa: BLE001 -- a look-alike block the kernels do not fit (rotary tables of another shape, an
                self.last_disagreement = repr(e)          # operand a kernel refuses): the module path, not an aborted quantisation
                return False
            y_m = module_forward(x, input_others)
        # bf16 rounding noise of the two paths is a few percent of the block's own contribution on random-init blocks (less on real
        # ones); a dropped multiplier, a missing / extra rotation or another norm placement changes it by tens of percent
        self.last_disagreement = disagreement(y_f, y_m, x)
        return self.last_disagreement <= (0.35 if act_quant else 0.25)

    # -- the attention half (shared with the sparse-MoE block) --------------------------------------------------------------
    def _attn_half_forward(self, x, others, ctx):
        """RMSNorm -> merged q/k/v GEMM -> [per-head norms] -> rotary -> attention -> o-proj + residual.  -> (x2 [T, H], saved)"""
        from .wrapper import act_quant_fwd_raw

        B, S, H = x.shape
        T = B * S
        x2d = x.reshape(T, H)
        if x2d.dtype != self.dtype:
            x2d = x2d.to(self.dtype)
        x2d = x2d.contiguous()
        cos, sin = self._cos_sin(others, B, S)
        mask = others.get("attention_mask")
        aq = self.aq

        def fq(t, plan):        # the GEMM's input: fake-quantised activations where the scheme has them
            return t if plan is None else act_quant_fwd_raw(t, plan)

        if ctx is not None:
            self._refresh_tn()
        h1, _ = ops.rmsnorm_fwd(x2d, self.w1, self.eps1, want_rstd=False)
        h1 = fq(h1, aq["qkv"])                       # (the block input needs no gradient: only the quantised form is kept)
        qkv = F.linear(h1, self.Wqkv, self.b_qkv)
        qkv_raw = rstd_qk = None
        if self.qk_norm is not None:                 # Qwen3: RMSNorm of every query / key head before the rotation
            wq, wk, eps_qk = self.qk_norm
            qkv_raw = qkv
            qkv, rstd_qk = ops.headnorm_fwd(qkv_raw, wq.to(self.dtype), wk.to(self.dtype), self.hq, self.hkv, self.hd, eps_qk)
        q2d, k2d, v2d = ops.rope_fwd(qkv, cos, sin, B, S, self.hq, self.hkv, self.hd)
        del qkv
        attn, leaves = self._attention(q2d, k2d, v2d, mask, B, S, grad=ctx is not None)
        attn2d = attn.detach().transpose(1, 2).reshape(T, self.hq * self.hd)
        attn_in = fq(attn2d, aq["o"])
        x2 = self._linear_residual(x2d, attn_in, self.Wo, self.b_o, inplace=ctx is not None and getattr(self, "_donated", False))
        saved = None
        if ctx is not None:
            saved = dict(h1=h1, attn=attn, leaves=leaves, attn2d=attn2d, attn_in=attn_in, cos=cos, sin=sin, B=B, S=S, qkv_raw=qkv_raw,
                         rstd_qk=rstd_qk)
        return x2, saved

    def _attn_half_backward(self, s, dx2):
        """dx2: gradient w.r.t. the post-attention residual stream (consumed).  Fills dWo and the merged dWqkv."""
        from .wrapper import act_quant_bwd_raw

        B, S = s["B"], s["S"]
        T = B * S
        L, aq = self.layers, self.aq

        def bq(g, x, plan):
            return g if plan is None else act_quant_bwd_raw(g, x, plan)

        self._dw(dx2, s.pop("attn_in"), self.dWo, [L["o"]])
        dattn = bq(self._dx(dx2, self.Wo, 0), s.pop("attn2d"), aq["o"])
        del dx2
        attn, leaves = s.pop("attn"), s.pop("leaves")
        dattn4 = dattn.view(B, S, self.hq, self.hd).transpose(1, 2)
        def tok(t):
            return t.transpose(1, 2).contiguous().view(T, self.hq * self.hd)

        done = None
        if isinstance(leaves[0], str):      # forward was ar_attn_fwd: (tag, q2d, k2d, v2d, out2d, lse[, mask])
            _, q2d, k2d, v2d, out2d, lse = leaves[:6]
            bias = leaves[6] if len(leaves) > 6 else None
            st = ops.mask_structure(bias, S) if bias is not None else None
            if bias is not None:
                bias = bias.to(q2d.dtype).expand(B, self.hq, S, S)
            if (bias is None or st is not None) and getattr(self, "flash_bwd", True):
                # the first-party deterministic backward, token-major like rope_bwd wants: head size 64 (Llama-3.2-1B, Qwen2-0.5B ...)
                # causal or under the calibration flow's structured mask, head size 128 under that mask
                done = ops.attn_bwd(q2d, k2d, v2d, out2d, lse, dattn, B, S, self.hq, self.hd, scale=self.scaling, mask_struct=st)
            if done is None:
                h4 = lambda t: t.view(B, S, self.hq, self.hd).transpose(1, 2)
                z = torch.zeros((), dtype=torch.int64)
                dq, dk, dv, _ = torch.ops.aten._scaled_dot_product_efficient_attention_backward(
                    dattn4, h4(q2d), h4(k2d), h4(v2d), bias, h4(out2d), lse, z, z, 0.0, (True, True, True, False), bias is None, scale=self.scaling)
        else:
            dq, dk, dv = torch.autograd.grad(attn, leaves, dattn4)
        del attn, leaves
        if done is None:
            done = (tok(dq), tok(dk), tok(dv))
            del dq, dk, dv
        del dattn
        dqkv = ops.rope_bwd(done[0], done[1], done[2], s["cos"], s["sin"], B, S, self.hq, self.hkv, self.hd)
        del done
        if self.qk_norm is not None:
            wq, wk, _ = self.qk_norm
            ops.headnorm_bwd_(dqkv, s.pop("qkv_raw"), wq.to(self.dtype), wk.to(self.dtype), s.pop("rstd_qk"), self.hq, self.hkv, self.hd)
        self._dw(dqkv, s.pop("h1"), self.dWqkv, [L["q"], L["k"], L["v"]])

    # -- the two directions of the dense block --------------------------------------------------------------------------------
    def _forward_impl(self, x, others, ctx):
        from .wrapper import act_quant_fwd_raw

        B, S, H = x.shape
        aq = self.aq

        def fq(t, plan):
            return t if plan is None else act_quant_fwd_raw(t, plan)

        x2, saved = self._attn_half_forward(x, others, ctx)
        h2, rstd2 = ops.rmsnorm_fwd(x2, self.w2, self.eps2, want_rstd=ctx is not None)
        h2_in = fq(h2, aq["gu"])
        gu = F.linear(h2_in, self.Wgu, self.b_gu)
        act = ops.swiglu_fwd(gu, self.Fdim)
        act_in = fq(act, aq["d"])
        y = self._linear_residual(x2, act_in, self.Wd, self.b_d)
        if ctx is not None:
            saved.update(x2=x2, rstd2=rstd2, h2=h2, h2_in=h2_in, gu=gu, act=act, act_in=act_in)
            ctx.saved = saved
        return y.view(B, S, H)

    def _backward_impl(self, ctx, dy):
        s = ctx.saved
        ctx.saved = None
        T = s["B"] * s["S"]
        L = self.layers
        dy2d = dy.reshape(T, self.H)
        if dy2d.dtype != self.dtype:
            dy2d = dy2d.to(self.dtype)
        dy2d = dy2d.contiguous()
        from .wrapper import act_quant_bwd_raw

        aq = self.aq

        def bq(g, x, plan):     # gradient w.r.t. the quantised activation -> gradient w.r.t. the activation
            return g if plan is None else act_quant_bwd_raw(g, x, plan)

        # MLP
        self._dw(dy2d, s.pop("act_in"), self.dWd, [L["d"]])
        da = bq(self._dx(dy2d, self.Wd, 2), s.pop("act"), aq["d"])
        dgu = ops.swiglu_bwd_(da, s.pop("gu"), self.Fdim)
        del da
        self._dw(dgu, s.pop("h2_in"), self.dWgu, [L["g"], L["u"]])
        dh2 = bq(self._dx(dgu, self.Wgu, 1), s.pop("h2"), aq["gu"])
        del dgu
        dx2 = ops.rmsnorm_bwd(dh2, s.pop("x2"), self.w2, s.pop("rstd2"), dres=dy2d, out=dh2)
        self._attn_half_backward(s, dx2)


class FusedOPTBlock(FusedLlamaBlock):
    """The same treatment for OPT-style decoder blocks (transformers/models/opt/modeling_opt.py OPTDecoderLayer with
    do_layer_norm_before: LayerNorm -> q/k/v (+bias, q scaled) -> causal attention -> out_proj + residual -> LayerNorm -> fc1 ->
    ReLU -> fc2 + residual).  LayerNorm forward / backward are csrc/ar_block.hip kernels (the module path runs them in fp32 with
    two dtype conversions around each), q/k/v run as one GEMM, the residual adds ride in the GEMM epilogues, weight gradients go
    straight into the arena through the MFMA kernel.  BASELINE configs[0] (OPT-125M) is this block."""

    @staticmethod
    def _parts(block):
        if not _class_in(block, OPT_FAMILY):
            return None
        try:
            attn = block.self_attn
            return (block.self_attn_layer_norm, block.final_layer_norm, attn,
                    [attn.q_proj, attn.k_proj, attn.v_proj, attn.out_proj, block.fc1, block.fc2])
        except AttributeError:
            return None

    @staticmethod
    def _shape_ok(block, n1, n2, attn, q, k, v, o, f1, f2):
        if not (isinstance(n1, torch.nn.LayerNorm) and isinstance(n2, torch.nn.LayerNorm)) or not getattr(block, "do_layer_norm_before", False):
            return None
        if n1.weight is None or n2.weight is None or not isinstance(getattr(block, "activation_fn", None), torch.nn.ReLU):
            return None
        if float(getattr(block, "dropout", 0.0) or 0.0) != 0.0 and block.training:
            return None
        hd = int(getattr(attn, "head_dim", 0))
        H = q.in_features
        if hd <= 0 or hd % 16 or H % hd or not (q.out_features == k.out_features == v.out_features == H == o.in_features == o.out_features):
            return None
        if not (k.in_features == v.in_features == H == f1.in_features == f2.out_features and f1.out_features == f2.in_features) or H % 8 or f1.out_features % 8:
            return None
        return hd

    @classmethod
    def try_build(cls, block, arenas, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None, use_mfma_dw=True,
                  tn_dx_gemm=True) -> Optional["FusedOPTBlock"]:
        from .wrapper import WrapperLinear

        parts = cls._parts(block)
        if parts is None or not arenas:
            return None
        n1, n2, attn, proj = parts
        if not all(isinstance(p, WrapperLinear) for p in proj):
            return None
        q, k, v, o, f1, f2 = proj
        hd = cls._shape_ok(block, n1, n2, attn, q, k, v, o, f1, f2)
        if hd is None:
            return None
        if any(p.padded or p.is_conv1d or not any(p.arena is a for a in arenas) for p in proj) or not (q.arena is k.arena is v.arena):
            return None
        if len({a.w_dtype for a in arenas}) != 1 or q.arena.w_dtype != amp_dtype or amp_dtype not in (torch.bfloat16, torch.float16):
            return None
        # OPTAttention declares k_proj, v_proj, q_proj in that order, and the arena follows declaration order: the merged GEMM
        # takes the three in whatever order they sit in the arena
        trio = sorted([("q", q), ("k", k), ("v", v)], key=lambda t: t[1]._off)
        if not (trio[1][1]._off == trio[0][1]._off + trio[0][1].numel and trio[2][1]._off == trio[1][1]._off + trio[1][1].numel):
            return None
        others = dict(input_others or {})
        if others.get("past_key_values") is not None:
            return None
        qkv_bias = [t[1].orig_layer.bias for t in trio]
        if any(b is not None for b in qkv_bias) and not all(b is not None for b in qkv_bias):
            return None
        try:
            plans = _act_plans(proj)
        except NotImplementedError:
            return None
        if not (plans[0] == plans[1] == plans[2]):
            return None
        self = cls()
        arena = q.arena
        self.block, self.arena, self.arenas, self.attn = block, arena, list(arenas), attn
        self.layers = dict(q=q, k=k, v=v, o=o, f1=f1, f2=f2)
        self.n1, self.n2 = n1, n2
        self.hq = self.hkv = q.out_features // hd
        self.hd, self.H, self.Fdim = hd, q.in_features, f1.out_features
        self.qscale = float(getattr(attn, "scaling", hd ** -0.5))
        self._set_q_scaling()
        self.dtype = arena.w_dtype
        self.sdpa_ctx = sdpa_ctx
        self.use_mfma_dw = bool(use_mfma_dw)
        self.aq = dict(qkv=plans[0], o=plans[3], f1=plans[4], f2=plans[5])
        n, first = q.numel + k.numel + v.numel, trio[0][1]._off
        self.order = [t[0] for t in trio]
        self.trio = [t[1] for t in trio]
        self.Wqkv = arena.Wq[first:first + n].view(3 * self.H, self.H)
        self.dWqkv = arena.dWq[first:first + n].view(3 * self.H, self.H)
        self.Wo, self.dWo = o.weight_q, o.weight_grad
        self.W1, self.dW1 = f1.weight_q, f1.weight_grad
        self.W2, self.dW2 = f2.weight_q, f2.weight_grad
        dt = self.dtype
        self.b_qkv = torch.cat([b.to(dt) for b in qkv_bias]) if qkv_bias[0] is not None else None
        self.b_o, self.b_1, self.b_2 = (None if p.orig_layer.bias is None else p.orig_layer.bias.to(dt) for p in (o, f1, f2))
        self.set_tn_dx(tn_dx_gemm)
        return self

    def _dx_weights(self):          # (OPT-125M's 768-wide weights stay below _tn_min_dim: no gain measured there)
        return (self.Wo, self.W1, self.W2)

    def _set_q_scaling(self):
        """OPTAttention computes q_proj(x) * head_dim^-0.5 in the activation dtype and calls the attention with scaling = 1.  When that
        factor is a power of two (head size 64: 0.125) the product is exact, so the factor is folded into the attention's softmax
        scale instead -- same bits, one elementwise pass less in each direction -- and q / k / v are read as column slices of the
        merged projection output (no copies).  Other head sizes keep the explicit multiply."""
        import math

        m, _ = math.frexp(self.qscale)
        self.fold_qscale = (m == 0.5 and self.qscale > 0)
        self.scaling = self.qscale if self.fold_qscale else 1.0

    @classmethod
    def try_build_plain(cls, block, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None) -> Optional["FusedOPTBlock"]:
        parts = cls._parts(block)
        if parts is None:
            return None
        n1, n2, attn, proj = parts
        if not all(type(p) is torch.nn.Linear for p in proj):
            return None
        q, k, v, o, f1, f2 = proj
        hd = cls._shape_ok(block, n1, n2, attn, q, k, v, o, f1, f2)
        if hd is None or any(p._forward_hooks or p._forward_pre_hooks for p in proj + [attn, n1, n2, block]):
            return None
        if any(p.weight.dtype != amp_dtype or p.weight.device.type != "cuda" for p in proj) or amp_dtype not in (torch.bfloat16, torch.float16):
            return None
        if (input_others or {}).get("past_key_values") is not None:
            return None
        b = [p.bias for p in (q, k, v)]
        if any(x is not None for x in b) and not all(x is not None for x in b):
            return None
        self = cls()
        self.block, self.arena, self.arenas, self.attn, self.layers = block, None, [], attn, {}
        self.n1, self.n2 = n1, n2
        self.hq = self.hkv = q.out_features // hd
        self.hd, self.H, self.Fdim = hd, q.in_features, f1.out_features
        self.qscale = float(getattr(attn, "scaling", hd ** -0.5))
        self._set_q_scaling()
        self.dtype = amp_dtype
        self.sdpa_ctx = sdpa_ctx
        self.aq = dict(qkv=None, o=None, f1=None, f2=None)
        self.order = ["q", "k", "v"]
        self.Wqkv = torch.cat([q.weight, k.weight, v.weight], dim=0)
        self.Wo, self.W1, self.W2 = o.weight, f1.weight, f2.weight
        self.b_qkv = None if q.bias is None else torch.cat([q.bias, k.bias, v.bias]).to(amp_dtype)
        self.b_o, self.b_1, self.b_2 = (None if p.bias is None else p.bias.to(amp_dtype) for p in (o, f1, f2))
        return self

    def _ln(self, n, x2d, want_stats):
        dt = self.dtype
        return ops.layernorm_fwd(x2d, n.weight.to(dt), None if n.bias is None else n.bias.to(dt), float(n.eps), want_stats=want_stats)

    def _forward_impl(self, x, others, ctx):
        from .wrapper import act_quant_fwd_raw

        B, S, H = x.shape
        T = B * S
        aq = self.aq

        def fq(t, plan):
            return t if plan is None else act_quant_fwd_raw(t, plan)

        x2d = x.reshape(T, H)
        if x2d.dtype != self.dtype:
            x2d = x2d.to(self.dtype)
        x2d = x2d.contiguous()
        mask = others.get("attention_mask")
        if ctx is not None:
            self._refresh_tn()
        h1, _, _ = self._ln(self.n1, x2d, False)
        h1 = fq(h1, aq["qkv"])
        qkv = F.linear(h1, self.Wqkv, self.b_qkv)
        at = {n: i * H for i, n in enumerate(self.order)}
        q2d = qkv[:, at["q"]:at["q"] + H]
        if not self.fold_qscale:
            q2d = q2d * self.qscale                         # OPTAttention: q_proj(x) * scaling, in the activation dtype
        k2d, v2d = qkv[:, at["k"]:at["k"] + H], qkv[:, at["v"]:at["v"] + H]      # column slices: the attention kernels take the stride
        del qkv
        attn, leaves = self._attention(q2d, k2d, v2d, mask, B, S, grad=ctx is not None)
        attn2d = attn.detach().transpose(1, 2).reshape(T, H)
        attn_in = fq(attn2d, aq["o"])
        x2 = self._linear_residual(x2d, attn_in, self.Wo, self.b_o, inplace=ctx is not None and getattr(self, "_donated", False))
        h2, mean2, rstd2 = self._ln(self.n2, x2, ctx is not None)
        h2_in = fq(h2, aq["f1"])
        a = torch.relu_(F.linear(h2_in, self.W1, self.b_1))     # (bias + ReLU as a GEMM epilogue, torch._addmm_activation: no gain measured)
        a_in = fq(a, aq["f2"])
        y = self._linear_residual(x2, a_in, self.W2, self.b_2)
        if ctx is not None:
            ctx.saved = dict(h1=h1, attn=attn, leaves=leaves, attn2d=attn2d, attn_in=attn_in, x2=x2, mean2=mean2, rstd2=rstd2, h2=h2,
                             h2_in=h2_in, a=a, a_in=a_in, B=B, S=S)
        return y.view(B, S, H)

    def _backward_impl(self, ctx, dy):
        from .wrapper import act_quant_bwd_raw

        s = ctx.saved
        ctx.saved = None
        B, S = s["B"], s["S"]
        T, H = B * S, self.H
        L, aq = self.layers, self.aq

        def bq(g, x, plan):
            return g if plan is None else act_quant_bwd_raw(g, x, plan)

        dy2d = dy.reshape(T, H)
        if dy2d.dtype != self.dtype:
            dy2d = dy2d.to(self.dtype)
        dy2d = dy2d.contiguous()
        # MLP
        self._dw(dy2d, s.pop("a_in"), self.dW2, [L["f2"]])
        a = s.pop("a")
        da = bq(self._dx(dy2d, self.W2, 2), a, aq["f2"])
        df = torch.ops.aten.threshold_backward(da, a, 0)     # ReLU
        del da, a
        self._dw(df, s.pop("h2_in"), self.dW1, [L["f1"]])
        dh2 = bq(self._dx(df, self.W1, 1), s.pop("h2"), aq["f1"])
        del df
        dt = self.dtype
        dx2 = ops.layernorm_bwd(dh2, s.pop("x2"), self.n2.weight.to(dt), s.pop("mean2"), s.pop("rstd2"), dres=dy2d, out=dh2)
        # attention
        self._dw(dx2, s.pop("attn_in"), self.dWo, [L["o"]])
        dattn = bq(self._dx(dx2, self.Wo, 0), s.pop("attn2d"), aq["o"])
        del dx2
        attn, leaves = s.pop("attn"), s.pop("leaves")
        dattn4 = dattn.view(B, S, self.hq, self.hd).transpose(1, 2)
        dqkv = torch.empty(T, 3 * H, dtype=self.dtype, device=dy2d.device)
        at = {n: i * H for i, n in enumerate(self.order)}
        grads = None
        if isinstance(leaves[0], str):
            _, q2d, k2d, v2d, out2d, lse = leaves[:6]
            bias = leaves[6] if len(leaves) > 6 else None
            st = ops.mask_structure(bias, S) if bias is not None else None
            if bias is not None:
                bias = bias.to(q2d.dtype).expand(B, self.hq, S, S)
            done = None
            if (bias is None or st is not None) and getattr(self, "flash_bwd", True):
                # hand-written deterministic backward (csrc/ar_attn_bwd.hip: head size 64, S % 256 == 0): reads q / k / v where the
                # merged projection left them and writes dq / dk / dv into their columns of dqkv -- no transposes, no copies
                # (causal, or the calibration flow's structured mask)
                done = ops.attn_bwd(q2d, k2d, v2d, out2d, lse, dattn, B, S, self.hq, self.hd, scale=self.scaling, mask_struct=st,
                                    dq=dqkv[:, at["q"]:at["q"] + H], dk=dqkv[:, at["k"]:at["k"] + H], dv=dqkv[:, at["v"]:at["v"] + H])
            if done is not None:
                if not self.fold_qscale:
                    dqkv[:, at["q"]:at["q"] + H].mul_(self.qscale)
            else:
                h4 = lambda t: t.view(B, S, self.hq, self.hd).transpose(1, 2)
                z = torch.zeros((), dtype=torch.int64)
                grads = torch.ops.aten._scaled_dot_product_efficient_attention_backward(
                    dattn4, h4(q2d), h4(k2d), h4(v2d), bias, h4(out2d), lse, z, z, 0.0, (True, True, True, False), bias is None, scale=self.scaling)[:3]
        else:
            grads = torch.autograd.grad(attn, leaves, dattn4)
        del attn, leaves
        if grads is not None:
            for n, g in zip("qkv", grads):
                dst = dqkv[:, at[n]:at[n] + H].view(B, S, self.hq, self.hd)
                if n == "q" and not self.fold_qscale:
                    torch.mul(g.transpose(1, 2), self.qscale, out=dst)
                else:
                    dst.copy_(g.transpose(1, 2))
        del grads, dattn
        self._dw(dqkv, s.pop("h1"), self.dWqkv, self.trio)


class FusedMoEBlock(FusedLlamaBlock):
    """Mixtral-style sparse-MoE decoder block (transformers MixtralDecoderLayer whose fused 3-D expert parameters were unfused by
    `moe_unfuse.unfuse_moe_experts` -- the reference's "linear_loop" experts, auto_round/modeling/fused_moe/
    moe_experts_interface.py): the attention half runs through the same kernels as the dense block; the expert half is ONE pass
    over rows sorted by expert instead of a Python loop of index / SiLU / product / index_add_ ops per expert:

        router (the module's own code, kept in a local autograd graph)      -> top-k weights [T, K], indices [T, K]
        one stable sort of the (slot, token) pairs, one host read of the per-expert counts
        xs = gather(h2)  [T K, H]  -> ONE activation fake-quant (gate and up share their input)
        per expert e with rows r_e:  GU[r_e] = xs_q[r_e] Wgu_e^T   (gate / up merged: adjacent in the arena)
        act = SwiGLU(GU) (one launch over all rows) -> ONE activation fake-quant
        per expert:                  D[r_e] = act_q[r_e] Wd_e^T
        y = x2 + sum_k w[t, k] D[pos[t, k]]          (ar_moe_combine: fp32 sum, one rounding, residual add fused, no atomics)

    and the mirrored backward (the routing-weight gradient is a row dot product, sent back through the router's local graph).

    Round 5: the per-expert GEMMs are GROUPED launches (`ops.gemm_nt_grouped` for the forward and -- through per-expert transposed
    weight copies refreshed once per iteration -- the input gradients, `ops.gemm_dw_grouped` for the weight gradients): one launch per
    projection whose row ranges are a device-side prefix sum of the routing counts, so the iteration has NO host read of the counts
    when every expert shares one activation-quant plan (MXFP4, INT; NVFP4's static per-expert scales still need the counts for their
    per-expert fake-quant launches), and its result does not depend on the library's choice of kernel per ragged row count (run-to-run
    reproducible).  Shapes the grouped kernels do not take (widths that are not multiples of 256) keep the per-expert loop."""

    capturable = False
    grouped = os.environ.get("AR_MOE_GROUPED", "1") != "0"     # (tests / A-B: False or AR_MOE_GROUPED=0 forces the per-expert loop of rounds 3-4)

    @classmethod
    def try_build(cls, block, arenas, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None, use_mfma_dw=True,
                  tn_dx_gemm=True) -> Optional["FusedMoEBlock"]:
        from .moe_unfuse import expert_children, is_linear_loop_experts
        from .wrapper import WrapperLinear

        if not _class_in(block, MOE_FAMILY) or not arenas:
            return None
        try:
            n1, n2, attn, moe = block.input_layernorm, block.post_attention_layernorm, block.self_attn, block.mlp
            q, k, v, o = attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj
            router, experts = moe.gate, moe.experts
        except AttributeError:
            return None
        # (the experts in their unfused "linear loop" form: this package's own unfusing or the reference's -- behind the reference's
        #  front door the block arrives with ITS numbered expert containers, moe_experts_interface.py:173-289)
        if not (_is_rmsnorm(n1) and _is_rmsnorm(n2)) or not is_linear_loop_experts(experts) or not _is_silu(getattr(experts, "act_fn", None)):
            return None
        if float(getattr(moe, "jitter_noise", 0.0) or 0.0) != 0.0 and block.training:
            return None
        rw = getattr(router, "weight", None)
        kids = expert_children(experts)
        if not isinstance(rw, torch.Tensor) or rw.dim() != 2 or not kids or len(kids) != rw.shape[0] or not hasattr(router, "top_k"):
            return None
        try:
            trip = [(e.gate_proj, e.up_proj, e.down_proj) for e in kids]
        except AttributeError:
            return None
        proj = [q, k, v, o] + [p for t in trip for p in t]
        if not all(isinstance(p, WrapperLinear) for p in proj):
            return None
        if any(p.padded or p.is_conv1d or p.orig_layer.bias is not None or not any(p.arena is a for a in arenas) for p in proj):
            return None
        if not (q.arena is k.arena is v.arena) or len({a.w_dtype for a in arenas}) != 1 or q.arena.w_dtype != amp_dtype \
                or amp_dtype not in (torch.bfloat16, torch.float16):
            return None
        if not (k._off == q._off + q.numel and v._off == k._off + k.numel):
            return None
        H, Fd = q.in_features, trip[0][0].out_features
        for g, u, d in trip:
            if not (g.arena is u.arena and u._off == g._off + g.numel and g.in_features == u.in_features == H == d.out_features
                    and g.out_features == u.out_features == Fd == d.in_features):
                return None
        try:
            plans = _act_plans(proj)
        except NotImplementedError:
            return None
        if not (plans[0] == plans[1] == plans[2]) or any(plans[4 + 3 * i] != plans[5 + 3 * i] for i in range(len(trip))):
            return None
        # one plan per expert and projection input: dynamic schemes (MXFP4, INT) give every expert the same one -- a single launch
        # over all sorted rows; NVFP4's static per-layer scale differs between experts -- one launch per expert's rows
        pl_gu_e = [plans[4 + 3 * i] for i in range(len(trip))]
        pl_d_e = [plans[6 + 3 * i] for i in range(len(trip))]
        if any((pl is None) != (pl_gu_e[0] is None) or (pl is not None and pl[0] != pl_gu_e[0][0]) for pl in pl_gu_e):
            return None
        if any((pl is None) != (pl_d_e[0] is None) or (pl is not None and pl[0] != pl_d_e[0][0]) for pl in pl_d_e):
            return None
        pl_gu, pl_d = pl_gu_e[0], pl_d_e[0]
        others = dict(input_others or {})
        pe = others.get("position_embeddings")
        hd = int(getattr(attn, "head_dim", 0))
        if others.get("past_key_values") is not None or hd <= 0 or hd % 16 or q.out_features % hd or k.out_features % hd or not _rotary_ok(pe, hd):
            return None
        hq, hkv = q.out_features // hd, k.out_features // hd
        if hq % hkv or o.in_features != hq * hd or Fd % 8 or H % 8 or k.out_features != v.out_features or _qk_norm(attn, hd) is not None:
            return None

        self = cls()
        arena = q.arena
        self.block, self.arena, self.arenas, self.attn = block, arena, list(arenas), attn
        self.layers = dict(q=q, k=k, v=v, o=o)
        self.trip = trip
        self.router, self.top_k, self.E = router, int(router.top_k), len(trip)
        self.w1, self.eps1 = n1.weight, float(n1.variance_epsilon)
        self.w2, self.eps2 = n2.weight, float(n2.variance_epsilon)
        self.hq, self.hkv, self.hd = hq, hkv, hd
        self.qk_norm = None
        self.H, self.Fdim = H, Fd
        self.scaling = getattr(attn, "scaling", None)
        self.dtype = arena.w_dtype
        self.sdpa_ctx = sdpa_ctx
        self.use_mfma_dw = bool(use_mfma_dw)
        self.aq = dict(qkv=plans[0], o=plans[3], gu=pl_gu, d=pl_d)
        self.pl_gu_e, self.pl_d_e = pl_gu_e, pl_d_e
        nqkv = q.numel + k.numel + v.numel
        self.Wqkv = arena.Wq[q._off:q._off + nqkv].view((hq + 2 * hkv) * hd, H)
        self.dWqkv = arena.dWq[q._off:q._off + nqkv].view((hq + 2 * hkv) * hd, H)
        self.Wo, self.dWo = o.weight_q, o.weight_grad
        self.Wgu = [g.arena.Wq[g._off:g._off + 2 * g.numel].view(2 * Fd, H) for g, u, d in trip]
        self.dWgu = [g.arena.dWq[g._off:g._off + 2 * g.numel].view(2 * Fd, H) for g, u, d in trip]
        self.Wd = [d.weight_q for g, u, d in trip]
        self.dWd = [d.weight_grad for g, u, d in trip]
        self.b_qkv = self.b_o = None
        self.set_tn_dx(tn_dx_gemm)
        # grouped expert GEMMs: every expert's layers in ONE arena (offsets relative to its flat buffers), widths the kernels take
        ea = trip[0][0].arena
        self._grp = None
        # (every expert offset a multiple of 8 elements: the grouped kernels move 16 bytes per LDS-DMA lane and store 8; an arena layer
        #  with an odd element count in front of the experts would misalign them -- ADVICE r05)
        if (all(l.arena is ea for t in trip for l in t) and ea.w_dtype == torch.bfloat16 and H % 256 == 0 and Fd % 256 == 0
                and all(l._off % 8 == 0 for t in trip for l in t)):
            dev = ea.Wq.device
            self._grp = dict(arena=ea,
                             off_gu=torch.tensor([g._off for g, u, d in trip], dtype=torch.int64, device=dev),
                             off_d=torch.tensor([d._off for g, u, d in trip], dtype=torch.int64, device=dev),
                             offT_gu=torch.arange(len(trip), dtype=torch.int64, device=dev) * (H * 2 * Fd),
                             offT_d=torch.arange(len(trip), dtype=torch.int64, device=dev) * (Fd * H),
                             WguT=None, WdT=None)
        self._uniform_plans = all(pl == pl_gu_e[0] for pl in pl_gu_e) and all(pl == pl_d_e[0] for pl in pl_d_e)
        return self

    def _use_grouped(self) -> bool:
        return self.grouped and self._grp is not None

    def _refresh_expert_transposes(self):
        """W_e^T copies for the grouped input-gradient GEMMs (dX = dY W_e is an "NN" product; the NT kernel wants the reduction
        dimension contiguous in both operands): [E, H, 2F] and [E, F, H], rewritten once per iteration from the fresh fake-quant weights
        (16 transposes, ~1 ms of a ~90 ms Mixtral iteration)."""
        g = self._grp
        E, H, Fd = self.E, self.H, self.Fdim
        if g["WguT"] is None:
            g["WguT"] = torch.empty((E, H, 2 * Fd), dtype=self.dtype, device=g["arena"].Wq.device)
            g["WdT"] = torch.empty((E, Fd, H), dtype=self.dtype, device=g["arena"].Wq.device)
        for e in range(E):
            ops.transpose16(self.Wgu[e], out=g["WguT"][e])
            ops.transpose16(self.Wd[e], out=g["WdT"][e])

    def _dx_weights(self):
        """Only the o-projection keeps a transposed copy.  (Transposed copies of every expert's merged gate/up and down weights were
        measured in round 3: 2.9 GB more HBM and 16 transposes per iteration for expert dX GEMMs in the library's fast layout --
        9.90 s per Mixtral block against 9.71 s without them.)"""
        return (self.Wo,)

    @staticmethod
    def _mm_rows(a, w, out):
        """out[M, N] = a[M, K] @ w[K, N] for ONE expert's rows.  The library GEMM works in rounds of 256 output tiles (256 x 256, one
        per CU) and has no stream-K form for these shapes: an expert with a few rows more than a whole number of rounds pays a whole
        extra round -- at N = 4096 (down-projection forward, gate/up input gradient) 4097..4608 rows cost 2.0-2.4x what 4096 do
        (profiles/r03_moe_expert_gemm_vs_rows.jsonl).  Where the product is only a few rounds deep the rows are cut at the last
        whole round and the remainder goes in a second, short call (measured: 720 -> 330 us and 1320 -> 830 us at 4100 rows)."""
        M, N = a.shape[0], w.shape[1]
        col_tiles = -(-N // 256)
        if col_tiles <= 16:                                             # (wider outputs: many rounds, the tail is noise -- measured)
            per_round = (256 // col_tiles) * 256                        # rows that fill the chip exactly once
            whole = (M // per_round) * per_round
            if 0 < whole < M and -(-M // 256) * col_tiles <= 6 * 256:
                torch.mm(a[:whole], w, out=out[:whole])
                torch.mm(a[whole:], w, out=out[whole:])
                return out
        return torch.mm(a, w, out=out)

    @staticmethod
    def _host_counts(r):
        """the per-expert row counts on the host, read (once per iteration) only when a per-expert loop needs them"""
        if r["counts"] is None:
            ro = r["row_off"].tolist()
            r["counts"] = [ro[i + 1] - ro[i] for i in range(len(ro) - 1)]
        return r["counts"]

    @staticmethod
    def _dw_done(layers) -> bool:
        """bookkeeping after a grouped weight-gradient launch wrote EVERY listed layer's gradient (zeros for an expert without rows)"""
        for lyr in layers:
            lyr._dw_accum[0] = True
            post = getattr(lyr, "_post_dw", None)
            if post is not None:
                post()
        return True

    @staticmethod
    def _rows_fq(t, plans, counts, raw, grad_of=None):
        """Activation fake-quant (raw = act_quant_fwd_raw) or its backward (raw = act_quant_bwd_raw, grad_of = the activation) of the
        sorted rows: one launch when every expert has the same plan, else one per expert's row segment."""
        if plans[0] is None:
            return t
        if all(pl == plans[0] for pl in plans):
            return raw(t, plans[0]) if grad_of is None else raw(t, grad_of, plans[0])
        out = torch.empty_like(t)
        start = 0
        for e, cnt in enumerate(counts):
            if cnt:
                rows = slice(start, start + cnt)
                if grad_of is None:
                    raw(t[rows], plans[e], out=out[rows])
                else:
                    raw(t[rows], grad_of[rows], plans[e], out=out[rows])
            start += cnt
        return out

    def _route(self, h2, grad):
        """The module's own router on the normalised stream (its backward through a local autograd graph), then the sorted-row
        bookkeeping: tok[p] / pos[t, k] / per-expert counts (the one host read)."""
        T = h2.shape[0]
        leaf = None
        if grad:
            with torch.enable_grad():
                leaf = h2.detach().requires_grad_(True)
                _, rw, ri = self.router(leaf)
        else:
            _, rw, ri = self.router(h2)
        with torch.no_grad():
            K = ri.shape[1]
            flat = ri.t().reshape(-1)                                   # index = slot * T + token (slot-major; the module path sorts token-major since round 5)
            order = torch.argsort(flat, stable=True)
            # per-expert row counts WITHOUT a host read (torch.bincount reads its input's maximum on the host): a compare-and-sum
            cnt = (flat.unsqueeze(1) == torch.arange(self.E, device=flat.device, dtype=flat.dtype)).sum(dim=0)
            row_off = torch.zeros(self.E + 1, dtype=torch.int32, device=flat.device)
            row_off[1:] = torch.cumsum(cnt, dim=0)
            # the counts on the host only where something still needs them: the per-expert loop, or per-expert activation-quant plans
            counts = None if (self._use_grouped() and self._uniform_plans) else cnt.tolist()
            tok = (order % T).contiguous()
            inv = torch.empty_like(order)
            inv[order] = torch.arange(order.numel(), device=order.device, dtype=order.dtype)
            pos = inv.view(K, T).t().contiguous()                       # [T, K]: row of token t's k-th routed copy
            w_tk = rw.detach().to(torch.float32).contiguous()           # [T, K]
            w_sorted = w_tk[tok, order // T].contiguous()               # [T K]
        return dict(leaf=leaf, rw=rw, counts=counts, row_off=row_off, tok=tok, pos=pos, w_tk=w_tk, w_sorted=w_sorted)

    def _forward_impl(self, x, others, ctx):
        from .wrapper import act_quant_fwd_raw

        B, S, H = x.shape
        x2, saved = self._attn_half_forward(x, others, ctx)
        h2, rstd2 = ops.rmsnorm_fwd(x2, self.w2, self.eps2, want_rstd=ctx is not None)
        r = self._route(h2, grad=ctx is not None)
        xs = ops.moe_expand(h2, r["tok"])
        xs_q = self._rows_fq(xs, self.pl_gu_e, r["counts"], act_quant_fwd_raw)
        R = xs.shape[0]
        GU = torch.empty((R, 2 * self.Fdim), dtype=self.dtype, device=x.device)
        grp = self._grp if self._use_grouped() else None
        if grp is None or not ops.gemm_nt_grouped(xs_q, grp["arena"].Wq, GU, r["row_off"], grp["off_gu"], 2 * self.Fdim, H):
            start = 0
            for e, cnt in enumerate(self._host_counts(r)):
                if cnt:
                    self._mm_rows(xs_q[start:start + cnt], self.Wgu[e].t(), GU[start:start + cnt])
                start += cnt
        act = ops.swiglu_fwd(GU, self.Fdim)
        act_q = self._rows_fq(act, self.pl_d_e, r["counts"], act_quant_fwd_raw)
        D = torch.empty((R, H), dtype=self.dtype, device=x.device)
        if grp is None or not ops.gemm_nt_grouped(act_q, grp["arena"].Wq, D, r["row_off"], grp["off_d"], H, self.Fdim):
            start = 0
            for e, cnt in enumerate(self._host_counts(r)):
                if cnt:
                    self._mm_rows(act_q[start:start + cnt], self.Wd[e].t(), D[start:start + cnt])
                start += cnt
        y = ops.moe_combine(D, r["pos"], r["w_tk"], res=x2)
        if ctx is not None:
            saved.update(x2=x2, rstd2=rstd2, route=r, xs=xs, xs_q=xs_q, GU=GU, act=act, act_q=act_q, D=D)
            ctx.saved = saved
        return y.view(B, S, H)

    def _backward_impl(self, ctx, dy):
        from .wrapper import act_quant_bwd_raw

        s = ctx.saved
        ctx.saved = None
        T = s["B"] * s["S"]
        dy2d = dy.reshape(T, self.H)
        if dy2d.dtype != self.dtype:
            dy2d = dy2d.to(self.dtype)
        dy2d = dy2d.contiguous()
        r = s.pop("route")
        tok, pos, counts = r["tok"], r["pos"], r["counts"]
        D = s.pop("D")
        dD = ops.moe_expand(dy2d, tok, scale=r["w_sorted"])             # d (down-projection output) = dt(w * dy[token])
        drw = ops.moe_rowdot(dy2d, tok, D)[pos]                         # [T, K]: d loss / d routing weight
        del D
        act_q = s.pop("act_q")
        dact_q = torch.empty_like(act_q)
        grp = self._grp if self._use_grouped() else None
        if grp is not None and any(l._dw_accum[0] for t in self.trip for l in t):
            grp = None                       # accumulating micro-batches: the per-expert addmm_ path
        if grp is not None:
            self._refresh_expert_transposes()
        d_layers = [t[2] for t in self.trip]
        if not (grp is not None and ops.gemm_dw_grouped(dD, act_q, grp["arena"].dWq, r["row_off"], grp["off_d"], self.Fdim) and self._dw_done(d_layers)):
            start = 0
            for e, cnt in enumerate(self._host_counts(r)):
                if cnt:
                    rows = slice(start, start + cnt)
                    self._dw(dD[rows], act_q[rows], self.dWd[e], [self.trip[e][2]])
                start += cnt
        if grp is None or not ops.gemm_nt_grouped(dD, grp["WdT"], dact_q, r["row_off"], grp["offT_d"], self.Fdim, self.H):
            start = 0
            for e, cnt in enumerate(self._host_counts(r)):
                if cnt:
                    rows = slice(start, start + cnt)
                    self._mm_rows(dD[rows], self.Wd[e], dact_q[rows])
                start += cnt
        del dD, act_q
        dact = self._rows_fq(dact_q, self.pl_d_e, counts, act_quant_bwd_raw, grad_of=s.pop("act"))
        dGU = ops.swiglu_bwd_(dact, s.pop("GU"), self.Fdim)
        del dact, dact_q
        xs_q = s.pop("xs_q")
        dxs_q = torch.empty_like(xs_q)
        gu_layers = [l for t in self.trip for l in t[:2]]
        if not (grp is not None and ops.gemm_dw_grouped(dGU, xs_q, grp["arena"].dWq, r["row_off"], grp["off_gu"], self.H) and self._dw_done(gu_layers)):
            start = 0
            for e, cnt in enumerate(self._host_counts(r)):
                if cnt:
                    rows = slice(start, start + cnt)
                    self._dw(dGU[rows], xs_q[rows], self.dWgu[e], [self.trip[e][0], self.trip[e][1]])
                start += cnt
        if grp is None or not ops.gemm_nt_grouped(dGU, grp["WguT"], dxs_q, r["row_off"], grp["offT_gu"], self.H, 2 * self.Fdim):
            start = 0
            for e, cnt in enumerate(self._host_counts(r)):
                if cnt:
                    rows = slice(start, start + cnt)
                    self._mm_rows(dGU[rows], self.Wgu[e], dxs_q[rows])
                start += cnt
        del dGU, xs_q
        dxs = self._rows_fq(dxs_q, self.pl_gu_e, counts, act_quant_bwd_raw, grad_of=s.pop("xs"))
        (dh2_router,) = torch.autograd.grad(r["rw"], r["leaf"], drw.to(r["rw"].dtype))
        dh2 = ops.moe_combine(dxs, pos, None, res=dh2_router.to(self.dtype).contiguous())      # experts' share + router's share
        del dxs, dxs_q
        dx2 = ops.rmsnorm_bwd(dh2, s.pop("x2"), self.w2, s.pop("rstd2"), dres=dy2d, out=dh2)
        self._attn_half_backward(s, dx2)


def build_fused_block(block, arenas, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None, use_mfma_dw=True, tn_dx_gemm=True):
    """The fused form of a wrapped block, whichever family recognises it (None: the generic module path)."""
    fb = FusedLlamaBlock.try_build(block, arenas, input_others, amp_dtype, sdpa_ctx=sdpa_ctx, use_mfma_dw=use_mfma_dw, tn_dx_gemm=tn_dx_gemm)
    if fb is None:
        fb = FusedOPTBlock.try_build(block, arenas, input_others, amp_dtype, sdpa_ctx=sdpa_ctx, use_mfma_dw=use_mfma_dw, tn_dx_gemm=tn_dx_gemm)
    if fb is None:
        fb = FusedMoEBlock.try_build(block, arenas, input_others, amp_dtype, sdpa_ctx=sdpa_ctx, use_mfma_dw=use_mfma_dw, tn_dx_gemm=tn_dx_gemm)
    return fb


def build_fused_block_plain(block, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None):
    for cls in (FusedLlamaBlock, FusedOPTBlock):
        fb = cls.try_build_plain(block, input_others, amp_dtype, sdpa_ctx=sdpa_ctx)
        if fb is not None:
            return fb
    return None


def mfma_dw_pays(M: int, N: int, K: int) -> bool:
    """Where the hand-written weight-gradient GEMM beats hipBLASLt on MI355X (tools/gemm_dw_probe.py, profiles/archive/r02_gemm_dw_*):
    256x256 tiles that fill the 256 CUs at least once, deep K."""
    # (few tiles x deep K go through the kernel's deterministic split-K form: OPT-125M's 768x768 / 3072x768 weights, k/v projections;
    #  a K that is not a multiple of 128 -- an expert's share of the tokens -- is completed with zero rows inside the kernel)
    return M % 256 == 0 and N % 256 == 0 and K >= 2048


class _FusedBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, token, fb, others):
        ctx.fb = fb
        return fb._forward_impl(x, others, ctx)

    @staticmethod
    def backward(ctx, dy):
        ctx.fb._backward_impl(ctx, dy)
        return None, None, None, None
