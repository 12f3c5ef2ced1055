"""`exact_rounding` for OPT-style decoder blocks (round 6; BASELINE configs[0] / the north-star's own model, OPT-125M): the block the
reference tunes through transformers' EAGER module code (auto_round/compressors/utils.py:109-172 `block_forward` around
transformers/models/opt/modeling_opt.py `OPTDecoderLayer` / `OPTAttention`) run as ONE autograd node with the module path's bits.

What the module path computes, op by op (eval mode: the dropouts are identities), and what runs here:

  self_attn_layer_norm / final_layer_norm   under autocast `layer_norm` is on torch's fp32 list: x.float() -> native_layer_norm in fp32
                                            -> the consuming linear casts to bf16.  Here: csrc/ar_exact_ln.hip, ATen's vectorised
                                            kernel restated (per-thread Welford over float4 vectors, shuffle-down combine over the
                                            64-lane wavefront, shared-memory combine over the 4 wavefronts; backward: two block
                                            sums in ATen's association) with the two casts folded in -- or, where that kernel does
                                            not prove bit-equal, torch's own ops in a local autograd graph.
  q_proj / k_proj / v_proj, out_proj,       `F.linear(x, Wq, bias)` exactly as `wrapper._QLinearFn` issues it, one call per layer (the
  fc1, fc2                                  library's bias epilogue included); weight gradients `torch.mm(dY^T, X)` into the arena,
                                            input gradients `torch.mm(dY, Wq)` -- the module path's own calls.
  q_proj(x) * scaling                       torch's multiply by a python scalar (its own rounding), backward the same.
  attention                                 transformers' `sdpa_attention_forward` on the layouts `OPTAttention` hands over
                                            (scaling = 1.0), in a local autograd graph: the library kernels the module path calls --
                                            or (round 6, option `attn`) csrc/ar_attn_exact.hip, the library attention's arithmetic
                                            restated from torch's gfx950 code objects: the same bits at 0.59 instead of 1.46 ms,
                                            and without the library forward's run-to-run slips at head size 64.
  residual adds, ReLU                       torch's add / relu / threshold_backward: one rounding each, as eager.

Nothing is assumed: `plan_against_module` runs real minibatches forward + backward through the module code and through this class on
the same frozen state and compares the block output and every weight gradient bit for bit -- first with every segment on torch's own
ops, then with each LayerNorm kernel switched on, kept only after two consecutive runs without a differing value.  What is gained over
the module path is launches and host work, not arithmetic: one autograd node instead of ~25, no module dispatch, no dtype-conversion
kernels around the norms (measured at OPT-125M, full recipe: 3.27 ms per iteration against the module path's 3.48; the loop is then
GPU-bound, 88 % of it the library attention and GEMMs whose bits are the reference's -- profiles/r06_opt125m_exact_kernel_stats.csv)."""
from __future__ import annotations

import contextlib
import warnings
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import ops
from .exact_block import _bits_equal, _count_diff, exact_attention_backward, exact_attention_forward
from .fused_block import OPT_FAMILY, FusedOPTBlock, _class_in, _FusedBlockFn

GEMM_OPTS = ("merged_qkv", "dw_qkv")      # q / k / v as ONE library GEMM forward (N = 3 H) and ONE weight-gradient GEMM -- kept when bit-equal
KERNEL_OPTS = ("ln1", "ln2", "attn")     # "attn": csrc/ar_attn_exact.hip instead of torch's SDPA (exact_block.py)
# flag bits of the LayerNorm kernels that an installed torch build may resolve either way (csrc/ar_exact_ln.hip): tried in this order
LN_VARIANTS = (0, 1, 2, 3)


class ExactOPTBlock(FusedOPTBlock):
    capturable = False          # the attention runs in a local autograd graph: the iteration is host-driven
    exact = True

    # -- construction -------------------------------------------------------------------------------------------------------
    @classmethod
    def try_build(cls, block, arenas, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None, amp=True, **_) -> Optional["ExactOPTBlock"]:
        if not _class_in(block, OPT_FAMILY):
            return None
        self = super().try_build(block, arenas, input_others, amp_dtype, sdpa_ctx=sdpa_ctx, use_mfma_dw=False, tn_dx_gemm=False)
        if self is None or not self._usable():
            return None
        self.amp = bool(amp)
        self.plan = self.base_plan()
        self.plan_report = None
        self._attn_verify = False
        self._attn_diffs = {}
        return self

    @classmethod
    def try_build_plain(cls, block, input_others, amp_dtype=torch.bfloat16, sdpa_ctx=None, amp=True) -> Optional["ExactOPTBlock"]:
        """The no-grad form for an UNWRAPPED block (targets, quantised-output forward): same kernels, forward-only proof."""
        import types

        if not _class_in(block, OPT_FAMILY):
            return None
        self = super().try_build_plain(block, input_others, amp_dtype, sdpa_ctx=sdpa_ctx)
        if self is None or not self._usable():
            return None
        attn = self.attn
        mods = dict(q=attn.q_proj, k=attn.k_proj, v=attn.v_proj, o=attn.out_proj, f1=block.fc1, f2=block.fc2)
        self.layers = {n: types.SimpleNamespace(weight_q=m.weight, orig_layer=m) for n, m in mods.items()}
        self.amp = bool(amp)
        self.plan = self.base_plan()
        self.plan_report = None
        self._attn_verify = False
        self._attn_diffs = {}
        return self

    def _usable(self) -> bool:
        if getattr(getattr(self.attn, "config", None), "_attn_implementation", None) != "sdpa":
            return False
        n1, n2 = self.n1, self.n2
        if any(n.bias is None or n.weight.dtype != self.dtype or n.bias.dtype != self.dtype or tuple(n.normalized_shape) != (self.H,)
               for n in (n1, n2)):
            return False
        # activation-quantised schemes: under autocast the module path fake-quantises the LayerNorm's FP32 output (each WrapperLinear
        # quantises the tensor it is handed, and the cast to the activation dtype happens inside its linear); this class hands the
        # norm's output over in the activation dtype -- other quantisation inputs, so those blocks keep the module path
        if self.aq.get("qkv") is not None or self.aq.get("f1") is not None:
            return False
        return not self.block.training

    @staticmethod
    def base_plan() -> Dict[str, object]:
        """every segment on torch's own ops (exact by construction)"""
        return dict(ln1=False, ln2=False, ln_flags=0, attn=False, attn_kb=0, merged_qkv=False, dw_qkv=False)

    def set_plan(self, plan):
        self.plan = {**self.base_plan(), **plan}

    def _refresh_tn(self):      # (the fused class's transposed dX weights are not used here)
        pass

    def _ctx(self, S):
        st = contextlib.ExitStack()
        if self.sdpa_ctx is not None:
            st.enter_context(self.sdpa_ctx(S))
        if self.amp:
            st.enter_context(torch.autocast(device_type=self.n1.weight.device.type, dtype=self.dtype))
        return st

    def _bias(self, key):
        b = self.layers[key].orig_layer.bias
        return None if b is None else b.to(self.dtype)

    def _dw_x(self, key, dY2d, X2d):
        """dW of layer `key` into the arena exactly as `_QLinearFn.backward` issues it (the library GEMM; `addmm_` when micro-batches
        accumulate)"""
        lyr = self.layers[key]
        if lyr._dw_accum[0]:
            lyr.weight_grad.addmm_(dY2d.t(), X2d)
        else:
            torch.mm(dY2d.t(), X2d, out=lyr.weight_grad)
            lyr._dw_accum[0] = True
        post = getattr(lyr, "_post_dw", None)
        if post is not None:
            post()

    def _ln_module(self, norm, x2d, B, S, grad):
        """the module code's LayerNorm on torch's ops: under autocast an fp32 `layer_norm` whose result the consuming linear casts to
        the activation dtype -> (h [T, H] in the activation dtype, (leaf, graph output) for the backward or None)"""
        with torch.enable_grad() if grad else contextlib.nullcontext():
            xl = x2d.detach().requires_grad_(grad)
            with self._ctx(S):
                hg = norm(xl.view(B, S, self.H))
            hg = hg.reshape(B * S, self.H)
            if hg.dtype != self.dtype:
                hg = hg.to(self.dtype)
        return hg.detach(), ((xl, hg) if grad else None)

    # -- forward -------------------------------------------------------------------------------------------------------------
    def _forward_impl(self, x, others, ctx):
        from transformers.integrations.sdpa_attention import sdpa_attention_forward

        from .wrapper import act_quant_fwd_raw

        P, L, aq = self.plan, self.layers, self.aq
        grad = ctx is not None
        B, S, H = x.shape
        T = B * S
        hq, hd = self.hq, self.hd
        x2d = x.reshape(T, H)
        if x2d.dtype != self.dtype:
            x2d = x2d.to(self.dtype)
        x2d = x2d.contiguous()

        def fq(t, plan):
            return t if plan is None else act_quant_fwd_raw(t, plan)

        # self_attn_layer_norm (the block input needs no gradient)
        res = ops.layernorm_fwd_exact(x2d, self.n1.weight, self.n1.bias, float(self.n1.eps), flags=int(P["ln_flags"]), want_stats=False) if P["ln1"] else None
        h1 = res[0] if res is not None else self._ln_module(self.n1, x2d, B, S, False)[0]
        h1_in = fq(h1, aq["qkv"])
        if P.get("merged_qkv"):      # one GEMM over the three projections' rows (adjacent in the arena): column slices go on
            qkv = F.linear(h1_in, self.Wqkv, self.b_qkv)
            at = {n: i * H for i, n in enumerate(self.order)}
            q2d = qkv[:, at["q"]:at["q"] + H] * self.qscale
            k2d, v2d = qkv[:, at["k"]:at["k"] + H], qkv[:, at["v"]:at["v"] + H]
            del qkv
        else:
            q2d = F.linear(h1_in, L["q"].weight_q, self._bias("q")) * self.qscale         # OPTAttention: q_proj(x) * scaling
            k2d = F.linear(h1_in, L["k"].weight_q, self._bias("k"))
            v2d = F.linear(h1_in, L["v"].weight_q, self._bias("v"))
        mask = others.get("attention_mask")
        q4, k4, v4 = (t.view(B, S, hq, hd).transpose(1, 2) for t in (q2d, k2d, v2d))
        xa = exact_attention_forward(q4, k4, v4, mask, 1.0, S, P.get("attn_kb", 0)) if P.get("attn") else None
        al = ao = None
        if xa is None and mask is not None and getattr(self, "materialise_mask_rows", False) and mask.shape[0] == 1 and B > 1:
            mask = mask.expand(B, *mask.shape[1:]).contiguous()      # (the quantizer handed the shared mask over un-materialised)
        if xa is None or getattr(self, "_attn_verify", False):
            with torch.enable_grad() if grad else contextlib.nullcontext():
                al = [t.detach().requires_grad_(grad) for t in (q4, k4, v4)]
                with self._ctx(S):
                    ao, _ = sdpa_attention_forward(self.attn, al[0], al[1], al[2], mask, dropout=0.0, scaling=1.0)
                    ao = ao.reshape(B, S, -1).contiguous()
        del q2d, k2d, v2d
        if xa is not None:
            if ao is not None:          # the proof: torch's attention output beside the first-party one
                self._attn_diffs["out"] = self._attn_diffs.get("out", 0) + _count_diff(xa[3].view(B, S, -1), ao.detach())
            a2d = xa[3].view(T, H)
        else:
            a2d = ao.detach().view(T, H)
        a_in = fq(a2d, aq["o"])
        x2 = x2d + F.linear(a_in, L["o"].weight_q, self._bias("o"))                   # residual + out_proj(attn)
        # final_layer_norm
        norm_graph, mean2, rstd2 = None, None, None
        res = ops.layernorm_fwd_exact(x2, self.n2.weight, self.n2.bias, float(self.n2.eps), flags=int(P["ln_flags"]), want_stats=grad) if P["ln2"] else None
        if res is not None:
            h2, mean2, rstd2 = res
        else:
            h2, norm_graph = self._ln_module(self.n2, x2, B, S, grad)
        h2_in = fq(h2, aq["f1"])
        a = torch.relu_(F.linear(h2_in, L["f1"].weight_q, self._bias("f1")))         # nn.ReLU()(fc1(.)): same values, in place
        f_in = fq(a, aq["f2"])
        y = x2 + F.linear(f_in, L["f2"].weight_q, self._bias("f2"))
        if grad:
            ctx.saved = dict(B=B, S=S, h1_in=h1_in, h1=h1, leaves=al, ao=ao, attn_x=xa, a2d=a2d, a_in=a_in, x2=x2, mean2=mean2, rstd2=rstd2,
                             norm_graph=norm_graph, h2=h2, h2_in=h2_in, a=a, f_in=f_in)
        return y.view(B, S, H)

    # -- backward ------------------------------------------------------------------------------------------------------------
    def _backward_impl(self, ctx, dy):
        from .wrapper import act_quant_bwd_raw

        s = ctx.saved
        ctx.saved = None
        P, L, aq = self.plan, self.layers, self.aq
        B, S = s["B"], s["S"]
        T, H = B * S, self.H

        def bq(g, x, plan):
            return g if plan is None else act_quant_bwd_raw(g, x, plan)

        dy2d = dy.reshape(T, H)
        if dy2d.dtype != self.dtype:
            dy2d = dy2d.to(self.dtype)
        dy2d = dy2d.contiguous()
        # y = x2 + fc2(relu(fc1(ln2(x2))))
        self._dw_x("f2", dy2d, s.pop("f_in"))
        a = s.pop("a")
        da = bq(torch.mm(dy2d, L["f2"].weight_q), a, aq["f2"])
        df = torch.ops.aten.threshold_backward(da, a, 0)         # ReluBackward0
        del da, a
        self._dw_x("f1", df, s.pop("h2_in"))
        dh2 = bq(torch.mm(df, L["f1"].weight_q), s.pop("h2"), aq["f1"])
        del df
        norm_graph = s.pop("norm_graph")
        x2 = s.pop("x2")
        if norm_graph is None:
            dx2 = ops.layernorm_bwd_exact(dh2, x2, self.n2.weight, s.pop("mean2"), s.pop("rstd2"), dres=dy2d, flags=int(P["ln_flags"]), out=dh2)
        else:
            xl, hg = norm_graph
            (gx,) = torch.autograd.grad(hg, xl, dh2)
            dx2 = gx + dy2d            # the two uses of x2 (norm input, residual): autograd's accumulation, one rounding
        del dh2, norm_graph, x2
        # x2 = x + out_proj(attn)
        self._dw_x("o", dx2, s.pop("a_in"))
        dattn = bq(torch.mm(dx2, L["o"].weight_q), s.pop("a2d"), aq["o"])
        del dx2
        al, ao, xa = s.pop("leaves"), s.pop("ao"), s.pop("attn_x")
        dqkv = torch.empty((T, 3 * H), dtype=self.dtype, device=dy2d.device) if P.get("dw_qkv") else None
        outs = None
        if dqkv is not None:
            at = {n: i * H for i, n in enumerate(self.order)}
            outs = tuple(dqkv[:, at[n]:at[n] + H] for n in "qkv")
        if xa is not None:
            gq4, gk4, gv4 = exact_attention_backward(xa, dattn.view(B, S, self.hq, self.hd), 1.0, outs=outs)
            if ao is not None:          # the proof: torch's gradients beside the first-party ones
                for name, mine, ref in zip(("dq", "dk", "dv"), (gq4, gk4, gv4), torch.autograd.grad(ao, al, dattn.view(B, S, H))):
                    self._attn_diffs[name] = self._attn_diffs.get(name, 0) + _count_diff(mine, ref)
        else:
            gq4, gk4, gv4 = torch.autograd.grad(ao, al, dattn.view(B, S, H))
        first_party = xa is not None
        del al, ao, xa, dattn
        h1_in = s.pop("h1_in")
        if dqkv is not None:          # the merged gradient buffer: one weight-gradient GEMM over the three projections' rows
            if not first_party:
                for n, g in zip("qkv", (gq4, gk4, gv4)):
                    outs["qkv".index(n)].view(B, S, self.hq, self.hd).copy_(g.transpose(1, 2))
            outs[0].mul_(self.qscale)                                  # MulBackward0 of q_proj(x) * scaling (its own rounding)
            acc = [l._dw_accum[0] for l in self.trio]
            if all(acc):
                self.dWqkv.addmm_(dqkv.t(), h1_in)
            elif not any(acc):
                torch.mm(dqkv.t(), h1_in, out=self.dWqkv)
            else:
                raise RuntimeError("q / k / v weight gradients in different accumulation states")
            for lyr in self.trio:
                lyr._dw_accum[0] = True
                post = getattr(lyr, "_post_dw", None)
                if post is not None:
                    post()
            return
        dq2d = gq4.transpose(1, 2).reshape(T, H) * self.qscale        # MulBackward0 of q_proj(x) * scaling
        dk2d = gk4.transpose(1, 2).reshape(T, H)
        dv2d = gv4.transpose(1, 2).reshape(T, H)
        self._dw_x("q", dq2d, h1_in)
        self._dw_x("k", dk2d, h1_in)
        self._dw_x("v", dv2d, h1_in)

    # -- the proof -------------------------------------------------------------------------------------------------------------
    def _run_once(self, x, others, dpred):
        for a in self.arenas:
            for l in a.layers:
                l._dw_accum[0] = False
        y = _FusedBlockFn.apply(x, self.arena.token, self, others)
        y.backward(dpred)
        return y.detach(), [a.dWq.clone() for a in self.arenas]

    def _ln_stats_match(self, x, flags) -> bool:
        """the LayerNorm kernel's fp32 row statistics and output against torch's own ops on the probe minibatch (a bf16 output
        alone could hide a last-bit difference in mean / rstd)"""
        x2d = x.reshape(-1, x.shape[-1]).to(self.dtype).contiguous()
        res = ops.layernorm_fwd_exact(x2d, self.n1.weight, self.n1.bias, float(self.n1.eps), flags=flags, want_stats=True)
        if res is None:
            return False
        xf = x2d.float()
        y, mean, rstd = torch.ops.aten.native_layer_norm(xf, [self.H], self.n1.weight.float(), self.n1.bias.float(), float(self.n1.eps))
        return _bits_equal(res[1], mean.view(-1)) and _bits_equal(res[2], rstd.view(-1)) and _bits_equal(res[0], y.to(self.dtype))

    def plan_against_module(self, module_forward, x, others, ref, want=None, second=None) -> Optional[dict]:
        """Same contract as ExactLlamaBlock.plan_against_module: -> the proven plan (installed), or None when not even the all-torch
        form reproduces the module path (the caller keeps the module path).  An option joins the plan after TWO consecutive runs
        without a differing value (the second on `second`'s minibatch); one more pair is granted after a failed pair -- the library
        attention under the comparison slips about once in 4000 calls at this shape (profiles/r06_opt_loop_flake2.json)."""
        for a in self.arenas:
            if not a.wq_fresh:
                a.qdq_forward()
        reset = lambda: [l._dw_accum.__setitem__(0, False) for a in self.arenas for l in a.layers]  # noqa: E731

        def module_reference(xb, ob, rb):
            reset()
            pred = module_forward(xb, ob)
            pred_c = pred if pred.is_contiguous() else pred.contiguous()
            dpred = torch.empty_like(pred_c)
            scratch = torch.zeros(1, dtype=torch.float32, device=xb.device)
            ops.mse_loss_fwd_bwd(pred_c, rb.to(pred_c.dtype), dpred=dpred, loss_accum=scratch, accum_scale=1.0, grad_scale=1000.0)
            pred_c.backward(dpred)
            return xb, ob, dpred, pred_c.detach(), [a.dWq.clone() for a in self.arenas]

        mbs = [module_reference(x, others, ref)]
        if second is not None:
            mbs.append(module_reference(*second))
        report = dict(errors={}, tried=[], kept=[], skipped={}, dropped={}, minibatches=len(mbs))

        def mismatches(plan, mb=0) -> int:
            xb, ob, dpred, y_ref, dw_ref = mbs[mb]
            self.set_plan(plan)
            try:
                y, dws = self._run_once(xb, ob, dpred)
            except (RuntimeError, ValueError, NotImplementedError) as e:
                report["errors"][",".join(k for k, v in plan.items() if v)] = repr(e)[:200]
                return -1
            return _count_diff(y, y_ref) + sum(_count_diff(a, b) for a, b in zip(dws, dw_ref))

        def proven(plan):
            n = mismatches(plan, 0)
            if n != 0:
                return False, n
            n = mismatches(plan, len(mbs) - 1)
            return n == 0, n

        plan = self.base_plan()
        ok, n_bad = proven(plan)
        if not ok:          # once more: the comparison itself sits on a library that slips now and then
            ok, n_bad2 = proven(plan)
            if ok:
                report.setdefault("kept_on_second_try", []).append("base")
                warnings.warn(f"exact_rounding: {type(self.block).__name__}: the all-torch form differed from the module path in its first pair "
                              f"of runs ({n_bad} values) and matched in the second pair -- the library under the comparison is not perfectly repeatable")
        if not ok:
            reset()
            self.plan_report = dict(report, usable=False, base_mismatches=n_bad)
            warnings.warn(f"exact_rounding: even with every segment on torch's own ops {type(self.block).__name__} differs from the module "
                          f"path ({n_bad} values); blocks of this kind keep the module path")
            return None
        # the LayerNorm kernels: which of the build-dependent forms (fast reciprocal, contraction, rsqrt) reproduces torch's statistics
        flags = next((f for f in LN_VARIANTS if self._ln_stats_match(x, f)), None)
        for opt in [o for o in KERNEL_OPTS if want is None or o in want]:
            if opt == "attn" and others.get("attention_mask") is None:
                report["skipped"][opt] = "no additive attention mask: the call is not the one the kernel restates"
                continue
            report["tried"].append(opt)
            if flags is None and opt != "attn":
                report["errors"][opt] = "no form of the LayerNorm kernel reproduces torch's row statistics / output on this stack"
                report["dropped"][opt] = {"stats": "differ"}
                continue
            trials = [dict(plan, **{opt: True, "ln_flags": flags})]
            if opt == "attn":       # the library picks its forward configuration by shape: the measured guess first, then the other key blocks
                guess = ops.attn_key_block_guess(self.hd, int(x.shape[1]))
                trials = [dict(plan, attn=True, attn_kb=kb) for kb in [0] + [b for b in (64, 32, 16) if b != guess]]
            worst = {}
            for attempt, trial in [(0, t) for t in trials] + [(1, trials[0])]:
                if opt in report["kept"]:
                    break
                if opt == "attn":               # torch's attention runs beside the kernels: outputs and gradients compared directly
                    self._attn_verify, self._attn_diffs = True, {}
                ok, n_bad = proven(trial)
                if opt == "attn":
                    self._attn_verify = False
                    direct = dict(self._attn_diffs)
                    report.setdefault("attn_direct", []).append(dict(direct, attn_kb=trial["attn_kb"]))
                    if set(direct) != {"out", "dq", "dk", "dv"} or any(direct.values()):
                        ok, n_bad = False, (n_bad if n_bad else sum(direct.values()) or -1)
                if ok:
                    plan = trial
                    report["kept"].append(opt)
                    if attempt:
                        report.setdefault("kept_on_second_try", []).append(opt)
                        warnings.warn(f"exact_rounding: option {opt} differed from the module path in its first pair of runs ({worst}) and "
                                      f"matched in the second pair -- kept; the library under the comparison is not perfectly repeatable")
                    break
                worst[f"{attempt}:{trial.get('attn_kb', 0)}" if opt == "attn" else str(attempt)] = n_bad
            if opt not in report["kept"]:
                report["dropped"][opt] = worst
        for opt in [o for o in GEMM_OPTS if want is None or o in want]:
            report["tried"].append(opt)
            trial = dict(plan, **{opt: True})
            worst = {}
            for attempt in (0, 1):
                ok, n_bad = proven(trial)
                if ok:
                    plan = trial
                    report["kept"].append(opt)
                    if attempt:
                        report.setdefault("kept_on_second_try", []).append(opt)
                    break
                worst[str(attempt)] = n_bad
            if opt not in report["kept"]:
                report["dropped"][opt] = worst
        if report["dropped"]:
            warnings.warn(f"exact_rounding: {type(self.block).__name__}: not bit-equal to the module path on this stack and left on torch's own "
                          f"ops (slower): {report['dropped']} (differing values per tried pair)")
        self.set_plan(plan)
        reset()
        self.plan_report = dict(report, usable=True, plan={k: (int(v) if k in ("ln_flags", "attn_kb") else bool(v)) for k, v in plan.items()})
        return plan

    def plan_forward_against_module(self, module_forward, x, others) -> Optional[dict]:
        """Forward-only proof for the no-grad form.  -> the plan (installed), or None (module path)."""
        with torch.no_grad():
            y_ref = module_forward(x, others).detach()

            def same(plan):
                self.set_plan(plan)
                try:
                    return _bits_equal(self._forward_impl(x, others, None).detach(), y_ref)
                except (RuntimeError, ValueError, NotImplementedError):
                    return False

            plan = self.base_plan()
            if not same(plan) and not same(plan):
                self.plan_report = dict(usable=False)
                return None
            flags = next((f for f in LN_VARIANTS if self._ln_stats_match(x, f)), None)
            kept = []
            for opt in KERNEL_OPTS:
                if opt == "attn":
                    if others.get("attention_mask") is None:
                        continue
                    ok = False
                    guess = ops.attn_key_block_guess(self.hd, int(x.shape[1]))
                    for kb in [0] + [b for b in (64, 32, 16) if b != guess]:
                        trial = dict(plan, attn=True, attn_kb=kb)
                        self._attn_verify, self._attn_diffs = True, {}
                        ok = same(trial) and same(trial) and self._attn_diffs.get("out", -1) == 0
                        self._attn_verify = False
                        if ok:
                            break
                elif flags is not None:
                    trial = dict(plan, **{opt: True, "ln_flags": flags})
                    ok = same(trial) and same(trial)
                else:
                    continue
                if ok:
                    plan = trial
                    kept.append(opt)
            self.set_plan(plan)
            self.plan_report = dict(usable=True, kept=kept, plan={k: (int(v) if k in ("ln_flags", "attn_kb") else bool(v)) for k, v in plan.items()})
        return plan
