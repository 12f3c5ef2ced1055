"""Model preparation for sparse-MoE blocks: transformers >= 5 stores a layer's experts as two fused 3-D parameters
(`experts.gate_up_proj [E, 2F, H]`, `experts.down_proj [E, H, F]`; Mixtral, Qwen3-MoE, ...), which no per-layer quantiser can
see.  Like the reference does before tuning (auto_round/modeling/fused_moe/moe_experts_interface.py: "linear_loop" experts,
`prepare_model_for_moe_quantization`) the experts are unfused into numbered children holding plain `nn.Linear`s

    experts.<e>.gate_proj / up_proj / down_proj          (checkpoint keys: ...experts.<e>.gate_proj.qweight, ...)

and the experts module gets a loop-over-hit-experts forward that calls them, so wrapper_block / pack_block / ShardWriter treat
every expert projection like any other linear.  Only the standard half-split `gate_up_proj` layout is handled (gate = first F
rows, up = last F rows); interleaved layouts (GPT-OSS) and architectures with their own expert classes are left untouched."""
from __future__ import annotations

from typing import List

import torch
from torch import nn


class ExpertContainer(nn.Module):
    """One expert's three projections as direct attributes."""


def _is_fused_experts(m: nn.Module) -> bool:
    gu, dn = getattr(m, "gate_up_proj", None), getattr(m, "down_proj", None)
    return (isinstance(gu, nn.Parameter) and isinstance(dn, nn.Parameter) and gu.dim() == 3 and dn.dim() == 3
            and gu.shape[0] == dn.shape[0] and gu.shape[1] == 2 * dn.shape[2] and gu.shape[2] == dn.shape[1]
            and not hasattr(m, "gate_up_proj_bias"))


def _linear_loop_forward(self, hidden_states: torch.Tensor, top_k_index: torch.Tensor, top_k_weights: torch.Tensor):
    """Same routing arithmetic as the fused implementation, one expert at a time through its nn.Linear children (which may be
    tuning wrappers).  Experts that received no token are not called (their parameters get no gradient).

    The (token, slot) pairs are grouped by expert with ONE stable sort and one host read of the per-expert counts; each expert's
    rows keep the order `torch.where(one_hot(top_k_index).permute(2, 1, 0)[e])` would give them (slot-major, then token) -- the
    reference's order -- so every GEMM sees the same rows in the same places, without a device->host synchronisation per expert."""
    out = torch.zeros_like(hidden_states)
    T, K = top_k_index.shape
    with torch.no_grad():
        flat = top_k_index.t().reshape(-1)                                  # index = slot * T + token
        order = torch.argsort(flat, stable=True)
        counts = torch.bincount(flat, minlength=self.num_experts).tolist()  # the one synchronisation
    start = 0
    for e, cnt in enumerate(counts):
        if cnt == 0 or e >= self.num_experts:
            start += cnt
            continue
        sel = order[start:start + cnt]
        start += cnt
        pos, tok = sel // T, sel % T
        ex = getattr(self, str(e))
        x = hidden_states[tok]
        h = self.act_fn(ex.gate_proj(x)) * ex.up_proj(x)
        h = ex.down_proj(h) * top_k_weights[tok, pos, None]
        out.index_add_(0, tok, h.to(out.dtype))
    return out


@torch.no_grad()
def unfuse_moe_experts(model: nn.Module) -> List[str]:
    """Unfuse every standard fused-experts module under `model` in place.  Returns the names of the converted modules.
    reference: auto_round/modeling/fused_moe/moe_experts_interface.py ("linear_loop" experts)."""
    done = []
    for name, m in list(model.named_modules()):
        if not _is_fused_experts(m):
            continue
        gu, dn = m.gate_up_proj.data, m.down_proj.data
        E, F2, H = gu.shape
        F = F2 // 2
        for e in range(E):
            c = ExpertContainer()
            for pname, w in (("gate_proj", gu[e, :F]), ("up_proj", gu[e, F:]), ("down_proj", dn[e])):
                lin = nn.Linear(w.shape[1], w.shape[0], bias=False, device="meta")
                lin.weight = nn.Parameter(w.clone(), requires_grad=False)
                setattr(c, pname, lin)
            m.add_module(str(e), c)
        del m.gate_up_proj, m.down_proj
        if not hasattr(m, "num_experts"):
            m.num_experts = E
        m.forward = _linear_loop_forward.__get__(m, type(m))
        m._ar_unfused = True
        done.append(name)
    return done


def expert_children(experts: nn.Module):
    """The per-expert containers of an unfused experts module / a ModuleList of experts, else None."""
    if isinstance(experts, (nn.ModuleList, list, tuple)):
        return list(experts)
    kids = [c for k, c in getattr(experts, "_modules", {}).items() if k.isdigit()]
    return kids or None


def is_linear_loop_experts(experts: nn.Module) -> bool:
    """An experts module in the unfused "linear loop" form -- this package's (`unfuse_moe_experts`) or the reference's
    (auto_round/modeling/fused_moe/moe_experts_interface.py:173-289 `linear_loop_experts_forward`: numbered children, each with
    `gate_proj` / `up_proj` / `down_proj`, an `act_fn`, `num_experts`, `_apply_gate` absent or the standard act(gate) * up): out[t] = sum_k w[t, k] *
    down_e(act(gate_e(x_t)) * up_e(x_t)).  What the fused MoE block computes in one sorted-row pass."""
    if getattr(experts, "_ar_unfused", False):
        return True
    kids = expert_children(experts)
    if not kids or isinstance(experts, (nn.ModuleList, list, tuple)):
        return False
    if int(getattr(experts, "num_experts", len(kids))) != len(kids) or not callable(getattr(experts, "act_fn", None)):
        return False
    if not all(all(hasattr(c, n) for n in ("gate_proj", "up_proj", "down_proj")) for c in kids):
        return False
    gate = getattr(experts, "_apply_gate", None)
    if gate is not None:
        # transformers' experts classes carry `_apply_gate(cat(gate, up))` and the reference's loop calls it when present
        # (moe_experts_interface.py:246-250); the standard one is act_fn(gate) * up -- anything else (GPT-OSS's clamped, interleaved
        # gate ...) is another function than the fused block's SwiGLU kernel
        try:
            probe = torch.linspace(-3.0, 3.0, 64, dtype=torch.float32).view(2, 32)
            with torch.no_grad():
                if not torch.allclose(gate(probe), experts.act_fn(probe[:, :16]) * probe[:, 16:], rtol=1e-6, atol=1e-6):
                    return False
        except Exception:  # noqa: BLE001
            return False
    return True
