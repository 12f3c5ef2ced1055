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
    """The reference's `linear_loop_experts_forward` (auto_round/modeling/fused_moe/moe_experts_interface.py:173-289) op for op -- the
    same tensors in the same order, so that a block unfused by THIS package computes, bit for bit, what the reference's unfused block
    computes (round 5: the reference-free flow's targets and module-path trajectory at Mixtral-8x7B's real width are the reference's):

        selected = hidden_states[token of every (token, slot) pair]         pairs in TOKEN-major order, S = tokens * top_k rows
        per expert e, rows sample_idx = the pairs routed to e, ascending:   down_e(act(gate_e(x)) * up_e(x))  -> index_copy_ into [S, H]
        out = (per-pair outputs * routing weights).view(tokens, top_k, H).sum(dim=1)

    The row ORDER inside an expert's GEMMs matters: for ragged row counts the library picks kernels that split K per tile, so a row's
    bits depend on the tile it lands in (rounds 3-4 grouped the pairs slot-major: same function, other bits).  One difference in how the
    row sets are found: one stable sort + ONE host read of the per-expert counts instead of a `nonzero` (a device->host synchronisation)
    per expert -- `order[start:start + cnt]` IS `nonzero(expert_ids == e)`.  Experts that received no token are not called (their
    parameters get no gradient)."""
    shape3 = None
    if hidden_states.dim() == 3:
        shape3 = hidden_states.shape
        hidden_states = hidden_states.view(-1, shape3[-1])
        top_k_index = top_k_index.view(-1, top_k_index.size(-1))
        top_k_weights = top_k_weights.view(-1, top_k_weights.size(-1))
    T, K = top_k_index.shape
    H = hidden_states.size(-1)
    dev = hidden_states.device
    token_idx = torch.arange(T, device=dev).unsqueeze(1).expand(-1, K).reshape(-1)
    sample_weights = top_k_weights.reshape(-1).to(hidden_states.dtype)
    expert_ids = top_k_index.reshape(-1)
    selected = hidden_states[token_idx]
    out_per_sample = torch.zeros_like(selected)
    with torch.no_grad():
        order = torch.argsort(expert_ids, stable=True)
        counts = torch.bincount(expert_ids, minlength=self.num_experts).tolist()      # the one synchronisation
    start = 0
    for e, cnt in enumerate(counts):
        if cnt == 0 or e >= self.num_experts:
            start += cnt
            continue
        sample_idx = order[start:start + cnt]
        start += cnt
        ex = getattr(self, str(e))
        x = selected.index_select(0, sample_idx)
        gate_out, up_out = ex.gate_proj(x), ex.up_proj(x)
        if hasattr(self, "_apply_gate"):
            gated = self._apply_gate(torch.cat([gate_out, up_out], dim=-1))
        else:
            gated = self.act_fn(gate_out) * up_out
        y = ex.down_proj(gated)
        out_per_sample.index_copy_(0, sample_idx, y.to(out_per_sample.dtype))
    out_per_sample = out_per_sample * sample_weights.unsqueeze(-1)
    out = out_per_sample.view(T, K, H).sum(dim=1)
    return out.view(shape3) if shape3 is not None else out


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
