"""SignSGD mirror (reference: auto_round/algorithms/quantization/sign_round/sign_sgd.py:255-389).

`param.add_(sign(grad), alpha=-lr)` with per-group learning rates held as 0-dim fp32 tensors, so the stock
`torch.optim.lr_scheduler.LinearLR` drives it exactly as it drives the reference optimizer (chained fp32 recurrence,
SURVEY App. A.4).  Two execution modes:

* fused (default inside `SignRoundQuantizer.quantize_block`): `step()` launches ONE `ar_qdq_int_bwd_sgd` per block
  arena -- the qdq backward, the sign step on V / min_scale / max_scale, the best-parameter snapshot and the next
  iteration's fake-quant forward -- and never materialises a gradient tensor;
* unfused (`fused=False`, or plain tensors with `.grad` set): one `ar_sign_sgd` launch per parameter, the literal
  restatement used by the step-level parity tests.

momentum (0 by default) takes the unfused route -- gradients materialised by `ar_qdq_int_bwd`, running buffers, `ar_sign_sgd` --
because the sign is then taken of the buffer; weight_decay is always 0 in the reference's quantizer and nesterov / dampening are
never set by it: they raise if requested.
"""
from __future__ import annotations

import torch
from torch.optim.optimizer import Optimizer

from . import ops


class SignSGD(Optimizer):
    """reference: SignSGD (algorithms/quantization/sign_round/sign_sgd.py:128; `step` :255-306) with `_single_tensor_sgd` (:356-389):
    `param.add_(sign(grad), alpha=-lr)` per parameter group; here one fused backward + update launch per arena."""

    def __init__(self, params, lr=None, momentum=0, dampening=0, weight_decay=0, nesterov=False, *, maximize=False,
                 foreach=None, differentiable=False, arenas=None, fused=None):
        if lr is None:
            raise ValueError("lr is required")
        if weight_decay != 0 or nesterov or maximize or dampening != 0:
            raise NotImplementedError("SignSGD on MI355X implements what the reference's quantizer can ask for: weight_decay=0 "
                                      "(sign_round/quantizer.py:422), dampening=0, nesterov=False, maximize=False")
        momentum = float(momentum or 0.0)
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=0, nesterov=False, maximize=False,
                        foreach=foreach, differentiable=differentiable)
        self.momentum = momentum
        super().__init__(params, defaults)
        self.arenas = list(arenas) if arenas else []
        self.fused = bool(self.arenas) if fused is None else fused
        self._lr_dev = {}
        self.snapshot_flag = None     # device int32[1]; set by the quantizer when best-param tracking is on
        self.fuse_next_fwd = True
        self.lr_override = None       # {id(arena): (lr_v [1], lr_mm [1])}: device scalars filled by ar_iter_begin (captured iterations)

    def _lr_tensor(self, key, value, device):
        t = self._lr_dev.get(key)
        if t is None:
            t = torch.empty(1, dtype=torch.float32, device=device)
            self._lr_dev[key] = t
        t.fill_(float(value))     # value travels as a kernel argument: no pinned-buffer reuse hazard
        return t

    def _arena_lrs(self, arena):
        """(lr_v, lr_mm) of one arena.  The quantizer tags each param group with `arena` (index) and `kind`
        ("round" | "minmax"), one pair per arena, because the auto learning rate depends on the layer bit-width
        (reference: per-layer lr groups, sign_round/quantizer.py:374-417)."""
        idx = self.arenas.index(arena)
        lr_v = lr_mm = None
        for g in self.param_groups:
            if g.get("arena", 0) != idx:
                continue
            if g.get("kind") == "minmax":
                lr_mm = g["lr"]
            else:
                lr_v = g["lr"]
        if lr_v is None:
            lr_v = self.defaults["lr"]
        if lr_mm is None:
            lr_mm = lr_v
        return lr_v, lr_mm

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self.fused:
            for a in self.arenas:
                if self.lr_override is not None:
                    lr_v_dev, lr_mm_dev = self.lr_override[id(a)]
                else:
                    lr_v, lr_mm = self._arena_lrs(a)
                    lr_v_dev, lr_mm_dev = self._lr_tensor(("v", id(a)), lr_v, a.device), self._lr_tensor(("mm", id(a)), lr_mm, a.device)
                a.backward_step(lr_v_dev, lr_mm_dev, snapshot_flag=self.snapshot_flag, fuse_next_fwd=self.fuse_next_fwd,
                                momentum=self.momentum)
            return loss
        if self.momentum:
            raise NotImplementedError("momentum is implemented on the arena path (quantize_block)")
        for gi, group in enumerate(self.param_groups):
            lr = group["lr"]
            for pi, p in enumerate(group["params"]):
                if p.grad is None:
                    continue
                lr_dev = self._lr_tensor((gi, p.device.index), lr, p.device)
                ops.sign_sgd_(p.data, p.grad.contiguous(), lr_dev)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        if self.fused:
            return      # gradients never exist as tensors; dWq is overwritten by the next backward
        super().zero_grad(set_to_none=set_to_none)
