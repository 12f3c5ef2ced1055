"""Optional attention plumbing for Hugging Face blocks on MI355X.

On ROCm 7.2 / torch 2.10 the AOTriton "efficient" SDPA kernels run the causal bf16 forward+backward of an 8x32x2048x128
problem in 2.7 ms, the "flash" ones in 5.0 ms (tools/sdpa_probe*.py).  The efficient kernels do not accept
`enable_gqa=True`, which is what transformers' stock "sdpa" path passes for grouped-query models, so torch silently
falls back to the flash kernels.  `register_mi355x_sdpa()` registers an attention function with transformers'
AttentionInterface that repeats K/V heads explicitly and calls SDPA under an efficient-first priority; select it with
`config._attn_implementation = "mi355x_sdpa"`.  Pure host-side plumbing: no arithmetic of the tuning path changes.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NAME = "mi355x_sdpa"


def efficient_backward_ok(seq: int) -> bool:
    """torch 2.10 + ROCm 7.2: the backward of the "efficient" SDPA kernels (aiter fmha_bwd behind AOTriton) returns wrong
    gradients (relative error ~1, NaNs) for token-major [B, S, H, D] operands when S % 256 == 128 and S > 128 (384, 640, 896, ...);
    contiguous [B, H, S, D] operands and every other length checked are right, and so are the "flash" and "math" backends
    (measured against fp32 autograd: tools/sdpa_backward_check.py, profiles/archive/r02_sdpa_backward_check.json).  Callers route those
    lengths to the flash kernels."""
    return seq <= 128 or seq % 256 != 128


def backend_order(prefer: str, seq=None):
    from torch.nn.attention import SDPBackend

    if prefer == "math":
        return [SDPBackend.MATH]
    if prefer == "flash" or (seq is not None and not efficient_backward_ok(int(seq))):
        return [SDPBackend.FLASH_ATTENTION, SDPBackend.MATH] if prefer != "flash" else [SDPBackend.FLASH_ATTENTION,
                                                                                       SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH]
    return [SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH]


def _repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    if n_rep == 1:
        return x
    b, h, s, d = x.shape
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def mi355x_sdpa_attention(module, query, key, value, attention_mask=None, dropout=0.0, scaling=None, is_causal=None,
                          **kwargs):
    from torch.nn.attention import SDPBackend, sdpa_kernel

    n_rep = getattr(module, "num_key_value_groups", 1)
    key, value = _repeat_kv(key, n_rep), _repeat_kv(value, n_rep)
    if attention_mask is not None and attention_mask.ndim == 4:
        attention_mask = attention_mask[:, :, :, : key.shape[-2]]
    if is_causal is None:
        is_causal = query.shape[2] > 1 and attention_mask is None and getattr(module, "is_causal", True)
    order = backend_order("efficient", query.shape[2])
    with sdpa_kernel(order, set_priority=True):
        out = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=dropout,
                                             scale=scaling, is_causal=bool(is_causal))
    return out.transpose(1, 2).contiguous(), None


def register_mi355x_sdpa() -> str:
    from transformers import AttentionInterface

    AttentionInterface.register(NAME, mi355x_sdpa_attention)
    return NAME
