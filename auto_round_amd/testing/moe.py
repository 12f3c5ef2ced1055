"""A Mixtral-style sparse-MoE decoder block with one nn.Linear per expert projection (w1 = gate, w3 = up, w2 = down),
i.e. the module layout the reference tunes after its `modeling/fused_moe` pass has unfused Hugging Face's 3-D expert
parameters (SURVEY 2.1 `modeling/`).  Attention / norms / rotary embedding are the stock Llama modules (Mixtral's are
the same computation); the router stays unquantised (bits = 16), as in the reference."""
from __future__ import annotations

import torch
import torch.nn.functional as F


class ExpertMLP(torch.nn.Module):
    def __init__(self, hidden, ffn):
        super().__init__()
        self.w1 = torch.nn.Linear(hidden, ffn, bias=False)
        self.w2 = torch.nn.Linear(ffn, hidden, bias=False)
        self.w3 = torch.nn.Linear(hidden, ffn, bias=False)

    def forward(self, x):
        return self.w2(F.silu(self.w1(x)) * self.w3(x))


class SparseMoeBlock(torch.nn.Module):
    """top-k routing with renormalised softmax weights (MixtralSparseMoeBlock semantics)."""

    def __init__(self, hidden, ffn, num_experts=8, top_k=2):
        super().__init__()
        self.gate = torch.nn.Linear(hidden, num_experts, bias=False)
        self.experts = torch.nn.ModuleList([ExpertMLP(hidden, ffn) for _ in range(num_experts)])
        self.top_k = top_k
        self.num_experts = num_experts

    def forward(self, hidden_states):
        b, s, h = hidden_states.shape
        x = hidden_states.reshape(-1, h)
        logits = self.gate(x)
        weights = F.softmax(logits, dim=-1, dtype=torch.float32)
        weights, selected = torch.topk(weights, self.top_k, dim=-1)
        weights = (weights / weights.sum(dim=-1, keepdim=True)).to(x.dtype)
        out = torch.zeros_like(x)
        mask = F.one_hot(selected, num_classes=self.num_experts).permute(2, 1, 0)   # [E, k, tokens]
        for e in range(self.num_experts):
            kpos, tok = torch.where(mask[e])
            if tok.numel() == 0:
                continue
            y = self.experts[e](x[tok]) * weights[tok, kpos, None]
            out.index_add_(0, tok, y.to(out.dtype))
        return out.reshape(b, s, h)


def build_moe_decoder_layer(hidden=4096, ffn=14336, heads=32, kv_heads=8, num_experts=8, top_k=2, device="cuda",
                            dtype=torch.bfloat16, attn="sdpa", seed=0):
    """-> (layer, rotary_embedding, config).  A LlamaDecoderLayer whose dense MLP is replaced by a SparseMoeBlock."""
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=max(64, hidden // 8), num_attention_heads=heads,
                      num_key_value_heads=kv_heads, num_hidden_layers=1, vocab_size=32000, rope_theta=1e6,
                      max_position_embeddings=8192)
    cfg._attn_implementation = attn
    with torch.device(device):
        layer = LlamaDecoderLayer(cfg, 0)
        layer.mlp = SparseMoeBlock(hidden, ffn, num_experts, top_k)
        layer = layer.to(dtype)
        rope = LlamaRotaryEmbedding(cfg)
    layer.eval()
    for p in layer.parameters():
        p.requires_grad_(False)
    return layer, rope, cfg


def set_scheme(layer, scheme: str, **overrides):
    """Attach the per-layer scheme attributes of a preset (auto_round_amd/schemes.py, same names and values as the
    reference's schemes.py:538-832) to every nn.Linear; `NAME_W` = the weight-only variant of an A4 preset (act_bits 16).
    The MoE router gate stays in 16 bit.  Returns the number of weights that will be quantised."""
    from ..schemes import apply_scheme, resolve_scheme

    weight_only = scheme.endswith("_W")
    cfg = resolve_scheme(scheme[:-2] if weight_only else scheme, **overrides)
    if weight_only:
        cfg.update(act_bits=16, act_data_type=None, act_group_size=None, act_sym=None, act_dynamic=None)
    done = apply_scheme(layer, cfg, skip=("mlp.gate",))
    return sum(m.weight.numel() for n, m in layer.named_modules() if n in done and done[n]["bits"] < 16)
