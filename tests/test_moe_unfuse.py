"""Unfusing of transformers' fused 3-D MoE expert parameters into per-expert nn.Linear (model preparation for cfg 5)."""
import pytest
import torch


def tiny_mixtral(layers=1, hidden=64, ffn=128, experts=4, top_k=2, seed=0):
    from transformers import MixtralConfig, MixtralForCausalLM

    torch.manual_seed(seed)
    cfg = MixtralConfig(hidden_size=hidden, intermediate_size=ffn, num_attention_heads=4, num_key_value_heads=2,
                        num_hidden_layers=layers, vocab_size=96, max_position_embeddings=64, num_local_experts=experts,
                        num_experts_per_tok=top_k, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    return MixtralForCausalLM(cfg).eval()


def test_unfused_experts_compute_the_same_function_and_expose_linears():
    from auto_round_amd.moe_unfuse import expert_children, unfuse_moe_experts
    from auto_round_amd.schemes import apply_scheme, resolve_scheme

    model = tiny_mixtral()
    tokens = torch.randint(0, 96, (2, 24), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        before = model(input_ids=tokens).logits
    gu = model.model.layers[0].mlp.experts.gate_up_proj.detach().clone()
    dn = model.model.layers[0].mlp.experts.down_proj.detach().clone()
    names = unfuse_moe_experts(model)
    assert names == ["model.layers.0.mlp.experts"]
    ex = model.model.layers[0].mlp.experts
    assert not hasattr(ex, "gate_up_proj") and len(expert_children(ex)) == 4
    assert torch.equal(ex.get_submodule("2").gate_proj.weight, gu[2, :128]) and torch.equal(ex.get_submodule("2").up_proj.weight, gu[2, 128:])
    assert torch.equal(ex.get_submodule("3").down_proj.weight, dn[3])
    with torch.no_grad():
        after = model(input_ids=tokens).logits
    assert torch.allclose(before, after, rtol=1e-4, atol=1e-5)          # fp32: one fused GEMM vs two, same math
    sd = model.state_dict()
    assert "model.layers.0.mlp.experts.1.up_proj.weight" in sd and "model.layers.0.mlp.experts.gate_up_proj" not in sd
    assert unfuse_moe_experts(model) == []                               # idempotent
    cfg = apply_scheme(model.model.layers[0], resolve_scheme("W4A16", group_size=32))
    assert cfg["mlp.experts.0.gate_proj"]["bits"] == 4 and "mlp.gate" not in cfg     # the router is not an nn.Linear: stays fp
    assert sum(c["bits"] == 4 for c in cfg.values()) == 4 + 3 * 4


def test_dense_models_are_left_alone():
    from transformers import LlamaConfig, LlamaForCausalLM

    from auto_round_amd.moe_unfuse import unfuse_moe_experts

    m = LlamaForCausalLM(LlamaConfig(hidden_size=32, intermediate_size=64, num_attention_heads=2, num_key_value_heads=2,
                                     num_hidden_layers=1, vocab_size=32))
    assert unfuse_moe_experts(m) == []


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/auto_round"), reason="reference tree not present (GPU box)")
def test_unfusing_equals_the_reference_preparation():
    """Same state-dict keys, same weights, bit-identical logits as the reference's prepare_model_for_moe_quantization
    (auto_round/modeling/fused_moe/moe_experts_interface.py) on a transformers Mixtral."""
    import copy
    import os
    import sys

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, "/root/reference"):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.modeling.fused_moe.moe_experts_interface import prepare_model_for_moe_quantization

    from auto_round_amd.moe_unfuse import unfuse_moe_experts

    base = tiny_mixtral()
    a, b = copy.deepcopy(base), copy.deepcopy(base)
    prepare_model_for_moe_quantization(a)
    unfuse_moe_experts(b)
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    tok = torch.randint(0, 96, (2, 16), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        assert torch.equal(a(input_ids=tok).logits, b(input_ids=tok).logits)


def test_expert_grouping_keeps_the_reference_row_order():
    """The loop groups (token, slot) pairs with one stable sort; every expert must see its rows in the order the reference's
    `nonzero(expert_ids == e)` over the token-major pair list yields (moe_experts_interface.py:224-233), empty experts included -- the
    GEMM row order is part of bit-for-bit parity (rounds 3-4 used the slot-major order of transformers' own one_hot loop: same
    function, other bits at ragged row counts).  Checked through the forward itself with recording experts."""
    from auto_round_amd.moe_unfuse import _linear_loop_forward

    class Rec(torch.nn.Module):
        def __init__(self, log, e):
            super().__init__()
            self.log, self.e = log, e
            self.gate_proj = self.up_proj = self.down_proj = self

        def forward(self, x):
            if x.shape[-1] == 4 and not self.log.get(("seen", self.e)):
                self.log[("seen", self.e)] = True
                self.log[self.e] = x[:, 0].clone()              # column 0 carries the token id
            return x

    g = torch.Generator().manual_seed(3)
    for trial in range(20):
        T, K, E = int(torch.randint(1, 33, (), generator=g)), 2, 6
        idx = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(T)])
        if trial % 3 == 0:
            idx = idx.clamp(max=2)                               # experts 3..5 stay empty
        log = {}
        holder = torch.nn.Module()
        holder.num_experts, holder.act_fn = E, torch.nn.Identity()
        for e in range(E):
            holder.add_module(str(e), Rec(log, e))
        hidden = torch.zeros(T, 4)
        hidden[:, 0] = torch.arange(T, dtype=torch.float32)
        _linear_loop_forward(holder, hidden, idx, torch.ones(T, K))
        # the reference's row sets (moe_experts_interface.py:224-233): pairs in token-major order, `nonzero(expert_ids == e)` ascending
        token_idx = torch.arange(T).unsqueeze(1).expand(-1, K).reshape(-1)
        expert_ids = idx.reshape(-1)
        for e in range(E):
            sample_idx = torch.nonzero(expert_ids == e, as_tuple=False).squeeze(-1)
            if sample_idx.numel() == 0:
                assert e not in log                              # not called
            else:
                assert torch.equal(log[e], token_idx[sample_idx].float()), (trial, e)


def test_unfused_forward_is_the_references_linear_loop_bit_for_bit_with_gradients():
    """Live against /root/reference (skipped where the tree is absent): this package's `_linear_loop_forward` and the reference's
    `linear_loop_experts_forward` on the same experts, routing and inputs -- outputs, input gradients and every expert weight gradient
    identical bits (same ops in the same order: on the GPU the library then sees the same GEMM operands row for row)."""
    import pytest

    from ref_tree import import_reference, reference_root

    if reference_root() is None:
        pytest.skip("reference tree not present")
    import_reference()
    from auto_round.modeling.fused_moe.moe_experts_interface import linear_loop_experts_forward

    from auto_round_amd.moe_unfuse import ExpertContainer, _linear_loop_forward

    g = torch.Generator().manual_seed(11)
    T, K, E, H, F = 37, 2, 5, 16, 24

    def build():
        torch.manual_seed(5)
        holder = torch.nn.Module()
        holder.num_experts, holder.act_fn = E, torch.nn.SiLU()
        for e in range(E):
            c = ExpertContainer()
            c.gate_proj, c.up_proj, c.down_proj = torch.nn.Linear(H, F, bias=False), torch.nn.Linear(H, F, bias=False), torch.nn.Linear(F, H, bias=False)
            holder.add_module(str(e), c.to(torch.bfloat16))
        return holder

    idx = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(T)])
    idx[idx == 3] = 1                                            # expert 3 stays empty
    w = torch.rand(T, K, generator=g).to(torch.bfloat16)
    x0 = torch.randn(T, H, generator=g).to(torch.bfloat16)
    outs = []
    for fn in (_linear_loop_forward, linear_loop_experts_forward):
        m = build()
        x = x0.clone().requires_grad_(True)
        wv = w.clone().requires_grad_(True)
        y = fn(m, x, idx, wv)
        y.backward(torch.ones_like(y) * 0.37)
        outs.append((y.detach(), x.grad, wv.grad, [p.grad for p in m.parameters()]))
    (y1, gx1, gw1, gp1), (y2, gx2, gw2, gp2) = outs
    eq = lambda a, b: (a is None and b is None) or torch.equal(a.view(torch.int16), b.view(torch.int16))  # noqa: E731
    assert eq(y1, y2) and eq(gx1, gx2) and eq(gw1, gw2)
    assert len(gp1) == len(gp2) and all(eq(a, b) for a, b in zip(gp1, gp2))


def test_the_references_linear_loop_experts_are_recognised_like_this_packages_own():
    """Behind the reference's front door a Mixtral block arrives with the REFERENCE's unfused experts (numbered containers with
    gate_proj / up_proj / down_proj, act_fn, num_experts; moe_experts_interface.py:173-289) -- the fused MoE block must take them."""
    import torch
    from torch import nn

    from auto_round_amd.moe_unfuse import is_linear_loop_experts

    class RefExperts(nn.Module):            # the structure the reference leaves behind (no `_ar_unfused` marker)
        def __init__(self, E=4, H=16, F=32):
            super().__init__()
            self.num_experts, self.act_fn = E, nn.SiLU()
            for e in range(E):
                c = nn.Module()
                c.gate_proj, c.up_proj, c.down_proj = nn.Linear(H, F, bias=False), nn.Linear(H, F, bias=False), nn.Linear(F, H, bias=False)
                self.add_module(str(e), c)

    ex = RefExperts()
    assert is_linear_loop_experts(ex)
    ex._apply_gate = lambda gu: ex.act_fn(gu[..., :gu.shape[-1] // 2]) * gu[..., gu.shape[-1] // 2:]      # transformers' standard gate
    assert is_linear_loop_experts(ex)
    ex._apply_gate = lambda x: x[..., :x.shape[-1] // 2]            # a custom gate: another function than SwiGLU
    assert not is_linear_loop_experts(ex)
    ex = RefExperts()
    del ex._modules["3"].down_proj
    assert not is_linear_loop_experts(ex)
    assert not is_linear_loop_experts(nn.ModuleList([nn.Linear(4, 4)]))
    ex = RefExperts()
    ex.num_experts = 5                      # containers missing
    assert not is_linear_loop_experts(ex)
    assert not is_linear_loop_experts(torch.nn.Linear(4, 4))


def test_live_the_references_own_unfusing_of_a_mixtral_block_is_recognised():
    """With the reference tree at hand: a tiny Mixtral through the REFERENCE's `prepare_model_for_moe_quantization`
    (moe_experts_interface.py) -- transformers' MixtralExperts keeps its standard `_apply_gate` -- is what the fused MoE block must take."""
    import pytest

    from ref_tree import import_reference, reference_root

    if reference_root() is None:
        pytest.skip("reference tree not present")
    import_reference()
    import auto_round.modeling.fused_moe.moe_experts_interface as mi

    from auto_round_amd.moe_unfuse import expert_children, is_linear_loop_experts
    from auto_round_amd.testing import t3_fixture as fx

    model = fx.build_model("mixtral_tiny")
    assert mi.prepare_model_for_moe_quantization(model)
    experts = fx.decoder_blocks(model)[0].mlp.experts
    assert hasattr(experts, "_apply_gate") and is_linear_loop_experts(experts)
    kids = expert_children(experts)
    assert len(kids) == experts.num_experts and all(hasattr(k, "gate_proj") for k in kids)
