"""CPU check of the ORCHESTRATION of auto_round_amd/exact_opt_block.py (round 6: `exact_rounding` for OPT-style blocks, BASELINE
configs[0]) with every segment on torch's own ops -- the form `plan_against_module` starts from.  The same seeded OPTDecoderLayer runs
through transformers' module code under autograd and through ExactOPTBlock._forward_impl / _backward_impl; the block output and all six
weight gradients must agree bit for bit (CPU kernels are deterministic and both sides issue the same ops in the same shapes).  The
LayerNorm HIP kernels themselves are compared with torch on the GPU (tests/test_gpu_exact_block.py)."""
import types

import pytest
import torch


def _layer(hidden=64, ffn=160, heads=4, seq=16, batch=2):
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer

    torch.manual_seed(0)
    cfg = OPTConfig(hidden_size=hidden, ffn_dim=ffn, num_attention_heads=heads, num_hidden_layers=1, vocab_size=32,
                    max_position_embeddings=64, word_embed_proj_dim=hidden)
    cfg._attn_implementation = "sdpa"
    blk = OPTDecoderLayer(cfg).to(torch.bfloat16).eval()
    for p in blk.parameters():
        torch.nn.init.normal_(p, std=0.3)
    x = torch.randn(batch, seq, hidden).to(torch.bfloat16)
    return cfg, blk, x


def _exact_over(blk, cfg, amp):
    from auto_round_amd.exact_opt_block import ExactOPTBlock

    attn = blk.self_attn
    mods = dict(q=attn.q_proj, k=attn.k_proj, v=attn.v_proj, o=attn.out_proj, f1=blk.fc1, f2=blk.fc2)
    layers = {n: types.SimpleNamespace(weight_q=m.weight.detach(), weight_grad=torch.zeros_like(m.weight), _dw_accum=[False], orig_layer=m)
              for n, m in mods.items()}
    eb = object.__new__(ExactOPTBlock)
    eb.block, eb.layers, eb.attn = blk, layers, attn
    eb.n1, eb.n2 = blk.self_attn_layer_norm, blk.final_layer_norm
    eb.hq = eb.hkv = cfg.num_attention_heads
    eb.hd = cfg.hidden_size // cfg.num_attention_heads
    eb.H, eb.Fdim, eb.dtype = cfg.hidden_size, cfg.ffn_dim, torch.bfloat16
    eb.qscale = float(attn.scaling)
    eb.aq = dict(qkv=None, o=None, f1=None, f2=None)
    eb.sdpa_ctx, eb.amp = None, amp
    eb.plan = ExactOPTBlock.base_plan()
    return eb, layers, mods


@pytest.mark.parametrize("mask_kind", ["none", "additive"])
@pytest.mark.parametrize("amp", [False, True])
def test_all_torch_plan_reproduces_the_module_code_bit_for_bit(mask_kind, amp):
    import contextlib

    cfg, blk, x = _layer()
    B, S, H = x.shape
    mask = None
    if mask_kind == "additive":     # what the reference's calibration flow hands over: a 0/1 additive bias in the amp dtype
        mask = torch.tril(torch.ones(S, S)).to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
    others = dict(attention_mask=mask)
    dy = torch.randn(B, S, H).to(torch.bfloat16)
    for p in blk.parameters():
        p.requires_grad_(True)
    ctxm = torch.autocast("cpu", dtype=torch.bfloat16) if amp else contextlib.nullcontext()
    with ctxm:
        y_ref = blk(x, attention_mask=mask)
    y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
    y_ref.backward(dy.to(y_ref.dtype))

    eb, layers, mods = _exact_over(blk, cfg, amp)
    ctx = types.SimpleNamespace(saved=None)
    with torch.no_grad():
        y = eb._forward_impl(x, others, ctx)
        eb._backward_impl(ctx, dy)
    assert y.dtype == y_ref.dtype and torch.equal(y.view(torch.int16), y_ref.detach().view(torch.int16))
    for n, m in mods.items():
        assert torch.equal(layers[n].weight_grad.view(torch.int16), m.weight.grad.view(torch.int16)), n


def test_the_no_grad_form_is_the_same_forward():
    cfg, blk, x = _layer()
    with torch.no_grad():
        y_ref = blk(x)
        y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
        eb, _, _ = _exact_over(blk, cfg, False)
        y = eb._forward_impl(x, {"attention_mask": None}, None)
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)) and not y.requires_grad


def test_gradients_accumulate_over_micro_batches_like_addmm():
    cfg, blk, x = _layer()
    dy = torch.randn_like(x)
    eb, layers, mods = _exact_over(blk, cfg, False)
    for _ in range(2):
        ctx = types.SimpleNamespace(saved=None)
        with torch.no_grad():
            eb._forward_impl(x, {"attention_mask": None}, ctx)
            eb._backward_impl(ctx, dy)
    for p in blk.parameters():
        p.requires_grad_(True)
    for _ in range(2):
        out = blk(x)
        (out[0] if isinstance(out, tuple) else out).backward(dy)
    for n, m in mods.items():       # autograd accumulates .grad += g (two roundings); the arena form is addmm_ (one): close
        assert torch.allclose(layers[n].weight_grad.float(), m.weight.grad.float(), rtol=2e-2, atol=1e-2), n
