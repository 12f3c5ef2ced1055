"""CPU check of the ORCHESTRATION of auto_round_amd/exact_block.py (which GEMM feeds which, gradient routing through the local
autograd graphs, the residual / accumulation order) with every segment on torch's own ops -- the form `plan_against_module` starts
from.  The same seeded Llama decoder layer runs through transformers' module code under autograd and through
ExactLlamaBlock._forward_impl / _backward_impl; the block output and all seven weight gradients must agree bit for bit (the CPU
kernels are deterministic and both sides issue the same ops).  The HIP kernels themselves are compared with torch on the GPU
(tests/test_gpu_exact_block.py)."""
import types

import pytest
import torch


def _layer(hidden=64, inter=160, heads=4, kv=2, seq=16, batch=2, bias=False):
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads, num_key_value_heads=kv,
                      num_hidden_layers=1, vocab_size=32, max_position_embeddings=64, attention_bias=bias, mlp_bias=bias)
    cfg._attn_implementation = "sdpa"
    blk = LlamaDecoderLayer(cfg, 0).to(torch.bfloat16)
    for p in blk.parameters():
        torch.nn.init.normal_(p, std=0.3)
    rot = LlamaRotaryEmbedding(cfg)
    x = torch.randn(batch, seq, hidden).to(torch.bfloat16)
    pos = torch.arange(seq)[None]
    cos, sin = rot(x, pos)
    return cfg, blk, x, (cos.to(torch.bfloat16), sin.to(torch.bfloat16))


def _exact_over(blk, cfg, pe):
    from auto_round_amd.exact_block import GEMM_OPTS, KERNEL_OPTS, ExactLlamaBlock

    attn, mlp = blk.self_attn, blk.mlp
    mods = dict(q=attn.q_proj, k=attn.k_proj, v=attn.v_proj, o=attn.o_proj, g=mlp.gate_proj, u=mlp.up_proj, d=mlp.down_proj)
    layers = {n: types.SimpleNamespace(weight_q=m.weight.detach(), weight_grad=torch.zeros_like(m.weight), _dw_accum=[False], orig_layer=m)
              for n, m in mods.items()}
    eb = object.__new__(ExactLlamaBlock)
    eb.block, eb.layers, eb.attn = blk, layers, attn
    eb.hq, eb.hkv, eb.hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.hidden_size // cfg.num_attention_heads
    eb.H, eb.Fdim, eb.dtype = cfg.hidden_size, cfg.intermediate_size, torch.bfloat16
    eb.aq = dict(qkv=None, o=None, gu=None, d=None)
    eb.w1, eb.eps1 = blk.input_layernorm.weight, blk.input_layernorm.variance_epsilon
    eb.w2, eb.eps2 = blk.post_attention_layernorm.weight, blk.post_attention_layernorm.variance_epsilon
    eb.sdpa_ctx, eb.amp, eb.qk_norm = None, False, None
    eb.plan = {k: False for k in KERNEL_OPTS + GEMM_OPTS}
    eb.plan["swiglu_contract"] = True
    eb._tnx = {}
    eb._last_sk = {}
    # merged gradient buffers as the arena lays them out: [q; k; v] and [gate; up] contiguous, the per-layer gradients are views
    eb.dWqkv = torch.zeros(sum(mods[n].weight.shape[0] for n in "qkv"), cfg.hidden_size, dtype=torch.bfloat16)
    eb.dWgu = torch.zeros(2 * cfg.intermediate_size, cfg.hidden_size, dtype=torch.bfloat16)
    r0 = 0
    for n in "qkv":
        r1 = r0 + mods[n].weight.shape[0]
        layers[n].weight_grad = eb.dWqkv[r0:r1]
        r0 = r1
    layers["g"].weight_grad, layers["u"].weight_grad = eb.dWgu[:cfg.intermediate_size], eb.dWgu[cfg.intermediate_size:]
    return eb, layers, mods


@pytest.mark.parametrize("mask_kind", ["none", "additive"])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("merged_dw", [False, True])
def test_all_torch_plan_reproduces_the_module_code_bit_for_bit(mask_kind, bias, merged_dw, monkeypatch):
    cfg, blk, x, pe = _layer(bias=bias)
    B, S, _ = x.shape
    mask = None
    if mask_kind == "additive":     # what the reference's calibration flow hands over: a 0/1 additive bias in the amp dtype
        mask = torch.tril(torch.ones(S, S)).to(torch.bfloat16)[None, None]
    others = dict(position_embeddings=pe, attention_mask=mask)
    dy = torch.randn(B, S, cfg.hidden_size).to(torch.bfloat16)

    for p in blk.parameters():
        p.requires_grad_(True)
    y_ref = blk(x, attention_mask=mask, position_embeddings=pe)
    y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
    y_ref.backward(dy)

    eb, layers, mods = _exact_over(blk, cfg, pe)
    if merged_dw:       # the merged weight-gradient buffers (one [tokens, q+k+v] / [tokens, 2F] operand); the MFMA kernel itself is
        from auto_round_amd import ops      # GPU-only: here the library GEMM takes the merged operands
        monkeypatch.setattr(ops, "gemm_dw", lambda *a, **k: False)
        eb.plan.update(dw_qkv=True, dw_gu=True)
    ctx = types.SimpleNamespace(saved=None)
    with torch.no_grad():
        y = eb._forward_impl(x, others, ctx)
        eb._backward_impl(ctx, dy)
    assert torch.equal(y.view(torch.int16), y_ref.detach().view(torch.int16))
    for n, m in mods.items():
        assert torch.equal(layers[n].weight_grad.view(torch.int16), m.weight.grad.view(torch.int16)), n


def test_gradients_accumulate_over_micro_batches_like_addmm():
    cfg, blk, x, pe = _layer()
    others = dict(position_embeddings=pe, attention_mask=None)
    dy = torch.randn_like(x)
    eb, layers, mods = _exact_over(blk, cfg, pe)
    for _ in range(2):
        ctx = types.SimpleNamespace(saved=None)
        with torch.no_grad():
            eb._forward_impl(x, others, ctx)
            eb._backward_impl(ctx, dy)
    for p in blk.parameters():
        p.requires_grad_(True)
    for _ in range(2):
        out = blk(x, position_embeddings=pe)
        (out[0] if isinstance(out, tuple) else out).backward(dy)
    # autograd accumulates .grad += g (two roundings); the arena form is addmm_ (one): close, not necessarily equal
    for n, m in mods.items():
        assert torch.allclose(layers[n].weight_grad.float(), m.weight.grad.float(), rtol=2e-2, atol=1e-2), n


def test_the_no_grad_form_is_the_same_forward():
    """`forward_nograd` (targets, quantised-output forward): `_forward_impl` without a context -- no autograd leaves, same bits."""
    cfg, blk, x, pe = _layer()
    others = dict(position_embeddings=pe, attention_mask=None)
    with torch.no_grad():
        y_ref = blk(x, position_embeddings=pe)
        y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
        eb, _, _ = _exact_over(blk, cfg, pe)
        y = eb._forward_impl(x, others, None)
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)) and not y.requires_grad


def test_streamk_plan_value_routes_the_weight_gradient_through_the_found_structure_or_the_library(monkeypatch):
    """dw_* = STREAMK: `_dw_x` asks streamk.find_on_device for the shape's structure and hands its cut table to ops.gemm_dw_sk; with no
    structure the library GEMM runs (and `_streamk_found` says so, which is what keeps the plan search from recording an unproven
    option).  Host logic only: both device calls are stubbed."""
    from auto_round_amd import exact_block, ops, streamk
    cfg, blk, x, pe = _layer()
    eb, layers, mods = _exact_over(blk, cfg, pe)
    T = x.shape[0] * x.shape[1]
    dY = torch.randn(T, layers["g"].weight_q.shape[0]).to(torch.bfloat16)
    X = torch.randn(T, layers["g"].weight_q.shape[1]).to(torch.bfloat16)
    want = torch.mm(dY.t(), X)
    calls = []

    def fake_sk(dY2d, X2d, out, kcut):
        calls.append(kcut)
        torch.mm(dY2d.t(), X2d, out=out)
        return True

    monkeypatch.setattr(ops, "gemm_dw_sk", fake_sk)
    eb.plan["dw_g"] = exact_block.STREAMK
    M, N = layers["g"].weight_q.shape
    st = streamk.Structure(4, 1, 32, 0, streamk.tile_order(1, 1, 1), torch.zeros(1, dtype=torch.int32).numpy())
    monkeypatch.setattr(streamk, "_found", {(None, M, N, T): (st, "kcut-table")})
    monkeypatch.setattr(streamk, "find_on_device", lambda a, b: streamk._found[(None, a.shape[1], b.shape[1], a.shape[0])])
    layers["g"]._dw_accum[0] = False
    eb._dw_x("g", dY, X)
    assert calls == ["kcut-table"] and torch.equal(layers["g"].weight_grad, want)
    assert eb._streamk_found("g") == [st] and eb._streamk_found("g")[0] is st
    # accumulating micro-batches never take the first-party form (the proof covered the plain product only)
    eb._dw_x("g", dY, X)
    assert len(calls) == 1
    # no structure for the shape: the library runs, and the plan search can see that nothing was proven
    monkeypatch.setattr(streamk, "_found", {(None, M, N, T): None})
    layers["g"]._dw_accum[0] = False
    layers["g"].weight_grad.zero_()
    eb._dw_x("g", dY, X)
    assert len(calls) == 1 and torch.equal(layers["g"].weight_grad, want)
    assert eb._streamk_found("g") is None
    # the merged gate | up gradient: one launch, every layer's rows in the structure of its own library GEMM
    F = layers["g"].weight_q.shape[0]
    dgu = torch.randn(T, 2 * F).to(torch.bfloat16)
    eb.dWgu = torch.empty(2 * F, N, dtype=torch.bfloat16)
    eb.plan["dw_gu"] = exact_block.STREAMK
    asked = []
    monkeypatch.setattr(streamk, "find_merged_on_device", lambda a, b, rows: asked.append(list(rows)) or "merged-table")
    monkeypatch.setattr(streamk, "_found", {(None, M, N, T): (st, "kcut-table")})
    for n in ("g", "u"):
        layers[n]._dw_accum[0] = False
    eb._dw_x("gu", dgu, X)
    assert asked == [[F, F]] and calls[-1] == "merged-table" and torch.equal(eb.dWgu, torch.mm(dgu.t(), X))
    assert eb._streamk_found("gu") == [st, st]                    # what THIS call launched: each layer's own structure at this (M, N, K)
    # a structure cached for another K is not this call's: with no table for the call's own K the library runs, and the proof is told so
    # (`_streamk_found` used to scan the cache for any K with the same (M, N))
    monkeypatch.setattr(streamk, "_found", {(None, M, N, T + 128): (st, "kcut-table")})
    monkeypatch.setattr(streamk, "find_merged_on_device", lambda a, b, rows: None)
    for n in ("g", "u"):
        layers[n]._dw_accum[0] = False
    n_calls = len(calls)
    eb._dw_x("gu", dgu, X)
    assert len(calls) == n_calls and torch.equal(eb.dWgu, torch.mm(dgu.t(), X))
    assert eb._streamk_found("gu") is None
