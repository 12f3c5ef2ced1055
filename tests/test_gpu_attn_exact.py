"""ar_attn_fwd_exact / ar_attn_bwd_exact (csrc/ar_attn_exact.hip) against torch's own attention on the GPU: the output, the log-sum-exp
rows and the q / k / v gradients of `F.scaled_dot_product_attention(q, k, v, attn_mask=<0 / 1 additive mask>)` -- the call
transformers' `sdpa_attention_forward` makes under the reference's `block_forward` (auto_round/compressors/utils.py:109-172,
calibration/llm.py:360-402, inputs.py:100-107) -- value for value, at the two minibatch shapes of the bit-identical paths (the library
picks its kernel configuration by shape: other shapes are not claimed, callers prove each call signature)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [pytest.param(8, 32, 2048, 128, 8, 128 ** -0.5, 1.0, id="llama3-8b-minibatch"),
          pytest.param(8, 12, 2048, 64, 12, 1.0, 0.35, id="opt-125m-minibatch"),
          pytest.param(8, 64, 2048, 128, 8, 128 ** -0.5, 1.0, id="llama3-70b-minibatch"),
          pytest.param(4, 32, 2048, 128, 8, 128 ** -0.5, 1.0, id="llama3-8b-batch4"),
          pytest.param(8, 32, 1024, 128, 8, 128 ** -0.5, 1.0, id="llama3-8b-seq1024"),
          pytest.param(8, 32, 4096, 128, 8, 128 ** -0.5, 1.0, id="llama3-8b-seq4096"),
          # other sequence lengths: the library's forward uses another key block there (ops.attn_key_block_guess)
          pytest.param(4, 32, 512, 128, 8, 128 ** -0.5, 1.0, id="llama3-8b-seq512-keyblock32"),
          pytest.param(4, 12, 512, 64, 12, 1.0, 0.35, id="opt-125m-seq512-keyblock64"),
          pytest.param(2, 12, 4096, 64, 12, 1.0, 0.35, id="opt-125m-seq4096-keyblock64"),
          pytest.param(4, 16, 768, 64, 16, 1.0, 0.35, id="head64-seq768-keyblock32"),
          # S % 256 != 0: workgroups of 4 waves in the backward
          pytest.param(4, 32, 384, 128, 8, 128 ** -0.5, 1.0, id="llama3-8b-seq384"),
          pytest.param(4, 12, 640, 64, 12, 1.0, 0.35, id="opt-125m-seq640"),
          pytest.param(2, 32, 1152, 128, 8, 128 ** -0.5, 1.0, id="llama3-8b-seq1152")]


def _case(B, H, S, D, hk, std, seed=0, valid=None):
    torch.manual_seed(seed)
    dev = "cuda"
    q = (torch.randn(B, S, H, D, device=dev) * std).to(torch.bfloat16).transpose(1, 2)
    k = (torch.randn(B, S, hk, D, device=dev) * std).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(B, S, hk, D, device=dev).to(torch.bfloat16).transpose(1, 2)
    valid = S - 1 if valid is None else valid
    idx = torch.arange(S, device=dev)
    keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < valid)
    mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()       # the calibration flow's 0 / 1 additive mask
    return q, k, v, mask


def _ndiff(a, b):
    a, b = a.contiguous(), b.contiguous()
    it = {2: torch.int16, 4: torch.int32}[a.element_size()]
    return int((a.view(it) != b.view(it)).sum())


@pytest.mark.parametrize("B,H,S,D,hk,scale,std", SHAPES)
def test_forward_and_backward_have_the_librarys_bits(B, H, S, D, hk, scale, std):
    from auto_round_amd import ops

    q, k, v, mask = _case(B, H, S, D, hk, std)
    st = ops.mask_structure(mask, S)
    assert st == (1.0, 0.0, S - 1)
    rep = H // hk
    ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
    ke = kl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else kl        # transformers' repeat_kv
    ve = vl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else vl
    o = F.scaled_dot_product_attention(ql, ke, ve, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale)
    ao = o.transpose(1, 2).contiguous()
    da = (torch.randn(B, S, H, D, device=q.device) * 0.02).to(torch.bfloat16)
    gq, gk, gv = torch.autograd.grad(ao, (ql, kl, vl), da)
    with torch.no_grad():
        lse_ref = torch.ops.aten._scaled_dot_product_efficient_attention(q, ke.detach(), ve.detach(), mask.expand(B, H, S, S), True, 0.0, False,
                                                                         scale=scale)[1]
        got = ops.attn_fwd_exact(q, k, v, st, scale, key_block=ops.attn_key_block_guess(D, S))
        assert got is not None
        mo, mlse = got
        assert _ndiff(mo, ao.detach()) == 0                      # bit for bit: 67 M values at Llama-3-8B's minibatch
        assert _ndiff(mlse, lse_ref[..., :S]) == 0
        from auto_round_amd.exact_block import exact_attention_backward

        dq4, dk4, dv4 = exact_attention_backward((q, k, v, mo, mlse, st), da, scale)
    assert _ndiff(dq4, gq) == 0
    assert _ndiff(dk4, gk) == 0
    assert _ndiff(dv4, gv) == 0


def test_other_key_padding_and_a_second_seed():
    """the mask's other parameter (keys at the end marked invalid) and other operand values: still the library's bits"""
    from auto_round_amd import ops

    B, H, S, D, hk, scale = 8, 12, 2048, 64, 12, 1.0
    q, k, v, mask = _case(B, H, S, D, hk, 0.5, seed=3, valid=S - 300)
    st = ops.mask_structure(mask, S)
    assert st == (1.0, 0.0, S - 300)
    ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
    ao = F.scaled_dot_product_attention(ql, kl, vl, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale).transpose(1, 2).contiguous()
    da = (torch.randn(B, S, H, D, device=q.device) * 0.05).to(torch.bfloat16)
    gq, gk, gv = torch.autograd.grad(ao, (ql, kl, vl), da)
    with torch.no_grad():
        mo, mlse = ops.attn_fwd_exact(q, k, v, st, scale)
        from auto_round_amd.exact_block import exact_attention_backward

        dq4, dk4, dv4 = exact_attention_backward((q, k, v, mo, mlse, st), da, scale)
    assert _ndiff(mo, ao.detach()) == 0
    assert (_ndiff(dq4, gq), _ndiff(dk4, gk), _ndiff(dv4, gv)) == (0, 0, 0)


@pytest.mark.parametrize("B,H,S,hk,valid", [(8, 32, 2048, 8, 2047), (2, 32, 512, 8, 400), (2, 16, 1152, 16, 1100)])
def test_launch_forms_give_equal_bits(B, H, S, hk, valid):
    """ar_attn_exact_config: the hand-pipelined key-side kernel (bit 16; selectable, not the default) and the workgroup -> XCD mapping before
    `xattn_map` (bit 32) must return the default form's bits -- output, log-sum-exp and all three gradients."""
    from auto_round_amd import _lib, ops
    from auto_round_amd.exact_block import exact_attention_backward

    D, scale = 128, 128 ** -0.5
    q, k, v, mask = _case(B, H, S, D, hk, 1.0, seed=5, valid=valid)
    st = ops.mask_structure(mask, S)
    da = (torch.randn(B, S, H, D, device=q.device) * 0.02).to(torch.bfloat16)
    lib = _lib.load()
    outs = {}
    try:
        for cfg in (0, 16, 32, 48):
            lib.ar_attn_exact_config(cfg)
            with torch.no_grad():
                mo, mlse = ops.attn_fwd_exact(q, k, v, st, scale)
                g = exact_attention_backward((q, k, v, mo, mlse, st), da, scale)
            outs[cfg] = [mo.clone(), mlse.clone()] + [t.clone() for t in g]
    finally:
        lib.ar_attn_exact_config(0)
    for cfg in (16, 32, 48):
        assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[cfg])), cfg


def test_refusals():
    """calls the kernels do not restate are refused (the caller keeps torch's SDPA), never answered approximately"""
    from auto_round_amd import ops

    q, k, v, mask = _case(1, 2, 256, 64, 2, 1.0)
    st = ops.mask_structure(mask, 256)
    assert ops.attn_fwd_exact(q, k, v, None, 1.0) is None                                        # no structured mask
    assert ops.attn_fwd_exact(q.float(), k.float(), v.float(), st, 1.0) is None                  # not bf16
    q96 = torch.randn(1, 2, 256, 96, device="cuda").to(torch.bfloat16)
    assert ops.attn_fwd_exact(q96, q96, q96, st, 1.0) is None                                    # head size
    assert ops.attn_fwd_exact(q, k, v, (0.3, 0.0, 255), 1.0) is None                             # mask value not a bf16 number
    hard = (0.0, float(torch.finfo(torch.bfloat16).min), 255)
    assert ops.attn_fwd_exact(q, k, v, hard, 1.0) is None                                        # hard (-inf like) mask
