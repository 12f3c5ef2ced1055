"""Host logic of auto_round_amd/attention.py and of the activation-quant plans (no GPU): which SDPA backend order a sequence length
gets -- torch 2.10 / ROCm 7.2's efficient backward is wrong for token-major operands at S % 256 == 128 (profiles/archive/r02_sdpa_backward_check.json)
-- and which fake-quant a layer's activation attributes select (reference: WrapperLinear._qdq_act, auto_round/wrapper.py:295-321)."""
import types

import pytest
import torch

from auto_round_amd.attention import backend_order, efficient_backward_ok


def test_efficient_backward_is_avoided_exactly_where_it_is_wrong():
    ok = [s for s in range(64, 4097, 64) if efficient_backward_ok(s)]
    bad = [s for s in range(64, 4097, 64) if not efficient_backward_ok(s)]
    assert 128 in ok and 256 in ok and 512 in ok and 2048 in ok and 64 in ok
    assert bad[:4] == [384, 640, 896, 1152] and all(s % 256 == 128 for s in bad)


def test_backend_order_by_preference_and_length():
    from torch.nn.attention import SDPBackend as B

    assert backend_order("efficient", 2048)[0] == B.EFFICIENT_ATTENTION
    assert backend_order("efficient", None)[0] == B.EFFICIENT_ATTENTION
    assert backend_order("efficient", 384) == [B.FLASH_ATTENTION, B.MATH]           # the efficient kernels are left out entirely
    assert backend_order("flash", 2048)[0] == B.FLASH_ATTENTION
    assert backend_order("math", 384) == [B.MATH]


def _layer(**kw):
    base = dict(act_bits=16, act_data_type="int", act_group_size=None, act_sym=True, act_dynamic=True, scale_dtype=torch.float16)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_activation_quant_plans_follow_the_layer_attributes():
    from auto_round_amd.wrapper import act_quant_plan

    assert act_quant_plan(_layer(), 4096) is None
    assert act_quant_plan(_layer(act_bits=8, act_group_size=32), 4096) == ("int", 8, 32, torch.float16, 1e-5, True)
    assert act_quant_plan(_layer(act_bits=8, act_group_size=-1), 4096)[2] == 4096                # per token
    assert act_quant_plan(_layer(act_bits=8, act_group_size=128), 64)[2] == 64                   # hidden < group: one group per row
    assert act_quant_plan(_layer(act_bits=4, act_group_size=128, act_sym=False, scale_dtype=torch.float32), 256) == (
        "int", 4, 128, torch.float32, 1e-8, False)
    assert act_quant_plan(_layer(act_bits=4, act_data_type="mx_fp", act_group_size=32), 4096) == ("mx",)
    assert act_quant_plan(_layer(act_bits=4, act_data_type="nv_fp4_with_static_gs", act_group_size=16), 4096) == ("nv",)
    with pytest.raises(NotImplementedError):
        act_quant_plan(_layer(act_bits=8, act_group_size=48), 4096 + 8)                           # groups must divide the row
    with pytest.raises(NotImplementedError):
        act_quant_plan(_layer(act_bits=8, act_dynamic=False, act_group_size=32), 4096)            # static int activations: not built


def test_the_sdpa_context_managers_leave_torch_as_they_found_it_and_pass_cpu_calls_through():
    """attention.reproducible_sdpa_forward / verified_sdpa_forward / guarded_sdpa patch `torch.nn.functional.scaled_dot_product_attention`
    for the length of a `with` block (the module code looks the function up at call time); CPU tensors -- and every call the patches are
    not about -- go straight to torch's function, and the original is back afterwards, also after an exception, also when nested."""
    import torch
    import torch.nn.functional as F

    from auto_round_amd import attention as at

    real = F.scaled_dot_product_attention
    q = torch.randn(1, 2, 8, 16)
    want = real(q, q, q)
    flag = torch.zeros(1, dtype=torch.bool)
    with at.reproducible_sdpa_forward():
        assert F.scaled_dot_product_attention is not real
        with at.reproducible_sdpa_forward():                     # nested: the inner one is a no-op
            with torch.no_grad():
                assert torch.equal(F.scaled_dot_product_attention(q, q, q), want)
        assert F.scaled_dot_product_attention is not real
    assert F.scaled_dot_product_attention is real
    with at.reproducible_sdpa_forward(False):
        assert F.scaled_dot_product_attention is real
    for ctx in (lambda: at.verified_sdpa_forward(flag), lambda: at.guarded_sdpa("before,after,touch")):
        try:
            with ctx():
                assert torch.equal(F.scaled_dot_product_attention(q, q, q), want)
                raise KeyError("boom")
        except KeyError:
            pass
        assert F.scaled_dot_product_attention is real and not bool(flag)
    with at.guarded_sdpa(""):
        assert F.scaled_dot_product_attention is real
