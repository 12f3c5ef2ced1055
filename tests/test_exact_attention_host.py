"""Host-side logic of the round-6 attention path, no GPU: `attention.exact_sdpa_attention` hands every call it cannot run on the
first-party kernels (CPU tensors here) to transformers' own `sdpa_attention_forward` unchanged, counts it as a fallback, and
materialises a shared one-row mask for that call when the quantizer handed it over un-materialised; `ops.attn_fwd_exact` /
`attn_bwd_exact` refuse instead of computing anything on the CPU (reference call: transformers/integrations/sdpa_attention.py under
auto_round/compressors/utils.py:109-172)."""
import types

import pytest
import torch


def _case(B=2, H=4, S=256, D=64):
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).transpose(1, 2) for _ in range(3))
    idx = torch.arange(S)
    keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < S - 1)
    return q, k, v, keep.to(torch.bfloat16)[None, None]


def test_fallback_is_transformers_own_attention_and_is_counted():
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    from auto_round_amd import attention as A

    q, k, v, mask1 = _case()
    mod = types.SimpleNamespace(num_key_value_groups=1, is_causal=True, training=False)
    A.exact_state.update(verify=False, diffs={}, calls=0, fallbacks=0, materialise=False)
    want, _ = sdpa_attention_forward(mod, q, k, v, mask1.expand(2, 1, 256, 256).contiguous(), dropout=0.0, scaling=1.0)
    got, _ = A.exact_sdpa_attention(mod, q, k, v, mask1.expand(2, 1, 256, 256).contiguous(), dropout=0.0, scaling=1.0)
    assert torch.equal(got, want)
    assert A.exact_state["fallbacks"] == 1 and A.exact_state["calls"] == 0
    # the un-materialised shared mask: expanded for the fallback call exactly as the module path's runner would have
    A.exact_state["materialise"] = True
    got1, _ = A.exact_sdpa_attention(mod, q, k, v, mask1, dropout=0.0, scaling=1.0)
    A.exact_state["materialise"] = False
    assert torch.equal(got1, want)


def test_ops_refuse_cpu_tensors_loudly_or_return_none():
    from auto_round_amd import _lib, ops

    q, k, v, mask1 = _case()
    st = (1.0, 0.0, 255)
    with pytest.raises(_lib.Mi355xLibraryError):
        ops.attn_fwd_exact(q, k, v, st, 1.0)                     # a CPU tensor: there is no CPU fallback
    assert ops.attn_fwd_exact(q, k, v, None, 1.0) is None       # no structured mask: the caller keeps torch's SDPA
    assert ops.attn_fwd_exact(q.float(), k.float(), v.float(), st, 1.0) is None


def test_registration_name_and_restore():
    from transformers import AttentionInterface

    from auto_round_amd import attention as A
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    assert A.register_exact_sdpa() == A.EXACT_NAME
    assert A.EXACT_NAME in AttentionInterface().valid_keys()
    qz = SignRoundQuantizer.__new__(SignRoundQuantizer)          # (the constructor insists on a HIP device; this is host logic only)
    qz._attn_restore = []
    cfg_obj = types.SimpleNamespace(_attn_implementation=A.EXACT_NAME)
    qz._attn_restore.append((cfg_obj, "sdpa"))
    A.exact_state["materialise"] = True
    qz._restore_module_attention()
    assert cfg_obj._attn_implementation == "sdpa" and not qz._attn_restore and A.exact_state["materialise"] is False


def test_key_block_table_is_the_measured_one():
    """ops.attn_key_block_guess: the forward key block per (head size, sequence length) measured against torch
    (profiles/r06_attn_exact_keyblock_probe.json) -- only ever a proof's first candidate"""
    import json
    import os

    from auto_round_amd import ops

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_attn_exact_keyblock_probe.json")
    rows = json.load(open(path))
    checked = 0
    for r in rows:
        exact = [kb for kb in (16, 32, 64) if r.get(f"kb{kb}") == [0, 0]]
        if not exact:
            continue                      # head size 128 at S <= 256: no key block reproduces the library there
        assert ops.attn_key_block_guess(r["D"], r["S"]) in exact, r
        checked += 1
    assert checked >= 12


def test_front_door_calibration_mask_is_the_reference_calibrators():
    """auto_round/calibration/llm.py:374-402 for a dataset that is not one of the reference's named ones: ones; trailing repeats of a
    sample's last token cleared together with the last position; the last position of every sample cleared"""
    from auto_round_amd.autoround import calibration_attention_mask

    ids = torch.tensor([[1, 2, 3, 4, 5], [7, 8, 9, 9, 9], [4, 4, 4, 4, 4], [1, 2, 2, 3, 2]])
    want = torch.tensor([[1, 1, 1, 1, 0], [1, 1, 0, 0, 0], [0, 0, 0, 0, 0], [1, 1, 1, 1, 0]])
    got = calibration_attention_mask(ids)
    assert got.dtype == torch.long and torch.equal(got, want)
