"""`ops.mask_structure` (host logic, runs on CPU tensors too): which attention masks the first-party masked forward takes -- the
calibration flow's structured 0 / 1 bias (auto_round/calibration/llm.py:360-402 + inputs.py:100-107) -- and which stay with torch SDPA."""
import torch

from auto_round_amd import ops


def _mask(S, cleared=1, b_in=1.0, b_out=0.0, dtype=torch.bfloat16):
    keep = torch.tril(torch.ones(S, S, dtype=torch.bool))
    if cleared:
        keep[:, S - cleared:] = False
    return torch.where(keep, torch.tensor(b_in), torch.tensor(b_out)).to(dtype)[None, None]


def test_the_calibration_masks_are_recognised_with_their_valid_length():
    assert ops.mask_structure(_mask(64, 1), 64) == (1.0, 0.0, 63)
    assert ops.mask_structure(_mask(64, 0), 64) == (1.0, 0.0, 64)
    assert ops.mask_structure(_mask(64, 9), 64) == (1.0, 0.0, 55)
    assert ops.mask_structure(_mask(64, 1, 0.5, -2.0, torch.float32), 64) == (0.5, -2.0, 63)
    assert ops.mask_structure(_mask(64, 1).expand(4, 1, 64, 64).contiguous(), 64) == (1.0, 0.0, 63)


def test_everything_else_keeps_torch_sdpa():
    assert ops.mask_structure(None, 64) is None
    assert ops.mask_structure(_mask(64, 1, 0.0, float("-inf")), 64) is None                  # hard mask
    assert ops.mask_structure(_mask(64, 1, 0.0, torch.finfo(torch.bfloat16).min), 64) is None   # transformers' additive min-value mask
    assert ops.mask_structure(_mask(64, 1)[:, :, :, :32], 64) is None                        # not [S, S]
    assert ops.mask_structure(_mask(64, 1).bool(), 64) is None                               # boolean: torch's own semantics
    m = _mask(64, 1).clone()
    m[0, 0, 10, 3] = 0.5
    assert ops.mask_structure(m, 64) is None                                                 # unstructured
    m = _mask(64, 1).clone()
    m[0, 0, :, 20] = 0.0                                                                     # a hole in the middle: not trailing padding
    assert ops.mask_structure(m, 64) is None
    two = _mask(64, 1).expand(2, 1, 64, 64).contiguous()
    two[1, 0, :, 40:] = 0.0                                                                  # per-sample padding
    assert ops.mask_structure(two, 64) is None


def test_the_verdict_follows_the_tensor_not_its_address():
    a = _mask(32, 1)
    assert ops.mask_structure(a, 32) == (1.0, 0.0, 31)
    a[0, 0, 5, 1] = 0.25                                                                     # same object, new version
    assert ops.mask_structure(a, 32) is None
    for cleared in (0, 2, 5):                                                                # fresh tensors, possibly recycled storage
        assert ops.mask_structure(_mask(32, cleared), 32) == (1.0, 0.0, 32 - cleared)
