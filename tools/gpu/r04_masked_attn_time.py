"""ar_attn_fwd_masked / ar_attn_bwd_masked vs torch SDPA (AOTriton efficient, additive mask) at the Llama-3-8B minibatch: ms per call."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops  # noqa: E402

dev = "cuda"
B, S, H, D = 8, 2048, 32, 128
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(B * S, H * D, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
keep = torch.tril(torch.ones(S, S, dtype=torch.bool, device=dev))
keep[:, -1] = False
mask = keep.to(torch.bfloat16)[None, None]
st = ops.mask_structure(mask, S)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


q4, k4, v4 = (t.view(B, S, H, D).transpose(1, 2) for t in (q, k, v))
out = dict(struct=st,
           ms_first_party_masked=timed(lambda: ops.attn_fwd(q, k, v, B, S, H, D, mask_struct=st)),
           ms_first_party_causal=timed(lambda: ops.attn_fwd(q, k, v, B, S, H, D)),
           ms_torch_sdpa_mask=timed(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, attn_mask=mask)),
           ms_torch_sdpa_causal=timed(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, is_causal=True)))
do = (0.1 * torch.randn(B * S, H * D, device=dev, generator=g)).to(torch.bfloat16)
o_mine, lse = ops.attn_fwd(q, k, v, B, S, H, D, mask_struct=st)
z = torch.zeros((), dtype=torch.int64)
bias = mask.expand(B, H, S, S)
o4, do4 = o_mine.view(B, S, H, D).transpose(1, 2), do.view(B, S, H, D).transpose(1, 2)
out["ms_bwd_first_party_masked"] = timed(lambda: ops.attn_bwd(q, k, v, o_mine, lse, do, B, S, H, D, mask_struct=st))
out["ms_bwd_torch_efficient_bias"] = timed(lambda: torch.ops.aten._scaled_dot_product_efficient_attention_backward(
    do4, q4, k4, v4, bias, o4, lse, z, z, 0.0, (True, True, True, False), False))
print(json.dumps(out))
os.makedirs("gpurun_out/r04i", exist_ok=True)
json.dump(out, open("gpurun_out/r04i/masked_attn_time.json", "w"), indent=1)
