"""Exact attention kernels at a QUARTER of the tuning minibatch (B = 2: the 2^29 counter ceiling) for rocprofv3 --pmc passes; with
`--cfg N` the launch form (ar_attn_exact_config).  Summarise with r06_pmc_attn_kernels.py --summarise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from auto_round_amd import ops, _lib
cfgs = [int(x) for x in sys.argv[sys.argv.index("--cfg") + 1].split(",")] if "--cfg" in sys.argv else [0]
lib = _lib.load()
torch.manual_seed(0)
B, H, S, D, hk = 2, 32, 2048, 128, 8
scale = D ** -0.5
q = torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
k = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
idx = torch.arange(S, device="cuda")
keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < S - 1)
mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
st = ops.mask_structure(mask, S)
da = (torch.randn(B, S, H, D, device="cuda") * 0.02).to(torch.bfloat16)
with torch.no_grad():
    for cfg in cfgs:
        lib.ar_attn_exact_config(cfg)
        for _ in range(3):
            o, lse = ops.attn_fwd_exact(q, k, v, st, scale)
            ops.attn_bwd_exact(q, k, v, o, lse, da, st, scale)
torch.cuda.synchronize()
