#!/usr/bin/env python
"""Self-contained reproducer (torch only) of the library attention race described in DESIGN.md section 5, for an upstream report.

torch 2.10.0+rocm7.0 on MI355X (gfx950): `F.scaled_dot_product_attention` at q / k / v [8, 12, 2048, 64] bf16 (token-major storage,
as a transformers OPT block hands them over) with an [8, 1, 2048, 2048] additive bf16 mask returns, on a fraction of its calls,
16 (sometimes 32) output values that differ from every other call's by as much as the values themselves (up to 2.4 on the N(0, 1)
operands below; 0.02-0.04 on an OPT-125M block's activations) -- identical inputs, nothing else on the GPU.  Measured with this script
(profiles/r06_sdpa_race_repro.txt): 36 of 2999 calls; with a 128 MB streaming kernel before every call 29 of 1499.  In a tuning loop the
rate depends on what else runs; AMD_SERIALIZE_KERNEL=3 does not remove it, so the race is inside the kernel.

    python tools/sdpa_race_repro.py [calls] [--flush]
"""
import sys

import torch
import torch.nn.functional as F


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3000
    flush = "--flush" in sys.argv
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    B, H, S, D = 8, 12, 2048, 64
    q, k, v = (torch.randn(B, S, H, D, device=dev, generator=g).to(torch.bfloat16).transpose(1, 2) for _ in range(3))
    keep = torch.tril(torch.ones(S, S, dtype=torch.bool, device=dev))
    keep[:, -1] = False
    mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()          # the reference's 0 / 1 additive bias
    big = torch.zeros(1 << 25, device=dev) if flush else None
    ref, bad = None, []
    stats = torch.zeros(n, 2, device=dev)
    with torch.no_grad():
        for i in range(n):
            if flush:
                big.add_(1.0)
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, scale=1.0, is_causal=False)
            if ref is None:
                ref = o.clone()
                continue
            ne = o.view(torch.int16) != ref.view(torch.int16)
            stats[i, 0] = ne.sum()
            stats[i, 1] = (o.float() - ref.float()).abs().max()
    torch.cuda.synchronize()
    st = stats.cpu()
    bad = st[:, 0].nonzero().flatten().tolist()
    print(f"torch {torch.__version__} on {torch.cuda.get_device_name(0)}: {len(bad)} of {n - 1} calls differ from call 0"
          f"{' (with a 128 MB streaming kernel before every call)' if flush else ''}")
    for i in bad[:10]:
        print(f"  call {i}: {int(st[i, 0])} values differ, max |diff| {float(st[i, 1]):.4f}")


if __name__ == "__main__":
    main()
