"""timing of csrc/ar_attn_bwd.hip against the library's attention backward at OPT-125M's tuning minibatch (8 x 12 x 2048 x 64)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from auto_round_amd import ops

B, H, S, D = 8, 12, 2048, 64
T, HD = B * S, H * D
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(T, 3 * HD, generator=g, device="cuda").to(torch.bfloat16)
do = (torch.randn(T, HD, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
sc = 0.125
out, lse = ops.attn_fwd(q, k, v, B, S, H, D, scale=sc)
dqkv = torch.empty_like(qkv)
h4 = lambda t: t.reshape(B, S, H, D).transpose(1, 2)
z = torch.zeros((), dtype=torch.int64)


def lib():
    return torch.ops.aten._scaled_dot_product_efficient_attention_backward(h4(do), h4(q), h4(k), h4(v), None, h4(out), lse, z, z, 0.0, (True, True, True, False), True, scale=sc)


def mine():
    return ops.attn_bwd(q, k, v, out, lse, do, B, S, H, D, scale=sc, dq=dqkv[:, :HD], dk=dqkv[:, HD:2 * HD], dv=dqkv[:, 2 * HD:])


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


a = mine()
b = lib()
err = [float((x.float() - y.transpose(1, 2).reshape(T, HD).float()).abs().max()) for x, y in zip(a, b[:3])]
flops = 2.5 * 4 * B * H * S * S * D / 2
rec = {"shape": [B, H, S, D], "ours_ms": timeit(mine), "library_ms": timeit(lib), "max_abs_diff_vs_library": err}
rec["ours_PFLOPs_at_5_gemm_flops"] = flops / rec["ours_ms"] / 1e12
rec["library_PFLOPs"] = flops / rec["library_ms"] / 1e12
print(json.dumps(rec))
