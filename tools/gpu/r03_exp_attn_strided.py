"""experiment: does aten::_scaled_dot_product_efficient_attention_backward accept q / k / v that are column slices of one merged
[T, 3H] projection output (token stride 3H) -- same results, same speed as contiguous copies?  (OPT-125M shape)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from auto_round_amd import ops

B, H, S, D = 8, 12, 2048, 64
T, HD = B * S, H * D
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(T, 3 * HD, generator=g, device="cuda").to(torch.bfloat16)
do = (torch.randn(T, HD, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
sc = 0.125
qc, kc, vc = (qkv[:, i * HD:(i + 1) * HD].contiguous() for i in range(3))
out, lse = ops.attn_fwd(qc, kc, vc, B, S, H, D, scale=sc)
h4 = lambda t: t.view(B, S, H, D).transpose(1, 2)
hs = lambda i: qkv.view(B, S, 3 * H, D)[:, :, i * H:(i + 1) * H].transpose(1, 2)
z = torch.zeros((), dtype=torch.int64)


def bwd(q, k, v):
    return torch.ops.aten._scaled_dot_product_efficient_attention_backward(h4(do), q, k, v, None, h4(out), lse, z, z, 0.0, (True, True, True, False), True, scale=sc)[:3]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = bwd(h4(qc), h4(kc), h4(vc))
rec = {"contiguous_ms": timeit(lambda: bwd(h4(qc), h4(kc), h4(vc)))}
try:
    got = bwd(hs(0), hs(1), hs(2))
    rec["strided_ms"] = timeit(lambda: bwd(hs(0), hs(1), hs(2)))
    rec["identical"] = [bool(torch.equal(a, b)) for a, b in zip(ref, got)]
    rec["max_abs_diff"] = [float((a.float() - b.float()).abs().max()) for a, b in zip(ref, got)]
    rec["grad_strides"] = [list(t.stride()) for t in got]
except Exception as e:
    rec["strided_error"] = repr(e)[:300]
rec["contig_grad_strides"] = [list(t.stride()) for t in ref]
print(json.dumps(rec))
