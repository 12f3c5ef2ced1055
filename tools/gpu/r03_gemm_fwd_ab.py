"""VERDICT r02 item 5 ("forward / dX GEMM: build it or bury it with numbers"), measured instead of argued: the forward products of the
Llama-3-8B block Y[T, N] = X[T, K] W[N, K]^T through (a) the library (hipBLASLt's NT kernel, what the product runs) and (b) THIS
repository's MFMA kernel -- csrc/ar_gemm.hip computes out[M, N] = A[k, M]^T B[k, N], so fed with X^T and W^T it computes the same
product; the transposed weight copy already exists for the input-gradient GEMMs, X^T costs one ar_transpose16 pass per call.
T = 16384 tokens, bf16, fp32 accumulate."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from auto_round_amd import ops

dev = torch.device("cuda:0")
T = 16384
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(torch.bfloat16)


def t_us(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000


for name, (N, K) in dict(qkv=(6144, 4096), o=(4096, 4096), gate_up=(28672, 4096), down=(4096, 14336)).items():
    X, W = rnd(T, K), rnd(N, K, sc=0.02)
    Wt = W.t().contiguous()
    Xt = torch.empty(K, T, dtype=torch.bfloat16, device=dev)
    out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
    ref = F.linear(X, W)
    assert ops.transpose16(X, out=Xt) is not None
    assert ops.gemm_dw(Xt, Wt, out, accumulate=False)
    err = (out.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
    lib = t_us(lambda: F.linear(X, W))
    ours = t_us(lambda: ops.gemm_dw(Xt, Wt, out, accumulate=False))
    tr = t_us(lambda: ops.transpose16(X, out=Xt))
    fl = 2.0 * T * N * K
    rec = dict(shape=name, T=T, N=N, K=K, hipblaslt_us=round(lib, 1), ours_kernel_us=round(ours, 1), transpose_x_us=round(tr, 1),
               hipblaslt_PF=round(fl / lib / 1e9, 3), ours_kernel_PF=round(fl / ours / 1e9, 3),
               ours_with_transpose_PF=round(fl / (ours + tr) / 1e9, 3), max_rel_diff=err)
    print(json.dumps(rec), flush=True)
