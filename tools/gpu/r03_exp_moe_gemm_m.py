"""Per-expert GEMM time against the expert's row count: does hipBLASLt's heuristic pick slower kernels for ragged M (Mixtral-8x7B
expert shapes, bf16)?  fwd: [M, K] @ W[N, K]^T ; dx: dY[M, N] @ W[N, K]"""
import json, sys, torch
H, F = 4096, 14336
dev = "cuda"
Wgu = torch.randn(2 * F, H, device=dev, dtype=torch.bfloat16)
Wd = torch.randn(H, F, device=dev, dtype=torch.bfloat16)


def t(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000


out = []
for M in (3840, 3901, 3968, 4000, 4037, 4096, 4100, 4163, 4224, 4352, 4608):
    x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
    a = torch.randn(M, F, device=dev, dtype=torch.bfloat16)
    g = torch.randn(M, 2 * F, device=dev, dtype=torch.bfloat16)
    GU = torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
    D = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    dA = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
    rec = dict(M=M,
               fwd_gu_us=t(lambda: torch.mm(x, Wgu.t(), out=GU)), fwd_d_us=t(lambda: torch.mm(a, Wd.t(), out=D)),
               dx_gu_us=t(lambda: torch.mm(g, Wgu, out=D)), dx_d_us=t(lambda: torch.mm(x, Wd, out=dA)))
    fl_gu, fl_d = 2.0 * M * H * 2 * F, 2.0 * M * H * F
    rec.update(fwd_gu_PF=fl_gu / rec["fwd_gu_us"] / 1e9, fwd_d_PF=fl_d / rec["fwd_d_us"] / 1e9, dx_gu_PF=fl_gu / rec["dx_gu_us"] / 1e9,
               dx_d_PF=fl_d / rec["dx_d_us"] / 1e9)
    out.append(rec)
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)
