"""Follow-up of r03_exp_moe_gemm_m.py: an expert with a few rows more than 4096 costs a second round of 256 x 256 tiles in the
down-projection forward and the gate/up input-gradient GEMM (N = 4096 output columns -> 16 tiles per 256 rows).  Is one call over M rows
slower than 4096 rows + the remainder, or two halves?"""
import json, torch
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


def chunks(M, plan):
    if plan == "one":
        return [(0, M)]
    if plan == "4096+rest":
        return [(0, 4096), (4096, M)] if M > 4096 else [(0, M)]
    if plan == "halves":
        h = (M // 2 + 7) // 8 * 8
        return [(0, h), (h, M)]
    if plan == "3968+rest":
        return [(0, 3968), (3968, M)]


for M in (4100, 4133, 4163, 4224, 4352, 4500):
    a = torch.randn(M, F, device=dev, dtype=torch.bfloat16)
    g = torch.randn(M, 2 * F, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
    D = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    GU = torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
    rec = dict(M=M)
    for plan in ("one", "4096+rest", "halves", "3968+rest"):
        cs = chunks(M, plan)
        rec["fwd_d " + plan] = round(t(lambda: [torch.mm(a[s:e], Wd.t(), out=D[s:e]) for s, e in cs]), 1)
        rec["dx_gu " + plan] = round(t(lambda: [torch.mm(g[s:e], Wgu, out=D[s:e]) for s, e in cs]), 1)
        rec["fwd_gu " + plan] = round(t(lambda: [torch.mm(x[s:e], Wgu.t(), out=GU[s:e]) for s, e in cs]), 1)
    print(json.dumps(rec), flush=True)
for M in (3700, 3800, 3840, 3880):
    a = torch.randn(M, F, device=dev, dtype=torch.bfloat16)
    g = torch.randn(M, 2 * F, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
    D = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    GU = torch.empty(M, 2 * F, device=dev, dtype=torch.bfloat16)
    rec = dict(M=M, fwd_gu=round(t(lambda: torch.mm(x, Wgu.t(), out=GU)), 1), fwd_d=round(t(lambda: torch.mm(a, Wd.t(), out=D)), 1),
               dx_gu=round(t(lambda: torch.mm(g, Wgu, out=D)), 1))
    print(json.dumps(rec), flush=True)
