#!/usr/bin/env python
"""Micro-benchmark of the quant kernels at BASELINE sizes (achieved algorithmic GB/s). GPU box only."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_round_amd import ops


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=218103808)   # Llama-3-8B block
    ap.add_argument("--gs", type=int, default=128)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--asym", action="store_true")
    a = ap.parse_args()
    n, gs = a.n // a.gs * a.gs, a.gs
    G = n // gs
    g = torch.Generator(device="cuda").manual_seed(0)
    W = (torch.randn(n, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    V = torch.rand(n, generator=g, device="cuda") - 0.5
    dWq = (torch.randn(n, generator=g, device="cuda") * 1e-3).to(torch.bfloat16)
    ms = torch.ones(G, device="cuda"); Ms = torch.ones(G, device="cuda")
    wmin, wmax = ops.group_minmax(W, gs)
    Wq = torch.empty_like(W)
    lr = torch.tensor([1e-9], device="cuda")
    sym = not a.asym
    res = {}
    t = timeit(lambda: ops.qdq_int_fwd(W, V, wmin, wmax, ms, Ms, gs=gs, bits=a.bits, sym=sym, out=Wq))
    res["fwd_ms"] = t; res["fwd_GBps"] = (8 * n + 12 * G) / t / 1e6
    t = timeit(lambda: ops.qdq_int_bwd_sgd_(dWq, W, V, wmin, wmax, ms, Ms, gs=gs, bits=a.bits, sym=sym, lr_v=lr, lr_mm=lr))
    res["bwd_sgd_ms"] = t; res["bwd_sgd_GBps"] = (12 * n + 8 * G) / t / 1e6
    t = timeit(lambda: ops.qdq_int_bwd_sgd_(dWq, W, V, wmin, wmax, ms, Ms, gs=gs, bits=a.bits, sym=sym, lr_v=lr, lr_mm=lr, Wq_next=Wq))
    res["bwd_sgd_fwd_ms"] = t; res["bwd_sgd_fwd_GBps"] = (14 * n + 8 * G) / t / 1e6
    t = timeit(lambda: Wq.copy_(W))
    res["copy_bf16_ms"] = t; res["copy_GBps"] = 4 * n / t / 1e6
    t = timeit(lambda: ops.group_minmax(W, gs))
    res["minmax_ms"] = t; res["minmax_GBps"] = 2 * n / t / 1e6
    # fp4 weight kernels (MXFP4 g32, NVFP4 g16)
    for name, mode, fgs in (("mx", 0, 32), ("nv", 1, 16)):
        Gf = n // fgs
        absmax, tmax = ops.group_absmax(W, fgs, want_tensor_max=True)
        gsc = (448.0 * 6.0 / tmax) if mode == 1 else None
        Msf = torch.ones(Gf, device="cuda")
        t = timeit(lambda: ops.qdq_fp4_fwd(W, V, absmax, Msf, mode=mode, gs=fgs, global_scale=gsc, out=Wq))
        res[f"{name}fp4_fwd_GBps"] = (8 * n + 8 * Gf) / t / 1e6
        t = timeit(lambda: ops.qdq_fp4_bwd_sgd_(dWq, W, V, absmax, Msf, mode=mode, gs=fgs, global_scale=gsc, lr_v=lr, lr_mm=lr))
        res[f"{name}fp4_bwd_sgd_GBps"] = (12 * n + 8 * Gf) / t / 1e6
        t = timeit(lambda: ops.qdq_fp4_fwd(W, None, None, None, mode=mode, gs=fgs, global_scale=gsc, out=Wq))
        res[f"{name}fp4_act_fwd_GBps"] = 4 * n / t / 1e6
        t = timeit(lambda: ops.fp4_act_bwd(dWq, W, mode=mode, gs=fgs, global_scale=gsc, out=Wq))
        res[f"{name}fp4_act_bwd_GBps"] = 6 * n / t / 1e6
    # dynamic symmetric int8 activation fake-quant: group 32 / 128 (lane groups) and per-token 4096 / 14336 (a wave / a
    # workgroup per group, the group parked in LDS); n = 218,103,808 is a multiple of all four
    for ags in (32, 128, 4096, 14336):
        t = timeit(lambda: ops.qdq_int_act_fwd(W, gs=ags, bits=8, out=Wq))
        res[f"int8_act_g{ags}_fwd_GBps"] = 4 * n / t / 1e6
        t = timeit(lambda: ops.int_act_bwd(dWq, W, gs=ags, bits=8, out=Wq))
        res[f"int8_act_g{ags}_bwd_GBps"] = 6 * n / t / 1e6
    # fused block kernels at the Llama-3-8B minibatch (8 x 2048 tokens, hidden 4096, ffn 14336, 32/8 heads of 128)
    del V, dWq, Wq
    T, H, Fd, hq, hkv, d = 16384, 4096, 14336, 32, 8, 128
    x = torch.randn(T, H, generator=g, device="cuda").to(torch.bfloat16)
    wn = torch.ones(H, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.rmsnorm_fwd(x, wn, 1e-5))
    res["rmsnorm_fwd_GBps"] = 4 * T * H / t / 1e6
    y, rstd = ops.rmsnorm_fwd(x, wn, 1e-5)
    t = timeit(lambda: ops.rmsnorm_bwd(y, x, wn, rstd, dres=x, out=y))
    res["rmsnorm_bwd_add_GBps"] = 8 * T * H / t / 1e6
    gu = torch.randn(T, 2 * Fd, generator=g, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: ops.swiglu_fwd(gu, Fd))
    res["swiglu_fwd_GBps"] = 6 * T * Fd / t / 1e6
    da = ops.swiglu_fwd(gu, Fd)
    t = timeit(lambda: ops.swiglu_bwd_(da, gu, Fd))
    res["swiglu_bwd_GBps"] = 10 * T * Fd / t / 1e6
    del gu, da
    qkv = torch.randn(T, (hq + 2 * hkv) * d, generator=g, device="cuda").to(torch.bfloat16)
    cs = torch.randn(1, 2048, d, generator=g, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: ops.rope_fwd(qkv, cs, cs, 8, 2048, hq, hkv, d))
    res["rope_fwd_GBps"] = 2 * T * d * ((hq + 2 * hkv) + 3 * hq) / t / 1e6
    q, k, v = ops.rope_fwd(qkv, cs, cs, 8, 2048, hq, hkv, d)
    t = timeit(lambda: ops.rope_bwd(q, k, v, cs, cs, 8, 2048, hq, hkv, d, out=qkv))
    res["rope_bwd_GBps"] = 2 * T * d * ((hq + 2 * hkv) + 3 * hq) / t / 1e6
    # round-2 additions: per-head q / k norm (Qwen3 form of the same projection), LayerNorm (hidden 4096 and OPT-125M's 768),
    # the 2-byte transpose of the merged gate / up weight, the attention forward at both head sizes
    wh = torch.ones(d, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.headnorm_fwd(qkv, wh, wh, hq, hkv, d, 1e-6))
    res["headnorm_fwd_GBps"] = 4 * T * (hq + 2 * hkv) * d / t / 1e6
    _, hr = ops.headnorm_fwd(qkv, wh, wh, hq, hkv, d, 1e-6)
    dq_ = torch.randn_like(qkv)
    t = timeit(lambda: ops.headnorm_bwd_(dq_, qkv, wh, wh, hr, hq, hkv, d))
    res["headnorm_bwd_GBps"] = 6 * T * (hq + hkv) * d / t / 1e6
    del dq_, hr
    for Hn in (4096, 768):
        xl = torch.randn(T, Hn, generator=g, device="cuda").to(torch.bfloat16)
        wl = torch.ones(Hn, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: ops.layernorm_fwd(xl, wl, wl, 1e-5))
        res[f"layernorm_h{Hn}_fwd_GBps"] = 4 * T * Hn / t / 1e6
        yl, mean, rs = ops.layernorm_fwd(xl, wl, wl, 1e-5)
        t = timeit(lambda: ops.layernorm_bwd(yl, xl, wl, mean, rs, dres=xl, out=yl))
        res[f"layernorm_h{Hn}_bwd_add_GBps"] = 8 * T * Hn / t / 1e6
    Wt_src = torch.randn(2 * Fd, H, generator=g, device="cuda").to(torch.bfloat16)
    Wt_dst = torch.empty(H, 2 * Fd, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.transpose16(Wt_src, out=Wt_dst))
    res["transpose16_GBps"] = 4 * 2 * Fd * H / t / 1e6
    del Wt_src, Wt_dst
    for heads, hd in ((32, 128), (12, 64)):
        qa, ka, va = (torch.randn(T, heads * hd, generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
        t = timeit(lambda: ops.attn_fwd(qa, ka, va, 8, 2048, heads, hd))
        res[f"attn_fwd_h{heads}_d{hd}_ms"] = t
        res[f"attn_fwd_h{heads}_d{hd}_TFLOPs"] = 4 * 8 * heads * 2048 * 2048 * hd / 2 / t / 1e9      # causal: half the score matrix
    print(json.dumps({k: round(v, 3) for k, v in res.items()}))


if __name__ == "__main__":
    main()
