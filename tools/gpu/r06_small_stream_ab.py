"""K1 / K2 at OPT-125M's stream size (7,077,888 weights: 57 / 85 / 99 MB per launch, inside the 256 MB Infinity Cache) under different
grid caps (library builds with -DAR_GRID_CAP=N under build_ab/, selected with AR_MI355X_LIB): one launch per workgroup-tile (default) vs
a persistent grid-stride grid.  200 launches per figure, event-timed; the Llama-3-8B block size next to it.

    AR_MI355X_LIB=build_ab/cap512/libar_mi355x.so python tools/gpu/r06_small_stream_ab.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from auto_round_amd import ops


def timeit(fn, iters=200, warmup=10):
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
    out = {"lib": os.environ.get("AR_MI355X_LIB", "default")}
    for name, n in (("opt125m", 7077888), ("llama8b", 218103808)):
        gs = 128
        G = n // gs
        g = torch.Generator(device="cuda").manual_seed(0)
        W = (torch.randn(n, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        V = torch.rand(n, generator=g, device="cuda") - 0.5
        dWq = (torch.randn(n, generator=g, device="cuda") * 1e-3).to(torch.bfloat16)
        ms = torch.ones(G, device="cuda")
        Ms = torch.ones(G, device="cuda")
        wmin, wmax = ops.group_minmax(W, gs)
        Wq = torch.empty_like(W)
        lr = torch.tensor([1e-9], device="cuda")
        it = 200 if n < 1e8 else 30
        t = timeit(lambda: ops.qdq_int_fwd(W, V, wmin, wmax, ms, Ms, gs=gs, bits=4, sym=True, out=Wq), it)
        out[f"{name}_k1_us"] = 1000 * t
        out[f"{name}_k1_frac"] = (8 * n + 12 * G) / t / 1e6 / 8000
        t = timeit(lambda: ops.qdq_int_bwd_sgd_(dWq, W, V, wmin, wmax, ms, Ms, gs=gs, bits=4, sym=True, lr_v=lr, lr_mm=lr, Wq_next=Wq), it)
        out[f"{name}_k2fwd_us"] = 1000 * t
        out[f"{name}_k2fwd_frac"] = (14 * n + 8 * G) / t / 1e6 / 8000
        del W, V, dWq, Wq
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
