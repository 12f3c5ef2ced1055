#!/usr/bin/env python
"""The launches behind profiles/pmc_traffic.json, for two separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE` passes
(MI355X_MICROARCH.md, HBM section): a calibration copy, the quant forward (K1) and the three variants of the fused backward +
sign-SGD kernel (K2 plain, K2 + best-parameter snapshot, K2 + next forward) at the Llama-3-8B block size, a few launches each, in
this fixed order.  `tools/rocprof_summary.py <db> --pmc-rows rows.csv` then lists the counter per dispatch in order and
`tools/pmc_traffic_merge.py` turns the two row files into the JSON bench.py reads.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_round_amd import ops

N, GS, REP = 218103808, 128, 4


def main():
    n, gs = N, GS
    G = n // gs
    g = torch.Generator(device="cuda").manual_seed(0)
    W = (torch.randn(n, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    V = torch.rand(n, generator=g, device="cuda") - 0.5
    dWq = (torch.randn(n, generator=g, device="cuda") * 1e-3).to(torch.bfloat16)
    ms, Ms = torch.ones(G, device="cuda"), torch.ones(G, device="cuda")
    wmin, wmax = ops.group_minmax(W, gs)
    Wq = torch.empty_like(W)
    bV, bmin, bmax = V.clone(), ms.clone(), Ms.clone()
    flag = torch.ones(1, dtype=torch.int32, device="cuda")
    lr = torch.tensor([1e-9], device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        Wq.copy_(W)                                                     # calibration: 2 B/elem read + 2 B/elem written
    for _ in range(REP):
        ops.qdq_int_fwd(W, V, wmin, wmax, ms, Ms, gs=gs, bits=4, sym=True, out=Wq)
    for _ in range(REP):
        ops.qdq_int_bwd_sgd_(dWq, W, V, wmin, wmax, ms, Ms, gs=gs, bits=4, sym=True, lr_v=lr, lr_mm=lr)
    for _ in range(REP):
        ops.qdq_int_bwd_sgd_(dWq, W, V, wmin, wmax, ms, Ms, gs=gs, bits=4, sym=True, lr_v=lr, lr_mm=lr, snapshot_flag=flag, best_V=bV,
                             best_min=bmin, best_max=bmax)
    for _ in range(REP):
        ops.qdq_int_bwd_sgd_(dWq, W, V, wmin, wmax, ms, Ms, gs=gs, bits=4, sym=True, lr_v=lr, lr_mm=lr, Wq_next=Wq)
    torch.cuda.synchronize()
    print("ok", n, G)


if __name__ == "__main__":
    main()
