#!/usr/bin/env python
"""(test infrastructure; lives under tests/ because it uses the oracle)
End-to-end parity + speed report on one MI355X: auto_round_amd (HIP path) vs oracle/torch_ref (the pinned torch
restatement of the reference loop, i.e. what the reference's eager path computes) on the SAME device, seeds, data and
index schedule, for full-size blocks and the full 200 iterations.  Writes one JSON line per workload.

Trajectories are chaotic (sign-SGD; GEMM summation order differs between `torch.mm(out=)` and autograd's linear
backward), so the report gives: loss curves (first/last/best), fraction of identical baked weights, identical integer
codes, scale agreement, and wall time of both paths."""
import argparse
import copy
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
from oracle import torch_ref as tr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="opt-125m")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--nsamples", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--group-size", type=int, default=128)
    ap.add_argument("--asym", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    w = B.WORKLOADS[a.workload]
    layer, rope, cfg, n_w = B.build_block(w, a.bits, a.group_size, not a.asym, dev, seed=1234)
    g = torch.Generator(device=dev).manual_seed(2)
    X = torch.randn(a.nsamples, a.seqlen, w["hidden"], generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    others = B.make_others(rope, a.seqlen, dev, X[:1])

    def fwd(blk, x, o):
        out = blk(x, **o)
        return out[0] if isinstance(out, (tuple, list)) else out

    q = SignRoundQuantizer(SignRoundConfig(iters=a.iters, batch_size=8, bits=a.bits, sdpa_backend="auto"), device=dev)
    Y = q.forward_all(layer, X, others)

    blk_m = copy.deepcopy(layer)
    random.seed(42)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    best_m = q.quantize_block(blk_m, X, others, Y, None, None)
    torch.cuda.synchronize(); t_mine = time.perf_counter() - t0
    st = dict(q.last_stats)

    blk_o = copy.deepcopy(layer)
    random.seed(42)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=a.iters, batch_size=8, forward=fwd)
    torch.cuda.synchronize(); t_ref = time.perf_counter() - t0

    same_w, same_int, same_scale, tot = 0, 0, 0, 0
    per_layer = {}
    for (n1, m1), (n2, m2) in zip(blk_o.named_modules(), blk_m.named_modules()):
        if isinstance(m1, torch.nn.Linear) and hasattr(m1, "scale"):
            gs = m1.weight.shape[1] // m1.scale.shape[1]
            s1 = m1.scale.float().to(dev).repeat_interleave(gs, 1); s2 = m2.scale.float().to(dev).repeat_interleave(gs, 1)
            i1 = torch.round(m1.weight.float() / s1); i2 = torch.round(m2.weight.float() / s2)
            n = m1.weight.numel()
            per_layer[n1] = {"weights_equal": float((m1.weight == m2.weight).float().mean()),
                             "ints_equal": float((i1 == i2).float().mean()),
                             "scales_equal": float((m1.scale == m2.scale).float().mean())}
            same_w += int((m1.weight == m2.weight).sum()); same_int += int((i1 == i2).sum()); tot += n
            same_scale += int((m1.scale == m2.scale).sum())
    out = {"workload": w["desc"], "iters": a.iters, "bits": a.bits, "group_size": a.group_size, "sym": not a.asym,
           "mi355x_path": {"seconds": t_mine, "init_loss": st["init_loss"], "best_loss": st["best_loss"], "best_iter": st["best_iter"]},
           "torch_ref_same_gpu": {"seconds": t_ref, "init_loss": info["losses"][0], "best_loss": info["best_loss"],
                                  "best_iter": info["best_iter"], "last_loss": info["losses"][-1]},
           "speedup_vs_eager_torch_on_same_gpu": t_ref / t_mine,
           "fraction_identical_baked_weights": same_w / tot, "fraction_identical_integer_codes": same_int / tot,
           "per_layer": per_layer}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
