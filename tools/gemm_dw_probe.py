"""A/B of the hand-written MFMA weight-gradient GEMM (ar_gemm_dw, csrc/ar_gemm.hip) against hipBLASLt's torch.mm(dY.t(), X) on
the Llama-3-8B layer shapes at K = 8 x 2048 tokens: correctness (vs an fp32 product), kernel time (interleaved rounds, device
events), PFLOP/s.  Prints one JSON object per (shape, variant)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_round_amd import ops  # noqa: E402
from auto_round_amd._lib import load  # noqa: E402

SHAPES = {"gate_up": (14336, 4096), "down": (4096, 14336), "q_o": (4096, 4096), "qkv_merged": (6144, 4096),
          "gate_up_merged": (28672, 4096), "kv": (1024, 4096), "opt_fc1": (3072, 768), "opt_qkv": (768, 768)}


def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="q_o,gate_up,down,qkv_merged,kv")
    ap.add_argument("--K", type=int, default=16384)
    ap.add_argument("--variants", default="2:2,2:1,2:0,1:2")       # sem:order
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda")
    lib = load()
    torch.manual_seed(0)
    # correctness first, on a shape small enough for an fp32 reference: asymmetric random operands
    for sem in (1, 2):
        lib.ar_gemm_dw_config(sem, 2)
        K, M, N = 256, 512, 256
        dY = torch.randn(K, M, device=dev).to(torch.bfloat16)
        X = torch.randn(K, N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ok = ops.gemm_dw(dY, X, out)
        ref = dY.float().t() @ X.float()
        err = (out.float() - ref).abs().max().item()
        exact = (out == ref.to(torch.bfloat16)).float().mean().item()
        print(json.dumps({"check": "small", "sem": sem, "took_kernel": ok, "max_abs_err": err, "frac_equal_to_rounded_fp32": exact}), flush=True)
    best_sem = None
    for sem in (1, 2):
        lib.ar_gemm_dw_config(sem, 2)
        out = torch.empty(512, 256, dtype=torch.bfloat16, device=dev)
        dY = torch.randn(256, 512, device=dev).to(torch.bfloat16)
        X = torch.randn(256, 256, device=dev).to(torch.bfloat16)
        ops.gemm_dw(dY, X, out)
        if (out.float() - dY.float().t() @ X.float()).abs().max().item() < 0.5:
            best_sem = sem
    print(json.dumps({"correct_sem": best_sem}), flush=True)
    if best_sem is None:
        return
    # accumulate + strided operands (column slices of wider buffers), multi-tile, all three tile orders
    for order in (0, 1, 2):
        lib.ar_gemm_dw_config(best_sem, order)
        K, M, N = 512, 1024, 2048
        big_y = torch.randn(K, M + 512, device=dev).to(torch.bfloat16)
        big_x = torch.randn(K, N + 256, device=dev).to(torch.bfloat16)
        dY, X = big_y[:, 256:256 + M], big_x[:, 256:256 + N]
        out = torch.randn(M, N, device=dev).to(torch.bfloat16)
        old = out.clone()
        assert ops.gemm_dw(dY, X, out, accumulate=True)
        ref = (old.float() + dY.float().t() @ X.float())
        err = ((out.float() - ref).abs() / (ref.abs() + 1.0)).max().item()
        print(json.dumps({"check": "strided+accumulate", "order": order, "max_rel_err": err}), flush=True)
    for name in args.shapes.split(","):
        M, N = SHAPES[name]
        K = args.K
        dY = torch.randn(K, M, device=dev).to(torch.bfloat16)
        X = torch.randn(K, N, device=dev).to(torch.bfloat16)
        out_t = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out_k = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        flops = 2.0 * M * N * K
        variants = [tuple(int(x) for x in v.split(":")) for v in args.variants.split(",") if int(v.split(":")[0]) == best_sem or v.startswith(str(best_sem))]
        variants = [(best_sem, o) for o in sorted({v[1] for v in variants})]

        def lib_mm():
            torch.mm(dY.t(), X, out=out_t)

        fns = {"hipblaslt": lib_mm}
        for sem, order in variants:
            def f(sem=sem, order=order):
                lib.ar_gemm_dw_config(sem, order)
                ops.gemm_dw(dY, X, out_k)
            fns[f"mfma_sem{sem}_order{order}"] = f
        for f in fns.values():       # warm-up
            f()
        torch.cuda.synchronize()
        times = {k: [] for k in fns}
        for _ in range(args.rounds):  # interleaved rounds
            for k, f in fns.items():
                times[k].append(timed(f, args.reps))
        lib_mm()
        fns[f"mfma_sem{best_sem}_order2"]() if f"mfma_sem{best_sem}_order2" in fns else None
        torch.cuda.synchronize()
        diff = (out_k.float() - out_t.float()).abs().max().item()
        scale = out_t.float().abs().mean().item()
        for k, ts in times.items():
            ms = sorted(ts)[len(ts) // 2]
            print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "impl": k, "ms_median": ms, "ms_min": min(ts),
                              "pflops": flops / ms / 1e12, "max_abs_diff_vs_hipblaslt": diff if k != "hipblaslt" else 0.0,
                              "mean_abs_out": scale}), flush=True)


if __name__ == "__main__":
    main()
