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
          "gate_up_merged": (28672, 4096), "kv": (1024, 4096), "opt_fc1": (3072, 768), "opt_fc2": (768, 3072), "opt_qkv": (768, 768)}


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
    ap.add_argument("--variants", default="v1_staggered:2,v3:2")       # kernel:order
    ap.add_argument("--ablate", action="store_true", help="also time the v1 ablations (no DMA / no reads / MFMA only): garbage outputs")
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
    lib.ar_gemm_dw_config(1, 2)
    KERNELS = {"v0": 10, "v1_staggered": 11, "v1_lockstep": 12, "v3": 17, "v4_l4": 18, "v4_l8": 19}
    ABLATIONS = {"abl_no_dma": 14, "abl_no_dma_no_reads": 15, "abl_mfma_only": 16}
    # accumulate + strided operands (column slices of wider buffers), multi-tile, all tile orders, every kernel
    for kname, code in list(KERNELS.items()):
        for order in (2,):
            lib.ar_gemm_dw_config(code, order)
            K, M, N = 512, 1024, 2048
            big_y = torch.randn(K, M + 512, device=dev).to(torch.bfloat16)
            big_x = torch.randn(K, N + 256, device=dev).to(torch.bfloat16)
            dY, X = big_y[:, 256:256 + M], big_x[:, 256:256 + N]
            out = torch.randn(M, N, device=dev).to(torch.bfloat16)
            old = out.clone()
            assert ops.gemm_dw(dY, X, out, accumulate=True)
            ref = (old.float() + dY.float().t() @ X.float())
            exact = (out == ref.to(torch.bfloat16)).float().mean().item()
            err = ((out.float() - ref).abs() / (ref.abs() + 1.0)).max().item()
            print(json.dumps({"check": "strided+accumulate", "kernel": kname, "order": order, "max_rel_err": err,
                              "frac_equal_to_rounded_fp32": exact}), flush=True)
    wanted = [v for v in args.variants.split(",")]
    if args.ablate:
        KERNELS = dict(KERNELS, **ABLATIONS)
        wanted += [f"{k}:2" for k in ABLATIONS]
    for name in args.shapes.split(","):
        M, N = SHAPES[name]
        K = args.K
        dY = torch.randn(K, M, device=dev).to(torch.bfloat16)
        X = torch.randn(K, N, device=dev).to(torch.bfloat16)
        out_t = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out_k = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        flops = 2.0 * M * N * K

        def lib_mm():
            torch.mm(dY.t(), X, out=out_t)

        fns = {"hipblaslt": lib_mm}
        for v in wanted:
            kname, order = v.split(":")
            def f(code=KERNELS[kname], order=int(order)):
                lib.ar_gemm_dw_config(code, order)
                ops.gemm_dw(dY, X, out_k)
            fns[f"mfma_{kname}_order{order}"] = f
        diffs = {}
        for k, f in fns.items():       # warm-up + agreement with the library product
            f()
            torch.cuda.synchronize()
            if k != "hipblaslt":
                diffs[k] = ((out_k.float() - out_t.float()).abs().max().item(), (out_k == out_t).float().mean().item())
        times = {k: [] for k in fns}
        for _ in range(args.rounds):  # interleaved rounds
            for k, f in fns.items():
                times[k].append(timed(f, args.reps))
        scale = out_t.float().abs().mean().item()
        for k, ts in times.items():
            ms = sorted(ts)[len(ts) // 2]
            print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "impl": k, "ms_median": ms, "ms_min": min(ts),
                              "pflops": flops / ms / 1e12, "max_abs_diff_vs_hipblaslt": diffs.get(k, (0.0, 1.0))[0],
                              "frac_bit_equal_to_hipblaslt": diffs.get(k, (0.0, 1.0))[1], "mean_abs_out": scale}), flush=True)


if __name__ == "__main__":
    main()
