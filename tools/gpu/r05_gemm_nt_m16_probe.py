"""The NT GEMM on v_mfma_f32_16x16x32_bf16 (ar_gemm_nt_config(3)) against the 32x32x16 form (config 0) and hipBLASLt on the eight forward /
input-gradient shapes of the Llama-3-8B block and in the grouped form at Mixtral-8x7B's expert shapes: bits against the library and
between the two forms, interleaved timing, run-to-run identical bits.

    python tools/gpu/r05_gemm_nt_m16_probe.py --out gpurun_out/r05/gemm_nt_m16_probe.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops  # noqa: E402
from auto_round_amd._lib import load  # noqa: E402


def ndiff(a, b):
    return int((a.contiguous().view(torch.int16) != b.contiguous().view(torch.int16)).sum())


def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    lib = load()
    T = 16384
    shapes = {"q_o_fwd": (T, 4096, 4096), "kv_fwd": (T, 1024, 4096), "gate_up_fwd": (T, 14336, 4096), "down_fwd": (T, 4096, 14336),
              "kv_dx": (T, 4096, 1024), "gate_up_dx": (T, 4096, 14336), "down_dx": (T, 14336, 4096), "gate_up_merged_fwd": (T, 28672, 4096),
              "ragged_rows": (1000, 512, 384)}
    res = {"device": torch.cuda.get_device_name(0), "dense": [], "grouped": []}
    for name, (M, N, K) in shapes.items():
        A, B = rnd((M, K), 1), rnd((N, K), 2, 0.05)
        ref = torch.mm(A, B.t())
        outs, fns = {}, {"hipblaslt": lambda o=torch.empty((M, N), dtype=torch.bfloat16, device="cuda"): torch.mm(A, B.t(), out=o)}
        for v in (0, 3):
            lib.ar_gemm_nt_config(v)
            o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert ops.gemm_nt(A, B, o)
            o2 = torch.empty_like(o)
            ops.gemm_nt(A, B, o2)
            assert ndiff(o, o2) == 0
            outs[v] = o

            def f(v=v, o=torch.empty((M, N), dtype=torch.bfloat16, device="cuda")):
                lib.ar_gemm_nt_config(v)
                ops.gemm_nt(A, B, o)
            fns[f"v{v}"] = f
        rec = dict(shape=name, M=M, N=N, K=K, nan=int(torch.isnan(outs[3].float()).sum()), v3_vs_v0_differing=ndiff(outs[3], outs[0]),
                   v3_vs_library_differing=ndiff(outs[3], ref), v0_vs_library_differing=ndiff(outs[0], ref))
        times = {k: [] for k in fns}
        for _ in range(args.rounds):
            for k, f in fns.items():
                times[k].append(timed(f, args.reps))
        for k, ts in times.items():
            ms = sorted(ts)[len(ts) // 2]
            rec[f"{k}_ms"] = round(ms, 4)
            rec[f"{k}_pflops"] = round(2.0 * M * N * K / ms / 1e12, 4)
        res["dense"].append(rec)
        print(json.dumps(rec), flush=True)
        del A, B, ref, outs, fns
        torch.cuda.empty_cache()
    counts = [4100, 3900, 4500, 3700, 4096, 4200, 4000, 4272]
    R, E = sum(counts), len(counts)
    for name, (N, K) in {"w1_w3_fwd": (28672, 4096), "w2_fwd": (4096, 14336), "w2_dx": (14336, 4096)}.items():
        A = rnd((R, K), 3)
        W = rnd((E * N, K), 4, 0.05)
        row_off = torch.tensor([0] + torch.tensor(counts).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
        b_off = torch.arange(E, dtype=torch.int64, device="cuda") * (N * K)
        outs, fns = {}, {}
        for v in (0, 3):
            lib.ar_gemm_nt_config(v)
            o = torch.full((R, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            assert ops.gemm_nt_grouped(A, W, o, row_off, b_off, N, K)
            outs[v] = o

            def f(v=v, o=torch.empty((R, N), dtype=torch.bfloat16, device="cuda")):
                lib.ar_gemm_nt_config(v)
                ops.gemm_nt_grouped(A, W, o, row_off, b_off, N, K)
            fns[f"v{v}"] = f
        lib0 = torch.mm(A[:counts[0]], W[:N].t())
        rec = dict(grouped=name, N=N, K=K, rows=R, nan=int(torch.isnan(outs[3].float()).sum()), v3_vs_v0_differing=ndiff(outs[3], outs[0]),
                   v3_expert0_vs_library_differing=ndiff(outs[3][:counts[0]], lib0))
        times = {k: [] for k in fns}
        for _ in range(args.rounds):
            for k, f in fns.items():
                times[k].append(timed(f, max(2, args.reps // 2)))
        for k, ts in times.items():
            ms = sorted(ts)[len(ts) // 2]
            rec[f"{k}_ms"] = round(ms, 4)
            rec[f"{k}_pflops"] = round(2.0 * R * N * K / ms / 1e12, 4)
        res["grouped"].append(rec)
        print(json.dumps(rec), flush=True)
        del A, W, outs, fns
        torch.cuda.empty_cache()
    lib.ar_gemm_nt_config(3)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
