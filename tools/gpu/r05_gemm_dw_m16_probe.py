"""The weight-gradient GEMM on v_mfma_f32_16x16x32_bf16 (k_gemm_dw6, ar_gemm_dw_config(32)) against the round 2-4 kernel on
32x32x16 (config 30) and against hipBLASLt: correctness on small asymmetric problems (strided operands, accumulate, ragged K, forced
split plans, a cut table), bit equality of the two first-party kernels and of each with the library at the Llama-3-8B shapes, kernel
times (interleaved rounds, device events) in the one-pass form, in the library's stream-K structure (ar_gemm_dw_sk with the structure
streamk.py finds) and in the grouped form at Mixtral-8x7B's expert shapes, run-to-run identical bits.

    python tools/gpu/r05_gemm_dw_m16_probe.py --out gpurun_out/r05/gemm_dw_m16_probe.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops, streamk  # noqa: E402
from auto_round_amd._lib import load  # noqa: E402

V = {"m32": 30, "m16": 32}


def ndiff(a, b):
    return int((a.contiguous().view(torch.int16) != b.contiguous().view(torch.int16)).sum())


def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    lib = load()
    res = {"device": torch.cuda.get_device_name(0), "small": [], "shapes": [], "grouped": []}

    def use(v):
        lib.ar_gemm_dw_config(V[v], -1)

    # ---- small problems: each variant against fp32 and against the other
    cases = [(256, 512, 256, False, False), (512, 1024, 2048, True, True), (1000, 512, 512, False, False), (96 + 37, 256, 256, False, False),
             (4096, 768, 768, False, False), (2048, 3072, 768, False, False)]
    for K, M, N, strided, accum in cases:
        if strided:
            by, bx = rnd((K, M + 512), 1), rnd((K, N + 256), 2, 0.05)
            dY, X = by[:, 256:256 + M], bx[:, 256:256 + N]
        else:
            dY, X = rnd((K, M), 1), rnd((K, N), 2, 0.05)
        ref = dY.float().t() @ X.float()
        outs = {}
        for v in V:
            use(v)
            for split in ([True, False, 2] if K >= 1024 else [True, False]):
                old = rnd((M, N), 3, 0.5)
                out = old.clone()
                ok = ops.gemm_dw(dY, X, out, accumulate=accum, split=split)
                want = (old.float() + ref) if accum else ref
                err = float(((out.float() - want).abs() / (want.abs() + 1.0)).max()) if ok else None
                outs[(v, str(split))] = out if ok else None
                res["small"].append(dict(K=K, M=M, N=N, strided=strided, accumulate=accum, variant=v, split=str(split), took=bool(ok), max_rel_err=err))
        for split in ("True", "False", "2"):
            a, b = outs.get(("m32", split)), outs.get(("m16", split))
            if a is not None and b is not None:
                res["small"].append(dict(K=K, M=M, N=N, split=split, m16_vs_m32_differing=ndiff(a, b), numel=a.numel()))
    for r in res["small"]:
        print(json.dumps(r), flush=True)

    # ---- a cut table on a small problem: both kernels must produce the same two-part sums
    K, M, N = 2048, 512, 1024
    dY, X = rnd((K, M), 4), rnd((K, N), 5, 0.05)
    tiles = (M // 256) * (N // 256)
    kcut = torch.zeros(tiles, dtype=torch.int32, device="cuda")
    kcut[1], kcut[3], kcut[6] = 32, 1024, 2016
    outs = {}
    for v in V:
        use(v)
        o = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        assert ops.gemm_dw_sk(dY, X, o, kcut)
        outs[v] = o
    want = torch.empty((M, N), dtype=torch.float32, device="cuda")
    for t in range(tiles):
        tm, tn = t // (N // 256), t % (N // 256)
        ys, xs = dY[:, tm * 256:(tm + 1) * 256].float(), X[:, tn * 256:(tn + 1) * 256].float()
        c = int(kcut[t])
        want[tm * 256:(tm + 1) * 256, tn * 256:(tn + 1) * 256] = ys.t() @ xs if c == 0 else (ys[:c].t() @ xs[:c]) + (ys[c:].t() @ xs[c:])
    rec = dict(check="cut_table", m16_vs_m32_differing=ndiff(outs["m16"], outs["m32"]),
               m16_max_abs_err=float((outs["m16"].float() - want).abs().max()), m32_max_abs_err=float((outs["m32"].float() - want).abs().max()))
    res["cut_table"] = rec
    print(json.dumps(rec), flush=True)

    # ---- Llama-3-8B shapes, K = 8 x 2048 tokens
    shapes = {"q_o": (4096, 4096), "down": (4096, 14336), "gate_up": (14336, 4096), "qkv_merged": (6144, 4096), "gate_up_merged": (28672, 4096)}
    if args.quick:
        shapes = {"q_o": shapes["q_o"], "gate_up": shapes["gate_up"]}
    K = 16384
    for name, (M, N) in shapes.items():
        dY, X = rnd((K, M), 11, 0.02), rnd((K, N), 12)
        flops = 2.0 * M * N * K
        lib_out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        torch.mm(dY.t(), X, out=lib_out)
        one, sk, fns = {}, {}, {"hipblaslt": lambda: torch.mm(dY.t(), X, out=lib_out)}
        st = streamk.find_on_device(dY, X) if M * N >= 4096 * 14336 and name != "gate_up_merged" else None
        kcut = None if st is None else st[1]
        for v in V:
            use(v)
            o = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            assert ops.gemm_dw(dY, X, o, split=False)
            one[v] = o
            o2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            ops.gemm_dw(dY, X, o2, split=False)
            assert ndiff(o, o2) == 0, "two launches, different bits"
            if kcut is not None:
                o3 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
                assert ops.gemm_dw_sk(dY, X, o3, kcut)
                sk[v] = o3

            def f1(v=v, o=torch.empty((M, N), dtype=torch.bfloat16, device="cuda")):
                use(v)
                ops.gemm_dw(dY, X, o, split=False)

            def f2(v=v, o=torch.empty((M, N), dtype=torch.bfloat16, device="cuda")):
                use(v)
                ops.gemm_dw(dY, X, o, split=True)

            fns[f"{v}_one_pass"] = f1
            fns[f"{v}_own_plan"] = f2
            if kcut is not None:
                def f3(v=v, o=torch.empty((M, N), dtype=torch.bfloat16, device="cuda")):
                    use(v)
                    ops.gemm_dw_sk(dY, X, o, kcut)
                fns[f"{v}_streamk"] = f3
        times = {k: [] for k in fns}
        for _ in range(args.rounds):
            for k, f in fns.items():
                times[k].append(timed(f, args.reps))
        rec = dict(shape=name, M=M, N=N, K=K, one_pass_m16_vs_m32_differing=ndiff(one["m16"], one["m32"]),
                   one_pass_m32_vs_library_differing=ndiff(one["m32"], lib_out), one_pass_m16_vs_library_differing=ndiff(one["m16"], lib_out),
                   numel=M * N, streamk_structure_found=kcut is not None)
        if kcut is not None:
            rec.update(streamk_m32_vs_library_differing=ndiff(sk["m32"], lib_out), streamk_m16_vs_library_differing=ndiff(sk["m16"], lib_out),
                       two_part_tiles=int((kcut > 0).sum()))
        for k, ts in times.items():
            ms = sorted(ts)[len(ts) // 2]
            rec[f"{k}_ms"] = round(ms, 4)
            rec[f"{k}_pflops"] = round(flops / ms / 1e12, 4)
        res["shapes"].append(rec)
        print(json.dumps(rec), flush=True)
        del dY, X, lib_out, one, sk, fns
        torch.cuda.empty_cache()

    # ---- grouped form at Mixtral-8x7B's expert shapes: 8 experts, 32768 routed rows, ragged counts
    if not args.quick:
        counts = [4100, 3900, 4500, 3700, 4096, 4200, 4000, 4272]
        R, E = sum(counts), len(counts)
        for name, (M, N) in {"w1_w3": (14336, 4096), "w2": (4096, 14336)}.items():
            dY, X = rnd((R, M), 21, 0.02), rnd((R, N), 22)
            row_off = torch.tensor([0] + torch.tensor(counts).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
            w_off = torch.arange(E, dtype=torch.int64, device="cuda") * (M * N)
            outs = {}
            fns = {}
            for v in V:
                use(v)
                W = torch.empty((E * M, N), dtype=torch.bfloat16, device="cuda")
                assert ops.gemm_dw_grouped(dY, X, W, row_off, w_off, N)
                outs[v] = W

                def f(v=v, W=W):
                    use(v)
                    ops.gemm_dw_grouped(dY, X, W, row_off, w_off, N)
                fns[v] = f
            ref0 = dY[:counts[0]].float().t() @ X[:counts[0]].float()
            rec = dict(grouped=name, M=M, N=N, rows=R, m16_vs_m32_differing=ndiff(outs["m16"], outs["m32"]),
                       m16_expert0_max_abs_err=float((outs["m16"][:M].float() - ref0).abs().max()), expert0_mean_abs=float(ref0.abs().mean()))
            times = {k: [] for k in fns}
            for _ in range(args.rounds):
                for k, f in fns.items():
                    times[k].append(timed(f, max(2, args.reps // 2)))
            for k, ts in times.items():
                ms = sorted(ts)[len(ts) // 2]
                rec[f"{k}_ms"] = round(ms, 4)
                rec[f"{k}_pflops"] = round(2.0 * M * N * R / ms / 1e12, 4)
            res["grouped"].append(rec)
            print(json.dumps(rec), flush=True)
            del dY, X, outs, fns
            torch.cuda.empty_cache()
    use("m32")
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
