"""Round 5: the NT forward / input-gradient GEMM (csrc/ar_gemm_nt.hip) and the grouped expert GEMMs, on the MI355X.

  * dense `ops.gemm_nt` vs the library (`torch.mm(A, B.t())`: hipBLASLt behind `F.linear`) -- bit for bit -- and vs an fp32 product;
    ragged M; 10 repeated launches per shape must return identical bits (race screen: the kernel keeps LDS-DMA in flight across
    barriers);
  * time: NT kernel vs library on Llama-3-8B's forward and dX shapes, interleaved rounds, random operands;
  * grouped `ops.gemm_nt_grouped` / `ops.gemm_dw_grouped` vs per-expert calls (bitwise vs this package's dense kernels, tolerance vs
    fp32), with empty groups and row counts that are not multiples of anything; time vs the Python loop of library GEMMs.

    python tools/gpu/r05_gemm_nt_probe.py --out gpurun_out/r05/gemm_nt_probe.json
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops  # noqa: E402


def rnd(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, device="cuda") * scale).to(torch.bfloat16)


def bits_diff(a, b):
    return int((a.contiguous().view(torch.int16) != b.contiguous().view(torch.int16)).sum())


def timeit(fn, rounds=5, inner=5):
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        e1.synchronize()
        best.append(e0.elapsed_time(e1) / inner)
    best.sort()
    return dict(ms_min=best[0], ms_median=best[len(best) // 2])


def dense_case(M, N, K, gen, repeat=10):
    A, B = rnd((M, K), gen), rnd((N, K), gen, 0.05)
    lib = torch.mm(A, B.t())
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ok = ops.gemm_nt(A, B, out)
    rec = dict(M=M, N=N, K=K, ran=bool(ok))
    if not ok:
        return rec
    torch.cuda.synchronize()
    rec["nan"] = int(torch.isnan(out.float()).sum())
    rec["bits_differ_from_library"] = bits_diff(out, lib)
    rows = min(M, 2048)
    ref = A[:rows].float() @ B.float().t()
    rec["max_abs_err_vs_fp32"] = float((out[:rows].float() - ref).abs().max())
    rec["library_max_abs_err_vs_fp32"] = float((lib[:rows].float() - ref).abs().max())
    rec["rms_ref"] = float(ref.pow(2).mean().sqrt())
    flaky = 0
    for _ in range(repeat):
        o2 = torch.empty_like(out)
        ops.gemm_nt(A, B, o2)
        flaky += int(bits_diff(o2, out) != 0)
    rec["repeat_runs_differing"] = flaky
    return rec


def time_case(name, M, N, K, gen):
    A, B = rnd((M, K), gen), rnd((N, K), gen, 0.05)
    out, lib = torch.empty((M, N), dtype=torch.bfloat16, device="cuda"), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * M * N * K
    for _ in range(3):
        ops.gemm_nt(A, B, out)
        torch.mm(A, B.t(), out=lib)
    res = dict(shape=name, M=M, N=N, K=K, equal_bits=bits_diff(out, lib) == 0)
    a, b = [], []
    for _ in range(4):          # interleaved rounds
        a.append(timeit(lambda: ops.gemm_nt(A, B, out), rounds=1, inner=10)["ms_min"])
        b.append(timeit(lambda: torch.mm(A, B.t(), out=lib), rounds=1, inner=10)["ms_min"])
    res.update(nt_ms=min(a), nt_ms_median=sorted(a)[len(a) // 2], lib_ms=min(b), lib_ms_median=sorted(b)[len(b) // 2])
    res.update(nt_pflops=fl / (res["nt_ms_median"] * 1e-3) / 1e15, lib_pflops=fl / (res["lib_ms_median"] * 1e-3) / 1e15)
    return res


def grouped_case(counts, N, K, gen, Mdw=None):
    """counts: rows per expert.  forward NT grouped vs per-expert dense kernel + library; dW grouped vs per-expert gemm_dw"""
    E, R = len(counts), sum(counts)
    A = rnd((R, K), gen)
    Wall = rnd((E * N, K), gen, 0.05)                    # the experts' [N, K] matrices, stacked (stride N*K elements)
    row_off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()), dtype=torch.int32, device="cuda")
    b_off = torch.tensor([e * N * K for e in range(E)], dtype=torch.int64, device="cuda")
    out = torch.full((R, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    ok = ops.gemm_nt_grouped(A, Wall, out, row_off, b_off, N, K)
    rec = dict(counts=counts, N=N, K=K, ran=bool(ok))
    if not ok:
        return rec
    torch.cuda.synchronize()
    rec["nan"] = int(torch.isnan(out.float()).sum())
    d_lib = d_dense = 0
    worst = 0.0
    s = 0
    for e, c in enumerate(counts):
        if c:
            We = Wall[e * N:(e + 1) * N]
            lib = torch.mm(A[s:s + c], We.t())
            d_lib += bits_diff(out[s:s + c], lib)
            mine = torch.empty_like(lib)
            if ops.gemm_nt(A[s:s + c], We, mine):
                d_dense += bits_diff(out[s:s + c], mine)
            ref = A[s:s + min(c, 512)].float() @ We.float().t()
            worst = max(worst, float((out[s:s + min(c, 512)].float() - ref).abs().max()))
        s += c
    rec.update(bits_differ_from_library=d_lib, bits_differ_from_dense_kernel=d_dense, max_abs_err_vs_fp32=worst)
    o2 = torch.empty_like(out)
    flaky = 0
    for _ in range(5):
        ops.gemm_nt_grouped(A, Wall, o2, row_off, b_off, N, K)
        flaky += int(bits_diff(o2, out) != 0)
    rec["repeat_runs_differing"] = flaky

    def loop_lib():
        s = 0
        for e, c in enumerate(counts):
            if c:
                torch.mm(A[s:s + c], Wall[e * N:(e + 1) * N].t(), out=o2[s:s + c])
            s += c

    t_g = timeit(lambda: ops.gemm_nt_grouped(A, Wall, out, row_off, b_off, N, K), rounds=3, inner=5)
    t_l = timeit(loop_lib, rounds=3, inner=5)
    fl = 2.0 * R * N * K
    rec.update(grouped_ms=t_g["ms_median"], library_loop_ms=t_l["ms_median"], grouped_pflops=fl / (t_g["ms_median"] * 1e-3) / 1e15,
               library_loop_pflops=fl / (t_l["ms_median"] * 1e-3) / 1e15)

    # ---- weight gradients: dW_e [Mdw, K] = dY[rows_e]^T @ A[rows_e]   (dY [R, Mdw]; "N" of the kernel = K here)
    Mdw = Mdw or N
    dY = rnd((R, Mdw), gen, 0.01)
    dW = torch.full((E * Mdw, K), float("nan"), dtype=torch.bfloat16, device="cuda")
    w_off = torch.tensor([e * Mdw * K for e in range(E)], dtype=torch.int64, device="cuda")
    ok = ops.gemm_dw_grouped(dY, A, dW, row_off, w_off, K)
    rec["dw_ran"] = bool(ok)
    if ok:
        torch.cuda.synchronize()
        rec["dw_nan"] = int(torch.isnan(dW.float()).sum())
        d_dense = 0
        worst = 0.0
        s = 0
        for e, c in enumerate(counts):
            mine = torch.zeros((Mdw, K), dtype=torch.bfloat16, device="cuda")
            if c:
                if c >= 96 and ops.gemm_dw(dY[s:s + c], A[s:s + c], mine, split=False):
                    d_dense += bits_diff(dW[e * Mdw:(e + 1) * Mdw], mine)
                ref = dY[s:s + c].float().t()[:256] @ A[s:s + c].float()
                worst = max(worst, float((dW[e * Mdw:e * Mdw + 256].float() - ref).abs().max()))
            else:
                d_dense += bits_diff(dW[e * Mdw:(e + 1) * Mdw], mine)       # an empty group writes zeros
            s += c
        rec.update(dw_bits_differ_from_dense_kernel=d_dense, dw_max_abs_err_vs_fp32=worst)
        dW2 = torch.empty_like(dW)

        def loop_dw():
            s = 0
            for e, c in enumerate(counts):
                if c:
                    torch.mm(dY[s:s + c].t(), A[s:s + c], out=dW2[e * Mdw:(e + 1) * Mdw])
                s += c

        t_g = timeit(lambda: ops.gemm_dw_grouped(dY, A, dW, row_off, w_off, K), rounds=3, inner=5)
        t_l = timeit(loop_dw, rounds=3, inner=5)
        fl = 2.0 * R * Mdw * K
        rec.update(dw_grouped_ms=t_g["ms_median"], dw_library_loop_ms=t_l["ms_median"], dw_grouped_pflops=fl / (t_g["ms_median"] * 1e-3) / 1e15,
                   dw_library_loop_pflops=fl / (t_l["ms_median"] * 1e-3) / 1e15)
    return rec


def ab_variants(gen):
    """A/B of where the LDS-DMA pieces are issued (between the MFMAs = 0 / at the end of the fragment-read part = 1), for the NT kernel
    and for the weight-gradient kernel: identical bits between the variants, race screen, interleaved timing rounds."""
    from auto_round_amd import _lib

    lib = _lib.load()
    out = dict(nt=[], dw=[])
    for name, M, N, K in [("o / q fwd", 16384, 4096, 4096), ("gate / up fwd", 16384, 14336, 4096), ("down fwd", 16384, 4096, 14336)]:
        A, B = rnd((M, K), gen), rnd((N, K), gen, 0.05)
        lib_out = torch.mm(A, B.t())
        outs, rec = {}, dict(shape=name, M=M, N=N, K=K)
        for v in (0, 1, 2):
            lib.ar_gemm_nt_config(v)
            o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            ops.gemm_nt(A, B, o)
            flaky = 0
            for _ in range(8):
                o2 = torch.empty_like(o)
                ops.gemm_nt(A, B, o2)
                flaky += int(bits_diff(o2, o) != 0)
            outs[v] = o
            rec[f"v{v}_bits_differ_from_library"] = bits_diff(o, lib_out)
            rec[f"v{v}_repeat_runs_differing"] = flaky
        t = {0: [], 1: [], 2: [], "lib": []}
        o = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        for _ in range(4):
            for v in (0, 1, 2):
                lib.ar_gemm_nt_config(v)
                t[v].append(timeit(lambda: ops.gemm_nt(A, B, o), rounds=1, inner=10)["ms_min"])
            t["lib"].append(timeit(lambda: torch.mm(A, B.t(), out=o), rounds=1, inner=10)["ms_min"])
        fl = 2.0 * M * N * K
        med = lambda x: sorted(x)[len(x) // 2]  # noqa: E731
        rec.update(v0_ms=med(t[0]), v1_ms=med(t[1]), v2_ms=med(t[2]), lib_ms=med(t["lib"]), v0_pflops=fl / med(t[0]) / 1e12, v1_pflops=fl / med(t[1]) / 1e12,
                   v2_pflops=fl / med(t[2]) / 1e12, lib_pflops=fl / med(t["lib"]) / 1e12)
        print(json.dumps(rec), flush=True)
        out["nt"].append(rec)
    lib.ar_gemm_nt_config(0)
    for name, M, N, K in [("o dW", 4096, 4096, 16384), ("qkv merged dW", 6144, 4096, 16384), ("down dW", 4096, 14336, 16384), ("gate dW", 14336, 4096, 16384),
                          ("gate|up merged dW", 28672, 4096, 16384)]:
        dY, X = rnd((K, M), gen, 0.01), rnd((K, N), gen)
        rec = dict(shape=name, M=M, N=N, K=K)
        outs = {}
        for v in (0, 1):
            lib.ar_gemm_dw_config(30 + v, -1)
            o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            ops.gemm_dw(dY, X, o, split=False)
            flaky = 0
            for _ in range(6):
                o2 = torch.empty_like(o)
                ops.gemm_dw(dY, X, o2, split=False)
                flaky += int(bits_diff(o2, o) != 0)
            outs[v] = o
            rec[f"v{v}_repeat_runs_differing"] = flaky
            rec[f"v{v}_nan"] = int(torch.isnan(o.float()).sum())
        rec["variants_bits_differ"] = bits_diff(outs[0], outs[1])
        t = {(0, False): [], (1, False): [], (0, True): [], (1, True): [], "lib": []}
        o = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        for _ in range(4):
            for v in (0, 1):
                lib.ar_gemm_dw_config(30 + v, -1)
                for sp in (False, True):
                    t[(v, sp)].append(timeit(lambda: ops.gemm_dw(dY, X, o, split=sp), rounds=1, inner=6)["ms_min"])
            t["lib"].append(timeit(lambda: torch.mm(dY.t(), X, out=o), rounds=1, inner=6)["ms_min"])
        fl = 2.0 * M * N * K
        med = lambda x: sorted(x)[len(x) // 2]  # noqa: E731
        rec.update(v0_one_pass_ms=med(t[(0, False)]), v1_one_pass_ms=med(t[(1, False)]), v0_own_plan_ms=med(t[(0, True)]), v1_own_plan_ms=med(t[(1, True)]),
                   lib_ms=med(t["lib"]), v0_one_pass_pflops=fl / med(t[(0, False)]) / 1e12, v1_one_pass_pflops=fl / med(t[(1, False)]) / 1e12,
                   v0_own_plan_pflops=fl / med(t[(0, True)]) / 1e12, v1_own_plan_pflops=fl / med(t[(1, True)]) / 1e12, lib_pflops=fl / med(t["lib"]) / 1e12)
        print(json.dumps(rec), flush=True)
        out["dw"].append(rec)
    lib.ar_gemm_dw_config(30, -1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--ab", action="store_true", help="only the A/B of the DMA-issue variants (NT and weight-gradient kernels) + the grouped cases")
    args = ap.parse_args()
    gen = torch.Generator(device="cuda").manual_seed(0)
    res = dict(device=torch.cuda.get_device_name(0), torch=torch.__version__, dense=[], timing=[], grouped=[])

    def flush():
        if args.out:
            with open(args.out, "w") as f:
                json.dump(res, f, indent=1)

    if args.ab:
        res["ab"] = ab_variants(gen)
        flush()
        v2_wins = sum(1 for r in res["ab"]["nt"] if r["v2_ms"] < r["v0_ms"] and r["v2_bits_differ_from_library"] == 0 and r["v2_repeat_runs_differing"] == 0)
        from auto_round_amd import _lib
        _lib.load().ar_gemm_nt_config(2 if v2_wins >= 2 else 0)
        res["grouped_variant"] = 2 if v2_wins >= 2 else 0
    for M, N, K in ([] if args.ab else [(256, 256, 128), (512, 512, 256), (1000, 256, 384), (4096, 1024, 4096), (16384, 4096, 4096), (16384, 4096, 14336)]):
        r = dense_case(M, N, K, gen)
        print(json.dumps(r), flush=True)
        res["dense"].append(r)
        flush()
    shapes = [("o / q fwd", 16384, 4096, 4096), ("k / v fwd", 16384, 1024, 4096), ("gate / up fwd", 16384, 14336, 4096), ("down fwd", 16384, 4096, 14336),
              ("qkv merged fwd", 16384, 6144, 4096), ("gate|up merged fwd", 16384, 28672, 4096), ("dX down (via W^T)", 16384, 14336, 4096),
              ("dX gate (via W^T)", 16384, 4096, 14336)]
    for name, M, N, K in ([] if args.ab else (shapes[:3] if args.quick else shapes)):
        r = time_case(name, M, N, K, gen)
        print(json.dumps(r), flush=True)
        res["timing"].append(r)
        flush()
    # Mixtral: 16384 tokens x top-2 = 32768 routed rows over 8 experts
    cases = [([300, 0, 1, 255, 257, 512, 100, 4096 - 1425], 512, 256, None),
             ([4100, 3900, 4300, 4096, 3800, 4188, 4200, 4184], 4096, 14336, 1024),          # down-projection forward (K = ffn), dW sampled at M = 1024
             ([4100, 3900, 4300, 4096, 3800, 4188, 4200, 4184], 28672, 4096, 2048)]          # merged gate | up forward
    for counts, N, K, Mdw in (cases[:2] if args.quick else cases):
        r = grouped_case(counts, N, K, gen, Mdw)
        print(json.dumps(r), flush=True)
        res["grouped"].append(r)
        flush()


if __name__ == "__main__":
    main()
